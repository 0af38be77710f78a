import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isr2_amd import ops
dev = torch.device("cuda:0")
part = torch.randn(256, 192, device=dev); W1 = torch.randn(6, 180, device=dev); b1 = torch.zeros(6, device=dev); W2 = torch.randn(180, 6, device=dev); b2 = torch.zeros(180, device=dev)
part = part[:, :180] if False else part
pp = ops.PoolPartials(part, 180, 1.0 / 65536)
for fused in (True, False):
    f = (lambda: ops.vec_mlp(pp, W1, b1, "relu", W2, b2, "sigmoid")) if fused else (lambda: ops.vec_mlp(pp.mean(), W1, b1, "relu", W2, b2, "sigmoid"))
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize()
    print("fused" if fused else "two launches", e0.elapsed_time(e1) / 200 * 1e3, "us")
