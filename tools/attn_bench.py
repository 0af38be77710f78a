import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from isr2_amd import ops
dev = torch.device("cuda:0")
H = W = 256; C = 180; heads = 6; d = 30
qkv = torch.randn(1, H, W, 3 * C, device=dev)
bias = torch.randn(heads, 256, 256, device=dev)
out = torch.empty(1, H, W, C, device=dev)
shift = int(sys.argv[1]) if len(sys.argv) > 1 else 0
def run():
    ops.window_attn(qkv, out, bias, q_off=0, k_off=C, v_off=2 * C, o_off=0, H=H, W=W, Hp=H, Wp=W, win=(16, 16), kwin=(16, 16),
                    shift=(shift, shift), use_mask=shift > 0, heads=heads, d=d, scale=d ** -0.5)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("attn shift", shift, "us per call:", round(e0.elapsed_time(e1) / 20 * 1e3, 1))
