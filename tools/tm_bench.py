import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, math
from isr2_amd import ops
from isr2_amd.prep import pack_token_mlp
dev = torch.device("cuda:0")
M, C, Hd = 65536, 180, 360
x = torch.randn(M, C, device=dev)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
pk = pack_token_mlp(torch.randn(Hd, C, device=dev) / math.sqrt(C), torch.zeros(Hd, device=dev), torch.randn(C, Hd, device=dev) / math.sqrt(Hd), torch.zeros(C, device=dev))
for _ in range(3): ops.token_mlp(x, g, b, pk)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.token_mlp(x, g, b, pk)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("FF_TM_DBG", "0"), "us per call:", e0.elapsed_time(e1) / 20 * 1e3)
