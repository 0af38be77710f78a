import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, math
from isr2_amd import ops
from isr2_amd.prep import pack_token_linear
dev = torch.device("cuda:0")
M, C = 65536, 180
N = int(sys.argv[1]) if len(sys.argv) > 1 else 540
x = torch.randn(M, C, device=dev)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
pk = pack_token_linear(torch.randn(N, C, device=dev) / math.sqrt(C), torch.zeros(N, device=dev))
for _ in range(3): ops.token_linear(x, pk, gamma=g, beta=b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.token_linear(x, pk, gamma=g, beta=b)
e1.record(); torch.cuda.synchronize()
print("N", N, "dbg", os.environ.get("FF_TM_DBG", "0"), "us per call:", round(e0.elapsed_time(e1) / 20 * 1e3, 1))
