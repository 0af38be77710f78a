"""Single-shape driver for rocprofv3 counter passes on the 1x1 (linear) path.  python3 tools/gemm_prof.py M K N [hint] [res]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isr2_amd import ops  # noqa: E402

M, K, N = (int(a) for a in sys.argv[1:4])
hint = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
x = torch.randn(1, 1, M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.05
b = torch.randn(N, device=dev)
res = torch.randn(1, 1, M, N, device=dev) if len(sys.argv) > 5 else None
for _ in range(5):
    ops.conv2d(x, w, b, ksize=(1, 1), pad=(0, 0), res=res, tile_hint=hint)
torch.cuda.synchronize()
