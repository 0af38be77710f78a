"""Single-shape driver for rocprofv3 counter passes on the NAFBlock front kernel.  python3 tools/naf_front_prof.py [C] [HW]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isr2_amd import ops  # noqa: E402
from isr2_amd.prep import pack_token_linear, pack_dw  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
x = torch.randn(1, hw, hw, C, device=dev)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
pk = pack_token_linear(torch.randn(2 * C, C, device=dev) / math.sqrt(C), torch.randn(2 * C, device=dev) * 0.1)
w2, b2 = pack_dw(torch.randn(2 * C, 1, 3, 3, device=dev) * 0.3), torch.randn(2 * C, device=dev) * 0.1
for _ in range(5):
    ops.naf_front(x, pk, g, b, w2, b2)
torch.cuda.synchronize()
