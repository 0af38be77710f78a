"""Single-kernel driver for rocprofv3 counter passes: runs ff_token_projmlp (HAT geometry, 65 536 tokens) a few times.
    python3 tools/pm_prof.py [mode = bf16 | bf16x3]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isr2_amd import ops  # noqa: E402
from isr2_amd.prep import pack_token_projmlp  # noqa: E402

dev = torch.device("cuda:0")
ops.set_gemm_mode(sys.argv[1] if len(sys.argv) > 1 else "bf16")
M, C, Hd = 65536, 180, 360
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)  # noqa: E731
att, x, c2 = [torch.empty(M, 192, device=dev)[:, :C] for _ in range(3)]
for t in (att, x, c2):
    t.copy_(r(M, C))
pk = pack_token_projmlp(r(C, C, sc=1 / math.sqrt(C)), r(C, sc=0.1), r(Hd, C, sc=1 / math.sqrt(C)), r(Hd, sc=0.1), r(C, Hd, sc=1 / math.sqrt(Hd)), r(C, sc=0.1))
gam, bet, scale = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.full((C,), 0.01, device=dev)
for _ in range(5):
    ops.token_projmlp(att, x, pk, gam, bet, c2=c2, c2_scale=scale)
torch.cuda.synchronize()
