"""Single-kernel driver for rocprofv3 counter passes: runs ff_win_attn_fused (HAT geometry) a few times.
    python3 tools/wf_prof.py [shift] [mode = bf16 | bf16x3]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isr2_amd import ops  # noqa: E402
from isr2_amd.prep import pack_win_attn, pack_win_rel  # noqa: E402

dev = torch.device("cuda:0")
H = W = 256
C, heads, d, ws = 180, 6, 30, 16
shift = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ops.set_gemm_mode(sys.argv[2] if len(sys.argv) > 2 else "bf16")
g = torch.Generator().manual_seed(0)
x = ops.empty_rows((1, H, W, C), dev)
x.copy_(torch.randn(1, H, W, C, generator=g).to(dev))
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
wqkv = (torch.randn(3 * C, C, generator=g) / math.sqrt(C)).to(dev)
pk = pack_win_attn(wqkv, torch.zeros(3 * C, device=dev), heads, d, d ** -0.5)
relp = pack_win_rel((torch.randn(heads, (2 * ws - 1) ** 2, generator=g) * 0.5).to(dev), ws, ws)
out = ops.empty_rows((1, H, W, C), dev)
for _ in range(5):
    ops.win_attn_fused(x, out, pk, relp, gamma=gam, beta=bet, H=H, W=W, Hp=H, Wp=W, win=(ws, ws), shift=(shift, shift),
                       use_mask=shift > 0, want_xn=True)
torch.cuda.synchronize()
