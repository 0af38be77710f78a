"""Isolated timing of HAT's OCAB attention stage (persistent kernel vs the two-stage kernel), 256 x 256 tokens.  python tools/ocab_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from isr2_amd import ops
from isr2_amd.prep import pack_rel_overlap
dev = torch.device("cuda:0")
ops.set_gemm_mode("bf16")
H = W = 256; C, heads, d, ws, ows = 180, 6, 30, 16, 24
qkv = torch.randn(1, H, W, 3 * C, device=dev)
table = torch.randn((ws + ows - 1) ** 2, heads, device=dev) * 0.5
rel = pack_rel_overlap(table, ws, ows)
out = ops.empty_rows((1, H, W, C), dev)
f = lambda: ops.ocab_attn(qkv, out, rel, q_off=0, k_off=C, v_off=2 * C, H=H, W=W, heads=heads, d=d, ws=ws, ows=ows, scale=d ** -0.5)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("ocab_attn persistent: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
