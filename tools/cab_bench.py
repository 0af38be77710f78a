"""HAT's CAB (conv3x3 180->60, GELU, conv3x3 60->180 + pool) at 256x256: two launches vs the fused kernel, plain bf16 (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from isr2_amd import ops
from isr2_amd.prep import pack_conv

ops.set_gemm_mode("bf16")
dev = torch.device("cuda:0")
x = torch.empty(1, 256, 256, 192, device=dev)[..., :180]
x.copy_(torch.randn(1, 256, 256, 180, device=dev))
w1, b1 = pack_conv(torch.randn(60, 180, 3, 3, device=dev) * 0.03), torch.randn(60, device=dev)
w2, b2 = pack_conv(torch.randn(180, 60, 3, 3, device=dev) * 0.05), torch.randn(180, device=dev)


def two():
    c1 = ops.conv2d(x, w1, b1, ksize=(3, 3), pad=(1, 1), act="gelu")
    return ops.conv2d(c1, w2, b2, ksize=(3, 3), pad=(1, 1), want_pool=True)


for name, fn in (("two launches", two), ("fused", lambda: ops.cab_fused(x, w1, b1, w2, b2))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:14s} {1e3 * e0.elapsed_time(e1) / 20:7.1f} us", flush=True)
