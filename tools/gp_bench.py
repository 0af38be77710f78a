"""ff_gemm_planes against ff_conv2d_bf16s (ops.linear) on NAFNet's deep-level 1x1 shapes.  python tools/gp_bench.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isr2_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dev = torch.device("cuda:0")
for name, M, K, N in [("c1 256->512 @256", 65536, 256, 512), ("c3 256->256 @256", 65536, 256, 256), ("c1 512->1024 @128", 16384, 512, 1024),
                      ("c3 512->512 @128", 16384, 512, 512), ("c1 1024->2048 @64", 4096, 1024, 2048), ("c3 1024->1024 @64", 4096, 1024, 1024),
                      ("dat fc2 360->180", 65536, 360, 180), ("hat-like 192->192", 65536, 192, 192)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / math.sqrt(K)
    b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    t_lin = timeit(lambda: ops.linear(x, w, b, res=res))
    planes = ops.split_rows(x)
    t_split = timeit(lambda: ops.split_rows(x))
    t_gp = timeit(lambda: ops.gemm_planes(planes, w, b, res=res))
    fl = 2.0 * M * N * K
    print(f"{name:22s} linear {t_lin:7.1f} us {fl / t_lin / 1e6:6.1f} TF | planes GEMM {t_gp:7.1f} us {fl / t_gp / 1e6:6.1f} TF | split_rows {t_split:6.1f} us", flush=True)
