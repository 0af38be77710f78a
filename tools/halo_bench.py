"""Per-shape timing of the LDS-resident 3x3 convolution in a given contraction mode (tuning aid).  usage: halo_bench.py [mode]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from isr2_amd import ops
from isr2_amd.prep import pack_conv

SHAPES = [("cab 180->60 @256", 256, 256, 180, 60), ("cab 60->180 @256", 256, 256, 60, 180), ("rhag 180->180 @256", 256, 256, 180, 180),
          ("refine 64->64 @1024", 1024, 1024, 64, 64), ("hier 76->64 @1024", 1024, 1024, 76, 64), ("edge 64->32 @1024", 1024, 1024, 64, 32),
          ("naf 64->64 @512", 512, 512, 64, 64)]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    ops.set_gemm_mode(mode)
    dev = torch.device("cuda:0")
    for name, H, W, ci, co in SHAPES:
        x = torch.randn(1, H, W, ci, device=dev)
        w = pack_conv(torch.randn(co, ci, 3, 3, device=dev) * 0.05)
        b = torch.randn(co, device=dev)
        for _ in range(3):
            ops.conv2d(x, w, b, ksize=(3, 3), pad=(1, 1), act="gelu")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.conv2d(x, w, b, ksize=(3, 3), pad=(1, 1), act="gelu")
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / n
        gb = 4.0 * H * W * (ci + co) / 1e9
        print(f"{mode:7s} {name:24s} {us:8.1f} us  {2.0 * H * W * co * 9 * ci / us / 1e6:7.1f} TF  min-traffic {gb * 1e3:6.1f} MB = {gb / 4e-6 / 1e6 * 1e3:5.1f} us @4TB/s", flush=True)


if __name__ == "__main__":
    main()
