"""Per-shape timing of ff_conv2d on the GPU box (tuning aid; not part of the product or the tests).
usage: python tools/gemm_bench.py [tile_hint ...]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from isr2_amd import ops

SHAPES = [  # name, B,H,W,Cin,Cout,k
    ("hat qkv 180->540", 1, 256, 256, 180, 540, 1), ("hat proj 180->180", 1, 256, 256, 180, 180, 1),
    ("hat fc1 180->360", 1, 256, 256, 180, 360, 1), ("hat fc2 360->180", 1, 256, 256, 360, 180, 1),
    ("dat fc1 180->720", 1, 256, 256, 180, 720, 1), ("cab 3x3 180->60", 1, 256, 256, 180, 60, 3),
    ("cab 3x3 60->180", 1, 256, 256, 60, 180, 3), ("rhag 3x3 180->180", 1, 256, 256, 180, 180, 3),
    ("naf c1 64->128 @1024", 1, 1024, 1024, 64, 128, 1), ("naf c3 64->64 @1024", 1, 1024, 1024, 64, 64, 1),
    ("naf c1 128->256 @512", 1, 512, 512, 128, 256, 1), ("naf c1 256->512 @256", 1, 256, 256, 256, 512, 1),
    ("naf c1 512->1024 @128", 1, 128, 128, 512, 1024, 1), ("naf c1 1024->2048 @64", 1, 64, 64, 1024, 2048, 1),
    ("naf c3 1024->1024 @64", 1, 64, 64, 1024, 1024, 1), ("refine 3x3 64->64 @1024", 1, 1024, 1024, 64, 64, 3),
    ("dat 3x3 180->180 b", 1, 256, 256, 180, 180, 3), ("last 3x3 64->3 @1024", 1, 1024, 1024, 64, 3, 3),
    ("naf 3x3 64->64 @512", 1, 512, 512, 64, 64, 3), ("edge 3x3 64->32 @1024", 1, 1024, 1024, 64, 32, 3),
    ("hier 3x3 73->64 @1024", 1, 1024, 1024, 73, 64, 3), ("up 3x3 64->256 @256", 1, 256, 256, 64, 256, 3),
    ("up 3x3 64->256 @512", 1, 512, 512, 64, 256, 3), ("conv 3->64 @1024", 1, 1024, 1024, 3, 64, 3),
]

def main():
    args = sys.argv[1:]
    if args and args[0] in ops.GEMM_MODES:
        ops.set_gemm_mode(args.pop(0))
    print("mode", ops.gemm_mode())
    if args and args[0] == "nohalo":
        args.pop(0); ops.set_halo(False)
    hints = [int(a) for a in args] or [0]
    dev = torch.device("cuda:0")
    for name, B, H, W, Ci, Co, k in SHAPES:
        x = torch.randn(B, H, W, Ci, device=dev)
        w = torch.randn(Co, k * k * Ci, device=dev) * 0.05
        b = torch.randn(Co, device=dev)
        row = []
        for hint in hints:
            try:
                for _ in range(2):
                    ops.conv2d(x, w, b, ksize=(k, k), pad=(k // 2, k // 2), tile_hint=hint)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 10
                e0.record()
                for _ in range(n):
                    ops.conv2d(x, w, b, ksize=(k, k), pad=(k // 2, k // 2), tile_hint=hint)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n
                tf = 2.0 * B * H * W * Co * k * k * Ci / ms / 1e9
                row.append(f"h{hint}: {ms*1e3:8.1f} us {tf:6.1f} TF")
            except Exception as e:
                row.append(f"h{hint}: ERR {e}")
        print(f"{name:28s} " + " | ".join(row), flush=True)

if __name__ == "__main__":
    main()
