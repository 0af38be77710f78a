"""Per-launch table of one forward (single stream, HIP events around every wrapper call), grouped by (wrapper, algorithmic FLOPs,
algorithmic bytes) = by shape: launches, total ms, average us, algorithmic GB/s and TFLOP/s, and the time at 4 TB/s / peak MFMA.
    python tools/launch_table.py [mode] [top]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from isr2_amd import ops
from isr2_amd.model import FreqFusionHIP
from isr2_amd.weights import synth_state_dict


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    ops.set_gemm_mode(mode)
    m = FreqFusionHIP(synth_state_dict(1234), "cuda:0")
    m.multi_stream = False
    lr = bench.make_tile(100).cuda()
    for _ in range(2):
        m(lr)
    agg = {}
    reps = 3
    for _ in range(reps):
        with ops.profile() as prof:
            m(lr)
        for name, ms, fl, by in prof.records():
            a = agg.setdefault((name, round(fl / 1e6), round(by / 1e4)), [0, 0.0, fl, by])
            a[0] += 1
            a[1] += ms
    tot = sum(a[1] for a in agg.values()) / reps
    print(f"mode {mode}: {tot:.2f} ms per forward in {sum(a[0] for a in agg.values()) // reps} wrapper calls")
    print(f"{'wrapper':22s} {'n':>4s} {'ms/fwd':>7s} {'avg us':>8s} {'GFLOP':>8s} {'MB':>8s} {'GB/s':>7s} {'TF':>6s} {'us@4TB/s':>9s}")
    for (name, _, _), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        n, ms, fl, by = a[0] / reps, a[1] / reps, a[2], a[3]
        us = 1e3 * ms / n
        print(f"{name:22s} {n:4.0f} {ms:7.2f} {us:8.1f} {fl / 1e9:8.2f} {by / 1e6:8.1f} {by / us / 1e3:7.0f} {fl / us / 1e6:6.1f} {by / 4e6:9.1f}")


if __name__ == "__main__":
    main()
