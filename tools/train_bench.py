"""Time the fusion-only training step (BASELINE config 5: batch 16 x 64 x 64 LR, experts precomputed) on one MI355X.
    python tools/train_bench.py [--batch 16] [--size 64] [--steps 5] [--mode bf16x3|f32]
Under rocprofv3 --kernel-trace --stats the per-kernel table of a step comes out of the same command."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", default="f32")
    a = ap.parse_args()
    from train_inputs import make_train_batch
    from isr2_amd import ops
    from isr2_amd.train import FusionTrainer
    from isr2_amd.weights import synth_state_dict
    ops.set_gemm_mode(a.mode)
    d = {k: torch.from_numpy(v).cuda() for k, v in make_train_batch(503, a.batch, a.size, a.size).items()}
    outs = {k: d["out_" + k] for k in ("hat", "dat", "nafnet")}
    feats = {k: d["feat_" + k] for k in ("hat", "dat", "nafnet")}
    tr = FusionTrainer(synth_state_dict(1234, parts=("fusion", "collab")), "cuda:0", gemm=a.mode)
    for _ in range(a.warmup):
        tr.step(d["lr"], d["hr"], outs, feats)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.step(d["lr"], d["hr"], outs, feats)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    print(f"train step B={a.batch} {a.size}x{a.size} mode={a.mode}: {ms:.2f} ms/step, loss {float(loss):.6f}, "
          f"peak HBM {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)


if __name__ == "__main__":
    main()
