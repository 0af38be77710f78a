"""Image-sharded multi-GPU runner of the plugin (BASELINE config 4; the reference's pattern is eval.py:162-170: the sorted
file list is cut into contiguous shards, one worker process per GPU).

    python -m isr2_amd.run_sharded --gpus 8 --input LR_DIR --output SR_DIR [--model_dir fusion.pth]

The parent touches no GPU: it starts one fresh child per GPU, each of which calls the plugin's
`main(model_dir, input_path, output_path, device)` with RANK / WORLD_SIZE set.  Inside, rank 0 reads the checkpoints and the
frozen weights travel in ONE RCCL broadcast; every rank then processes its own shard of whole images and writes its own
PNGs.  There is no per-image collective; one tiny all_gather of (count, seconds) closes the run.
"""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def worker(args) -> None:
    import torch
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from models.team29_FreqFusion import main as plugin_main
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n = max(torch.cuda.device_count(), 1)
    plugin_main(model_dir=args.model_dir, input_path=args.input, output_path=args.output, device=torch.device("cuda", local % n))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser("isr2_amd.run_sharded")
    ap.add_argument("--gpus", type=int, default=0, help="ranks to start (default: every visible GPU)")
    ap.add_argument("--input", required=True)
    ap.add_argument("--output", required=True)
    ap.add_argument("--model_dir", default=os.path.join("checkpoints", "phase5_single_gpu", "championship_sr_phase5_single_gpu",
                                                        "best_epoch0050_psnr30.05.pth"))
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    if args.worker or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        worker(args)
        return 0
    from .parallel import spawn_ranks, visible_gpu_count
    ndev = visible_gpu_count()                       # sysfs / visibility lists only: no HIP call in the launcher parent
    n = args.gpus or ndev
    if n < 1:
        print("run_sharded: no GPU visible (pass --gpus N to start N ranks anyway)", file=sys.stderr)
        return 2
    if 0 <= ndev < n and os.environ.get("FF_DIST_BACKEND", "nccl") == "nccl":
        print(f"run_sharded: --gpus {n} but only {ndev} GPU(s) visible (one rank per GPU over RCCL)", file=sys.stderr)
        return 2
    # fail in the parent, before any rank exists, on what would otherwise kill rank 0 alone and leave its peers in the broadcast
    if not os.path.isdir(args.input):
        print(f"run_sharded: --input {args.input!r} is not a directory", file=sys.stderr)
        return 2
    if not os.path.isfile(args.model_dir) and os.environ.get("FF_ALLOW_SYNTH", "0") != "1":
        print(f"run_sharded: fusion checkpoint {os.path.abspath(args.model_dir)!r} not found "
              "(--model_dir is a FILE path, relative to the current directory; FF_ALLOW_SYNTH=1 runs on seeded synthetic weights)", file=sys.stderr)
        return 2
    if n == 1:
        worker(args)
        return 0
    cmd = [sys.executable, "-m", "isr2_amd.run_sharded", "--worker", "--gpus", str(n), "--input", args.input, "--output", args.output,
           "--model_dir", args.model_dir]
    env = {"PYTHONPATH": ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")}
    return spawn_ranks(cmd, n, env)


if __name__ == "__main__":
    sys.exit(main())
