"""Per-launch timeline of one block of an expert (kernel class, microseconds), from HIP events around every launch.
python tools/block_timeline.py dat|hat|naf"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from isr2_amd import ops  # noqa: E402
from isr2_amd.model import FreqFusionHIP  # noqa: E402
from isr2_amd.weights import synth_state_dict  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "dat"
dev = torch.device("cuda:0")
model = FreqFusionHIP(synth_state_dict(1234), dev)
model.multi_stream = False
lr = bench.make_tile(100).to(dev)
exp = {"dat": model.dat, "hat": model.hat, "naf": model.nafnet}[which]
for _ in range(2):
    exp.forward(lr)
torch.cuda.synchronize()
with ops.profile() as prof:
    exp.forward(lr)
recs = prof.records()
tot = sum(r[1] for r in recs)
print(f"{which}: {len(recs)} launches, {tot:.2f} ms")
lo, hi = (8, 8 + 40) if which != "naf" else (0, 60)
for i, (name, ms, fl, by) in enumerate(recs[:int(os.environ.get("FF_TL_N", "120"))]):
    print(f"{i:4d} {name:18s} {ms * 1e3:8.1f} us  {fl / max(ms, 1e-9) / 1e9:7.1f} TF  {by / max(ms, 1e-9) / 1e6:7.0f} GB/s")
