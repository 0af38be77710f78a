"""Per-launch timeline of the fusion stack (everything after the experts) on the bench tile.  python tools/fusion_timeline.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from isr2_amd import ops  # noqa: E402
from isr2_amd.model import FreqFusionHIP  # noqa: E402
from isr2_amd.weights import synth_state_dict  # noqa: E402

dev = torch.device("cuda:0")
model = FreqFusionHIP(synth_state_dict(1234), dev)
model.multi_stream = False
lr = bench.make_tile(100).to(dev)
ex = model.experts(lr)
for _ in range(2):
    model.fusion.forward(lr, ex)
torch.cuda.synchronize()
with ops.profile() as prof:
    model.fusion.forward(lr, ex)
recs = prof.records()
print(f"fusion: {len(recs)} launches, {sum(r[1] for r in recs):.2f} ms")
for i, (name, ms, fl, by) in enumerate(recs):
    print(f"{i:4d} {name:18s} {ms * 1e3:8.1f} us  {fl / max(ms, 1e-9) / 1e9:7.1f} TF  {by / max(ms, 1e-9) / 1e6:7.0f} GB/s")
