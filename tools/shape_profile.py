"""Per-shape time of every ff_conv2d / linear call of one forward (eager, single stream, HIP events): tuning aid."""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from isr2_amd import ops
from isr2_amd.weights import synth_state_dict
from isr2_amd.model import FreqFusionHIP
dev = torch.device("cuda:0")
m = FreqFusionHIP(synth_state_dict(1234), dev); m.multi_stream = False
lr = torch.from_numpy(np.random.default_rng(2).random((1, 3, 256, 256), dtype=np.float32)).to(dev)
m(lr); m(lr); torch.cuda.synchronize()
rec = []
def wrap(name, fn):
    def w(x, wt, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(x, wt, *a, **k); e1.record()
        ks = k.get("ksize", (1, 1))
        rec.append((name, tuple(x.shape[-3:]) if x.dim() == 4 else (x.numel() // x.shape[-1], x.shape[-1]), tuple(wt.shape), ks, e0, e1))
        return r
    return w
import isr2_amd.experts as E, isr2_amd.fusion as Fz
orig_c, orig_l = ops.conv2d, ops.linear
ops.conv2d, ops.linear = wrap("conv2d", orig_c), wrap("linear", orig_l)
m(lr); torch.cuda.synchronize()
ops.conv2d, ops.linear = orig_c, orig_l
agg = collections.defaultdict(lambda: [0, 0.0])
for name, xs, ws, ks, e0, e1 in rec:
    a = agg[(name, xs, ws, ks)]; a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print("total ms", round(tot, 2), "calls", len(rec))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{v[1]:7.2f} ms  x{v[0]:3d}  {1e3 * v[1] / v[0]:7.1f} us  {k}")
