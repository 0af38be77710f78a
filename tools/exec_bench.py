"""C-level executor on the bench tile: export a plan for 256x256, replay it through ff_forward (eager and captured in a HIP graph)
and compare with the Python host's three-stream graph.  python tools/exec_bench.py"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from isr2_amd import plan  # noqa: E402
from isr2_amd.model import FreqFusionHIP  # noqa: E402
from isr2_amd.weights import synth_state_dict  # noqa: E402


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


dev = torch.device("cuda:0")
model = FreqFusionHIP(synth_state_dict(1234), dev)
lr = bench.make_tile(100).to(dev)
with tempfile.TemporaryDirectory() as d:
    t0 = time.perf_counter()
    info = plan.export_plan(model, lr, os.path.join(d, "ff256"))
    print(f"export: {time.perf_counter() - t0:.1f} s", info)
    nat = plan.NativeModel(os.path.join(d, "ff256.ffplan"), os.path.join(d, "ff256.ffwts"))
ref = model(lr)
out = nat(lr)
torch.cuda.synchronize()
print("bit-equal to the Python-sequenced forward:", bool(torch.equal(out, ref)))
print(f"ff_forward, eager launches from C : {timeit(lambda: nat(lr)):.2f} ms")
sin, sout = lr.clone(), torch.empty_like(out)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    nat.L.ff_forward(nat.h, sin.data_ptr(), 1, 256, 256, sout.data_ptr(), torch.cuda.current_stream().cuda_stream)
print(f"ff_forward captured in a HIP graph : {timeit(g.replay):.2f} ms ({info['streams']} streams)")
with tempfile.TemporaryDirectory() as d:
    info1 = plan.export_plan(model, lr, os.path.join(d, "ff256s"), multi_stream=False)
    nat1 = plan.NativeModel(os.path.join(d, "ff256s.ffplan"), os.path.join(d, "ff256s.ffwts"))
print("single-stream plan:", info1, "bit-equal:", bool(torch.equal(nat1(lr), ref)))
print(f"ff_forward, single-stream plan     : {timeit(lambda: nat1(lr)):.2f} ms")
nat1.close()
print(f"Python host, model.graphed         : {timeit(lambda: model.graphed(lr)):.2f} ms (three streams)")
model.multi_stream = False
model._graphs.clear()
print(f"Python host, single-stream graph   : {timeit(lambda: model.graphed(lr)):.2f} ms")
nat.close()
