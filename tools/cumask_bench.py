"""Experiment: two tiles in flight on DISJOINT halves of the chip (hipExtStreamCreateWithCUMask), so that kernels of the two lanes
run concurrently instead of queueing behind each other (every big kernel of the path fills the whole chip with one 8-wave workgroup
per CU: two lanes on ordinary streams overlap only at kernel tails).   python tools/cumask_bench.py [mode]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["FF_STREAMS"] = "0"            # one launch stream per forward: the lane's graph is a linear chain on the lane's queue
import torch
import bench
from isr2_amd import ops
from isr2_amd.model import FreqFusionHIP
from isr2_amd.weights import synth_state_dict

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def run(model, lr, streams, steps=20):
    graphs = []
    for s in streams:
        with torch.cuda.stream(s):
            model(lr)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = model(lr)
        graphs.append((g, out))
    torch.cuda.synchronize()
    for i in range(4):
        with torch.cuda.stream(streams[i % len(streams)]):
            graphs[i % len(streams)][0].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % len(streams)]):
            graphs[i % len(streams)][0].replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    ops.set_gemm_mode(mode)
    m = FreqFusionHIP(synth_state_dict(1234), "cuda:0")
    lr = bench.make_tile(100).cuda()
    m(lr)
    full = [0xFFFFFFFF] * 8
    print(f"mode {mode}, single-stream forwards", flush=True)
    print(f"one lane, whole chip                : {run(m, lr, [torch.cuda.Stream()]):7.2f} ms per tile", flush=True)
    print(f"two lanes, ordinary streams         : {run(m, lr, [torch.cuda.Stream(), torch.cuda.Stream()]):7.2f}", flush=True)
    lo, hi = [0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4
    print(f"two lanes, mask low / high 128 bits : {run(m, lr, [masked_stream(lo), masked_stream(hi)]):7.2f}", flush=True)
    ev, od = [0x55555555] * 8, [0xAAAAAAAA] * 8
    print(f"two lanes, mask even / odd bits     : {run(m, lr, [masked_stream(ev), masked_stream(od)]):7.2f}", flush=True)
    a, b = [0x0F0F0F0F] * 8, [0xF0F0F0F0] * 8
    print(f"two lanes, mask nibbles             : {run(m, lr, [masked_stream(a), masked_stream(b)]):7.2f}", flush=True)
    print(f"two lanes, both full masks          : {run(m, lr, [masked_stream(full), masked_stream(full)]):7.2f}", flush=True)
    q = [[0xFFFFFFFF if i // 2 == k else 0 for i in range(8)] for k in range(4)]
    print(f"four lanes, 64-bit quarters         : {run(m, lr, [masked_stream(w) for w in q]):7.2f}", flush=True)


if __name__ == "__main__":
    main()
