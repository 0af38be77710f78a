"""C-level executor (ff_create / ff_upload / ff_finalize / ff_forward / ff_destroy).  CPU: the lifetime packing
of the plan's workspace and the plan loader's validation (no GPU call is made for a rejected file).  GPU: a plan exported
from the Python host replays through the C API bit for bit."""
import ctypes
import os
import struct

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def test_workspace_packing_never_aliases_live_buffers():
    from isr2_amd.plan import _pack_offsets
    rng = np.random.default_rng(0)
    work = []
    for _ in range(300):
        f = int(rng.integers(0, 1000))
        work.append([int(rng.integers(1, 5000)) * 16, f, f + int(rng.integers(0, 40))])
    work.append([4096, -1, -1])                                   # allocated but never referenced: takes no space
    offs, total = _pack_offsets(work)
    live = [(o, o + (w[0] + 255) // 256 * 256, w[1], w[2]) for o, w in zip(offs, work) if w[1] >= 0]
    for i, (o1, e1, f1, l1) in enumerate(live):
        assert o1 % 256 == 0 and e1 <= total
        for o2, e2, f2, l2 in live[i + 1:]:
            if not (l1 < f2 or l2 < f1):                          # lifetimes overlap -> bytes must not
                assert e1 <= o2 or e2 <= o1
    assert total < sum(w[0] for w in work)                        # reuse happened


def test_workspace_packing_keeps_concurrent_streams_apart():
    """Inside a fork..join region buffers of different streams never share bytes even when their tape intervals are disjoint
    (the tape lists one stream's launches after the other's, the device runs them side by side)."""
    from isr2_amd.plan import _pack_offsets
    work = [[4096, 10, 20, {1}], [4096, 30, 40, {2}], [4096, 50, 60, {0}], [4096, 61, 70, {0}], [4096, 2, 3, {0}], [4096, 90, 95, {0}]]
    offs, total = _pack_offsets(work, regions=[(5, 80)])
    spans = [(o, o + 4096) for o in offs]
    def disjoint(a, b):
        return spans[a][1] <= spans[b][0] or spans[b][1] <= spans[a][0]
    assert disjoint(0, 1) and disjoint(0, 2) and disjoint(1, 2) and disjoint(0, 3) and disjoint(1, 3)
    assert offs[2] == offs[3]                                     # same stream, disjoint lifetimes: reuse
    assert offs[4] == offs[5] == offs[0] or total <= 3 * 4096     # outside the region the plain lifetime rule applies
    offs1, total1 = _pack_offsets([w[:3] for w in work])          # no stream sets, no regions: one slot serves all
    assert total1 == 4096


def test_create_rejects_bad_plans(tmp_path):
    from isr2_amd import lib
    L = lib.load()
    h = ctypes.c_void_p()
    assert L.ff_create(str(tmp_path / "missing.ffplan").encode(), ctypes.byref(h)) != 0 and not h.value
    p = tmp_path / "garbage.ffplan"
    p.write_bytes(b"not a plan at all")
    assert L.ff_create(str(p).encode(), ctypes.byref(h)) != 0
    assert b"FFPLAN3" in L.ff_last_error()
    p = tmp_path / "abi.ffplan"
    p.write_bytes(b"FFPLAN3\0" + struct.pack("<iI", 999, 0))
    assert L.ff_create(str(p).encode(), ctypes.byref(h)) != 0 and b"ABI" in L.ff_last_error()
    p = tmp_path / "name.ffplan"
    p.write_bytes(b"FFPLAN3\0" + struct.pack("<iI", L.ff_abi_version(), 1) + struct.pack("<I", 9) + b"ff_nosuch")
    assert L.ff_create(str(p).encode(), ctypes.byref(h)) != 0 and b"ff_nosuch" in L.ff_last_error()
    assert L.ff_forward(None, None, 1, 1, 1, None, None) != 0     # null handle is an error, not a crash
    assert L.ff_destroy(None) == 0

    # a structurally valid plan whose ONE call does not fit the entry point's prototype: rejected before anything is
    # allocated or launched -- wrong arity, a float where a pointer belongs, an input offset beyond the lr tensor
    K_INT, K_FLT, K_NULL, K_WEIGHT, K_WORK, K_INPUT, K_OUTPUT, K_STREAM = range(8)
    proto = lib.parse_header()["ff_avgpool2"][1]                 # (const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, void* stream)
    assert len(proto) == 9

    def arg(kind, a=0, b=0, f=None):
        return struct.pack("<B3x", kind) + (struct.pack("<d8x", f) if f is not None else struct.pack("<qq", a, b))

    def plan_with(args):
        blob = b"FFPLAN3\0" + struct.pack("<iI", L.ff_abi_version(), 1) + struct.pack("<I", 11) + b"ff_avgpool2"
        blob += struct.pack("<I", 0)                                           # no weight slots
        blob += struct.pack("<q4i4i", 1 << 20, 1, 3, 8, 8, 1, 3, 32, 32)       # workspace bytes, in / out shapes
        blob += struct.pack("<I", 1) + struct.pack("<HHH", 0, len(args), 0) + b"".join(args)
        return blob
    good = [arg(K_INPUT, 0, 0), arg(K_INT, 3), arg(K_WORK, 0, 0), arg(K_INT, 3), arg(K_INT, 1), arg(K_INT, 8), arg(K_INT, 8), arg(K_INT, 3), arg(K_STREAM)]
    cases = {"arity": good[:-2] + [good[-1]], "kind": [arg(K_FLT, f=1.0)] + good[1:], "input offset": [arg(K_INPUT, 0, 4 * 3 * 8 * 8)] + good[1:],
             "output offset": good[:2] + [arg(K_OUTPUT, 0, 4 * 3 * 32 * 32 + 4)] + good[3:], "stream kind": good[:-1] + [arg(K_INT, 0)]}
    for what, args in cases.items():
        p = tmp_path / "bad_call.ffplan"
        p.write_bytes(plan_with(args))
        assert L.ff_create(str(p).encode(), ctypes.byref(h)) != 0 and not h.value, what
        err = L.ff_last_error()
        assert (b"argument" in err or b"reference" in err), (what, err)


@pytest.mark.gpu
def test_native_executor_replays_the_python_forward_bit_for_bit(tmp_path, synth_sd):
    """export_plan on the 48x48 reference-golden input, then ff_create / ff_upload / ff_finalize / ff_forward through ctypes:
    equal to the Python-sequenced forward bit for bit (same kernels, same arguments, library-owned memory), equal to the
    REFERENCE golden within the parity bar, repeatable, and graph-capturable."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from isr2_amd import ops, plan
    from isr2_amd.lib import FFError
    from isr2_amd.model import FreqFusionHIP
    ops.set_gemm_mode("bf16x3")
    dev = torch.device("cuda:0")
    model = FreqFusionHIP(synth_sd, dev)
    g = np.load(os.path.join(HERE, "golden", "c48_u8.npz"))
    lr = torch.from_numpy(g["lr"]).to(dev)
    stem = str(tmp_path / "ff48")
    info = plan.export_plan(model, lr, stem)
    print("plan:", info)
    assert info["calls"] > 1000 and info["workspace_bytes"] < info["activation_bytes_unpacked"] / 4
    assert info["streams"] == 3                                          # the host's three-stream schedule is part of the plan
    ref = model(lr)
    nat = plan.NativeModel(stem + ".ffplan", stem + ".ffwts")
    try:
        assert nat.io_shapes() == ((1, 3, 48, 48), (1, 3, 192, 192))
        out = nat(lr)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), "C executor differs from the Python-sequenced forward"
        assert (out.cpu() - torch.from_numpy(g["full/final"])).abs().max().item() < 2e-4
        lr2 = torch.rand(1, 3, 48, 48, device=dev)                       # another input through the same plan
        assert torch.equal(nat(lr2), model(lr2))
        static_in, static_out = lr.clone(), torch.empty_like(out)
        gr = torch.cuda.CUDAGraph()
        L = nat.L
        with torch.cuda.graph(gr):
            assert L.ff_forward(nat.h, static_in.data_ptr(), 1, 48, 48, static_out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(static_out, ref)
        with pytest.raises(FFError):
            nat(torch.zeros(1, 3, 40, 48, device=dev))                   # a plan is for one input shape
    finally:
        nat.close()
    # the same model as a single-stream plan: fewer workspace bytes, same result
    info1 = plan.export_plan(model, lr, stem + "_1s", multi_stream=False)
    assert info1["streams"] == 1 and info1["workspace_bytes"] <= info["workspace_bytes"]
    nat1 = plan.NativeModel(stem + "_1s.ffplan", stem + "_1s.ffwts")
    try:
        assert torch.equal(nat1(lr), ref)
    finally:
        nat1.close()
    h = ctypes.c_void_p()
    L = plan._lib.load()
    assert L.ff_create((stem + ".ffplan").encode(), ctypes.byref(h)) == 0
    assert L.ff_finalize(h) != 0 and b"never uploaded" in L.ff_last_error()     # slots must be filled first
    assert L.ff_upload(h, b"no.such.slot", ctypes.c_char_p(b"x"), 1) != 0
    L.ff_destroy(h)
