"""Launch plans for the C-level executor (ff_create / ff_upload / ff_finalize / ff_forward / ff_destroy).

The Python host (experts.py / fusion.py) is the single description of the kernel sequence.  `export_plan` runs it ONCE for one
input shape with every C-ABI call intercepted, and writes

    <stem>.ffplan   the tape: for every launch the entry point and its arguments, pointers rewritten as
                    (weight slot, offset) | (workspace, offset) | (input, offset) | (output, offset);
                    the prepared-weight slot table; the workspace size after liveness-based packing
    <stem>.ffwts    the prepared weights (the tensors prep.py / the model constructors made), one record per slot

csrc/ff_executor.hip replays a plan without Python: ff_create(plan) allocates the slots and the workspace, ff_upload fills the
slots (from the .ffwts records or from any host / device buffer), ff_forward(lr_dev, out_dev, stream) issues the ~1700
launches on one stream.  `NativeModel` below is the ctypes binding a C or C++ caller would mirror.

A plan is specific to (input shape, contraction mode, kernel library build): the header stores the ABI version and the list
of entry-point names it uses; ff_create refuses a plan whose names the library does not export.
"""
from __future__ import annotations

import ctypes
import os
import struct
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as _lib

MAGIC_PLAN, MAGIC_WTS = b"FFPLAN3\0", b"FFWTS01\0"
FID_MARK = 0xFFFF
K_INT, K_FLT, K_NULL, K_WEIGHT, K_WORK, K_INPUT, K_OUTPUT, K_STREAM = range(8)
_PTR_TYPES = ("const float*", "float*", "void*", "const void*", "const unsigned char*", "unsigned char*", "double*", "const double*")
ALIGN = 256


# ---------------------------------------------------------------------------------------------------------------------
# recording
def _walk_tensors(obj, path: str, out: Dict[int, Tuple[str, torch.Tensor]], seen: set):
    """Every CUDA tensor reachable from the model objects (lists / tuples / dicts / attributes, and the prepared planes that
    ops.PREPARED holds for the weight tensors: split planes, halo / small-conv images, interleaved bias tables)."""
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            st = obj.untyped_storage()
            out.setdefault(st.data_ptr(), (path, obj))
        from . import ops as _ops
        for kind, val in _ops.PREPARED.of(obj):
            _walk_tensors(val, f"{path}.<{kind}>", out, seen)
        return
    if isinstance(obj, dict):
        for k, v in obj.items():
            _walk_tensors(v, f"{path}.{k}", out, seen)
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            _walk_tensors(v, f"{path}.{i}", out, seen)
    elif hasattr(obj, "__dict__") and not isinstance(obj, (type, torch.cuda.Stream, torch.cuda.CUDAGraph)):
        for k, v in vars(obj).items():
            if k in ("_graphs", "_side"):
                continue
            _walk_tensors(v, f"{path}.{k}", out, seen)


class _Intervals:
    """address -> (kind, id, base) lookup over half-open [start, end) ranges; a new range evicts the ones it overlaps (the
    caching allocator hands the same addresses to later tensors)."""

    def __init__(self):
        self.items: List[Tuple[int, int, int, int]] = []          # (start, end, kind, id)

    def add(self, start: int, nbytes: int, kind: int, ident: int):
        end = start + max(nbytes, 1)
        self.items = [it for it in self.items if it[1] <= start or it[0] >= end]
        self.items.append((start, end, kind, ident))

    def find(self, p: int):
        for s, e, k, i in self.items:
            if s <= p < e:
                return k, i, p - s
        return None


class _Recorder:
    def __init__(self, real, protos):
        self.real, self.protos = real, protos
        self.iv = _Intervals()
        self.calls: List[Tuple[str, list, int]] = []            # (entry point | "fork" | "join", args, stream index)
        self.work: List[list] = []                              # per activation buffer: [nbytes, first_call, last_call, {streams}]
        self.streams: Dict[int, int] = {}                       # hipStream_t value -> index (0 = the caller's stream)

    def note_alloc(self, t: torch.Tensor):
        if isinstance(t, torch.Tensor) and t.is_cuda:
            st = t.untyped_storage()
            self.work.append([st.nbytes(), -1, -1, set()])
            self.iv.add(st.data_ptr(), st.nbytes(), K_WORK, len(self.work) - 1)

    def __getattr__(self, name):
        fn = getattr(self.real, name)
        proto = self.protos.get(name)
        if proto is None or not proto[1] or proto[1][-1] != "void*" or name in ("ff_create", "ff_upload", "ff_forward"):
            return fn

        def wrapped(*args):
            sidx = self.streams.setdefault(int(args[-1] or 0), len(self.streams))
            enc = []
            for a, ty in zip(args[:-1], proto[1][:-1]):
                if ty in _PTR_TYPES:
                    if a is None or a == 0:
                        enc.append((K_NULL, 0, 0))
                        continue
                    hit = self.iv.find(int(a))
                    if hit is None:
                        raise _lib.FFError(f"plan recording: {name} got a pointer {int(a):#x} that belongs to no registered tensor "
                                           "(an allocation or a copy outside the intercepted path)")
                    kind, ident, off = hit
                    if kind == K_WORK:
                        w = self.work[ident]
                        w[1] = len(self.calls) if w[1] < 0 else w[1]
                        w[2] = len(self.calls)
                        w[3].add(sidx)
                    enc.append((kind, ident, off))
                elif ty in ("float", "double"):
                    enc.append((K_FLT, float(a), 0))
                else:
                    enc.append((K_INT, int(a), 0))
            enc.append((K_STREAM, 0, 0))
            self.calls.append((name, enc, sidx))
            return fn(*args)
        return wrapped

    def mark(self, what: str):
        self.calls.append((what, [], 0))


def _pack_offsets(work: List[list], regions: List[Tuple[int, int]] = ()) -> Tuple[List[int], int]:
    """Workspace offsets by lifetime: buffers whose [first, last] launch intervals overlap never share bytes; inside a fork..join
    region (launch indices) the streams run concurrently, so two buffers touched there by DIFFERENT streams never share bytes
    either, whatever their tape order.  Greedy by decreasing size, lowest fitting offset."""
    def in_region(w):
        return any(not (w[2] < a or w[1] > b) for a, b in regions)

    order = sorted((i for i, w in enumerate(work) if w[1] >= 0), key=lambda i: -work[i][0])
    placed: List[Tuple[int, int, int]] = []                     # (offset, end, buffer index)
    offs = [0] * len(work)
    total = 0
    reg = {i: in_region(work[i]) for i in order}
    for i in order:
        size = (work[i][0] + ALIGN - 1) // ALIGN * ALIGN
        f, l = work[i][1], work[i][2]
        si = work[i][3] if len(work[i]) > 3 else {0}
        busy = []
        for o, e, j in placed:
            wj = work[j]
            sj = wj[3] if len(wj) > 3 else {0}
            if not (wj[2] < f or wj[1] > l) or (reg[i] and reg[j] and (si != sj or len(si) > 1)):
                busy.append((o, e))
        busy.sort()
        cur = 0
        for o, e in busy:
            if o - cur >= size:
                break
            cur = max(cur, e)
        offs[i] = cur
        placed.append((cur, cur + size, i))
        total = max(total, cur + size)
    return offs, total


@torch.no_grad()
def export_plan(model, lr: torch.Tensor, stem: str, multi_stream: bool = True, verify: bool = True) -> dict:
    """Record model.forward(lr) and write <stem>.ffplan / <stem>.ffwts.  multi_stream: keep the host's three-stream schedule
    (the experts side by side between a fork and a join marker; the executor replays them on two internal streams).
    Returns a summary dict."""
    dev = model.dev
    lr = lr.to(dev, torch.float32).contiguous()
    ms = model.multi_stream
    model.multi_stream = bool(multi_stream)
    try:
        ref = model.forward(lr).clone()                             # also runs every lazy weight preparation
        torch.cuda.synchronize(dev)
        found: Dict[int, Tuple[str, torch.Tensor]] = {}
        _walk_tensors(model, "model", found, set())
        # the zero-padded concat / input buffers that outlive a forward (ops.persistent_zeros) are plan slots too: their untouched
        # parts must be zero on the executor's side, exactly as here
        from . import ops as _ops
        for (owner, role, _dev), t in sorted(_ops._PERSIST_EAGER.items(), key=lambda kv: kv[0][1]):
            if t.is_cuda and t.device == torch.device(dev) and t.data_ptr() not in found:
                found[t.data_ptr()] = (f"persist.{role}.{'x'.join(str(v) for v in t.shape)}", t)
        real = _lib.load()
        rec = _Recorder(real, _lib.parse_header())
        slots = []                                                  # (name, storage ptr, nbytes)
        for ptr, (path, t) in sorted(found.items(), key=lambda kv: kv[1][0]):
            st = t.untyped_storage()
            rec.iv.add(ptr, st.nbytes(), K_WEIGHT, len(slots))
            slots.append((path, ptr, st.nbytes(), t))
        lst = lr.untyped_storage()
        rec.iv.add(lst.data_ptr(), lst.nbytes(), K_INPUT, 0)
        lr_off = lr.data_ptr() - lst.data_ptr()
        assert lr_off == 0

        orig = {n: getattr(torch, n) for n in ("empty", "empty_like", "zeros", "zeros_like")}

        def hook(fn, zeroing):
            def f(*a, **k):
                t = fn(*a, **k)
                if zeroing and isinstance(t, torch.Tensor) and t.is_cuda:
                    # a buffer zeroed DURING a forward is a memset the tape does not carry: the replay would read stale bytes
                    raise _lib.FFError("plan recording: the forward zero-fills a device buffer per call (torch.zeros / zeros_like inside the "
                                       "recorded pass); make it a persistent buffer (ops.persistent_zeros) or an ff_* fill")
                rec.note_alloc(t)
                return t
            return f
        for n, fn in orig.items():
            setattr(torch, n, hook(fn, n.startswith("zeros")))
        _lib._lib = rec
        model._marker = rec.mark
        rec.streams[int(torch.cuda.current_stream(dev).cuda_stream)] = 0
        try:
            out = model.forward(lr)
        finally:
            model._marker = None
            _lib._lib = real
            for n, fn in orig.items():
                setattr(torch, n, fn)
        torch.cuda.synchronize(dev)
        if not torch.equal(out, ref):
            raise _lib.FFError("plan recording changed the result")
    finally:
        model.multi_stream = ms

    # the returned tensor's buffer becomes the OUTPUT space
    hit = rec.iv.find(out.data_ptr())
    if hit is None or hit[0] != K_WORK or hit[2] != 0:
        raise _lib.FFError("plan recording: the output tensor is not a fresh allocation")
    out_buf = hit[1]
    regions, start = [], None
    for i, (n, _, _) in enumerate(rec.calls):
        if n == "fork":
            start = i
        elif n == "join" and start is not None:
            regions.append((start, i))
            start = None
    offs, wbytes = _pack_offsets([w if i != out_buf else [0, -1, -1, set()] for i, w in enumerate(rec.work)], regions)
    # The host issues one expert after the other inside a region; an eager (un-captured) replay would reach the last stream's
    # first launch only after queueing everything else.  Interleave the region's launches across its streams in proportion
    # (order within a stream kept, so the lifetimes used for the packing above still hold).
    calls = list(rec.calls)
    for a, b in regions:
        per = {}
        for c in calls[a + 1:b]:
            per.setdefault(c[2], []).append(c)
        merged, pos = [], {k: 0 for k in per}
        while any(pos[k] < len(per[k]) for k in per):
            k = min((k for k in per if pos[k] < len(per[k])), key=lambda k: (pos[k] + 1) / len(per[k]))
            merged.append(per[k][pos[k]])
            pos[k] += 1
        calls[a + 1:b] = merged
    used_slots = sorted({e[1] for _, enc, _ in rec.calls for e in enc if e[0] == K_WEIGHT})
    remap = {s: i for i, s in enumerate(used_slots)}
    names = sorted({n for n, _, _ in rec.calls if n not in ("fork", "join")})
    nstreams = max(len(rec.streams), 1)
    if nstreams > 3:
        raise _lib.FFError(f"plan recording: {nstreams} streams seen, the executor replays at most 3")
    fid = {n: i for i, n in enumerate(names)}

    with open(stem + ".ffplan", "wb") as f:
        f.write(MAGIC_PLAN)
        f.write(struct.pack("<iI", int(real.ff_abi_version()), len(names)))
        for n in names:
            b = n.encode()
            f.write(struct.pack("<I", len(b)) + b)
        f.write(struct.pack("<I", len(used_slots)))
        for s in used_slots:
            b = slots[s][0].encode()
            f.write(struct.pack("<I", len(b)) + b + struct.pack("<q", slots[s][2]))
        f.write(struct.pack("<q4i4i", wbytes, *lr.shape, *out.shape))
        f.write(struct.pack("<I", len(calls)))
        for n, enc, sidx in calls:
            if n in ("fork", "join"):
                f.write(struct.pack("<HHH", FID_MARK, 0, 0 if n == "fork" else 1))
                continue
            f.write(struct.pack("<HHH", fid[n], len(enc), sidx))
            for kind, a, b in enc:
                if kind == K_FLT:
                    f.write(struct.pack("<Bxxxdq", kind, a, 0))
                elif kind == K_WEIGHT:
                    f.write(struct.pack("<Bxxxqq", kind, remap[a], b))
                elif kind == K_WORK:
                    if a == out_buf:
                        f.write(struct.pack("<Bxxxqq", K_OUTPUT, 0, b))
                    else:
                        f.write(struct.pack("<Bxxxqq", kind, 0, offs[a] + b))
                else:
                    f.write(struct.pack("<Bxxxqq", kind, int(a), int(b)))
    with open(stem + ".ffwts", "wb") as f:
        f.write(MAGIC_WTS + struct.pack("<I", len(used_slots)))
        for s in used_slots:
            path, ptr, nbytes, t = slots[s]
            raw = torch.empty(0, dtype=torch.uint8, device=dev).set_(t.untyped_storage(), 0, (nbytes,))   # the storage's bytes
            b = path.encode()
            f.write(struct.pack("<I", len(b)) + b + struct.pack("<q", nbytes))
            f.write(raw.cpu().numpy().tobytes())
    # Self-check of the WRITTEN files: the recording pass above still ran every host-side op, so it cannot show that
    # the tape alone reproduces the forward.  Load the plan through the executor and replay it twice -- the second time on a
    # workspace the first one left dirty -- and require the recorded forward's result, bit for bit.
    if verify:
        nat = NativeModel(stem + ".ffplan", stem + ".ffwts")
        try:
            for rep in range(2):
                got = nat(lr)
                torch.cuda.synchronize(dev)
                if not torch.equal(got, ref):
                    raise _lib.FFError(f"plan self-check: replay {rep + 1} of the written plan differs from the recorded forward "
                                       f"(max |d| = {float((got - ref).abs().max()):.3e}): a buffer or a copy the tape does not carry")
        finally:
            nat.close()
    return {"calls": sum(1 for n, _, _ in rec.calls if n not in ("fork", "join")), "streams": nstreams, "entry_points": len(names), "slots": len(used_slots),
            "weight_bytes": sum(slots[s][2] for s in used_slots), "workspace_bytes": wbytes,
            "activation_bytes_unpacked": sum(w[0] for w in rec.work if w[1] >= 0), "in_shape": tuple(lr.shape), "out_shape": tuple(out.shape)}


def read_weights(path: str):
    """Iterate (slot name, bytes) records of a .ffwts file."""
    with open(path, "rb") as f:
        if f.read(8) != MAGIC_WTS:
            raise _lib.FFError(f"{path}: not an FFWTS file")
        (n,) = struct.unpack("<I", f.read(4))
        for _ in range(n):
            (ln,) = struct.unpack("<I", f.read(4))
            name = f.read(ln).decode()
            (nb,) = struct.unpack("<q", f.read(8))
            yield name, f.read(nb)


# ---------------------------------------------------------------------------------------------------------------------
# the binding a C caller mirrors
class NativeModel:
    """ff_create -> ff_upload (every slot) -> ff_finalize -> ff_forward ... -> ff_destroy, through ctypes."""

    def __init__(self, plan_path: str, weights_path: Optional[str] = None):
        self.L = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self.L.ff_create(plan_path.encode(), ctypes.byref(h)))
        self.h = h
        if weights_path:
            for name, blob in read_weights(weights_path):
                buf = ctypes.create_string_buffer(blob, len(blob))
                _lib.check(self.L.ff_upload(self.h, name.encode(), ctypes.cast(buf, ctypes.c_void_p), len(blob)))
            _lib.check(self.L.ff_finalize(self.h))

    def io_shapes(self):
        a, b = (ctypes.c_int * 4)(), (ctypes.c_int * 4)()
        _lib.check(self.L.ff_model_io_shape(self.h, a, b))
        return tuple(a), tuple(b)

    def forward(self, lr: torch.Tensor) -> torch.Tensor:
        ishape, oshape = self.io_shapes()
        if tuple(lr.shape) != ishape or not lr.is_cuda or lr.dtype != torch.float32 or not lr.is_contiguous():
            raise _lib.FFError(f"NativeModel.forward: expected a contiguous CUDA float32 tensor of shape {ishape}")
        out = torch.empty(oshape, device=lr.device, dtype=torch.float32)
        _lib.check(self.L.ff_forward(self.h, lr.data_ptr(), ishape[0], ishape[2], ishape[3], out.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream))
        return out

    def close(self):
        if self.h:
            self.L.ff_destroy(self.h)
            self.h = None

    __call__ = forward
