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

MAGIC_PLAN, MAGIC_WTS = b"FFPLAN2\0", b"FFWTS01\0"
K_INT, K_FLT, K_NULL, K_WEIGHT, K_WORK, K_INPUT, K_OUTPUT, K_STREAM = range(8)
_PTR_TYPES = ("const float*", "float*", "void*", "const void*", "const unsigned char*", "unsigned char*", "double*", "const double*")
ALIGN = 256


# ---------------------------------------------------------------------------------------------------------------------
# recording
def _walk_tensors(obj, path: str, out: Dict[int, Tuple[str, torch.Tensor]], seen: set):
    """Every CUDA tensor reachable from the model objects (lists / tuples / dicts / attributes, and the prepared planes that
    ops.py caches on weight tensors as _ff_split / _ff_halo / _ff_quad)."""
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            st = obj.untyped_storage()
            out.setdefault(st.data_ptr(), (path, obj))
        for a in ("_ff_split", "_ff_halo", "_ff_quad", "_ff_small"):
            if hasattr(obj, a):
                _walk_tensors(getattr(obj, a), f"{path}.{a}", out, seen)
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
        self.calls: List[Tuple[str, list]] = []
        self.work: List[List[int]] = []                         # per activation buffer: [nbytes, first_call, last_call]

    def note_alloc(self, t: torch.Tensor):
        if isinstance(t, torch.Tensor) and t.is_cuda:
            st = t.untyped_storage()
            self.work.append([st.nbytes(), -1, -1])
            self.iv.add(st.data_ptr(), st.nbytes(), K_WORK, len(self.work) - 1)

    def __getattr__(self, name):
        fn = getattr(self.real, name)
        proto = self.protos.get(name)
        if proto is None or not proto[1] or proto[1][-1] != "void*" or name in ("ff_create", "ff_upload", "ff_forward"):
            return fn

        def wrapped(*args):
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
                    enc.append((kind, ident, off))
                elif ty in ("float", "double"):
                    enc.append((K_FLT, float(a), 0))
                else:
                    enc.append((K_INT, int(a), 0))
            enc.append((K_STREAM, 0, 0))
            self.calls.append((name, enc))
            return fn(*args)
        return wrapped


def _pack_offsets(work: List[List[int]]) -> Tuple[List[int], int]:
    """Workspace offsets by lifetime: buffers whose [first, last] launch intervals overlap never share bytes.  Greedy by
    decreasing size, lowest fitting offset."""
    order = sorted((i for i, w in enumerate(work) if w[1] >= 0), key=lambda i: -work[i][0])
    placed: List[Tuple[int, int, int, int]] = []                # (offset, end, first, last)
    offs = [0] * len(work)
    total = 0
    for i in order:
        size = (work[i][0] + ALIGN - 1) // ALIGN * ALIGN
        f, l = work[i][1], work[i][2]
        busy = sorted((o, e) for o, e, pf, pl in placed if not (pl < f or pf > l))
        cur = 0
        for o, e in busy:
            if o - cur >= size:
                break
            cur = max(cur, e)
        offs[i] = cur
        placed.append((cur, cur + size, f, l))
        total = max(total, cur + size)
    return offs, total


@torch.no_grad()
def export_plan(model, lr: torch.Tensor, stem: str) -> dict:
    """Record model.forward(lr) (single stream) and write <stem>.ffplan / <stem>.ffwts.  Returns a summary dict."""
    dev = model.dev
    lr = lr.to(dev, torch.float32).contiguous()
    ms = model.multi_stream
    model.multi_stream = False
    try:
        ref = model.forward(lr).clone()                             # also runs every lazy weight preparation
        torch.cuda.synchronize(dev)
        found: Dict[int, Tuple[str, torch.Tensor]] = {}
        _walk_tensors(model, "model", found, set())
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

        def hook(fn):
            def f(*a, **k):
                t = fn(*a, **k)
                rec.note_alloc(t)
                return t
            return f
        for n, fn in orig.items():
            setattr(torch, n, hook(fn))
        _lib._lib = rec
        try:
            out = model.forward(lr)
        finally:
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
    offs, wbytes = _pack_offsets([w if i != out_buf else [0, -1, -1] for i, w in enumerate(rec.work)])
    used_slots = sorted({e[1] for _, enc in rec.calls for e in enc if e[0] == K_WEIGHT})
    remap = {s: i for i, s in enumerate(used_slots)}
    names = sorted({n for n, _ in rec.calls})
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
        f.write(struct.pack("<I", len(rec.calls)))
        for n, enc in rec.calls:
            f.write(struct.pack("<HH", fid[n], len(enc)))
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
    return {"calls": len(rec.calls), "entry_points": len(names), "slots": len(used_slots),
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
