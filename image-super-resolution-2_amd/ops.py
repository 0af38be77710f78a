"""Torch-tensor front-end of the C ABI (include/ff_kernels.h).

PyTorch is plumbing here: it owns device memory and the stream.  Every function below only validates
shapes/strides and forwards raw pointers to libff_hip.so; there is no PyTorch compute fallback -- a
missing library or a CPU tensor raises.
Tensors are fp32, channel-last.  "rows view": any tensor whose last dim is contiguous and whose
leading dims collapse to a single row stride (e.g. a channel slice t[..., a:b] of an NHWC buffer).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import lib as _lib

ACT = {None: 0, "none": 0, "gelu": 1, "relu": 2, "lrelu": 3, "sigmoid": 4}
T = torch.Tensor

# ---- optional per-launch instrumentation (bench.py / profiling only; None in the product path) ----------
_PROF = None          # list collecting (kernel_class, start_event, end_event, algorithmic_flops, algorithmic_bytes)
_META = [0.0, 0.0]    # [flops, bytes] noted by the wrapper currently running


def _note(flops: float = 0.0, nbytes: float = 0.0):
    _META[0] += flops
    _META[1] += nbytes


def _numel(*ts) -> float:
    """fp32-equivalent element count (a bf16 tensor counts half: _note multiplies by 4 bytes)."""
    return float(sum(t.numel() * (t.element_size() / 4.0) for t in ts if t is not None))


def _instrument(fn):
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        if _PROF is None:
            return fn(*a, **k)
        _META[0] = _META[1] = 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        _PROF.append((fn.__name__, e0, e1, _META[0], _META[1]))
        return r
    return wrapped


class profile:
    """with ops.profile() as p: ...  -> p.records() = [(kernel_class, ms, flops, bytes)] (events on the current stream)."""

    def __enter__(self):
        global _PROF
        self._rec = []
        _PROF = self._rec
        return self

    def __exit__(self, *exc):
        global _PROF
        _PROF = None
        torch.cuda.synchronize()

    def records(self):
        return [(n, e0.elapsed_time(e1), f, b) for n, e0, e1, f, b in self._rec]


def _L():
    return _lib.load()


# ---- contraction precision ---------------------------------------------------------------------------
# "f32"    : v_mfma_f32_32x32x2_f32, exact fp32 fma chain (parity mode, 157 TFLOP/s ceiling)
# "bf16x3" : split-operand bf16 MFMA, 3 terms, fp32-grade results (default; 833 TFLOP/s algorithmic ceiling)
# "bf16x2" : exact activations x bf16 weights;  "bf16": plain bf16 operands, fp32 accumulate
import os as _os

GEMM_MODES = {"f32": 0, "bf16": 1, "bf16x2": 2, "bf16x3": 3}
_GEMM_MODE = _os.environ.get("FF_GEMM", "bf16x3")


def set_gemm_mode(mode: str):
    global _GEMM_MODE
    if mode not in GEMM_MODES:
        raise _lib.FFError(f"unknown GEMM mode {mode!r}; expected one of {sorted(GEMM_MODES)}")
    _GEMM_MODE = mode


def _nterms() -> int:
    """MFMA terms per product of the fused token / halo / NAFNet kernels: 3 (split bf16, fp32-grade) or 1 (plain bf16)."""
    if _GEMM_MODE not in ("bf16x3", "bf16"):
        raise _lib.FFError(f"the fused bf16 kernels exist for the 'bf16x3' and 'bf16' contraction modes, not {_GEMM_MODE!r}")
    return 3 if _GEMM_MODE == "bf16x3" else 1


def fused_modes() -> bool:
    """True in the contraction modes that have the fused (token-stationary, window-resident, LDS-resident 3x3) kernels."""
    return _GEMM_MODE in ("bf16x3", "bf16")


def gemm_mode() -> str:
    return _GEMM_MODE


_SMALL = _os.environ.get("FF_SMALL_CONV", "1") != "0"     # fp32 VALU kernel for 3x3 convolutions with Cout <= 16, Cin <= 64
_HALO_ALL = _os.environ.get("FF_HALO", "1") == "2"          # tuning aid: every eligible 3x3 through the halo kernel
_HALO = _os.environ.get("FF_HALO", "1") != "0"     # LDS-resident 3x3 convolution (csrc/conv3x3_halo.hip); 0 = generic implicit GEMM


def set_halo(on: bool) -> None:
    global _HALO
    _HALO = bool(on)


_PAD_PITCH = _os.environ.get("FF_PAD_PITCH", "1") != "0"


_PAD_ROWS = _os.environ.get("FF_PAD_ROWS", "1") != "0"


def empty_rows_bf16(shape, device) -> T:
    """bf16 activation buffer [..., C] for an intermediate whose only consumer rounds it to bf16 anyway (plain-bf16 mode); the row
    pitch is a multiple of 64 elements (128 bytes)."""
    C = int(shape[-1])
    return torch.empty(tuple(shape[:-1]) + ((C + 63) // 64 * 64,), device=device, dtype=torch.bfloat16)[..., :C]


def empty_rows(shape, device) -> T:
    """fp32 activation buffer [..., C].  The row pitch is rounded up (FF_PAD_ROWS=0 disables) to a multiple of 32 floats when C is
    not one already (180 -> 192, 60 -> 64, 720 -> 736): every 32-channel segment a GEMM epilogue stores is then a whole
    128-byte line instead of straddling two.  Kernels take the pitch as their ld argument; the padding is never touched."""
    C = int(shape[-1])
    if _PAD_ROWS and C >= 48 and C % 32 != 0:
        return torch.empty(tuple(shape[:-1]) + ((C + 31) // 32 * 32,), device=device, dtype=torch.float32)[..., :C]
    return torch.empty(tuple(shape), device=device, dtype=torch.float32)


def empty_like_rows(x: T) -> T:
    return empty_rows(tuple(x.shape), x.device)


# Replay lane (model.graphed_async): two captured graphs of one shape may be in flight at once, so every buffer that outlives a
# forward (NAFNet's zero-padded input, the hierarchical fusion's concat buffers) is keyed by the lane that is being captured.
_LANE = 0


def lane() -> int:
    return _LANE


def set_lane(k: int) -> None:
    global _LANE
    _LANE = int(k)


# Buffers that must outlive a forward because part of them is zeroed ONCE and never written again (the hierarchical fusion's
# 76-channel concat buffers, NAFNet's zero-padded input).  Two ownership regimes:
#   * inside model._graph_for (capture key set): the buffer belongs to THAT graph entry -- its address is baked into the captured
#     launches -- and is dropped with it when the entry is evicted (drop_persistent);
#   * eager forwards (no capture key): ONE buffer per (owner, role, device), replaced and re-zeroed when the size changes, so a
#     directory of images with many distinct shapes does not accumulate a buffer per shape.
_CAPTURE_KEY = None
_PERSIST_GRAPH: Dict[tuple, Dict[tuple, T]] = {}
_PERSIST_EAGER: Dict[tuple, T] = {}


_POOL_MLP = _os.environ.get("FF_POOL_MLP", "1") != "0"     # pool finish + channel-attention MLP in one launch


def set_capture_key(key) -> None:
    global _CAPTURE_KEY
    _CAPTURE_KEY = key


def persistent_zeros(owner: int, role: str, shape, device) -> T:
    shape = tuple(int(v) for v in shape)
    if _CAPTURE_KEY is not None:
        slot = _PERSIST_GRAPH.setdefault(_CAPTURE_KEY, {})
        k = (owner, role, shape, str(device))
        buf = slot.get(k)
        if buf is None:
            buf = slot[k] = torch.zeros(shape, device=device, dtype=torch.float32)
        return buf
    k = (owner, role, str(device))
    buf = _PERSIST_EAGER.get(k)
    if buf is None or tuple(buf.shape) != shape:
        _PERSIST_EAGER.pop(k, None)                         # release the old size before the new one is allocated
        buf = _PERSIST_EAGER[k] = torch.zeros(shape, device=device, dtype=torch.float32)
    return buf


def drop_persistent(key=None, owner: int = None) -> None:
    """Release the buffers of one evicted graph entry (key), or everything an owner object holds (owner = id(obj))."""
    if key is not None:
        _PERSIST_GRAPH.pop(key, None)
    if owner is not None:
        for slot in _PERSIST_GRAPH.values():
            for k in [k for k in slot if k[0] == owner]:
                del slot[k]
        for k in [k for k in _PERSIST_EAGER if k[0] == owner]:
            del _PERSIST_EAGER[k]


def persistent_bytes() -> int:
    n = sum(b.numel() for b in _PERSIST_EAGER.values())
    n += sum(b.numel() for slot in _PERSIST_GRAPH.values() for b in slot.values())
    return 4 * n


class PreparedWeights:
    """Kernel-side images of weight tensors (bf16 hi/lo planes, LDS tile images, interleaved bias tables), built once per weight
    and kind.  An explicit registry instead of ad-hoc attributes on the tensors: it is thread-safe, it can be enumerated (the plan
    exporter ships these images), entries die with their weight (weak references), and `clear()` releases the device memory."""

    def __init__(self):
        import threading
        self._d: Dict[Tuple[int, str], tuple] = {}
        self._lock = threading.Lock()

    def get(self, w: T, kind: str, build):
        key = (id(w), kind)
        with self._lock:
            ent = self._d.get(key)
        if ent is not None and ent[0]() is w:
            return ent[1]
        val = build()
        import weakref
        ref = weakref.ref(w, lambda _r, key=key: self._d.pop(key, None))
        with self._lock:
            self._d[key] = (ref, val)
        return val

    def peek(self, w: T, kind: str):
        ent = self._d.get((id(w), kind))
        return ent[1] if ent is not None and ent[0]() is w else None

    def put(self, w: T, kind: str, val):
        self._d.pop((id(w), kind), None)
        return self.get(w, kind, lambda: val)

    def of(self, w: T):
        """Every prepared image of this weight: [(kind, value)]."""
        with self._lock:
            return [(k[1], e[1]) for k, e in self._d.items() if k[0] == id(w) and e[0]() is w]

    def nbytes(self) -> int:
        tot = 0

        def walk(v):
            nonlocal tot
            if isinstance(v, torch.Tensor):
                tot += v.numel() * v.element_size()
            elif isinstance(v, (tuple, list)):
                for u in v:
                    walk(u)
        with self._lock:
            for e in self._d.values():
                walk(e[1])
        return tot

    def clear(self):
        with self._lock:
            self._d.clear()


PREPARED = PreparedWeights()


def _split_weight(w: T, dynamic: bool, cin: int):
    """bf16 hi/lo planes [N][Kp] of a packed fp32 weight [N][taps*cin]; cached on the tensor object unless
    `dynamic`.  Per-tap padded (TAP) K layout when cin >= 32 and cin % 4 == 0, flat otherwise."""
    nterms = GEMM_MODES[_GEMM_MODE]
    cached = None if dynamic else PREPARED.peek(w, "split")
    if cached is not None and cached[4] >= nterms:
        return cached
    N, K = w.shape
    if cin >= 32 and cin % 4 == 0:
        Cp = (cin + 31) // 32 * 32
        Kp = (K // cin) * Cp
    else:
        Cp, Kp = 0, (K + 31) // 32 * 32
    hi = torch.empty((N, Kp), device=w.device, dtype=torch.bfloat16)
    lo = torch.empty((N, Kp), device=w.device, dtype=torch.bfloat16) if nterms == 3 else None
    _lib.check(_L().ff_split_bf16(w.data_ptr(), N, K, Kp, cin, Cp, hi.data_ptr(), lo.data_ptr() if lo is not None else None,
                                  _stream()))
    ent = (hi, lo, Kp, Cp, nterms)
    if not dynamic:
        PREPARED.put(w, "split", ent)
    return ent


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: T, name: str):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or (t.dtype != torch.float32 and not (t.dtype == torch.bfloat16 and name.endswith("|bf16ok"))):
        raise _lib.FFError(f"{name.split('|')[0]}: expected a CUDA(HIP) float32 tensor")


def _ptr(t: Optional[T]) -> Optional[int]:
    if t is None:
        return None
    _chk(t, "arg")
    return t.data_ptr()


def rows_view(t: T, name: str = "tensor") -> Tuple[int, int, int, int]:
    """-> (ptr, ld, rows, C) for a tensor whose leading dims collapse to rows of stride ld."""
    _chk(t, name)
    if t.dim() == 1:
        return t.data_ptr(), t.shape[0], 1, t.shape[0]
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        raise _lib.FFError(f"{name}: last dim must be contiguous")
    ld = t.stride(-2)
    rows = 1
    expect = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != expect:
            raise _lib.FFError(f"{name}: leading dims do not collapse to a constant row stride {tuple(t.shape)} {t.stride()}")
        expect = expect * t.shape[d]
        rows *= t.shape[d]
    return t.data_ptr(), ld, rows, t.shape[-1]


def _nhwc(t: T, name: str):
    """-> (ptr, ld, B, H, W, C) for a 4-D NHWC rows view."""
    if t.dim() != 4:
        raise _lib.FFError(f"{name}: expected [B,H,W,C]")
    p, ld, rows, c = rows_view(t, name)
    return p, ld, t.shape[0], t.shape[1], t.shape[2], c


class PoolPartials:
    """Per-workgroup pool partial sums written by a conv epilogue: mean[c] = inv_count * sum_r part[r][c].  vec_mlp() consumes
    them directly (ff_pool_vec_mlp: finish + channel-attention MLP in one launch); .mean() finishes them alone."""

    def __init__(self, part: T, C: int, inv_count: float):
        self.part, self.C, self.inv_count = part, C, inv_count

    def mean(self) -> T:
        pooled = torch.empty((1, self.C), device=self.part.device, dtype=torch.float32)
        _lib.check(_L().ff_pool_finish(self.part.data_ptr(), self.part.shape[0], self.part.shape[1], self.C, self.inv_count,
                                       pooled.data_ptr(), _stream()))
        return pooled


def conv2d(x: T, w: T, bias: Optional[T] = None, *, ksize=(1, 1), stride=(1, 1), pad=(0, 0), act=None,
           res: Optional[T] = None, mul: Optional[T] = None, alpha: float = 1.0, shuffle: int = 0,
           out: Optional[T] = None, tile_hint: int = 0, dynamic_w: bool = False, want_pool: bool = False, out_bf16: bool = False):
    """x [B,H,W,Cin] (rows view), w packed [Cout, KH*KW*Cin] -> [B,Ho,Wo,Cout] (or pixel-shuffled).
    want_pool (B == 1): also return the global average pool [1, Cout] of the output -> (out, pooled); the LDS-resident 3x3
    kernel produces it from its epilogue, any other path falls back to ff_pool_mean on the output.  want_pool="partials":
    the second value is a PoolPartials when the epilogue produced partial sums (else the finished pool)."""
    in_bf16 = x.dtype == torch.bfloat16
    xp, ldi, B, H, W, Cin = _nhwc(x, "conv2d.x|bf16ok" if in_bf16 else "conv2d.x")
    KH, KW = ksize
    Cout = w.shape[0]
    if w.dim() != 2 or w.shape[1] != KH * KW * Cin or not w.is_contiguous():
        raise _lib.FFError(f"conv2d: packed weight must be [Cout, {KH * KW * Cin}], got {tuple(w.shape)}")
    Ho = (H + 2 * pad[0] - KH) // stride[0] + 1
    Wo = (W + 2 * pad[1] - KW) // stride[1] + 1
    oshape = (B, Ho * 2, Wo * 2, Cout // 4) if shuffle == 2 else (B, Ho, Wo, Cout)
    if out is None:
        out = empty_rows_bf16(oshape, x.device) if out_bf16 else empty_rows(oshape, x.device)
    elif tuple(out.shape) != oshape:
        raise _lib.FFError(f"conv2d: out shape {tuple(out.shape)} != {oshape}")
    out_bf16 = out.dtype == torch.bfloat16
    op, ldo, *_ = _nhwc(out, "conv2d.out|bf16ok" if out_bf16 else "conv2d.out")
    halo_ok = (_HALO and _GEMM_MODE in ("bf16x3", "bf16") and (KH, KW) == (3, 3) and tuple(stride) == (1, 1) and tuple(pad) == (1, 1)
               and not dynamic_w and Cin >= 32 and Cin % 4 == 0 and ldi % 4 == 0 and xp % 16 == 0
               and (Cout <= 64 or Cin >= 128 or 128 < Cout <= 192 or _HALO_ALL) and H * W >= 1024 and xp != op and B * H * W * ldi < 2 ** 31)
    if (in_bf16 or out_bf16) and not (halo_ok and _GEMM_MODE == "bf16" and shuffle == 0 and res is None):
        raise _lib.FFError("conv2d: bf16 input / output rows exist for the LDS-resident 3x3 kernel in plain-bf16 mode only")
    rp, ldr = None, 0
    if res is not None:
        if tuple(res.shape) != oshape:
            raise _lib.FFError(f"conv2d: res shape {tuple(res.shape)} != {oshape}")
        rp, ldr, *_ = _nhwc(res, "conv2d.res")
    if (_SMALL and _GEMM_MODE != "f32" and (KH, KW) == (3, 3) and tuple(stride) == (1, 1) and tuple(pad) == (1, 1) and Cout <= 16
            and Cin * (1 if Cout == 1 else (4 if Cout <= 4 else 16)) <= 128 and shuffle == 0 and mul is None and not dynamic_w and H * W >= 4096 and xp != op):
        # small-channel tail convolutions: exact fp32 on the VALU from an LDS halo tile (csrc/conv3x3_small.hip).  Measured on
        # MI355X at 1024 x 1024: 16->3 205 -> <70 us, 8->1 160 -> <70, 16->1 204 -> <70, 32->3 166 -> 95, 6->16 173 -> 126; but
        # 64->3 202 -> 370 and 32->16 187 -> 199 (VALU-bound: Cin * CT FMAs per pixel and tap), hence the Cin * CT <= 128 rule
        from . import prep as _prep
        ws = PREPARED.get(w, "small", lambda: _prep.pack_conv3x3_small(w, Cin))
        _lib.check(_L().ff_conv3x3_small(xp, ldi, ws.data_ptr(), ws.shape[2], _ptr(bias), rp, ldr, op, ldo, B, H, W, Cin, Cout, ACT[act],
                                         float(alpha), _stream()))
    elif _GEMM_MODE == "f32":
        _lib.check(_L().ff_conv2d(xp, w.data_ptr(), _ptr(bias), _ptr(mul), rp, op, B, H, W, Cin, ldi, Ho, Wo, Cout, ldo, ldr,
                                  KH, KW, stride[0], stride[1], pad[0], pad[1], ACT[act], float(alpha), shuffle, tile_hint,
                                  _stream()))
    elif (_HALO and _GEMM_MODE in ("bf16x3", "bf16") and (KH, KW) == (3, 3) and tuple(stride) == (1, 1) and tuple(pad) == (1, 1)
          and not dynamic_w and Cin >= 32 and Cin % 4 == 0 and ldi % 4 == 0 and xp % 16 == 0
          and (Cout <= 64 or Cin >= 128 or 128 < Cout <= 192 or _HALO_ALL)   # measured (profiles/r01_conv3x3_halo_vs_igemm.txt); 64 -> 256 ties
          and H * W >= 1024 and xp != op and B * H * W * ldi < 2 ** 31):
        from . import prep as _prep

        nt_ = _nterms()

        def _mk_halo():
            bn = _prep.halo_bn(Cout)
            return (_prep.pack_conv3x3_halo(w, Cin, bn, nt_), bn)
        img = PREPARED.get(w, "halo" if nt_ == 3 else "halo1", _mk_halo)
        part = None
        if want_pool and B == 1 and shuffle == 0 and Cout <= img[1]:
            prow = int(_L().ff_conv3x3_halo_pool_rows(B, H, W, Cout, img[1], nt_))
            part = torch.empty((prow, img[1]), device=x.device, dtype=torch.float32)
        _lib.check(_L().ff_conv3x3_halo(xp, ldi, img[0].data_ptr(), img[1], _ptr(bias), _ptr(mul), rp, ldr, op, ldo, B, H, W,
                                        Cin, Cout, ACT[act], float(alpha), shuffle, _ptr(part), _nterms(),
                                        (1 if in_bf16 else 0) | (2 if out_bf16 else 0), _stream()))
        if part is not None:
            pp = PoolPartials(part, Cout, 1.0 / float(H * W))
            _note(2.0 * B * Ho * Wo * Cout * KH * KW * Cin, 4.0 * (_numel(x, w, out, res)))
            return out, (pp if want_pool == "partials" else pp.mean())
    else:
        aligned = (ldi % 4 == 0) and (xp % 16 == 0)
        hi, lo, Kp, Cp, _ = _split_weight(w, dynamic_w, Cin if aligned else 0)
        _lib.check(_L().ff_conv2d_bf16s(xp, hi.data_ptr(), lo.data_ptr() if lo is not None else None, Kp, Cp, _ptr(bias), _ptr(mul),
                                        rp, op, B, H, W, Cin, ldi, Ho, Wo, Cout, ldo, ldr, KH, KW, stride[0], stride[1],
                                        pad[0], pad[1], ACT[act], float(alpha), shuffle, GEMM_MODES[_GEMM_MODE], tile_hint,
                                        None, _stream()))
    _note(2.0 * B * Ho * Wo * Cout * KH * KW * Cin, 4.0 * (_numel(x, w, out, res)))
    if want_pool:
        return out, pool_mean(out)
    return out


def cab_fused(x: T, w1: T, b1: T, w2: T, b2: T, partials: bool = False):
    """HAT's CAB in one launch (plain bf16 only): conv3x3(x, w1) + b1 -> GELU -> conv3x3(., w2) + b2 and the global average pool of the
    result.  x [1,H,W,Cin] rows view, w1 / w2 packed [Cmid, 9*Cin] / [Cout, 9*Cmid] -> (out [1,H,W,Cout], pooled [1,Cout])."""
    if _GEMM_MODE != "bf16":
        raise _lib.FFError("cab_fused exists for the plain-bf16 contraction mode only")
    xp, ldi, B, H, W, Cin = _nhwc(x, "cab_fused.x")
    Cmid, Cout = w1.shape[0], w2.shape[0]
    if B != 1 or w1.shape[1] != 9 * Cin or w2.shape[1] != 9 * Cmid:
        raise _lib.FFError("cab_fused: expects one image and packed 3x3 weights [Cmid, 9*Cin], [Cout, 9*Cmid]")
    from . import prep as _prep
    i1 = PREPARED.get(w1, "cab1", lambda: _prep.pack_conv3x3_halo(w1, Cin, 64, 1))
    i2 = PREPARED.get(w2, "cab2", lambda: _prep.pack_conv3x3_halo(w2, Cmid, 192, 1))
    out = empty_rows((1, H, W, Cout), x.device)
    op, ldo, *_ = _nhwc(out, "cab_fused.out")
    prow = int(_L().ff_cab_fused_pool_rows(H, W))
    part = torch.empty((prow, 192), device=x.device, dtype=torch.float32)
    _lib.check(_L().ff_cab_fused(xp, ldi, i1.data_ptr(), b1.data_ptr(), i2.data_ptr(), b2.data_ptr(), op, ldo, H, W, Cin, Cmid, Cout,
                                 part.data_ptr(), _stream()))
    pp = PoolPartials(part, Cout, 1.0 / float(H * W))
    _note(2.0 * H * W * 9 * (Cmid * Cin + Cout * Cmid), 4.0 * H * W * (Cin + Cout))
    return out, (pp if partials else pp.mean())


def linear(x: T, w: T, bias: Optional[T] = None, *, act=None, res: Optional[T] = None, mul: Optional[T] = None,
           alpha: float = 1.0, out: Optional[T] = None, dynamic_w: bool = False, kmul: Optional[T] = None,
           gate_pairs: bool = False) -> T:
    """x [..., K] rows view, w [N, K] -> [..., N].  kmul [K]: scale of the input channels (W (kmul * x); split-bf16 modes).
    gate_pairs: SimpleGate in the epilogue, [..., N / 2] out with out[j] = y[2j] * y[2j+1] (rows of w interleaved by the caller)."""
    xp, ldi, rows, K = rows_view(x, "linear.x")
    N = w.shape[0]
    if w.dim() != 2 or w.shape[1] != K or not w.is_contiguous():
        raise _lib.FFError(f"linear: weight must be [N, {K}], got {tuple(w.shape)}")
    if (kmul is not None or gate_pairs) and _GEMM_MODE == "f32":
        raise _lib.FFError("linear: kmul / gate_pairs exist in the split-bf16 kernels only")
    No = N // 2 if gate_pairs else N
    oshape = tuple(x.shape[:-1]) + (No,)
    if out is None:
        out = empty_rows(oshape, x.device)
    op, ldo, orows, oc = rows_view(out, "linear.out")
    if orows != rows or oc != No:
        raise _lib.FFError("linear: out shape mismatch")
    rp, ldr = None, 0
    if res is not None:
        rp, ldr, rrows, rc = rows_view(res, "linear.res")
        if rrows != rows or rc != N:
            raise _lib.FFError("linear: res shape mismatch")
    if _GEMM_MODE == "f32":
        _lib.check(_L().ff_conv2d(xp, w.data_ptr(), _ptr(bias), _ptr(mul), rp, op, 1, 1, rows, K, ldi, 1, rows, N, ldo, ldr,
                                  1, 1, 1, 1, 0, 0, ACT[act], float(alpha), 0, 0, _stream()))
    else:
        aligned = (ldi % 4 == 0) and (xp % 16 == 0)
        hi, lo, Kp, Cp, _ = _split_weight(w, dynamic_w, K if aligned else 0)
        if kmul is not None and (kmul.numel() != K or not kmul.is_contiguous()):
            raise _lib.FFError(f"linear: kmul must be a contiguous [{K}] vector")
        _lib.check(_L().ff_conv2d_bf16s(xp, hi.data_ptr(), lo.data_ptr() if lo is not None else None, Kp, Cp, _ptr(bias), _ptr(mul),
                                        rp, op, 1, 1, rows, K, ldi, 1, rows, N, ldo, ldr, 1, 1, 1, 1, 0, 0, ACT[act],
                                        float(alpha), 1 if gate_pairs else 0, GEMM_MODES[_GEMM_MODE], 0, _ptr(kmul), _stream()))
    _note(2.0 * rows * N * K, 4.0 * (rows * K + N * K + rows * No * (2 if res is not None else 1)))
    return out


def window_attn(qkv: T, out: T, biasT: T, *, q_off: int, k_off: int, v_off: int, o_off: int, H: int, W: int, Hp: int,
                Wp: int, win: Tuple[int, int], kwin: Tuple[int, int], shift: Tuple[int, int], use_mask: bool, heads: int,
                d: int, scale: float, rel_table: Optional[T] = None) -> T:
    qp, ldq, B, h_, w_, _ = _nhwc(qkv, "window_attn.qkv")
    op, ldo, *_ = _nhwc(out, "window_attn.out")
    if (h_, w_) != (H, W) or tuple(out.shape[:3]) != (B, H, W):
        raise _lib.FFError("window_attn: qkv/out spatial dims mismatch")
    nk = kwin[0] * kwin[1]
    if tuple(biasT.shape) != (heads, nk, 256) or not biasT.is_contiguous():
        raise _lib.FFError(f"window_attn: biasT must be [{heads},{nk},256]")
    if q_off + heads * d > qkv.shape[-1] or k_off + heads * d > qkv.shape[-1] or v_off + heads * d > qkv.shape[-1] \
            or o_off + heads * d > out.shape[-1]:
        raise _lib.FFError("window_attn: channel offsets out of range")
    if _GEMM_MODE == "f32":
        _lib.check(_L().ff_window_attn(qp, ldq, q_off, k_off, v_off, op, ldo, o_off, biasT.data_ptr(), B, H, W, Hp, Wp, win[0],
                                       win[1], kwin[0], kwin[1], shift[0], shift[1], int(use_mask), heads, d, float(scale),
                                       _stream()))
    else:
        relp = None
        if rel_table is not None and tuple(kwin) != tuple(win):
            # overlapping keys: the rotated table of prep.pack_rel_overlap, gathered in the kernel through a per-key offset table
            n = (win[0] + kwin[0] - 1) * (win[1] + kwin[1] - 1)
            if tuple(rel_table.shape) != (heads, n) or not rel_table.is_contiguous() or kwin[1] % 8 or tuple(shift) != (0, 0):
                raise _lib.FFError(f"window_attn: overlapping-window rel_table must be [heads, {n}] (prep.pack_rel_overlap), kw % 8 == 0, no shift")
            relp = rel_table.data_ptr()
        elif rel_table is not None and tuple(kwin) == tuple(win) and win[1] & (win[1] - 1) == 0 and win[1] >= 8:
            if tuple(rel_table.shape) != (heads, (2 * win[0] - 1) * (2 * win[1] - 1)) or not rel_table.is_contiguous():
                raise _lib.FFError("window_attn: rel_table must be [heads, (2wh-1)*(2ww-1)]")
            relp = rel_table.data_ptr()
        from . import prep as _prep
        bq = PREPARED.get(biasT, "quad", lambda: _prep.quad_bias(biasT))          # load-time relayout, once per table
        _lib.check(_L().ff_window_attn_bf16s(qp, ldq, q_off, k_off, v_off, op, ldo, o_off, bq.data_ptr(), B, H, W, Hp, Wp,
                                             win[0], win[1], kwin[0], kwin[1], shift[0], shift[1], int(use_mask), heads, d,
                                             float(scale), 1 if _GEMM_MODE == "bf16" else 3, relp, _stream()))
    nwin = B * (Hp // win[0]) * (Wp // win[1])
    _note(4.0 * nwin * heads * 256 * nk * d, 4.0 * (4.0 * B * H * W * heads * d + heads * nk * 256))
    return out


def ocab_attn(qkv: T, out: T, rel_rotated: T, *, q_off: int, k_off: int, v_off: int, o_off: int = 0, H: int, W: int, heads: int, d: int,
              ws: int, ows: int, scale: float) -> T:
    """HAT OCAB attention stage in one persistent-workgroup launch (plain bf16; csrc/ocab_attn.hip): 16x16 query windows against
    24x24 zero-padded key windows, bias from the rotated compact table of prep.pack_rel_overlap [heads, 39*39]."""
    if _GEMM_MODE != "bf16":
        raise _lib.FFError("ocab_attn exists for the plain-bf16 contraction mode only")
    qp, ldq, B, h_, w_, _ = _nhwc(qkv, "ocab_attn.qkv")
    out_bf16 = out.dtype == torch.bfloat16
    op, ldo, *_ = _nhwc(out, "ocab_attn.out|bf16ok" if out_bf16 else "ocab_attn.out")
    if (h_, w_) != (H, W) or tuple(out.shape[:3]) != (B, H, W) or tuple(rel_rotated.shape) != (heads, (ws + ows - 1) ** 2) or not rel_rotated.is_contiguous():
        raise _lib.FFError("ocab_attn: shape mismatch")
    _lib.check(_L().ff_ocab_attn(qp, ldq, q_off, k_off, v_off, op, ldo, o_off, rel_rotated.data_ptr(), B, H, W, heads, d, ws, ows,
                                 float(scale), int(out_bf16), _stream()))
    nwin = B * (H // ws) * (W // ws)
    _note(4.0 * nwin * heads * ws * ws * ows * ows * d, 4.0 * B * H * W * heads * d * 4)
    return out


def win_attn_fused(x: T, out: T, pk: dict, rel_padded: T, *, gamma: Optional[T], beta: Optional[T], eps: float = 1e-5,
                   H: int, W: int, Hp: int, Wp: int, win: Tuple[int, int], shift: Tuple[int, int], use_mask: bool,
                   head0: int = 0, nheads: Optional[int] = None, o_off: int = 0, zero_pad: bool = False,
                   want_xn: bool = False, v_out: Optional[T] = None, v_off: int = 0):
    """LayerNorm + qkv projection + window attention of `nheads` heads in one launch (csrc/win_attn_fused.hip); pk from
    prep.pack_win_attn, rel_padded from prep.pack_win_rel.  x [B,H,W,K] rows view; out [B,H,W,>= o_off + heads*d].
    want_xn: also return the normalised rows; v_out: receives v of the processed heads at channel v_off + g*d."""
    xp, ldx, B, h_, w_, K = _nhwc(x, "win_attn_fused.x")
    out_bf16 = out.dtype == torch.bfloat16
    op, ldo, *_ = _nhwc(out, "win_attn_fused.out|bf16ok" if out_bf16 else "win_attn_fused.out")
    xn_bf16 = want_xn == "bf16"
    if (out_bf16 or xn_bf16) and _GEMM_MODE != "bf16":
        raise _lib.FFError("win_attn_fused: bf16 outputs exist for the plain-bf16 contraction mode only")
    if (h_, w_) != (H, W) or tuple(out.shape[:3]) != (B, H, W) or K != pk["K"]:
        raise _lib.FFError("win_attn_fused: x/out dims mismatch")
    nheads = pk["heads"] - head0 if nheads is None else nheads
    if tuple(rel_padded.shape[:2]) != (pk["heads"], 2 * win[0] - 1) or not rel_padded.is_contiguous():
        raise _lib.FFError("win_attn_fused: rel_padded must be [heads][2wh-1][stride]")
    if _GEMM_MODE not in ("bf16x3", "bf16"):
        raise _lib.FFError("win_attn_fused exists for the bf16 contraction modes only")
    xn, xnp, ldxn = None, None, 0
    if want_xn:
        xn = empty_rows_bf16(tuple(x.shape), x.device) if xn_bf16 else empty_like_rows(x)
        xnp, ldxn, _, _ = rows_view(xn, "win_attn_fused.xn|bf16ok" if xn_bf16 else "win_attn_fused.xn")
    vp, ldv = None, 0
    if v_out is not None:
        vp, ldv, vb, vh, vw, _ = _nhwc(v_out, "win_attn_fused.v_out")
        if (vb, vh, vw) != (B, H, W):
            raise _lib.FFError("win_attn_fused: v_out dims mismatch")
    _lib.check(_L().ff_win_attn_fused(xp, ldx, op, ldo, o_off, _ptr(gamma), _ptr(beta), float(eps), pk["w"].data_ptr(),
                                      pk["b"].data_ptr(), rel_padded.data_ptr(), rel_padded.shape[1], rel_padded.shape[2], B, H, W,
                                      Hp, Wp, win[0], win[1], shift[0], shift[1], int(use_mask), head0, nheads, pk["d"], K,
                                      int(zero_pad), xnp, ldxn, vp, ldv, v_off, 1 if _GEMM_MODE == "bf16" else 3, int(out_bf16), int(xn_bf16),
                                      _stream()))
    nwin = B * (Hp // win[0]) * (Wp // win[1])
    _note(2.0 * nwin * 256 * K * 3 * nheads * pk["d"] + 4.0 * nwin * nheads * 256 * 256 * pk["d"],
          4.0 * B * H * W * (K * (1 + (0 if not want_xn else (0.5 if xn_bf16 else 1.0))) + nheads * pk["d"] * ((0.5 if out_bf16 else 1.0) + (1 if v_out is not None else 0))))
    return (out, xn) if want_xn else out


def token_mlp(x: T, gamma: T, beta: T, pk: dict, eps: float = 1e-5) -> T:
    """x + fc2(GELU(fc1(LayerNorm(x)))) in one launch (bf16x3 only); pk from prep.pack_token_mlp."""
    xp, ldx, rows, K = rows_view(x, "token_mlp.x")
    if K != pk["K"] or pk["N"] != K:
        raise _lib.FFError("token_mlp: shape mismatch")
    out = empty_like_rows(x)
    op_, ldo_, _, _ = rows_view(out, "token_mlp.out")
    _lib.check(_L().ff_token_mlp(xp, ldx, op_, ldo_, rows, K, pk["ht"], pk["N"], gamma.data_ptr(), beta.data_ptr(),
                                 float(eps), pk["w"].data_ptr(), pk["b1"].data_ptr(), pk["b2"].data_ptr(), _nterms(), _stream()))
    _note(4.0 * rows * K * pk["ht"] * 32, 8.0 * rows * K)
    return out


def token_projmlp(att: T, x: T, pk: dict, gamma: T, beta: T, *, c2: Optional[T] = None, c2_scale: Optional[T] = None,
                  eps: float = 1e-5) -> T:
    """x1 = x + proj(att) + c2 * c2_scale; return x1 + fc2(GELU(fc1(LayerNorm(x1)))) in one launch (bf16x3); pk from
    prep.pack_token_projmlp."""
    att_bf16 = att.dtype == torch.bfloat16
    c2_bf16 = c2 is not None and c2.dtype == torch.bfloat16
    if (att_bf16 or c2_bf16) and _GEMM_MODE != "bf16":
        raise _lib.FFError("token_projmlp: bf16 att / c2 rows exist for the plain-bf16 contraction mode only")
    ap, lda, rows, K = rows_view(att, "token_projmlp.att|bf16ok" if att_bf16 else "token_projmlp.att")
    xp, ldx, xr, xk = rows_view(x, "token_projmlp.x")
    if xr != rows or xk != K or K != pk["mlp"]["K"]:
        raise _lib.FFError("token_projmlp: shape mismatch")
    cp, ldc = None, 0
    if c2 is not None:
        cp, ldc, cr, ck = rows_view(c2, "token_projmlp.c2|bf16ok" if c2_bf16 else "token_projmlp.c2")
        if cr != rows or ck != K or c2_scale is None or c2_scale.numel() != K:
            raise _lib.FFError("token_projmlp: c2 shape mismatch")
    out = empty_like_rows(x)
    op_, ldo_, _, _ = rows_view(out, "token_projmlp.out")
    m = pk["mlp"]
    _lib.check(_L().ff_token_projmlp(ap, lda, xp, ldx, cp, ldc, _ptr(c2_scale), op_, ldo_, rows, K, m["ht"], pk["proj"]["w"].data_ptr(),
                                     pk["proj"]["b"].data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), m["w"].data_ptr(),
                                     m["b1"].data_ptr(), m["b2"].data_ptr(), _nterms(), (1 if att_bf16 else 0) | (2 if c2_bf16 else 0), _stream()))
    _note(2.0 * rows * K * K + 4.0 * rows * K * m["ht"] * 32, 4.0 * rows * K * (2 + (0.5 if att_bf16 else 1.0) + (0 if c2 is None else (0.5 if c2_bf16 else 1.0))))
    return out


def token_linear(x: T, pk: dict, *, gamma: Optional[T] = None, beta: Optional[T] = None, eps: float = 1e-5, act=None,
                 res: Optional[T] = None, res2: Optional[T] = None, res2_scale: Optional[T] = None, want_xn: bool = False,
                 stats_range: Optional[Tuple[int, int]] = None, stats_eps: float = 1e-5):
    """res + res2*scale + act(LayerNorm?(x) @ W^T + b) for K <= 192 in one launch (bf16x3); pk from prep.pack_token_linear.
    want_xn: also return LayerNorm(x) (written by the same launch) -> (out, xn).
    stats_range=(lo, hi): also return [rows, 2] (mean, rstd) of the activated output channels [lo, hi) -> (out, stats)."""
    xp, ldx, rows, K = rows_view(x, "token_linear.x")
    if K != pk["K"]:
        raise _lib.FFError("token_linear: K mismatch")
    N = pk["N"]
    # wide outputs (qkv, fc1) get a row pitch that is a multiple of 32 floats: every 32-column store segment of the kernel
    # is then one whole 128-byte line (no partial-line writes); consumers take the pitch as their ld argument
    ldo = (N + 31) // 32 * 32 if ((_PAD_PITCH and N > 192) or (_PAD_ROWS and N >= 48)) else N
    out = torch.empty(tuple(x.shape[:-1]) + (ldo,), device=x.device, dtype=torch.float32)[..., :N]
    rp, ldr, r2p, ldr2 = None, 0, None, 0
    if res is not None:
        rp, ldr, rr, rc = rows_view(res, "token_linear.res")
        if rr != rows or rc != N:
            raise _lib.FFError("token_linear: res shape mismatch")
    if res2 is not None:
        r2p, ldr2, rr, rc = rows_view(res2, "token_linear.res2")
        if rr != rows or rc != N or res2_scale is None or res2_scale.numel() != N:
            raise _lib.FFError("token_linear: res2 shape mismatch")
    xn, xnp, ldxn = None, None, 0
    if want_xn:
        if gamma is None:
            raise _lib.FFError("token_linear: want_xn needs LayerNorm parameters")
        xn = empty_like_rows(x)
        xnp, ldxn, _, _ = rows_view(xn, "token_linear.xn")
    stats, sp, slo, shi = None, None, 0, 0
    if stats_range is not None:
        slo, shi = stats_range
        stats = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
        sp = stats.data_ptr()
    _lib.check(_L().ff_token_linear(xp, ldx, out.data_ptr(), ldo, rows, K, pk["kpad"], N, pk["nt"], _ptr(gamma), _ptr(beta), float(eps),
                                    pk["w"].data_ptr(), _ptr(pk["b"]), ACT[act], rp, ldr, r2p, ldr2, _ptr(res2_scale), xnp, ldxn,
                                    sp, slo, shi, float(stats_eps), _nterms(), _stream()))
    _note(2.0 * rows * N * K, 4.0 * (rows * K * (2 if want_xn else 1) + rows * N * (1 + (res is not None) + (res2 is not None))))
    if stats is not None:
        return out, stats
    return (out, xn) if want_xn else out


def dwconv3_gate_pool(t: T, w_tap: T, bias: T):
    """t [1,H,W,2C] -> (dw3x3(t)[..., :C] * dw3x3(t)[..., C:], its per-channel mean [1,C]) in one pass."""
    tp, ldi, B, H, W, C2 = _nhwc(t, "dwconv3_gate_pool.t")
    if B != 1 or C2 % 2 or tuple(w_tap.shape) != (9, C2):
        raise _lib.FFError("dwconv3_gate_pool: expects B == 1, [9, 2C] tap-major weights")
    C = C2 // 2
    out = torch.empty((1, H, W, C), device=t.device, dtype=torch.float32)
    pooled = torch.empty((1, C), device=t.device, dtype=torch.float32)
    nwork = int(_L().ff_dwconv3_gate_pool_workspace(C))
    work = torch.empty(nwork, device=t.device, dtype=torch.float32)
    _lib.check(_L().ff_dwconv3_gate_pool(tp, ldi, out.data_ptr(), C, H, W, C, w_tap.data_ptr(), bias.data_ptr(), pooled.data_ptr(),
                                         work.data_ptr(), nwork, _stream()))
    _note(2.0 * H * W * C2 * 9, 4.0 * H * W * (C2 + C))
    return out, pooled


def naf_front(x: T, pk: dict, ln_g: T, ln_b: T, w_tap: T, dw_bias: T, eps: float = 1e-6):
    """x [1,H,W,C] (C = 64 / 128) -> (SimpleGate(dw3x3(conv1(LayerNorm2d(x)))) [1,H,W,C], its per-channel mean [1,C]) in one launch;
    pk = prep.pack_token_linear(conv1 weight [2C, C], bias), w_tap [9, 2C] tap-major depth-wise weights."""
    xp, ldx, B, H, W, C = _nhwc(x, "naf_front.x")
    if B != 1 or C not in (64, 128) or pk["N"] != 2 * C or pk["K"] != C or pk["kpad"] != C or tuple(w_tap.shape) != (9, 2 * C):
        raise _lib.FFError("naf_front: expects B == 1, C in (64, 128), conv1 packed as [2C, C], [9, 2C] tap-major weights")
    out = torch.empty((1, H, W, C), device=x.device, dtype=torch.float32)
    pooled = torch.empty((1, C), device=x.device, dtype=torch.float32)
    nwork = int(_L().ff_naf_front_workspace(H, W, C))
    work = torch.empty(nwork, device=x.device, dtype=torch.float32)
    _lib.check(_L().ff_naf_front(xp, ldx, H, W, C, ln_g.data_ptr(), ln_b.data_ptr(), float(eps), pk["w"].data_ptr(), pk["b"].data_ptr(),
                                 w_tap.data_ptr(), dw_bias.data_ptr(), out.data_ptr(), C, pooled.data_ptr(), work.data_ptr(), nwork,
                                 _nterms(), _stream()))
    _note(2.0 * H * W * 2 * C * (C + 9), 4.0 * H * W * 2 * C)
    return out, pooled


def naf_ffn(y: T, pk: dict, ln_g: T, ln_b: T, out_scale: T, eps: float = 1e-6) -> T:
    """y + out_scale * conv5(SimpleGate(conv4(LayerNorm(y)))) in one launch (C = 64 / 128, bf16x3)."""
    yp, ldy, rows, C = rows_view(y, "naf_ffn.y")
    if C != pk["C"]:
        raise _lib.FFError("naf_ffn: channel mismatch")
    out = torch.empty(tuple(y.shape), device=y.device, dtype=torch.float32)
    _lib.check(_L().ff_naf_ffn(yp, ldy, out.data_ptr(), C, rows, C, ln_g.data_ptr(), ln_b.data_ptr(), float(eps), pk["w"].data_ptr(),
                               pk["b4"].data_ptr(), pk["b5"].data_ptr(), out_scale.data_ptr(), _nterms(), _stream()))
    _note(2.0 * rows * C * 3 * C, 8.0 * rows * C)
    return out


def layernorm(x: T, gamma: T, beta: T, eps: float = 1e-5, out: Optional[T] = None) -> T:
    xp, ldi, rows, C = rows_view(x, "layernorm.x")
    if out is None:
        out = empty_like_rows(x)
    op, ldo, orows, oc = rows_view(out, "layernorm.out")
    if orows != rows or oc != C or gamma.numel() != C or beta.numel() != C:
        raise _lib.FFError("layernorm: shape mismatch")
    _lib.check(_L().ff_layernorm(xp, ldi, op, ldo, rows, C, gamma.data_ptr(), beta.data_ptr(), float(eps), _stream()))
    _note(0.0, 8.0 * rows * C)
    return out


def pool_partials(x: T):
    """First stage of the global average pool of one image [1,H,W,C] -> PoolPartials (vec_mlp finishes them in its own launch)."""
    xp, ld, B, H, W, C = _nhwc(x, "pool_partials.x")
    if B != 1 or not _POOL_MLP:
        return pool_mean(x)
    P = H * W
    rows = int(_L().ff_pool_partial_rows(P))
    part = torch.empty((rows, C), device=x.device, dtype=torch.float32)
    _lib.check(_L().ff_pool_partials(xp, ld, P, C, part.data_ptr(), part.numel(), _stream()))
    return PoolPartials(part, C, 1.0 / float(P))


def pool_mean(x: T) -> T:
    """[B,H,W,C] rows view -> [B,C]."""
    xp, ld, B, H, W, C = _nhwc(x, "pool_mean.x")
    P = H * W
    nwork = int(_L().ff_pool_mean_workspace(B, P, C))
    work = torch.empty(nwork, device=x.device, dtype=torch.float32)
    out = torch.empty((B, C), device=x.device, dtype=torch.float32)
    _lib.check(_L().ff_pool_mean(xp, ld, B, P, C, out.data_ptr(), work.data_ptr(), nwork, _stream()))
    _note(0.0, 4.0 * B * P * C)
    return out


def vec_mlp(v, W1: T, b1: Optional[T], act1, W2: Optional[T] = None, b2: Optional[T] = None, act2=None,
            post: float = 1.0) -> T:
    """act2(W2 . act1(W1 . v + b1) + b2) * post per row of v [B, Cin] (W2 None: one layer).  v may be a PoolPartials: the pool is
    then finished inside the same launch (two-layer form with a hidden width <= 64), otherwise finished first."""
    if isinstance(v, PoolPartials):
        if W2 is not None and W1.shape[0] <= 64 and v.C <= 512 and _POOL_MLP:
            if W1.shape[1] != v.C or W2.shape[1] != W1.shape[0]:
                raise _lib.FFError("vec_mlp: weight shape mismatch")
            out = torch.empty((1, W2.shape[0]), device=v.part.device, dtype=torch.float32)
            _lib.check(_L().ff_pool_vec_mlp(v.part.data_ptr(), v.part.shape[0], v.part.shape[1], v.inv_count, v.C, W1.data_ptr(), _ptr(b1),
                                            W1.shape[0], ACT[act1], W2.data_ptr(), _ptr(b2), W2.shape[0], ACT[act2], float(post),
                                            out.data_ptr(), None, _stream()))
            return out
        v = v.mean()
    B, Cin = v.shape
    Ch = W1.shape[0]
    Cout = W2.shape[0] if W2 is not None else Ch
    if W1.shape[1] != Cin or (W2 is not None and W2.shape[1] != Ch):
        raise _lib.FFError("vec_mlp: weight shape mismatch")
    out = torch.empty((B, Cout), device=v.device, dtype=torch.float32)
    _lib.check(_L().ff_vec_mlp(v.data_ptr(), B, Cin, W1.data_ptr(), _ptr(b1), Ch, ACT[act1], _ptr(W2), _ptr(b2), Cout,
                               ACT[act2], float(post), out.data_ptr(), _stream()))
    return out


def pixel_mlp(x: T, W1: T, b1: Optional[T], act1, w2: T, b2: float, act2) -> T:
    """[..., C] rows -> [..., 1]: act2(w2 . act1(W1 x + b1) + b2) per row, hidden width <= 16 (fp32, one launch)."""
    xp, ldi, rows, C = rows_view(x, "pixel_mlp.x")
    Hd = W1.shape[0]
    if tuple(W1.shape) != (Hd, C) or w2.numel() != Hd or not W1.is_contiguous():
        raise _lib.FFError("pixel_mlp: weight shape mismatch")
    out = torch.empty(tuple(x.shape[:-1]) + (1,), device=x.device, dtype=torch.float32)
    _lib.check(_L().ff_pixel_mlp(xp, ldi, rows, C, Hd, W1.data_ptr(), _ptr(b1), ACT[act1], w2.reshape(-1).data_ptr(), float(b2), ACT[act2],
                                 out.data_ptr(), _stream()))
    _note(2.0 * rows * C * Hd, 4.0 * rows * (C + 1))
    return out


def dwconv2d(x: T, w_tap: T, bias: Optional[T] = None, *, ksize=(3, 3), stride=(1, 1), pad=(1, 1),
             post_scale: Optional[T] = None, post_shift: Optional[T] = None, act=None, out: Optional[T] = None,
             mul_in: Optional[T] = None) -> T:
    xp, ldi, B, H, W, C = _nhwc(x, "dwconv2d.x")
    KH, KW = ksize
    if tuple(w_tap.shape) != (KH * KW, C) or not w_tap.is_contiguous():
        raise _lib.FFError(f"dwconv2d: tap-major weight must be [{KH * KW},{C}]")
    Ho = (H + 2 * pad[0] - KH) // stride[0] + 1
    Wo = (W + 2 * pad[1] - KW) // stride[1] + 1
    if out is None:
        out = empty_rows((B, Ho, Wo, C), x.device)
    op, ldo, *_ = _nhwc(out, "dwconv2d.out")
    mp, ldm = None, 0
    if mul_in is not None:
        mp, ldm, mb, mh, mw, mc = _nhwc(mul_in, "dwconv2d.mul_in")
        if (mb, mh, mw, mc) != (B, Ho, Wo, C):
            raise _lib.FFError("dwconv2d: mul_in shape mismatch")
    _lib.check(_L().ff_dwconv2d(xp, ldi, op, ldo, B, H, W, C, Ho, Wo, w_tap.data_ptr(), _ptr(bias), KH, KW, stride[0],
                                stride[1], pad[0], pad[1], _ptr(post_scale), _ptr(post_shift), ACT[act], mp, ldm, _stream()))
    _note(2.0 * B * Ho * Wo * C * KH * KW, 4.0 * (B * H * W * C + B * Ho * Wo * C))
    return out


def token_linear_gated(x: T, x2: T, pk: dict, cm: T, gb2: float, *, res: Optional[T] = None) -> T:
    """res + W (x * cm[channel] + x2 * sm[token]) + b with sm = sigmoid(gw2 . gelu(GW1 x + gb1) + gb2) in one launch (bf16x3);
    pk from prep.pack_token_linear_gated."""
    xp, ldx, rows, K = rows_view(x, "token_linear_gated.x")
    x2p, ldx2, r2, k2 = rows_view(x2, "token_linear_gated.x2")
    if r2 != rows or k2 != K or K != pk["K"] or cm.numel() != K:
        raise _lib.FFError("token_linear_gated: shape mismatch")
    N = pk["N"]
    out = empty_rows(tuple(x.shape[:-1]) + (N,), x.device)
    op, ldo, _, _ = rows_view(out, "token_linear_gated.out")
    rp, ldr = None, 0
    if res is not None:
        rp, ldr, rr, rc = rows_view(res, "token_linear_gated.res")
        if rr != rows or rc != N:
            raise _lib.FFError("token_linear_gated: res shape mismatch")
    _lib.check(_L().ff_token_linear_gated(xp, ldx, x2p, ldx2, cm.data_ptr(), None, pk["gb1"].data_ptr(), pk["gw2"].data_ptr(), float(gb2),
                                          op, ldo, rows, K, N, pk["nt"], pk["w"].data_ptr(), pk["b"].data_ptr(), rp, ldr, _nterms(), _stream()))
    _note(2.0 * rows * N * K + 2.0 * rows * 32 * K, 4.0 * rows * (2 * K + N * (2 if res is not None else 1)))
    return out


def dwconv3x3_ln(x: T, w_tap: T, bias: Optional[T], stats: T, gamma: T, beta: T, mul_in: Optional[T] = None) -> T:
    """(dw3x3(LayerNorm(x)) + bias) * mul_in with the LayerNorm applied on load from per-token (mean, rstd) `stats`."""
    xp, ldi, B, H, W, C = _nhwc(x, "dwconv3x3_ln.x")
    if tuple(w_tap.shape) != (9, C) or tuple(stats.shape) != (B * H * W, 2) or gamma.numel() != C or beta.numel() != C:
        raise _lib.FFError("dwconv3x3_ln: shape mismatch")
    out = empty_rows((B, H, W, C), x.device)
    op, ldo, *_ = _nhwc(out, "dwconv3x3_ln.out")
    mp, ldm = None, 0
    if mul_in is not None:
        mp, ldm, mb, mh, mw, mc = _nhwc(mul_in, "dwconv3x3_ln.mul_in")
        if (mb, mh, mw, mc) != (B, H, W, C):
            raise _lib.FFError("dwconv3x3_ln: mul_in shape mismatch")
    _lib.check(_L().ff_dwconv3x3_ln(xp, ldi, op, ldo, B, H, W, C, w_tap.data_ptr(), _ptr(bias), stats.data_ptr(), gamma.data_ptr(),
                                    beta.data_ptr(), mp, ldm, _stream()))
    _note(2.0 * B * H * W * C * 9, 4.0 * B * H * W * C * (3 if mul_in is not None else 2))
    return out


def sgfn_tail(y: T, c2: int, w_tap: T, dw_bias: Optional[T], stats: T, gamma: T, beta: T, w2: T, b2: Optional[T], res: Optional[T] = None) -> T:
    """res + fc2(y[..., :c2] * (dw3x3(LayerNorm(y[..., c2:2 c2])) + dw_bias)) + b2 in one launch (plain bf16; csrc/sgfn_tail.hip).
    y [B,H,W,>= 2 c2] rows view (fc1's output), stats [B*H*W, 2] = (mean, rstd) of y[..., c2:], w_tap [9, c2], w2 [N, c2]."""
    if _GEMM_MODE != "bf16":
        raise _lib.FFError("sgfn_tail exists for the plain-bf16 contraction mode only")
    yp, ldh, B, H, W, Cy = _nhwc(y, "sgfn_tail.y")
    N = w2.shape[0]
    if Cy < 2 * c2 or tuple(w_tap.shape) != (9, c2) or tuple(stats.shape) != (B * H * W, 2) or gamma.numel() != c2 or beta.numel() != c2 \
            or w2.shape[1] != c2:
        raise _lib.FFError("sgfn_tail: shape mismatch")
    from . import prep as _prep
    tiles = PREPARED.get(w2, "sgfn", lambda: _prep.pack_sgfn_fc2(w2))
    out = empty_rows((B, H, W, N), y.device)
    op, ldo, *_ = _nhwc(out, "sgfn_tail.out")
    rp, ldr = None, 0
    if res is not None:
        if tuple(res.shape) != (B, H, W, N):
            raise _lib.FFError("sgfn_tail: res shape mismatch")
        rp, ldr, *_ = _nhwc(res, "sgfn_tail.res")
    _lib.check(_L().ff_sgfn_tail(yp, ldh, c2, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w_tap.data_ptr(), _ptr(dw_bias),
                                 tiles.data_ptr(), tiles.shape[0], _ptr(b2), rp, ldr, op, ldo, B, H, W, N, _stream()))
    _note(2.0 * B * H * W * c2 * (9 + N), 4.0 * B * H * W * (2 * c2 + 2 * N))
    return out


def mix2(a: T, b: Optional[T] = None, *, ka: float = 1.0, kb: float = 1.0, ca: Optional[T] = None,
         cb: Optional[T] = None, pa: Optional[T] = None, pb: Optional[T] = None, clamp01: bool = False,
         out: Optional[T] = None) -> T:
    ap, lda, rows, C = rows_view(a, "mix2.a")
    bp, ldb = None, 0
    if b is not None:
        bp, ldb, brows, bc = rows_view(b, "mix2.b")
        if brows != rows or bc != C:
            raise _lib.FFError("mix2: a/b shape mismatch")
    if out is None:
        out = empty_like_rows(a)
    op, ldo, orows, oc = rows_view(out, "mix2.out")
    if orows != rows or oc != C:
        raise _lib.FFError("mix2: out shape mismatch")

    def pvec(p, nm):
        if p is None:
            return None, 0
        pp, ldp, prows, pc = rows_view(p, nm)
        if prows != rows or pc != 1:
            raise _lib.FFError(f"{nm}: per-row vector must be [rows,1]")
        return pp, ldp

    pap, ldpa = pvec(pa, "mix2.pa")
    pbp, ldpb = pvec(pb, "mix2.pb")
    for v_, nm in ((ca, "ca"), (cb, "cb")):
        if v_ is not None and v_.numel() != C:
            raise _lib.FFError(f"mix2: {nm} must have {C} elements")
    _lib.check(_L().ff_mix2(op, ldo, ap, lda, bp, ldb, rows, C, float(ka), float(kb), _ptr(ca), _ptr(cb), pap, ldpa, pbp,
                            ldpb, int(clamp01), _stream()))
    _note(0.0, 4.0 * rows * C * (3 if b is not None else 2))
    return out


def fma3(a: Optional[T], b: T, c: T, alpha: float = 1.0, out: Optional[T] = None) -> T:
    bp, ldb, rows, C = rows_view(b, "fma3.b")
    cp, ldc, crows, cc = rows_view(c, "fma3.c")
    if crows != rows or cc != C:
        raise _lib.FFError("fma3: b/c shape mismatch")
    ap, lda = None, 0
    if a is not None:
        ap, lda, arows, ac = rows_view(a, "fma3.a")
        if arows != rows or ac != C:
            raise _lib.FFError("fma3: a shape mismatch")
    if out is None:
        out = empty_like_rows(b)
    op, ldo, orows, oc = rows_view(out, "fma3.out")
    if orows != rows or oc != C:
        raise _lib.FFError("fma3: out shape mismatch")
    _lib.check(_L().ff_fma3(op, ldo, ap, lda, bp, ldb, cp, ldc, rows, C, float(alpha), _stream()))
    _note(0.0, 4.0 * rows * C * (4 if a is not None else 3))
    return out


def affine(x: T, scale: T, shift: T, act=None, out: Optional[T] = None) -> T:
    xp, ldi, rows, C = rows_view(x, "affine.x")
    if out is None:
        out = empty_like_rows(x)
    op, ldo, orows, oc = rows_view(out, "affine.out")
    if orows != rows or oc != C or scale.numel() != C or shift.numel() != C:
        raise _lib.FFError("affine: shape mismatch")
    _lib.check(_L().ff_affine(op, ldo, xp, ldi, rows, C, scale.data_ptr(), shift.data_ptr(), ACT[act], _stream()))
    _note(0.0, 8.0 * rows * C)
    return out


def u8_to_f32_image(img_u8: T) -> T:
    """uint8 HWC [H,W,3] on the device -> fp32 [1,3,H,W] in [0,1] (reference io.py:64-68)."""
    if not (isinstance(img_u8, torch.Tensor) and img_u8.is_cuda and img_u8.dtype == torch.uint8 and img_u8.dim() == 3
            and img_u8.shape[2] == 3 and img_u8.is_contiguous()):
        raise _lib.FFError("u8_to_f32_image: expected a contiguous CUDA(HIP) uint8 [H,W,3] tensor")
    H, W, _ = img_u8.shape
    out = torch.empty((1, 3, H, W), device=img_u8.device, dtype=torch.float32)
    _lib.check(_L().ff_u8hwc_to_f32nchw(img_u8.data_ptr(), out.data_ptr(), H, W, _stream()))
    return out


def f32_to_u8_image(x: T) -> T:
    """fp32 [1,3,H,W] (or [3,H,W]) -> uint8 HWC [H,W,3]: clamp, *255, round half to even (reference io.py:71-76)."""
    _chk(x, "f32_to_u8_image.x")
    if x.dim() == 4:
        if x.shape[0] != 1:
            raise _lib.FFError("f32_to_u8_image: batch must be 1")
        x = x[0]
    if x.dim() != 3 or x.shape[0] != 3:
        raise _lib.FFError("f32_to_u8_image: expected [1,3,H,W] or [3,H,W]")
    x = x.contiguous()
    _, H, W = x.shape
    out = torch.empty((H, W, 3), device=x.device, dtype=torch.uint8)
    _lib.check(_L().ff_f32nchw_to_u8hwc(x.data_ptr(), out.data_ptr(), H, W, _stream()))
    return out


def nchw_to_nhwc(x: T, Hp: Optional[int] = None, Wp: Optional[int] = None, add: Optional[T] = None,
                 pad_mode: str = "zero", out: Optional[T] = None) -> T:
    _chk(x, "nchw_to_nhwc.x")
    x = x.contiguous()
    B, C, H, W = x.shape
    Hp, Wp = Hp or H, Wp or W
    if out is None:
        out = empty_rows((B, Hp, Wp, C), x.device)
    op, ldo, *_ = _nhwc(out, "nchw_to_nhwc.out")
    _lib.check(_L().ff_nchw_to_nhwc(x.data_ptr(), op, B, C, H, W, Hp, Wp, ldo, _ptr(add), 1 if pad_mode == "reflect" else 0,
                                    _stream()))
    return out


def nhwc_to_nchw(x: T, H: Optional[int] = None, W: Optional[int] = None, add: Optional[T] = None, clamp01: bool = False,
                 out: Optional[T] = None) -> T:
    xp, ldi, B, Hs, Ws, C = _nhwc(x, "nhwc_to_nchw.x")
    H, W = H or Hs, W or Ws
    if out is None:
        out = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    else:
        _chk(out, "nhwc_to_nchw.out")
        if tuple(out.shape) != (B, C, H, W) or not out.is_contiguous():
            raise _lib.FFError(f"nhwc_to_nchw: out must be a contiguous [{B},{C},{H},{W}] tensor")
    _lib.check(_L().ff_nhwc_to_nchw(xp, out.data_ptr(), B, C, H, W, Hs, Ws, ldi, _ptr(add), int(clamp01), _stream()))
    return out


def _aten_scale(n_in: int, n_out: int, scale_factor: Optional[float]) -> float:
    """ATen area_pixel_compute_scale (align_corners=False): 1/scale_factor if given else in/out, in fp32."""
    import numpy as np
    if scale_factor is not None and scale_factor > 0:
        return float(np.float32(1.0) / np.float32(scale_factor))
    return float(np.float32(n_in) / np.float32(n_out))


def resize(x: T, size: Tuple[int, int], *, mode: str = "bilinear", scale_factor: Optional[float] = None,
           layout: str = "nhwc", out: Optional[T] = None, mul: float = 1.0) -> T:
    """x NHWC rows view [B,H,W,C] or planar [B,C,H,W] (layout='nchw'); out is always NHWC (may be a channel slice)."""
    Ho, Wo = size
    if layout == "nhwc":
        xp, ld, B, Hi, Wi, C = _nhwc(x, "resize.x")
        isb, isc, isy, isx = Hi * Wi * ld, 1, Wi * ld, ld
    else:
        _chk(x, "resize.x")
        x = x.contiguous()
        B, C, Hi, Wi = x.shape
        xp = x.data_ptr()
        isb, isc, isy, isx = C * Hi * Wi, Hi * Wi, Wi, 1
    if out is None:
        out = torch.empty((B, Ho, Wo, C), device=x.device, dtype=torch.float32)
    op, ldo, ob, oh, ow, oc = _nhwc(out, "resize.out")      # out may be larger than (Ho, Wo): the top-left region is written
    if ob != B or oc != C or oh < Ho or ow < Wo:
        raise _lib.FFError("resize: out shape mismatch")
    sh = _aten_scale(Hi, Ho, scale_factor)
    sw = _aten_scale(Wi, Wo, scale_factor)
    _lib.check(_L().ff_resize(xp, isb, isc, isy, isx, Hi, Wi, op, oh * ow * ldo, 1, ow * ldo, ldo, Ho, Wo, B, C, sh, sw,
                              0 if mode == "bilinear" else 1, float(mul), _stream()))
    _note(0.0, 4.0 * B * C * (Hi * Wi + Ho * Wo))
    return out


def avgpool2(x: T) -> T:
    xp, ldi, B, H, W, C = _nhwc(x, "avgpool2.x")
    out = torch.empty((B, H // 2, W // 2, C), device=x.device, dtype=torch.float32)
    _lib.check(_L().ff_avgpool2(xp, ldi, out.data_ptr(), C, B, H, W, C, _stream()))
    return out


def dct8_bands(x_planar: T, dct_mat: T, masks3: T, band_scale: T, out: T, ch_off: int):
    _chk(x_planar, "dct8_bands.x")
    C, H, W = x_planar.shape
    op, ldo, *_ = _nhwc(out, "dct8_bands.out")
    _lib.check(_L().ff_dct8_bands(x_planar.data_ptr(), C, H, W, dct_mat.data_ptr(), masks3.data_ptr(), band_scale.data_ptr(),
                                  op, ldo, ch_off, _stream()))


def dwt_pass(x_planar: T, axis: int, lo8: T, hi8: T) -> Tuple[T, T]:
    _chk(x_planar, "dwt_pass.x")
    C, H, W = x_planar.shape
    Ho = (H + 6) // 2 + 1 if axis == 0 else H
    Wo = (W + 6) // 2 + 1 if axis == 1 else W
    lo = torch.empty((C, Ho, Wo), device=x_planar.device, dtype=torch.float32)
    hi = torch.empty_like(lo)
    _lib.check(_L().ff_dwt_pass(x_planar.data_ptr(), C, H, W, axis, lo8.data_ptr(), hi8.data_ptr(), lo.data_ptr(),
                                hi.data_ptr(), _stream()))
    return lo, hi


def fft_bands(x_planar: T, twW: Tuple[T, T], twH: Tuple[T, T], mask_logits: T, temp: float, band_scale2: T, out: T,
              ch_lo: int, ch_hi: int):
    _chk(x_planar, "fft_bands.x")
    C, H, W = x_planar.shape
    nwork = 4 * C * H * (W // 2 + 1)
    work = torch.empty(nwork, device=x_planar.device, dtype=torch.float32)
    op, ldo, *_ = _nhwc(out, "fft_bands.out")
    msz = mask_logits.shape[-1]
    _lib.check(_L().ff_fft_bands(x_planar.data_ptr(), C, H, W, twW[0].data_ptr(), twW[1].data_ptr(), twH[0].data_ptr(),
                                 twH[1].data_ptr(), mask_logits.data_ptr(), msz, float(temp), band_scale2.data_ptr(),
                                 work.data_ptr(), nwork, op, ldo, ch_lo, ch_hi, _stream()))


def chan_attn_weights(qkv: T, q_off: int, k_off: int, temperature: T) -> T:
    qp, ld, rows, C = rows_view(qkv, "chan_attn_weights.qkv")
    nwork = int(_L().ff_chan_attn_workspace(rows))
    work = torch.empty(nwork, device=qkv.device, dtype=torch.float32)
    wbd = torch.empty((180, 180), device=qkv.device, dtype=torch.float32)
    _lib.check(_L().ff_chan_attn_weights(qp, ld, q_off, k_off, rows, temperature.data_ptr(), wbd.data_ptr(), work.data_ptr(),
                                         nwork, _stream()))
    return wbd


def chan_qkv_attn(x: T, pk: dict, gamma: T, beta: T, temperature: T, eps: float = 1e-5):
    """DAT channel attention front end in two launches (ff_chan_qkv + ff_chan_attn_finish): x [..., 180] ->
    (v [..., 180], block-diagonal attention matrix [180, 180]); q and k never reach memory."""
    xp, ldx, rows, K = rows_view(x, "chan_qkv_attn.x")
    if K != pk["K"]:
        raise _lib.FFError("chan_qkv_attn: K mismatch")
    v = empty_like_rows(x)
    vp, ldv, _, _ = rows_view(v, "chan_qkv_attn.v")
    nwork = int(_L().ff_chan_qkv_workspace(rows))
    work = torch.empty(nwork, device=x.device, dtype=torch.float32)
    wbd = torch.empty((180, 180), device=x.device, dtype=torch.float32)
    _lib.check(_L().ff_chan_qkv(xp, ldx, rows, K, gamma.data_ptr(), beta.data_ptr(), float(eps), pk["w"].data_ptr(), pk["b"].data_ptr(),
                                vp, ldv, work.data_ptr(), nwork, _nterms(), _stream()))
    _lib.check(_L().ff_chan_attn_finish(work.data_ptr(), nwork, (rows + 255) // 256, temperature.data_ptr(), wbd.data_ptr(), _stream()))
    _note(2.0 * rows * 576 * K + 2.0 * rows * 6 * 32 * 32, 8.0 * rows * K)
    return v, wbd


def band_mha_core(qkv: T, P: int, nbands: int, heads: int) -> T:
    _chk(qkv, "band_mha_core.qkv")
    E = qkv.shape[-1] // 3
    out = torch.empty((P * nbands, E), device=qkv.device, dtype=torch.float32)
    _lib.check(_L().ff_band_mha_core(qkv.data_ptr(), out.data_ptr(), P, nbands, heads, _stream()))
    return out


def band_weight(x: T, att: T, imp: T) -> T:
    out = torch.empty_like(x)
    P = x.numel() // x.shape[-1]
    _lib.check(_L().ff_band_weight(x.data_ptr(), att.data_ptr(), imp.data_ptr(), out.data_ptr(), P, x.shape[-1] // 3, _stream()))
    return out


def freq_guidance(b3: T) -> T:
    P = b3.numel() // 9
    out = torch.empty(tuple(b3.shape[:-1]) + (3,), device=b3.device, dtype=torch.float32)
    _lib.check(_L().ff_freq_guidance(b3.data_ptr(), out.data_ptr(), P, _stream()))
    return out


def dynamic_gates(graw: T, dif: T) -> T:
    P = graw.numel() // 3
    out = torch.empty_like(graw)
    _lib.check(_L().ff_dynamic_gates(graw.data_ptr(), dif.data_ptr(), out.data_ptr(), P, _stream()))
    return out


def fuse_blend(experts9: T, hier3: T, guide3: T, gates3: T, dif1: T) -> T:
    _, Hh, Wh, _ = experts9.shape
    _, Hl, Wl, _ = guide3.shape
    out = torch.empty((1, Hh, Wh, 3), device=experts9.device, dtype=torch.float32)
    _lib.check(_L().ff_fuse_blend(experts9.data_ptr(), hier3.data_ptr(), guide3.data_ptr(), gates3.data_ptr(), dif1.data_ptr(),
                                  out.data_ptr(), Hh, Wh, Hl, Wl, _stream()))
    return out


def tile_accum(tile: T, wy: T, wx: T, acc: T, wsum: T, sy: int, sx: int):
    """acc [1,C,H,W] += tile [1,C,th,tw] * (wy x wx) at (sy, sx); wsum [1,1,H,W] += wy x wx."""
    for t_, n_ in ((tile, "tile"), (wy, "wy"), (wx, "wx"), (acc, "acc"), (wsum, "wsum")):
        _chk(t_, "tile_accum." + n_)
    _, C, th, tw = tile.shape
    _, _, H, W = acc.shape
    _lib.check(_L().ff_tile_accum(tile.contiguous().data_ptr(), C, th, tw, wy.data_ptr(), wx.data_ptr(), acc.data_ptr(),
                                  wsum.data_ptr(), H, W, sy, sx, _stream()))


def tile_normalize(acc: T, wsum: T):
    _, C, H, W = acc.shape
    _lib.check(_L().ff_tile_normalize(acc.data_ptr(), wsum.data_ptr(), C, H, W, _stream()))


for _n in ("conv2d", "cab_fused", "sgfn_tail", "ocab_attn", "linear", "win_attn_fused", "token_projmlp", "token_linear_gated", "token_mlp", "token_linear", "pixel_mlp", "dwconv3_gate_pool", "naf_front", "naf_ffn", "window_attn", "layernorm", "pool_mean", "vec_mlp", "dwconv2d", "dwconv3x3_ln", "mix2", "fma3", "affine",
           "nchw_to_nhwc", "nhwc_to_nchw", "resize", "avgpool2", "dct8_bands", "dwt_pass", "fft_bands", "chan_attn_weights", "chan_qkv_attn",
           "band_mha_core", "band_weight", "freq_guidance", "dynamic_gates", "fuse_blend", "tile_accum", "tile_normalize"):
    globals()[_n] = _instrument(globals()[_n])
