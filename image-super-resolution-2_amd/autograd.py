"""A minimal reverse-mode tape over the HIP kernels: what torch.autograd does for the reference's fusion-only training step
(train.py:308-356, `loss.backward()`), rebuilt on this library's own kernels so that no ATen compute op runs in a step.

    with Tape() as tape:
        y = conv2d(x, W, b, ksize=(3, 3), act="gelu") ...        # every op appends its backward closure
        tape.backward(loss_grad_of=y, grad=dy)

A `Var` is a device tensor (dense rows [..., C], or a channel slice of one: last dim contiguous, constant row pitch) plus an
optional gradient.  Gradients are accumulated (`_acc`); parameter Vars carry a pre-allocated gradient view into the trainer's flat
gradient buffer, zeroed once per step.  All compute goes through include/ff_kernels.h entry points; torch supplies memory only.
Every reduction is two-stage in a fixed order: a step is bit-reproducible.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import lib as _lib
from . import ops

T = torch.Tensor
K_FULL, K_ROW, K_GC, K_SCALAR, K_NONE = 0, 1, 2, 3, 4
U = dict(gelu=0, relu=1, sigmoid=2, softplus=3, abs=4, clamp01=5, scale=6, recip_eps=7, clamp_min=8, exp=9, exp_bwd_y=24,
         gelu_bwd=16, relu_bwd=17, sigmoid_bwd_y=18, softplus_bwd=19, abs_bwd=20, clamp01_bwd=21, recip_bwd=22, clamp_min_bwd=23)


def _L():
    return _lib.load()


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def _rv(t: T):
    return ops.rows_view(t, "autograd")


class Var:
    __slots__ = ("data", "grad", "req", "owned", "name")

    def __init__(self, data: T, req: bool = False, grad: Optional[T] = None, name: str = ""):
        self.data, self.req, self.grad, self.owned, self.name = data, req, grad, grad is not None, name

    @property
    def shape(self):
        return tuple(self.data.shape)

    def __repr__(self):
        return f"Var({self.name or ''}{tuple(self.data.shape)}, req={self.req})"


class Tape:
    """Backward closures in forward order; backward() runs them reversed."""
    current: Optional["Tape"] = None

    def __init__(self):
        self.fns: List[Callable[[], None]] = []

    def __enter__(self):
        self._prev = Tape.current
        Tape.current = self
        return self

    def __exit__(self, *exc):
        Tape.current = self._prev

    def backward(self):
        for fn in reversed(self.fns):
            fn()
        self.fns.clear()


def _rec(fn: Callable[[], None], *inputs: Var):
    if Tape.current is not None and any(isinstance(v, Var) and v.req for v in inputs):
        Tape.current.fns.append(fn)
        return True
    return False


def const(t: T) -> Var:
    return Var(t, False)


# ------------------------------------------------------------------------------------------------------ raw kernel calls
def k_fma(a: Optional[T], b: Optional[T] = None, b_kind: int = K_NONE, c: Optional[T] = None, c_kind: int = K_NONE, c_scale: float = 1.0,
          out: Optional[T] = None, rpg: int = 0, clamp01: bool = False, like: Optional[T] = None) -> T:
    """out = a*b + c_scale*c with broadcast operands (ff_ew_fma); a None = 1."""
    ref = a if a is not None else (like if like is not None else out)
    if out is None:
        out = torch.empty(tuple(ref.shape), device=ref.device, dtype=torch.float32)
    op, ldo, rows, C = _rv(out)
    ap, lda = (None, 0)
    if a is not None:
        ap, lda, ar, ac = _rv(a)
        if (ar, ac) != (rows, C):
            raise _lib.FFError(f"k_fma: a {tuple(a.shape)} vs out {tuple(out.shape)}")

    def operand(t, kind):
        if kind == K_NONE or t is None:
            return None, 0, K_NONE
        if kind == K_FULL:
            p, ld, r, c_ = _rv(t)
            if (r, c_) != (rows, C):
                raise _lib.FFError(f"k_fma: full operand {tuple(t.shape)} vs out {tuple(out.shape)}")
            return p, ld, kind
        if kind == K_ROW:
            p, ld, r, c_ = _rv(t)
            if r != rows or c_ != 1:
                raise _lib.FFError(f"k_fma: per-row operand must be [rows,1], got {tuple(t.shape)} for {rows} rows")
            return p, ld, kind
        if kind == K_GC:
            if not t.is_contiguous() or t.shape[-1] != C or rpg <= 0 or t.numel() != (rows // rpg) * C:
                raise _lib.FFError(f"k_fma: group operand must be contiguous [rows/rpg, C], got {tuple(t.shape)}")
            return t.data_ptr(), C, kind
        if t.numel() != 1:
            raise _lib.FFError("k_fma: scalar operand must have one element")
        return t.data_ptr(), 1, kind
    bp, ldb, bk = operand(b, b_kind)
    cp, ldc, ck = operand(c, c_kind)
    _lib.check(_L().ff_ew_fma(op, ldo, ap, lda, bp, ldb, bk, cp, ldc, ck, float(c_scale), rows, C, int(rpg), int(clamp01), _st()))
    return out


def k_unary(op: str, x: T, g: Optional[T] = None, p0: float = 0.0, out: Optional[T] = None) -> T:
    if out is None:
        out = torch.empty(tuple(x.shape), device=x.device, dtype=torch.float32)
    xp, ldx, rows, C = _rv(x)
    op_, ldo, _, _ = _rv(out)
    gp, ldg = (None, 0)
    if g is not None:
        gp, ldg, gr, gc = _rv(g)
        if (gr, gc) != (rows, C):
            raise _lib.FFError("k_unary: gradient shape mismatch")
    _lib.check(_L().ff_ew_unary(U[op], xp, ldx, gp, ldg, op_, ldo, rows, C, float(p0), _st()))
    return out


def k_reduce_cols(x: T, y: Optional[T] = None, y_kind: int = K_NONE, rpg: int = 0, scale: float = 1.0, out: Optional[T] = None,
                  accumulate: bool = False) -> T:
    """[G, C] = scale * sum over the rows of each group of x (* y)."""
    xp, ldx, rows, C = _rv(x)
    rpg = rpg or rows
    G = rows // rpg
    if out is None:
        out = torch.empty((G, C), device=x.device, dtype=torch.float32)
        accumulate = False
    if out.numel() != G * C or not out.is_contiguous():
        raise _lib.FFError("k_reduce_cols: out must be contiguous [G, C]")
    yp, ldy, yk = None, 0, 4
    if y is not None:
        yp, ldy, yr, yc = _rv(y)
        yk = 0 if y_kind in (K_FULL, K_NONE) else 1
        if yr != rows or (yk == 0 and yc != C) or (yk == 1 and yc != 1):
            raise _lib.FFError("k_reduce_cols: y shape mismatch")
    nw = int(_L().ff_reduce_cols_workspace(rows, C, rpg))
    work = torch.empty(max(nw, 1), device=x.device, dtype=torch.float32)
    _lib.check(_L().ff_reduce_cols(xp, ldx, yp, ldy, yk, rows, C, rpg, float(scale), out.data_ptr(), int(accumulate), work.data_ptr(), nw, _st()))
    return out


def k_reduce_rows(x: T, y: Optional[T] = None, y_kind: int = K_NONE, scale: float = 1.0, out: Optional[T] = None, accumulate: bool = False) -> T:
    """[..., 1] = scale * sum over channels of x (* y full, or * y per column)."""
    xp, ldx, rows, C = _rv(x)
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (1,), device=x.device, dtype=torch.float32)
        accumulate = False
    op, ldo, orows, oc = _rv(out)
    if orows != rows or oc != 1:
        raise _lib.FFError("k_reduce_rows: out must be [rows, 1]")
    yp, ldy, yk = None, 0, 4
    if y is not None:
        if y_kind == K_GC or (y.dim() == 1 and y.numel() == C):
            yp, ldy, yk = y.data_ptr(), 0, 2
        else:
            yp, ldy, yr, yc = _rv(y)
            yk = 0
            if (yr, yc) != (rows, C):
                raise _lib.FFError("k_reduce_rows: y shape mismatch")
    _lib.check(_L().ff_reduce_rows(xp, ldx, yp, ldy, yk, rows, C, float(scale), op, ldo, int(accumulate), _st()))
    return out


def k_sum_all(x: T, y: Optional[T] = None, scale: float = 1.0, out: Optional[T] = None, accumulate: bool = False) -> T:
    """[1] = scale * sum(x * y) over everything (fixed order: columns first, then the C partials)."""
    cols = k_reduce_cols(x, y, K_FULL if y is not None else K_NONE)                  # [1, C]
    if out is None:
        out = torch.empty(1, device=x.device, dtype=torch.float32)
        accumulate = False
    return k_reduce_rows(cols, scale=scale, out=out.reshape(1, 1), accumulate=accumulate).reshape(1)


def _acc(v: Var, g: T, owned: bool = True):
    """v.grad += g.  `owned`: g is a fresh tensor nobody else refers to (it may become v.grad itself)."""
    if not v.req:
        return
    if v.grad is None:
        v.grad, v.owned = g, owned
    elif v.owned:
        k_fma(v.grad, c=g, c_kind=K_FULL, out=v.grad)
    else:
        v.grad, v.owned = k_fma(v.grad, c=g, c_kind=K_FULL), True


def _grad_buf(v: Var) -> T:
    """A dense, owned, zero-initialised gradient buffer for v (slices accumulate into it)."""
    if v.grad is None:
        v.grad, v.owned = torch.zeros(tuple(v.data.shape), device=v.data.device, dtype=torch.float32), True
    elif not v.owned:
        v.grad, v.owned = k_fma(v.grad), True                                       # private copy
    return v.grad


# ------------------------------------------------------------------------------------------------------ pointwise ops
def unary(op: str, x: Var, p0: float = 0.0) -> Var:
    """gelu / relu / sigmoid / softplus / abs / clamp01 / recip_eps (1/(x+p0)) / clamp_min (max(x, p0))."""
    y = Var(k_unary(op, x.data, p0=p0), x.req)

    def bwd():
        if y.grad is None:
            return
        if op == "sigmoid":
            g = k_unary("sigmoid_bwd_y", y.data, y.grad)
        elif op == "recip_eps":
            g = k_unary("recip_bwd", y.data, y.grad)
        elif op == "exp":
            g = k_unary("exp_bwd_y", y.data, y.grad)
        else:
            g = k_unary(op + "_bwd", x.data, y.grad, p0=p0)
        _acc(x, g)
    _rec(bwd, x)
    return y


def scale(x: Var, k: float) -> Var:
    y = Var(k_unary("scale", x.data, p0=k), x.req)

    def bwd():
        if y.grad is not None:
            _acc(x, k_unary("scale", y.grad, p0=k))
    _rec(bwd, x)
    return y


def add(a: Var, b: Var, sb: float = 1.0, clamp01: bool = False) -> Var:
    """a + sb * b (optionally clamped to [0,1]; the clamp's gradient mask is inclusive, as torch.clamp's)."""
    pre = k_fma(a.data, c=b.data, c_kind=K_FULL, c_scale=sb)
    y = Var(k_unary("clamp01", pre) if clamp01 else pre, a.req or b.req)

    def bwd():
        if y.grad is None:
            return
        g, fresh = (k_unary("clamp01_bwd", pre, y.grad), True) if clamp01 else (y.grad, y.owned)
        sole = not (a.req and b.req)                                  # one receiver: the tensor can be handed over
        if a.req:
            _acc(a, g, owned=fresh and sole)
        if b.req:
            if sb == 1.0:
                _acc(b, g, owned=fresh and sole)
            else:
                _acc(b, k_unary("scale", g, p0=sb))
    _rec(bwd, a, b)
    return y


def mul(a: Var, b: Var) -> Var:
    y = Var(k_fma(a.data, b.data, K_FULL), a.req or b.req)

    def bwd():
        if y.grad is None:
            return
        if a.req:
            _acc(a, k_fma(y.grad, b.data, K_FULL))
        if b.req:
            _acc(b, k_fma(y.grad, a.data, K_FULL))
    _rec(bwd, a, b)
    return y


def mul_row(a: Var, r: Var) -> Var:
    """a [rows, C] * r [rows, 1] (per-pixel gate)."""
    y = Var(k_fma(a.data, r.data, K_ROW), a.req or r.req)

    def bwd():
        if y.grad is None:
            return
        if a.req:
            _acc(a, k_fma(y.grad, r.data, K_ROW))
        if r.req:
            _acc(r, k_reduce_rows(y.grad, a.data))
    _rec(bwd, a, r)
    return y


def mul_scalar(a: Var, s: Var) -> Var:
    """a * s with s a one-element device parameter (LKABlock.scale1, residual_scale, ...)."""
    y = Var(k_fma(a.data, s.data, K_SCALAR), a.req or s.req)

    def bwd():
        if y.grad is None:
            return
        if a.req:
            _acc(a, k_fma(y.grad, s.data, K_SCALAR))
        if s.req:
            _acc(s, k_sum_all(y.grad, a.data).reshape(s.data.shape))
    _rec(bwd, a, s)
    return y


def mul_gc(a: Var, s: Var, rpg: int) -> Var:
    """a [rows, C] * s [rows/rpg, C] (one value per image group and channel)."""
    y = Var(k_fma(a.data, s.data, K_GC, rpg=rpg), a.req or s.req)

    def bwd():
        if y.grad is None:
            return
        if a.req:
            _acc(a, k_fma(y.grad, s.data, K_GC, rpg=rpg))
        if s.req:
            _acc(s, k_reduce_cols(y.grad, a.data, K_FULL, rpg=rpg).reshape(s.data.shape))
    _rec(bwd, a, s)
    return y


def sum_ch(x: Var, k: float = 1.0) -> Var:
    """[..., C] -> [..., 1]: k * sum over channels."""
    y = Var(k_reduce_rows(x.data, scale=k), x.req)

    def bwd():
        if y.grad is not None:
            g = torch.empty(tuple(x.data.shape), device=x.data.device, dtype=torch.float32)
            _acc(x, k_fma(None, y.grad, K_ROW, out=g) if k == 1.0 else k_unary("scale", k_fma(None, y.grad, K_ROW, out=g), p0=k))
    _rec(bwd, x)
    return y


def slice_ch(x: Var, a: int, b: int) -> Var:
    """Channel slice x[..., a:b] (a view; its gradient lands in the same slice of x's gradient)."""
    y = Var(x.data[..., a:b], x.req)

    def bwd():
        if y.grad is None:
            return
        gb = _grad_buf(x)
        k_fma(gb[..., a:b], c=y.grad, c_kind=K_FULL, out=gb[..., a:b])
    _rec(bwd, x)
    return y


def cat_ch(xs: Sequence[Var], pad_to: int = 0) -> Var:
    """Concatenate along channels (optionally zero-padded to pad_to channels)."""
    C = sum(v.data.shape[-1] for v in xs)
    Ct = max(C, pad_to)
    lead = tuple(xs[0].data.shape[:-1])
    out = (torch.zeros if Ct > C else torch.empty)(lead + (Ct,), device=xs[0].data.device, dtype=torch.float32)
    o = 0
    offs = []
    for v in xs:
        c = v.data.shape[-1]
        k_fma(v.data, out=out[..., o:o + c])
        offs.append((o, c))
        o += c
    y = Var(out, any(v.req for v in xs))

    def bwd():
        if y.grad is None:
            return
        for v, (o_, c_) in zip(xs, offs):
            if v.req:
                _acc(v, y.grad[..., o_:o_ + c_], owned=False)
    _rec(bwd, *xs)
    return y


def reshape(x: Var, shape) -> Var:
    """Reinterpret a DENSE tensor's shape (free)."""
    if not x.data.is_contiguous():
        raise _lib.FFError("reshape: needs a dense tensor")
    y = Var(x.data.reshape(shape), x.req)

    def bwd():
        if y.grad is not None:
            g = y.grad if y.grad.is_contiguous() else k_fma(y.grad)
            _acc(x, g.reshape(x.data.shape), owned=y.owned and g is y.grad or g is not y.grad)
    _rec(bwd, x)
    return y


def permute_rows(x: Var, A: int, Bd: int, C: int) -> Var:
    """[A][B][C] -> [B][A][C] on a dense tensor."""
    out = torch.empty((Bd, A, C), device=x.data.device, dtype=torch.float32)
    _lib.check(_L().ff_permute_rows(x.data.data_ptr(), out.data_ptr(), A, Bd, C, _st()))
    y = Var(out, x.req)

    def bwd():
        if y.grad is not None:
            g = y.grad if y.grad.is_contiguous() else k_fma(y.grad)
            gi = torch.empty((A, Bd, C), device=g.device, dtype=torch.float32)
            _lib.check(_L().ff_permute_rows(g.data_ptr(), gi.data_ptr(), Bd, A, C, _st()))
            _acc(x, gi.reshape(x.data.shape))
    _rec(bwd, x)
    return y


# ------------------------------------------------------------------------------------------------------ structured ops
def _flipT(w: T, Cout: int, Cin: int, KH: int, KW: int) -> T:
    wt = torch.empty((Cin, KH * KW * Cout), device=w.device, dtype=torch.float32)
    _lib.check(_L().ff_conv_weight_flipT(w.data_ptr(), wt.data_ptr(), Cout, Cin, KH, KW, _st()))
    return wt


def conv2d(x: Var, w: Var, bias: Optional[Var] = None, ksize=(1, 1), act: Optional[str] = None) -> Var:
    """nn.Conv2d(stride 1, 'same' padding) / nn.Linear on rows: x [B,H,W,Cin] (or [rows,Cin] for 1x1), w packed [Cout, KH*KW*Cin]."""
    KH, KW = ksize
    pad = (KH // 2, KW // 2)
    x4 = x.data if x.data.dim() == 4 else x.data.reshape(1, 1, -1, x.data.shape[-1]) if x.data.is_contiguous() else None
    if x4 is None:                                               # strided 2-D rows view: ops.linear addresses it in place
        if (KH, KW) != (1, 1):
            raise _lib.FFError("conv2d: a spatial kernel needs a [B,H,W,C] input")
        z = ops.linear(x.data, w.data, bias.data if bias is not None else None, dynamic_w=True)
    else:
        z = ops.conv2d(x4, w.data, bias.data if bias is not None else None, ksize=ksize, pad=pad, dynamic_w=True)
        if x.data.dim() != 4:
            z = z.reshape(tuple(x.data.shape[:-1]) + (w.data.shape[0],))
    yv = k_unary(act, z) if act else z
    y = Var(yv, x.req or w.req or (bias is not None and bias.req))
    Cout, Cin = w.data.shape[0], x.data.shape[-1]

    def bwd():
        if y.grad is None:
            return
        if act == "sigmoid":
            gz = k_unary("sigmoid_bwd_y", yv, y.grad)
        elif act:
            gz = k_unary(act + "_bwd", z, y.grad)
        else:
            gz = y.grad
        gzp, ldz, grows, _ = _rv(gz)
        if bias is not None and bias.req:
            k_reduce_cols(gz, out=_grad_buf(bias).reshape(1, Cout), accumulate=True)
        if w.req:
            xp, ldx, xrows, _ = _rv(x.data)
            if x.data.dim() == 4:
                B_, H_, W_ = x.data.shape[:3]
            else:
                B_, H_, W_ = 1, 1, xrows
            nw = int(_L().ff_conv2d_wgrad_workspace(B_, H_, W_, Cin, Cout, KH, KW))
            work = torch.empty(nw, device=gz.device, dtype=torch.float32)
            _lib.check(_L().ff_conv2d_wgrad(xp, ldx, gzp, ldz, _grad_buf(w).data_ptr(), B_, H_, W_, Cin, Cout, KH, KW, pad[0], pad[1], 1,
                                            work.data_ptr(), nw, _st()))
        if x.req:
            wt = _flipT(w.data, Cout, Cin, KH, KW)
            if x.data.dim() == 4:
                gx = ops.conv2d(gz, wt, None, ksize=ksize, pad=pad, dynamic_w=True)
            else:
                gx = ops.linear(gz, wt, None, dynamic_w=True)
            _acc(x, gx)
    _rec(bwd, x, w, bias) if bias is not None else _rec(bwd, x, w)
    return y


def dwconv2d(x: Var, w: Var, ksize=(3, 3)) -> Var:
    """Depth-wise convolution, stride 1, 'same' zero padding, tap-major weights [KH*KW, C], no bias (the LKA chain)."""
    KH, KW = ksize
    pad = (KH // 2, KW // 2)
    y = Var(ops.dwconv2d(x.data, w.data, None, ksize=ksize, pad=pad), x.req or w.req)
    B, H, W_, C = x.data.shape

    def bwd():
        if y.grad is None:
            return
        g = y.grad
        gp, ldg, _, _ = _rv(g)
        if w.req:
            xp, ldx, _, _ = _rv(x.data)
            nw = int(_L().ff_dwconv2d_wgrad_workspace(B, H, W_, C, KH, KW))
            work = torch.empty(nw, device=g.device, dtype=torch.float32)
            _lib.check(_L().ff_dwconv2d_wgrad(xp, ldx, gp, ldg, _grad_buf(w).data_ptr(), B, H, W_, C, KH, KW, 1, work.data_ptr(), nw, _st()))
        if x.req:
            wr = torch.empty_like(w.data)
            _lib.check(_L().ff_taps_reverse(w.data.data_ptr(), wr.data_ptr(), KH * KW, C, _st()))
            _acc(x, ops.dwconv2d(g.reshape(B, H, W_, C) if g.dim() != 4 else g, wr, None, ksize=ksize, pad=pad))
    _rec(bwd, x, w)
    return y


def resize(x: Var, size: Tuple[int, int], scale_factor: Optional[float] = None) -> Var:
    """F.interpolate(mode='bilinear', align_corners=False) on NHWC."""
    B, Hi, Wi, C = x.data.shape
    Ho, Wo = size
    y = Var(ops.resize(x.data, size, scale_factor=scale_factor), x.req)

    def bwd():
        if y.grad is None:
            return
        gp, ldg, _, _ = _rv(y.grad)
        gx = torch.empty((B, Hi, Wi, C), device=x.data.device, dtype=torch.float32)
        _lib.check(_L().ff_resize_bilinear_adj(gp, ldg, Ho, Wo, gx.data_ptr(), C, Hi, Wi, B, C, ops._aten_scale(Hi, Ho, scale_factor),
                                               ops._aten_scale(Wi, Wo, scale_factor), 1.0, 0, _st()))
        _acc(x, gx)
    _rec(bwd, x)
    return y


def avgpool2(x: Var) -> Var:
    B, H, W_, C = x.data.shape
    y = Var(ops.avgpool2(x.data), x.req)

    def bwd():
        if y.grad is None:
            return
        gp, ldg, _, _ = _rv(y.grad)
        gx = torch.empty((B, H, W_, C), device=x.data.device, dtype=torch.float32)
        _lib.check(_L().ff_avgpool2_adj(gp, ldg, gx.data_ptr(), C, B, H, W_, C, 0, _st()))
        _acc(x, gx)
    _rec(bwd, x)
    return y


def layernorm(x: Var, gamma: Var, beta: Var, eps: float = 1e-5) -> Var:
    y = Var(ops.layernorm(x.data, gamma.data, beta.data, eps), x.req or gamma.req)
    C = x.data.shape[-1]

    def bwd():
        if y.grad is None:
            return
        xp, ldx, rows, _ = _rv(x.data)
        gp, ldg, _, _ = _rv(y.grad)
        gx = torch.empty(tuple(x.data.shape), device=x.data.device, dtype=torch.float32)
        gb = torch.empty((2, C), device=x.data.device, dtype=torch.float32)
        nw = int(_L().ff_layernorm_bwd_workspace(rows, C))
        work = torch.empty(nw, device=x.data.device, dtype=torch.float32)
        _lib.check(_L().ff_layernorm_bwd(xp, ldx, gp, ldg, gamma.data.data_ptr(), float(eps), gx.data_ptr(), C, rows, C, gb.data_ptr(), 0,
                                         work.data_ptr(), nw, _st()))
        _acc(x, gx)
        _acc(gamma, gb[0].reshape(gamma.data.shape), owned=False)
        _acc(beta, gb[1].reshape(beta.data.shape), owned=False)
    _rec(bwd, x, gamma, beta)
    return y


def batchnorm_train(x: Var, gamma: Var, beta: Var, running_mean: Optional[T], running_var: Optional[T], groups: int = 1,
                    eps: float = 1e-5, momentum: float = 0.1) -> Var:
    """nn.BatchNorm2d in training mode over `groups` consecutive row groups of x [rows, C] (one module CALL per group, in order):
    batch statistics per group, running statistics updated call by call (unbiased variance), two-pass variance."""
    xp, ldx, rows, C = _rv(x.data)
    rpg = rows // groups
    dev = x.data.device
    s1 = k_reduce_cols(x.data, rpg=rpg)                                               # [G, C]
    mean = k_unary("scale", s1, p0=1.0 / rpg)
    xc = k_fma(x.data, c=mean, c_kind=K_GC, c_scale=-1.0, rpg=rpg)
    s2 = k_reduce_cols(xc, xc, K_FULL, rpg=rpg)
    mr = torch.empty((groups, C, 2), device=dev, dtype=torch.float32)
    sc = torch.empty((groups, C), device=dev, dtype=torch.float32)
    sh = torch.empty((groups, C), device=dev, dtype=torch.float32)
    _lib.check(_L().ff_bn_train_finish(s1.data_ptr(), s2.data_ptr(), 1, groups, C, rpg, gamma.data.data_ptr(), beta.data.data_ptr(), float(eps),
                                       float(momentum), running_mean.data_ptr() if running_mean is not None else None,
                                       running_var.data_ptr() if running_var is not None else None, mr.data_ptr(), sc.data_ptr(), sh.data_ptr(), _st()))
    y = Var(k_fma(x.data, sc, K_GC, sh, K_GC, rpg=rpg), x.req or gamma.req)

    def bwd():
        if y.grad is None:
            return
        g = y.grad
        gp, ldg, _, _ = _rv(g)
        xh = torch.empty(tuple(x.data.shape), device=dev, dtype=torch.float32)
        _lib.check(_L().ff_bn_xhat(xp, ldx, mr.data_ptr(), xh.data_ptr(), C, rows, C, rpg, _st()))
        sdy = k_reduce_cols(g, rpg=rpg)
        sdyxh = k_reduce_cols(g, xh, K_FULL, rpg=rpg)
        if gamma.req:
            _acc(gamma, k_reduce_cols(sdyxh).reshape(gamma.data.shape))              # sum over the groups (shared module)
            _acc(beta, k_reduce_cols(sdy).reshape(beta.data.shape))
        if x.req:
            gx = torch.empty(tuple(x.data.shape), device=dev, dtype=torch.float32)
            _lib.check(_L().ff_bn_train_bwd(xp, ldx, gp, ldg, mr.data_ptr(), gamma.data.data_ptr(), sdy.data_ptr(), sdyxh.data_ptr(),
                                            gx.data_ptr(), C, rows, C, rpg, _st()))
            _acc(x, gx)
    _rec(bwd, x, gamma, beta)
    return y


def band_mha(qkv: Var, P: int, ntok: int, heads: int, drop_p: float = 0.0, seed: int = 0) -> Var:
    """The per-pixel attention core of nn.MultiheadAttention over ntok tokens (9 bands / 3 experts), d = 16 per head."""
    E = qkv.data.shape[-1] // 3
    out = torch.empty((P * ntok, E), device=qkv.data.device, dtype=torch.float32)
    _lib.check(_L().ff_band_mha_train(qkv.data.data_ptr(), out.data_ptr(), P, ntok, heads, float(drop_p), int(seed), _st()))
    y = Var(out, qkv.req)

    def bwd():
        if y.grad is None:
            return
        g = y.grad if y.grad.is_contiguous() else k_fma(y.grad)
        dq = torch.empty(tuple(qkv.data.shape), device=g.device, dtype=torch.float32)
        nw = int(_L().ff_band_mha_bwd_workspace(P, ntok, heads))
        work = torch.empty(nw, device=g.device, dtype=torch.float32)
        _lib.check(_L().ff_band_mha_bwd(qkv.data.data_ptr(), g.data_ptr(), dq.data_ptr(), P, ntok, heads, float(drop_p), int(seed),
                                        work.data_ptr(), nw, _st()))
        _acc(qkv, dq)
    _rec(bwd, qkv)
    return y


def dynamic_gates(graw: Var, dif: Var) -> Var:
    """fusion_network.py:226-234 on [P,3] sigmoid gates and the [P,1] difficulty map."""
    y = Var(ops.dynamic_gates(graw.data, dif.data), graw.req or dif.req)
    P = graw.data.numel() // 3

    def bwd():
        if y.grad is None:
            return
        g = y.grad if y.grad.is_contiguous() else k_fma(y.grad)
        dg = torch.empty(tuple(graw.data.shape), device=g.device, dtype=torch.float32)
        dd = torch.empty(tuple(dif.data.shape), device=g.device, dtype=torch.float32)
        _lib.check(_L().ff_dynamic_gates_bwd(graw.data.data_ptr(), dif.data.data_ptr(), g.data_ptr(), dg.data_ptr(), dd.data_ptr(), P, _st()))
        _acc(graw, dg)
        _acc(dif, dd)
    _rec(bwd, graw, dif)
    return y


def mean_pool(x: Var) -> Var:
    """AdaptiveAvgPool2d(1): [B,H,W,C] -> [B,C]."""
    B, H, W_, C = x.data.shape
    rpg = H * W_
    y = Var(k_reduce_cols(x.data, rpg=rpg, scale=1.0 / rpg), x.req)

    def bwd():
        if y.grad is not None:
            g = y.grad if y.grad.is_contiguous() else k_fma(y.grad)
            gx = torch.empty(tuple(x.data.shape), device=x.data.device, dtype=torch.float32)
            k_fma(None, g, K_GC, out=gx, rpg=rpg)
            _acc(x, k_unary("scale", gx, p0=1.0 / rpg, out=gx))
    _rec(bwd, x)
    return y


# ------------------------------------------------------------------------------------------------------ learnable FFT mask
class FFT2:
    """rfft2 / irfft2 (norm='ortho') of planar real images [planes, H, W] by the library's DFT kernels; twiddles cached per size."""

    def __init__(self, dev):
        self.dev = dev
        self._tw = {}

    def tw(self, n: int):
        import numpy as np
        if n not in self._tw:
            ang = 2.0 * np.pi * np.arange(n, dtype=np.float64) / n
            self._tw[n] = (torch.from_numpy(np.cos(ang).astype(np.float32)).to(self.dev), torch.from_numpy(np.sin(ang).astype(np.float32)).to(self.dev))
        return self._tw[n]

    def rfft2(self, x: T) -> T:
        C, H, W_ = x.shape
        Wf = W_ // 2 + 1
        spec = torch.empty((C, H, Wf, 2), device=x.device, dtype=torch.float32)
        work = torch.empty(C * H * Wf * 2, device=x.device, dtype=torch.float32)
        (wc, ws), (hc, hs) = self.tw(W_), self.tw(H)
        _lib.check(_L().ff_rfft2(x.data_ptr(), C, H, W_, wc.data_ptr(), ws.data_ptr(), hc.data_ptr(), hs.data_ptr(), work.data_ptr(), work.numel(),
                                 spec.data_ptr(), _st()))
        return spec

    def irfft2(self, spec: T, W_: int) -> T:
        C, H, Wf, _ = spec.shape
        out = torch.empty((C, H, W_), device=spec.device, dtype=torch.float32)
        work = torch.empty(C * H * Wf * 2, device=spec.device, dtype=torch.float32)
        (wc, ws), (hc, hs) = self.tw(W_), self.tw(H)
        _lib.check(_L().ff_irfft2(spec.data_ptr(), C, H, W_, wc.data_ptr(), ws.data_ptr(), hc.data_ptr(), hs.data_ptr(), work.data_ptr(), work.numel(),
                                  out.data_ptr(), _st()))
        return out


def fft_lowpass(fft: FFT2, x_planes: T, X: T, m: Var) -> Var:
    """irfft2(rfft2(x) * m) for a real mask m [H, Wf] shared by all planes (multi_domain_frequency.py:362-376).  x is an input
    (no gradient); dm = sum_planes Re(rfft2(g) conj X) with the conjugate-paired columns doubled (torch's c2r backward)."""
    C, H, W_ = x_planes.shape
    Wf = W_ // 2 + 1
    Y = torch.empty_like(X)
    _lib.check(_L().ff_spec_mask_mul(X.data_ptr(), m.data.data_ptr(), Y.data_ptr(), C, H, Wf, _st()))
    y = Var(fft.irfft2(Y, W_), m.req)

    def bwd():
        if y.grad is None:
            return
        g = y.grad if y.grad.is_contiguous() else k_fma(y.grad)
        gY = fft.rfft2(g.reshape(C, H, W_))
        dm = torch.empty((H, Wf), device=g.device, dtype=torch.float32)
        _lib.check(_L().ff_spec_mask_grad(gY.data_ptr(), X.data_ptr(), dm.data_ptr(), C, H, W_, 0, _st()))
        _acc(m, dm.reshape(m.data.shape))
    _rec(bwd, m)
    return y


def slice_rows(x: Var, a: int, b: int) -> Var:
    """x[a:b] along the first axis of a dense tensor (a view; the gradient lands in the same rows of x's gradient)."""
    y = Var(x.data[a:b], x.req)

    def bwd():
        if y.grad is None:
            return
        gb = _grad_buf(x)
        k_fma(gb[a:b], c=y.grad, c_kind=K_FULL, out=gb[a:b])
    _rec(bwd, x)
    return y


def add_const(x: Var, c: float, one: T) -> Var:
    """x + c (`one`: a one-element device tensor holding 1.0)."""
    y = Var(k_fma(x.data, c=one, c_kind=K_SCALAR, c_scale=c), x.req)

    def bwd():
        if y.grad is not None:
            _acc(x, y.grad, owned=y.owned)
    _rec(bwd, x)
    return y


def planes_to_nhwc(x: Var, B: int) -> Var:
    """planar [B*C, H, W] -> NHWC [B, H, W, C]."""
    BC, H, W_ = x.data.shape
    C = BC // B
    y = Var(ops.nchw_to_nhwc(x.data.reshape(B, C, H, W_)), x.req)

    def bwd():
        if y.grad is not None:
            g = y.grad if y.grad.dim() == 4 else y.grad.reshape(B, H, W_, C)
            _acc(x, ops.nhwc_to_nchw(g).reshape(BC, H, W_))
    _rec(bwd, x)
    return y
