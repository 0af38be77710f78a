"""Host-side, one-off weight preparation (runs at model load, never in the timed path):
repacking torch conv weights for the NHWC implicit-GEMM kernel, folding eval-mode BatchNorm,
expanding relative-position tables into per-layer bias tensors, DAT's DynamicPosBias MLP."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

T = torch.Tensor


def pack_conv(w: T) -> T:
    """[Cout, Cin, KH, KW] -> [Cout, KH*KW*Cin] with k = (ky*KW + kx)*Cin + ci."""
    co, ci, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).contiguous()


def pack_dw(w: T) -> T:
    """depth-wise [C, 1, KH, KW] -> tap-major [KH*KW, C]."""
    c, _, kh, kw = w.shape
    return w.reshape(c, kh * kw).t().contiguous()


def bn_scale_shift(sd: Dict[str, T], p: str, eps: float = 1e-5) -> Tuple[T, T]:
    """eval BatchNorm as y = x*scale + shift."""
    scale = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + eps)
    shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return scale.contiguous(), shift.contiguous()


def fold_bn_after_conv(w2d: T, b: Optional[T], scale: T, shift: T) -> Tuple[T, T]:
    """BN(conv(x)) with packed weight [Cout, K]: -> (w', b')."""
    w = w2d * scale[:, None]
    bb = shift if b is None else b * scale + shift
    return w.contiguous(), bb.contiguous()
