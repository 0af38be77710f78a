"""Host sequencing of the fusion stack (everything after the three experts) on the HIP kernels.

Mirrors, stage by stage, the reference's eval path src/models/enhanced_fusion.py:418-429 (bands),
:502-591 (fuse), :593-647 (dynamic selection), :653-688 (refine + edge); the modules it calls are
cited at each class.  Tensors are NHWC fp32 on the GPU; B = 1 (one image / tile per call).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
from .prep import pack_conv, pack_dw, bn_scale_shift, fold_bn_after_conv

T = torch.Tensor
SD = Dict[str, T]

DB4_LO = [-0.010597401784997278, 0.032883011666982945, 0.030841381835986965, -0.18703481171888114,
          -0.027983769416983849, 0.63088076792959036, 0.71484657055291582, 0.23037781330885523]
DB4_HI = [-0.23037781330885523, 0.71484657055291582, -0.63088076792959036, -0.027983769416983849,
          0.18703481171888114, 0.030841381835986965, -0.032883011666982945, -0.010597401784997278]


def _dct_matrix(n: int = 8) -> T:
    d = torch.zeros(n, n)
    for k in range(n):
        for i in range(n):
            d[k, i] = np.sqrt(1.0 / n) if k == 0 else np.sqrt(2.0 / n) * np.cos(np.pi * k * (2 * i + 1) / (2 * n))
    return d


def _zigzag_masks(n: int = 8) -> T:
    order = torch.zeros(n, n, dtype=torch.long)
    idx = 0
    for s in range(2 * n - 1):
        rng = range(min(s, n - 1), max(0, s - n + 1) - 1, -1) if s % 2 == 0 else range(max(0, s - n + 1), min(s, n - 1) + 1)
        for i in rng:
            order[i, s - i] = idx
            idx += 1
    lo, hi = (n * n) // 3, 2 * (n * n) // 3
    return torch.stack([(order < lo).float(), ((order >= lo) & (order < hi)).float(), (order >= hi).float()])


class FreqBands:
    """MultiDomainFrequencyDecomposition.decompose (multi_domain_frequency.py:578-591): LR image
    [1,3,H,W] -> nine bands as one NHWC tensor [1,H,W,27] (DCT 0-8, DWT 9-20, FFT 21-26)."""

    def __init__(self, sd: SD, dev):
        m = "multi_domain_freq."
        self.dev = dev
        self.dct = _dct_matrix().to(dev).contiguous()
        self.masks = _zigzag_masks().to(dev).contiguous()
        self.dct_scale = sd[m + "dct.band_scale"].contiguous()
        self.sub_scale = [float(v) for v in sd[m + "dwt.subband_scale"].cpu()]
        self.lo = torch.tensor(DB4_LO, dtype=torch.float32, device=dev)
        self.hi = torch.tensor(DB4_HI, dtype=torch.float32, device=dev)
        self.mask_logits = sd[m + "fft.freq_mask_logits"].reshape(64, 64).contiguous()
        self.temp = max(float(sd[m + "fft.temperature"].cpu()), 1.0)
        self.fft_scale = sd[m + "fft.band_scale"].contiguous()
        self._tw: Dict[int, tuple] = {}

    def _twiddle(self, n: int):
        if n not in self._tw:
            ang = 2.0 * np.pi * np.arange(n, dtype=np.float64) / n
            self._tw[n] = (torch.from_numpy(np.cos(ang).astype(np.float32)).to(self.dev),
                           torch.from_numpy(np.sin(ang).astype(np.float32)).to(self.dev))
        return self._tw[n]

    def __call__(self, lr: T) -> T:
        _, c, h, w = lr.shape
        x = lr.reshape(c, h, w).contiguous()
        out = torch.empty((1, h, w, 27), device=lr.device, dtype=torch.float32)
        ops.dct8_bands(x, self.dct, self.masks, self.dct_scale, out, 0)
        lo_r, hi_r = ops.dwt_pass(x, 1, self.lo, self.hi)
        ll, lh = ops.dwt_pass(lo_r, 0, self.lo, self.hi)
        hl, hh = ops.dwt_pass(hi_r, 0, self.lo, self.hi)
        for i, sb in enumerate((ll, lh, hl, hh)):
            ops.resize(sb.unsqueeze(0), (h, w), layout="nchw", out=out[..., 9 + 3 * i:12 + 3 * i], mul=self.sub_scale[i])
        ops.fft_bands(x, self._twiddle(w), self._twiddle(h), self.mask_logits, self.temp, self.fft_scale, out, 21, 24)
        return out
