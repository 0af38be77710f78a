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

from . import lib as _lib
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


def _conv_b(sd: SD, name: str, dev):
    b = sd.get(name + ".bias")
    return pack_conv(sd[name + ".weight"]).to(dev), (b.to(dev).contiguous() if b is not None else None)


def _lin1x1(sd: SD, name: str, dev):
    w = sd[name + ".weight"]
    b = sd.get(name + ".bias")
    return w.reshape(w.shape[0], -1).contiguous().to(dev), (b.to(dev).contiguous() if b is not None else None)


class CrossBandLKA:
    """EnhancedCrossBandWithLKA (large_kernel_attention.py:156-244): per-pixel 9-token attention across the
    bands, then the LKA block (BN -> DW5x5 -> DW1x21 -> DW21x1 -> 1x1 -> BN -> sigmoid gate; BN -> FFN)
    on every band.  The nine bands are batched as 9*64 = 576 depth-wise channels / 9P GEMM rows."""

    def __init__(self, sd: SD, dev, p: str = "cross_band_attn", nb: int = 9, heads: int = 4):
        self.nb, self.heads, self.dev = nb, heads, dev
        self.proj = _lin1x1(sd, p + ".band_proj", dev)
        self.norm = (sd[p + ".norm.weight"].to(dev), sd[p + ".norm.bias"].to(dev))
        self.in_proj = (sd[p + ".band_attention.in_proj_weight"].to(dev).contiguous(), sd[p + ".band_attention.in_proj_bias"].to(dev))
        self.out_proj = (sd[p + ".band_attention.out_proj.weight"].to(dev).contiguous(), sd[p + ".band_attention.out_proj.bias"].to(dev))
        q = p + ".lka_block"
        sdd = {k: v.to(dev) for k, v in sd.items() if k.startswith(q)}
        s1, t1 = bn_scale_shift(sdd, q + ".norm1")
        s2, t2 = bn_scale_shift(sdd, q + ".norm2")
        self.bn1 = (s1.repeat(nb).contiguous(), t1.repeat(nb).contiguous())
        self.bn2 = (s2, t2)
        self.dw5 = pack_dw(sdd[q + ".lka.local_conv.weight"]).repeat(1, nb).contiguous()
        self.dwh = pack_dw(sdd[q + ".lka.h_conv.weight"]).repeat(1, nb).contiguous()
        self.dwv = pack_dw(sdd[q + ".lka.v_conv.weight"]).repeat(1, nb).contiguous()
        sb, tb = bn_scale_shift(sdd, q + ".lka.bn")
        self.pw = fold_bn_after_conv(sdd[q + ".lka.pw_conv.weight"].reshape(64, 64), None, sb, tb)
        self.scale1, self.scale2 = float(sdd[q + ".scale1"].cpu()), float(sdd[q + ".scale2"].cpu())
        self.ffn0 = _lin1x1(sd, q + ".ffn.0", dev)
        self.ffn2 = _lin1x1(sd, q + ".ffn.2", dev)
        # token-stationary forms (split-bf16 modes): the eval BatchNorm in front of the FFN is a per-channel affine and folds into
        # ffn.0 (W (s*x + t) + b = (W diag s) x + (W t + b)); scale2 folds into ffn.2; norm + in_proj and out_proj + residual are
        # ff_token_linear launches with the LayerNorm in the prologue -- three bandwidth passes over the 151 MB token matrix fewer
        from .prep import pack_token_linear
        w0, b0 = self.ffn0
        self.tl_ffn0 = pack_token_linear((w0 * s2[None, :]).contiguous(), (b0 + w0 @ t2).contiguous())
        w2, b2 = self.ffn2
        self.tl_ffn2 = pack_token_linear((w2 * self.scale2).contiguous(), (b2 * self.scale2).contiguous())
        self.tl_in = pack_token_linear(*self.in_proj)
        self.tl_out = pack_token_linear(*self.out_proj)
        self.outp = _lin1x1(sd, p + ".out_proj", dev)

    def __call__(self, bands: T) -> T:
        """bands [1,H,W,27] -> enhanced bands [1,H,W,27]."""
        _, H, W, _ = bands.shape
        nb, E = self.nb, 64
        P = H * W
        tok = torch.empty((1, H, W, nb * E), device=bands.device, dtype=torch.float32)
        for i in range(nb):
            ops.linear(bands[..., 3 * i:3 * i + 3], *self.proj, out=tok[..., E * i:E * (i + 1)])
        rows = tok.reshape(P * nb, E)                                   # token (pixel, band) rows
        fast = ops.fused_modes()
        if fast:
            qkv = ops.token_linear(rows, self.tl_in, gamma=self.norm[0], beta=self.norm[1])          # LayerNorm + in_proj
        else:
            qkv = ops.linear(ops.layernorm(rows, *self.norm), *self.in_proj)
        att = ops.band_mha_core(qkv, P, nb, self.heads)
        x = ops.token_linear(att, self.tl_out, res=rows) if fast else ops.linear(att, *self.out_proj, res=rows)      # [9P, 64]
        ximg = x.reshape(1, H, W, nb * E)
        # LKA block on all bands at once
        t = ops.affine(ximg, *self.bn1)
        a = ops.dwconv2d(t, self.dw5, None, ksize=(5, 5), pad=(2, 2))
        a = ops.dwconv2d(a, self.dwh, None, ksize=(1, 21), pad=(0, 10))
        a = ops.dwconv2d(a, self.dwv, None, ksize=(21, 1), pad=(10, 0))
        a = ops.linear(a.reshape(P * nb, E), *self.pw, act="sigmoid")
        x = ops.fma3(x, t.reshape(P * nb, E), a, self.scale1)           # x + s1 * (norm1(x) * attn)
        if fast:
            f = ops.token_linear(x, self.tl_ffn0, act="gelu")           # BatchNorm folded in
            x = ops.token_linear(f, self.tl_ffn2, res=x)                # scale2 folded in
        else:
            t2 = ops.affine(x, *self.bn2)
            f = ops.linear(t2, *self.ffn0, act="gelu")
            x = ops.linear(f, *self.ffn2, res=x, alpha=self.scale2)
        ximg = x.reshape(1, H, W, nb * E)
        out = torch.empty_like(bands)
        for i in range(nb):
            ops.linear(ximg[..., E * i:E * (i + 1)], *self.outp, res=bands[..., 3 * i:3 * i + 3], out=out[..., 3 * i:3 * i + 3])
        return out


class Collaborative:
    """EnhancedCollaborativeWithLKA in eval mode (large_kernel_attention.py:250-419), live only in the cached-mode forward
    (enhanced_fusion.py:756-812): expert features [1,C,h,w] -> one gain per expert and colour channel.  Same building blocks as
    CrossBandLKA with 3 tokens of 128 channels per pixel; the three experts are batched as 3*128 depth-wise channels / 3P rows."""

    def __init__(self, sd: SD, dev, p: str = "collaborative", heads: int = 8):
        self.heads, self.dev, E = heads, dev, 128
        self.align = [_lin1x1(sd, f"{p}.align_layers.{n}", dev) for n in ("hat", "dat", "nafnet")]
        self.norm1 = (sd[p + ".norm1.weight"].to(dev), sd[p + ".norm1.bias"].to(dev))
        self.norm2 = (sd[p + ".norm2.weight"].to(dev), sd[p + ".norm2.bias"].to(dev))
        self.in_proj = (sd[p + ".cross_attn.in_proj_weight"].to(dev).contiguous(), sd[p + ".cross_attn.in_proj_bias"].to(dev))
        self.out_proj = (sd[p + ".cross_attn.out_proj.weight"].to(dev).contiguous(), sd[p + ".cross_attn.out_proj.bias"].to(dev))
        self.ffn0 = (sd[p + ".ffn.0.weight"].to(dev).contiguous(), sd[p + ".ffn.0.bias"].to(dev))
        self.ffn2 = (sd[p + ".ffn.2.weight"].to(dev).contiguous(), sd[p + ".ffn.2.bias"].to(dev))
        q = p + ".lka_global"
        sdd = {k: v.to(dev) for k, v in sd.items() if k.startswith(q)}
        s1, t1 = bn_scale_shift(sdd, q + ".norm1")
        s2, t2 = bn_scale_shift(sdd, q + ".norm2")
        self.bn1 = (s1.repeat(3).contiguous(), t1.repeat(3).contiguous())
        self.dw5 = pack_dw(sdd[q + ".lka.local_conv.weight"]).repeat(1, 3).contiguous()
        self.dwh = pack_dw(sdd[q + ".lka.h_conv.weight"]).repeat(1, 3).contiguous()
        self.dwv = pack_dw(sdd[q + ".lka.v_conv.weight"]).repeat(1, 3).contiguous()
        sb, tb = bn_scale_shift(sdd, q + ".lka.bn")
        self.pw = fold_bn_after_conv(sdd[q + ".lka.pw_conv.weight"].reshape(E, E), None, sb, tb)
        self.scale1, self.scale2 = float(sdd[q + ".scale1"].cpu()), float(sdd[q + ".scale2"].cpu())
        w0, b0 = _lin1x1(sd, q + ".ffn.0", dev)
        self.lffn0 = ((w0 * s2[None, :]).contiguous(), (b0 + w0 @ t2).contiguous())      # eval BatchNorm folded into ffn.0
        self.lffn2 = _lin1x1(sd, q + ".ffn.2", dev)
        self.mod0 = [_lin1x1(sd, f"{p}.modulation.{i}.0", dev) for i in range(3)]
        self.mod3 = [_lin1x1(sd, f"{p}.modulation.{i}.3", dev) for i in range(3)]
        self.g_scale = torch.full((3,), 0.2, device=dev)
        self.g_shift = torch.full((3,), 0.9, device=dev)                                   # 1 + 0.2 (m - 0.5) = 0.9 + 0.2 m

    def __call__(self, feats: Dict[str, T], hr_hw, taps: Optional[dict] = None):
        """feats: NCHW {hat [1,180,h,w], dat [1,180,h,w], nafnet [1,64,h,w]} -> three device vectors [3] of per-channel gains."""
        E, ne = 128, 3
        shapes = {tuple(feats[k].shape[2:]) for k in ("hat", "dat", "nafnet")}
        if len(shapes) != 1:
            raise _lib.FFError(f"collaborative: the cached features must share one resolution (cache.py writes them at LR size), got {shapes}")
        h, w = next(iter(shapes))
        P = h * w
        tok = torch.empty((1, h, w, ne * E), device=self.dev, dtype=torch.float32)
        for i, k in enumerate(("hat", "dat", "nafnet")):
            ops.linear(ops.nchw_to_nhwc(feats[k].to(self.dev, torch.float32)), *self.align[i], out=tok[..., E * i:E * (i + 1)])
        rows = tok.reshape(P * ne, E)
        qkv = ops.linear(ops.layernorm(rows, *self.norm1), *self.in_proj)
        att = ops.band_mha_core(qkv, P, ne, self.heads)
        x = ops.linear(att, *self.out_proj, res=rows)
        f = ops.linear(ops.layernorm(x, *self.norm2), *self.ffn0, act="gelu")
        x = ops.linear(f, *self.ffn2, res=x)
        # LKA block, the three experts at once
        ximg = x.reshape(1, h, w, ne * E)
        t = ops.affine(ximg, *self.bn1)
        a = ops.dwconv2d(t, self.dw5, None, ksize=(5, 5), pad=(2, 2))
        a = ops.dwconv2d(a, self.dwh, None, ksize=(1, 21), pad=(0, 10))
        a = ops.dwconv2d(a, self.dwv, None, ksize=(21, 1), pad=(10, 0))
        a = ops.linear(a.reshape(P * ne, E), *self.pw, act="sigmoid")
        x = ops.fma3(x, t.reshape(P * ne, E), a, self.scale1)
        f = ops.linear(x, *self.lffn0, act="gelu")
        x = ops.linear(f, *self.lffn2, res=x, alpha=self.scale2)
        ximg = x.reshape(1, h, w, ne * E)
        gains = []
        for i in range(ne):
            up = ops.resize(ximg[..., E * i:E * (i + 1)], hr_hw)                          # [1,H,W,128] bilinear
            m = ops.pool_mean(ops.linear(up, *self.mod0[i], act="gelu"))                   # [1,32]
            m = ops.vec_mlp(m, self.mod3[i][0], self.mod3[i][1], "sigmoid")                # [1,3]
            if taps is not None:
                taps[f"collab.mod{i}"] = m
            gains.append(ops.affine(m, self.g_scale, self.g_shift).reshape(3))
        return gains


class BandFusion:
    """AdaptiveBandFusionModule (multi_domain_frequency.py:415-526): 9 -> 3 guidance bands."""

    def __init__(self, sd: SD, dev, p: str = "multi_domain_freq.band_fusion"):
        import torch.nn.functional as F
        imp = torch.cat([F.softplus(sd[p + ".dct_importance"]), F.softplus(sd[p + ".dwt_importance"]),
                         F.softplus(sd[p + ".fft_importance"])]).float().cpu()
        self.imp = (imp / (imp.sum() + 1e-8)).to(dev).contiguous()
        # nine 3->1 3x3 band-attention convs as one block-diagonal 27->9 conv
        wbd = torch.zeros(9, 27, 3, 3)
        bbd = torch.zeros(9)
        for i in range(9):
            wbd[i, 3 * i:3 * i + 3] = sd[f"{p}.band_attention.{i}.conv.0.weight"][0].cpu()
            bbd[i] = sd[f"{p}.band_attention.{i}.conv.0.bias"][0].cpu()
        self.att = (pack_conv(wbd).to(dev), bbd.to(dev))
        self.tr0, self.tr2 = _lin1x1(sd, p + ".fusion_transform.0", dev), _lin1x1(sd, p + ".fusion_transform.2", dev)
        self.gt0, self.gt2 = _lin1x1(sd, p + ".fusion_gate.0", dev), _lin1x1(sd, p + ".fusion_gate.2", dev)
        self.res = _lin1x1(sd, p + ".dct_residual", dev)

    def __call__(self, xb: T) -> T:
        """xb [1,H,W,27] -> guidance [1,H,W,9] = (low rgb, mid rgb, high rgb)."""
        att = ops.conv2d(xb, *self.att, ksize=(3, 3), pad=(1, 1), act="sigmoid")           # [1,H,W,9]
        wb = ops.band_weight(xb, att, self.imp)
        tr = ops.linear(ops.linear(wb, *self.tr0, act="gelu"), *self.tr2)
        gt = ops.linear(ops.linear(wb, *self.gt0, act="gelu"), *self.gt2, act="sigmoid")
        r = ops.linear(xb[..., :9], *self.res, alpha=0.3)
        return ops.fma3(r, tr, gt)


class HierFusion:
    """HierarchicalMultiResolutionFusion (hierarchical_fusion.py:67-197)."""

    def __init__(self, sd: SD, dev, p: str = "multi_res_fusion"):
        self.st = {}
        for name in ("stage1", "stage2", "stage3"):
            self.st[name] = dict(c0=_conv_b(sd, f"{p}.{name}_conv.0", dev), c2=_conv_b(sd, f"{p}.{name}_conv.2", dev),
                                 g0=_lin1x1(sd, f"{p}.{name}_gate.gate.0", dev), g2=_lin1x1(sd, f"{p}.{name}_gate.gate.2", dev),
                                 r0=_conv_b(sd, f"{p}.{name}_res.block.0", dev), r2=_conv_b(sd, f"{p}.{name}_res.block.2", dev),
                                 rs=float(sd[f"{p}.{name}_res.scale"].cpu()))
        self.rgb0, self.rgb2 = _conv_b(sd, p + ".to_rgb.0", dev), _conv_b(sd, p + ".to_rgb.2", dev)
        self.w12, self.w23 = float(sd[p + ".residual_weight_1_2"].cpu()), float(sd[p + ".residual_weight_2_3"].cpu())
        # stage 2 / 3 read cat(features 64, experts 9) = 73 channels (hierarchical_fusion.py:160,176).  The concat buffers are
        # given 76 channels (three that stay zero) and the weights three zero input columns per tap, so the convolution takes
        # the 16-byte-aligned / LDS-resident path instead of the scalar gather (915 -> ~500 us at 1024x1024).
        for name in ("stage2", "stage3"):
            w, b = self.st[name]["c0"]
            w76 = torch.zeros(w.shape[0], 9, 76, device=w.device)
            w76[:, :, :73] = w.reshape(w.shape[0], 9, 73)
            self.st[name]["c0"] = (w76.reshape(w.shape[0], 9 * 76).contiguous(), b)

    def _cat_buf(self, h: int, w: int, dev, key: str) -> T:
        """[1,h,w,76] concat buffer whose channels 73..75 are zero (zeroed when the buffer is made; never written again).
        Owned by the graph entry being captured, or -- for eager forwards -- one buffer per role that is replaced on a size
        change (ops.persistent_zeros): HBM stays bounded over a directory of differently sized images."""
        return ops.persistent_zeros(id(self), key, (1, h, w, 76), dev)

    def __del__(self):
        try:
            ops.drop_persistent(owner=id(self))
        except Exception:                                   # interpreter shutdown
            pass

    def _stage(self, x: T, name: str) -> T:
        k = self.st[name]
        x = ops.conv2d(x, *k["c0"], ksize=(3, 3), pad=(1, 1), act="gelu")
        x = ops.conv2d(x, *k["c2"], ksize=(3, 3), pad=(1, 1), act="gelu")
        if ops.gemm_mode() != "f32" and k["g0"][0].shape[0] <= 16:
            # SpatialGate (hierarchical_fusion.py:25-43): conv1x1 C -> C/4 -> GELU -> conv1x1 -> 1 -> sigmoid per pixel in ONE pass over x
            # (fp32 VALU, ff_pixel_mlp) instead of two N <= 16 GEMM launches that re-read the 64-channel tensor
            if "g2b" not in k:
                k["g2b"] = float(k["g2"][1].reshape(-1)[0].cpu())
            gate = ops.pixel_mlp(x, k["g0"][0], k["g0"][1], "gelu", k["g2"][0], k["g2b"], "sigmoid")
        else:
            gate = ops.linear(ops.linear(x, *k["g0"], act="gelu"), *k["g2"], act="sigmoid")    # [.,1]
        x = ops.mix2(x, pa=gate)
        r = ops.conv2d(x, *k["r0"], ksize=(3, 3), pad=(1, 1), act="gelu")
        return ops.conv2d(r, *k["r2"], ksize=(3, 3), pad=(1, 1), res=x, alpha=k["rs"])

    def __call__(self, experts9: T) -> T:
        _, fh, fw, _ = experts9.shape
        s1, s2 = (max(fh // 4, 1), max(fw // 4, 1)), (max(fh // 2, 1), max(fw // 2, 1))
        dev = experts9.device
        f1 = self._stage(ops.resize(experts9, s1), "stage1")                               # [1,h/4,w/4,64]
        in2 = self._cat_buf(s2[0], s2[1], dev, "in2")
        ops.resize(f1, s2, out=in2[..., :64])
        ops.resize(experts9, s2, out=in2[..., 64:73])
        f2 = self._stage(in2, "stage2")
        f2 = ops.mix2(f2, in2[..., :64], kb=self.w12)
        in3 = self._cat_buf(fh, fw, dev, "in3")
        ops.resize(f2, (fh, fw), out=in3[..., :64])
        ops.mix2(experts9, out=in3[..., 64:73])
        f3 = self._stage(in3, "stage3")                                                    # [1,fh,fw,32]
        f3 = ops.mix2(f3, in3[..., :32], kb=self.w23)
        o = ops.conv2d(f3, *self.rgb0, ksize=(3, 3), pad=(1, 1), act="gelu")
        return ops.conv2d(o, *self.rgb2, ksize=(3, 3), pad=(1, 1), act="sigmoid")


class DynamicSelection:
    """MultiScaleFeatureExtractor + DynamicExpertSelector (fusion_network.py:543-607, 167-236)."""

    def __init__(self, sd: SD, dev):
        self.br = {}
        for n in ("conv_1x", "conv_2x", "conv_4x"):
            sdd = {k: v.to(dev) for k, v in sd.items() if k.startswith(f"multiscale.{n}.2")}
            self.br[n] = (pack_conv(sd[f"multiscale.{n}.0.weight"]).to(dev), *bn_scale_shift(sdd, f"multiscale.{n}.2"))
        self.fuse = _lin1x1(sd, "multiscale.fusion", dev)
        d = "dynamic_selector."
        self.d0, self.d2, self.d4 = (_conv_b(sd, d + f"difficulty_estimator.{i}", dev) for i in (0, 2, 4))
        self.g0, self.g2 = _conv_b(sd, d + "expert_gate.0", dev), _lin1x1(sd, d + "expert_gate.2", dev)

    def _branch(self, x: T, n: str, out: T):
        w, sc, sh = self.br[n]
        y = ops.conv2d(x, w, None, ksize=(3, 3), pad=(1, 1), act="relu")
        return ops.affine(y, sc, sh, out=out)

    def __call__(self, lr_nhwc: T):
        _, H, W, _ = lr_nhwc.shape
        dev = lr_nhwc.device
        cat = torch.empty((1, H, W, 192), device=dev, dtype=torch.float32)
        self._branch(lr_nhwc, "conv_1x", cat[..., :64])
        for n, sf, sl in (("conv_2x", 0.5, slice(64, 128)), ("conv_4x", 0.25, slice(128, 192))):
            hs, ws = int(math.floor(H * sf)), int(math.floor(W * sf))
            xs = ops.resize(lr_nhwc, (hs, ws), scale_factor=sf)
            f = self._branch(xs, n, torch.empty((1, hs, ws, 64), device=dev, dtype=torch.float32))
            ops.resize(f, (H, W), out=cat[..., sl])
        feats = ops.linear(cat, *self.fuse)
        dif = ops.conv2d(lr_nhwc, *self.d0, ksize=(3, 3), pad=(1, 1), act="relu")
        dif = ops.conv2d(dif, *self.d2, ksize=(3, 3), pad=(1, 1), act="relu")
        dif = ops.conv2d(dif, *self.d4, ksize=(3, 3), pad=(1, 1), act="sigmoid")           # [1,H,W,1]
        g = ops.conv2d(feats, *self.g0, ksize=(3, 3), pad=(1, 1), act="relu")
        g = ops.linear(g, *self.g2, act="sigmoid")                                         # [1,H,W,3]
        return ops.dynamic_gates(g, dif), dif


class EdgeRefine:
    """LaplacianPyramidRefinement (edge_enhancement.py:126-260)."""

    def __init__(self, sd: SD, dev, p: str = "edge_refine", levels: int = 3):
        co = torch.arange(5, dtype=torch.float32) - 2
        g = torch.exp(-(co ** 2) / (2 * 1.5 ** 2))
        g = g / g.sum()
        self.gauss = (g[:, None] * g[None, :]).reshape(25, 1).repeat(1, 3).contiguous().to(dev)
        self.levels = levels
        self.lw = [float(v) for v in torch.softmax(sd[p + ".level_weights"].float().cpu(), dim=0)]
        self.strength = float(sd[p + ".edge_strength"].cpu())
        self.ref = []
        for i in range(levels):
            q = f"{p}.edge_refiners.{i}"
            self.ref.append(dict(c1=_conv_b(sd, q + ".conv1", dev), c2=_conv_b(sd, q + ".conv2", dev), c3=_conv_b(sd, q + ".conv3", dev),
                                 pj=_lin1x1(sd, q + ".proj", dev), a0=_lin1x1(sd, q + ".attn.attn.0", dev),
                                 a2=_conv_b(sd, q + ".attn.attn.2", dev)))
        self.f0, self.f2 = _conv_b(sd, p + ".fusion.0", dev), _conv_b(sd, p + ".fusion.2", dev)
        self.e0, self.e2 = _conv_b(sd, p + ".edge_gate.0", dev), _conv_b(sd, p + ".edge_gate.2", dev)

    def __call__(self, img: T) -> T:
        _, H, W, _ = img.shape
        dev = img.device
        pyr, cur = [], img
        for lv in range(self.levels):
            if lv < self.levels - 1:
                blur = ops.dwconv2d(cur, self.gauss, None, ksize=(5, 5), pad=(2, 2))
                down = ops.avgpool2(blur)
                up = ops.resize(down, (cur.shape[1], cur.shape[2]))
                pyr.append(ops.mix2(cur, up, kb=-1.0))
                cur = down
            else:
                pyr.append(cur)
        feats = torch.empty((1, H, W, 32 * self.levels), device=dev, dtype=torch.float32)
        for lv, lap in enumerate(pyr):
            k = self.ref[lv]
            o = ops.conv2d(lap, *k["c1"], ksize=(3, 3), pad=(1, 1), act="gelu")
            o = ops.conv2d(o, *k["c2"], ksize=(3, 3), pad=(1, 1), act="gelu")
            idn = ops.linear(lap, *k["pj"])
            o = ops.conv2d(o, *k["c3"], ksize=(3, 3), pad=(1, 1), res=idn)
            a = ops.linear(o, *k["a0"], act="gelu")
            a = ops.conv2d(a, *k["a2"], ksize=(3, 3), pad=(1, 1), act="sigmoid")           # [.,1]
            sl = feats[..., 32 * lv:32 * (lv + 1)]
            if (o.shape[1], o.shape[2]) != (H, W):
                o = ops.mix2(o, pa=a)
                ops.resize(o, (H, W), out=sl, mul=self.lw[lv])
            else:
                ops.mix2(o, pa=a, ka=self.lw[lv], out=sl)
        e = ops.conv2d(feats, *self.f0, ksize=(3, 3), pad=(1, 1), act="gelu")
        edge = ops.conv2d(e, *self.f2, ksize=(3, 3), pad=(1, 1))                           # [1,H,W,3]
        cat = torch.empty((1, H, W, 6), device=dev, dtype=torch.float32)
        ops.mix2(img, out=cat[..., :3])
        ops.mix2(edge, out=cat[..., 3:])
        gate = ops.conv2d(cat, *self.e0, ksize=(3, 3), pad=(1, 1), act="gelu")
        gate = ops.conv2d(gate, *self.e2, ksize=(3, 3), pad=(1, 1), act="sigmoid")         # [1,H,W,1]
        return ops.mix2(img, edge, kb=self.strength, pb=gate, clamp01=True)


class FusionHIP:
    """Everything after the experts (enhanced_fusion.py:729-742 in eval mode)."""

    def __init__(self, sd: SD, dev):
        sd = {k: v.to(dev, torch.float32) for k, v in sd.items() if not k.startswith("expert_ensemble.")}
        self.dev = dev
        self.bands = FreqBands(sd, dev)
        self.xband = CrossBandLKA(sd, dev)
        self.bfuse = BandFusion(sd, dev)
        self.hier = HierFusion(sd, dev)
        self.dyn = DynamicSelection(sd, dev)
        self.refine = [_conv_b(sd, f"refine_net.{i}", dev) for i in (0, 2, 4, 6)]
        self.res_scale = float(sd["residual_scale"].cpu())
        self.edge = EdgeRefine(sd, dev)
        # the cached-mode forward's collaborative block, when the checkpoint carries it (synthetic weights: parts += "collab")
        self.collab = Collaborative(sd, dev) if "collaborative.norm1.weight" in sd else None

    def pre(self, lr: T) -> dict:
        """Everything that depends on the LR input only (frequency bands, cross-band attention, band fusion -> guidance,
        dynamic-selection gates, the bilinear LR skip): independent of the experts, so model.py runs it beside them."""
        _, _, h, w = lr.shape
        raw = self.bands(lr)
        xb = self.xband(raw)
        b3 = self.bfuse(xb)
        guide = ops.freq_guidance(b3)
        lr_nhwc = ops.nchw_to_nhwc(lr)
        gates, dif = self.dyn(lr_nhwc)
        up = ops.resize(lr_nhwc, (4 * h, 4 * w), mul=self.res_scale)
        return dict(raw=raw, xb=xb, b3=b3, guide=guide, lr_nhwc=lr_nhwc, gates=gates, dif=dif, up=up)

    def forward(self, lr: T, experts: Dict[str, T], taps: Optional[dict] = None, pre: Optional[dict] = None,
                out: Optional[T] = None, feats: Optional[Dict[str, T]] = None) -> T:
        """lr NCHW [1,3,h,w]; experts: dict of NCHW [1,3,4h,4w] -> SR NCHW [1,3,4h,4w].  feats (cached-mode forward only,
        enhanced_fusion.py:794): the hooked expert features; every expert output is then scaled by its collaborative gain."""
        _, _, h, w = lr.shape
        dev = lr.device
        if pre is None:
            pre = self.pre(lr)
        raw, xb, b3, guide, gates, dif = pre["raw"], pre["xb"], pre["b3"], pre["guide"], pre["gates"], pre["dif"]
        e9 = torch.empty((1, 4 * h, 4 * w, 9), device=dev, dtype=torch.float32)
        for i, k in enumerate(("hat", "dat", "nafnet")):
            ops.nchw_to_nhwc(experts[k], out=e9[..., 3 * i:3 * i + 3])
        if feats is not None:
            if self.collab is None:
                raise _lib.FFError("expert features were passed but the checkpoint holds no collaborative.* weights")
            gains = self.collab(feats, (4 * h, 4 * w), taps)
            for i, k in enumerate(("hat", "dat", "nafnet")):
                ops.mix2(e9[..., 3 * i:3 * i + 3], ca=gains[i], clamp01=True, out=e9[..., 3 * i:3 * i + 3])       # :415-416
                if taps is not None:
                    taps[f"collab.out.{k}"] = e9[..., 3 * i:3 * i + 3]
        hier = self.hier(e9)
        fused = ops.fuse_blend(e9, hier, guide, gates, dif)
        if taps is not None:
            for i in range(9):
                taps[f"bands.raw{i}"] = raw[..., 3 * i:3 * i + 3]
                taps[f"bands.xb{i}"] = xb[..., 3 * i:3 * i + 3]
            for i in range(3):
                taps[f"bands.g{i}"] = b3[..., 3 * i:3 * i + 3]
            taps.update({"fusion.hier": hier, "fusion.gates": gates, "fusion.difficulty": dif, "fusion.fused1": fused})
        r = ops.conv2d(fused, *self.refine[0], ksize=(3, 3), pad=(1, 1), act="gelu")
        r = ops.conv2d(r, *self.refine[1], ksize=(3, 3), pad=(1, 1), act="gelu")
        r = ops.conv2d(r, *self.refine[2], ksize=(3, 3), pad=(1, 1), act="gelu")
        f = ops.conv2d(r, *self.refine[3], ksize=(3, 3), pad=(1, 1), res=fused, alpha=0.1)
        f = ops.mix2(f, pre["up"], clamp01=True)
        if taps is not None:
            taps["fusion.pre_edge"] = f
        return ops.nhwc_to_nchw(self.edge(f), out=out)
