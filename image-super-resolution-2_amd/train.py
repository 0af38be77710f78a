"""The fusion-only training step on the HIP kernels (SURVEY 8f rank 1 / BASELINE config 5).

Drop-in for the body of the reference's `train_epoch_cached` loop (train.py:308-356) with the experts precomputed:

    sr   = model.forward_with_precomputed(lr, expert_imgs, expert_feats)      # TRAIN mode (enhanced_fusion.py:756-812)
    loss = mean |clamp(sr, 0, 1) - hr|                                        # CombinedLoss, stage-1 weights {l1: 1}
    loss.backward(); clip_grad_norm_(params, 1.0); AdamW.step(); ema.update() # train.py:338-351, checkpoint_manager.py:400-407

    tr = FusionTrainer(state_dict, "cuda:0")          # reference-keyed fusion (+ collaborative) state dict
    loss = tr.step(lr, hr, outs, feats)               # one optimizer step; returns the loss (device scalar)

Train-mode semantics that differ from the eval path (fusion.py): BatchNorm uses batch statistics -- per CALL, and the shared LKA
block is called once per band / expert (large_kernel_attention.py:236-240, 406-411) -- and updates its running statistics;
nn.MultiheadAttention drops attention weights with p = 0.1 (counter-based masks here: the reference's RNG stream cannot be
reproduced, so dropout-on is statistically, not bit-wise, the same: "parity unpinned (RNG stream)"); the collaborative block is live.
All arithmetic runs in this library's kernels through the tape in autograd.py; parameters, gradients, Adam moments and the EMA
shadow are five flat fp32 buffers (1.02 M values) in the kernels' own weight layouts, so the optimizer is ONE launch and the
multi-GPU gradient exchange ONE 4 MB all-reduce.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import lib as _lib
from . import ops
from . import autograd as ag
from .autograd import Var
from .fusion import DB4_LO, DB4_HI, _dct_matrix, _zigzag_masks

T = torch.Tensor
HP = dict(lr=1.5e-4, betas=(0.9, 0.999), eps=1.0e-8, weight_decay=1.0e-4, clip=1.0, ema_decay=0.9995)   # configs/train_config.yaml:102-125
DW_KEYS = ("lka.local_conv.weight", "lka.h_conv.weight", "lka.v_conv.weight")
BN_SUFFIX = ("running_mean", "running_var", "num_batches_tracked")


def _is_buffer(name: str) -> bool:
    return name.endswith(BN_SUFFIX)


def to_native(name: str, t: T) -> T:
    """Reference layout -> the kernels' layout (a permutation): conv OIHW -> [O, KH*KW*I]; depth-wise [C,1,KH,KW] -> [KH*KW, C]."""
    if t.dim() == 4 and name.endswith(DW_KEYS):
        return t.reshape(t.shape[0], -1).t().contiguous()
    if t.dim() == 4:
        return t.permute(0, 2, 3, 1).reshape(t.shape[0], -1).contiguous()
    return t.contiguous()


def from_native(name: str, t: T, ref_shape) -> T:
    ref_shape = tuple(ref_shape)
    if len(ref_shape) == 4 and name.endswith(DW_KEYS):
        return t.reshape(ref_shape[2] * ref_shape[3], ref_shape[0]).t().reshape(ref_shape).contiguous()
    if len(ref_shape) == 4:
        o, i, kh, kw = ref_shape
        return t.reshape(o, kh, kw, i).permute(0, 3, 1, 2).contiguous()
    return t.reshape(ref_shape).contiguous()


def trainable_names(sd: Dict[str, T]) -> List[str]:
    """Parameters that receive a gradient on this path, in state-dict order (everything but BatchNorm buffers; parameters that are
    dead under the shipped MODEL_CONFIG -- freq_router.*, expert_weights, band_importance -- are not in the synthetic dict and, when a
    real checkpoint carries them, are kept aside unchanged: the reference's AdamW skips tensors whose .grad is None)."""
    dead = ("freq_router.", "expert_weights", "band_importance", "expert_ensemble.")
    fixed = ("multi_domain_freq.dct.dct_basis", "multi_domain_freq.dct.low_mask", "multi_domain_freq.dct.mid_mask", "multi_domain_freq.dct.high_mask",
             "multi_domain_freq.dwt.lo_", "multi_domain_freq.dwt.hi_", "edge_refine.gaussian.kernel")
    return [k for k in sd if not _is_buffer(k) and not k.startswith(dead) and not k.startswith(fixed)]


class FusionTrainer:
    def __init__(self, state_dict: Dict[str, T], device="cuda:0", hp: Optional[dict] = None, dropout: float = 0.1, seed: int = 0,
                 gemm: Optional[str] = "f32"):
        """gemm: contraction mode of the step's convolutions / linear layers -- "f32" (exact fp32 MFMA, the default: the reference
        trains in fp32, configs/train_config.yaml:100-101, and the step is not GEMM-bound: 90 vs 85 ms), "bf16x3", or None = whatever
        ops.gemm_mode() says (the weight gradients are exact fp32 MFMA in every mode)."""
        dev = torch.device(device)
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise _lib.FFError("FusionTrainer needs an MI355X (torch device 'cuda'); there is no CPU fallback")
        _lib.load()
        if "collaborative.norm1.weight" not in state_dict:
            raise _lib.FFError("FusionTrainer: the state dict holds no collaborative.* weights (cached-mode training needs them)")
        self.dev, self.hp, self.dropout, self.seed = dev, dict(HP, **(hp or {})), float(dropout), int(seed)
        self.gemm = gemm
        self.step_count = 0
        self.key_order = list(state_dict.keys())
        self.names = trainable_names(state_dict)
        self.ref_shape = {k: tuple(state_dict[k].shape) for k in self.names}
        self.other = OrderedDict((k, v.clone()) for k, v in state_dict.items()                                    # dead / fixed tensors
                                 if k not in self.ref_shape and not _is_buffer(k) and not k.startswith("expert_ensemble."))
        offs, n = {}, 0
        for k in self.names:
            offs[k] = n
            n += (state_dict[k].numel() + 3) // 4 * 4                # 16-byte aligned segments
        self.offs, self.n = offs, n
        with torch.cuda.device(dev):
            self.P = torch.zeros(n, device=dev)
            for k in self.names:
                self.P[offs[k]:offs[k] + state_dict[k].numel()].copy_(to_native(k, state_dict[k].float()).reshape(-1))
            self.G = torch.zeros(n, device=dev)
            self.M = torch.zeros(n, device=dev)
            self.V = torch.zeros(n, device=dev)
            self.EMA = self.P.clone()                                 # EMAModel.__init__: shadow = param.data.clone()
            self.buffers = OrderedDict((k, v.to(dev, torch.float32).clone()) for k, v in state_dict.items() if k.endswith(("running_mean", "running_var")))
            self.nbt = {k[:-len("running_mean")] + "num_batches_tracked": int(state_dict.get(k[:-len("running_mean")] + "num_batches_tracked", 0))
                        for k in state_dict if k.endswith("running_mean")}
            self.hyper = torch.zeros(10, device=dev)
            self.sqnorm = torch.zeros(1, device=dev)
            self.loss = torch.zeros(1, device=dev)
            self.work = torch.empty(1024, device=dev)
            self.one = torch.ones(1, device=dev)
            self.dct, self.masks = _dct_matrix().to(dev).contiguous(), _zigzag_masks().to(dev).contiguous()
            self.ones3 = torch.ones(3, device=dev)
            self.lo, self.hi = torch.tensor(DB4_LO, device=dev), torch.tensor(DB4_HI, device=dev)
            co = torch.arange(5, dtype=torch.float32) - 2
            g = torch.exp(-(co ** 2) / (2 * 1.5 ** 2))
            g = g / g.sum()
            self.gauss = (g[:, None] * g[None, :]).reshape(25, 1).repeat(1, 3).contiguous().to(dev)
            self.fft = ag.FFT2(dev)
        self.last_grad_norm = None

    # ---------------------------------------------------------------------------------------------------- parameters
    def _native_shape(self, k: str):
        s = self.ref_shape[k]
        if len(s) == 4 and k.endswith(DW_KEYS):
            return (s[2] * s[3], s[0])
        if len(s) == 4:
            return (s[0], s[2] * s[3] * s[1])
        return s if len(s) else (1,)

    def p(self, k: str) -> Var:
        """The parameter as a tape variable: value and gradient are views into the flat buffers."""
        v = self._vars.get(k)
        if v is None:
            o, ns = self.offs[k], self._native_shape(k)
            cnt = int(np.prod(ns))
            v = self._vars[k] = Var(self.P[o:o + cnt].reshape(ns), True, self.G[o:o + cnt].reshape(ns), name=k)
        return v

    def conv(self, x: Var, name: str, k: int = 1, act: Optional[str] = None, bias: bool = True) -> Var:
        return ag.conv2d(x, self.p(name + ".weight"), self.p(name + ".bias") if bias else None, ksize=(k, k), act=act)

    def bn(self, x: Var, name: str, groups: int = 1) -> Var:
        """BatchNorm2d in training mode on rows [.., C]; `groups` consecutive calls of the same module."""
        shp = x.data.shape
        x2 = ag.reshape(x, (-1, shp[-1]))
        y = ag.batchnorm_train(x2, self.p(name + ".weight"), self.p(name + ".bias"), self.buffers[name + ".running_mean"],
                               self.buffers[name + ".running_var"], groups=groups)
        self.nbt[name + ".num_batches_tracked"] += groups
        return ag.reshape(y, tuple(shp))

    # ---------------------------------------------------------------------------------------------------- forward pieces
    def _bands(self, lr: T) -> List[Var]:
        """MultiDomainFrequencyDecomposition.decompose (multi_domain_frequency.py:578-591): nine [B,h,w,3] bands.  The transforms of
        the input are constants; the learnable parts are the band scales and the FFT mask (logits, temperature)."""
        B, _, h, w = lr.shape
        dev = lr.device
        m = "multi_domain_freq."
        raw = torch.empty((B, h, w, 9), device=dev, dtype=torch.float32)
        for b in range(B):                                            # DCT with unit scales (the kernel is per image)
            ops.dct8_bands(lr[b], self.dct, self.masks, self.ones3, raw[b:b + 1], 0)
        bands = []
        dct_s = self.p(m + "dct.band_scale")
        rawv = ag.const(raw)
        for i in range(3):
            bands.append(ag.mul_scalar(ag.slice_ch(rawv, 3 * i, 3 * i + 3), ag.slice_ch(dct_s, i, i + 1)))
        planes = lr.reshape(B * 3, h, w)
        lo_r, hi_r = ops.dwt_pass(planes, 1, self.lo, self.hi)
        subs = (*ops.dwt_pass(lo_r, 0, self.lo, self.hi), *ops.dwt_pass(hi_r, 0, self.lo, self.hi))
        dwt_s = self.p(m + "dwt.subband_scale")
        for i, sb in enumerate(subs):
            up = ops.resize(sb.reshape(B, 3, sb.shape[1], sb.shape[2]), (h, w), layout="nchw")
            bands.append(ag.mul_scalar(ag.const(up), ag.slice_ch(dwt_s, i, i + 1)))
        # FFT: mask = sigmoid(bilinear(logits 64x64 -> h x (w/2+1)) * clamp(temperature, min=1)); low = irfft2(X mask), high = x - low
        Wf = w // 2 + 1
        X = self.fft.rfft2(planes)
        z = ag.resize(ag.reshape(self.p(m + "fft.freq_mask_logits"), (1, 64, 64, 1)), (h, Wf))
        temp = ag.unary("clamp_min", self.p(m + "fft.temperature"), 1.0)
        mask = ag.unary("sigmoid", ag.mul_scalar(z, temp))
        low = ag.planes_to_nhwc(ag.fft_lowpass(self.fft, planes, X, ag.reshape(mask, (h, Wf))), B)
        x_nhwc = ag.const(ops.nchw_to_nhwc(lr))
        high = ag.add(x_nhwc, low, -1.0)
        fs = self.p(m + "fft.band_scale")
        bands.append(ag.mul_scalar(low, ag.slice_ch(fs, 0, 1)))
        bands.append(ag.mul_scalar(high, ag.slice_ch(fs, 1, 2)))
        return bands

    def _lka_block(self, x: Var, name: str, groups: int) -> Var:
        """LKABlock.forward (large_kernel_attention.py:143-149) on [G*B, h, w, C], one module call per group of B images."""
        t = self.bn(x, name + ".norm1", groups)
        a = ag.dwconv2d(t, self.p(name + ".lka.local_conv.weight"), (5, 5))
        a = ag.dwconv2d(a, self.p(name + ".lka.h_conv.weight"), (1, 21))
        a = ag.dwconv2d(a, self.p(name + ".lka.v_conv.weight"), (21, 1))
        a = self.conv(a, name + ".lka.pw_conv", bias=False)
        a = ag.unary("sigmoid", self.bn(a, name + ".lka.bn", groups))
        x = ag.add(x, ag.mul_scalar(ag.mul(t, a), self.p(name + ".scale1")))
        t2 = self.bn(x, name + ".norm2", groups)
        f = self.conv(self.conv(t2, name + ".ffn.0", act="gelu"), name + ".ffn.2")
        return ag.add(x, ag.mul_scalar(f, self.p(name + ".scale2")))

    def _mha_block(self, tok: Var, P: int, ntok: int, heads: int, norm: str, attn: str, layer: int) -> Var:
        """norm -> nn.MultiheadAttention(self-attention over ntok tokens per pixel) ; returns the attention output (no residual)."""
        tn = ag.layernorm(tok, self.p(norm + ".weight"), self.p(norm + ".bias"))
        qkv = ag.conv2d(tn, self.p(attn + ".in_proj_weight"), self.p(attn + ".in_proj_bias"))
        seed = (self.seed * 1000003 + self.step_count) * 16 + layer
        a = ag.band_mha(qkv, P, ntok, heads, self.dropout, seed)
        return ag.conv2d(a, self.p(attn + ".out_proj.weight"), self.p(attn + ".out_proj.bias"))

    def _cross_band(self, bands: List[Var], B: int, h: int, w: int) -> Var:
        """EnhancedCrossBandWithLKA.forward (large_kernel_attention.py:207-244) -> [B,h,w,27]."""
        p, P = "cross_band_attn", B * h * w
        b27 = ag.cat_ch(bands)
        rows = ag.reshape(b27, (P * 9, 3))
        tok = self.conv(rows, p + ".band_proj")
        a = ag.add(self._mha_block(tok, P, 9, 4, p + ".norm", p + ".band_attention", 0), tok)
        img = ag.reshape(ag.permute_rows(a, P, 9, 64), (9 * B, h, w, 64))                  # band-major: one LKA call per band
        img = self._lka_block(img, p + ".lka_block", 9)
        o = self.conv(img, p + ".out_proj")
        o = ag.reshape(ag.permute_rows(ag.reshape(o, (9, P, 3)), 9, P, 3), (B, h, w, 27))
        return ag.add(o, b27)

    def _band_fusion(self, xb: Var) -> Var:
        """AdaptiveBandFusionModule.forward (multi_domain_frequency.py:478-526) -> [B,h,w,9] (low, mid, high guidance)."""
        p = "multi_domain_freq.band_fusion"
        imp = ag.cat_ch([ag.reshape(self.p(p + ".dct_importance"), (1, 3)), ag.reshape(self.p(p + ".dwt_importance"), (1, 4)),
                         ag.reshape(self.p(p + ".fft_importance"), (1, 2))])
        sp = ag.unary("softplus", imp)
        impn = ag.mul_row(sp, ag.unary("recip_eps", ag.sum_ch(sp), 1e-8))
        wb = []
        for i in range(9):
            bd = ag.slice_ch(xb, 3 * i, 3 * i + 3)
            att = self.conv(bd, f"{p}.band_attention.{i}.conv.0", 3, act="sigmoid")
            wb.append(ag.mul_scalar(ag.mul_row(bd, att), ag.slice_ch(impn, i, i + 1)))
        cat = ag.cat_ch(wb)
        tr = self.conv(self.conv(cat, p + ".fusion_transform.0", act="gelu"), p + ".fusion_transform.2")
        gt = self.conv(self.conv(cat, p + ".fusion_gate.0", act="gelu"), p + ".fusion_gate.2", act="sigmoid")
        res = self.conv(ag.slice_ch(xb, 0, 9), p + ".dct_residual")
        return ag.add(ag.mul(tr, gt), res, 0.3)

    def _collaborative(self, feats: Dict[str, T], outs: Dict[str, T], B: int, h: int, w: int) -> List[Var]:
        """EnhancedCollaborativeWithLKA.forward (large_kernel_attention.py:332-419) -> three modulated expert outputs [B,4h,4w,3]."""
        p, P, E = "collaborative", B * h * w, 128
        al = []
        for k in ("hat", "dat", "nafnet"):
            f = feats[k]
            if tuple(f.shape[2:]) != (h, w):
                raise _lib.FFError("collaborative: the cached features must be at the LR resolution (cache.py writes them so)")
            al.append(self.conv(ag.const(ops.nchw_to_nhwc(f)), f"{p}.align_layers.{k}"))
        tok = ag.reshape(ag.cat_ch(al), (P * 3, E))
        tok = ag.add(tok, self._mha_block(tok, P, 3, 8, p + ".norm1", p + ".cross_attn", 1))
        f = ag.conv2d(ag.layernorm(tok, self.p(p + ".norm2.weight"), self.p(p + ".norm2.bias")), self.p(p + ".ffn.0.weight"), self.p(p + ".ffn.0.bias"), act="gelu")
        tok = ag.add(tok, ag.conv2d(f, self.p(p + ".ffn.2.weight"), self.p(p + ".ffn.2.bias")))
        img = ag.reshape(ag.permute_rows(tok, P, 3, E), (3 * B, h, w, E))
        img = self._lka_block(img, p + ".lka_global", 3)
        res = []
        for i, k in enumerate(("hat", "dat", "nafnet")):
            up = ag.resize(ag.slice_rows(img, i * B, (i + 1) * B), (4 * h, 4 * w))
            m = ag.mean_pool(self.conv(up, f"{p}.modulation.{i}.0", act="gelu"))                       # [B, 32]
            m = self.conv(m, f"{p}.modulation.{i}.3", act="sigmoid")                                  # [B, 3]
            gain = ag.add_const(ag.scale(m, 0.2), 0.9, self.one)                                       # 1 + 0.2 (m - 0.5)
            o = ag.const(ops.nchw_to_nhwc(outs[k]))
            res.append(ag.unary("clamp01", ag.mul_gc(o, gain, 16 * h * w)))
        return res

    def _hier(self, e9: Var) -> Var:
        """HierarchicalMultiResolutionFusion.forward (hierarchical_fusion.py:131-197)."""
        p = "multi_res_fusion"
        _, fh, fw, _ = e9.data.shape
        s1, s2 = (max(fh // 4, 1), max(fw // 4, 1)), (max(fh // 2, 1), max(fw // 2, 1))

        def stage(x, name):
            x = self.conv(self.conv(x, f"{p}.{name}_conv.0", 3, act="gelu"), f"{p}.{name}_conv.2", 3, act="gelu")
            gate = self.conv(self.conv(x, f"{p}.{name}_gate.gate.0", act="gelu"), f"{p}.{name}_gate.gate.2", act="sigmoid")
            x = ag.mul_row(x, gate)
            r = self.conv(self.conv(x, f"{p}.{name}_res.block.0", 3, act="gelu", bias=False), f"{p}.{name}_res.block.2", 3, bias=False)
            return ag.add(x, ag.mul_scalar(r, self.p(f"{p}.{name}_res.scale")))

        f1 = stage(ag.resize(e9, s1), "stage1")
        f1u = ag.resize(f1, s2)
        f2 = ag.add(stage(ag.cat_ch([f1u, ag.resize(e9, s2)]), "stage2"), ag.mul_scalar(f1u, self.p(p + ".residual_weight_1_2")))
        f2u = ag.resize(f2, (fh, fw))
        f3 = stage(ag.cat_ch([f2u, e9]), "stage3")
        f3 = ag.add(f3, ag.mul_scalar(ag.slice_ch(f2u, 0, f3.data.shape[-1]), self.p(p + ".residual_weight_2_3")))
        return self.conv(self.conv(f3, p + ".to_rgb.0", 3, act="gelu"), p + ".to_rgb.2", 3, act="sigmoid")

    def _dynamic(self, lr_nhwc: T) -> Tuple[Var, Var]:
        """MultiScaleFeatureExtractor + DynamicExpertSelector (fusion_network.py:578-607, 199-236) -> gates [B,h,w,3], difficulty [B,h,w,1]."""
        _, H, W_, _ = lr_nhwc.shape
        x = ag.const(lr_nhwc)

        def branch(t, n):
            return self.bn(self.conv(t, f"multiscale.{n}.0", 3, act="relu", bias=False), f"multiscale.{n}.2")

        fs = [branch(x, "conv_1x")]
        for n, sf in (("conv_2x", 0.5), ("conv_4x", 0.25)):
            hs, ws = int(math.floor(H * sf)), int(math.floor(W_ * sf))
            fs.append(ag.resize(branch(ag.const(ops.resize(lr_nhwc, (hs, ws), scale_factor=sf)), n), (H, W_)))
        feats = self.conv(ag.cat_ch(fs), "multiscale.fusion", bias=False)
        d = "dynamic_selector."
        dif = self.conv(self.conv(self.conv(x, d + "difficulty_estimator.0", 3, act="relu"), d + "difficulty_estimator.2", 3, act="relu"),
                        d + "difficulty_estimator.4", 3, act="sigmoid")
        g = self.conv(self.conv(feats, d + "expert_gate.0", 3, act="relu"), d + "expert_gate.2", act="sigmoid")
        return ag.dynamic_gates(g, dif), dif

    def _edge(self, img: Var) -> Var:
        """LaplacianPyramidRefinement.forward (edge_enhancement.py:222-260)."""
        p, levels = "edge_refine", 3
        _, H, W_, _ = img.data.shape
        gauss = ag.const(self.gauss)
        pyr, cur = [], img
        for lv in range(levels):
            if lv < levels - 1:
                down = ag.avgpool2(ag.dwconv2d(cur, gauss, (5, 5)))
                pyr.append(ag.add(cur, ag.resize(down, tuple(cur.data.shape[1:3])), -1.0))
                cur = down
            else:
                pyr.append(cur)
        e = ag.unary("exp", ag.reshape(self.p(p + ".level_weights"), (1, 3)))
        lw = ag.mul_row(e, ag.unary("recip_eps", ag.sum_ch(e), 0.0))                                   # softmax over the three levels
        feats = []
        for lv, lap in enumerate(pyr):
            q = f"{p}.edge_refiners.{lv}"
            o = self.conv(self.conv(lap, q + ".conv1", 3, act="gelu"), q + ".conv2", 3, act="gelu")
            o = ag.add(self.conv(o, q + ".conv3", 3), self.conv(lap, q + ".proj"))
            a = self.conv(self.conv(o, q + ".attn.attn.0", act="gelu"), q + ".attn.attn.2", 3, act="sigmoid")
            o = ag.mul_row(o, a)
            if tuple(o.data.shape[1:3]) != (H, W_):
                o = ag.resize(o, (H, W_))
            feats.append(ag.mul_scalar(o, ag.slice_ch(lw, lv, lv + 1)))
        edge = self.conv(self.conv(ag.cat_ch(feats), p + ".fusion.0", 3, act="gelu"), p + ".fusion.2", 3)
        gate = self.conv(self.conv(ag.cat_ch([img, edge]), p + ".edge_gate.0", 3, act="gelu"), p + ".edge_gate.2", 3, act="sigmoid")
        return ag.add(img, ag.mul_scalar(ag.mul_row(edge, gate), self.p(p + ".edge_strength")), clamp01=True)

    def forward(self, lr: T, outs: Dict[str, T], feats: Dict[str, T]) -> Var:
        """forward_with_precomputed in training mode -> SR as an NHWC tape variable [B,4h,4w,3] (call inside `with Tape()`)."""
        B, _, h, w = lr.shape
        self._vars = {}
        bands = self._bands(lr)
        xb = self._cross_band(bands, B, h, w)
        b3 = self._band_fusion(xb)
        enh = self._collaborative(feats, outs, B, h, w)
        # fuse_experts (enhanced_fusion.py:502-591, hierarchical branch)
        hr = (4 * h, 4 * w)
        mags = [ag.sum_ch(ag.unary("abs", ag.slice_ch(b3, 3 * i, 3 * i + 3)), 1.0 / 3.0) for i in range(3)]     # low, mid, high
        rt = ag.unary("recip_eps", ag.add(ag.add(mags[0], mags[1]), mags[2]), 1e-8)
        hier = self._hier(ag.cat_ch(enh))
        weighted = None
        for e, mi in zip(enh, (2, 1, 0)):                                                              # high -> hat, mid -> dat, low -> nafnet
            t = ag.mul_row(e, ag.resize(ag.mul(mags[mi], rt), hr))
            weighted = t if weighted is None else ag.add(weighted, t)
        fused = ag.add(ag.scale(hier, 0.7), weighted, 0.3)
        lr_nhwc = ops.nchw_to_nhwc(lr)
        gates, dif = self._dynamic(lr_nhwc)
        gates_hr, dif_hr = ag.resize(gates, hr), ag.resize(dif, hr)
        dyn = None
        for i, e in enumerate(enh):
            t = ag.mul_row(e, ag.slice_ch(gates_hr, i, i + 1))
            dyn = t if dyn is None else ag.add(dyn, t)
        dyn = ag.mul_row(dyn, ag.unary("recip_eps", ag.sum_ch(gates_hr), 1e-8))
        fused = ag.add(fused, ag.mul_row(ag.add(dyn, fused, -1.0), ag.scale(dif_hr, 0.3)))           # fused (1 - 0.3 d) + dyn 0.3 d
        # refine_output (enhanced_fusion.py:653-688)
        r = self.conv(self.conv(self.conv(fused, "refine_net.0", 3, act="gelu"), "refine_net.2", 3, act="gelu"), "refine_net.4", 3, act="gelu")
        fused = ag.add(fused, self.conv(r, "refine_net.6", 3), 0.1)
        up = ag.mul_scalar(ag.const(ops.resize(lr_nhwc, hr)), self.p("residual_scale"))
        fused = ag.add(fused, up, clamp01=True)
        return self._edge(fused)

    # ---------------------------------------------------------------------------------------------------- step
    def _check_batch(self, lr, hr, outs, feats):
        if lr.dim() != 4 or lr.shape[1] != 3:
            raise _lib.FFError(f"expected lr of shape [B,3,h,w], got {tuple(lr.shape)}")
        B, _, h, w = lr.shape
        if h % 8 or w % 8 or h < 16 or w < 16:
            raise _lib.FFError("training patches must be multiples of 8 and at least 16 pixels wide (config 5: 64x64)")
        want = (B, 3, 4 * h, 4 * w)
        for k in ("hat", "dat", "nafnet"):
            if tuple(outs[k].shape) != want:
                raise _lib.FFError(f"expert output {k}: expected {want}, got {tuple(outs[k].shape)}")
        if hr is not None and tuple(hr.shape) != want:
            raise _lib.FFError(f"hr: expected {want}, got {tuple(hr.shape)}")

    def _to_dev(self, t):
        return t.to(self.dev, torch.float32).contiguous()

    def forward_backward(self, lr: T, hr: T, outs: Dict[str, T], feats: Dict[str, T]):
        """Zero the gradient buffer, run forward + L1 + backward.  Returns (sr NHWC tensor, loss [1] device tensor)."""
        lr, hr = self._to_dev(lr), self._to_dev(hr)
        outs = {k: self._to_dev(v) for k, v in outs.items()}
        feats = {k: self._to_dev(v) for k, v in feats.items()}
        self._check_batch(lr, hr, outs, feats)
        old_mode = ops.gemm_mode()
        if self.gemm is not None:
            ops.set_gemm_mode(self.gemm)
        try:
            return self._forward_backward(lr, hr, outs, feats)
        finally:
            ops.set_gemm_mode(old_mode)

    def _forward_backward(self, lr, hr, outs, feats):
        with torch.cuda.device(self.dev):
            self.G.zero_()
            with ag.Tape() as tape:
                sr = self.forward(lr, outs, feats)
                hr_nhwc = ops.nchw_to_nhwc(hr)
                dsr = torch.empty_like(sr.data)
                _lib.check(_lib.load().ff_l1_loss_grad(sr.data.data_ptr(), hr_nhwc.data_ptr(), dsr.data_ptr(), sr.data.numel(), self.loss.data_ptr(),
                                                       self.work.data_ptr(), 1024, ag._st()))
                sr.grad, sr.owned = dsr, True
                tape.backward()
        self._vars = {}
        return sr.data, self.loss

    def optimizer_step(self, lr_now: Optional[float] = None):
        """clip_grad_norm_(max_norm) + AdamW + EMA over the flat buffers: two launches."""
        hp = self.hp
        self.step_count += 1
        t = self.step_count
        lr_now = hp["lr"] if lr_now is None else lr_now
        b1, b2 = hp["betas"]
        bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        vals = [lr_now, b1, b2, hp["eps"], hp["weight_decay"], hp["clip"] or 0.0, hp["ema_decay"], float(t), lr_now / bc1, math.sqrt(bc2)]
        self.hyper.copy_(torch.tensor(vals, dtype=torch.float32), non_blocking=True)
        L = _lib.load()
        with torch.cuda.device(self.dev):
            _lib.check(L.ff_grad_sqnorm(self.G.data_ptr(), self.n, self.sqnorm.data_ptr(), self.work.data_ptr(), 1024, ag._st()))
            _lib.check(L.ff_adamw_ema_step(self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(), self.EMA.data_ptr(), self.n,
                                           self.hyper.data_ptr(), self.sqnorm.data_ptr(), ag._st()))

    def step(self, lr: T, hr: T, outs: Dict[str, T], feats: Dict[str, T], lr_now: Optional[float] = None, grad_hook=None) -> T:
        """One training step.  grad_hook(flat_grad) runs between backward and the optimizer (the multi-GPU all-reduce)."""
        _, loss = self.forward_backward(lr, hr, outs, feats)
        if grad_hook is not None:
            grad_hook(self.G)
        self.optimizer_step(lr_now)
        return loss

    # ---------------------------------------------------------------------------------------------------- state
    def _export(self, flat: T) -> "OrderedDict[str, T]":
        out = OrderedDict()
        host = flat.detach().cpu()
        for k in self.names:
            cnt = int(np.prod(self.ref_shape[k])) if len(self.ref_shape[k]) else 1
            out[k] = from_native(k, host[self.offs[k]:self.offs[k] + cnt], self.ref_shape[k])
        return out

    def grads(self) -> "OrderedDict[str, T]":
        """Gradients in the reference's parameter layout (host tensors)."""
        return self._export(self.G)

    def state_dict(self) -> "OrderedDict[str, T]":
        """Reference-keyed model state: parameters, BatchNorm buffers (with num_batches_tracked), untouched tensors."""
        sd = self._export(self.P)
        for k, v in self.buffers.items():
            sd[k] = v.detach().cpu().clone()
        for k, n in self.nbt.items():
            sd[k] = torch.tensor(n, dtype=torch.long)
        for k, v in self.other.items():
            sd[k] = v.clone()
        return sd

    def ema_shadow(self) -> "OrderedDict[str, T]":
        sd = self._export(self.EMA)
        for k, v in self.other.items():
            if not k.startswith(("multi_domain_freq.dct.", "multi_domain_freq.dwt.", "edge_refine.gaussian")):
                sd[k] = v.clone()                                     # requires_grad parameters without gradient: the shadow never moves
        return sd

    def grad_norm(self) -> float:
        return float(torch.sqrt(self.sqnorm).cpu())


# ======================================================================================================================
# Checkpoint contract of the training side (reference src/utils/checkpoint_manager.py:81-165 save, :185-239 load, train.py:961-966)
def is_parameter_key(k: str) -> bool:
    """Parameters (as opposed to registered buffers) among the keys of the reference model's state dict."""
    if k.endswith(BN_SUFFIX):
        return False
    return not k.startswith(("multi_domain_freq.dct.dct_basis", "multi_domain_freq.dct.low_mask", "multi_domain_freq.dct.mid_mask",
                             "multi_domain_freq.dct.high_mask", "multi_domain_freq.dwt.lo_", "multi_domain_freq.dwt.hi_", "edge_refine.gaussian.kernel"))


def build_checkpoint(model_sd: Dict[str, T], exp_avg: Dict[str, T], exp_avg_sq: Dict[str, T], ema_shadow: Dict[str, T], step: int, hp: dict,
                     epoch: int, metrics: Optional[dict] = None, lr_now: Optional[float] = None, scheduler_state: Optional[dict] = None,
                     stage: Optional[int] = None) -> dict:
    """The dict CheckpointManager.save_checkpoint writes, from host tensors in the reference's layouts:
        {epoch, model_state_dict, optimizer_state_dict, [scheduler_state_dict], ema_state_dict{shadow, decay}, metrics, timestamp}
    `model_sd` must be a COMPLETE state dict of the (cached-mode) reference model in its own key order -- the optimizer state is
    indexed by the position of each parameter in model.parameters(), which is the order of the parameter keys of that dict.
    Parameters without gradient (exp_avg has no entry: freq_router.*, expert_weights, band_importance) get no optimizer state, exactly as
    torch.optim.AdamW leaves them."""
    from datetime import datetime
    pkeys = [k for k in model_sd if is_parameter_key(k) and not k.startswith("expert_ensemble.")]
    state = {}
    for i, k in enumerate(pkeys):
        if k in exp_avg:
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": exp_avg[k].clone(), "exp_avg_sq": exp_avg_sq[k].clone()}
    # the param_group of THIS torch version's AdamW (so that optimizer.load_state_dict finds every key it expects)
    probe = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=hp["lr"], betas=tuple(hp["betas"]), eps=hp["eps"], weight_decay=hp["weight_decay"])
    group = dict(probe.state_dict()["param_groups"][0])
    group["params"] = list(range(len(pkeys)))
    group["lr"] = float(hp["lr"] if lr_now is None else lr_now)
    group.setdefault("initial_lr", float(hp["lr"]))                  # what CosineAnnealingWarmRestarts adds to the groups (train.py:897)
    ck = {"epoch": int(epoch), "model_state_dict": OrderedDict((k, v.clone()) for k, v in model_sd.items()),
          "optimizer_state_dict": {"state": state, "param_groups": [group]}, "metrics": dict(metrics or {}),
          "timestamp": datetime.now().isoformat(),
          "ema_state_dict": {"shadow": {k: v.clone() for k, v in ema_shadow.items()}, "decay": float(hp["ema_decay"])}}
    if scheduler_state is not None:
        ck["scheduler_state_dict"] = dict(scheduler_state)
    if stage is not None:
        ck["stage"] = stage
    return ck


def _save_atomic(ck: dict, path: str):
    tmp = os.path.splitext(path)[0] + ".tmp"
    torch.save(ck, tmp)
    os.replace(tmp, path)                                             # checkpoint_manager.py:136-139: temp file, then rename


def _trainer_save(self, path: str, epoch: int, metrics: Optional[dict] = None, lr_now: Optional[float] = None,
                  scheduler_state: Optional[dict] = None) -> str:
    """Write a checkpoint in the reference's layout (loadable by its CheckpointManager.load_checkpoint when this trainer was built
    from a complete reference state dict; a later FusionTrainer.load_checkpoint resumes bit-exactly either way)."""
    ck = build_checkpoint(self.state_dict_in_order(), self._export(self.M), self._export(self.V), self.ema_shadow(), self.step_count, self.hp,
                          epoch, metrics, lr_now, scheduler_state)
    ck["ff_trainer"] = {"seed": self.seed, "dropout": self.dropout, "step_count": self.step_count}
    _save_atomic(ck, path)
    return path


def _state_dict_in_order(self) -> "OrderedDict[str, T]":
    """state_dict() in the key order of the state dict the trainer was built from (the reference model's own order when that was a
    reference state dict): parameter positions index the optimizer state."""
    sd = self.state_dict()
    out = OrderedDict((k, sd[k]) for k in self.key_order if k in sd)
    for k, v in sd.items():
        out.setdefault(k, v)
    return out


def _trainer_load(self, path_or_ckpt, load_optimizer: bool = True):
    """Resume from a checkpoint of the reference's layout (its own CheckpointManager files, or ones written by save_checkpoint):
    model_state_dict (module. / model. prefixes stripped, as the plugin does), optimizer state by parameter position, EMA shadow."""
    ck = torch.load(path_or_ckpt, map_location="cpu", weights_only=True) if isinstance(path_or_ckpt, (str, os.PathLike)) else path_or_ckpt
    msd = OrderedDict()
    for k, v in ck["model_state_dict"].items():
        for pre in ("module.", "model."):
            if k.startswith(pre):
                k = k[len(pre):]
        msd[k] = v
    missing = [k for k in self.names if k not in msd]
    if missing:
        raise _lib.FFError(f"checkpoint lacks {len(missing)} trainable tensors, e.g. {missing[:3]}")

    def put(flat, d, names):
        for k in names:
            if tuple(d[k].shape) != self.ref_shape[k]:
                raise _lib.FFError(f"checkpoint tensor {k}: shape {tuple(d[k].shape)} != {self.ref_shape[k]}")
            flat[self.offs[k]:self.offs[k] + d[k].numel()].copy_(to_native(k, d[k].float()).reshape(-1))
    put(self.P, msd, self.names)
    for k in self.buffers:
        if k in msd:
            self.buffers[k].copy_(msd[k].float())
    for k in self.nbt:
        if k in msd:
            self.nbt[k] = int(msd[k])
    for k in list(self.other):
        if k in msd:
            self.other[k] = msd[k].clone()
    self.key_order = list(msd.keys())
    shadow = (ck.get("ema_state_dict") or {}).get("shadow")
    if shadow:
        put(self.EMA, shadow, [k for k in self.names if k in shadow])
        if "decay" in ck["ema_state_dict"]:
            self.hp["ema_decay"] = float(ck["ema_state_dict"]["decay"])
    else:
        self.EMA.copy_(self.P)                                        # train.py:965: EMA re-initialised from the model weights
    if load_optimizer and "optimizer_state_dict" in ck:
        osd = ck["optimizer_state_dict"]
        pkeys = [k for k in msd if is_parameter_key(k) and not k.startswith("expert_ensemble.")]
        steps = set()
        self.M.zero_()
        self.V.zero_()
        for i, st in osd["state"].items():
            k = pkeys[int(i)]
            if k in self.offs:
                put(self.M, {k: st["exp_avg"]}, [k])
                put(self.V, {k: st["exp_avg_sq"]}, [k])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise _lib.FFError(f"optimizer state carries different step counts {sorted(steps)}: per-parameter steps are not supported")
        self.step_count = steps.pop() if steps else 0
        g = osd["param_groups"][0]
        self.hp.update(lr=float(g.get("initial_lr", g["lr"])), betas=tuple(g["betas"]), eps=float(g["eps"]), weight_decay=float(g["weight_decay"]))
    if "ff_trainer" in ck:
        self.seed, self.dropout = int(ck["ff_trainer"]["seed"]), float(ck["ff_trainer"]["dropout"])
    return ck


FusionTrainer.save_checkpoint = _trainer_save
FusionTrainer.load_checkpoint = _trainer_load
FusionTrainer.state_dict_in_order = _state_dict_in_order


# ======================================================================================================================
# Multi-GPU: data-parallel training, one process per GPU, ONE flat all-reduce of the 4 MB gradient buffer per optimizer step
# (the reference itself has no distributed code).  Semantics = torch DDP without SyncBatchNorm: every
# rank runs forward + backward on its own shard of the batch (BatchNorm batch statistics are per rank), gradients are averaged.
def allreduce_mean_(flat: T, world: int) -> T:
    import torch.distributed as dist
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / world) if not flat.is_cuda else _scale_inplace(flat, 1.0 / world)
    return flat


def _scale_inplace(flat: T, k: float):
    ag.k_unary("scale", flat.reshape(1, -1), p0=k, out=flat.reshape(1, -1))


def distributed_step(tr: "FusionTrainer", lr, hr, outs, feats, world: int, lr_now: Optional[float] = None) -> T:
    """One data-parallel step: local forward/backward on this rank's shard, all-reduce(mean) of the flat gradient, identical
    optimizer step on every rank (parameters stay bit-identical across ranks: same averaged gradient, same deterministic kernels)."""
    return tr.step(lr, hr, outs, feats, lr_now, grad_hook=lambda g: allreduce_mean_(g, world))
