"""The three frozen experts sequenced on the HIP kernels: HAT-L, DAT, NAFNet-SR.

Each class takes the reference-keyed state dict, prepares device weights once (repacking, BN folding,
position-bias tables) and exposes forward(lr NCHW [1,3,h,w]) -> SR NCHW [1,3,4h,4w] clamped to [0,1]
with the semantics of ExpertEnsemble.forward_hat/_dat/_nafnet (src/models/expert_loader.py:591-674):
reflect-pad to x16, run, crop, clamp.  Activations are NHWC fp32 device tensors; every arithmetic
step is a libff_hip.so kernel (ops.py), PyTorch only allocates.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import ops
from .prep import (pack_conv, pack_dw, bn_scale_shift, fold_bn_after_conv, pack_token_mlp, pack_token_linear, pack_naf_ffn,
                   pack_win_attn, pack_win_rel, pack_token_projmlp, pack_chan_qkv, pack_token_linear_gated, pack_rel_overlap)

T = torch.Tensor
SD = Dict[str, T]
RGB_MEAN = (0.4488, 0.4371, 0.4040)


def _ceil_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class _Weights:
    """Small helper: fetch tensors by reference key, move to the device once."""

    def __init__(self, sd: SD, dev, prefix: str):
        self.sd, self.dev, self.p = sd, dev, prefix

    def __call__(self, name: str) -> T:
        return self.sd[self.p + name].to(self.dev, torch.float32).contiguous()

    def conv(self, name: str) -> T:
        return pack_conv(self(name + ".weight"))

    def opt(self, name: str) -> Optional[T]:
        k = self.p + name
        return self.sd[k].to(self.dev, torch.float32).contiguous() if k in self.sd else None


def _tl(blk: dict, key: str) -> dict:
    """Lazily packed ff_token_linear weights of blk[key] = (W [N,K], b)."""
    pk = blk.get(key + "_tl")
    if pk is None:
        pk = blk[key + "_tl"] = pack_token_linear(blk[key][0], blk[key][1])
    return pk


# plain bf16: intermediates that are consumed only as bf16 MFMA operands live in HBM as bf16 (needs the v2 attention / proj + MLP kernels)
_BF16_ROWS = (os.environ.get("FF_BF16_ROWS", "1") != "0" and os.environ.get("FF_WF_V2", "1") != "0" and os.environ.get("FF_PM_V2", "1") != "0")
_OCAB_V2 = os.environ.get("FF_OCAB_V2", "1") != "0"          # plain bf16: HAT's overlapping cross-attention in the persistent per-window kernel
_SGFN_TAIL = os.environ.get("FF_SGFN_TAIL", "1") != "0"      # plain bf16: DAT's SpatialGate + fc2 + residual in one launch
_CAB_FUSED = os.environ.get("FF_CAB_FUSED", "0") == "1"        # opt-in: HAT's conv branch in one launch (csrc/cab_fused.hip); bit-identical to the two launches, measured no faster


def _fast() -> bool:
    """The token-stationary fused kernels exist for the default split-bf16 contraction only."""
    return ops.fused_modes()


# Window-resident attention block (LayerNorm + qkv + attention in one launch, csrc/win_attn_fused.hip); FF_WIN_FUSED=0 keeps
# the two-stage form (token_linear qkv -> window_attn) for A/B measurements.
_WIN_FUSED = os.environ.get("FF_WIN_FUSED", "1") != "0"
# Attention projection + residuals + norm2 + MLP in one launch (csrc/token_mlp.hip: token_projmlp_kernel); 0 = two launches
_PROJ_MLP = os.environ.get("FF_PROJ_MLP", "1") != "0"
# DAT SpatialGate: LayerNorm statistics from the fc1 epilogue, normalisation on load in the depth-wise conv: 101 us against
# 45 + 72 us for the separate LayerNorm pass + plain strip kernel (tools/block_timeline.py).  FF_LN_ON_LOAD=0 restores those.
_LN_ON_LOAD = os.environ.get("FF_LN_ON_LOAD", "1") != "0"
# DAT channel attention: LayerNorm + qkv + gram / norms in one launch (csrc/chan_qkv.hip); 0 = token_linear qkv + VALU gram kernel
_CHAN_FUSED = os.environ.get("FF_CHAN_FUSED", "1") != "0"
# DAT: spatial gate + channel gate + mix + projection + residual in one launch (ff_token_linear_gated); 0 = pixel_mlp, mix2, token_linear
_GATED_PROJ = os.environ.get("FF_GATED_PROJ", "1") != "0"


def _pm(blk: dict) -> dict:
    pk = blk.get("projmlp_pk")
    if pk is None:
        pk = blk["projmlp_pk"] = pack_token_projmlp(blk["proj"][0], blk["proj"][1], blk["fc1"][0], blk["fc1"][1],
                                                    blk["fc2"][0], blk["fc2"][1])
    return pk


def _wf(blk: dict, heads: int, d: int) -> dict:
    """Lazily packed ff_win_attn_fused weights of blk["qkv"] = (W [3C, C], b)."""
    pk = blk.get("qkv_wf")
    if pk is None:
        pk = blk["qkv_wf"] = pack_win_attn(blk["qkv"][0], blk["qkv"][1], heads, d, d ** -0.5)
    return pk


# =============================================================================================== HAT
# Gathering the attention bias from the compact relative-position table in LDS removes the 400 MB / launch bias stream but
# measured SLOWER on MI355X (120 vs 108 us per launch: the kernel is issue / LDS bound, not bandwidth bound), so the
# expanded quad-interleaved table stays the default; FF_REL_BIAS=1 selects the gather.
_REL_BIAS = os.environ.get("FF_REL_BIAS", "0") == "1"
_NAF_FRONT = os.environ.get("FF_NAF_FRONT", "1") == "1"      # NAFBlock front half (norm1 + conv1 + conv2 + gate + pool) in one launch at C = 64 / 128
# OCAB (576 keys per window): FF_REL_OCAB=1 gathers the bias from the rotated compact table in LDS instead of streaming the expanded
# one (3.5 MB per head).  Bit-identical, 430 MB less traffic per launch, but 314 vs 207 us on MI355X (two dependent LDS reads per score):
# the expanded table stays the default.
_REL_OCAB = os.environ.get("FF_REL_OCAB", "0") == "1"


def _rel_index_sa(ws: int) -> T:
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    yy, xx = ys.reshape(-1), xs.reshape(-1)
    return (yy[:, None] - yy[None, :] + ws - 1) * (2 * ws - 1) + (xx[:, None] - xx[None, :] + ws - 1)


def _rel_index_oca(ws: int, ows: int) -> T:
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    ye, xe = torch.meshgrid(torch.arange(ows), torch.arange(ows), indexing="ij")
    dy = ye.reshape(-1)[None, :] - ys.reshape(-1)[:, None] + ws - ows + 1
    dx = xe.reshape(-1)[None, :] - xs.reshape(-1)[:, None] + ws - ows + 1
    return dy * (ws + ows - 1) + dx


class HatHIP:
    """HAT-L (hat_arch.py:710-984): 12 RHAG x (6 HAB + OCAB), dim 180, 6 heads, window 16."""

    def __init__(self, sd: SD, dev, prefix: str = "expert_ensemble.hat.", groups: int = 12, depth: int = 6, ws: int = 16,
                 heads: int = 6, conv_scale: float = 0.01):
        w = _Weights(sd, dev, prefix)
        self.dev, self.ws, self.heads, self.groups, self.depth, self.conv_scale = dev, ws, heads, groups, depth, conv_scale
        self.ows = ws + ws // 2
        self.neg_mean = torch.tensor([-m for m in RGB_MEAN], device=dev)
        rpi_sa = _rel_index_sa(ws).reshape(-1).to(dev)
        rpi_oca = _rel_index_oca(ws, self.ows).reshape(-1).to(dev)
        n, nk = ws * ws, self.ows * self.ows

        def lin(name):
            return w(name + ".weight"), w(name + ".bias")

        self.conv_first = (w.conv("conv_first"), w("conv_first.bias"))
        self.pe_norm = (w("patch_embed.norm.weight"), w("patch_embed.norm.bias"))
        self.blocks: List[List[dict]] = []
        self.ocab: List[dict] = []
        self.gconv = []
        for g in range(groups):
            blks = []
            for b in range(depth):
                q = f"layers.{g}.residual_group.blocks.{b}."
                tbl = w(q + "attn.relative_position_bias_table")
                bias = tbl[rpi_sa].reshape(n, n, heads).permute(2, 1, 0).contiguous()      # [heads][key][query]
                blks.append(dict(rel=tbl.t().contiguous(),                                 # [heads][(2ws-1)^2] compact table
                    n1=(w(q + "norm1.weight"), w(q + "norm1.bias")), qkv=lin(q + "attn.qkv"), proj=lin(q + "attn.proj"),
                    bias=bias, cab0=(w.conv(q + "conv_block.cab.0"), w(q + "conv_block.cab.0.bias")),
                    cab2=(w.conv(q + "conv_block.cab.2"), w(q + "conv_block.cab.2.bias")),
                    ca1=(w(q + "conv_block.cab.3.attention.1.weight").reshape(-1, 180).contiguous(), w(q + "conv_block.cab.3.attention.1.bias")),
                    ca2=(w(q + "conv_block.cab.3.attention.3.weight").reshape(180, -1).contiguous(), w(q + "conv_block.cab.3.attention.3.bias")),
                    n2=(w(q + "norm2.weight"), w(q + "norm2.bias")), fc1=lin(q + "mlp.fc1"), fc2=lin(q + "mlp.fc2"),
                    shift=0 if b % 2 == 0 else ws // 2))
            self.blocks.append(blks)
            q = f"layers.{g}.residual_group.overlap_attn."
            tbl = w(q + "relative_position_bias_table")
            self.ocab.append(dict(
                n1=(w(q + "norm1.weight"), w(q + "norm1.bias")), qkv=lin(q + "qkv"), proj=lin(q + "proj"),
                bias=tbl[rpi_oca].reshape(n, nk, heads).permute(2, 1, 0).contiguous(), rel=pack_rel_overlap(tbl, ws, self.ows),
                n2=(w(q + "norm2.weight"), w(q + "norm2.bias")), fc1=lin(q + "mlp.fc1"), fc2=lin(q + "mlp.fc2")))
            self.gconv.append((w.conv(f"layers.{g}.conv"), w(f"layers.{g}.conv.bias")))
        self.norm = (w("norm.weight"), w("norm.bias"))
        self.after = (w.conv("conv_after_body"), w("conv_after_body.bias"))
        self.before_up = (w.conv("conv_before_upsample.0"), w("conv_before_upsample.0.bias"))
        self.up0 = (w.conv("upsample.0"), w("upsample.0.bias"))
        self.up2 = (w.conv("upsample.2"), w("upsample.2.bias"))
        self.last = (w.conv("conv_last"), (w("conv_last.bias") - self.neg_mean).contiguous())   # "+ mean" folded into the bias

    # -- blocks -------------------------------------------------------------------------------
    def _mlp(self, x: T, blk: dict) -> T:
        if ops.fused_modes():                                # fused LN + fc1 + GELU + fc2 + residual, hidden stays on chip
            if "mlp_pk" not in blk:
                blk["mlp_pk"] = pack_token_mlp(blk["fc1"][0], blk["fc1"][1], blk["fc2"][0], blk["fc2"][1])
            return ops.token_mlp(x, blk["n2"][0], blk["n2"][1], blk["mlp_pk"])
        xn = ops.layernorm(x, *blk["n2"])
        h = ops.linear(xn, *blk["fc1"], act="gelu")
        return ops.linear(h, *blk["fc2"], res=x)

    def hab(self, x: T, blk: dict) -> T:
        _, H, W, C = x.shape
        d = C // self.heads
        # plain bf16: att, the normalised rows and conv1's output are consumed only as MFMA operands (rounded to bf16 there), so they are
        # stored as bf16 -- bit-identical results, half the bytes; conv2's output enters x1 times conv_scale = 0.01 and goes the same way
        b16 = (_BF16_ROWS and ops.gemm_mode() == "bf16" and _fast() and _WIN_FUSED and _PROJ_MLP and not _CAB_FUSED and x.shape[0] == 1
               and H * W >= 1024 and C % 4 == 0)
        att = ops.empty_rows_bf16(tuple(x.shape), x.device) if b16 else ops.empty_like_rows(x)
        s = blk["shift"]
        if _fast() and _WIN_FUSED:                                 # norm1 + qkv + (S)W-MSA of all six heads in one launch
            if "relp" not in blk:
                blk["relp"] = pack_win_rel(blk["rel"], self.ws, self.ws)
            _, xn = ops.win_attn_fused(x, att, _wf(blk, self.heads, d), blk["relp"], gamma=blk["n1"][0], beta=blk["n1"][1], H=H, W=W,
                                       Hp=H, Wp=W, win=(self.ws, self.ws), shift=(s, s), use_mask=s > 0, want_xn="bf16" if b16 else True)
        else:
            if _fast():                                            # LayerNorm inside the qkv launch; xn (conv branch) is its side output
                qkv, xn = ops.token_linear(x, _tl(blk, "qkv"), gamma=blk["n1"][0], beta=blk["n1"][1], want_xn=True)
            else:
                xn = ops.layernorm(x, *blk["n1"])
                qkv = ops.linear(xn, *blk["qkv"])
            ops.window_attn(qkv, att, blk["bias"], q_off=0, k_off=C, v_off=2 * C, o_off=0, H=H, W=W, Hp=H, Wp=W, win=(self.ws, self.ws),
                            kwin=(self.ws, self.ws), shift=(s, s), use_mask=s > 0, heads=self.heads, d=d, scale=d ** -0.5,
                            rel_table=blk["rel"] if _REL_BIAS else None)
        if _CAB_FUSED and ops.gemm_mode() == "bf16" and blk["cab0"][0].shape[0] <= 64 and C <= 192 and x.shape[0] == 1:
            c2, c2mean = ops.cab_fused(xn, blk["cab0"][0], blk["cab0"][1], blk["cab2"][0], blk["cab2"][1], partials=True)   # conv -> GELU -> conv + pool, one launch
        else:
            c1 = ops.conv2d(xn, *blk["cab0"], ksize=(3, 3), pad=(1, 1), act="gelu", out_bf16=b16)
            c2, c2mean = ops.conv2d(c1, *blk["cab2"], ksize=(3, 3), pad=(1, 1), want_pool="partials", out_bf16=b16)   # pool partials from the conv epilogue
        gate = ops.vec_mlp(c2mean, *blk["ca1"], "relu", *blk["ca2"], "sigmoid", post=self.conv_scale)
        if _fast() and _PROJ_MLP:                                  # proj + both residuals + norm2 + MLP: x1 never reaches memory
            return ops.token_projmlp(att, x, _pm(blk), blk["n2"][0], blk["n2"][1], c2=c2, c2_scale=gate.reshape(-1))
        if _fast():                                                # shortcut + conv_x*conv_scale + proj(attention), one launch
            x = ops.token_linear(att, _tl(blk, "proj"), res=x, res2=c2, res2_scale=gate.reshape(-1))
        else:
            t = ops.mix2(x, c2, cb=gate)
            x = ops.linear(att, *blk["proj"], res=t)
        return self._mlp(x, blk)

    def ocab_block(self, x: T, blk: dict) -> T:
        _, H, W, C = x.shape
        d = C // self.heads
        if _fast():
            qkv = ops.token_linear(x, _tl(blk, "qkv"), gamma=blk["n1"][0], beta=blk["n1"][1])      # LayerNorm fused
        else:
            qkv = ops.linear(ops.layernorm(x, *blk["n1"]), *blk["qkv"])
        ocab_v2 = _OCAB_V2 and ops.gemm_mode() == "bf16" and (self.ws, self.ows, d) == (16, 24, 30) and H % 16 == 0 and W % 16 == 0
        att = ops.empty_rows_bf16(tuple(x.shape), x.device) if (ocab_v2 and _BF16_ROWS and _fast() and _PROJ_MLP) else ops.empty_like_rows(x)
        if ocab_v2:
            ops.ocab_attn(qkv, att, blk["rel"], q_off=0, k_off=C, v_off=2 * C, H=H, W=W, heads=self.heads, d=d, ws=self.ws, ows=self.ows,
                          scale=d ** -0.5)                 # persistent per-window kernel, compact bias table in LDS (csrc/ocab_attn.hip)
        else:
            ops.window_attn(qkv, att, blk["bias"], q_off=0, k_off=C, v_off=2 * C, o_off=0, H=H, W=W, Hp=H, Wp=W, win=(self.ws, self.ws),
                            kwin=(self.ows, self.ows), shift=(0, 0), use_mask=False, heads=self.heads, d=d, scale=d ** -0.5,
                            rel_table=blk["rel"] if _REL_OCAB else None)
        if _fast() and _PROJ_MLP:
            return ops.token_projmlp(att, x, _pm(blk), blk["n2"][0], blk["n2"][1])
        x = ops.token_linear(att, _tl(blk, "proj"), res=x) if _fast() else ops.linear(att, *blk["proj"], res=x)
        return self._mlp(x, blk)

    def group(self, x: T, g: int) -> T:
        y = x
        for blk in self.blocks[g]:
            y = self.hab(y, blk)
        y = self.ocab_block(y, self.ocab[g])
        return ops.conv2d(y, *self.gconv[g], ksize=(3, 3), pad=(1, 1), res=x)

    def _tail(self, x: T) -> T:
        x = ops.conv2d(x, *self.before_up, ksize=(3, 3), pad=(1, 1), act="lrelu")
        x = ops.conv2d(x, *self.up0, ksize=(3, 3), pad=(1, 1), shuffle=2)
        x = ops.conv2d(x, *self.up2, ksize=(3, 3), pad=(1, 1), shuffle=2)
        return ops.conv2d(x, *self.last, ksize=(3, 3), pad=(1, 1))

    def forward(self, lr: T, taps: Optional[dict] = None, feats: Optional[dict] = None) -> T:
        """feats (optional dict): receives feats["hat"] = the conv_after_body output [1,Hp,Wp,180] (NHWC), the tensor the
        reference's forward hook captures for the cached-expert files (expert_loader.py:838)."""
        _, _, h, w = lr.shape
        Hp, Wp = _ceil_to(h, self.ws), _ceil_to(w, self.ws)
        xin = ops.nchw_to_nhwc(lr, Hp, Wp, add=self.neg_mean, pad_mode="reflect")
        feat = ops.conv2d(xin, *self.conv_first, ksize=(3, 3), pad=(1, 1))
        x = ops.layernorm(feat, *self.pe_norm)
        for g in range(self.groups):
            if taps is not None and g == 0:
                y = x
                for b, blk in enumerate(self.blocks[0]):
                    y = self.hab(y, blk)
                    taps[f"hat.g0.b{b}"] = y
                y = self.ocab_block(y, self.ocab[0])
                taps["hat.g0.ocab"] = y
                x = ops.conv2d(y, *self.gconv[0], ksize=(3, 3), pad=(1, 1), res=x)
                taps["hat.g0.out"] = x
            else:
                x = self.group(x, g)
        x = ops.layernorm(x, *self.norm)
        if feats is not None:
            feats["hat"] = ops.conv2d(x, *self.after, ksize=(3, 3), pad=(1, 1))
            x = ops.mix2(feats["hat"], feat)
        else:
            x = ops.conv2d(x, *self.after, ksize=(3, 3), pad=(1, 1), res=feat)
        sr = self._tail(x)
        return ops.nhwc_to_nchw(sr, 4 * h, 4 * w, clamp01=True)


# =============================================================================================== DAT
def _dat_dpb_bias(sd: SD, p: str, hs: int, wsz: int, heads: int) -> T:
    """DynamicPosBias (dat_arch.py:177-212) is input independent: evaluate it once on the host."""
    import torch.nn.functional as F
    ys, xs = torch.meshgrid(torch.arange(1 - hs, hs), torch.arange(1 - wsz, wsz), indexing="ij")
    t = F.linear(torch.stack([ys.reshape(-1), xs.reshape(-1)], dim=1).float(), sd[p + ".pos_proj.weight"].float().cpu(),
                 sd[p + ".pos_proj.bias"].float().cpu())
    for j in ("pos1", "pos2", "pos3"):
        t = F.layer_norm(t, (t.shape[-1],), sd[f"{p}.{j}.0.weight"].float().cpu(), sd[f"{p}.{j}.0.bias"].float().cpu())
        t = F.linear(F.relu(t), sd[f"{p}.{j}.2.weight"].float().cpu(), sd[f"{p}.{j}.2.bias"].float().cpu())
    cy, cx = torch.meshgrid(torch.arange(hs), torch.arange(wsz), indexing="ij")
    cy, cx = cy.reshape(-1), cx.reshape(-1)
    idx = (cy[:, None] - cy[None, :] + hs - 1) * (2 * wsz - 1) + (cx[:, None] - cx[None, :] + wsz - 1)
    n = hs * wsz
    # expanded [heads][key][query] for the f32 kernel, compact [heads][(2hs-1)*(2wsz-1)] for the LDS gather of the bf16 kernel
    return t[idx.reshape(-1)].reshape(n, n, heads).permute(2, 1, 0).contiguous(), t.t().contiguous()


def dat_should_shift(g: int, b: int) -> bool:
    return (g % 2 == 0 and b > 0 and (b - 2) % 4 == 0) or (g % 2 != 0 and b % 4 == 0)


class DatHIP:
    """DAT (dat_arch.py:864-1028): 6 residual groups x 6 DATB (spatial / channel alternating), dim 180."""

    def __init__(self, sd: SD, dev, prefix: str = "expert_ensemble.dat.", groups: int = 6, depth: int = 6, split=(8, 32),
                 heads: int = 6):
        w = _Weights(sd, dev, prefix)
        self.dev, self.groups, self.depth, self.split, self.heads = dev, groups, depth, split, heads
        self.neg_mean = torch.tensor([-m for m in RGB_MEAN], device=dev)
        C = 180

        def lin(name):
            return w(name + ".weight"), w(name + ".bias")

        def folded_1x1(conv, bn, cin):
            wt = w(conv + ".weight").reshape(-1, cin)
            sc, sh = bn_scale_shift({k: v.to(dev) for k, v in sd.items() if k.startswith(prefix + bn)}, prefix + bn)
            return fold_bn_after_conv(wt, w(conv + ".bias"), sc, sh)

        self.conv_first = (w.conv("conv_first"), w("conv_first.bias"))
        self.before = (w("before_RG.1.weight"), w("before_RG.1.bias"))
        self.blocks: List[List[dict]] = []
        self.gconv = []
        for g in range(groups):
            blks = []
            for b in range(depth):
                q = f"layers.{g}.blocks.{b}."
                sc, sh = bn_scale_shift({k: v.to(dev) for k, v in sd.items() if k.startswith(prefix + q + "attn.dwconv.1")},
                                        prefix + q + "attn.dwconv.1")
                ci1 = folded_1x1(q + "attn.channel_interaction.1", q + "attn.channel_interaction.2", C)
                si0 = folded_1x1(q + "attn.spatial_interaction.0", q + "attn.spatial_interaction.1", C)
                blk = dict(
                    spatial=(b % 2 == 0), shifted=dat_should_shift(g, b),
                    n1=(w(q + "norm1.weight"), w(q + "norm1.bias")), qkv=lin(q + "attn.qkv"), proj=lin(q + "attn.proj"),
                    dw=(pack_dw(w(q + "attn.dwconv.0.weight")), w(q + "attn.dwconv.0.bias"), sc, sh),
                    ci1=ci1, ci4=(w(q + "attn.channel_interaction.4.weight").reshape(C, -1).contiguous(), w(q + "attn.channel_interaction.4.bias")),
                    si0=si0, si3=(w(q + "attn.spatial_interaction.3.weight").reshape(1, -1).contiguous(), w(q + "attn.spatial_interaction.3.bias")),
                    n2=(w(q + "norm2.weight"), w(q + "norm2.bias")), fc1=lin(q + "ffn.fc1"),
                    sgn=(w(q + "ffn.sg.norm.weight"), w(q + "ffn.sg.norm.bias")),
                    sgc=(pack_dw(w(q + "ffn.sg.conv.weight")), w(q + "ffn.sg.conv.bias")), fc2=lin(q + "ffn.fc2"))
                if blk["spatial"]:
                    pb = [_dat_dpb_bias(sd, prefix + q + "attn.attns.0.pos", split[0], split[1], heads // 2),
                          _dat_dpb_bias(sd, prefix + q + "attn.attns.1.pos", split[1], split[0], heads // 2)]
                    blk["bias"] = [pb[0][0].to(dev), pb[1][0].to(dev)]
                    blk["rel"] = [pb[0][1].to(dev), pb[1][1].to(dev)]
                else:
                    blk["temp"] = w(q + "attn.temperature").reshape(-1).contiguous()
                blk["si3b"] = float(blk["si3"][1].reshape(-1)[0].cpu())                    # scalar bias of the 11 -> 1 layer
                blks.append(blk)
            self.blocks.append(blks)
            self.gconv.append((w.conv(f"layers.{g}.conv"), w(f"layers.{g}.conv.bias")))
        self.norm = (w("norm.weight"), w("norm.bias"))
        self.after = (w.conv("conv_after_body"), w("conv_after_body.bias"))
        self.before_up = (w.conv("conv_before_upsample.0"), w("conv_before_upsample.0.bias"))
        self.up0 = (w.conv("upsample.0"), w("upsample.0.bias"))
        self.up2 = (w.conv("upsample.2"), w("upsample.2.bias"))
        self.last = (w.conv("conv_last"), (w("conv_last.bias") - self.neg_mean).contiguous())

    def block(self, x: T, blk: dict) -> T:
        _, H, W, C = x.shape
        half, hh = C // 2, self.heads // 2
        d = half // hh
        fused = blk["spatial"] and _fast() and _WIN_FUSED
        if fused:
            # norm1 + qkv + both window-attention branches: two launches (head groups 0-2 on 8x32 windows, 3-5 on 32x8), each
            # also writes its half of v for the depth-wise conv branch (dat_arch.py:524); q and k never reach memory
            m = max(self.split)
            Hp, Wp = _ceil_to(H, m), _ceil_to(W, m)
            att = ops.empty_like_rows(x)
            v = ops.empty_like_rows(x)
            pk = _wf(blk, self.heads, d)
            if "relp" not in blk:
                blk["relp"] = []
                for br in range(2):
                    wh, ww = (self.split[0], self.split[1]) if br == 0 else (self.split[1], self.split[0])
                    r6 = torch.zeros(self.heads, blk["rel"][br].shape[1], device=x.device)
                    r6[hh * br:hh * br + hh] = blk["rel"][br]
                    blk["relp"].append(pack_win_rel(r6, wh, ww))
            for br in range(2):
                wh, ww = (self.split[0], self.split[1]) if br == 0 else (self.split[1], self.split[0])
                sh = (wh // 2, ww // 2) if blk["shifted"] else (0, 0)
                ops.win_attn_fused(x, att, pk, blk["relp"][br], gamma=blk["n1"][0], beta=blk["n1"][1], H=H, W=W, Hp=Hp, Wp=Wp,
                                   win=(wh, ww), shift=sh, use_mask=blk["shifted"], head0=hh * br, nheads=hh, zero_pad=True,
                                   v_out=v, v_off=0)
        elif not blk["spatial"] and _fast() and _CHAN_FUSED:
            # norm1 + qkv + per-head gram / norms over all tokens in one launch: only v reaches memory (dat_arch.py:617-641)
            if "cq_pk" not in blk:
                blk["cq_pk"] = pack_chan_qkv(blk["qkv"][0], blk["qkv"][1], self.heads, C // self.heads)
            v, wbd = ops.chan_qkv_attn(x, blk["cq_pk"], blk["n1"][0], blk["n1"][1], blk["temp"])
            qkv = None
        else:
            if _fast():
                qkv = ops.token_linear(x, _tl(blk, "qkv"), gamma=blk["n1"][0], beta=blk["n1"][1])   # LayerNorm fused
            else:
                qkv = ops.linear(ops.layernorm(x, *blk["n1"]), *blk["qkv"])          # [1,H,W,3C] = q | k | v
            v = qkv[..., 2 * C:]
        conv_x = ops.dwconv2d(v, blk["dw"][0], blk["dw"][1], post_scale=blk["dw"][2], post_shift=blk["dw"][3], act="gelu")
        if blk["spatial"]:
            if not fused:
                m = max(self.split)
                Hp, Wp = _ceil_to(H, m), _ceil_to(W, m)
                att = ops.empty_like_rows(x)
                for br in range(2):
                    wh, ww = (self.split[0], self.split[1]) if br == 0 else (self.split[1], self.split[0])
                    sh = (wh // 2, ww // 2) if blk["shifted"] else (0, 0)
                    ops.window_attn(qkv, att, blk["bias"][br], q_off=br * half, k_off=C + br * half, v_off=2 * C + br * half,
                                    o_off=br * half, H=H, W=W, Hp=Hp, Wp=Wp, win=(wh, ww), kwin=(wh, ww), shift=sh,
                                    use_mask=blk["shifted"], heads=hh, d=d, scale=d ** -0.5,
                                    rel_table=blk["rel"][br] if _REL_BIAS else None)
            ch_in, sp_in = conv_x, att
        else:
            if qkv is not None:
                wbd = ops.chan_attn_weights(qkv, 0, C, blk["temp"])
            att = ops.linear(v, wbd, dynamic_w=True)
            ch_in, sp_in = att, conv_x
        cm = ops.vec_mlp(ops.pool_partials(ch_in), *blk["ci1"], "gelu", *blk["ci4"], "sigmoid")          # [1,C]; pool finish inside the MLP launch
        if _fast() and _GATED_PROJ:
            # spatial gate (from sp_in, applied to ch_in) + channel gate (applied to sp_in) + projection + residual in one launch
            if "gp_pk" not in blk:
                blk["gp_pk"] = pack_token_linear_gated(blk["proj"][0], blk["proj"][1], blk["si0"][0], blk["si0"][1], blk["si3"][0])
            x = ops.token_linear_gated(sp_in, ch_in, blk["gp_pk"], cm.reshape(-1), blk["si3b"], res=x)
            return self._sgfn(x, blk)
        if _fast():                                                # 180 -> 11 -> 1 per pixel in one pass (fp32)
            sm = ops.pixel_mlp(sp_in, blk["si0"][0], blk["si0"][1], "gelu", blk["si3"][0], blk["si3b"], "sigmoid")
        else:
            sm = ops.linear(ops.linear(sp_in, *blk["si0"], act="gelu"), *blk["si3"], act="sigmoid")      # [1,H,W,1]
        if blk["spatial"]:
            fused = ops.mix2(att, conv_x, ca=cm, pb=sm)
        else:
            fused = ops.mix2(att, conv_x, pa=sm, cb=cm)
        x = ops.token_linear(fused, _tl(blk, "proj"), res=x) if _fast() else ops.linear(fused, *blk["proj"], res=x)
        return self._sgfn(x, blk)

    def _sgfn(self, x: T, blk: dict) -> T:
        """x + SGFN(norm2(x)) (dat_arch.py:155-170, :736)."""
        if _fast() and _LN_ON_LOAD:
            # fc1 + GELU also emits the SpatialGate LayerNorm statistics of its upper half; the depth-wise conv normalises on load
            c2 = blk["fc1"][0].shape[0] // 2
            y, stats = ops.token_linear(x, _tl(blk, "fc1"), gamma=blk["n2"][0], beta=blk["n2"][1], act="gelu", stats_range=(c2, 2 * c2))
            if _SGFN_TAIL and ops.gemm_mode() == "bf16" and blk["fc2"][0].shape[0] <= 192 and c2 % 4 == 0 and c2 <= 512:
                # SpatialGate + fc2 + residual in one launch: the gate product never reaches memory (csrc/sgfn_tail.hip)
                return ops.sgfn_tail(y, c2, blk["sgc"][0], blk["sgc"][1], stats, blk["sgn"][0], blk["sgn"][1], blk["fc2"][0], blk["fc2"][1], res=x)
            z = ops.dwconv3x3_ln(y[..., c2:], blk["sgc"][0], blk["sgc"][1], stats, blk["sgn"][0], blk["sgn"][1], mul_in=y[..., :c2])
            return ops.linear(z, *blk["fc2"], res=x)
        if _fast():
            y = ops.token_linear(x, _tl(blk, "fc1"), gamma=blk["n2"][0], beta=blk["n2"][1], act="gelu")
        else:
            y = ops.linear(ops.layernorm(x, *blk["n2"]), *blk["fc1"], act="gelu")
        c2 = y.shape[-1] // 2
        gte = ops.layernorm(y[..., c2:], *blk["sgn"])
        z = ops.dwconv2d(gte, *blk["sgc"], mul_in=y[..., :c2])                     # dw3x3(LN(x2)) * x1 in one pass
        return ops.linear(z, *blk["fc2"], res=x)

    def forward(self, lr: T, taps: Optional[dict] = None, feats: Optional[dict] = None) -> T:
        _, _, h, w = lr.shape
        Hp, Wp = _ceil_to(h, 16), _ceil_to(w, 16)
        xin = ops.nchw_to_nhwc(lr, Hp, Wp, add=self.neg_mean, pad_mode="reflect")
        feat = ops.conv2d(xin, *self.conv_first, ksize=(3, 3), pad=(1, 1))
        x = ops.layernorm(feat, *self.before)
        for g in range(self.groups):
            y = x
            for b, blk in enumerate(self.blocks[g]):
                y = self.block(y, blk)
                if taps is not None and g < 2:
                    taps[f"dat.g{g}.b{b}"] = y
            x = ops.conv2d(y, *self.gconv[g], ksize=(3, 3), pad=(1, 1), res=x)
        x = ops.layernorm(x, *self.norm)
        if feats is not None:                                  # expert_loader.py:850: hook on conv_after_body
            feats["dat"] = ops.conv2d(x, *self.after, ksize=(3, 3), pad=(1, 1))
            x = ops.mix2(feats["dat"], feat)
        else:
            x = ops.conv2d(x, *self.after, ksize=(3, 3), pad=(1, 1), res=feat)
        x = ops.conv2d(x, *self.before_up, ksize=(3, 3), pad=(1, 1), act="lrelu")
        x = ops.conv2d(x, *self.up0, ksize=(3, 3), pad=(1, 1), shuffle=2)
        x = ops.conv2d(x, *self.up2, ksize=(3, 3), pad=(1, 1), shuffle=2)
        sr = ops.conv2d(x, *self.last, ksize=(3, 3), pad=(1, 1))
        return ops.nhwc_to_nchw(sr, 4 * h, 4 * w, clamp01=True)


# =============================================================================================== NAFNet-SR
class NafnetHIP:
    """NAFNetSR (nafnet/__init__.py:117-139 + nafnet_arch.py:137-225): bicubic x4, then the width-64
    UNet (enc 2/2/4/8, mid 12, dec 2/2/2/2) at HR resolution."""

    def __init__(self, sd: SD, dev, prefix: str = "expert_ensemble.nafnet.nafnet.", enc=(2, 2, 4, 8), mid: int = 12,
                 dec=(2, 2, 2, 2)):
        w = _Weights(sd, dev, prefix)
        self.dev, self.enc_n, self.mid_n, self.dec_n = dev, enc, mid, dec

        def blk(q):
            def c1(n):
                wt = w(q + n + ".weight")
                return wt.reshape(wt.shape[0], wt.shape[1]).contiguous(), w(q + n + ".bias")
            return dict(n1=(w(q + "norm1.weight"), w(q + "norm1.bias")), n2=(w(q + "norm2.weight"), w(q + "norm2.bias")),
                        c1=c1("conv1"), c2=(pack_dw(w(q + "conv2.weight")), w(q + "conv2.bias")), c3=c1("conv3"),
                        sca=c1("sca.1"), c4=c1("conv4"), c5=c1("conv5"),
                        beta=w(q + "beta").reshape(-1).contiguous(), gamma=w(q + "gamma").reshape(-1).contiguous())

        self.intro = (w.conv("intro"), w("intro.bias"))
        self.ending = (w.conv("ending"), w("ending.bias"))
        self.encs = [[blk(f"encoders.{l}.{b}.") for b in range(n)] for l, n in enumerate(enc)]
        self.mids = [blk(f"middle_blks.{b}.") for b in range(mid)]
        self.decs = [[blk(f"decoders.{l}.{b}.") for b in range(n)] for l, n in enumerate(dec)]
        self.downs = [(w.conv(f"downs.{l}"), w(f"downs.{l}.bias")) for l in range(len(enc))]
        self.ups = [w(f"ups.{l}.0.weight").reshape(w(f"ups.{l}.0.weight").shape[0], -1).contiguous() for l in range(len(dec))]

    def _padded_input(self, B: int, Hp: int, Wp: int, dev) -> T:
        """[B,Hp,Wp,3] buffer whose padding (check_image_size, nafnet_arch.py:220-225: zeros right/bottom) was zeroed ONCE when
        the buffer was made; the resampler only ever writes the top-left H x W region, so no fill runs per forward.  Owned by the
        graph entry being captured (its address is in the captured launches) or, for eager forwards, one buffer that is replaced
        when the size changes (ops.persistent_zeros)."""
        return ops.persistent_zeros(id(self), "naf_in", (B, Hp, Wp, 3), dev)

    def __del__(self):
        try:
            ops.drop_persistent(owner=id(self))
        except Exception:                                   # interpreter shutdown
            pass

    def block(self, x: T, k: dict) -> T:
        c = x.shape[-1]
        flash = _fast() and c in (64, 128)                                   # HR levels: the bandwidth-bound ones
        if flash and _NAF_FRONT and c == 64:                                 # LayerNorm2d + conv1 + conv2 + SimpleGate + pool sums: one launch (313 vs 381 us at
                                                                             # 1024 x 1024; the 128-channel form fits one workgroup per CU only and ties: 222 vs 217)
            g, pooled = ops.naf_front(x, _tl(k, "c1"), k["n1"][0], k["n1"][1], *k["c2"])
        else:
            if flash:
                t = ops.token_linear(x, _tl(k, "c1"), gamma=k["n1"][0], beta=k["n1"][1], eps=1e-6)    # LayerNorm2d + conv1
            else:
                t = ops.linear(ops.layernorm(x, *k["n1"], eps=1e-6), *k["c1"])
            g, pooled = ops.dwconv3_gate_pool(t, *k["c2"])                   # conv2 + SimpleGate + SCA pool sums
        sca = ops.vec_mlp(pooled, *k["sca"], None)                           # [1,c]
        nf = ops.gemm_mode() != "f32"                                        # the split-bf16 GEMM carries both NAFBlock fusions
        if nf:                                                               # conv3(g * sca): the scale rides on the A operand
            y = ops.linear(g, *k["c3"], res=x, mul=k["beta"], kmul=sca.reshape(-1))
        else:
            w3 = ops.mix2(k["c3"][0], ca=sca.reshape(-1))                    # conv3(g * sca) == (W3 . diag(sca)) g
            y = ops.linear(g, w3, k["c3"][1], res=x, mul=k["beta"], dynamic_w=True)
        if flash:
            if "ffn_pk" not in k:
                k["ffn_pk"] = pack_naf_ffn(k["c4"][0], k["c4"][1], k["c5"][0], k["c5"][1])
            return ops.naf_ffn(y, k["ffn_pk"], k["n2"][0], k["n2"][1], k["gamma"])
        if nf:                                                               # conv4 + SimpleGate in one launch: the 2c-wide tensor stays on chip
            if "c4i" not in k:                                               # rows interleaved: 2j <- j, 2j+1 <- j + c (chunk(2, dim=1) halves)
                w4, b4 = k["c4"]
                k["c4i"] = (torch.stack((w4[:c], w4[c:]), dim=1).reshape(2 * c, -1).contiguous(),
                            torch.stack((b4[:c], b4[c:]), dim=1).reshape(-1).contiguous())
            g = ops.linear(ops.layernorm(y, *k["n2"], eps=1e-6), *k["c4i"], gate_pairs=True)
        else:
            t = ops.linear(ops.layernorm(y, *k["n2"], eps=1e-6), *k["c4"])
            g = ops.fma3(None, t[..., :c], t[..., c:])
        return ops.linear(g, *k["c5"], res=y, mul=k["gamma"])

    def forward(self, lr: T, taps: Optional[dict] = None, feats: Optional[dict] = None) -> T:
        _, _, h, w = lr.shape
        H, W = 4 * h, 4 * w
        m = 2 ** len(self.enc_n)
        Hp, Wp = _ceil_to(H, m), _ceil_to(W, m)
        inp = self._padded_input(1, Hp, Wp, lr.device)
        ops.resize(lr, (H, W), mode="bicubic", scale_factor=4.0, layout="nchw", out=inp)
        x = ops.conv2d(inp, *self.intro, ksize=(3, 3), pad=(1, 1))
        skips = []
        for lvl, blks in enumerate(self.encs):
            for b, k in enumerate(blks):
                x = self.block(x, k)
                if taps is not None and lvl == 0:
                    taps[f"naf.enc0.b{b}"] = x
            skips.append(x)
            x = ops.conv2d(x, *self.downs[lvl], ksize=(2, 2), stride=(2, 2), pad=(0, 0))
        for k in self.mids:
            x = self.block(x, k)
        if taps is not None:
            taps["naf.mid"] = x
        for lvl, blks in enumerate(self.decs):
            x = ops.conv2d(x, self.ups[lvl], None, shuffle=2, res=skips[-1 - lvl])
            for k in blks:
                x = self.block(x, k)
        if feats is not None:
            feats["nafnet"] = x                                # expert_loader.py:866: the INPUT of the ending conv, [1,Hp,Wp,64]
        x = ops.conv2d(x, *self.ending, ksize=(3, 3), pad=(1, 1), res=inp)
        return ops.nhwc_to_nchw(x, H, W, clamp01=True)
