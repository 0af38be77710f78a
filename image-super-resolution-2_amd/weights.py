"""Checkpoint layout of the FreqFusion x4 inference path + deterministic synthetic weights.

`param_spec()` enumerates every tensor the eval path reads, under the SAME state-dict keys the
reference model exposes (reference: models/team29_FreqFusion/io.py:141-177 loads
`model_state_dict` into CompleteEnhancedFusionSR whose experts live under
`expert_ensemble.{hat,dat,nafnet}`; key layout follows src/models/hat/hat_arch.py:710-880,
src/models/dat/dat_arch.py:864-990, src/models/nafnet/nafnet_arch.py:137-193,
src/models/enhanced_fusion.py:79-330).  tests/test_weights_spec.py checks it against the
manifest dumped from the reference (tests/golden/state_manifest.json).

`synth_state_dict(seed)` fills that layout with seeded values (numpy Philox keyed by the
parameter name) chosen so that no block is vacuous: the reference's default init zeroes
NAFBlock beta/gamma, BN running stats are 0/1, LKA scales 0.1 ....
No real checkpoints exist offline; the plugin loads real ones through the same keys.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np
import torch

HAT_DEPTH, HAT_GROUPS, DAT_DEPTH, DAT_GROUPS = 6, 12, 6, 6
EMBED = 180
Spec = Tuple[str, Tuple[int, ...], tuple]


def _hat_spec(p: str) -> List[Spec]:
    s: List[Spec] = []

    def lin(n, o, i):
        s.append((f"{p}{n}.weight", (o, i), ("w", 1.0)))
        s.append((f"{p}{n}.bias", (o,), ("b",)))

    def conv(n, o, i, k=3, gain=1.0):
        s.append((f"{p}{n}.weight", (o, i, k, k), ("w", gain)))
        s.append((f"{p}{n}.bias", (o,), ("b",)))

    def ln(n, c=EMBED):
        s.append((f"{p}{n}.weight", (c,), ("g",)))
        s.append((f"{p}{n}.bias", (c,), ("beta",)))

    conv("conv_first", EMBED, 3)
    ln("patch_embed.norm")
    for g in range(HAT_GROUPS):
        for b in range(HAT_DEPTH):
            q = f"layers.{g}.residual_group.blocks.{b}."
            ln(q + "norm1")
            s.append((f"{p}{q}attn.relative_position_bias_table", (31 * 31, 6), ("tbl",)))
            lin(q + "attn.qkv", 3 * EMBED, EMBED)
            lin(q + "attn.proj", EMBED, EMBED)
            conv(q + "conv_block.cab.0", EMBED // 3, EMBED)
            conv(q + "conv_block.cab.2", EMBED, EMBED // 3)
            conv(q + "conv_block.cab.3.attention.1", EMBED // 30, EMBED, 1)
            conv(q + "conv_block.cab.3.attention.3", EMBED, EMBED // 30, 1)
            ln(q + "norm2")
            lin(q + "mlp.fc1", 2 * EMBED, EMBED)
            lin(q + "mlp.fc2", EMBED, 2 * EMBED)
        q = f"layers.{g}.residual_group.overlap_attn."
        s.append((f"{p}{q}relative_position_bias_table", (39 * 39, 6), ("tbl",)))
        ln(q + "norm1")
        lin(q + "qkv", 3 * EMBED, EMBED)
        lin(q + "proj", EMBED, EMBED)
        ln(q + "norm2")
        lin(q + "mlp.fc1", 2 * EMBED, EMBED)
        lin(q + "mlp.fc2", EMBED, 2 * EMBED)
        conv(f"layers.{g}.conv", EMBED, EMBED)
    ln("norm")
    conv("conv_after_body", EMBED, EMBED)
    conv("conv_before_upsample.0", 64, EMBED)
    conv("upsample.0", 256, 64)
    conv("upsample.2", 256, 64)
    conv("conv_last", 3, 64, gain=0.15)
    return s


def _dat_spec(p: str) -> List[Spec]:
    s: List[Spec] = []

    def lin(n, o, i):
        s.append((f"{p}{n}.weight", (o, i), ("w", 1.0)))
        s.append((f"{p}{n}.bias", (o,), ("b",)))

    def conv(n, o, i, k=3, gain=1.0, groups=1):
        s.append((f"{p}{n}.weight", (o, i // groups, k, k), ("w", gain)))
        s.append((f"{p}{n}.bias", (o,), ("b",)))

    def ln(n, c=EMBED):
        s.append((f"{p}{n}.weight", (c,), ("g",)))
        s.append((f"{p}{n}.bias", (c,), ("beta",)))

    def bn(n, c):
        ln(n, c)
        s.append((f"{p}{n}.running_mean", (c,), ("rm",)))
        s.append((f"{p}{n}.running_var", (c,), ("rv",)))

    conv("conv_first", EMBED, 3)
    ln("before_RG.1")
    for g in range(DAT_GROUPS):
        for b in range(DAT_DEPTH):
            q = f"layers.{g}.blocks.{b}."
            ln(q + "norm1")
            lin(q + "attn.qkv", 3 * EMBED, EMBED)
            lin(q + "attn.proj", EMBED, EMBED)
            if b % 2 == 0:
                for br in range(2):
                    r = f"{q}attn.attns.{br}.pos."
                    lin(r + "pos_proj", 5, 2)
                    for j, o in (("pos1", 5), ("pos2", 5), ("pos3", 3)):
                        ln(f"{r}{j}.0", 5)
                        lin(f"{r}{j}.2", o, 5)
            else:
                s.append((f"{p}{q}attn.temperature", (6, 1, 1), ("sc", 1.0, 0.3)))
            conv(q + "attn.dwconv.0", EMBED, EMBED, 3, groups=EMBED)
            bn(q + "attn.dwconv.1", EMBED)
            conv(q + "attn.channel_interaction.1", EMBED // 8, EMBED, 1)
            bn(q + "attn.channel_interaction.2", EMBED // 8)
            conv(q + "attn.channel_interaction.4", EMBED, EMBED // 8, 1)
            conv(q + "attn.spatial_interaction.0", EMBED // 16, EMBED, 1)
            bn(q + "attn.spatial_interaction.1", EMBED // 16)
            conv(q + "attn.spatial_interaction.3", 1, EMBED // 16, 1)
            lin(q + "ffn.fc1", 4 * EMBED, EMBED)
            ln(q + "ffn.sg.norm", 2 * EMBED)
            conv(q + "ffn.sg.conv", 2 * EMBED, 2 * EMBED, 3, groups=2 * EMBED)
            lin(q + "ffn.fc2", EMBED, 2 * EMBED)
            ln(q + "norm2")
        conv(f"layers.{g}.conv", EMBED, EMBED)
    ln("norm")
    conv("conv_after_body", EMBED, EMBED)
    conv("conv_before_upsample.0", 64, EMBED)
    conv("upsample.0", 256, 64)
    conv("upsample.2", 256, 64)
    conv("conv_last", 3, 64, gain=0.15)
    return s


def _naf_block(s: List[Spec], q: str, c: int):
    def conv(n, o, i, k=1, groups=1):
        s.append((f"{q}{n}.weight", (o, i // groups, k, k), ("w", 1.0)))
        s.append((f"{q}{n}.bias", (o,), ("b",)))

    s.append((f"{q}beta", (1, c, 1, 1), ("n", 0.3)))
    s.append((f"{q}gamma", (1, c, 1, 1), ("n", 0.3)))
    conv("conv1", 2 * c, c)
    conv("conv2", 2 * c, 2 * c, 3, groups=2 * c)
    conv("conv3", c, c)
    conv("sca.1", c, c)
    conv("conv4", 2 * c, c)
    conv("conv5", c, c)
    for n in ("norm1", "norm2"):
        s.append((f"{q}{n}.weight", (c,), ("g",)))
        s.append((f"{q}{n}.bias", (c,), ("beta",)))


NAF_ENC, NAF_MID, NAF_DEC, NAF_WIDTH = (2, 2, 4, 8), 12, (2, 2, 2, 2), 64


def _nafnet_spec(p: str) -> List[Spec]:
    s: List[Spec] = []
    s.append((f"{p}intro.weight", (NAF_WIDTH, 3, 3, 3), ("w", 1.0)))
    s.append((f"{p}intro.bias", (NAF_WIDTH,), ("b",)))
    s.append((f"{p}ending.weight", (3, NAF_WIDTH, 3, 3), ("w", 0.05)))
    s.append((f"{p}ending.bias", (3,), ("b",)))
    c = NAF_WIDTH
    for lvl, n in enumerate(NAF_ENC):
        for b in range(n):
            _naf_block(s, f"{p}encoders.{lvl}.{b}.", c)
        c *= 2
    cc = c
    for lvl, n in enumerate(NAF_DEC):
        cc //= 2
        for b in range(n):
            _naf_block(s, f"{p}decoders.{lvl}.{b}.", cc)
    for b in range(NAF_MID):
        _naf_block(s, f"{p}middle_blks.{b}.", c)
    cc = c
    for lvl in range(len(NAF_DEC)):
        s.append((f"{p}ups.{lvl}.0.weight", (2 * cc, cc, 1, 1), ("w", 1.0)))
        cc //= 2
    cc = NAF_WIDTH
    for lvl in range(len(NAF_ENC)):
        s.append((f"{p}downs.{lvl}.weight", (2 * cc, cc, 2, 2), ("w", 1.0)))
        s.append((f"{p}downs.{lvl}.bias", (2 * cc,), ("b",)))
        cc *= 2
    return s


def _fusion_spec() -> List[Spec]:
    s: List[Spec] = []

    def conv(n, o, i, k=3, bias=True, gain=1.0, kh=None, kw=None, groups=1):
        kh = k if kh is None else kh
        kw = k if kw is None else kw
        s.append((f"{n}.weight", (o, i // groups, kh, kw), ("w", gain)))
        if bias:
            s.append((f"{n}.bias", (o,), ("b",)))

    def bn(n, c):
        s.append((f"{n}.weight", (c,), ("g",)))
        s.append((f"{n}.bias", (c,), ("beta",)))
        s.append((f"{n}.running_mean", (c,), ("rm",)))
        s.append((f"{n}.running_var", (c,), ("rv",)))

    s.append(("residual_scale", (), ("sc", 0.1, 0.2)))
    # multi-domain frequency decomposition (multi_domain_frequency.py:66-526)
    m = "multi_domain_freq."
    s.append((m + "dct.band_scale", (3,), ("sc", 1.0, 0.2)))
    s.append((m + "dwt.subband_scale", (4,), ("sc", 1.0, 0.2)))
    s.append((m + "fft.freq_mask_logits", (1, 1, 64, 64), ("fftmask",)))
    s.append((m + "fft.temperature", (), ("sc", 5.0, 0.2)))
    s.append((m + "fft.band_scale", (2,), ("sc", 1.0, 0.2)))
    s.append((m + "band_fusion.dct_importance", (3,), ("sc", 1.0, 0.2)))
    s.append((m + "band_fusion.dwt_importance", (4,), ("sc", 0.8, 0.2)))
    s.append((m + "band_fusion.fft_importance", (2,), ("sc", 0.6, 0.2)))
    for i in range(9):
        conv(f"{m}band_fusion.band_attention.{i}.conv.0", 1, 3)
    conv(m + "band_fusion.fusion_transform.0", 64, 27, 1)
    conv(m + "band_fusion.fusion_transform.2", 9, 64, 1)
    conv(m + "band_fusion.fusion_gate.0", 64, 27, 1)
    conv(m + "band_fusion.fusion_gate.2", 9, 64, 1)
    conv(m + "band_fusion.dct_residual", 9, 9, 1)
    # cross-band attention + LKA (large_kernel_attention.py:38-244)
    c = "cross_band_attn."
    conv(c + "band_proj", 64, 3, 1)
    s.append((c + "band_attention.in_proj_weight", (192, 64), ("w", 1.0)))
    s.append((c + "band_attention.in_proj_bias", (192,), ("b",)))
    s.append((c + "band_attention.out_proj.weight", (64, 64), ("w", 1.0)))
    s.append((c + "band_attention.out_proj.bias", (64,), ("b",)))
    s.append((c + "norm.weight", (64,), ("g",)))
    s.append((c + "norm.bias", (64,), ("beta",)))
    s.append((c + "lka_block.scale1", (), ("sc", 0.1, 0.2)))
    s.append((c + "lka_block.scale2", (), ("sc", 0.1, 0.2)))
    bn(c + "lka_block.norm1", 64)
    conv(c + "lka_block.lka.local_conv", 64, 64, 5, bias=False, groups=64)
    conv(c + "lka_block.lka.h_conv", 64, 64, bias=False, kh=1, kw=21, groups=64)
    conv(c + "lka_block.lka.v_conv", 64, 64, bias=False, kh=21, kw=1, groups=64)
    conv(c + "lka_block.lka.pw_conv", 64, 64, 1, bias=False)
    bn(c + "lka_block.lka.bn", 64)
    bn(c + "lka_block.norm2", 64)
    conv(c + "lka_block.ffn.0", 128, 64, 1)
    conv(c + "lka_block.ffn.2", 64, 128, 1)
    conv(c + "out_proj", 3, 64, 1)
    # hierarchical fusion (hierarchical_fusion.py:67-129)
    h = "multi_res_fusion."
    s.append((h + "residual_weight_1_2", (), ("sc", 0.2, 0.2)))
    s.append((h + "residual_weight_2_3", (), ("sc", 0.2, 0.2)))
    for st, cin, c1, c2 in (("stage1", 9, 64, 64), ("stage2", 73, 64, 64), ("stage3", 73, 64, 32)):
        conv(f"{h}{st}_conv.0", c1, cin)
        conv(f"{h}{st}_conv.2", c2, c1)
        conv(f"{h}{st}_gate.gate.0", c2 // 4, c2, 1)
        conv(f"{h}{st}_gate.gate.2", 1, c2 // 4, 1)
        s.append((f"{h}{st}_res.scale", (), ("sc", 0.1, 0.2)))
        conv(f"{h}{st}_res.block.0", c2, c2, bias=False)
        conv(f"{h}{st}_res.block.2", c2, c2, bias=False)
    conv(h + "to_rgb.0", 16, 32)
    conv(h + "to_rgb.2", 3, 16)
    # multi-scale extractor + dynamic selector (fusion_network.py:167-236,543-607)
    for n in ("conv_1x", "conv_2x", "conv_4x"):
        conv(f"multiscale.{n}.0", 64, 3, bias=False)
        bn(f"multiscale.{n}.2", 64)
    conv("multiscale.fusion", 64, 192, 1, bias=False)
    d = "dynamic_selector."
    conv(d + "difficulty_estimator.0", 64, 3)
    conv(d + "difficulty_estimator.2", 32, 64)
    conv(d + "difficulty_estimator.4", 1, 32)
    conv(d + "expert_gate.0", 64, 64)
    conv(d + "expert_gate.2", 3, 64, 1)
    # refinement (enhanced_fusion.py:266-290)
    conv("refine_net.0", 64, 3)
    conv("refine_net.2", 64, 64)
    conv("refine_net.4", 64, 64)
    conv("refine_net.6", 3, 64)
    # Laplacian edge refinement (edge_enhancement.py:92-180)
    e = "edge_refine."
    s.append((e + "level_weights", (3,), ("sc", 1.0 / 3.0, 0.3)))
    s.append((e + "edge_strength", (), ("sc", 0.15, 0.2)))
    for i in range(3):
        r = f"{e}edge_refiners.{i}."
        conv(r + "conv1", 32, 3)
        conv(r + "conv2", 32, 32)
        conv(r + "conv3", 32, 32)
        conv(r + "proj", 32, 3, 1)
        conv(r + "attn.attn.0", 8, 32, 1)
        conv(r + "attn.attn.2", 1, 8, 3)
    conv(e + "fusion.0", 32, 96)
    conv(e + "fusion.2", 3, 32)
    conv(e + "edge_gate.0", 16, 6)
    conv(e + "edge_gate.2", 1, 16)
    return s


def _collab_spec() -> List[Spec]:
    """EnhancedCollaborativeWithLKA (large_kernel_attention.py:250-330): read only by the cached-mode forward
    (`forward_with_precomputed`, enhanced_fusion.py:756-812) -- the eval forward never passes expert features, so the default
    `parts` of param_spec()/synth_state_dict() leave these keys out (tests/test_weights_spec.py lists them as dead at inference)."""
    s: List[Spec] = []
    c = "collaborative."

    def conv(n, o, i, k=1, bias=True, kh=None, kw=None, groups=1):
        kh = k if kh is None else kh
        kw = k if kw is None else kw
        s.append((f"{c}{n}.weight", (o, i // groups, kh, kw), ("w", 1.0)))
        if bias:
            s.append((f"{c}{n}.bias", (o,), ("b",)))

    def bn(n, ch):
        s.append((f"{c}{n}.weight", (ch,), ("g",)))
        s.append((f"{c}{n}.bias", (ch,), ("beta",)))
        s.append((f"{c}{n}.running_mean", (ch,), ("rm",)))
        s.append((f"{c}{n}.running_var", (ch,), ("rv",)))

    D = 128
    conv("align_layers.hat", D, EMBED)
    conv("align_layers.dat", D, EMBED)
    conv("align_layers.nafnet", D, 64)
    s.append((c + "cross_attn.in_proj_weight", (3 * D, D), ("w", 1.0)))
    s.append((c + "cross_attn.in_proj_bias", (3 * D,), ("b",)))
    s.append((c + "cross_attn.out_proj.weight", (D, D), ("w", 1.0)))
    s.append((c + "cross_attn.out_proj.bias", (D,), ("b",)))
    for n in ("norm1", "norm2"):
        s.append((f"{c}{n}.weight", (D,), ("g",)))
        s.append((f"{c}{n}.bias", (D,), ("beta",)))
    s.append((c + "ffn.0.weight", (2 * D, D), ("w", 1.0)))
    s.append((c + "ffn.0.bias", (2 * D,), ("b",)))
    s.append((c + "ffn.2.weight", (D, 2 * D), ("w", 1.0)))
    s.append((c + "ffn.2.bias", (D,), ("b",)))
    s.append((c + "lka_global.scale1", (), ("sc", 0.1, 0.2)))
    s.append((c + "lka_global.scale2", (), ("sc", 0.1, 0.2)))
    bn("lka_global.norm1", D)
    conv("lka_global.lka.local_conv", D, D, 5, bias=False, groups=D)
    conv("lka_global.lka.h_conv", D, D, bias=False, kh=1, kw=21, groups=D)
    conv("lka_global.lka.v_conv", D, D, bias=False, kh=21, kw=1, groups=D)
    conv("lka_global.lka.pw_conv", D, D, 1, bias=False)
    bn("lka_global.lka.bn", D)
    bn("lka_global.norm2", D)
    conv("lka_global.ffn.0", 2 * D, D)
    conv("lka_global.ffn.2", D, 2 * D)
    for i in range(3):
        conv(f"modulation.{i}.0", D // 4, D)
        conv(f"modulation.{i}.3", 3, D // 4)
    return s


HAT_PREFIX = "expert_ensemble.hat."
DAT_PREFIX = "expert_ensemble.dat."
NAF_PREFIX = "expert_ensemble.nafnet.nafnet."


def param_spec(parts=("hat", "dat", "nafnet", "fusion")) -> List[Spec]:
    out: List[Spec] = []
    if "hat" in parts:
        out += _hat_spec(HAT_PREFIX)
    if "dat" in parts:
        out += _dat_spec(DAT_PREFIX)
    if "nafnet" in parts:
        out += _nafnet_spec(NAF_PREFIX)
    if "fusion" in parts:
        out += _fusion_spec()
    if "collab" in parts:
        out += _collab_spec()
    return out


def _rng(name: str, seed: int) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.Philox(key=int.from_bytes(h[:8], "little")))


def _fill(name: str, shape, kind, seed: int) -> np.ndarray:
    g = _rng(name, seed)
    k = kind[0]
    n = int(np.prod(shape)) if len(shape) else 1
    if k == "w":
        fan_in = int(np.prod(shape[1:]))
        v = g.standard_normal(n, dtype=np.float32) * np.float32(kind[1] / np.sqrt(fan_in))
    elif k == "b":
        v = g.standard_normal(n, dtype=np.float32) * np.float32(0.05)
    elif k == "g":
        v = np.float32(1.0) + g.standard_normal(n, dtype=np.float32) * np.float32(0.1)
    elif k == "beta":
        v = g.standard_normal(n, dtype=np.float32) * np.float32(0.05)
    elif k == "rm":
        v = g.standard_normal(n, dtype=np.float32) * np.float32(0.1)
    elif k == "rv":
        v = (np.float32(0.5) + g.random(n, dtype=np.float32)).astype(np.float32)
    elif k == "tbl":
        v = g.standard_normal(n, dtype=np.float32) * np.float32(0.3)
    elif k == "n":
        v = g.standard_normal(n, dtype=np.float32) * np.float32(kind[1])
    elif k == "sc":
        v = np.float32(kind[1]) * (np.float32(1.0) + np.float32(kind[2]) * (2 * g.random(n, dtype=np.float32) - 1))
    elif k == "fftmask":
        # radial low-pass logits (multi_domain_frequency.py:336-348) plus a seeded perturbation
        size = shape[-1]
        y = np.linspace(-1, 1, size, dtype=np.float32)
        rad = np.sqrt(y[None, :] ** 2 + y[:, None] ** 2)
        v = (3.0 * (0.5 - rad)).astype(np.float32).reshape(-1) + g.standard_normal(n, dtype=np.float32) * np.float32(0.3)
    else:
        raise ValueError(kind)
    return np.asarray(v, dtype=np.float32).reshape(shape)


def synth_state_dict(seed: int = 1234, parts=("hat", "dat", "nafnet", "fusion")) -> "OrderedDict[str, torch.Tensor]":
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape, kind in param_spec(parts):
        sd[name] = torch.from_numpy(_fill(name, shape, kind, seed))
    return sd


def count_params(parts=("hat", "dat", "nafnet", "fusion")) -> int:
    return sum(int(np.prod(s)) if len(s) else 1 for _, s, _ in param_spec(parts))
