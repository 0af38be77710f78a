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


def pack_sgfn_fc2(w2: T) -> T:
    """fc2 [N, K] of ff_sgfn_tail -> bf16 tiles [ceil(K/32)][192][32] (tile c, row n, column kk = w2[n][32 c + kk]; zero padded)."""
    n, k = w2.shape
    assert n <= 192
    ht = (k + 31) // 32
    t = torch.zeros(192, ht * 32, device=w2.device, dtype=torch.float32)
    t[:n, :k] = w2
    return t.reshape(192, ht, 32).permute(1, 0, 2).contiguous().to(torch.bfloat16)


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


def split_bf16(w: T) -> Tuple[T, T]:
    """fp32 -> (hi, lo) bf16 with hi = rne(w), lo = rne(w - hi): the operand split of the bf16x3 kernels."""
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


def pack_token_mlp(w1: T, b1: T, w2: T, b2: T) -> dict:
    """Weights of ff_token_mlp: fc1 [hidden, K], fc2 [N, hidden] -> padded / tiled / permuted bf16 planes."""
    hidden, K = w1.shape
    N = w2.shape[0]
    assert K <= 192 and N <= 192 and w2.shape[1] == hidden
    ht = (hidden + 31) // 32
    dev = w1.device
    w1p = torch.zeros(ht * 32, 192, device=dev)
    w1p[:hidden, :K] = w1
    b1p = torch.zeros(ht * 32, device=dev)
    b1p[:hidden] = b1
    w2p = torch.zeros(192, ht * 32, device=dev)
    w2p[:N, :hidden] = w2
    pos = torch.arange(32)
    src = (pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1)          # stored[pos] = w2[.., swap23(pos)]
    w2t = w2p.reshape(192, ht, 32)[:, :, src.to(dev)].permute(1, 0, 2).contiguous()   # [ht][192][32]
    w1h, w1l = split_bf16(w1p.reshape(ht, 32 * 192))
    w2h, w2l = split_bf16(w2t.reshape(ht, 192 * 32))
    tiles = torch.stack([w1h, w1l, w2h, w2l], dim=1).contiguous()                     # [ht][4][6144]
    return dict(ht=ht, K=K, N=N, w=tiles, b1=b1p.contiguous(), b2=b2.contiguous())


def quad_bias(biasT: T) -> T:
    """[heads][nk][256] key-major attention bias -> quad-interleaved [heads][nk/4][256][4] for ff_window_attn_bf16s."""
    heads, nk, nq = biasT.shape
    assert nk % 4 == 0 and nq == 256
    return biasT.reshape(heads, nk // 4, 4, nq).permute(0, 1, 3, 2).contiguous()


def halo_bn(cout: int) -> int:
    """Output channels per workgroup of ff_conv3x3_halo: least padding, ties to the wider tile."""
    import os
    force = os.environ.get("FF_HALO_BN")                      # tuning switch: "32" / "64" force narrower output blocks
    if force and int(force) in (32, 64) and cout > int(force):
        return int(force)
    if cout <= 32:
        return 32
    return min((192, 128, 64), key=lambda bn: (-(-cout // bn) * bn, -bn))


def pack_conv3x3_halo(wp: T, cin: int, bn: int, nterms: int = 3) -> T:
    """Weight image of ff_conv3x3_halo from a packed 3x3 weight [Cout, 9*cin] (tap-major, pack_conv):
    bf16 [nblk][nchunk][9][64/wk][bn rows x (wk hi | wk lo | 8 pad)], every record padded to 1 KiB; wk = channels per
    weight tile = 32 for bn 192, 64 otherwise (csrc/conv3x3_halo.hip launch table).  nterms = 1 (plain bf16): compact rows
    (wk hi | 8 pad) -- the kernel's LDS image then holds no lo plane."""
    cout = wp.shape[0]
    assert wp.shape[1] == 9 * cin
    wk = 32 if bn == 192 else 64
    nh = 64 // wk
    nblk, nchunk = -(-cout // bn), -(-cin // 64)
    w = torch.zeros(nblk * bn, 9, nchunk * 64, device=wp.device)
    w[:cout, :, :cin] = wp.reshape(cout, 9, cin)
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    rw = (2 * wk if nterms == 3 else wk) + 8
    rows = torch.zeros(nblk, nchunk, 9, nh, bn, rw, device=wp.device, dtype=torch.bfloat16)
    rows[..., :wk] = hi.reshape(nblk, bn, 9, nchunk, nh, wk).permute(0, 3, 2, 4, 1, 5)
    if nterms == 3:
        rows[..., wk:2 * wk] = lo.reshape(nblk, bn, 9, nchunk, nh, wk).permute(0, 3, 2, 4, 1, 5)
    rec = bn * rw
    slot = (rec * 2 + 1023) // 1024 * 1024
    img = torch.zeros(nblk * nchunk * 9 * nh, slot // 2, device=wp.device, dtype=torch.bfloat16)
    img[:, :rec] = rows.reshape(nblk * nchunk * 9 * nh, rec)
    return img.contiguous()


def pack_rel_overlap(table: T, wh: int, kh: int) -> T:
    """Relative-position table of an OVERLAPPING window attention (HAT OCAB: hat_arch.py:882-899 builds the index with the offset
    a = wh - kh + 1 < 0, so the reference gathers with negative indices that PyTorch wraps) -> [heads][n] rotated so that the
    kernel's non-negative index i = m + (original index) reads U[i] = T[(i - m) mod n]; m = ((wh-1) - a) * (rel_w + 1).
    table: [(wh+kh-1)^2, heads] as stored in the checkpoint (square windows)."""
    rel_w = wh + kh - 1
    n = rel_w * rel_w
    if tuple(table.shape[:1]) != (n,):
        raise ValueError(f"pack_rel_overlap: expected a table of {n} rows, got {tuple(table.shape)}")
    a = wh - kh + 1
    m = ((wh - 1) - a) * (rel_w + 1)
    idx = (torch.arange(n, device=table.device) - m) % n
    return table[idx].t().contiguous()


def pack_token_linear(w: T, b: Optional[T]) -> dict:
    """Weights of ff_token_linear: [N, K<=192] -> bf16 [NT][2][32][kpad] hi/lo tiles (kpad = 64/128/192), bias padded."""
    N, K = w.shape
    assert K <= 192
    kpad = 64 if K <= 64 else (128 if K <= 128 else 192)
    nt = (N + 31) // 32
    wp = torch.zeros(nt * 32, kpad, device=w.device)
    wp[:N, :K] = w
    hi, lo = split_bf16(wp.reshape(nt, 32 * kpad))
    bp = None
    if b is not None:
        bp = torch.zeros(nt * 32, device=w.device)
        bp[:N] = b
    return dict(nt=nt, K=K, kpad=kpad, N=N, w=torch.stack([hi, lo], dim=1).contiguous(), b=bp)


def pack_naf_ffn(w4: T, b4: T, w5: T, b5: T) -> dict:
    """Weights of ff_naf_ffn: conv4 [2C, C], conv5 [C, C] -> per gated 32-channel tile g the record
    [W4a_hi, W4a_lo, W4b_hi, W4b_lo (32 x C: rows 32g.. and C+32g..), W5_hi, W5_lo (C x 32, columns 32g.. permuted)]."""
    C = w5.shape[0]
    assert C in (64, 128) and tuple(w4.shape) == (2 * C, C) and tuple(w5.shape) == (C, C)
    gt = C // 32
    pos = torch.arange(32)
    src = ((pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1)).to(w4.device)
    recs = []
    for g in range(gt):
        a = w4[32 * g:32 * g + 32].reshape(-1)
        b = w4[C + 32 * g:C + 32 * g + 32].reshape(-1)
        w5t = w5[:, 32 * g:32 * g + 32][:, src].reshape(-1)
        ah, al = split_bf16(a)
        bh, bl = split_bf16(b)
        wh, wl = split_bf16(w5t)
        recs.append(torch.cat([ah, al, bh, bl, wh, wl]))
    return dict(C=C, w=torch.stack(recs).contiguous(), b4=b4.contiguous(), b5=b5.contiguous())


def win_rel_stride(ww: int) -> int:
    """Row stride of the compact bias table of ff_win_attn_fused (>= 2 ww - 1, == ww mod 32: bank-conflict free gathers)."""
    return {8: 40, 16: 48, 32: 64}[ww]


def pack_win_rel(rel: T, wh: int, ww: int) -> T:
    """[heads][(2wh-1)*(2ww-1)] compact relative-position table -> [heads][2wh-1][stride] (zero padded rows)."""
    heads = rel.shape[0]
    rows, cols, st = 2 * wh - 1, 2 * ww - 1, win_rel_stride(ww)
    out = torch.zeros(heads, rows, st, device=rel.device, dtype=torch.float32)
    out[:, :, :cols] = rel.reshape(heads, rows, cols)
    return out.contiguous()


def pack_win_attn(wqkv: T, bqkv: Optional[T], heads: int, d: int, scale: float) -> dict:
    """Weights of ff_win_attn_fused from a fused qkv projection [3C, K] (rows q | k | v, head-major inside each third):
    tile 3g+j = rows of head g of q / k / v, padded to 32 rows x 192 columns, bf16 hi/lo planes; softmax scale folded into q."""
    C3, K = wqkv.shape
    C = C3 // 3
    assert K <= 192 and heads * d == C and d <= 32
    dev = wqkv.device
    wt = torch.zeros(3 * heads, 32, 192, device=dev)
    bt = torch.zeros(3 * heads, 32, device=dev)
    for g in range(heads):
        for j in range(3):
            rows = slice(j * C + g * d, j * C + (g + 1) * d)
            sc = scale if j == 0 else 1.0
            wt[3 * g + j, :d, :K] = wqkv[rows] * sc
            if bqkv is not None:
                bt[3 * g + j, :d] = bqkv[rows] * sc
    hi, lo = split_bf16(wt.reshape(3 * heads, 32 * 192))
    return dict(w=torch.stack([hi, lo], dim=1).contiguous(), b=bt.reshape(-1).contiguous(), heads=heads, d=d, K=K)


def projmlp_k_perm() -> T:
    """fc1 K-column order of ff_token_projmlp: operand position 16 st + 8 hh + j holds channel
    32 (st >> 1) + 8 (2 (st & 1) + (j >> 2)) + 4 hh + (j & 3) -- the order in which the projection's accumulator registers
    hold a token's channels (csrc/token_mlp.hip)."""
    k = torch.arange(192)
    st, hh, j = k // 16, (k % 16) // 8, k % 8
    return 32 * (st // 2) + 8 * (2 * (st % 2) + j // 4) + 4 * hh + (j % 4)


def pack_token_projmlp(wp: T, bp: T, w1: T, b1: T, w2: T, b2: T) -> dict:
    """Weights of ff_token_projmlp: the projection in ff_token_linear's format, the MLP in ff_token_mlp's with fc1's columns
    permuted (zero columns for the K padding)."""
    hidden, K = w1.shape
    assert K <= 192 and tuple(wp.shape) == (K, K)
    w1p = torch.zeros(hidden, 192, device=w1.device)
    w1p[:, :K] = w1
    w1perm = w1p[:, projmlp_k_perm().to(w1.device)].contiguous()
    mlp = pack_token_mlp(w1perm, b1, w2, b2)
    mlp["K"] = K
    return dict(proj=pack_token_linear(wp, bp), mlp=mlp)


def pack_chan_qkv(wqkv: T, bqkv: Optional[T], heads: int = 6, d: int = 30) -> dict:
    """Weights of ff_chan_qkv from DAT's fused qkv projection [3C, K]: tiles q_0, k_0, ..., q_5, k_5 (each head's 30 rows padded to
    32), then the v rows in six 32-row tiles; bf16 hi/lo planes, K padded to 192."""
    C3, K = wqkv.shape
    C = C3 // 3
    assert C == heads * d == 180 and K <= 192
    dev = wqkv.device
    wt = torch.zeros(18, 32, 192, device=dev)
    bt = torch.zeros(18, 32, device=dev)
    for h in range(heads):
        for j in range(2):
            rows = slice(j * C + h * d, j * C + (h + 1) * d)
            wt[2 * h + j, :d, :K] = wqkv[rows]
            if bqkv is not None:
                bt[2 * h + j, :d] = bqkv[rows]
    for t in range(6):
        n = min(32, C - 32 * t)
        wt[12 + t, :n, :K] = wqkv[2 * C + 32 * t:2 * C + 32 * t + n]
        if bqkv is not None:
            bt[12 + t, :n] = bqkv[2 * C + 32 * t:2 * C + 32 * t + n]
    hi, lo = split_bf16(wt.reshape(18, 32 * 192))
    return dict(w=torch.stack([hi, lo], dim=1).contiguous(), b=bt.reshape(-1).contiguous(), K=K)


def pack_token_linear_gated(w: T, b: Optional[T], gw1: T, gb1: Optional[T], gw2: T) -> dict:
    """Weights of ff_token_linear_gated: the projection [N, K] as in pack_token_linear plus one more tile with the spatial gate's
    first layer GW1 [hidden <= 32, K]; its bias and the second layer's weights zero padded to 32."""
    N, K = w.shape
    hd = gw1.shape[0]
    assert K <= 192 and hd <= 32 and gw1.shape[1] == K
    nt = (N + 31) // 32
    wp = torch.zeros((nt + 1) * 32, 192, device=w.device)
    wp[:N, :K] = w
    wp[nt * 32:nt * 32 + hd, :K] = gw1
    hi, lo = split_bf16(wp.reshape(nt + 1, 32 * 192))
    bp = torch.zeros(nt * 32, device=w.device)
    if b is not None:
        bp[:N] = b
    g1 = torch.zeros(32, device=w.device)
    if gb1 is not None:
        g1[:hd] = gb1
    g2 = torch.zeros(32, device=w.device)
    g2[:hd] = gw2.reshape(-1)
    return dict(nt=nt, K=K, N=N, w=torch.stack([hi, lo], dim=1).contiguous(), b=bp, gb1=g1, gw2=g2)


def small_ct(cout: int) -> int:
    return 1 if cout == 1 else (4 if cout <= 4 else 16)


def pack_conv3x3_small(wp: T, cin: int) -> T:
    """Weight image of ff_conv3x3_small from a packed 3x3 weight [Cout, 9*cin] (tap-major, pack_conv): fp32 [9][ceil4(cin)][CT]."""
    cout = wp.shape[0]
    ct, cp = small_ct(cout), (cin + 3) // 4 * 4
    w = torch.zeros(9, cp, ct, device=wp.device)
    w[:, :cin, :cout] = wp.reshape(cout, 9, cin).permute(1, 2, 0)
    return w.contiguous()
