"""Micro-benchmark: window-resident attention block (ff_win_attn_fused) against the two-stage form it replaces
(ff_token_linear qkv + ff_window_attn_bf16s) on the bench tile's 256x256 token grid.  python tools/wf_bench.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isr2_amd import ops  # noqa: E402
from isr2_amd.prep import pack_win_attn, pack_win_rel, pack_token_linear  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    H = W = 256
    C, heads, d, ws = 180, 6, 30, 16
    g = torch.Generator().manual_seed(0)
    x = ops.empty_rows((1, H, W, C), dev)
    x.copy_(torch.randn(1, H, W, C, generator=g).to(dev))
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    wqkv = (torch.randn(3 * C, C, generator=g) / math.sqrt(C)).to(dev)
    bqkv = torch.zeros(3 * C, device=dev)
    table = (torch.randn(heads, (2 * ws - 1) ** 2, generator=g) * 0.5).to(dev)
    pk = pack_win_attn(wqkv, bqkv, heads, d, d ** -0.5)
    relp = pack_win_rel(table, ws, ws)
    out = ops.empty_rows((1, H, W, C), dev)
    for shift in (0, 8):
        t = timeit(lambda: ops.win_attn_fused(x, out, pk, relp, gamma=gam, beta=bet, H=H, W=W, Hp=H, Wp=W, win=(ws, ws),
                                              shift=(shift, shift), use_mask=shift > 0, want_xn=True))
        print(f"fused HAT block, shift {shift}: {t:.1f} us")
    tl = pack_token_linear(wqkv, bqkv)
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    yy, xx = ys.reshape(-1), xs.reshape(-1)
    rpi = ((yy[:, None] - yy[None, :] + ws - 1) * (2 * ws - 1) + (xx[:, None] - xx[None, :] + ws - 1)).reshape(-1).to(dev)
    bias = table.t()[rpi].reshape(256, 256, heads).permute(2, 1, 0).contiguous()
    t1 = timeit(lambda: ops.token_linear(x, tl, gamma=gam, beta=bet, want_xn=True))
    qkv, _ = ops.token_linear(x, tl, gamma=gam, beta=bet, want_xn=True)
    for shift in (0, 8):
        t2 = timeit(lambda: ops.window_attn(qkv, out, bias, q_off=0, k_off=C, v_off=2 * C, o_off=0, H=H, W=W, Hp=H, Wp=W, win=(ws, ws),
                                            kwin=(ws, ws), shift=(shift, shift), use_mask=shift > 0, heads=heads, d=d, scale=d ** -0.5))
        print(f"two-stage, shift {shift}: token_linear qkv {t1:.1f} us + window_attn {t2:.1f} us = {t1 + t2:.1f} us")
    # DAT branches (3 heads each, 8x32 / 32x8)
    for br, (wh, ww) in enumerate(((8, 32), (32, 8))):
        rel6 = (torch.randn(6, (2 * wh - 1) * (2 * ww - 1), generator=g) * 0.5).to(dev)
        rp = pack_win_rel(rel6, wh, ww)
        vout = torch.empty(1, H, W, 544, device=dev)[..., :540]
        t = timeit(lambda: ops.win_attn_fused(x, out, pk, rp, gamma=gam, beta=bet, H=H, W=W, Hp=H, Wp=W, win=(wh, ww), shift=(0, 0),
                                              use_mask=False, head0=3 * br, nheads=3, zero_pad=True, v_out=vout, v_off=360))
        print(f"fused DAT branch {wh}x{ww}: {t:.1f} us")


if __name__ == "__main__":
    main()
