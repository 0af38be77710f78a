"""Host-side weight preparation (prep.py) checked on CPU against its documented layouts and against the sizes the
C ABI reports -- no GPU needed (the library is only asked for sizes / argument validation)."""
import numpy as np
import pytest
import torch


def test_quad_bias_layout():
    from isr2_amd.prep import quad_bias
    heads, nk = 3, 24
    b = torch.arange(heads * nk * 256, dtype=torch.float32).reshape(heads, nk, 256)
    q = quad_bias(b)
    assert tuple(q.shape) == (heads, nk // 4, 256, 4) and q.is_contiguous()
    rng = np.random.default_rng(0)
    for _ in range(200):
        h, k, qi = int(rng.integers(heads)), int(rng.integers(nk)), int(rng.integers(256))
        assert q[h, k // 4, qi, k % 4] == b[h, k, qi]          # include/ff_kernels.h: biasT[head][key / 4][query][key % 4]


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("cout,cin", [(60, 180), (180, 60), (180, 180), (3, 64), (256, 64), (200, 36)])
def test_halo_weight_image_matches_abi_size_and_layout(cout, cin, nterms):
    from isr2_amd import lib
    from isr2_amd.prep import halo_bn, pack_conv, pack_conv3x3_halo
    w = torch.randn(cout, cin, 3, 3)
    bn = halo_bn(cout)
    assert bn in (32, 64, 128, 192) and -(-cout // bn) * bn <= min(-(-cout // b) * b for b in (64, 128, 192)) or bn == 32
    img = pack_conv3x3_halo(pack_conv(w), cin, bn, nterms)
    assert img.dtype == torch.bfloat16 and img.is_contiguous()
    assert img.numel() * 2 == lib.load().ff_conv3x3_halo_weight_bytes(cout, cin, bn, nterms)
    # spot-check records: [nblk][chunk][tap][half][bn rows x (wk hi | wk lo | 8 pad)] (nterms 1: no lo part), each padded to 1 KiB
    wk = 32 if bn == 192 else 64
    rw = (2 * wk if nterms == 3 else wk) + 8
    nh, nchunk = 64 // wk, -(-cin // 64)
    rec = img.reshape(-(-cout // bn), nchunk, 9, nh, -1)
    rng = np.random.default_rng(1)
    for _ in range(100):
        co, ci, tap = int(rng.integers(cout)), int(rng.integers(cin)), int(rng.integers(9))
        blk, row = divmod(co, bn)
        chunk, r = divmod(ci, 64)
        half, k = divmod(r, wk)
        rowv = rec[blk, chunk, tap, half, row * rw:(row + 1) * rw].float()
        val = w[co, ci, tap // 3, tap % 3]
        hi = val.to(torch.bfloat16).float()
        assert rowv[k] == hi and (nterms == 1 or rowv[wk + k] == (val - hi).to(torch.bfloat16).float())
    assert lib.load().ff_conv3x3_halo_weight_bytes(cout, cin, 100, nterms) == -1 and lib.load().ff_conv3x3_halo_weight_bytes(cout, cin, bn, 2) == -1


def test_token_linear_pack_shapes():
    from isr2_amd.prep import pack_token_linear
    for n, k, kpad in ((540, 180, 192), (11, 64, 64), (256, 128, 128)):
        pk = pack_token_linear(torch.randn(n, k), torch.randn(n))
        assert pk["kpad"] == kpad and pk["nt"] == -(-n // 32) and tuple(pk["w"].shape) == (pk["nt"], 2, 32 * kpad)
        assert pk["b"].numel() == pk["nt"] * 32 and torch.all(pk["b"][n:] == 0)
        w = pk["w"].reshape(pk["nt"], 2, 32, kpad).float()
        assert torch.all(w[:, :, :, k:] == 0)                                   # K padding is zero
        full = (w[:, 0] + w[:, 1]).reshape(-1, kpad)                            # hi + lo reconstructs to ~2^-16
        assert torch.all(full[n:] == 0)
