"""GPU parity of every C-ABI kernel against plain PyTorch fp32 ops (run with -m gpu on the MI355X box).
Tolerances: fp32 MFMA is an exact-fp32 fma chain, so contractions agree with torch to ~1e-5 relative of
the accumulated magnitude; pointwise ops to a few ulp."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from isr2_amd import lib
    lib.load()
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    return torch.device("cuda:0")


@pytest.fixture(params=["f32", "bf16x3"])
def gemm_mode(request, dev):
    """Contraction precision of ff_conv2d: exact fp32 MFMA, or the split-operand bf16 MFMA (default)."""
    from isr2_amd import ops
    old = ops.gemm_mode()
    ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode(old)


GEMM_TOL = {"f32": 2e-5, "bf16x3": 6e-5, "bf16": 2e-2}   # relative to max|ref|; bf16x3 drops the O(2^-16) lo*lo term, bf16 rounds operands to 8 bits


def rnd(*shape, dev, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def close(a, b, tol, name=""):
    err = (a - b).abs().max().item()
    ref = max(1.0, b.abs().max().item())
    assert err <= tol * ref, f"{name}: max|d|={err:.3e} ref={ref:.3e}"


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p,act", [
    (1, 16, 16, 180, 180, 3, 1, 1, None),      # RHAG conv
    (1, 19, 23, 180, 60, 3, 1, 1, "gelu"),     # CAB conv1, ragged M
    (1, 16, 16, 60, 180, 3, 1, 1, None),
    (2, 12, 20, 3, 64, 3, 1, 1, "relu"),       # scalar path (Cin=3)
    (1, 16, 16, 73, 64, 3, 1, 1, "gelu"),      # scalar path (Cin=73)
    (1, 32, 32, 64, 128, 2, 2, 0, None),       # NAFNet down
    (1, 16, 16, 64, 3, 3, 1, 1, "sigmoid"),    # tiny Cout
    (1, 8, 8, 1024, 2048, 1, 1, 0, None),      # 128x128 tile config
    (1, 24, 24, 32, 1, 3, 1, 1, "sigmoid"),
])
def test_conv2d_matches_torch(dev, gemm_mode, B, H, W, Cin, Cout, k, s, p, act):
    from isr2_amd import ops
    from isr2_amd.prep import pack_conv
    x = rnd(B, Cin, H, W, dev=dev, seed=1)
    w = rnd(Cout, Cin, k, k, dev=dev, seed=2, scale=1.0 / math.sqrt(Cin * k * k))
    b = rnd(Cout, dev=dev, seed=3, scale=0.1)
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    ref = {"gelu": F.gelu, "relu": F.relu, "sigmoid": torch.sigmoid, None: lambda t: t}[act](ref)
    out = ops.conv2d(x.permute(0, 2, 3, 1).contiguous(), pack_conv(w), b, ksize=(k, k), stride=(s, s), pad=(p, p), act=act)
    close(out.permute(0, 3, 1, 2), ref, GEMM_TOL[gemm_mode], "conv2d")


def test_conv2d_epilogue_residual_mul_alpha_and_slices(dev, gemm_mode):
    from isr2_amd import ops
    from isr2_amd.prep import pack_conv
    B, H, W, Cin, Cout = 1, 20, 20, 64, 64
    wide = rnd(B, H, W, 100, dev=dev, seed=4)                 # input is a channel slice [10:74] of a wider buffer
    x = wide[..., 10:74]
    w = rnd(Cout, Cin, 1, 1, dev=dev, seed=5, scale=0.1)
    b = rnd(Cout, dev=dev, seed=6, scale=0.1)
    mul = rnd(Cout, dev=dev, seed=7)
    res = rnd(B, H, W, Cout, dev=dev, seed=8)
    obuf = torch.zeros(B, H, W, 96, device=dev)
    out = ops.conv2d(x, pack_conv(w), b, act="lrelu", res=res, mul=mul, alpha=0.3, out=obuf[..., 16:80])
    ref = res + 0.3 * mul * F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2), w, b), 0.01).permute(0, 2, 3, 1)
    close(out, ref, GEMM_TOL[gemm_mode], "epilogue")
    assert obuf[..., :16].abs().max() == 0 and obuf[..., 80:].abs().max() == 0


@pytest.mark.parametrize("B,H,W,Cin,Cout,act", [
    (1, 48, 48, 180, 60, "gelu"),      # CAB conv1: 3 channel chunks, bn 64 (16x16-pixel tiles)
    (2, 37, 45, 128, 180, None),       # ragged tiles, batch 2, bn 192 (8x16 tiles)
    (1, 40, 33, 180, 180, "lrelu"),    # RHAG conv
    (1, 64, 64, 64, 3, "sigmoid"),     # bn 32
    (1, 35, 64, 128, 128, None),       # bn 128
    (1, 32, 32, 128, 256, "relu"),     # two 128-wide n-blocks
    (1, 32, 40, 36, 40, None),         # Cin < 64 (zero-padded chunk), ragged n-block
    (1, 32, 40, 132, 200, None),       # ragged last chunk, two ragged n-blocks
])
def test_conv3x3_halo_matches_torch(dev, B, H, W, Cin, Cout, act):
    """LDS-resident 3x3 kernel (bf16x3 only) against PyTorch fp32, with residual / per-channel scale / strided views."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_conv
    prev = ops.gemm_mode()
    ops.set_gemm_mode("bf16x3")
    try:
        wide = rnd(B, H, W, Cin + 8, dev=dev, seed=60)
        x = wide[..., 4:4 + Cin]
        w = rnd(Cout, Cin, 3, 3, dev=dev, seed=61, scale=1.0 / math.sqrt(9 * Cin))
        b = rnd(Cout, dev=dev, seed=62, scale=0.1)
        mul = rnd(Cout, dev=dev, seed=63)
        res = rnd(B, H, W, Cout, dev=dev, seed=64)
        wp = pack_conv(w)
        out = ops.conv2d(x, wp, b, ksize=(3, 3), pad=(1, 1), act=act, res=res, mul=mul, alpha=0.7)
        assert ops.PREPARED.peek(wp, "halo") is not None, "halo kernel was not selected"
        f = {"gelu": F.gelu, "relu": F.relu, "sigmoid": torch.sigmoid, "lrelu": lambda t: F.leaky_relu(t, 0.01), None: lambda t: t}[act]
        ref = res + 0.7 * mul * f(F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1)).permute(0, 2, 3, 1)
        close(out, ref, GEMM_TOL["bf16x3"], "conv3x3 halo")
        if B == 1:                       # global average pool of the stored output from the same launch (epilogue partials)
            out3, pooled = ops.conv2d(x, wp, b, ksize=(3, 3), pad=(1, 1), act=act, res=res, mul=mul, alpha=0.7, want_pool=True)
            assert torch.equal(out3, out)
            close(pooled, out.mean(dim=(1, 2)), 2e-5, "pooled conv output")
        ops.set_halo(False)
        out2 = ops.conv2d(x, pack_conv(w), b, ksize=(3, 3), pad=(1, 1), act=act, res=res, mul=mul, alpha=0.7)
        close(out, out2, GEMM_TOL["bf16x3"], "halo vs implicit GEMM")
    finally:
        ops.set_halo(True)
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("hw,cin", [((12, 14), 64), ((36, 40), 128)])   # implicit GEMM; LDS-resident 3x3 kernel (bf16x3)
def test_conv2d_pixel_shuffle(dev, gemm_mode, hw, cin):
    from isr2_amd import ops
    from isr2_amd.prep import pack_conv
    x = rnd(1, cin, hw[0], hw[1], dev=dev, seed=9)
    w = rnd(256, cin, 3, 3, dev=dev, seed=10, scale=0.05)
    b = rnd(256, dev=dev, seed=11, scale=0.1)
    ref = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2)
    out = ops.conv2d(x.permute(0, 2, 3, 1).contiguous(), pack_conv(w), b, ksize=(3, 3), pad=(1, 1), shuffle=2)
    close(out.permute(0, 3, 1, 2), ref, GEMM_TOL[gemm_mode], "shuffle")


def test_linear_matches_torch(dev, gemm_mode):
    from isr2_amd import ops
    x = rnd(1000, 180, dev=dev, seed=12)
    w = rnd(540, 180, dev=dev, seed=13, scale=0.07)
    b = rnd(540, dev=dev, seed=14, scale=0.1)
    close(ops.linear(x, w, b), F.linear(x, w, b), GEMM_TOL[gemm_mode], "linear")
    close(ops.linear(x, w, b, act="gelu"), F.gelu(F.linear(x, w, b)), GEMM_TOL[gemm_mode], "linear+gelu")
    close(ops.linear(x, w.clone(), b, dynamic_w=True), F.linear(x, w, b), GEMM_TOL[gemm_mode], "linear dynamic weight")


@pytest.mark.parametrize("hint", [1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("M,K,N,k", [(4096, 1024, 2048, 1), (1111, 192, 200, 1), (900, 180, 96, 1), (30 * 30, 64, 130, 3), (24 * 24, 128, 64, 2)])
def test_gemm_tile_forms_agree_with_torch(dev, hint, M, K, N, k):
    """Every tile form of the bf16x3 implicit GEMM (4- and 8-wave tiles, 32- and 64-deep chunks; hints 8 / 9 fall back to the 32-deep
    form when K does not pad to 64) on ragged and exact shapes, 1x1 / 3x3 / 2x2-stride-2, with residual and per-channel scale."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_conv
    prev = ops.gemm_mode()
    ops.set_gemm_mode("bf16x3")
    ops.set_halo(False)
    try:
        hw = int(round(math.sqrt(M))) if k > 1 else None
        B, H, W = (1, hw, hw) if k > 1 else (1, 1, M)
        x = rnd(B, K, H, W, dev=dev, seed=120, scale=1.0)
        w = rnd(N, K, k, k, dev=dev, seed=121, scale=1.0 / math.sqrt(K * k * k))
        b = rnd(N, dev=dev, seed=122, scale=0.1)
        mul = rnd(N, dev=dev, seed=123)
        st, pd = (2, 0) if k == 2 else (1, k // 2)
        ref = F.conv2d(x, w, b, stride=st, padding=pd)
        res = rnd(*ref.permute(0, 2, 3, 1).shape, dev=dev, seed=124)
        out = ops.conv2d(x.permute(0, 2, 3, 1).contiguous(), pack_conv(w), b, ksize=(k, k), stride=(st, st), pad=(pd, pd), res=res, mul=mul,
                         tile_hint=hint)
        close(out, res + (mul.view(1, -1, 1, 1) * ref).permute(0, 2, 3, 1), GEMM_TOL["bf16x3"], f"tile form {hint}")
    finally:
        ops.set_halo(True)
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("C,H,W", [(64, 64, 96), (128, 40, 48), (64, 37, 53), (128, 9, 17)])
def test_naf_front_fused(dev, C, H, W):
    """LayerNorm2d + conv1 + depth-wise 3x3 + SimpleGate + average pool in one launch against the PyTorch fp32 chain (ragged tiles:
    the zero padding of the depth-wise conv applies to conv1's OUTPUT, nafnet_arch.py:88-98)."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_token_linear, pack_dw
    prev = ops.gemm_mode()
    ops.set_gemm_mode("bf16x3")
    try:
        x = rnd(1, C, H, W, dev=dev, seed=140, scale=1.5) + 0.2
        g, b = rnd(C, dev=dev, seed=141) * 0.1 + 1, rnd(C, dev=dev, seed=142) * 0.1
        w1, b1 = rnd(2 * C, C, dev=dev, seed=143, scale=1.0 / math.sqrt(C)), rnd(2 * C, dev=dev, seed=144, scale=0.1)
        w2, b2 = rnd(2 * C, 1, 3, 3, dev=dev, seed=145, scale=0.3), rnd(2 * C, dev=dev, seed=146, scale=0.1)
        mu = x.mean(1, keepdim=True)
        var = (x - mu).pow(2).mean(1, keepdim=True)
        xn = (x - mu) / (var + 1e-6).sqrt() * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
        t = F.conv2d(F.conv2d(xn, w1.view(2 * C, C, 1, 1), b1), w2, b2, padding=1, groups=2 * C)
        ref = t[:, :C] * t[:, C:]
        xh = x.permute(0, 2, 3, 1).contiguous()
        got, pooled = ops.naf_front(xh, pack_token_linear(w1, b1), g, b, pack_dw(w2), b2)
        close(got.permute(0, 3, 1, 2), ref, 2 * GEMM_TOL["bf16x3"], "naf front")
        close(pooled, ref.mean(dim=(2, 3)), 2 * GEMM_TOL["bf16x3"], "naf front pool")
        # and against the two-launch path it replaces
        t2 = ops.token_linear(xh, pack_token_linear(w1, b1), gamma=g, beta=b, eps=1e-6)
        g2, p2 = ops.dwconv3_gate_pool(t2, pack_dw(w2), b2)
        close(got, g2, 1e-5, "naf front vs two launches")
        close(pooled, p2, 1e-5, "naf front pool vs two launches")
    finally:
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("M,K,N", [(4096, 1024, 2048), (1000, 256, 512), (333, 64, 128)])
def test_linear_input_scale_and_simple_gate_epilogue(dev, M, K, N):
    """The two NAFBlock fusions of the split-bf16 GEMM (nafnet_arch.py:88-106): conv3(x * sca) as a scale on the A operand, and
    conv4 + SimpleGate with the chunk(2) halves interleaved so that the product happens in the epilogue."""
    from isr2_amd import ops
    prev = ops.gemm_mode()
    ops.set_gemm_mode("bf16x3")
    try:
        x = rnd(M, K, dev=dev, seed=130)
        w = rnd(N, K, dev=dev, seed=131, scale=1.0 / math.sqrt(K))
        b = rnd(N, dev=dev, seed=132, scale=0.1)
        km = rnd(K, dev=dev, seed=133) * 0.5 + 1.0
        res = rnd(M, N, dev=dev, seed=134)
        mul = rnd(N, dev=dev, seed=135)
        close(ops.linear(x, w, b, res=res, mul=mul, kmul=km), res + mul * F.linear(x * km, w, b), GEMM_TOL["bf16x3"], "kmul")
        c = N // 2
        wi = torch.stack((w[:c], w[c:]), dim=1).reshape(N, K).contiguous()
        bi = torch.stack((b[:c], b[c:]), dim=1).reshape(-1).contiguous()
        y = F.linear(x, w, b)
        got = ops.linear(x, wi, bi, gate_pairs=True)
        assert got.shape == (M, c)
        close(got, y[:, :c] * y[:, c:], 2 * GEMM_TOL["bf16x3"], "simple gate epilogue")
        wide = torch.zeros(M, c + 12, device=dev)                       # into a channel slice of a wider buffer
        ops.linear(x, wi, bi, gate_pairs=True, out=wide[:, 4:4 + c])
        assert torch.equal(wide[:, 4:4 + c], got) and wide[:, :4].abs().max() == 0 and wide[:, 4 + c:].abs().max() == 0
    finally:
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("M", [65536, 1000, 77])
def test_token_mlp_fused(dev, M):
    """LN + fc1 + GELU + fc2 + residual in one launch (bf16x3) against the PyTorch fp32 chain."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_token_mlp
    C, Hd = 180, 360
    x = rnd(M, C, dev=dev, seed=90, scale=1.5) + 0.3
    g, b = rnd(C, dev=dev, seed=91) * 0.1 + 1, rnd(C, dev=dev, seed=92) * 0.1
    w1, b1 = rnd(Hd, C, dev=dev, seed=93, scale=1.0 / math.sqrt(C)), rnd(Hd, dev=dev, seed=94, scale=0.1)
    w2, b2 = rnd(C, Hd, dev=dev, seed=95, scale=1.0 / math.sqrt(Hd)), rnd(C, dev=dev, seed=96, scale=0.1)
    ref = x + F.linear(F.gelu(F.linear(F.layer_norm(x, (C,), g, b, 1e-5), w1, b1)), w2, b2)
    out = ops.token_mlp(x, g, b, pack_token_mlp(w1, b1, w2, b2))
    close(out, ref, 6e-5, "token_mlp")


@pytest.mark.parametrize("M,N,ln,act,nres", [(65536, 540, True, None, 0), (1000, 180, False, None, 2), (777, 720, True, "gelu", 0),
                                            (300, 11, False, "gelu", 0), (256, 180, False, None, 1), (500, 182, False, None, 2)])
def test_token_linear(dev, M, N, ln, act, nres):
    """Token-stationary linear (+LayerNorm prologue, +GELU, +one or two residuals) against the PyTorch fp32 chain."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_token_linear
    C = 180
    x = rnd(M, C, dev=dev, seed=100, scale=1.5) + 0.3
    g, b = rnd(C, dev=dev, seed=101) * 0.1 + 1, rnd(C, dev=dev, seed=102) * 0.1
    w, bias = rnd(N, C, dev=dev, seed=103, scale=1.0 / math.sqrt(C)), rnd(N, dev=dev, seed=104, scale=0.1)
    res, res2, rs2 = rnd(M, N, dev=dev, seed=105), rnd(M, N, dev=dev, seed=106), rnd(N, dev=dev, seed=107)
    xin = F.layer_norm(x, (C,), g, b, 1e-5) if ln else x
    ref = F.linear(xin, w, bias)
    if act == "gelu":
        ref = F.gelu(ref)
    if nres >= 1:
        ref = ref + res
    if nres == 2:
        ref = ref + res2 * rs2
    out = ops.token_linear(x, pack_token_linear(w, bias), gamma=g if ln else None, beta=b if ln else None, act=act,
                           res=res if nres >= 1 else None, res2=res2 if nres == 2 else None, res2_scale=rs2 if nres == 2 else None)
    close(out, ref, 6e-5, "token_linear")
    if ln:                               # LayerNorm'ed rows as a side output of the same launch
        out2, xn = ops.token_linear(x, pack_token_linear(w, bias), gamma=g, beta=b, act=act, res=res if nres >= 1 else None,
                                    res2=res2 if nres == 2 else None, res2_scale=rs2 if nres == 2 else None, want_xn=True)
        assert torch.equal(out, out2)
        close(xn, xin, 1e-5, "token_linear xn side output")


@pytest.mark.parametrize("C,H,W", [(64, 40, 56), (128, 33, 20), (512, 16, 16), (1024, 8, 8)])
def test_naf_dwconv_gate_pool(dev, C, H, W):
    from isr2_amd import ops
    from isr2_amd.prep import pack_dw
    t = rnd(1, 2 * C, H, W, dev=dev, seed=110)
    w, b = rnd(2 * C, 1, 3, 3, dev=dev, seed=111, scale=0.3), rnd(2 * C, dev=dev, seed=112, scale=0.1)
    d = F.conv2d(t, w, b, padding=1, groups=2 * C)
    ref = d[:, :C] * d[:, C:]
    g, pooled = ops.dwconv3_gate_pool(t.permute(0, 2, 3, 1).contiguous(), pack_dw(w), b)
    close(g.permute(0, 3, 1, 2), ref, 1e-5, "dwconv gate")
    close(pooled, ref.mean(dim=(2, 3)), 1e-5, "pooled")


@pytest.mark.parametrize("C,M", [(64, 65536), (64, 300), (128, 4096), (128, 77)])
def test_naf_ffn_fused(dev, C, M):
    """y + gamma * conv5(SimpleGate(conv4(LayerNorm2d(y)))) in one launch against the PyTorch fp32 chain."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_naf_ffn
    y = rnd(M, C, dev=dev, seed=120, scale=1.3) + 0.2
    g, b = rnd(C, dev=dev, seed=121) * 0.1 + 1, rnd(C, dev=dev, seed=122) * 0.1
    w4, b4 = rnd(2 * C, C, dev=dev, seed=123, scale=1.0 / math.sqrt(C)), rnd(2 * C, dev=dev, seed=124, scale=0.1)
    w5, b5 = rnd(C, C, dev=dev, seed=125, scale=1.0 / math.sqrt(C)), rnd(C, dev=dev, seed=126, scale=0.1)
    gam = rnd(C, dev=dev, seed=127, scale=0.3)
    t = F.linear(F.layer_norm(y, (C,), g, b, 1e-6), w4, b4)
    ref = y + gam * F.linear(t[:, :C] * t[:, C:], w5, b5)
    out = ops.naf_ffn(y, pack_naf_ffn(w4, b4, w5, b5), g, b, gam)
    close(out, ref, 6e-5, "naf_ffn")


@pytest.mark.parametrize("K,N", [(64, 128), (128, 256)])
def test_token_linear_small_k(dev, K, N):
    from isr2_amd import ops
    from isr2_amd.prep import pack_token_linear
    M = 5000
    x = rnd(M, K, dev=dev, seed=130) + 0.1
    g, b = rnd(K, dev=dev, seed=131) * 0.1 + 1, rnd(K, dev=dev, seed=132) * 0.1
    w, bias = rnd(N, K, dev=dev, seed=133, scale=1.0 / math.sqrt(K)), rnd(N, dev=dev, seed=134, scale=0.1)
    ref = F.linear(F.layer_norm(x, (K,), g, b, 1e-6), w, bias)
    close(ops.token_linear(x, pack_token_linear(w, bias), gamma=g, beta=b, eps=1e-6), ref, 6e-5, "token_linear small K")


@pytest.mark.parametrize("C,eps", [(180, 1e-5), (64, 1e-6), (360, 1e-5), (1024, 1e-6), (128, 1e-6)])
def test_layernorm(dev, C, eps):
    from isr2_amd import ops
    x = rnd(777, C, dev=dev, seed=15, scale=2.0) + 0.5
    g = rnd(C, dev=dev, seed=16) * 0.1 + 1
    b = rnd(C, dev=dev, seed=17) * 0.1
    close(ops.layernorm(x, g, b, eps), F.layer_norm(x, (C,), g, b, eps), 1e-5, "ln")


def test_layernorm_slice(dev):
    from isr2_amd import ops
    y = rnd(500, 720, dev=dev, seed=18)
    g = rnd(360, dev=dev, seed=19) * 0.1 + 1
    b = rnd(360, dev=dev, seed=20) * 0.1
    close(ops.layernorm(y[:, 360:], g, b), F.layer_norm(y[:, 360:], (360,), g, b), 1e-5, "ln slice")


def _hat_bias(table, ws, ows, heads):
    from oracle import freqfusion_oracle as O
    rpi = O.hat_rel_index_sa(ws) if ows == ws else O.hat_rel_index_oca(ws, ows)
    return table[rpi.reshape(-1).to(table.device)].reshape(ws * ws, ows * ows, heads).permute(2, 0, 1).contiguous()


@pytest.mark.parametrize("H,W,shift", [(32, 48, 0), (32, 48, 8), (16, 16, 8)])
def test_window_attn_hat(dev, gemm_mode, H, W, shift):
    """W-MSA / SW-MSA against the oracle's window attention math evaluated with torch on the GPU."""
    from isr2_amd import ops
    from oracle import freqfusion_oracle as O
    heads, d, ws, C = 6, 30, 16, 180
    qkv = rnd(1, H, W, 3 * C, dev=dev, seed=21)
    table = rnd((2 * ws - 1) ** 2, heads, dev=dev, seed=22, scale=0.5)
    bias = _hat_bias(table, ws, ws, heads)                          # (heads, nq, nk)
    out = torch.zeros(1, H, W, C, device=dev)
    ops.window_attn(qkv, out, bias.transpose(1, 2).contiguous(), q_off=0, k_off=C, v_off=2 * C, o_off=0, H=H, W=W, Hp=H, Wp=W,
                    win=(ws, ws), kwin=(ws, ws), shift=(shift, shift), use_mask=shift > 0, heads=heads, d=d, scale=d ** -0.5)
    xi = qkv
    if shift:
        xi = torch.roll(xi, shifts=(-shift, -shift), dims=(1, 2))
    xw = O._win_split(xi, ws, ws).reshape(-1, ws * ws, 3, heads, d).permute(2, 0, 3, 1, 4)
    mask = O._region_mask(H, W, ws, ws, ws // 2, ws // 2).to(dev) if shift else None
    o = O._softmax_attn(xw[0] * d ** -0.5, xw[1], xw[2], bias, mask).transpose(1, 2).reshape(-1, ws * ws, C)
    o = O._win_merge(o, ws, ws, H, W)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    close(out, o, GEMM_TOL[gemm_mode], "window_attn")
    if gemm_mode != "f32":          # same bias values gathered from the compact (2ws-1)^2 table in LDS: bit-identical
        out2 = torch.zeros(1, H, W, C, device=dev)
        ops.window_attn(qkv, out2, bias.transpose(1, 2).contiguous(), q_off=0, k_off=C, v_off=2 * C, o_off=0, H=H, W=W, Hp=H, Wp=W,
                        win=(ws, ws), kwin=(ws, ws), shift=(shift, shift), use_mask=shift > 0, heads=heads, d=d, scale=d ** -0.5,
                        rel_table=table.t().contiguous())
        assert torch.equal(out, out2), "relative-position gather differs from the expanded table"


def test_window_attn_ocab(dev, gemm_mode):
    from isr2_amd import ops
    from oracle import freqfusion_oracle as O
    heads, d, ws, ows, C, H, W = 6, 30, 16, 24, 180, 32, 32
    qkv = rnd(1, H, W, 3 * C, dev=dev, seed=23)
    table = rnd((ws + ows - 1) ** 2, heads, dev=dev, seed=24, scale=0.5)
    bias = _hat_bias(table, ws, ows, heads)
    out = torch.zeros(1, H, W, C, device=dev)
    ops.window_attn(qkv, out, bias.transpose(1, 2).contiguous(), q_off=0, k_off=C, v_off=2 * C, o_off=0, H=H, W=W, Hp=H, Wp=W,
                    win=(ws, ws), kwin=(ows, ows), shift=(0, 0), use_mask=False, heads=heads, d=d, scale=d ** -0.5)
    q = O._win_split(qkv[..., :C], ws, ws)
    kv = qkv[..., C:].permute(0, 3, 1, 2)
    nwin = (H // ws) * (W // ws)
    kvw = F.unfold(kv, kernel_size=ows, stride=ws, padding=(ows - ws) // 2)
    kvw = kvw.reshape(1, 2, C, ows * ows, nwin).permute(1, 0, 4, 3, 2).reshape(2, nwin, ows * ows, C)
    qh = q.reshape(-1, ws * ws, heads, d).transpose(1, 2) * d ** -0.5
    kh = kvw[0].reshape(-1, ows * ows, heads, d).transpose(1, 2)
    vh = kvw[1].reshape(-1, ows * ows, heads, d).transpose(1, 2)
    o = O._softmax_attn(qh, kh, vh, bias, None).transpose(1, 2).reshape(-1, ws * ws, C)
    close(out, O._win_merge(o, ws, ws, H, W), GEMM_TOL[gemm_mode], "ocab")
    if gemm_mode != "f32":          # the same bias gathered in the kernel from the rotated (ws+ows-1)^2 table: bit-identical
        from isr2_amd.prep import pack_rel_overlap
        out2 = torch.zeros(1, H, W, C, device=dev)
        ops.window_attn(qkv, out2, bias.transpose(1, 2).contiguous(), q_off=0, k_off=C, v_off=2 * C, o_off=0, H=H, W=W, Hp=H, Wp=W,
                        win=(ws, ws), kwin=(ows, ows), shift=(0, 0), use_mask=False, heads=heads, d=d, scale=d ** -0.5,
                        rel_table=pack_rel_overlap(table, ws, ows))
        assert torch.equal(out, out2), "overlapping-window relative-position gather differs from the expanded table"


@pytest.mark.parametrize("H,W,shifted", [(32, 64, False), (48, 48, True), (64, 32, True)])
def test_window_attn_dat_branches(dev, gemm_mode, H, W, shifted):
    """DAT 8x32 / 32x8 branches on channel halves, zero tokens beyond (H, W), optional shift + mask."""
    from isr2_amd import ops
    from oracle import freqfusion_oracle as O
    C, half, hh, d = 180, 90, 3, 30
    m = 32
    Hp, Wp = H + (m - H % m) % m, W + (m - W % m) % m
    qkv = rnd(1, H, W, 3 * C, dev=dev, seed=25)
    out = torch.zeros(1, H, W, C, device=dev)
    qkv5 = F.pad(qkv.reshape(1, H, W, 3, C), (0, 0, 0, 0, 0, Wp - W, 0, Hp - H))
    for br in range(2):
        wh, ww = (8, 32) if br == 0 else (32, 8)
        sh, sw = wh // 2, ww // 2
        rel = rnd(hh, (2 * wh - 1) * (2 * ww - 1), dev=dev, seed=26 + br, scale=0.5)          # DynamicPosBias output per offset
        cy, cx = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
        cy, cx = cy.reshape(-1).to(dev), cx.reshape(-1).to(dev)
        idx = (cy[:, None] - cy[None, :] + wh - 1) * (2 * ww - 1) + (cx[:, None] - cx[None, :] + ww - 1)   # dat_arch.py:300-318
        bias = rel[:, idx]                                                                     # (heads, nq, nk)
        ops.window_attn(qkv, out, bias.transpose(1, 2).contiguous(), q_off=br * half, k_off=C + br * half,
                        v_off=2 * C + br * half, o_off=br * half, H=H, W=W, Hp=Hp, Wp=Wp, win=(wh, ww), kwin=(wh, ww),
                        shift=(sh, sw) if shifted else (0, 0), use_mask=shifted, heads=hh, d=d, scale=d ** -0.5)
        if gemm_mode != "f32":
            out2 = torch.zeros_like(out)
            ops.window_attn(qkv, out2, bias.transpose(1, 2).contiguous(), q_off=br * half, k_off=C + br * half,
                            v_off=2 * C + br * half, o_off=br * half, H=H, W=W, Hp=Hp, Wp=Wp, win=(wh, ww), kwin=(wh, ww),
                            shift=(sh, sw) if shifted else (0, 0), use_mask=shifted, heads=hh, d=d, scale=d ** -0.5, rel_table=rel)
            assert torch.equal(out[..., br * half:(br + 1) * half], out2[..., br * half:(br + 1) * half]), "rel gather != expanded"
        t = qkv5[..., br * half:(br + 1) * half]
        if shifted:
            t = torch.roll(t, shifts=(-sh, -sw), dims=(1, 2))
        tw = [O._win_split(t[:, :, :, i], wh, ww).reshape(-1, wh * ww, hh, d).transpose(1, 2) for i in range(3)]
        mask = O._region_mask(Hp, Wp, wh, ww, sh, sw).to(dev) if shifted else None
        o = O._softmax_attn(tw[0] * d ** -0.5, tw[1], tw[2], bias, mask).transpose(1, 2).reshape(-1, wh * ww, half)
        o = O._win_merge(o, wh, ww, Hp, Wp)
        if shifted:
            o = torch.roll(o, shifts=(sh, sw), dims=(1, 2))
        close(out[..., br * half:(br + 1) * half], o[:, :H, :W], GEMM_TOL[gemm_mode], f"dat branch {br}")


@pytest.mark.parametrize("P,C,Hd", [(65536, 180, 11), (1000, 64, 16), (77, 192, 3), (1048576, 32, 8), (5000, 64, 16)])
def test_pixel_mlp(dev, P, C, Hd):
    """DAT spatial-interaction gate (dat_arch.py:585-590) as one per-pixel kernel against the PyTorch chain."""
    from isr2_amd import ops
    wide = rnd(P, C + 12, dev=dev, seed=150)
    x = wide[:, 4:4 + C]
    W1, b1 = rnd(Hd, C, dev=dev, seed=151, scale=1.0 / math.sqrt(C)), rnd(Hd, dev=dev, seed=152, scale=0.1)
    w2, b2 = rnd(1, Hd, dev=dev, seed=153, scale=0.5), 0.37
    ref = torch.sigmoid(F.linear(F.gelu(F.linear(x, W1, b1)), w2) + b2)
    close(ops.pixel_mlp(x, W1, b1, "gelu", w2, b2, "sigmoid"), ref, 1e-5, "pixel_mlp")


def test_pool_and_vec_mlp(dev):
    from isr2_amd import ops
    x = rnd(2, 37, 41, 180, dev=dev, seed=30)
    pm = ops.pool_mean(x)
    close(pm, x.mean(dim=(1, 2)), 1e-5, "pool")
    W1, b1 = rnd(6, 180, dev=dev, seed=31, scale=0.1), rnd(6, dev=dev, seed=32, scale=0.1)
    W2, b2 = rnd(180, 6, dev=dev, seed=33, scale=0.3), rnd(180, dev=dev, seed=34, scale=0.1)
    out = ops.vec_mlp(pm, W1, b1, "relu", W2, b2, "sigmoid", post=0.01)
    ref = torch.sigmoid(F.linear(F.relu(F.linear(pm, W1, b1)), W2, b2)) * 0.01
    close(out, ref, 1e-5, "vec_mlp2")
    Wb, bb = rnd(1024, 1024, dev=dev, seed=35, scale=0.03), rnd(1024, dev=dev, seed=36, scale=0.1)
    v = rnd(1, 1024, dev=dev, seed=37)
    close(ops.vec_mlp(v, Wb, bb, None), F.linear(v, Wb, bb), 1e-5, "vec_mlp1")


@pytest.mark.parametrize("C,kh,kw,sy,sx", [(180, 3, 3, 1, 1), (64, 5, 5, 1, 1), (64, 1, 21, 1, 1), (64, 21, 1, 1, 1),
                                           (3, 5, 5, 1, 1), (576, 3, 3, 1, 1)])
def test_dwconv(dev, C, kh, kw, sy, sx):
    from isr2_amd import ops
    from isr2_amd.prep import pack_dw
    x = rnd(1, C, 33, 29, dev=dev, seed=40)
    w = rnd(C, 1, kh, kw, dev=dev, seed=41, scale=0.3)
    b = rnd(C, dev=dev, seed=42, scale=0.1)
    ps, pt = rnd(C, dev=dev, seed=43) * 0.1 + 1, rnd(C, dev=dev, seed=44) * 0.1
    ref = F.gelu(F.conv2d(x, w, b, stride=(sy, sx), padding=(kh // 2, kw // 2), groups=C) * ps[None, :, None, None] + pt[None, :, None, None])
    out = ops.dwconv2d(x.permute(0, 2, 3, 1).contiguous(), pack_dw(w), b, ksize=(kh, kw), stride=(sy, sx), pad=(kh // 2, kw // 2),
                       post_scale=ps, post_shift=pt, act="gelu")
    close(out.permute(0, 3, 1, 2), ref, 1e-5, "dwconv")
    if (kh, kw) == (3, 3):                                  # fused SGFN gate product (dat_arch.py:123), strided mul_in slice
        wide = rnd(1, 33, 29, 2 * C, dev=dev, seed=45)
        out = ops.dwconv2d(x.permute(0, 2, 3, 1).contiguous(), pack_dw(w), b, mul_in=wide[..., :C])
        ref = F.conv2d(x, w, b, padding=1, groups=C) * wide[..., :C].permute(0, 3, 1, 2)
        close(out.permute(0, 3, 1, 2), ref, 1e-5, "dwconv*mul")


def test_pointwise(dev):
    from isr2_amd import ops
    a, b = rnd(300, 64, dev=dev, seed=50), rnd(300, 64, dev=dev, seed=51)
    ca, cb = rnd(64, dev=dev, seed=52), rnd(64, dev=dev, seed=53)
    pa, pb = rnd(300, 1, dev=dev, seed=54), rnd(300, 1, dev=dev, seed=55)
    close(ops.mix2(a, b, ka=0.7, kb=0.3, ca=ca, cb=cb, pa=pa, pb=pb), 0.7 * a * ca * pa + 0.3 * b * cb * pb, 1e-6, "mix2")
    close(ops.mix2(a, clamp01=True), a.clamp(0, 1), 0, "clamp")
    y = rnd(300, 128, dev=dev, seed=56)
    close(ops.fma3(None, y[:, :64], y[:, 64:]), y[:, :64] * y[:, 64:], 1e-6, "gate")
    close(ops.fma3(a, b, y[:, 64:], 0.1), a + 0.1 * b * y[:, 64:], 1e-6, "fma3")
    close(ops.affine(a, ca, cb, "gelu"), F.gelu(a * ca + cb), 1e-6, "affine")


def test_layout_roundtrip_and_reflect(dev):
    from isr2_amd import ops
    x = rnd(1, 3, 42, 52, dev=dev, seed=60)
    add = torch.tensor([-0.4488, -0.4371, -0.4040], device=dev)
    y = ops.nchw_to_nhwc(x, 48, 64, add=add, pad_mode="reflect")
    ref = F.pad(x + add[None, :, None, None], (0, 12, 0, 6), mode="reflect").permute(0, 2, 3, 1)
    close(y, ref, 1e-7, "reflect")
    z = ops.nchw_to_nhwc(x, 48, 64)
    close(z, F.pad(x, (0, 12, 0, 6)).permute(0, 2, 3, 1), 0, "zero pad")
    back = ops.nhwc_to_nchw(y, 42, 52, add=-add, clamp01=True)
    close(back, x.clamp(0, 1), 1e-6, "roundtrip")


@pytest.mark.parametrize("Hi,Wi,Ho,Wo,sf", [(16, 20, 64, 80, None), (64, 80, 16, 20, None), (21, 26, 10, 13, 0.5),
                                            (42, 52, 10, 13, 0.25), (33, 33, 256, 129, None), (40, 40, 40, 40, None)])
def test_bilinear(dev, Hi, Wi, Ho, Wo, sf):
    from isr2_amd import ops
    x = rnd(1, 5, Hi, Wi, dev=dev, seed=61)
    if sf is None:
        ref = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False)
    else:
        ref = F.interpolate(x, scale_factor=sf, mode="bilinear", align_corners=False)
        assert ref.shape[-2:] == (Ho, Wo)
    out = ops.resize(x.permute(0, 2, 3, 1).contiguous(), (Ho, Wo), scale_factor=sf)
    close(out.permute(0, 3, 1, 2), ref, 2e-6, "bilinear nhwc")
    out2 = ops.resize(x, (Ho, Wo), scale_factor=sf, layout="nchw")
    close(out2.permute(0, 3, 1, 2), ref, 2e-6, "bilinear planar")


def test_bicubic_x4_and_avgpool(dev):
    from isr2_amd import ops
    x = rnd(1, 3, 23, 31, dev=dev, seed=62).clamp(-1, 1) * 0.5 + 0.5
    ref = F.interpolate(x, scale_factor=4, mode="bicubic", align_corners=False)
    out = ops.resize(x, (92, 124), mode="bicubic", scale_factor=4, layout="nchw")
    close(out.permute(0, 3, 1, 2), ref, 3e-6, "bicubic")
    y = rnd(1, 3, 31, 30, dev=dev, seed=63)
    close(ops.avgpool2(y.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2), F.avg_pool2d(y, 2, 2), 1e-6, "avgpool")


@pytest.mark.parametrize("H,W", [(48, 48), (42, 52), (64, 37)])
def test_freq_bands(dev, H, W):
    """DCT / DWT / FFT bands against the CPU oracle (which is pinned to the reference)."""
    from isr2_amd import ops
    from isr2_amd.fusion import FreqBands
    from isr2_amd.weights import synth_state_dict
    from oracle import freqfusion_oracle as O
    sd = synth_state_dict(1234, parts=("fusion",))
    lr = torch.from_numpy(np.random.default_rng(3).random((1, 3, H, W), dtype=np.float32))
    ref = torch.cat(O.freq_decompose(sd, lr), dim=1).permute(0, 2, 3, 1)
    fb = FreqBands({k: v.to(dev) for k, v in sd.items()}, dev)
    out = fb(lr.to(dev))
    for i in range(9):
        close(out[..., 3 * i:3 * i + 3].cpu(), ref[..., 3 * i:3 * i + 3], 1e-5, f"band {i}")


def test_chan_attn_weights(dev):
    from isr2_amd import ops
    N, C, heads, d = 3000, 180, 6, 30
    qkv = rnd(N, 3 * C, dev=dev, seed=70)
    temp = (rnd(heads, dev=dev, seed=71) * 0.2 + 1.0).contiguous()
    wbd = ops.chan_attn_weights(qkv, 0, C, temp)
    q = qkv[:, :C].reshape(N, heads, d).permute(1, 2, 0)
    k = qkv[:, C:2 * C].reshape(N, heads, d).permute(1, 2, 0)
    a = torch.softmax((F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)) * temp[:, None, None], dim=-1)
    ref = torch.block_diag(*[a[h] for h in range(heads)])
    close(wbd, ref, 1e-5, "chan attn")


def test_band_mha_and_fusion_pointwise(dev):
    from isr2_amd import ops
    P, nb, heads, E = 500, 9, 4, 64
    qkv = rnd(P * nb, 3 * E, dev=dev, seed=72)
    out = ops.band_mha_core(qkv, P, nb, heads)
    q, k, v = [t.reshape(P, nb, heads, 16).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    ref = (torch.softmax((q * 0.25) @ k.transpose(-2, -1), dim=-1) @ v).transpose(1, 2).reshape(P * nb, E)
    close(out, ref, 1e-5, "band mha")
    x, att, imp = rnd(P, 27, dev=dev, seed=73), rnd(P, 9, dev=dev, seed=74), rnd(9, dev=dev, seed=75)
    close(ops.band_weight(x, att, imp), (x.reshape(P, 9, 3) * att[:, :, None] * imp[None, :, None]).reshape(P, 27), 1e-6, "band weight")
    b3 = rnd(1, 20, 25, 9, dev=dev, seed=76)
    m = b3.reshape(1, 20, 25, 3, 3).abs().mean(-1)
    tot = m.sum(-1, keepdim=True) + 1e-8
    close(ops.freq_guidance(b3), torch.stack([m[..., 2], m[..., 1], m[..., 0]], -1) / tot, 1e-6, "guidance")


def test_fuse_blend(dev):
    from isr2_amd import ops
    Hl, Wl = 12, 15
    Hh, Wh = 4 * Hl, 4 * Wl
    E = rnd(1, Hh, Wh, 9, dev=dev, seed=80).abs()
    hier = rnd(1, Hh, Wh, 3, dev=dev, seed=81).abs()
    guide = torch.softmax(rnd(1, Hl, Wl, 3, dev=dev, seed=82), -1)
    gates = torch.sigmoid(rnd(1, Hl, Wl, 3, dev=dev, seed=83))
    dif = torch.sigmoid(rnd(1, Hl, Wl, 1, dev=dev, seed=84))
    out = ops.fuse_blend(E, hier, guide, gates, dif)

    def up(t):
        return F.interpolate(t.permute(0, 3, 1, 2), size=(Hh, Wh), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)

    g, gt, df = up(guide), up(gates), up(dif)
    ex = [E[..., 0:3], E[..., 3:6], E[..., 6:9]]
    f0 = hier * 0.7 + 0.3 * sum(e * g[..., i:i + 1] for i, e in enumerate(ex))
    dyn = sum(e * gt[..., i:i + 1] for i, e in enumerate(ex)) / (gt.sum(-1, keepdim=True) + 1e-8)
    close(out, f0 * (1 - 0.3 * df) + dyn * (0.3 * df), 2e-6, "fuse blend")


# ---------------------------------------------------------------------------------------------------------------
# Bench-size shapes (one 256x256 LR tile = 65 536 tokens): the tile variants that only large M selects.
@pytest.mark.parametrize("M,K,N", [(65536, 360, 180),     # DAT fc2: cfg 4 (256x192, 8 waves) because N <= 192
                                   (65536, 320, 360),     # cfg 4 through r192 = 168 > 128 and K >= 320
                                   (65536, 384, 540),     # cfg 4 through r192 = 156
                                   (65536, 180, 540),     # wide N at K = 180: 128x128 tiles
                                   (16384 + 37, 720, 180)])   # ragged M just above the cfg-4 threshold
def test_linear_at_bench_size(dev, gemm_mode, M, K, N):
    """ops.linear at M = 65 536 against torch fp32 (the 256x192 tile of conv_gemm_bf16.hip needs M >= 16384
    and was never compared with anything)."""
    from isr2_amd import ops
    x = rnd(M, K, dev=dev, seed=200)
    w = rnd(N, K, dev=dev, seed=201, scale=1.0 / math.sqrt(K))
    b = rnd(N, dev=dev, seed=202, scale=0.1)
    res = rnd(M, N, dev=dev, seed=203)
    ref = res + F.gelu(F.linear(x, w, b))
    close(ops.linear(x, w, b, act="gelu", res=res), ref, GEMM_TOL[gemm_mode], "linear @ bench size")


@pytest.mark.parametrize("halo", [True, False])
@pytest.mark.parametrize("Cin,Cout,hw", [(180, 180, 256), (180, 60, 256), (64, 64, 512), (76, 64, 256)])
def test_conv3x3_at_bench_size(dev, gemm_mode, halo, Cin, Cout, hw):
    """3x3 convolutions on the bench tile's grids (256x256 tokens for HAT/DAT, 512x512 for the fusion stack): the LDS-resident
    kernel with pool partials over 256+ workgroups, and the implicit GEMM's large-M tile (halo off)."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_conv
    x = rnd(1, Cin, hw, hw, dev=dev, seed=210)
    w = rnd(Cout, Cin, 3, 3, dev=dev, seed=211, scale=1.0 / math.sqrt(9 * Cin))
    b = rnd(Cout, dev=dev, seed=212, scale=0.1)
    ref = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1)
    ops.set_halo(halo)
    try:
        out, pooled = ops.conv2d(x.permute(0, 2, 3, 1).contiguous(), pack_conv(w), b, ksize=(3, 3), pad=(1, 1), want_pool=True)
    finally:
        ops.set_halo(True)
    close(out, ref, GEMM_TOL[gemm_mode], "conv3x3 @ bench size")
    close(pooled, ref.mean(dim=(1, 2)), 2e-5, "global average pool from the conv epilogue @ bench size")


def test_chan_attn_weights_at_bench_size(dev):
    """DAT channel attention statistics over 65 536 tokens (the split-K reduction that only this size exercises)."""
    from isr2_amd import ops
    N, C, heads, d = 65536, 180, 6, 30
    qkv = torch.empty(N, 544, device=dev)[:, :540]                  # padded pitch, as token_linear produces it
    qkv.copy_(rnd(N, 3 * C, dev=dev, seed=220))
    temp = (rnd(heads, dev=dev, seed=221) * 0.2 + 1.0).contiguous()
    wbd = ops.chan_attn_weights(qkv, 0, C, temp)
    q = qkv[:, :C].double().reshape(N, heads, d).permute(1, 2, 0)
    k = qkv[:, C:2 * C].double().reshape(N, heads, d).permute(1, 2, 0)
    a = torch.softmax((F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)) * temp.double()[:, None, None], dim=-1)
    close(wbd, torch.block_diag(*[a[h] for h in range(heads)]).float(), 1e-5, "chan attn @ 65536 tokens")


@pytest.mark.parametrize("C,H,W", [(64, 1024, 1024), (128, 512, 512)])
def test_naf_dwconv_gate_pool_at_bench_size(dev, C, H, W):
    """NAFNet dw3x3 + SimpleGate + pool with 1024 row-strip partials (the HR levels of the bench tile)."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_dw
    t = rnd(1, 2 * C, H, W, dev=dev, seed=230)
    w = rnd(2 * C, 1, 3, 3, dev=dev, seed=231, scale=0.3)
    b = rnd(2 * C, dev=dev, seed=232, scale=0.1)
    y = F.conv2d(t, w, b, padding=1, groups=2 * C)
    ref = (y[:, :C] * y[:, C:]).permute(0, 2, 3, 1)
    out, pooled = ops.dwconv3_gate_pool(t.permute(0, 2, 3, 1).contiguous(), pack_dw(w), b)
    close(out, ref, 1e-5, "dw3x3 gate")
    close(pooled, ref.double().mean(dim=(1, 2)).float(), 1e-5, "dw3x3 gate pool")


@pytest.mark.parametrize("H,W", [(256, 256), (339, 510)])
def test_freq_bands_at_bench_size(dev, H, W):
    """256-point direct DFT, the 64x64 -> 256x129 mask resample, DCT/DWT at the bench tile (and config 3's image size)."""
    from isr2_amd.fusion import FreqBands
    from isr2_amd.weights import synth_state_dict
    from oracle import freqfusion_oracle as O
    sd = synth_state_dict(1234, parts=("fusion",))
    lr = torch.from_numpy(np.random.default_rng(5).random((1, 3, H, W), dtype=np.float32))
    ref = torch.cat(O.freq_decompose(sd, lr), dim=1).permute(0, 2, 3, 1)
    out = FreqBands({k: v.to(dev) for k, v in sd.items()}, dev)(lr.to(dev))
    for i in range(9):
        close(out[..., 3 * i:3 * i + 3].cpu(), ref[..., 3 * i:3 * i + 3], 2e-5, f"band {i}")


def test_dynamic_gates_matches_reference_formula(dev):
    """ff_dynamic_gates against the formula of DynamicExpertSelector.forward (fusion_network.py:222-234): threshold
    0.7 - 0.4 d, sigmoid(10 (g - th)), max-gate floor 0.9 on every gate within 1 % of the per-pixel maximum."""
    from isr2_amd import ops
    P = 40000
    g = torch.sigmoid(rnd(1, 200, 200, 3, dev=dev, seed=240) * 2.0)
    d = torch.sigmoid(rnd(1, 200, 200, 1, dev=dev, seed=241))
    g[0, 0, :8] = torch.tensor([0.5, 0.5, 0.5], device=dev)               # exact three-way ties
    g[0, 1, :8, 1] = g[0, 1, :8, 0]                                        # exact two-way ties
    out = ops.dynamic_gates(g.contiguous(), d.contiguous())
    th = 0.7 - 0.4 * d
    s = torch.sigmoid(10.0 * (g - th))
    mx = s.max(dim=-1, keepdim=True).values
    ref = torch.maximum(s, (s >= mx * 0.99).float() * 0.9)
    dlt = (out - ref).abs()
    # a gate sitting within an ulp of 0.99 * max may fall on either side of the hard comparison: bound the fraction of such pixels
    assert (dlt > 1e-5).float().mean().item() <= 1e-4, (dlt > 1e-5).float().mean().item()
    assert (out.max(dim=-1).values >= 0.9 - 1e-6).all(), "at least one expert must be selected per pixel"
    assert out.numel() == 3 * P


# ---------------------------------------------------------------------------------------------------------------
# Window-resident attention block (csrc/win_attn_fused.hip): LayerNorm + qkv projection + attention in one launch
def _ref_ln_qkv(x, g, b, wqkv, bqkv, eps=1e-5):
    xn = F.layer_norm(x, (x.shape[-1],), g, b, eps)
    return xn, F.linear(xn, wqkv, bqkv)


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
@pytest.mark.parametrize("H,W,shift", [(32, 48, 0), (32, 48, 8), (16, 16, 8), (256, 256, 8)])
def test_win_attn_fused_hat(dev, mode, H, W, shift):
    """HAB attention half (hat_arch.py:272-303 + :165-192) against torch fp32: norm1, qkv, (S)W-MSA with relative-position
    bias and shift mask; also the normalised rows side output.  256x256 is the bench tile's token grid."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_win_attn, pack_win_rel
    from oracle import freqfusion_oracle as O
    prev = ops.gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        heads, d, ws, C = 6, 30, 16, 180
        x = torch.empty(1, H, W, 192, device=dev)[..., :C]
        x.copy_(rnd(1, H, W, C, dev=dev, seed=300, scale=1.3) + 0.2)
        g, b = rnd(C, dev=dev, seed=301) * 0.1 + 1, rnd(C, dev=dev, seed=302) * 0.1
        wqkv, bqkv = rnd(3 * C, C, dev=dev, seed=303, scale=1.0 / math.sqrt(C)), rnd(3 * C, dev=dev, seed=304, scale=0.1)
        table = rnd((2 * ws - 1) ** 2, heads, dev=dev, seed=305, scale=0.5)
        pk = pack_win_attn(wqkv, bqkv, heads, d, d ** -0.5)
        relp = pack_win_rel(table.t().contiguous(), ws, ws)
        out = ops.empty_rows((1, H, W, C), dev)
        out.zero_()
        _, xn = ops.win_attn_fused(x, out, pk, relp, gamma=g, beta=b, H=H, W=W, Hp=H, Wp=W, win=(ws, ws), shift=(shift, shift),
                                   use_mask=shift > 0, want_xn=True)
        xn_ref, qkv = _ref_ln_qkv(x, g, b, wqkv, bqkv)
        close(xn, xn_ref, 2e-6, "normalised rows")
        bias = _hat_bias(table, ws, ws, heads)
        xi = torch.roll(qkv, shifts=(-shift, -shift), dims=(1, 2)) if shift else qkv
        xw = O._win_split(xi, ws, ws).reshape(-1, ws * ws, 3, heads, d).permute(2, 0, 3, 1, 4)
        mask = O._region_mask(H, W, ws, ws, ws // 2, ws // 2).to(dev) if shift else None
        o = O._softmax_attn(xw[0] * d ** -0.5, xw[1], xw[2], bias, mask).transpose(1, 2).reshape(-1, ws * ws, C)
        o = O._win_merge(o, ws, ws, H, W)
        if shift:
            o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
        close(out, o, GEMM_TOL[mode], "fused window attention")
    finally:
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
@pytest.mark.parametrize("H,W,shifted", [(32, 64, False), (48, 48, True), (64, 32, True), (256, 256, True)])
def test_win_attn_fused_dat_branches(dev, mode, H, W, shifted):
    """DAT spatial attention (dat_arch.py:501-548, :290-342): qkv projection of norm1(x), 8x32 / 32x8 branches on channel
    halves as head groups 0-2 / 3-5, zero q/k/v beyond (H, W), shift + mask, DynamicPosBias table, v side output."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_win_attn, pack_win_rel
    from oracle import freqfusion_oracle as O
    prev = ops.gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        C, half, hh, d = 180, 90, 3, 30
        m = 32
        Hp, Wp = H + (m - H % m) % m, W + (m - W % m) % m
        x = torch.empty(1, H, W, 192, device=dev)[..., :C]
        x.copy_(rnd(1, H, W, C, dev=dev, seed=310, scale=1.3) + 0.2)
        g, b = rnd(C, dev=dev, seed=311) * 0.1 + 1, rnd(C, dev=dev, seed=312) * 0.1
        wqkv, bqkv = rnd(3 * C, C, dev=dev, seed=313, scale=1.0 / math.sqrt(C)), rnd(3 * C, dev=dev, seed=314, scale=0.1)
        pk = pack_win_attn(wqkv, bqkv, 6, d, d ** -0.5)
        _, qkv = _ref_ln_qkv(x, g, b, wqkv, bqkv)
        out = torch.zeros(1, H, W, C, device=dev)
        vout = torch.zeros(1, H, W, 3 * C, device=dev)
        qkv5 = F.pad(qkv.reshape(1, H, W, 3, C), (0, 0, 0, 0, 0, Wp - W, 0, Hp - H))
        for br in range(2):
            wh, ww = (8, 32) if br == 0 else (32, 8)
            sh, sw = wh // 2, ww // 2
            rel = rnd(hh, (2 * wh - 1) * (2 * ww - 1), dev=dev, seed=316 + br, scale=0.5)
            rel6 = torch.zeros(6, rel.shape[1], device=dev)
            rel6[3 * br:3 * br + 3] = rel
            ops.win_attn_fused(x, out, pk, pack_win_rel(rel6, wh, ww), gamma=g, beta=b, H=H, W=W, Hp=Hp, Wp=Wp, win=(wh, ww),
                               shift=(sh, sw) if shifted else (0, 0), use_mask=shifted, head0=3 * br, nheads=3, zero_pad=True,
                               v_out=vout, v_off=2 * C)
            cy, cx = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
            cy, cx = cy.reshape(-1).to(dev), cx.reshape(-1).to(dev)
            idx = (cy[:, None] - cy[None, :] + wh - 1) * (2 * ww - 1) + (cx[:, None] - cx[None, :] + ww - 1)
            bias = rel[:, idx]
            t = qkv5[..., br * half:(br + 1) * half]
            if shifted:
                t = torch.roll(t, shifts=(-sh, -sw), dims=(1, 2))
            tw = [O._win_split(t[:, :, :, i], wh, ww).reshape(-1, wh * ww, hh, d).transpose(1, 2) for i in range(3)]
            mask = O._region_mask(Hp, Wp, wh, ww, sh, sw).to(dev) if shifted else None
            o = O._softmax_attn(tw[0] * d ** -0.5, tw[1], tw[2], bias, mask).transpose(1, 2).reshape(-1, wh * ww, half)
            o = O._win_merge(o, wh, ww, Hp, Wp)
            if shifted:
                o = torch.roll(o, shifts=(sh, sw), dims=(1, 2))
            close(out[..., br * half:(br + 1) * half], o[:, :H, :W], GEMM_TOL[mode], f"fused dat branch {br}")
        close(vout[..., 2 * C:], qkv[..., 2 * C:], GEMM_TOL[mode], "v side output")
    finally:
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
@pytest.mark.parametrize("M,with_conv", [(65536, True), (1000, True), (77, False), (4096, False)])
def test_token_projmlp_fused(dev, mode, M, with_conv):
    """proj + shortcut + conv_x * scale + norm2 + MLP in one launch (hat_arch.py:303-307) against the PyTorch fp32 chain."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_token_projmlp
    C, Hd = 180, 360
    att = torch.empty(M, 192, device=dev)[:, :C]
    att.copy_(rnd(M, C, dev=dev, seed=400))
    x = torch.empty(M, 192, device=dev)[:, :C]
    x.copy_(rnd(M, C, dev=dev, seed=401, scale=1.5) + 0.3)
    c2 = torch.empty(M, 192, device=dev)[:, :C]
    c2.copy_(rnd(M, C, dev=dev, seed=402))
    scale = (rnd(C, dev=dev, seed=403).abs() * 0.01).contiguous()
    g, b = rnd(C, dev=dev, seed=404) * 0.1 + 1, rnd(C, dev=dev, seed=405) * 0.1
    wp, bp = rnd(C, C, dev=dev, seed=406, scale=1.0 / math.sqrt(C)), rnd(C, dev=dev, seed=407, scale=0.1)
    w1, b1 = rnd(Hd, C, dev=dev, seed=408, scale=1.0 / math.sqrt(C)), rnd(Hd, dev=dev, seed=409, scale=0.1)
    w2, b2 = rnd(C, Hd, dev=dev, seed=410, scale=1.0 / math.sqrt(Hd)), rnd(C, dev=dev, seed=411, scale=0.1)
    x1 = x + F.linear(att, wp, bp) + (c2 * scale if with_conv else 0.0)
    ref = x1 + F.linear(F.gelu(F.linear(F.layer_norm(x1, (C,), g, b, 1e-5), w1, b1)), w2, b2)
    pk = pack_token_projmlp(wp, bp, w1, b1, w2, b2)
    prev = ops.gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        out = ops.token_projmlp(att, x, pk, g, b, c2=c2 if with_conv else None, c2_scale=scale if with_conv else None)
    finally:
        ops.set_gemm_mode(prev)
    close(out, ref, GEMM_TOL[mode], "token_projmlp")


@pytest.mark.parametrize("H,W", [(256, 256), (37, 45)])
def test_sgfn_layernorm_on_load(dev, H, W):
    """DAT SGFN first half (dat_arch.py:163-166 fc1 + GELU, :117-123 SpatialGate): ff_token_linear's LayerNorm statistics side
    output + ff_dwconv3x3_ln (normalise on load, zero padding after the normalisation) against the PyTorch fp32 chain."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_token_linear, pack_dw
    C, Hd = 180, 720
    M = H * W
    x = torch.empty(1, H, W, 192, device=dev)[..., :C]
    x.copy_(rnd(1, H, W, C, dev=dev, seed=500, scale=1.5) + 0.3)
    g, b = rnd(C, dev=dev, seed=501) * 0.1 + 1, rnd(C, dev=dev, seed=502) * 0.1
    w1, b1 = rnd(Hd, C, dev=dev, seed=503, scale=1.0 / math.sqrt(C)), rnd(Hd, dev=dev, seed=504, scale=0.1)
    g2, bb2 = rnd(Hd // 2, dev=dev, seed=505) * 0.1 + 1, rnd(Hd // 2, dev=dev, seed=506) * 0.1
    wd, bd = rnd(Hd // 2, 1, 3, 3, dev=dev, seed=507, scale=0.3), rnd(Hd // 2, dev=dev, seed=508, scale=0.1)
    y_ref = F.gelu(F.linear(F.layer_norm(x, (C,), g, b, 1e-5), w1, b1))
    x1, x2 = y_ref[..., :Hd // 2], y_ref[..., Hd // 2:]
    x2n = F.layer_norm(x2, (Hd // 2,), g2, bb2, 1e-5).permute(0, 3, 1, 2)
    ref = x1 * F.conv2d(x2n, wd, bd, padding=1, groups=Hd // 2).permute(0, 2, 3, 1)
    y, stats = ops.token_linear(x, pack_token_linear(w1, b1), gamma=g, beta=b, act="gelu", stats_range=(Hd // 2, Hd))
    close(y, y_ref, 6e-5, "fc1 + gelu")
    mean_ref = x2.mean(-1).reshape(M)
    rstd_ref = 1.0 / torch.sqrt(x2.var(-1, unbiased=False).reshape(M) + 1e-5)
    close(stats[:, 0], mean_ref, 3e-5, "token mean")
    assert ((stats[:, 1] - rstd_ref).abs() / rstd_ref).max().item() < 2e-4
    z = ops.dwconv3x3_ln(y[..., Hd // 2:], pack_dw(wd), bd, stats, g2, bb2, mul_in=y[..., :Hd // 2])
    close(z, ref, 1e-4, "SpatialGate with LayerNorm on load")


@pytest.mark.parametrize("M", [65536, 3000, 256, 77])
def test_chan_qkv_attn_fused(dev, M):
    """DAT channel attention front end (dat_arch.py:617-641): LayerNorm + qkv + per-head L2-normalised gram + temperature + softmax,
    ff_chan_qkv + ff_chan_attn_finish against the PyTorch fp32 chain; v side output; deterministic."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_chan_qkv
    C, heads, d = 180, 6, 30
    x = torch.empty(M, 192, device=dev)[:, :C]
    x.copy_(rnd(M, C, dev=dev, seed=700, scale=1.4) + 0.2)
    g, b = rnd(C, dev=dev, seed=701) * 0.1 + 1, rnd(C, dev=dev, seed=702) * 0.1
    wqkv, bqkv = rnd(3 * C, C, dev=dev, seed=703, scale=1.0 / math.sqrt(C)), rnd(3 * C, dev=dev, seed=704, scale=0.1)
    temp = (rnd(heads, dev=dev, seed=705) * 0.2 + 1.0).contiguous()
    v, wbd = ops.chan_qkv_attn(x, pack_chan_qkv(wqkv, bqkv), g, b, temp)
    qkv = F.linear(F.layer_norm(x, (C,), g, b, 1e-5), wqkv, bqkv).double()
    q = qkv[:, :C].reshape(M, heads, d).permute(1, 2, 0)
    k = qkv[:, C:2 * C].reshape(M, heads, d).permute(1, 2, 0)
    a = torch.softmax((F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)) * temp.double()[:, None, None], dim=-1)
    close(wbd, torch.block_diag(*[a[h] for h in range(heads)]).float(), 2e-5, "fused channel attention matrix")
    close(v, qkv[:, 2 * C:].float(), GEMM_TOL["bf16x3"], "v side output")
    v2, wbd2 = ops.chan_qkv_attn(x, pack_chan_qkv(wqkv, bqkv), g, b, temp)
    assert torch.equal(wbd, wbd2) and torch.equal(v, v2), "not deterministic"


@pytest.mark.parametrize("M", [65536, 1000, 77])
def test_token_linear_gated(dev, M):
    """DAT adaptive interaction + projection + residual in one launch (dat_arch.py:541-559 / :649-666) against the PyTorch chain:
    sm = sigmoid(w2 . gelu(W1 s + b1) + b2), out = res + proj(s * cm + o * sm)."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_token_linear_gated
    C, Hd = 180, 11
    sp = torch.empty(M, 192, device=dev)[:, :C]
    sp.copy_(rnd(M, C, dev=dev, seed=800))
    ch = torch.empty(M, 192, device=dev)[:, :C]
    ch.copy_(rnd(M, C, dev=dev, seed=801))
    res = torch.empty(M, 192, device=dev)[:, :C]
    res.copy_(rnd(M, C, dev=dev, seed=802))
    cm = torch.sigmoid(rnd(C, dev=dev, seed=803)).contiguous()
    w1, b1 = rnd(Hd, C, dev=dev, seed=804, scale=1.0 / math.sqrt(C)), rnd(Hd, dev=dev, seed=805, scale=0.1)
    w2, b2 = rnd(1, Hd, dev=dev, seed=806, scale=0.5), 0.13
    wp, bp = rnd(C, C, dev=dev, seed=807, scale=1.0 / math.sqrt(C)), rnd(C, dev=dev, seed=808, scale=0.1)
    sm = torch.sigmoid(F.linear(F.gelu(F.linear(sp, w1, b1)), w2) + b2)
    ref = res + F.linear(sp * cm + ch * sm, wp, bp)
    out = ops.token_linear_gated(sp, ch, pack_token_linear_gated(wp, bp, w1, b1, w2), cm, b2, res=res)
    close(out, ref, GEMM_TOL["bf16x3"], "token_linear_gated")


def test_bilinear_vec4_nhwc(dev):
    """float4 channel-last bilinear path (C % 4 == 0, 16-byte aligned rows) incl. writing into a channel slice of a wider buffer."""
    from isr2_amd import ops
    x = rnd(1, 64, 37, 45, dev=dev, seed=900)
    ref = F.interpolate(x, size=(74, 90), mode="bilinear", align_corners=False)
    buf = torch.zeros(1, 74, 90, 76, device=dev)
    ops.resize(x.permute(0, 2, 3, 1).contiguous(), (74, 90), out=buf[..., :64], mul=0.5)
    close(buf[..., :64].permute(0, 3, 1, 2), 0.5 * ref, 2e-6, "bilinear vec4")
    assert buf[..., 64:].abs().max() == 0
    down = ops.resize(x.permute(0, 2, 3, 1).contiguous(), (18, 22), scale_factor=0.5)
    close(down.permute(0, 3, 1, 2), F.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=False), 2e-6, "bilinear vec4 down")


@pytest.mark.parametrize("kh,kw", [(5, 5), (1, 21), (21, 1)])
def test_dwconv_large_kernels_at_bench_size(dev, kh, kw):
    """The register-tiled depth-wise kernels of the LKA chain on its real geometry (9 x 64 channels at 256 x 256)."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_dw
    C = 576
    x = rnd(1, C, 256, 256, dev=dev, seed=910)
    w = rnd(C, 1, kh, kw, dev=dev, seed=911, scale=0.2)
    ref = F.conv2d(x, w, None, padding=(kh // 2, kw // 2), groups=C)
    out = ops.dwconv2d(x.permute(0, 2, 3, 1).contiguous(), pack_dw(w), None, ksize=(kh, kw), pad=(kh // 2, kw // 2))
    close(out.permute(0, 3, 1, 2), ref, 1e-5, "dwconv large kernel")


@pytest.mark.parametrize("Cin,Cout,H,W,act,with_res", [(32, 3, 1024, 1024, None, True), (8, 16, 256, 260, "gelu", False), (16, 3, 100, 75, "sigmoid", False),
                                                       (8, 1, 128, 128, "sigmoid", False), (6, 16, 96, 64, "gelu", False), (7, 9, 64, 64, "sigmoid", False)])
def test_conv3x3_small_matches_torch(dev, Cin, Cout, H, W, act, with_res):
    """fp32 VALU 3x3 convolution for small channel counts (the fusion stack's tail layers) against torch fp32, incl. ragged tiles,
    unaligned Cin (6, 27), residual + alpha, strided input slices."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_conv
    prev = ops.gemm_mode()
    ops.set_gemm_mode("bf16x3")
    try:
        wide = rnd(1, H, W, Cin + 4, dev=dev, seed=920)
        x = wide[..., :Cin]
        w = rnd(Cout, Cin, 3, 3, dev=dev, seed=921, scale=1.0 / math.sqrt(9 * Cin))
        b = rnd(Cout, dev=dev, seed=922, scale=0.1)
        res = rnd(1, H, W, Cout, dev=dev, seed=923) if with_res else None
        f = {"gelu": F.gelu, "sigmoid": torch.sigmoid, None: lambda t: t}[act]
        ref = f(F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1)).permute(0, 2, 3, 1) * 0.1
        if with_res:
            ref = ref + res
        wp = pack_conv(w)
        out = ops.conv2d(x, wp, b, ksize=(3, 3), pad=(1, 1), act=act, res=res, alpha=0.1)
        assert ops.PREPARED.peek(wp, "small") is not None, "small-channel kernel was not selected"
        close(out, ref, 3e-6, "conv3x3 small")
    finally:
        ops.set_gemm_mode(prev)


# ---------------------------------------------------------------------------------------------------------------
# Plain-bf16 forms (nterms = 1) of the fused kernels: BASELINE configs[1] names bf16.  One MFMA per product instead of three; results
# are bf16-grade (operands rounded to 8 mantissa bits, fp32 accumulation).
BF16_TOL = 1.5e-2     # relative to max|ref| for K <= 1620 contractions of O(1) operands


def _both_modes(fn):
    """fn() evaluated in 'bf16x3' and in 'bf16': -> (split result, plain result)."""
    from isr2_amd import ops
    prev = ops.gemm_mode()
    try:
        ops.set_gemm_mode("bf16x3")
        a = fn()
        ops.set_gemm_mode("bf16")
        b = fn()
    finally:
        ops.set_gemm_mode(prev)
    return a, b


def test_plain_bf16_forms_of_the_fused_kernels(dev):
    from isr2_amd import ops
    from isr2_amd.prep import (pack_token_linear, pack_token_mlp, pack_token_projmlp, pack_chan_qkv, pack_token_linear_gated, pack_dw,
                               pack_conv, pack_naf_ffn)
    C, Hd, M = 180, 360, 3000
    x = rnd(M, C, dev=dev, seed=1000, scale=1.5) + 0.3
    g, b = rnd(C, dev=dev, seed=1001) * 0.1 + 1, rnd(C, dev=dev, seed=1002) * 0.1
    w1, b1 = rnd(Hd, C, dev=dev, seed=1003, scale=1.0 / math.sqrt(C)), rnd(Hd, dev=dev, seed=1004, scale=0.1)
    w2, b2 = rnd(C, Hd, dev=dev, seed=1005, scale=1.0 / math.sqrt(Hd)), rnd(C, dev=dev, seed=1006, scale=0.1)
    wp, bp = rnd(C, C, dev=dev, seed=1007, scale=1.0 / math.sqrt(C)), rnd(C, dev=dev, seed=1008, scale=0.1)
    att, res = rnd(M, C, dev=dev, seed=1009), rnd(M, C, dev=dev, seed=1010)

    def check(name, pair, ref, tol=BF16_TOL):
        s3, s1 = pair
        close(s3, ref, 1e-4, name + " (bf16x3)")
        close(s1, ref, tol, name + " (bf16)")
        assert not torch.equal(s3, s1), name + ": the plain-bf16 form returned the split result (nterms not honoured?)"

    # token_linear: LayerNorm + fc1 + GELU, and proj + residual
    ref = F.gelu(F.linear(F.layer_norm(x, (C,), g, b, 1e-5), w1, b1))
    check("token_linear ln+gelu", _both_modes(lambda: ops.token_linear(x, pack_token_linear(w1, b1), gamma=g, beta=b, act="gelu")), ref)
    check("token_linear proj+res", _both_modes(lambda: ops.token_linear(att, pack_token_linear(wp, bp), res=res)), res + F.linear(att, wp, bp))
    # token_mlp / token_projmlp
    ref = x + F.linear(F.gelu(F.linear(F.layer_norm(x, (C,), g, b, 1e-5), w1, b1)), w2, b2)
    check("token_mlp", _both_modes(lambda: ops.token_mlp(x, g, b, pack_token_mlp(w1, b1, w2, b2))), ref)
    x1 = x + F.linear(att, wp, bp)
    ref = x1 + F.linear(F.gelu(F.linear(F.layer_norm(x1, (C,), g, b, 1e-5), w1, b1)), w2, b2)
    check("token_projmlp", _both_modes(lambda: ops.token_projmlp(att, x, pack_token_projmlp(wp, bp, w1, b1, w2, b2), g, b)), ref)
    # gated projection (DAT)
    cm = torch.sigmoid(rnd(C, dev=dev, seed=1011)).contiguous()
    gw1, gb1, gw2 = rnd(11, C, dev=dev, seed=1012, scale=1.0 / math.sqrt(C)), rnd(11, dev=dev, seed=1013, scale=0.1), rnd(1, 11, dev=dev, seed=1014, scale=0.5)
    sm = torch.sigmoid(F.linear(F.gelu(F.linear(x, gw1, gb1)), gw2) + 0.13)
    ref = res + F.linear(x * cm + att * sm, wp, bp)
    check("token_linear_gated", _both_modes(lambda: ops.token_linear_gated(x, att, pack_token_linear_gated(wp, bp, gw1, gb1, gw2), cm, 0.13, res=res)), ref)
    # channel-attention front end: the v side output and the attention matrix
    wqkv, bqkv = rnd(3 * C, C, dev=dev, seed=1015, scale=1.0 / math.sqrt(C)), rnd(3 * C, dev=dev, seed=1016, scale=0.1)
    temp = (rnd(6, dev=dev, seed=1017) * 0.2 + 1.0).contiguous()
    (v3, a3), (v1, a1) = _both_modes(lambda: ops.chan_qkv_attn(x, pack_chan_qkv(wqkv, bqkv), g, b, temp))
    qkv = F.linear(F.layer_norm(x, (C,), g, b, 1e-5), wqkv, bqkv)
    check("chan_qkv v", (v3, v1), qkv[:, 2 * C:])
    close(a1, a3, 2e-2, "chan attention matrix bf16 vs bf16x3")
    # LDS-resident 3x3 convolution
    xc = rnd(1, 40, 48, 64, dev=dev, seed=1018)
    wc, bc = rnd(60, 64, 3, 3, dev=dev, seed=1019, scale=1.0 / math.sqrt(576)), rnd(60, dev=dev, seed=1020, scale=0.1)
    wpk = pack_conv(wc)
    ref = F.gelu(F.conv2d(xc.permute(0, 3, 1, 2), wc, bc, padding=1)).permute(0, 2, 3, 1)
    pair = _both_modes(lambda: ops.conv2d(xc, wpk, bc, ksize=(3, 3), pad=(1, 1), act="gelu"))
    assert ops.PREPARED.peek(wpk, "halo") is not None and ops.PREPARED.peek(wpk, "halo1") is not None     # both weight images were built
    check("conv3x3 halo", pair, ref)
    for ci_, co_, hw_ in ((180, 60, (37, 45)), (60, 180, (40, 33)), (180, 180, (33, 20)), (76, 64, (64, 70)), (64, 32, (50, 40))):   # ragged tiles, every launch form
        xq = rnd(1, hw_[0], hw_[1], ci_, dev=dev, seed=1040 + ci_)
        wq, bq = rnd(co_, ci_, 3, 3, dev=dev, seed=1041 + co_, scale=1.0 / math.sqrt(9 * ci_)), rnd(co_, dev=dev, seed=1042, scale=0.1)
        wqp = pack_conv(wq)
        rq = rnd(1, hw_[0], hw_[1], co_, dev=dev, seed=1043)
        refq = rq + F.conv2d(xq.permute(0, 3, 1, 2), wq, bq, padding=1).permute(0, 2, 3, 1)
        check(f"conv3x3 halo {ci_}->{co_}", _both_modes(lambda: ops.conv2d(xq, wqp, bq, ksize=(3, 3), pad=(1, 1), res=rq)), refq)
    # NAFBlock halves
    Cn = 64
    xn_ = rnd(1, Cn, 40, 56, dev=dev, seed=1021, scale=1.5) + 0.2
    gn, bn_ = rnd(Cn, dev=dev, seed=1022) * 0.1 + 1, rnd(Cn, dev=dev, seed=1023) * 0.1
    nw1, nb1 = rnd(2 * Cn, Cn, dev=dev, seed=1024, scale=1.0 / math.sqrt(Cn)), rnd(2 * Cn, dev=dev, seed=1025, scale=0.1)
    nw2, nb2 = rnd(2 * Cn, 1, 3, 3, dev=dev, seed=1026, scale=0.3), rnd(2 * Cn, dev=dev, seed=1027, scale=0.1)
    mu = xn_.mean(1, keepdim=True)
    ln = (xn_ - mu) / ((xn_ - mu).pow(2).mean(1, keepdim=True) + 1e-6).sqrt() * gn.view(1, -1, 1, 1) + bn_.view(1, -1, 1, 1)
    t = F.conv2d(F.conv2d(ln, nw1.view(2 * Cn, Cn, 1, 1), nb1), nw2, nb2, padding=1, groups=2 * Cn)
    ref = (t[:, :Cn] * t[:, Cn:]).permute(0, 2, 3, 1)
    xh = xn_.permute(0, 2, 3, 1).contiguous()
    (g3, p3), (g1, p1) = _both_modes(lambda: ops.naf_front(xh, pack_token_linear(nw1, nb1), gn, bn_, pack_dw(nw2), nb2))
    check("naf_front", (g3, g1), ref, tol=3e-2)
    w4, b4 = rnd(2 * Cn, Cn, dev=dev, seed=1028, scale=1.0 / math.sqrt(Cn)), rnd(2 * Cn, dev=dev, seed=1029, scale=0.1)
    w5, b5 = rnd(Cn, Cn, dev=dev, seed=1030, scale=1.0 / math.sqrt(Cn)), rnd(Cn, dev=dev, seed=1031, scale=0.1)
    gam = rnd(Cn, dev=dev, seed=1032) * 0.3
    y = rnd(2000, Cn, dev=dev, seed=1033, scale=1.5) + 0.2
    lny = F.layer_norm(y, (Cn,), gn, bn_, 1e-6)
    h = F.linear(lny, w4, b4)
    ref = y + gam * F.linear(h[:, :Cn] * h[:, Cn:], w5, b5)
    check("naf_ffn", _both_modes(lambda: ops.naf_ffn(y, pack_naf_ffn(w4, b4, w5, b5), gn, bn_, gam)), ref, tol=3e-2)


@pytest.mark.parametrize("H,W", [(256, 256), (48, 64), (37, 45), (33, 32), (40, 100)])
def test_cab_fused_equals_the_two_launch_path(dev, H, W):
    """HAT's CAB in one launch (csrc/cab_fused.hip, plain bf16): conv3x3 180 -> 60, GELU, conv3x3 60 -> 180 and the average pool,
    bit-identical to the two LDS-resident 3x3 launches it replaces (same K order, same rounding points), incl. ragged tiles -- the
    first convolution's values OUTSIDE the image must be the second one's zero padding, not conv1 of padded input."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_conv
    C, Cm = 180, 60
    prev = ops.gemm_mode()
    ops.set_gemm_mode("bf16")
    try:
        x = torch.empty(1, H, W, 192, device=dev)[..., :C]
        x.copy_(rnd(1, H, W, C, dev=dev, seed=1100, scale=1.2) + 0.1)
        w1, b1 = rnd(Cm, C, 3, 3, dev=dev, seed=1101, scale=1.0 / math.sqrt(9 * C)), rnd(Cm, dev=dev, seed=1102, scale=0.2)
        w2, b2 = rnd(C, Cm, 3, 3, dev=dev, seed=1103, scale=1.0 / math.sqrt(9 * Cm)), rnd(C, dev=dev, seed=1104, scale=0.2)
        w1p, w2p = pack_conv(w1), pack_conv(w2)
        got, pooled = ops.cab_fused(x, w1p, b1, w2p, b2)
        c1 = ops.conv2d(x, w1p, b1, ksize=(3, 3), pad=(1, 1), act="gelu")
        want, pooled2 = ops.conv2d(c1, w2p, b2, ksize=(3, 3), pad=(1, 1), want_pool=True)
        assert torch.equal(got, want), float((got - want).abs().max())
        close(pooled, pooled2, 1e-6, "pooled")
        ref = F.conv2d(F.gelu(F.conv2d(x.permute(0, 3, 1, 2), w1, b1, padding=1)), w2, b2, padding=1).permute(0, 2, 3, 1)
        close(got, ref, BF16_TOL, "cab fused vs fp32")
        close(pooled, ref.mean(dim=(1, 2)), BF16_TOL, "pooled vs fp32")
    finally:
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("rows,C,Ch", [(256, 180, 6), (1024, 180, 22), (7, 64, 4), (300, 200, 64)])
def test_pool_vec_mlp_fused(dev, rows, C, Ch):
    """Pool finish + channel-attention MLP in one launch (hat_arch.py:50-54; dat_arch.py:412-416) against the two launches it
    replaces and against torch; also the first pool stage alone (ff_pool_partials)."""
    from isr2_amd import ops
    part = rnd(rows, C, dev=dev, seed=900, scale=3.0)
    W1, b1 = rnd(Ch, C, dev=dev, seed=901, scale=0.2), rnd(Ch, dev=dev, seed=902, scale=0.1)
    W2, b2 = rnd(C, Ch, dev=dev, seed=903, scale=0.5), rnd(C, dev=dev, seed=904, scale=0.1)
    pp = ops.PoolPartials(part, C, 1.0 / 4096.0)
    fused = ops.vec_mlp(pp, W1, b1, "relu", W2, b2, "sigmoid", post=0.01)
    two = ops.vec_mlp(pp.mean(), W1, b1, "relu", W2, b2, "sigmoid", post=0.01)
    v = part.double().sum(0, keepdim=True) / 4096.0
    ref = torch.sigmoid(F.linear(torch.relu(F.linear(v, W1.double(), b1.double())), W2.double(), b2.double())) * 0.01
    close(fused, ref.float(), 2e-6, "fused vs torch")
    close(fused, two, 2e-6, "fused vs two launches")
    x = rnd(1, 37, 29, C, dev=dev, seed=905)
    p2 = ops.pool_partials(x)
    close(p2.mean(), x.mean((1, 2)), 2e-6, "pool partials")


@pytest.mark.parametrize("H,W", [(256, 256), (37, 45), (8, 32), (19, 70)])
def test_sgfn_tail_fused(dev, H, W):
    """DAT SGFN tail in one launch (dat_arch.py:117-123 SpatialGate, :163-170 fc2, :736 residual; plain bf16) against the PyTorch
    fp32 chain: LayerNorm statistics from ff_token_linear's epilogue, zero padding AFTER the normalisation, ragged tiles."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_token_linear, pack_dw
    C, Hd = 180, 720
    c2 = Hd // 2
    prev = ops.gemm_mode()
    ops.set_gemm_mode("bf16")
    try:
        x = torch.empty(1, H, W, 192, device=dev)[..., :C]
        x.copy_(rnd(1, H, W, C, dev=dev, seed=910, scale=1.5) + 0.3)
        g, b = rnd(C, dev=dev, seed=911) * 0.1 + 1, rnd(C, dev=dev, seed=912) * 0.1
        w1, b1 = rnd(Hd, C, dev=dev, seed=913, scale=1.0 / math.sqrt(C)), rnd(Hd, dev=dev, seed=914, scale=0.1)
        g2, bb2 = rnd(c2, dev=dev, seed=915) * 0.1 + 1, rnd(c2, dev=dev, seed=916) * 0.1
        wd, bd = rnd(c2, 1, 3, 3, dev=dev, seed=917, scale=0.3), rnd(c2, dev=dev, seed=918, scale=0.1)
        w2, b2 = rnd(C, c2, dev=dev, seed=919, scale=1.0 / math.sqrt(c2)), rnd(C, dev=dev, seed=920, scale=0.1)
        y, stats = ops.token_linear(x, pack_token_linear(w1, b1), gamma=g, beta=b, act="gelu", stats_range=(c2, Hd))
        out = ops.sgfn_tail(y, c2, pack_dw(wd), bd, stats, g2, bb2, w2, b2, res=x)
        # reference from the SAME fc1 output (the fused kernel is what is under test), fp64
        yd = y.double()
        x1, x2 = yd[..., :c2], yd[..., c2:]
        x2n = F.layer_norm(x2, (c2,), g2.double(), bb2.double(), 1e-5).permute(0, 3, 1, 2)
        gate = x1 * F.conv2d(x2n, wd.double(), bd.double(), padding=1, groups=c2).permute(0, 2, 3, 1)
        ref = (x.double() + F.linear(gate, w2.double(), b2.double())).float()
        close(out, ref, GEMM_TOL["bf16"], "sgfn tail")
        # and against the two-launch bf16 path it replaces
        z = ops.dwconv3x3_ln(y[..., c2:], pack_dw(wd), bd, stats, g2, bb2, mul_in=y[..., :c2])
        two = ops.linear(z, w2, b2, res=x)
        close(out, two, 4e-3, "sgfn tail vs two launches")
    finally:
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("H,W", [(32, 32), (48, 80), (256, 256)])
def test_ocab_attn_persistent(dev, H, W):
    """HAT OCAB attention in the persistent per-window kernel (hat_arch.py:392-438; plain bf16) against the unfold-based torch chain:
    zero keys outside the image still enter the softmax, bias through the reference's wrapped (negative) table indices."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_rel_overlap
    from oracle import freqfusion_oracle as O
    heads, d, ws, ows, C = 6, 30, 16, 24, 180
    prev = ops.gemm_mode()
    ops.set_gemm_mode("bf16")
    try:
        qkv = rnd(1, H, W, 3 * C, dev=dev, seed=930)
        table = rnd((ws + ows - 1) ** 2, heads, dev=dev, seed=931, scale=0.5)
        bias = _hat_bias(table, ws, ows, heads)
        out = ops.empty_rows((1, H, W, C), dev)
        out.zero_()
        ops.ocab_attn(qkv, out, pack_rel_overlap(table, ws, ows), q_off=0, k_off=C, v_off=2 * C, H=H, W=W, heads=heads, d=d, ws=ws, ows=ows,
                      scale=d ** -0.5)
        q = O._win_split(qkv[..., :C], ws, ws)
        kv = qkv[..., C:].permute(0, 3, 1, 2)
        nwin = (H // ws) * (W // ws)
        kvw = F.unfold(kv, kernel_size=ows, stride=ws, padding=(ows - ws) // 2)
        kvw = kvw.reshape(1, 2, C, ows * ows, nwin).permute(1, 0, 4, 3, 2).reshape(2, nwin, ows * ows, C)
        qh = q.reshape(-1, ws * ws, heads, d).transpose(1, 2) * d ** -0.5
        kh = kvw[0].reshape(-1, ows * ows, heads, d).transpose(1, 2)
        vh = kvw[1].reshape(-1, ows * ows, heads, d).transpose(1, 2)
        o = O._softmax_attn(qh, kh, vh, bias, None).transpose(1, 2).reshape(-1, ws * ws, C)
        close(out, O._win_merge(o, ws, ws, H, W), GEMM_TOL["bf16"], "ocab persistent")
        two = torch.zeros(1, H, W, C, device=dev)
        ops.window_attn(qkv, two, bias.transpose(1, 2).contiguous(), q_off=0, k_off=C, v_off=2 * C, o_off=0, H=H, W=W, Hp=H, Wp=W,
                        win=(ws, ws), kwin=(ows, ows), shift=(0, 0), use_mask=False, heads=heads, d=d, scale=d ** -0.5)
        close(out, two, 4e-3, "ocab persistent vs two-stage kernel")
    finally:
        ops.set_gemm_mode(prev)


def test_bf16_intermediate_rows_are_bit_identical(dev):
    """Plain-bf16 mode: att, the normalised rows and conv1's output stored as bf16 give the SAME results as the fp32 rows (their
    consumers round to bf16 as MFMA operands); conv2's bf16 output changes x1 by at most 2^-9 * conv_scale * |c2|."""
    from isr2_amd import ops
    from isr2_amd.prep import pack_win_attn, pack_win_rel, pack_token_projmlp, pack_conv
    prev = ops.gemm_mode()
    ops.set_gemm_mode("bf16")
    try:
        H = W = 64
        heads, d, ws, C, Hd = 6, 30, 16, 180, 360
        x = torch.empty(1, H, W, 192, device=dev)[..., :C]
        x.copy_(rnd(1, H, W, C, dev=dev, seed=940, scale=1.3) + 0.2)
        g, b = rnd(C, dev=dev, seed=941) * 0.1 + 1, rnd(C, dev=dev, seed=942) * 0.1
        pk = pack_win_attn(rnd(3 * C, C, dev=dev, seed=943, scale=1.0 / math.sqrt(C)), rnd(3 * C, dev=dev, seed=944, scale=0.1), heads, d, d ** -0.5)
        relp = pack_win_rel(rnd(heads, (2 * ws - 1) ** 2, dev=dev, seed=945, scale=0.5), ws, ws)
        w1, b1 = pack_conv(rnd(60, C, 3, 3, dev=dev, seed=946, scale=0.03)), rnd(60, dev=dev, seed=947, scale=0.1)
        w2, b2 = pack_conv(rnd(C, 60, 3, 3, dev=dev, seed=948, scale=0.05)), rnd(C, dev=dev, seed=949, scale=0.1)
        pm = pack_token_projmlp(rnd(C, C, dev=dev, seed=950, scale=1.0 / math.sqrt(C)), rnd(C, dev=dev, seed=951, scale=0.1),
                                rnd(Hd, C, dev=dev, seed=952, scale=1.0 / math.sqrt(C)), rnd(Hd, dev=dev, seed=953, scale=0.1),
                                rnd(C, Hd, dev=dev, seed=954, scale=1.0 / math.sqrt(Hd)), rnd(C, dev=dev, seed=955, scale=0.1))
        g2, bb2 = rnd(C, dev=dev, seed=956) * 0.1 + 1, rnd(C, dev=dev, seed=957) * 0.1
        scale = torch.full((C,), 0.01, device=dev)

        def block(b16, c2_b16):
            att = ops.empty_rows_bf16((1, H, W, C), dev) if b16 else ops.empty_rows((1, H, W, C), dev)
            _, xn = ops.win_attn_fused(x, att, pk, relp, gamma=g, beta=b, H=H, W=W, Hp=H, Wp=W, win=(ws, ws), shift=(8, 8), use_mask=True,
                                       want_xn="bf16" if b16 else True)
            c1 = ops.conv2d(xn, w1, b1, ksize=(3, 3), pad=(1, 1), act="gelu", out_bf16=b16)
            c2, _ = ops.conv2d(c1, w2, b2, ksize=(3, 3), pad=(1, 1), want_pool=True, out_bf16=c2_b16)
            return ops.token_projmlp(att, x, pm, g2, bb2, c2=c2, c2_scale=scale), c2
        ref, c2f = block(False, False)
        same, c2s = block(True, False)
        assert same.dtype == torch.float32 and c2s.dtype == torch.float32
        assert torch.equal(c2f, c2s), "bf16 xn / c1 rows changed the convolution branch"
        assert torch.equal(ref, same), "bf16 att / xn / c1 rows changed the block output"
        near, c2h = block(True, True)
        assert c2h.dtype == torch.bfloat16
        err = (near - ref).abs().max().item()
        print("bf16 c2 rows: max|d| =", err, "max|ref| =", ref.abs().max().item(), "max|c2| =", c2f.abs().max().item())
        # x1 moves by <= 2^-9 * 0.01 * |c2| = 1e-4; that flips the bf16 rounding of some normalised values, i.e. it re-draws a little of
        # the bf16 noise the mode has anyway (measured 8e-4 of max|ref| here; the mode's bar against fp32 is 2e-2)
        assert err <= 2e-3 * max(1.0, ref.abs().max().item())
    finally:
        ops.set_gemm_mode(prev)
