"""Training-side kernels (csrc/train_ops.hip) and the tape (isr2_amd/autograd.py) against torch.autograd on the CPU, op by op:
forward values and every gradient.  fp32 ('f32' contraction mode) unless noted; tolerances relative to max(1, |ref|)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module", autouse=True)
def _mode():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from isr2_amd import ops
    old = ops.gemm_mode()
    ops.set_gemm_mode("f32")
    yield
    ops.set_gemm_mode(old)


def rnd(*shape, seed=0, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def close(got, ref, tol=TOL, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert tuple(got.shape) == tuple(ref.shape), (what, got.shape, ref.shape)
    d = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert d < tol, (what, d)


def run_tape(build, inputs, gy):
    """inputs: dict name -> CPU tensor (all get gradients).  build(vars) -> Var.  Returns (y, grads dict) from the HIP tape."""
    from isr2_amd import autograd as ag
    vs = {k: ag.Var(v.cuda().contiguous(), True) for k, v in inputs.items()}
    with ag.Tape() as tape:
        y = build(vs)
        y.grad, y.owned = gy.cuda().contiguous(), True
        tape.backward()
    torch.cuda.synchronize()
    return y.data.cpu(), {k: (v.grad.cpu() if v.grad is not None else None) for k, v in vs.items()}


def ref_run(fn, inputs, gy):
    ts = {k: v.clone().requires_grad_(True) for k, v in inputs.items()}
    y = fn(ts)
    y.backward(gy)
    return y.detach(), {k: t.grad for k, t in ts.items()}


def pack(w):     # OIHW -> [O, KH*KW*I]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def unpack(wp, shape):
    o, i, kh, kw = shape
    return wp.reshape(o, kh, kw, i).permute(0, 3, 1, 2)


@pytest.mark.parametrize("cin,cout,k,act", [(3, 64, 3, "gelu"), (73, 64, 3, "gelu"), (64, 32, 3, None), (32, 1, 3, "sigmoid"), (64, 3, 1, "sigmoid"),
                                            (27, 64, 1, "gelu"), (6, 16, 3, "relu"), (96, 32, 3, "gelu")])
def test_conv2d_forward_and_gradients(cin, cout, k, act):
    from isr2_amd import autograd as ag
    B, H, W = 2, 20, 24
    x, w, b = rnd(B, cin, H, W, seed=1), rnd(cout, cin, k, k, seed=2, scale=0.2), rnd(cout, seed=3, scale=0.1)
    gy = rnd(B, cout, H, W, seed=4)
    actf = {None: lambda t: t, "gelu": F.gelu, "relu": F.relu, "sigmoid": torch.sigmoid}[act]
    yr, gr = ref_run(lambda t: actf(F.conv2d(t["x"], t["w"], t["b"], padding=k // 2)), dict(x=x, w=w, b=b), gy)
    y, g = run_tape(lambda v: ag.conv2d(v["x"], v["w"], v["b"], ksize=(k, k), act=act),
                    dict(x=x.permute(0, 2, 3, 1).contiguous(), w=pack(w), b=b), gy.permute(0, 2, 3, 1).contiguous())
    close(y.permute(0, 3, 1, 2), yr, what="y")
    close(g["x"].permute(0, 3, 1, 2), gr["x"], what="dx")
    close(unpack(g["w"], w.shape), gr["w"], tol=5e-5, what="dw")
    close(g["b"], gr["b"], tol=5e-5, what="db")


def test_linear_on_rows_and_strided_slice():
    from isr2_amd import autograd as ag
    M, K, N = 4 * 36 * 9, 3, 64
    x, w, b, gy = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    yr, gr = ref_run(lambda t: F.linear(t["x"], t["w"], t["b"]), dict(x=x, w=w, b=b), gy)
    y, g = run_tape(lambda v: ag.conv2d(v["x"], v["w"], v["b"]), dict(x=x, w=w, b=b), gy)
    close(y, yr); close(g["x"], gr["x"]); close(g["w"], gr["w"], tol=5e-5); close(g["b"], gr["b"], tol=5e-5)
    # a channel slice of a wider NHWC tensor as the input of a 3x3 convolution (band attention: 3 of 27 channels)
    xb, w3, gy3 = rnd(2, 16, 16, 27, seed=5), rnd(1, 3, 3, 3, seed=6), rnd(2, 16, 16, 1, seed=7)
    yr, gr = ref_run(lambda t: torch.sigmoid(F.conv2d(t["x"].permute(0, 3, 1, 2)[:, 6:9], t["w"], None, padding=1)).permute(0, 2, 3, 1),
                     dict(x=xb, w=w3), gy3)
    y, g = run_tape(lambda v: ag.conv2d(ag.slice_ch(v["x"], 6, 9), v["w"], None, ksize=(3, 3), act="sigmoid"), dict(x=xb, w=pack(w3)), gy3)
    close(y, yr); close(g["x"], gr["x"]); close(unpack(g["w"], w3.shape), gr["w"], tol=5e-5)


@pytest.mark.parametrize("k", [(5, 5), (1, 21), (21, 1), (3, 3)])
def test_dwconv_gradients(k):
    from isr2_amd import autograd as ag
    B, H, W, C = 3, 24, 28, 64
    x, w, gy = rnd(B, C, H, W, seed=1), rnd(C, 1, k[0], k[1], seed=2, scale=0.3), rnd(B, C, H, W, seed=3)
    yr, gr = ref_run(lambda t: F.conv2d(t["x"], t["w"], None, padding=(k[0] // 2, k[1] // 2), groups=C), dict(x=x, w=w), gy)
    y, g = run_tape(lambda v: ag.dwconv2d(v["x"], v["w"], k), dict(x=x.permute(0, 2, 3, 1).contiguous(), w=w.reshape(C, -1).t().contiguous()),
                    gy.permute(0, 2, 3, 1).contiguous())
    close(y.permute(0, 3, 1, 2), yr); close(g["x"].permute(0, 3, 1, 2), gr["x"])
    close(g["w"].t().reshape(w.shape), gr["w"], tol=5e-5)


@pytest.mark.parametrize("hi,wi,ho,wo,sf", [(16, 16, 64, 64, None), (64, 48, 16, 12, None), (64, 48, 32, 24, None), (16, 20, 8, 10, 0.5),
                                            (64, 64, 16, 16, 0.25), (64, 64, 64, 33, None), (17, 13, 40, 31, None)])
def test_resize_adjoint(hi, wi, ho, wo, sf):
    from isr2_amd import autograd as ag
    B, C = 2, 5
    x, gy = rnd(B, C, hi, wi, seed=1), rnd(B, C, ho, wo, seed=2)
    if sf is None:
        fn = lambda t: F.interpolate(t["x"], size=(ho, wo), mode="bilinear", align_corners=False)
    else:
        fn = lambda t: F.interpolate(t["x"], scale_factor=sf, mode="bilinear", align_corners=False)
    yr, gr = ref_run(fn, dict(x=x), gy)
    y, g = run_tape(lambda v: ag.resize(v["x"], (ho, wo), scale_factor=sf), dict(x=x.permute(0, 2, 3, 1).contiguous()), gy.permute(0, 2, 3, 1).contiguous())
    close(y.permute(0, 3, 1, 2), yr); close(g["x"].permute(0, 3, 1, 2), gr["x"])


def test_avgpool_layernorm_meanpool():
    from isr2_amd import autograd as ag
    x, gy = rnd(2, 3, 32, 24, seed=1), rnd(2, 3, 16, 12, seed=2)
    yr, gr = ref_run(lambda t: F.avg_pool2d(t["x"], 2, 2), dict(x=x), gy)
    y, g = run_tape(lambda v: ag.avgpool2(v["x"]), dict(x=x.permute(0, 2, 3, 1).contiguous()), gy.permute(0, 2, 3, 1).contiguous())
    close(y.permute(0, 3, 1, 2), yr); close(g["x"].permute(0, 3, 1, 2), gr["x"])
    for C in (64, 128):
        x, ga, be, gy = rnd(700, C, seed=3), rnd(C, seed=4) * 0.2 + 1, rnd(C, seed=5) * 0.1, rnd(700, C, seed=6)
        yr, gr = ref_run(lambda t: F.layer_norm(t["x"], (C,), t["g"], t["b"]), dict(x=x, g=ga, b=be), gy)
        y, g = run_tape(lambda v: ag.layernorm(v["x"], v["g"], v["b"]), dict(x=x, g=ga, b=be), gy)
        close(y, yr); close(g["x"], gr["x"]); close(g["g"], gr["g"], tol=5e-5); close(g["b"], gr["b"], tol=5e-5)
    x, gy = rnd(3, 10, 12, 32, seed=7), rnd(3, 32, seed=8)
    yr, gr = ref_run(lambda t: t["x"].mean(dim=(1, 2)), dict(x=x), gy)
    y, g = run_tape(lambda v: ag.mean_pool(v["x"]), dict(x=x), gy)
    close(y, yr); close(g["x"], gr["x"])


@pytest.mark.parametrize("G", [1, 3])
def test_batchnorm_training_mode(G):
    """Batch statistics per group of images (one module call per group), running statistics updated call by call."""
    from isr2_amd import autograd as ag
    B, H, W, C = 2, 12, 10, 64
    x = rnd(G * B, H, W, C, seed=1) * 1.5 + 0.7
    ga, be, gy = rnd(C, seed=2) * 0.2 + 1, rnd(C, seed=3) * 0.1, rnd(G * B, H, W, C, seed=4)
    rm0, rv0 = rnd(C, seed=5) * 0.1, rnd(C, seed=6).abs() + 0.5
    rm, rv = rm0.clone(), rv0.clone()

    def ref(t):
        outs = []
        for g_ in range(G):
            xi = t["x"][g_ * B:(g_ + 1) * B].permute(0, 3, 1, 2)
            outs.append(F.batch_norm(xi, rm, rv, t["g"], t["b"], True, 0.1, 1e-5).permute(0, 2, 3, 1))
        return torch.cat(outs, 0)
    yr, gr = ref_run(ref, dict(x=x, g=ga, b=be), gy)
    rmd, rvd = rm0.clone().cuda(), rv0.clone().cuda()

    def hip(v):
        x2 = ag.reshape(v["x"], (-1, C))
        return ag.reshape(ag.batchnorm_train(x2, v["g"], v["b"], rmd, rvd, groups=G), (G * B, H, W, C))
    y, g = run_tape(hip, dict(x=x, g=ga, b=be), gy)
    close(y, yr); close(g["x"], gr["x"]); close(g["g"], gr["g"], tol=5e-5); close(g["b"], gr["b"], tol=5e-5)
    close(rmd, rm, what="running_mean"); close(rvd, rv, what="running_var")


@pytest.mark.parametrize("nt,heads", [(9, 4), (3, 8)])
def test_band_mha_gradients(nt, heads):
    from isr2_amd import autograd as ag
    P, E = 300, heads * 16
    qkv, gy = rnd(P * nt, 3 * E, seed=1), rnd(P * nt, E, seed=2)

    def ref(t):
        q, k, v = [u.reshape(P, nt, heads, 16).transpose(1, 2) for u in t["qkv"].chunk(3, dim=-1)]
        a = torch.softmax((q * 0.25) @ k.transpose(-2, -1), dim=-1) @ v
        return a.transpose(1, 2).reshape(P * nt, E)
    yr, gr = ref_run(ref, dict(qkv=qkv), gy)
    y, g = run_tape(lambda v: ag.band_mha(v["qkv"], P, nt, heads, 0.0, 0), dict(qkv=qkv), gy)
    close(y, yr); close(g["qkv"], gr["qkv"])


def test_band_mha_dropout_is_consistent_between_forward_and_backward():
    """Dropout on the attention weights (p = 0.1): the kept fraction is right, E[out] is preserved, and the backward uses the
    forward's mask -- checked by finite differences through the HIP forward itself (the mask is a pure function of the seed)."""
    from isr2_amd import autograd as ag
    P, nt, heads, E = 2000, 9, 4, 64
    qkv = rnd(P * nt, 3 * E, seed=1)
    y0, _ = run_tape(lambda v: ag.band_mha(v["qkv"], P, nt, heads, 0.0, 0), dict(qkv=qkv), torch.zeros(P * nt, E))
    y1, _ = run_tape(lambda v: ag.band_mha(v["qkv"], P, nt, heads, 0.1, 1234), dict(qkv=qkv), torch.zeros(P * nt, E))
    y2, _ = run_tape(lambda v: ag.band_mha(v["qkv"], P, nt, heads, 0.1, 1234), dict(qkv=qkv), torch.zeros(P * nt, E))
    y3, _ = run_tape(lambda v: ag.band_mha(v["qkv"], P, nt, heads, 0.1, 99), dict(qkv=qkv), torch.zeros(P * nt, E))
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    assert abs((y1 - y0).mean().item()) < 5e-3                                   # unbiased: kept weights are scaled by 1 / (1 - p)
    gy = rnd(P * nt, E, seed=2)
    _, g = run_tape(lambda v: ag.band_mha(v["qkv"], P, nt, heads, 0.1, 1234), dict(qkv=qkv), gy)
    d = rnd(P * nt, 3 * E, seed=3)
    eps = 1e-2
    yp, _ = run_tape(lambda v: ag.band_mha(v["qkv"], P, nt, heads, 0.1, 1234), dict(qkv=qkv + eps * d), gy)
    ym, _ = run_tape(lambda v: ag.band_mha(v["qkv"], P, nt, heads, 0.1, 1234), dict(qkv=qkv - eps * d), gy)
    fd = ((yp - ym).double() * gy.double()).sum().item() / (2 * eps)
    an = (g["qkv"].double() * d.double()).sum().item()
    assert abs(fd - an) <= 2e-3 * max(1.0, abs(an)), (fd, an)


def test_pointwise_vocabulary():
    from isr2_amd import autograd as ag
    x, y2, gy = rnd(500, 9, seed=1), rnd(500, 9, seed=2), rnd(500, 9, seed=3)
    r, s, gc = rnd(500, 1, seed=4), rnd(1, seed=5), rnd(5, 9, seed=6)
    one = torch.ones(1).cuda()

    def ref(t):
        a = F.gelu(t["x"]) * t["y"] + 0.3 * torch.sigmoid(t["y"])
        a = a * t["r"] * t["s"] + F.softplus(t["x"]) - t["x"].abs()
        a = a * t["gc"].repeat_interleave(100, dim=0)
        a = (a + 0.25).clamp(0, 1) + F.relu(t["y"]) * 1.0 / (t["x"].abs().sum(dim=1, keepdim=True) + 1e-8)
        return a + torch.exp(0.1 * t["x"]) + t["s"].clamp(min=0.5) * t["x"][:, 2:5].sum(dim=1, keepdim=True)

    def hip(v):
        a = ag.add(ag.mul(ag.unary("gelu", v["x"]), v["y"]), ag.unary("sigmoid", v["y"]), 0.3)
        a = ag.add(ag.add(ag.mul_scalar(ag.mul_row(a, v["r"]), v["s"]), ag.unary("softplus", v["x"])), ag.unary("abs", v["x"]), -1.0)
        a = ag.mul_gc(a, v["gc"], 100)
        a = ag.unary("clamp01", ag.add_const(a, 0.25, one))
        a = ag.add(a, ag.mul_row(ag.unary("relu", v["y"]), ag.unary("recip_eps", ag.sum_ch(ag.unary("abs", v["x"])), 1e-8)))
        a = ag.add(a, ag.unary("exp", ag.scale(v["x"], 0.1)))
        col = ag.mul_scalar(ag.sum_ch(ag.slice_ch(v["x"], 2, 5)), ag.unary("clamp_min", v["s"], 0.5))
        ones = ag.const(torch.ones(500, 9).cuda())
        return ag.add(a, ag.mul_row(ones, col))
    ins = dict(x=x, y=y2, r=r, s=s, gc=gc)
    yr, gr = ref_run(ref, ins, gy)
    y, g = run_tape(hip, ins, gy)
    close(y, yr)
    for k in ins:
        close(g[k], gr[k], tol=5e-5, what=k)


def test_cat_slice_rows_permute_dynamic_gates():
    from isr2_amd import autograd as ag
    a, b, gy = rnd(2, 8, 8, 64, seed=1), rnd(2, 8, 8, 9, seed=2), rnd(2, 8, 8, 76, seed=3)
    yr, gr = ref_run(lambda t: F.pad(torch.cat([t["a"], t["b"]], -1), (0, 3)), dict(a=a, b=b), gy)
    y, g = run_tape(lambda v: ag.cat_ch([v["a"], v["b"]], pad_to=76), dict(a=a, b=b), gy)
    close(y, yr); close(g["a"], gr["a"]); close(g["b"], gr["b"])
    x, gy = rnd(6, 5, 4, 8, seed=4), rnd(2, 5, 4, 8, seed=5)
    yr, gr = ref_run(lambda t: t["x"][2:4] * 2.0, dict(x=x), gy)
    y, g = run_tape(lambda v: ag.scale(ag.slice_rows(v["x"], 2, 4), 2.0), dict(x=x), gy)
    close(y, yr); close(g["x"], gr["x"])
    x, gy = rnd(30 * 9, 16, seed=6), rnd(9, 30, 16, seed=7)
    yr, gr = ref_run(lambda t: t["x"].reshape(30, 9, 16).transpose(0, 1), dict(x=x), gy)
    y, g = run_tape(lambda v: ag.permute_rows(v["x"], 30, 9, 16), dict(x=x), gy)
    close(y, yr); close(g["x"], gr["x"])
    graw, dif, gy = torch.sigmoid(rnd(4000, 3, seed=8)), torch.sigmoid(rnd(4000, 1, seed=9)), rnd(4000, 3, seed=10)

    def ref(t):
        gg = torch.sigmoid(10.0 * (t["g"] - (0.7 - 0.4 * t["d"])))
        top = gg.max(dim=1, keepdim=True)[0]
        return torch.maximum(gg, (gg >= top * 0.99).float() * 0.9)
    yr, gr = ref_run(ref, dict(g=graw, d=dif), gy)
    y, g = run_tape(lambda v: ag.dynamic_gates(v["g"], v["d"]), dict(g=graw, d=dif), gy)
    close(y, yr); close(g["g"], gr["g"], tol=1e-4); close(g["d"], gr["d"], tol=1e-4)


@pytest.mark.parametrize("h,w", [(16, 16), (64, 64), (24, 40)])
def test_fft_mask_gradient(h, w):
    """irfft2(rfft2(x) * sigmoid(T * bilinear(logits))) and its gradient wrt the 64x64 logits and the temperature."""
    from isr2_amd import autograd as ag
    B = 2
    x, logits, temp, gy = rnd(B * 3, h, w, seed=1), rnd(1, 1, 64, 64, seed=2), torch.tensor([2.5]), rnd(B * 3, h, w, seed=3)
    Wf = w // 2 + 1

    def ref(t):
        xf = torch.fft.rfft2(x, norm="ortho")
        m = torch.sigmoid(F.interpolate(t["l"], size=(h, Wf), mode="bilinear", align_corners=False) * t["t"].clamp(min=1.0))
        return torch.fft.irfft2(xf * m[0], s=(h, w), norm="ortho")
    yr, gr = ref_run(ref, dict(l=logits, t=temp), gy)
    fft = ag.FFT2(torch.device("cuda:0"))
    xd = x.cuda()
    X = fft.rfft2(xd)
    close(X.cpu(), torch.view_as_real(torch.fft.rfft2(x, norm="ortho")), tol=2e-5, what="rfft2 (re, im)")

    def hip(v):
        z = ag.resize(ag.reshape(v["l"], (1, 64, 64, 1)), (h, Wf))
        m = ag.unary("sigmoid", ag.mul_scalar(z, ag.unary("clamp_min", v["t"], 1.0)))
        return ag.fft_lowpass(fft, xd, X, ag.reshape(m, (h, Wf)))
    y, g = run_tape(hip, dict(l=logits.reshape(1, 4096), t=temp), gy)
    close(y, yr, tol=5e-5); close(g["l"].reshape(1, 1, 64, 64), gr["l"], tol=5e-5); close(g["t"], gr["t"], tol=5e-5)


def test_l1_adamw_ema_against_torch():
    """ff_l1_loss_grad, ff_grad_sqnorm, ff_adamw_ema_step against F.l1_loss / clip_grad_norm_ / torch.optim.AdamW / the EMA formula."""
    import math
    from isr2_amd import lib as L, autograd as ag
    lib = L.load()
    n = 100003
    sr, hr = rnd(n, seed=1) * 0.6 + 0.5, torch.rand(n, generator=torch.Generator().manual_seed(2))
    srt = sr.clone().requires_grad_(True)
    lossr = (srt.clamp(0, 1) - hr).abs().mean()
    lossr.backward()
    d = {k: v.cuda() for k, v in dict(sr=sr, hr=hr).items()}
    dsr, loss, work = torch.empty(n).cuda(), torch.zeros(1).cuda(), torch.empty(1024).cuda()
    L.check(lib.ff_l1_loss_grad(d["sr"].data_ptr(), d["hr"].data_ptr(), dsr.data_ptr(), n, loss.data_ptr(), work.data_ptr(), 1024, ag._st()))
    assert abs(loss.item() - lossr.item()) < 1e-6
    close(dsr, srt.grad, tol=1e-9)
    for clip in (1.0, 0.05):
        p0, gs = rnd(n, seed=3), [rnd(n, seed=10 + i) * 0.01 for i in range(3)]
        pt = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([pt], lr=1.5e-4, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8)
        ema_r = p0.clone()
        P, M, V, E = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda(), p0.clone().cuda()
        hyper, sq = torch.zeros(10).cuda(), torch.zeros(1).cuda()
        for t, g in enumerate(gs, 1):
            pt.grad = g.clone()
            gn = torch.nn.utils.clip_grad_norm_([pt], clip)
            opt.step()
            ema_r = 0.9995 * ema_r + (1 - 0.9995) * pt.data
            G = g.clone().cuda()
            hyper.copy_(torch.tensor([1.5e-4, 0.9, 0.999, 1e-8, 1e-4, clip, 0.9995, t, 1.5e-4 / (1 - 0.9 ** t), math.sqrt(1 - 0.999 ** t)]))
            L.check(lib.ff_grad_sqnorm(G.data_ptr(), n, sq.data_ptr(), work.data_ptr(), 1024, ag._st()))
            L.check(lib.ff_adamw_ema_step(P.data_ptr(), G.data_ptr(), M.data_ptr(), V.data_ptr(), E.data_ptr(), n, hyper.data_ptr(), sq.data_ptr(), ag._st()))
            assert abs(math.sqrt(sq.item()) - float(gn)) < 1e-5 * float(gn)
        dp, de = (P.cpu() - pt.data).abs(), (E.cpu() - ema_r).abs()
        assert dp.max().item() < 2e-6 and dp.mean().item() < 1e-7, (dp.max().item(), dp.mean().item())   # a few ulp at |p| ~ 4; 3 steps move p by 4.5e-4
        assert de.max().item() < 1e-6 and de.mean().item() < 1e-7
