"""Parity at the BENCHMARKED sizes: the 256x256 tile bench.py times and BASELINE config 3's
510x339 image cut into 256/32 tiles, against goldens produced by the imported reference itself
(tests/golden/make_golden_big.py -> t256_nat.npz, config3_510x339.npz), plus the CPU oracle run on the box at 128x128
and a B = 2 batch against the reference's batched forward (b2_48.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

# (max deviation relative to max(1,|ref|) on every tap / sample / crop, min PSNR vs the reference output in dB)
BARS = {"f32": (5e-5, 120.0), "bf16x3": (2e-4, 110.0)}
GATE_FLIP_FRAC = 1e-3


@pytest.fixture(scope="module", params=["f32", "bf16x3"])
def model(request, synth_sd):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from isr2_amd import ops
    from isr2_amd.model import FreqFusionHIP
    old = ops.gemm_mode()
    ops.set_gemm_mode(request.param)
    m = FreqFusionHIP(synth_sd, "cuda:0")
    m.mode = request.param
    yield m
    ops.set_gemm_mode(old)


def _as_ref_layout(t, shape):
    t = t.detach().float().cpu()
    shape = tuple(int(v) for v in shape)
    if t.dim() == 4 and len(shape) == 4 and tuple(t.shape) != shape:          # NHWC -> NCHW
        t = t.permute(0, 3, 1, 2)
    t = t.contiguous()
    if tuple(t.shape) != shape:                                                # NHWC tokens -> (b, h*w, c)
        t = t.reshape(shape)
    return t


def _psnr_from_samples(got, ref):
    mse = float(((got.double() - ref.double()) ** 2).mean())
    return 10.0 * np.log10(1.0 / max(mse, 1e-30))


def test_tile256_against_reference_golden(model):
    """The bench workload (bench.make_tile(100), 256x256 -> 1024x1024) against the reference's own forward on that tile:
    every intermediate tap (4096 samples each + whole-tensor L2), 65 536 samples and four 64x64 crops of the final image
    and of each expert output."""
    g = np.load(os.path.join(GOLD, "t256_nat.npz"))
    lr = torch.from_numpy(g["lr"]).cuda()
    taps = {}
    out = model(lr, taps)
    taps["final"] = out
    tol, min_psnr = BARS[model.mode]
    worst = {}
    for n in sorted({k.split("/")[1] for k in g.files if k.startswith("tap/")}):
        t = _as_ref_layout(taps[n], g[f"tap/{n}/shape"])
        ref = torch.from_numpy(g[f"tap/{n}/val"])
        got = t.reshape(-1)[torch.from_numpy(g[f"tap/{n}/idx"])]
        if n == "fusion.gates":
            flips = ((got - ref).abs() > 1e-3).float().mean().item()
            print("t256", model.mode, "fusion.gates: fraction differing by > 1e-3 =", flips)
            assert flips <= GATE_FLIP_FRAC, flips
            continue
        worst[n] = (got - ref).abs().max().item() / max(1.0, float(ref.abs().max()))
        l2_ref = float(g[f"tap/{n}/stats"][2])
        l2 = float(torch.sqrt((t.double() ** 2).sum()))
        assert abs(l2 - l2_ref) <= 1e-4 * max(l2_ref, 1.0), (n, l2, l2_ref)
    print("t256", model.mode, "worst taps:", sorted(worst.items(), key=lambda kv: -kv[1])[:8])
    bad = {k: v for k, v in worst.items() if not v < tol}
    assert not bad, bad
    corners = g["crop_corners"]
    for k in ("final", "expert.hat", "expert.dat", "expert.nafnet"):
        t = taps[k].detach().float().cpu()
        ref = torch.from_numpy(g[f"big/{k}/val"])
        got = t.reshape(-1)[torch.from_numpy(g[f"big/{k}/idx"])]
        assert (got - ref).abs().max().item() < tol, (k, (got - ref).abs().max().item())
        for c, (y, x) in enumerate(corners):
            d = (t[0, :, y:y + 64, x:x + 64] - torch.from_numpy(g[f"crop/{k}"][c])).abs().max().item()
            assert d < tol, (k, c, d)
        if k == "final":
            psnr = _psnr_from_samples(got, ref)
            print("t256", model.mode, "PSNR(hip, reference) over 65536 samples =", psnr)
            assert psnr >= min_psnr


def test_config3_tiled_against_reference_golden(model):
    """BASELINE config 3: one 510x339 image through the plugin's _tiled_forward(256, 32) -- six 256x256 model calls blended
    on the device -- against the reference's own `_tiled_forward(model, lr, 256, 32)` (io.py:82-121) on the same image."""
    import models.team29_FreqFusion.io as plug
    g = np.load(os.path.join(GOLD, "config3_510x339.npz"))
    lr = torch.from_numpy(g["lr"]).cuda()
    out = plug._tiled_forward(model, lr, tile_size=256, overlap=32, scale=4, device=lr.device).cpu()
    assert tuple(out.shape) == tuple(int(v) for v in g["shape"]) == (1, 3, 1356, 2040)
    tol, min_psnr = BARS[model.mode]
    ref = torch.from_numpy(g["big/val"])
    got = out.reshape(-1)[torch.from_numpy(g["big/idx"])]
    d = (got - ref).abs().max().item()
    psnr = _psnr_from_samples(got, ref)
    print("config3", model.mode, "max|d| over 65536 samples =", d, "PSNR =", psnr)
    assert d < tol and psnr >= min_psnr
    for c, (y, x) in enumerate(g["crop_corners"]):                           # includes crops that straddle the blend seams
        dc = (out[0, :, y:y + 64, x:x + 64] - torch.from_numpy(g["crops"][c])).abs().max().item()
        assert dc < tol, (c, int(y), int(x), dc)
    l2 = float(torch.sqrt((out.double() ** 2).sum()))
    assert abs(l2 - float(g["stats"][2])) <= 1e-5 * float(g["stats"][2])


@pytest.fixture(scope="module")
def oracle128(synth_sd):
    """The CPU oracle on a 128x128 1/f tile, taps included (about 10-20 s on the box's host cores), shared by both modes."""
    from oracle import freqfusion_oracle as O
    import bench
    lr = bench.make_tile(7, 128)
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    taps = {}
    out = O.forward(synth_sd, lr, taps)
    taps["final"] = out
    return lr, taps


def test_against_oracle_128(model, oracle128):
    """HIP vs the oracle at 128x128 on whole tensors (every tap the oracle records), not samples."""
    from oracle import freqfusion_oracle as O
    lr, otaps = oracle128
    taps = {}
    out = model(lr.cuda(), taps)
    taps["final"] = out
    tol, min_psnr = BARS[model.mode]
    worst = {}
    for n, ref in otaps.items():
        if n not in taps:
            continue
        t = _as_ref_layout(taps[n], ref.shape)
        if n == "fusion.gates":
            flips = ((t - ref).abs() > 1e-3).float().mean().item()
            print("oracle128", model.mode, "fusion.gates flips", flips)
            assert flips <= GATE_FLIP_FRAC
            continue
        worst[n] = (t - ref).abs().max().item() / max(1.0, float(ref.abs().max()))
    print("oracle128", model.mode, "taps compared:", len(worst), "worst:", sorted(worst.items(), key=lambda kv: -kv[1])[:6])
    assert len(worst) >= 20
    bad = {k: v for k, v in worst.items() if not v < tol}
    assert not bad, bad
    psnr = O.psnr(out.cpu(), otaps["final"])
    print("oracle128", model.mode, "PSNR(hip, oracle) =", psnr)
    assert psnr >= min_psnr


def test_batch2_matches_reference_and_single(model):
    """B = 2: the reference's batched forward on two different 48x48 images (b2_48.npz), and bit-equality of the batched
    HIP forward with two B = 1 forwards (every op on the path is per-image: pools, channel attention and FFT are per sample)."""
    g = np.load(os.path.join(GOLD, "b2_48.npz"))
    lr = torch.from_numpy(g["lr"]).cuda()
    out = model(lr)
    assert tuple(out.shape) == (2, 3, 192, 192)
    tol, min_psnr = BARS[model.mode]
    ref = torch.from_numpy(g["out"])
    assert (out.cpu() - ref).abs().max().item() < tol
    one = torch.cat([model(lr[0:1]), model(lr[1:2])], 0)
    assert torch.equal(out, one), "batched forward differs from two single forwards"


# ---------------------------------------------------------------------------------------------------------------------
# Plain-bf16 contraction mode (FF_GEMM=bf16, BASELINE configs[1] names bf16): one MFMA term instead of three.  It is not the
# headline mode, but it is a product mode, so it carries its own bar: PSNR against the REFERENCE's output
# >= 60 dB (a 30 dB PSNR-vs-ground-truth then moves by < 0.005 dB) and every tap within BF16_TAP_TOL of the reference.
BF16_BAR = (6e-2, 60.0)      # measured: worst tap 0.035 (NAFNet's 1024-channel bottleneck), 69 dB at all three sizes


@pytest.fixture(scope="module")
def model_bf16(synth_sd):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from isr2_amd import ops
    from isr2_amd.model import FreqFusionHIP
    old = ops.gemm_mode()
    ops.set_gemm_mode("bf16")
    m = FreqFusionHIP(synth_sd, "cuda:0")
    m.mode = "bf16"
    yield m
    ops.set_gemm_mode(old)


@pytest.mark.parametrize("case", ["c48_u8", "c42x52_nat", "t256_nat"])
def test_plain_bf16_mode_against_reference_goldens(model_bf16, case):
    g = np.load(os.path.join(GOLD, case + ".npz"))
    lr = torch.from_numpy(g["lr"]).cuda()
    taps = {}
    out = model_bf16(lr, taps)
    taps["final"] = out
    tol, min_psnr = BF16_BAR
    worst = {}
    for n in sorted({k.split("/")[1] for k in g.files if k.startswith("tap/")}):
        if n == "fusion.gates" or n not in taps:
            continue                                                    # hard threshold: covered by the final image's bar
        t = _as_ref_layout(taps[n], g[f"tap/{n}/shape"])
        ref = torch.from_numpy(g[f"tap/{n}/val"])
        got = t.reshape(-1)[torch.from_numpy(g[f"tap/{n}/idx"])]
        worst[n] = (got - ref).abs().max().item() / max(1.0, float(ref.abs().max()))
    print(case, "bf16 worst taps:", sorted(worst.items(), key=lambda kv: -kv[1])[:8])
    bad = {k: v for k, v in worst.items() if not v < tol}
    assert not bad, bad
    if f"full/final" in g.files:
        psnr = _psnr_from_samples(out.cpu().reshape(-1), torch.from_numpy(g["full/final"]).reshape(-1))
    else:
        psnr = _psnr_from_samples(out.cpu().reshape(-1)[torch.from_numpy(g["big/final/idx"])], torch.from_numpy(g["big/final/val"]))
    print(case, "bf16 PSNR(hip, reference) =", psnr)
    assert psnr >= min_psnr
    if case == "t256_nat":
        # the north-star bar itself, |PSNR(build, GT) - PSNR(reference, GT)| <= 0.01 dB, at synthesised ground truths that put the
        # reference at 25 / 30 / 35 / 40 dB (bench.delta_psnr_vs_ground_truth: the evidence the bench line carries for its bf16 headline)
        import bench
        got = out.cpu().reshape(-1)[torch.from_numpy(g["big/final/idx"])].double().numpy()
        rep = bench.delta_psnr_vs_ground_truth(got, g["big/final/val"].astype(np.float64))
        print("bf16 delta PSNR vs synthesised ground truth:", {k: round(v["delta_dB"], 5) for k, v in rep["points"].items()})
        assert rep["worst_abs_delta_dB"] <= 0.01, rep


def test_whole_image_510x339_against_reference_golden(model):
    """The WHOLE-IMAGE branch above 256x256 (VERDICT r2 missing #2): the reference runs every image as one `model(lr)` first
    (models/team29_FreqFusion/io.py:219-221).  510x339 = 172 890 tokens: reflect pad to 512x352 in HAT / DAT, split-K channel
    attention, 510- / 339-point DFTs, NAFNet at 2040x1356 -- against the reference's own forward on that image
    (tests/golden/whole510_339.npz, make_golden_big.py whole510), as a model call and through the plugin's _forward_image."""
    import models.team29_FreqFusion.io as plug
    g = np.load(os.path.join(GOLD, "whole510_339.npz"))
    lr = torch.from_numpy(g["lr"]).cuda()
    tol, min_psnr = BARS[model.mode]
    ref = torch.from_numpy(g["big/val"])
    idx = torch.from_numpy(g["big/idx"])
    for how in ("model", "plugin"):
        out = (model(lr) if how == "model" else plug._forward_image(model, lr, "whole510.png", lr.device, {})).cpu()
        assert tuple(out.shape) == tuple(int(v) for v in g["shape"]) == (1, 3, 1356, 2040)
        got = out.reshape(-1)[idx]
        d = (got - ref).abs().max().item()
        psnr = _psnr_from_samples(got, ref)
        print("whole510", how, model.mode, "max|d| over 65536 samples =", d, "PSNR =", psnr)
        assert d < tol and psnr >= min_psnr
        for c, (y, x) in enumerate(g["crop_corners"]):
            dc = (out[0, :, y:y + 64, x:x + 64] - torch.from_numpy(g["crops"][c])).abs().max().item()
            assert dc < tol, (how, c, int(y), int(x), dc)
        l2 = float(torch.sqrt((out.double() ** 2).sum()))
        assert abs(l2 - float(g["stats"][2])) <= 1e-5 * float(g["stats"][2])


def test_plain_bf16_whole_image_510x339(model_bf16):
    """The plain-bf16 kernels (persistent OCAB attention, rescheduled window attention / proj + MLP, SGFN tail) on the whole 510x339
    image: 172 890 tokens, reflect pad to 512x352 in HAT / DAT, ragged 8x32 tiles in the SGFN tail -- against the reference's forward
    (tests/golden/whole510_339.npz) at the bf16 bar."""
    g = np.load(os.path.join(GOLD, "whole510_339.npz"))
    lr = torch.from_numpy(g["lr"]).cuda()
    tol, min_psnr = BF16_BAR
    out = model_bf16(lr).cpu()
    assert tuple(out.shape) == (1, 3, 1356, 2040)
    got = out.reshape(-1)[torch.from_numpy(g["big/idx"])]
    ref = torch.from_numpy(g["big/val"])
    d = (got - ref).abs().max().item()
    psnr = _psnr_from_samples(got, ref)
    print("whole510 bf16: max|d| over 65536 samples =", d, "PSNR =", psnr)
    assert d < tol and psnr >= min_psnr
    for c, (y, x) in enumerate(g["crop_corners"]):
        dc = (out[0, :, y:y + 64, x:x + 64] - torch.from_numpy(g["crops"][c])).abs().max().item()
        assert dc < tol, (c, int(y), int(x), dc)
