"""End-to-end GPU parity: the HIP path against (a) the committed reference goldens and (b) the CPU oracle,
stage by stage.  Bar (BASELINE north_star): within 0.01 dB PSNR of the reference in fp32; we additionally
require every intermediate tap within 2e-4 (relative to max(1,|ref|)) and PSNR(hip, reference) >= 110 dB
(measured: 4.3e-5 / 123.8 dB for bf16x3, 3e-6 / 143 dB for f32)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TAP_TOL = 5e-4


# per contraction mode: (max tap deviation relative to max(1,|ref|), min PSNR vs the reference in dB)
BARS = {"f32": (5e-5, 125.0), "bf16x3": (2e-4, 110.0)}
GATE_FLIP_FRAC = 1e-3      # dynamic-selection gates have a hard threshold: at most 0.1 % of pixels may differ by > 1e-3


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


def _to_nchw_like(t, ref_shape):
    t = t.detach().float().cpu()
    if t.dim() == 4 and tuple(t.shape) != tuple(ref_shape) and tuple(t.permute(0, 3, 1, 2).shape) == tuple(ref_shape):
        t = t.permute(0, 3, 1, 2)
    return t.contiguous()


@pytest.mark.parametrize("case", ["c48_u8", "c42x52_nat"])
def test_against_reference_goldens(model, case):
    from oracle import freqfusion_oracle as O
    g = np.load(os.path.join(HERE, "golden", case + ".npz"))
    lr = torch.from_numpy(g["lr"]).cuda()
    taps = {}
    out = model(lr, taps).cpu()
    taps["final"] = out
    worst = {}
    for k in ("expert.hat", "expert.dat", "expert.nafnet", "final"):
        ref = torch.from_numpy(g["full/" + k])
        worst[k] = (taps[k].cpu() - ref).abs().max().item()
    names = sorted({k.split("/")[1] for k in g.files if k.startswith("tap/")})
    for n in names:
        shape = tuple(int(v) for v in g[f"tap/{n}/shape"])
        t = taps[n].detach().float().cpu()
        if t.dim() == 4 and len(shape) == 4 and tuple(t.shape) != shape:          # NHWC -> NCHW
            t = t.permute(0, 3, 1, 2)
        if t.dim() == 4 and len(shape) == 3:                                        # NHWC tokens -> (b, h*w, c)
            t = t.reshape(shape)
        t = t.contiguous()
        if tuple(t.shape) != shape:                                                # padded token grids (DAT taps are unpadded)
            t = t.reshape(shape)
        ref = torch.from_numpy(g[f"tap/{n}/val"])
        got = t.reshape(-1)[torch.from_numpy(g[f"tap/{n}/idx"])]
        if n == "fusion.gates":
            # hard threshold (fusion_network.py:232-234: gates >= 0.99 max): single pixels may legitimately land on the other
            # side, so the bar is on the FRACTION of pixels that differ, not on the worst one
            flips = ((got - ref).abs() > 1e-3).float().mean().item()
            print(case, model.mode, "fusion.gates: fraction differing by > 1e-3 =", flips)
            assert flips <= GATE_FLIP_FRAC, flips
            continue
        worst[n] = (got - ref).abs().max().item() / max(1.0, float(ref.abs().max()))
    tol, min_psnr = BARS[model.mode]
    bad = {k: v for k, v in worst.items() if not v < tol}
    print(case, model.mode, "worst taps:", sorted(worst.items(), key=lambda kv: -kv[1])[:8])
    assert not bad, bad
    psnr = O.psnr(out, torch.from_numpy(g["full/final"]))
    print(case, model.mode, "PSNR(hip, reference) =", psnr)
    assert psnr >= min_psnr
    if "png_u8" in g.files:
        arr = (out.squeeze(0).clamp(0, 1).permute(1, 2, 0).numpy() * 255.0).round().astype(np.uint8)
        diff = np.abs(arr.astype(np.int16) - g["png_u8"].astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-3


def test_against_oracle_odd_size(model, synth_sd):
    """A size no golden covers (ragged 37x29: reflect pad 11/3, DAT pad to 64, NAFNet pad 148->160, odd FFT width)."""
    from oracle import freqfusion_oracle as O
    lr = torch.from_numpy(np.random.default_rng(11).random((1, 3, 37, 29), dtype=np.float32))
    ref = O.forward(synth_sd, lr)
    out = model(lr.cuda()).cpu()
    tol, min_psnr = BARS[model.mode]
    print("odd size", model.mode, "max|d| =", (out - ref).abs().max().item(), "PSNR =", O.psnr(out, ref))
    assert (out - ref).abs().max().item() < 5 * tol
    assert O.psnr(out, ref) >= min_psnr
