"""Pin the CPU oracle against fixtures produced by the imported reference (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import freqfusion_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 2e-4   # fp32 re-association across ~150 residual blocks; observed <= 2e-5


@pytest.fixture(scope="module", params=["c48_u8", "c42x52_nat"])
def case(request, synth_sd):
    g = np.load(os.path.join(HERE, "golden", request.param + ".npz"))
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    taps = {}
    out = O.forward(synth_sd, torch.from_numpy(g["lr"]), taps)
    taps["final"] = out
    return g, taps


def test_final_and_experts_match_reference(case):
    g, taps = case
    for k in ("final", "expert.hat", "expert.dat", "expert.nafnet"):
        ref = torch.from_numpy(g["full/" + k])
        assert (taps[k] - ref).abs().max().item() < TOL, k
    assert O.psnr(taps["final"], torch.from_numpy(g["full/final"])) > 100.0


def test_intermediate_taps_match_reference(case):
    g, taps = case
    names = sorted({k.split("/")[1] for k in g.files if k.startswith("tap/")})
    assert len(names) > 40
    for n in names:
        t = taps[n].reshape(-1)
        idx = torch.from_numpy(g[f"tap/{n}/idx"])
        ref = torch.from_numpy(g[f"tap/{n}/val"])
        scale = max(1.0, float(ref.abs().max()))
        assert (t[idx] - ref).abs().max().item() < TOL * scale, n
        l2 = float(g[f"tap/{n}/stats"][2])
        assert abs(float(t.double().norm()) - l2) <= 1e-4 * max(l2, 1.0), n


def test_png_bytes_config1(synth_sd):
    g = np.load(os.path.join(HERE, "golden", "c48_u8.npz"))
    out = O.forward(synth_sd, torch.from_numpy(g["lr"]))
    arr = (out.squeeze(0).clamp(0, 1).permute(1, 2, 0).numpy() * 255.0).round().astype(np.uint8)
    diff = np.abs(arr.astype(np.int16) - g["png_u8"].astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3


def test_tiled_forward_matches_reference():
    g = np.load(os.path.join(HERE, "golden", "tiled_37x53.npz"))

    def standin(t):
        up = torch.nn.functional.interpolate(t, scale_factor=4, mode="bilinear", align_corners=False)
        return up * 0.9 + 0.05 * t.mean()

    out = O.tiled_forward(standin, torch.from_numpy(g["lr"]), tile=16, overlap=4, scale=4)
    assert (out - torch.from_numpy(g["out"])).abs().max().item() < 1e-6
