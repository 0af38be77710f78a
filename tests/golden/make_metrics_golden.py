"""PSNR / SSIM fixtures from the IMPORTED reference metrics (build container only).

    python tests/golden/make_metrics_golden.py      -> tests/golden/metrics.npz

Loads /root/reference/src/utils/metrics.py by file path (its package __init__ pulls tensorboard / lpips) and evaluates
rgb_to_y, calculate_psnr, calculate_ssim and calculate_psnr_ssim_batch (metrics.py:30-52, 76-126, 129-260, 263-290) on seeded
image pairs.  scikit-image is not installed here, so calculate_ssim takes the module's own torch path
(calculate_ssim_torch: 11x11 Gaussian sigma 1.5, zero padding, mean of the SSIM map); the skimage branch of the reference is
not pinned by these fixtures.  Only the arrays below are committed."""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_metrics", os.path.join(os.environ.get("FF_REFERENCE_ROOT", "/root/reference"), "src", "utils", "metrics.py"))
M = importlib.util.module_from_spec(spec)
spec.loader.exec_module(M)
assert not M.SKIMAGE_AVAILABLE


def main():
    rng = np.random.default_rng(2026)
    blob = {}
    cases = []
    for i, (h, w) in enumerate([(67, 53), (128, 96), (40, 40)]):
        hr = rng.random((3, h, w), dtype=np.float32)
        hr = (hr + np.roll(hr, 1, 1) + np.roll(hr, 1, 2)) / 3.0                       # some spatial correlation
        sr = hr + rng.normal(0, 0.03 * (i + 1), size=hr.shape).astype(np.float32)     # leaves [0,1] in places: exercises the clamp
        blob[f"sr{i}"], blob[f"hr{i}"] = sr, hr
        a, b = torch.from_numpy(sr), torch.from_numpy(hr)
        for crop in (0, 4):
            for ych in (False, True):
                cases.append((i, crop, int(ych), M.calculate_psnr(a, b, crop, ych), M.calculate_ssim(a, b, crop, ych)))
        blob[f"y{i}"] = M.rgb_to_y(a).numpy()
    blob["cases"] = np.array(cases, dtype=np.float64)                                 # rows: image, crop, y_channel, psnr, ssim
    sr_b = torch.from_numpy(np.stack([blob["sr2"], blob["hr2"] * 0.9 + 0.05]))
    hr_b = torch.from_numpy(np.stack([blob["hr2"], blob["hr2"]]))
    blob["batch"] = np.array(M.calculate_psnr_ssim_batch(sr_b, hr_b, 4, True), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **blob)
    print(blob["cases"], blob["batch"])


if __name__ == "__main__":
    main()
