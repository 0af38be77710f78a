"""PSNR / SSIM on the device -- same names, arguments and conventions as the reference's
src/utils/metrics.py: rgb_to_y (:30-52), calculate_psnr (:76-126), calculate_ssim (:193-260, its torch path :129-190: 11x11
Gaussian, sigma 1.5, zero padding), calculate_psnr_ssim_batch (:263-290; Y channel, crop 4 by default).
Images are CUDA(HIP) float32 tensors [B,C,H,W] or [C,H,W] in [0,1]; nothing but the final scalar crosses PCIe.
(The reference switches to scikit-image's structural_similarity when that package is installed -- a different window; that
branch is not reproduced here and is not pinned by the fixtures: scikit-image is absent from this environment.)"""
from __future__ import annotations

import math
from typing import Tuple

import torch

from . import lib as _lib

T = torch.Tensor
_GAUSS = {}


def _gauss11(dev) -> T:
    g = _GAUSS.get(str(dev))
    if g is None:
        w = torch.tensor([math.exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)])      # metrics.py:153-157
        g = _GAUSS[str(dev)] = (w / w.sum()).float().to(dev)
    return g


def _prep(img1: T, img2: T):
    if img1.shape != img2.shape:
        raise _lib.FFError(f"Image shapes must match: {tuple(img1.shape)} vs {tuple(img2.shape)}")
    for t in (img1, img2):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
            raise _lib.FFError("metrics: expected CUDA(HIP) float32 tensors")
    if img1.dim() == 3:
        img1, img2 = img1.unsqueeze(0), img2.unsqueeze(0)
    if img1.dim() != 4 or img1.shape[1] not in (1, 3):
        raise _lib.FFError("metrics: expected [B,C,H,W] or [C,H,W] with C = 1 or 3")
    return img1.contiguous(), img2.contiguous()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def rgb_to_y(img: T) -> T:
    """ITU-R BT.601 luma in [16/255, 235/255] (metrics.py:30-52); host-side helper for tests (the kernels convert on the fly)."""
    r, g, b = (img[0:1], img[1:2], img[2:3]) if img.dim() == 3 else (img[:, 0:1], img[:, 1:2], img[:, 2:3])
    return (65.481 * r + 128.553 * g + 24.966 * b + 16.0) / 255.0


def _per_image(fn, img1: T, img2: T, crop_border: int, test_y_channel: bool, extra=()):
    L = _lib.load()
    B, C, H, W = img1.shape
    if H <= 2 * crop_border or W <= 2 * crop_border:
        raise _lib.FFError("metrics: crop_border removes the whole image")
    nwork = int(L.ff_metric_workspace(C, H, W, crop_border))
    work = torch.empty(nwork, device=img1.device, dtype=torch.float64)
    out = torch.empty(B, device=img1.device, dtype=torch.float64)
    for b in range(B):
        _lib.check(fn(img1[b].data_ptr(), img2[b].data_ptr(), C, H, W, int(crop_border), int(bool(test_y_channel)), *extra,
                      work.data_ptr(), nwork, out[b:b + 1].data_ptr(), _stream()))
    return out


def mse_per_image(img1: T, img2: T, crop_border: int = 0, test_y_channel: bool = False) -> T:
    """Device tensor [B] (float64) of mean squared errors -- stays on the GPU."""
    img1, img2 = _prep(img1, img2)
    return _per_image(_lib.load().ff_psnr_mse, img1, img2, crop_border, test_y_channel)


def calculate_psnr(img1: T, img2: T, crop_border: int = 0, test_y_channel: bool = False) -> float:
    """metrics.py:76-126 -- one MSE over the whole (batched) tensor, inf below 1e-10."""
    mse = float(mse_per_image(img1, img2, crop_border, test_y_channel).mean().item())
    return float("inf") if mse < 1e-10 else 10 * math.log10(1.0 / mse)


def ssim_per_image(img1: T, img2: T, crop_border: int = 0, test_y_channel: bool = False) -> T:
    img1, img2 = _prep(img1, img2)
    return _per_image(_lib.load().ff_ssim_mean, img1, img2, crop_border, test_y_channel, extra=(_gauss11(img1.device).data_ptr(),))


def calculate_ssim(img1: T, img2: T, crop_border: int = 0, test_y_channel: bool = False) -> float:
    """metrics.py:193-260 (torch path): mean of the SSIM map over batch, channels and pixels."""
    return float(ssim_per_image(img1, img2, crop_border, test_y_channel).mean().item())


def calculate_psnr_ssim_batch(sr_images: T, hr_images: T, crop_border: int = 4, test_y_channel: bool = True) -> Tuple[float, float]:
    """metrics.py:263-290: per-image PSNR (infinite ones skipped) and SSIM, averaged."""
    mse = mse_per_image(sr_images, hr_images, crop_border, test_y_channel).cpu().tolist()
    ssim = ssim_per_image(sr_images, hr_images, crop_border, test_y_channel).cpu().tolist()
    ps = [10 * math.log10(1.0 / m) for m in mse if m >= 1e-10]
    return (sum(ps) / len(ps) if ps else float("inf")), sum(ssim) / len(ssim)
