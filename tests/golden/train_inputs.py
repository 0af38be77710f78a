"""Seeded inputs of the fusion-only training step (BASELINE config 5), shared by the fixture generator
(make_golden_train.py), the tests and bench.py's training side block.  Pure numpy element-wise arithmetic on Philox
streams, so every machine regenerates the same bits and only the OUTPUTS need to be stored as fixtures.

    lr     [B,3,h,w]      1/f-spectrum noise clipped to [0,1] (the natural-statistics tile of SURVEY 8d)
    hr     [B,3,4h,4w]    nearest x4 of lr plus detail noise, clipped
    outs   {hat,dat,nafnet: [B,3,4h,4w]}   hr plus per-expert noise, clipped (what CachedSRDataset's expert_imgs hold)
    feats  {hat [B,180,h,w], dat [B,180,h,w], nafnet [B,64,h,w]}   N(0, 0.5) (the cached hook features, cached_dataset.py:135-200)
"""
import numpy as np


def _natural(rng, h, w):
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.fftfreq(w)[None, :]
    amp = 1.0 / np.maximum(np.sqrt(fy ** 2 + fx ** 2), 1.0 / max(h, w))
    out = []
    for _ in range(3):
        ph = rng.random((h, w)) * 2 * np.pi
        img = np.real(np.fft.ifft2(amp * np.exp(1j * ph)))
        img = (img - img.mean()) / (img.std() + 1e-8) * 0.2 + 0.5
        out.append(np.clip(img, 0, 1))
    return np.stack(out)


def make_train_batch(seed: int, B: int, h: int, w: int):
    """-> dict of float32 numpy arrays: lr, hr, out_hat, out_dat, out_nafnet, feat_hat, feat_dat, feat_nafnet."""
    rng = np.random.Generator(np.random.Philox(seed))
    # the FFT-based 1/f image is rounded to 1/4096 so that libm / FFT differences between machines cannot reach the stored bits
    lr = np.stack([_natural(rng, h, w) for _ in range(B)])
    lr = (np.round(lr * 4096.0) / 4096.0).astype(np.float32)
    up = lr.repeat(4, axis=2).repeat(4, axis=3)
    hr = np.clip(up + 0.04 * rng.standard_normal(up.shape, dtype=np.float32), 0.0, 1.0).astype(np.float32)
    d = {"lr": lr, "hr": hr}
    for k, s in (("hat", 0.02), ("dat", 0.03), ("nafnet", 0.05)):
        d["out_" + k] = np.clip(hr + s * rng.standard_normal(up.shape, dtype=np.float32), 0.0, 1.0).astype(np.float32)
    for k, c in (("hat", 180), ("dat", 180), ("nafnet", 64)):
        d["feat_" + k] = (0.5 * rng.standard_normal((B, c, h, w), dtype=np.float32)).astype(np.float32)
    return d
