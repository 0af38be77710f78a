"""End-to-end timings of the plugin path on BASELINE configs 3 and 4 (synthetic 1/f images, seeded synthetic weights):
   * config 3: DIV2K-val-sized 510x339 LR images -- main() (whole-image forward, PNG decode / encode included) and the
     256-tile / 32-overlap inference the config names (io._tiled_forward, device-resident), lanes on and off;
   * config 4: one 2040x1356 LR image through main() and through 256 / 32 tiles.
python tools/plugin_bench.py [n_images]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("FF_ALLOW_SYNTH", "1")
import bench  # noqa: E402
import models.team29_FreqFusion.io as plug  # noqa: E402
from isr2_amd.model import FreqFusionHIP  # noqa: E402
from isr2_amd.weights import synth_state_dict  # noqa: E402
from PIL import Image  # noqa: E402


def pink(h, w, seed):
    rng = np.random.default_rng(seed)
    f = np.fft.rfft2(rng.standard_normal((3, h, w)))
    fy, fx = np.fft.fftfreq(h)[:, None], np.fft.rfftfreq(w)[None, :]
    f /= np.maximum(np.sqrt(fy ** 2 + fx ** 2), 1.0 / max(h, w))
    x = np.fft.irfft2(f, s=(h, w))
    x = (x - x.min()) / (x.max() - x.min())
    return (x.transpose(1, 2, 0) * 255).astype(np.uint8)


def sync_time(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, r


n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as d:
    for name, (h, w), cnt in (("config 3 (510x339)", (339, 510), n), ("config 4 (2040x1356)", (1356, 2040), max(n // 2, 2))):
        src, dst = os.path.join(d, name[:8].replace(" ", "") + "_in"), os.path.join(d, name[:8].replace(" ", "") + "_out")
        os.makedirs(src)
        for i in range(cnt):
            Image.fromarray(pink(h, w, 100 + i)).save(os.path.join(src, f"im{i:03d}.png"))
        mp = 16 * h * w / 1e6
        plug.main(os.path.join(d, "no_ckpt.pth"), src, dst, dev)          # untimed: module load, allocator growth, page cache
        t1, _ = sync_time(lambda: plug.main(os.path.join(d, "no_ckpt.pth"), src, dst, dev))
        # the same images twice as many times: the difference is the steady-state cost per image (model build, first-shape
        # allocations and graph capture are paid once per main() call)
        for i in range(cnt):
            os.link(os.path.join(src, f"im{i:03d}.png"), os.path.join(src, f"jm{i:03d}.png"))
        t2, _ = sync_time(lambda: plug.main(os.path.join(d, "no_ckpt.pth"), src, dst, dev))
        per = (t2 - t1) / cnt
        print(f"{name}: main() over {cnt} images {t1:.2f} s, over {2 * cnt} images {t2:.2f} s -> {per:.3f} s per further image "
              f"= {mp / per:.2f} output MPix/s end to end (PNG decode + encode included), fixed cost {t1 - cnt * per:.1f} s", flush=True)
    model = FreqFusionHIP(synth_state_dict(1234), dev)
    for name, (h, w) in (("config 3 (510x339)", (339, 510)), ("config 4 (2040x1356)", (1356, 2040))):
        lr = torch.from_numpy(pink(h, w, 7)).to(dev).permute(2, 0, 1).unsqueeze(0).float() / 255
        mp = 16 * h * w / 1e6
        for lanes in ("1", "0"):
            os.environ["FF_TILE_LANES"] = lanes
            plug._tiled_forward(model, lr, tile_size=256, overlap=32, scale=4, device=dev)
            t, _ = sync_time(lambda: plug._tiled_forward(model, lr, tile_size=256, overlap=32, scale=4, device=dev))
            print(f"{name}: 256/32 tiles, device-resident, lanes={'2' if lanes == '1' else '1'}: {t * 1e3:.1f} ms = {mp / t:.2f} output MPix/s", flush=True)
        if h * w <= 600 * 400:
            model(lr)
            t, _ = sync_time(lambda: model.graphed(lr))
            t, _ = sync_time(lambda: model.graphed(lr))
            print(f"{name}: whole-image forward (graph replay): {t * 1e3:.1f} ms = {mp / t:.2f} output MPix/s", flush=True)
        else:
            model(lr)
            t, _ = sync_time(lambda: model(lr))
            print(f"{name}: whole-image forward (eager): {t * 1e3:.1f} ms = {mp / t:.2f} output MPix/s", flush=True)
