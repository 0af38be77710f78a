"""Goldens at the BENCHMARKED sizes, generated from the IMPORTED reference (build container only).

    python tests/golden/make_golden_big.py tile256     # one 256x256 LR tile (bench.py's rank-0 tile), ~3 min
    python tests/golden/make_golden_big.py config3     # BASELINE config 3: reference io._tiled_forward(model, lr, 256, 32)
                                                       #   on one 510x339 1/f image (6 tiles), ~15-20 min
    python tests/golden/make_golden_big.py whole510    # the WHOLE-IMAGE branch (io.py:219-221) on the same 510x339 image: one
                                                       #   model(lr) call, 172 890 tokens, ~8-10 min (+ the oracle for its pinning)
    python tests/golden/make_golden_big.py b2          # a B=2 48x48 batch through the reference (batched-forward parity)
    python tests/golden/make_golden_big.py hooks48     # the cached-expert features of forward_all_with_hooks
    python tests/golden/make_golden_big.py precomp48   # the cached-mode forward (forward_with_precomputed, SURVEY 8f rank 1's forward
                                                       #   half) on those features + expert outputs, collaborative block live

A 1024x1024x3 fp32 output is 12.6 MB, so only data a test can check cheaply is stored:
  * for every tap (same names as make_golden.py): 4096 seeded samples + (mean, mean|x|, L2) over the whole tensor,
  * for `final` and each expert output: 65536 seeded samples and four 64x64 HR crops (three fixed + the worst-case
    corner), and for config 3 crops that straddle the blend seams,
  * the oracle-vs-reference deviation at this size (the oracle's pinning at 256x256).
Only these arrays are committed; the reference's code never leaves this container.
"""
import os
import sys
import json
import time
import contextlib
import io as _io

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from _ref_import import build_reference_model, install_shims  # noqa: E402
from make_golden import make_input, summarize, load_synth_into_reference, reference_taps, sample_idx  # noqa: E402
from oracle import freqfusion_oracle as O  # noqa: E402

NBIG = 65536
CROP = 64


def big_samples(name, t):
    f = t.detach().float().reshape(-1).numpy()
    import hashlib
    s = int.from_bytes(hashlib.sha256(("big/" + name).encode()).digest()[:4], "little")
    idx = np.random.default_rng(s).integers(0, f.size, size=NBIG)
    return idx.astype(np.int64), f[idx].astype(np.float32)


def crops_of(t, corners):
    return np.stack([t[0, :, y:y + CROP, x:x + CROP].numpy() for (y, x) in corners]).astype(np.float32)


def tile256(model, sd):
    lr = make_input("natural", 256, 256, 100)          # == bench.make_tile(100): the tile bench.py times on rank 0
    t0 = time.time()
    ref = reference_taps(model, lr)
    t_ref = time.time() - t0
    print(f"reference forward+taps at 256x256: {t_ref:.1f} s", flush=True)
    blob = {"lr": lr.numpy()}
    corners = [(0, 0), (480, 480), (960, 960), (100, 900)]
    blob["crop_corners"] = np.array(corners, dtype=np.int64)
    for k, v in ref.items():
        s = summarize(k, v)
        for kk, vv in s.items():
            blob[f"tap/{k}/{kk}"] = vv
    for k in ("final", "expert.hat", "expert.dat", "expert.nafnet"):
        i, v = big_samples(k, ref[k])
        blob[f"big/{k}/idx"], blob[f"big/{k}/val"] = i, v
        blob[f"crop/{k}"] = crops_of(ref[k], corners)
    np.savez_compressed(os.path.join(HERE, "t256_nat.npz"), **blob)
    print("wrote t256_nat.npz", flush=True)
    # pin the oracle at this size
    t0 = time.time()
    otaps = {}
    out = O.forward(sd, lr, otaps)
    otaps["final"] = out
    t_or = time.time() - t0
    dev = {k: float((otaps[k] - v).abs().max()) for k, v in ref.items() if k in otaps}
    flips = float(((otaps["fusion.gates"] - ref["fusion.gates"]).abs() > 1e-3).float().mean()) if "fusion.gates" in otaps else None
    rep = {"max_abs_dev": dev, "psnr_final": O.psnr(out, ref["final"]), "gates_frac_gt_1e-3": flips,
           "reference_seconds_8_threads": t_ref, "oracle_seconds_8_threads": t_or}
    print("oracle-vs-reference at 256x256:", sorted(dev.items(), key=lambda kv: -kv[1])[:6], "PSNR", rep["psnr_final"], flush=True)
    return rep


def config3(model, sd):
    install_shims()
    with contextlib.redirect_stdout(_io.StringIO()):
        import importlib
        plug = importlib.import_module("models.team29_FreqFusion.io")
    h, w = 339, 510
    lr = make_input("natural", h, w, 31)
    _orig = torch.cuda.empty_cache
    torch.cuda.empty_cache = lambda: None
    t0 = time.time()
    try:
        with torch.no_grad():
            out = plug._tiled_forward(model, lr, tile_size=256, overlap=32, scale=4, device="cpu")
    finally:
        torch.cuda.empty_cache = _orig
    dt = time.time() - t0
    print(f"reference _tiled_forward(510x339, 256, 32): {dt:.1f} s", flush=True)
    # tile origins: x in {0,224,254}, y in {0,83}; seams (HR) around x=896..1024, 1016..1152, y=332..1024
    corners = [(0, 0), (300, 880), (332, 1000), (600, 1100), (1292, 1976), (1000, 1500), (340, 20), (980, 940)]
    blob = {"lr": lr.numpy(), "crop_corners": np.array(corners, dtype=np.int64), "crops": crops_of(out, corners)}
    i, v = big_samples("config3", out)
    blob["big/idx"], blob["big/val"] = i, v
    f = out.reshape(-1).numpy().astype(np.float64)
    blob["stats"] = np.array([f.mean(), np.abs(f).mean(), np.sqrt((f ** 2).sum())])
    blob["shape"] = np.array(out.shape, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "config3_510x339.npz"), **blob)
    print("wrote config3_510x339.npz", flush=True)
    return {"reference_seconds_8_threads": dt, "shape": list(out.shape)}


def whole510(model, sd):
    """The reference's first choice for every image is ONE forward over the whole image (models/team29_FreqFusion/io.py:219-221);
    config3() above pins only the tiled fallback.  Same 510x339 1/f image (seed 31): reflect pad to 512x352 inside the experts,
    510/339-point FFT bands, channel attention over 172 890 tokens."""
    h, w = 339, 510
    lr = make_input("natural", h, w, 31)
    t0 = time.time()
    with torch.no_grad():
        out = model(lr)
    dt = time.time() - t0
    print(f"reference model(lr) on the whole 510x339 image: {dt:.1f} s", flush=True)
    corners = [(0, 0), (300, 880), (332, 1000), (600, 1100), (1292, 1976), (1000, 1500), (340, 20), (980, 940)]
    blob = {"lr": lr.numpy(), "crop_corners": np.array(corners, dtype=np.int64), "crops": crops_of(out, corners)}
    i, v = big_samples("whole510", out)
    blob["big/idx"], blob["big/val"] = i, v
    f = out.reshape(-1).numpy().astype(np.float64)
    blob["stats"] = np.array([f.mean(), np.abs(f).mean(), np.sqrt((f ** 2).sum())])
    blob["shape"] = np.array(out.shape, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "whole510_339.npz"), **blob)
    print("wrote whole510_339.npz", flush=True)
    rep = {"reference_seconds_8_threads": dt, "shape": list(out.shape)}
    if os.environ.get("FF_WHOLE510_ORACLE", "1") == "1":
        t0 = time.time()
        oout = O.forward(sd, lr)
        rep["oracle_seconds_8_threads"] = time.time() - t0
        rep["oracle_vs_reference_max_abs"] = float((oout - out).abs().max())
        rep["oracle_vs_reference_psnr"] = O.psnr(oout, out)
        print("oracle-vs-reference on the whole 510x339 image:", rep["oracle_vs_reference_max_abs"], rep["oracle_vs_reference_psnr"], flush=True)
    return rep


def b2(model, sd):
    lr = torch.cat([make_input("natural", 48, 48, 41), make_input("uniform", 48, 48, 42)], 0)
    with torch.no_grad():
        out = model(lr)
        one = torch.cat([model(lr[0:1]), model(lr[1:2])], 0)
    blob = {"lr": lr.numpy(), "out": out.numpy().astype(np.float32)}
    np.savez_compressed(os.path.join(HERE, "b2_48.npz"), **blob)
    return {"ref_batched_vs_single_max_abs": float((out - one).abs().max())}


def hooks48(model, sd):
    """ExpertEnsemble.forward_all_with_hooks (expert_loader.py:894-951) on the 48x48 uint8 case: the cached-expert features
    (conv_after_body outputs of HAT / DAT, the input of NAFNet's ending conv, each bilinearly resized to LR resolution)."""
    lr = make_input("u8", 48, 48, 0)
    ens = model.expert_ensemble
    with contextlib.redirect_stdout(_io.StringIO()):
        outputs, feats = ens.forward_all_with_hooks(lr)
    blob = {"lr": lr.numpy()}
    # (the SR outputs of the same case are in c48_u8.npz)
    for k, v in feats.items():
        blob["feat/" + k] = v.detach().numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "hooks48.npz"), **blob)
    return {k: list(v.shape) for k, v in feats.items()}


def precomp48(model, sd):
    """CompleteEnhancedFusionSR.forward_with_precomputed (enhanced_fusion.py:756-812) in eval mode on the 48x48 uint8 case: the
    expert outputs and hooked features are the ones c48_u8.npz / hooks48.npz hold (checked here); the collaborative block
    (large_kernel_attention.py:250-419), dead in the plain eval forward, gets the seeded `collab` weights."""
    from isr2_amd.weights import synth_state_dict
    from make_golden import SEED
    sdc = synth_state_dict(SEED, parts=("hat", "dat", "nafnet", "fusion", "collab"))
    res = model.load_state_dict(sdc, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys[:5]
    model.eval()
    lr = make_input("u8", 48, 48, 0)
    ens = model.expert_ensemble
    with contextlib.redirect_stdout(_io.StringIO()), torch.no_grad():
        outputs, feats = ens.forward_all_with_hooks(lr)
        final, inter = model.forward_with_precomputed(lr, outputs, feats, return_intermediates=True)
    g = np.load(os.path.join(HERE, "c48_u8.npz"))
    hk = np.load(os.path.join(HERE, "hooks48.npz"))
    for k in ("hat", "dat", "nafnet"):
        assert np.abs(outputs[k].numpy() - g[f"full/expert.{k}"]).max() < 1e-6, k       # same inputs as the committed fixtures
        assert np.abs(feats[k].numpy() - hk[f"feat/{k}"]).max() < 1e-6, k
    blob = {"final": final.numpy().astype(np.float32)}
    for k, v in inter["enhanced_outputs"].items():
        blob["enhanced/" + k] = v.numpy().astype(np.float32)
    # pin the oracle's restatement of the same call
    otaps = {}
    oout = O.forward_with_precomputed(sdc, lr, {k: v for k, v in outputs.items()}, {k: v for k, v in feats.items()}, otaps)
    for i in range(3):
        blob[f"mod{i}"] = otaps[f"collab.mod{i}"].numpy().astype(np.float32)
    rep = {"oracle_vs_reference_final_max_abs": float((oout - final).abs().max()),
           "oracle_vs_reference_enhanced_max_abs": {k: float((otaps[f"collab.out.{k}"] - inter["enhanced_outputs"][k]).abs().max()) for k in ("hat", "dat", "nafnet")},
           "gain_range": [float(min(blob[f"mod{i}"].min() for i in range(3))), float(max(blob[f"mod{i}"].max() for i in range(3)))],
           "plain_vs_collab_final_max_abs": float((final - torch.from_numpy(g["full/final"])).abs().max())}
    np.savez_compressed(os.path.join(HERE, "precomp48.npz"), **blob)
    print("precomp48:", rep, flush=True)
    return rep


def main():
    what = sys.argv[1:] or ["tile256"]
    torch.manual_seed(0)
    torch.set_num_threads(int(os.environ.get("FF_THREADS", "8")))
    model, ens = build_reference_model()
    sd = load_synth_into_reference(model)
    rp = os.path.join(HERE, "big_pinning_report.json")
    report = json.load(open(rp)) if os.path.exists(rp) else {}
    for wname in what:
        report[wname] = {"tile256": tile256, "config3": config3, "whole510": whole510, "b2": b2, "hooks48": hooks48, "precomp48": precomp48}[wname](model, sd)
        json.dump(report, open(rp, "w"), indent=1)


if __name__ == "__main__":
    main()
