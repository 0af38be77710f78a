"""Fixtures of the fusion-only TRAINING step (SURVEY 8f rank 1 / BASELINE config 5), generated from the IMPORTED reference
(build container only; the reference's code never leaves it -- only the arrays below are committed).

    python tests/golden/make_golden_train.py b2        # B=2, 16x16 LR  -> train_b2_16.npz     (seconds)
    python tests/golden/make_golden_train.py b4        # B=4, 32x32 LR  -> train_b4_32.npz     (~1 min)
    python tests/golden/make_golden_train.py b16       # B=16, 64x64 LR -> train_b16_64.npz    (config 5's shape; one step, minutes, ~30 GB)

What is run, on the reference's own modules (train.py:308-356 `train_epoch_cached`, accumulation_steps = 1):
    model = CompleteEnhancedFusionSR(expert_ensemble=None, **io.MODEL_CONFIG flags)      # cached mode, train.py:683-705
    model.train();  sr = model.forward_with_precomputed(lr, expert_imgs, expert_feats).clamp(0, 1)
    loss = mean |sr - hr|                    # CombinedLoss with the stage-1 weights {l1: 1.0} (configs/train_config.yaml:133-143,
                                             # perceptual_loss.py:86-105, :1235-1237); the class itself needs torchvision (absent)
    loss.backward(); clip_grad_norm_(model.parameters(), 1.0); AdamW(lr 1.5e-4, betas (0.9, 0.999), wd 1e-4, eps 1e-8).step();
    EMAModel(decay 0.9995).update(model)     # src/utils/checkpoint_manager.py:400-407
with the seeded synthetic fusion + collaborative weights (isr2_amd.weights.synth_state_dict) and seeded inputs
(train_inputs.make_train_batch).  The two nn.MultiheadAttention modules' `dropout` attribute is set to 0.0 on the instantiated
model (no reference file is touched), so the results do not depend on torch's RNG stream: dropout-on is "parity unpinned (RNG
stream)" -- the build draws its own counter-based masks.

Stored per case: sr, loss, pre-clip gradient norm; every BatchNorm's running_mean / running_var / num_batches_tracked after the
first forward; for EVERY parameter that received a gradient its L2 norm and either the whole gradient (numel <= 8192) or 256
seeded samples; the same for parameters and EMA shadow after 3 optimizer steps; the names of parameters without gradient.
"""
import os
import sys
import json
import time
import hashlib
import contextlib
import io as _io

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from _ref_import import install_shims  # noqa: E402
from train_inputs import make_train_batch  # noqa: E402
from isr2_amd.weights import synth_state_dict  # noqa: E402

SEED = 1234
FULL_MAX = 8192
NS = 256
HP = dict(lr=1.5e-4, betas=(0.9, 0.999), weight_decay=1.0e-4, eps=1.0e-8, clip=1.0, ema_decay=0.9995)   # configs/train_config.yaml:102-125
CASES = {"b2": (2, 16, 16, 501, "train_b2_16.npz", 3), "b4": (4, 32, 32, 502, "train_b4_32.npz", 3), "b16": (16, 64, 64, 503, "train_b16_64.npz", 1)}


def build_cached_mode_model():
    install_shims()
    with contextlib.redirect_stdout(_io.StringIO()):
        import importlib
        from src.models.enhanced_fusion import CompleteEnhancedFusionSR
        plug = importlib.import_module("models.team29_FreqFusion.io")
        cfg = plug.MODEL_CONFIG
        model = CompleteEnhancedFusionSR(
            expert_ensemble=None, num_experts=cfg["num_experts"], num_bands=cfg["num_bands"], block_size=cfg["block_size"],
            upscale=cfg["scale"], fusion_dim=cfg["fusion_dim"], num_heads=cfg["num_heads"], refine_depth=cfg["refine_depth"],
            refine_channels=cfg["refine_channels"], enable_hierarchical=True, enable_multi_domain_freq=True, enable_lka=True,
            enable_edge_enhance=True, enable_dynamic_selection=True, enable_cross_band_attn=True, enable_adaptive_bands=True,
            enable_multi_resolution=True, enable_collaborative=True)
    sd = synth_state_dict(SEED, parts=("fusion", "collab"))
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys[:5]
    # the only state the synthetic dict does not carry: parameters that are dead on this path and BatchNorm step counters
    live_missing = [k for k in res.missing_keys if not k.endswith("num_batches_tracked")]
    model.cross_band_attn.band_attention.dropout = 0.0
    model.collaborative.cross_attn.dropout = 0.0
    return model, sd, live_missing


def sidx(name, numel, n=NS):
    s = int.from_bytes(hashlib.sha256(("train/" + name).encode()).digest()[:4], "little")
    return np.random.default_rng(s).integers(0, numel, size=n).astype(np.int64)


def put(blob, prefix, name, t):
    f = t.detach().float().reshape(-1).numpy()
    blob[f"{prefix}/{name}/l2"] = np.array(np.sqrt((f.astype(np.float64) ** 2).sum()))
    if f.size <= FULL_MAX:
        blob[f"{prefix}/{name}/full"] = f.astype(np.float32)
    else:
        idx = sidx(name, f.size)
        blob[f"{prefix}/{name}/idx"] = idx
        blob[f"{prefix}/{name}/val"] = f[idx].astype(np.float32)


def run_case(tag):
    B, h, w, seed, fname, steps = CASES[tag]
    torch.manual_seed(0)
    model, sd, live_missing = build_cached_mode_model()
    model.train()
    d = {k: torch.from_numpy(v) for k, v in make_train_batch(seed, B, h, w).items()}
    outs = {k: d["out_" + k] for k in ("hat", "dat", "nafnet")}
    feats = {k: d["feat_" + k] for k in ("hat", "dat", "nafnet")}
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    opt = torch.optim.AdamW(model.parameters(), lr=HP["lr"], betas=HP["betas"], weight_decay=HP["weight_decay"], eps=HP["eps"])
    from src.utils.checkpoint_manager import EMAModel
    ema = EMAModel(model, decay=HP["ema_decay"])
    blob = {"meta": np.array(json.dumps({"B": B, "h": h, "w": w, "input_seed": seed, "weight_seed": SEED, "steps": steps, "hp": HP,
                                         "dropout": 0.0, "dead_parameters_not_in_synth": live_missing}))}
    losses, norms = [], []
    for step in range(steps):
        t0 = time.time()
        opt.zero_grad()
        sr = model.forward_with_precomputed(d["lr"], outs, feats)
        sr = sr.clamp(0, 1)
        loss = (sr - d["hr"]).abs().mean()
        loss.backward()
        if step == 0:
            blob["sr"] = sr.detach().numpy().astype(np.float32) if sr.numel() <= (1 << 20) else np.zeros(0, np.float32)
            if sr.numel() > (1 << 20):
                idx = sidx("sr", sr.numel(), 65536)
                blob["sr/idx"], blob["sr/val"] = idx, sr.detach().reshape(-1).numpy()[idx].astype(np.float32)
            nograd = []
            for n, p in named:
                if p.grad is None:
                    nograd.append(n)
                else:
                    put(blob, "grad", n, p.grad)
            blob["nograd"] = np.array(json.dumps(nograd))
            for n, b in model.named_buffers():
                if n.endswith(("running_mean", "running_var", "num_batches_tracked")):
                    blob["bn/" + n] = b.detach().numpy().copy()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), HP["clip"])
        opt.step()
        ema.update(model)
        losses.append(float(loss))
        norms.append(float(gn))
        print(f"{tag} step {step}: loss {float(loss):.6f}  grad norm {float(gn):.6f}  ({time.time() - t0:.1f} s)", flush=True)
    blob["losses"] = np.array(losses, dtype=np.float64)
    blob["grad_norms"] = np.array(norms, dtype=np.float64)
    for n, p in named:
        put(blob, f"param{steps}", n, p.data)
        put(blob, f"ema{steps}", n, ema.shadow[n])
    for n, b in model.named_buffers():
        if n.endswith(("running_mean", "running_var", "num_batches_tracked")):
            blob[f"bn{steps}/" + n] = b.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, fname), **blob)
    print("wrote", fname, os.path.getsize(os.path.join(HERE, fname)) // 1024, "KiB;", len(named), "trainable tensors,",
          sum(p.numel() for _, p in named), "values;", len(json.loads(str(blob["nograd"]))), "without gradient", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("FF_THREADS", "8")))
    for t in (sys.argv[1:] or ["b2"]):
        run_case(t)
