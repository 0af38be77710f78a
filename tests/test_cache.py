"""Cached-expert feature format: files written by isr2_amd.cache against the reference's own reader
(CachedSRDataset, loaded by file path when /root/reference is present -- build container only), and the HIP extractor against
the reference's forward_all_with_hooks golden (tests/golden/hooks48.npz)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _fake_sample(seed=0, h=12, w=10):
    g = torch.Generator().manual_seed(seed)
    lr, hr = torch.rand(3, h, w, generator=g), torch.rand(3, 4 * h, 4 * w, generator=g)
    outs = {k: torch.rand(1, 3, 4 * h, 4 * w, generator=g) for k in ("hat", "dat", "nafnet")}
    feats = {"hat": torch.rand(1, 180, h, w, generator=g), "dat": torch.rand(1, 180, h, w, generator=g),
             "nafnet": torch.rand(1, 64, h, w, generator=g)}
    return lr, hr, outs, feats


def test_cache_roundtrip_and_layout(tmp_path):
    from isr2_amd import cache
    lr, hr, outs, feats = _fake_sample()
    p1, p2 = cache.write_sample(str(tmp_path), "img_001", lr, hr, outs, feats)
    assert os.path.basename(p1) == "img_001_hat_part.pt" and os.path.basename(p2) == "img_001_rest_part.pt"
    a = torch.load(p1, weights_only=True)
    b = torch.load(p2, weights_only=True)
    assert sorted(a) == ["features", "filename", "hr", "lr", "outputs"] and sorted(b) == ["features", "filename", "outputs"]
    assert list(a["outputs"]) == ["hat"] and sorted(b["outputs"]) == ["dat", "nafnet"]
    s = cache.read_sample(str(tmp_path), "img_001")
    assert torch.equal(s["lr"], lr) and torch.equal(s["hr"], hr) and s["filename"] == "img_001"
    for k in outs:
        assert torch.equal(s["expert_imgs"][k], outs[k][0]) and torch.equal(s["expert_feats"][k], feats[k][0])
    assert cache.list_stems(str(tmp_path)) == ["img_001"]
    with pytest.raises(KeyError):
        cache.write_sample(str(tmp_path), "bad", lr, hr, {"hat": outs["hat"]}, feats)


def test_reference_reader_loads_our_files(tmp_path):
    """The reference's CachedSRDataset (src/data/cached_dataset.py) reads files written by isr2_amd.cache: same keys, shapes
    and values.  Needs the reference tree (this container); skipped where it is absent (GPU box)."""
    ref = os.path.join(os.environ.get("FF_REFERENCE_ROOT", "/root/reference"), "src", "data", "cached_dataset.py")
    if not os.path.exists(ref):
        pytest.skip("reference tree not present")
    from isr2_amd import cache
    spec = importlib.util.spec_from_file_location("ref_cached_dataset", ref)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for i in range(2):
        lr, hr, outs, feats = _fake_sample(seed=i)
        cache.write_sample(str(tmp_path), f"img_{i:03d}", lr, hr, outs, feats)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        ds = mod.CachedSRDataset(str(tmp_path), augment=False, repeat_factor=1, load_features=True)
    assert len(ds) == 2
    item = ds[1]
    lr, hr, outs, feats = _fake_sample(seed=1)
    assert torch.equal(item["lr"], lr) and torch.equal(item["hr"], hr) and item["filename"] == "img_001"
    for k in ("hat", "dat", "nafnet"):
        assert torch.equal(item["expert_imgs"][k], outs[k][0]) and torch.equal(item["expert_feats"][k], feats[k][0])


def test_oracle_hook_features_match_reference_golden(synth_sd):
    from oracle import freqfusion_oracle as O
    g = np.load(os.path.join(HERE, "golden", "hooks48.npz"))
    _, feats = O.experts_forward_with_features(synth_sd, torch.from_numpy(g["lr"]))
    for k in ("hat", "dat", "nafnet"):
        ref = torch.from_numpy(g["feat/" + k])
        assert (feats[k] - ref).abs().max().item() < 5e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
def test_hip_extractor_matches_reference_hook_features(tmp_path, synth_sd):
    """model.experts_with_features (HIP) against the features the reference's forward_all_with_hooks captured on the same
    input (expert_loader.py:894-951), then through the extractor into cache files and back."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from isr2_amd import cache
    from isr2_amd.model import FreqFusionHIP
    g = np.load(os.path.join(HERE, "golden", "hooks48.npz"))
    gold = np.load(os.path.join(HERE, "golden", "c48_u8.npz"))
    model = FreqFusionHIP(synth_sd, "cuda:0")
    lr = torch.from_numpy(g["lr"])
    outs, feats = model.experts_with_features(lr.cuda())
    for k, c in (("hat", 180), ("dat", 180), ("nafnet", 64)):
        ref = torch.from_numpy(g["feat/" + k])
        assert tuple(feats[k].shape) == (1, c, 48, 48)
        assert (feats[k].cpu() - ref).abs().max().item() < 2e-4 * max(1.0, float(ref.abs().max())), k
        assert (outs[k].cpu() - torch.from_numpy(gold["full/expert." + k])).abs().max().item() < 2e-4
    assert cache.extract(model, [("a", lr, None), ("b", lr[0], torch.zeros(3, 192, 192))], str(tmp_path)) == 2
    s = cache.read_sample(str(tmp_path), "b")
    assert torch.equal(s["expert_feats"]["nafnet"], feats["nafnet"][0].cpu()) and tuple(s["expert_imgs"]["hat"].shape) == (3, 192, 192)
