"""The plugin's checkpoint loader against real-format FILES (VERDICT r2 missing #3): what a user of the drop-in does first.

Restated contracts (the tests write files the way the upstream projects publish them, then load them through the plugin):
  * expert checkpoints: BasicSR containers `params_ema` / `params` / `state_dict` / `model` / a raw state dict, `module.`
    (DDP) prefixes removed, tensors copied where name AND shape match (reference src/models/expert_loader.py:127-157);
  * NAFNet-SIDD: bare backbone keys (`intro.weight`, `encoders.0.0.conv1.weight`, ...) loaded into the inner `.nafnet`
    (reference src/models/nafnet/__init__.py:84-115, expert_loader.py:494-515);
  * fusion checkpoint: `model_state_dict` of the whole CompleteEnhancedFusionSR (or a raw dict), `module.` / `model.` prefixes
    stripped, name + shape match, everything else ignored (reference models/team29_FreqFusion/io.py:164-177).
The CPU part checks `_build_state_dict` returns exactly the tensors written; the GPU part runs `main()` from such files and
compares with the model built directly from the same tensors.
"""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

FILE_SEED = 4321          # != the plugin's fallback seed (1234): a tensor that was NOT loaded from the file is detected


@pytest.fixture(scope="module")
def file_sd():
    from isr2_amd.weights import synth_state_dict
    return synth_state_dict(FILE_SEED)


def _write_checkpoints(root, sd, hat_fmt="params_ema", dat_fmt="params", naf_fmt="raw", fusion_fmt="model_state_dict",
                       fusion_prefix="", with_junk=True):
    """pretrained/{hat,dat,nafnet}/<published file names> + fusion.pth under `root`; returns (pretrained_dir, fusion_path)."""
    from isr2_amd.weights import HAT_PREFIX, DAT_PREFIX, NAF_PREFIX
    import models.team29_FreqFusion.io as plug
    pre = os.path.join(root, "pretrained")

    def container(fmt, d):
        if fmt == "raw":
            return d
        return {fmt: d, "iter": 800000} if fmt != "params_ema" else {"params": OrderedDict((k, torch.zeros_like(v)) for k, v in d.items()), "params_ema": d}

    for name, prefix, fmt, ddp in (("hat", HAT_PREFIX, hat_fmt, True), ("dat", DAT_PREFIX, dat_fmt, False), ("nafnet", NAF_PREFIX, naf_fmt, False)):
        sub, fname, _ = plug.EXPERT_FILES[name]
        os.makedirs(os.path.join(pre, sub), exist_ok=True)
        d = OrderedDict((("module." if ddp else "") + k[len(prefix):], v.clone()) for k, v in sd.items() if k.startswith(prefix))
        assert len(d) > 50, name
        if with_junk:
            d["not_in_the_model.weight"] = torch.ones(3)                              # ignored: no such name
            first = next(iter(d))
            d["shape_mismatch_probe"] = d[first]
        torch.save(container(fmt, d), os.path.join(pre, sub, fname))
    fus = OrderedDict((fusion_prefix + k, v.clone()) for k, v in sd.items() if not k.startswith("expert_ensemble."))
    if with_junk:
        fus[fusion_prefix + "collaborative.some_training_only_tensor"] = torch.zeros(7)  # ignored
        k0 = next(k for k in fus if k.endswith("refine_net.0.weight"))
        fus[k0 + "_wrong_shape"] = fus[k0][:1]
    path = os.path.join(root, "fusion_best.pth")
    if fusion_fmt == "model_state_dict":
        torch.save({"epoch": 50, "model_state_dict": fus, "optimizer_state_dict": {"state": {}, "param_groups": []},
                    "ema_state_dict": {"shadow": {}, "decay": 0.9995}, "metrics": {"psnr": 30.05}}, path)
    else:
        torch.save(fus, path)
    return pre, path


@pytest.mark.parametrize("fmts", [
    dict(hat_fmt="params_ema", dat_fmt="params", naf_fmt="raw", fusion_fmt="model_state_dict", fusion_prefix="module."),
    dict(hat_fmt="state_dict", dat_fmt="model", naf_fmt="params", fusion_fmt="raw", fusion_prefix="model."),
])
def test_build_state_dict_reads_every_container_format(tmp_path, file_sd, fmts):
    import models.team29_FreqFusion.io as plug
    pre, fusion = _write_checkpoints(str(tmp_path), file_sd, **fmts)
    sd = plug._build_state_dict(fusion, pre, verbose=False)
    assert set(sd) == set(file_sd)
    wrong = [k for k, v in file_sd.items() if not torch.equal(sd[k], v)]
    assert not wrong, wrong[:5]


def test_shape_mismatch_and_unknown_keys_keep_the_fallback(tmp_path, file_sd):
    """A tensor whose shape differs is skipped (the model keeps its initial value, reference expert_loader.py:148-153); here the
    initial value is the seeded stand-in for the reference's random init."""
    import models.team29_FreqFusion.io as plug
    from isr2_amd.weights import synth_state_dict, HAT_PREFIX
    pre, fusion = _write_checkpoints(str(tmp_path), file_sd, with_junk=False)
    sub, fname, _ = plug.EXPERT_FILES["hat"]
    p = os.path.join(pre, sub, fname)
    ck = torch.load(p, weights_only=True)
    key = "module.conv_first.weight"
    ck["params_ema"][key] = ck["params_ema"][key][:, :, :1, :1].clone()          # wrong shape -> skipped
    torch.save(ck, p)
    sd = plug._build_state_dict(fusion, pre, verbose=False)
    base = synth_state_dict(plug.SYNTH_SEED)
    assert torch.equal(sd[HAT_PREFIX + "conv_first.weight"], base[HAT_PREFIX + "conv_first.weight"])
    assert torch.equal(sd[HAT_PREFIX + "conv_first.bias"], file_sd[HAT_PREFIX + "conv_first.bias"])


def test_fusion_checkpoint_is_mandatory_and_must_match(tmp_path, file_sd, monkeypatch):
    import models.team29_FreqFusion.io as plug
    monkeypatch.setenv("FF_ALLOW_SYNTH", "0")
    pre, fusion = _write_checkpoints(str(tmp_path), file_sd, with_junk=False)
    with pytest.raises(FileNotFoundError):
        plug._build_state_dict(os.path.join(str(tmp_path), "absent.pth"), pre, verbose=False)
    bad = os.path.join(str(tmp_path), "other_model.pth")
    torch.save({"model_state_dict": {"backbone.weight": torch.zeros(3)}}, bad)
    with pytest.raises(RuntimeError):
        plug._build_state_dict(bad, pre, verbose=False)


@pytest.mark.gpu
def test_main_from_real_format_files_equals_direct_model(tmp_path, file_sd, monkeypatch):
    """main() fed from checkpoint FILES == FreqFusionHIP built from the same tensors (bit-equal PNG bytes)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    import models.team29_FreqFusion.io as plug
    from isr2_amd import ops
    from isr2_amd.model import FreqFusionHIP
    pre, fusion = _write_checkpoints(str(tmp_path), file_sd, fusion_prefix="module.")
    monkeypatch.setenv("FREQFUSION_PRETRAINED", pre)
    monkeypatch.setenv("FF_ALLOW_SYNTH", "0")
    monkeypatch.setenv("FF_PNG_LEVEL", "6")
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    lr_u8 = np.random.default_rng(3).integers(0, 256, (40, 48, 3), dtype=np.uint8)
    Image.fromarray(lr_u8).save(src / "img.png")
    ops.set_gemm_mode("bf16x3")
    plug.main(model_dir=fusion, input_path=str(src), output_path=str(dst), device=torch.device("cuda:0"))
    got = np.array(Image.open(dst / "img.png").convert("RGB"))
    model = FreqFusionHIP(file_sd, "cuda:0")
    lr = plug._load_image(str(src / "img.png"), torch.device("cuda:0"))
    want = ops.f32_to_u8_image(model(lr)).cpu().numpy()
    assert got.shape == want.shape == (160, 192, 3)
    assert np.array_equal(got, want)
    # and it is NOT what the fallback weights give (the files were really read)
    base = FreqFusionHIP(__import__("isr2_amd.weights", fromlist=["x"]).synth_state_dict(plug.SYNTH_SEED), "cuda:0")
    other = ops.f32_to_u8_image(base(lr)).cpu().numpy()
    assert not np.array_equal(got, other)
