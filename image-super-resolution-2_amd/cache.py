"""Cached-expert feature files: the on-disk contract of the reference's CachedSRDataset
(src/data/cached_dataset.py:8-25 layout, :135-200 reader) and an extractor that fills it from the HIP experts.

    {cache_dir}/{stem}_hat_part.pt    {'outputs': {'hat': [1,3,4h,4w]}, 'features': {'hat': [1,180,h,w]},
                                       'lr': [3,h,w], 'hr': [3,4h,4w], 'filename': stem}
    {cache_dir}/{stem}_rest_part.pt   {'outputs': {'dat', 'nafnet'}, 'features': {'dat': [1,180,h,w], 'nafnet': [1,64,h,w]},
                                       'filename': stem}

Plain `torch.save` dicts of CPU float32 tensors: the reference's reader loads them unchanged (its `_normalize_keys` passes
hat / dat / nafnet through, `:160-175` squeezes the batch dimension).  The frozen experts are the expensive 95 % of the path;
writing their outputs once is what lets the fusion-only training step (8f rank 1) skip them.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional, Tuple

import torch

T = torch.Tensor
PRIMARY_SUFFIX, REST_SUFFIX = "_hat_part.pt", "_rest_part.pt"


def write_sample(cache_dir: str, stem: str, lr: T, hr: T, outputs: Dict[str, T], features: Dict[str, T]) -> Tuple[str, str]:
    """One training sample -> the two part files.  lr [3,h,w] / hr [3,4h,4w] (or with a leading batch dim of 1)."""
    os.makedirs(cache_dir, exist_ok=True)
    cpu = lambda t: t.detach().to("cpu", torch.float32).contiguous()      # noqa: E731
    sq = lambda t: t.squeeze(0) if t.dim() == 4 else t                    # noqa: E731
    for k in ("hat", "dat", "nafnet"):
        if k not in outputs or k not in features:
            raise KeyError(f"cache sample {stem}: missing expert {k}")
    p1 = os.path.join(cache_dir, stem + PRIMARY_SUFFIX)
    p2 = os.path.join(cache_dir, stem + REST_SUFFIX)
    torch.save({"outputs": {"hat": cpu(outputs["hat"])}, "features": {"hat": cpu(features["hat"])},
                "lr": cpu(sq(lr)), "hr": cpu(sq(hr)), "filename": stem}, p1)
    torch.save({"outputs": {k: cpu(outputs[k]) for k in ("dat", "nafnet")},
                "features": {k: cpu(features[k]) for k in ("dat", "nafnet")}, "filename": stem}, p2)
    return p1, p2


def read_sample(cache_dir: str, stem: str, load_features: bool = True) -> dict:
    """What CachedSRDataset.__getitem__ returns without augmentation (cached_dataset.py:135-200): lr, hr, expert_imgs
    (batch dim squeezed), expert_feats, filename.  Tensors only: loaded with weights_only=True."""
    a = torch.load(os.path.join(cache_dir, stem + PRIMARY_SUFFIX), weights_only=True)
    b = torch.load(os.path.join(cache_dir, stem + REST_SUFFIX), weights_only=True)
    sq = lambda t: t.squeeze(0) if t.dim() == 4 else t                    # noqa: E731
    imgs = {k: sq(v) for d in (a["outputs"], b["outputs"]) for k, v in d.items()}
    res = {"lr": a["lr"], "hr": a["hr"], "expert_imgs": imgs, "filename": stem}
    if load_features:
        res["expert_feats"] = {k: sq(v) for d in (a.get("features", {}), b.get("features", {})) for k, v in d.items()}
    return res


def list_stems(cache_dir: str):
    return sorted(f[:-len(PRIMARY_SUFFIX)] for f in os.listdir(cache_dir) if f.endswith(PRIMARY_SUFFIX)
                  and os.path.exists(os.path.join(cache_dir, f[:-len(PRIMARY_SUFFIX)] + REST_SUFFIX)))


@torch.no_grad()
def extract(model, samples: Iterable[Tuple[str, T, Optional[T]]], cache_dir: str) -> int:
    """Run the HIP experts (model.experts_with_features = the reference's forward_all_with_hooks) over (stem, lr [1,3,h,w] or
    [3,h,w], hr or None) samples and write the cache files.  Returns the number of samples written."""
    n = 0
    for stem, lr, hr in samples:
        lr4 = lr if lr.dim() == 4 else lr.unsqueeze(0)
        outs, feats = model.experts_with_features(lr4)
        if hr is None:
            hr = torch.zeros(3, 4 * lr4.shape[-2], 4 * lr4.shape[-1])
        write_sample(cache_dir, stem, lr4, hr, outs, feats)
        n += 1
    return n
