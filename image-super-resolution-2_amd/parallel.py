"""Multi-GPU sharding of the path: one process per GPU, work items (images / tiles) are independent, so the
only collective is the one-off broadcast of the frozen weights (RCCL over xGMI on the GPU node; the same
code runs over gloo on CPU tensors in the tests).  Mirrors how the reference shards a file list over
workers (eval.py:166-170: contiguous ranges) -- there is no per-image communication to mirror.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

T = torch.Tensor


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of rank's share; the first n_items % world ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(max(n_items, 0), world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def shard_list(items: Sequence, rank: int, world: int) -> List:
    b, e = shard_range(len(items), rank, world)
    return list(items[b:e])


def flat_size(spec) -> int:
    return sum(int(np.prod(s)) if len(s) else 1 for _, s, _ in spec)


def broadcast_state_dict(sd_on_src, spec, rank: int, world: int, device, src: int = 0) -> "OrderedDict[str, T]":
    """Every rank returns the state dict of `spec` (name, shape, kind) holding rank `src`'s values.
    ONE flat fp32 broadcast (690 MB for the full model): per-tensor collectives would be latency bound."""
    import torch.distributed as dist
    total = flat_size(spec)
    flat = torch.empty(total, device=device, dtype=torch.float32)
    if rank == src:
        off = 0
        for n, s, _ in spec:
            k = int(np.prod(s)) if len(s) else 1
            flat[off:off + k].copy_(sd_on_src[n].reshape(-1).to(device, torch.float32))
            off += k
    if world > 1:
        dist.broadcast(flat, src=src)
    out: "OrderedDict[str, T]" = OrderedDict()
    off = 0
    for n, s, _ in spec:
        k = int(np.prod(s)) if len(s) else 1
        out[n] = flat[off:off + k].reshape(tuple(s))
        off += k
    return out
