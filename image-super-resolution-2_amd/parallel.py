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


ALIGN = 64          # floats: every tensor starts on a 256-byte boundary of the flat buffer (kernels need 16-byte aligned rows)


def _numel(shape) -> int:
    return int(np.prod(shape)) if len(shape) else 1


def _offsets(spec):
    offs, off = [], 0
    for _, s, _ in spec:
        offs.append(off)
        off += (_numel(s) + ALIGN - 1) // ALIGN * ALIGN
    return offs, off


def flat_size(spec) -> int:
    """Floats in the broadcast buffer (each tensor padded to a multiple of ALIGN)."""
    return _offsets(spec)[1]


def broadcast_state_dict(sd_on_src, spec, rank: int, world: int, device, src: int = 0) -> "OrderedDict[str, T]":
    """Every rank returns the state dict of `spec` (name, shape, kind) holding rank `src`'s values.
    ONE flat fp32 broadcast (690 MB for the full model): per-tensor collectives would be latency bound.
    The returned tensors are views into the flat buffer, each 256-byte aligned."""
    import torch.distributed as dist
    offs, total = _offsets(spec)
    flat = torch.zeros(total, device=device, dtype=torch.float32)
    if rank == src:
        for (n, s, _), off in zip(spec, offs):
            flat[off:off + _numel(s)].copy_(sd_on_src[n].reshape(-1).to(device, torch.float32))
    if world > 1:
        dist.broadcast(flat, src=src)
    out: "OrderedDict[str, T]" = OrderedDict()
    for (n, s, _), off in zip(spec, offs):
        out[n] = flat[off:off + _numel(s)].reshape(tuple(s))
    return out


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpu_count() -> int:
    """GPUs this process may use, WITHOUT any HIP / torch.cuda call (a launcher parent must stay GPU-free: its children are
    started by fork+exec).  The visibility lists (ROCR_VISIBLE_DEVICES, HIP_VISIBLE_DEVICES, CUDA_VISIBLE_DEVICES) win when
    set; otherwise the KFD topology is counted: a node is a GPU when its `simd_count` property is non-zero.  -1 = unknown
    (no sysfs topology readable): the caller lets the ranks themselves fail."""
    import os
    counts = []
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            counts.append(len([t for t in v.split(",") if t.strip() != ""]))
    if counts:
        return min(counts)
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(root)
    except OSError:
        return -1
    n = 0
    for d in nodes:
        try:
            for line in open(os.path.join(root, d, "properties")):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except (OSError, ValueError):
            return -1
    return n


def spawn_ranks(cmd: List[str], world: int, extra_env: Dict[str, str] = None, timeout: float = None, poll: float = 0.2) -> int:
    """Start `world` fresh child processes of `cmd`, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    in their environment, as torch.distributed.run would set them) and return the job's exit code: 0 when every rank exits 0,
    otherwise the FIRST non-zero code seen -- at which point the surviving ranks are ended at once (they would sit in the
    weight broadcast or the closing all_gather until the process-group timeout, 10-30 minutes).
    The caller must not have touched the GPU: children are started with subprocess (fork+exec of a process that never
    initialised HIP), never by re-exec'ing a GPU process.  Rank 0 inherits stdout (its JSON line / progress is the job's);
    the other ranks' stdout goes to stderr so a single machine-readable line stays on stdout."""
    import os
    import subprocess
    import sys
    import time
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else sys.stderr))
    deadline = None if timeout is None else time.monotonic() + timeout
    code = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0:
                    code = rc
                    print(f"spawn_ranks: rank {procs.index(p)} exited with {rc}; ending the other {len(live)} rank(s)", file=sys.stderr)
                    return code
            if deadline is not None and time.monotonic() > deadline:
                raise subprocess.TimeoutExpired(cmd, timeout)
            if live:
                time.sleep(poll)
    finally:
        for p in procs:                      # exactly the PIDs started above
            if p.poll() is None:
                p.kill()
                p.wait()
    return code
