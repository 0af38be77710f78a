"""N > 1 path on CPU: two gloo ranks exercise the weight broadcast and the work sharding that bench.py and
the sharded plugin driver use on the GPU node (there over RCCL)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def test_shard_range_partitions_exactly():
    from isr2_amd.parallel import shard_range, shard_list
    for n in (0, 1, 5, 8, 100, 601):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b, e = shard_range(n, r, world)
                assert 0 <= b <= e <= n
                seen += list(range(b, e))
            assert seen == list(range(n))
            sizes = [len(shard_list(list(range(n)), r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from isr2_amd.parallel import broadcast_state_dict, shard_range
    from isr2_amd.weights import param_spec, synth_state_dict
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = param_spec(parts=("fusion",))
    src = synth_state_dict(1234, parts=("fusion",)) if rank == 0 else None
    sd = broadcast_state_dict(src, spec, rank, world, torch.device("cpu"))
    ref = synth_state_dict(1234, parts=("fusion",))
    ok = all(torch.equal(sd[k], ref[k]) for k in ref)
    # work sharding: every rank sums its own items, one all_reduce verifies full coverage (test only)
    b, e = shard_range(601, rank, world)
    t = torch.tensor([float(sum(range(b, e)))], dtype=torch.float64)
    dist.all_reduce(t)
    q.put((rank, ok, float(t.item())))
    dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res), "broadcast state dict differs from rank 0's"
    assert all(r[2] == float(sum(range(601))) for r in res)
