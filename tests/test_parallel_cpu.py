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


# ---- image-sharded plugin driver (BASELINE config 4), two gloo ranks on CPU ---------------------------------------------
def _plugin_worker(rank, world, port, src, dst, logdir):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "FF_DIST_BACKEND": "gloo", "FF_IO_THREADS": "0"})
    import models.team29_FreqFusion.io as plug

    class StandIn:                                   # the HIP model needs a GPU; sharding / naming / collectives do not
        def __call__(self, x):
            return torch.nn.functional.interpolate(x, scale_factor=4, mode="nearest")

    seen = []
    orig_load = plug._load_image
    plug._build_and_load = lambda model_dir, device, rank=0, world=1: StandIn()
    plug._load_image = lambda path, device=None: (seen.append(os.path.basename(path)), orig_load(path, None))[1]
    plug.main(model_dir="unused.pth", input_path=src, output_path=dst, device=torch.device("cpu"))
    with open(os.path.join(logdir, f"rank{rank}.txt"), "w") as f:
        f.write("\n".join(seen))


def test_two_rank_plugin_shards_every_image_exactly_once(tmp_path):
    """main() under WORLD_SIZE=2: the sorted file list is cut into contiguous shards (reference eval.py:166-170), every image is
    read by exactly one rank and written exactly once, names are preserved, and the closing all_gather agrees on the total."""
    import numpy as np
    from PIL import Image
    src, dst, logs = tmp_path / "in", tmp_path / "out", tmp_path / "logs"
    for d in (src, dst, logs):
        d.mkdir()
    names = [f"img_{i:03d}.png" for i in range(7)]
    rng = np.random.default_rng(0)
    for n in names:
        Image.fromarray(rng.integers(0, 256, size=(6, 5, 3), dtype=np.uint8)).save(src / n)
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_plugin_worker, args=(r, 2, port, str(src), str(dst), str(logs))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    per_rank = [open(logs / f"rank{r}.txt").read().split() for r in range(2)]
    assert per_rank[0] == names[:4] and per_rank[1] == names[4:]              # contiguous, first rank takes the extra one
    assert sorted(os.listdir(dst)) == names
    for n in names:                                                            # nearest x4 of the input: content is the right image's
        a, b = np.array(Image.open(src / n)), np.array(Image.open(dst / n))
        assert b.shape == (24, 20, 3) and np.array_equal(b[::4, ::4], a)


def test_spawn_ranks_sets_rank_environment_and_reports_failure(tmp_path):
    """parallel.spawn_ranks (what `bench.py --gpus N` and run_sharded use to start themselves): one fresh child per rank with
    the torch.distributed.run environment; a failing rank makes the job fail."""
    import sys
    from isr2_amd.parallel import spawn_ranks
    code = ("import os,sys; open(os.path.join(sys.argv[1], 'r' + os.environ['RANK']), 'w').write("
            "os.environ['WORLD_SIZE'] + ' ' + os.environ['LOCAL_RANK'] + ' ' + os.environ['MASTER_ADDR']); "
            "sys.exit(3 if os.environ['RANK'] == sys.argv[2] else 0)")
    assert spawn_ranks([sys.executable, "-c", code, str(tmp_path), "-1"], 3) == 0
    assert sorted(os.listdir(tmp_path)) == ["r0", "r1", "r2"]
    assert open(tmp_path / "r2").read() == "3 2 127.0.0.1"
    assert spawn_ranks([sys.executable, "-c", code, str(tmp_path), "1"], 2) == 3


def test_fusion_checkpoint_is_mandatory(tmp_path, monkeypatch):
    """a missing fusion checkpoint must raise (reference io.py:164 torch.load fails), not silently run on synthetic
    weights; FF_ALLOW_SYNTH=1 is the explicit opt-in the tests and the bench use."""
    import models.team29_FreqFusion.io as plug
    monkeypatch.delenv("FF_ALLOW_SYNTH", raising=False)
    with pytest.raises(FileNotFoundError):
        plug._build_state_dict(str(tmp_path / "missing.pth"), str(tmp_path / "no_pretrained"), verbose=False)
    torch.save({"model_state_dict": {"unrelated.weight": torch.zeros(3)}}, tmp_path / "empty.pth")
    with pytest.raises(RuntimeError):
        plug._build_state_dict(str(tmp_path / "empty.pth"), str(tmp_path / "no_pretrained"), verbose=False)
    monkeypatch.setenv("FF_ALLOW_SYNTH", "1")
    with pytest.warns(UserWarning):
        sd = plug._build_state_dict(str(tmp_path / "missing.pth"), str(tmp_path / "no_pretrained"), verbose=False)
    assert len(sd) > 1000


def test_spawn_ranks_ends_survivors_when_a_rank_dies(tmp_path):
    """a rank that dies early must not leave its peers waiting (in the real job: inside the weight broadcast until the
    process-group timeout).  Rank 1 exits with 5 at once, rank 0 would sleep for 10 minutes: the job returns 5 within seconds."""
    import sys
    import time
    from isr2_amd.parallel import spawn_ranks
    code = ("import os,sys,time; r = os.environ['RANK']; open(os.path.join(sys.argv[1], 'pid' + r), 'w').write(str(os.getpid())); "
            "sys.exit(5) if r == '1' else time.sleep(600)")
    t0 = time.monotonic()
    assert spawn_ranks([sys.executable, "-c", code, str(tmp_path)], 2) == 5
    assert time.monotonic() - t0 < 60
    time.sleep(0.2)
    pid0 = int(open(tmp_path / "pid0").read())
    assert not os.path.exists(f"/proc/{pid0}") or open(f"/proc/{pid0}/stat").read().split()[2] == "Z"   # the sleeper is gone


def test_visible_gpu_count_uses_no_gpu_call(monkeypatch):
    """The launcher parent counts GPUs from the visibility lists / the KFD topology in sysfs, never through HIP."""
    from isr2_amd import parallel
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    n = parallel.visible_gpu_count()
    assert n >= -1                                                  # this container: no KFD -> -1 (unknown) or 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert parallel.visible_gpu_count() == 3
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4,5")
    assert parallel.visible_gpu_count() == 2
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "")
    assert parallel.visible_gpu_count() == 0
    import inspect
    src = inspect.getsource(parallel.visible_gpu_count) + inspect.getsource(parallel.spawn_ranks)
    assert "torch.cuda" not in src.replace("torch.cuda call", "") and "hip" not in src.lower().replace("hip_visible_devices", "").replace("hip /", "").replace("initialised hip", "")


def test_run_sharded_checks_paths_before_spawning(tmp_path, monkeypatch, capsys):
    """A bad --model_dir fails in the parent (exit 2), before any rank is started."""
    from isr2_amd import run_sharded
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    monkeypatch.delenv("FF_ALLOW_SYNTH", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    (tmp_path / "in").mkdir()
    called = []
    monkeypatch.setattr("isr2_amd.parallel.spawn_ranks", lambda *a, **k: called.append(a) or 0)
    rc = run_sharded.main(["--gpus", "2", "--input", str(tmp_path / "in"), "--output", str(tmp_path / "out"), "--model_dir", str(tmp_path / "nope.pth")])
    assert rc == 2 and not called
    assert "not found" in capsys.readouterr().err
    rc = run_sharded.main(["--gpus", "2", "--input", str(tmp_path / "missing_in"), "--output", str(tmp_path / "out"), "--model_dir", str(tmp_path / "nope.pth")])
    assert rc == 2 and not called
    torch.save({"model_state_dict": {}}, tmp_path / "f.pth")
    rc = run_sharded.main(["--gpus", "2", "--input", str(tmp_path / "in"), "--output", str(tmp_path / "out"), "--model_dir", str(tmp_path / "f.pth")])
    assert rc == 0 and len(called) == 1
