"""Training-side checkpoint contract (VERDICT r2 missing #7; reference src/utils/checkpoint_manager.py:81-165 save, :185-239 load,
train.py:943-966 resume) and the data-parallel gradient exchange.

  * CPU, build container only (needs /root/reference): a checkpoint assembled by isr2_amd.train.build_checkpoint loads through the
    reference's OWN CheckpointManager.load_checkpoint into its model, torch.optim.AdamW, CosineAnnealingWarmRestarts and EMAModel --
    strict state-dict load, optimizer moments attached to the right parameters, and the restored optimizer can step.
  * GPU: save -> load into a fresh FusionTrainer -> the next step is bit-identical to the uninterrupted run.
  * GPU, two ranks over gloo sharing the card: distributed_step on two half-batches equals, bit for bit, a single-process step on
    the average of the two half-batch gradients (DDP semantics without SyncBatchNorm).
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, GOLD)
REF = os.environ.get("FF_REFERENCE_ROOT", "/root/reference")


def test_allreduce_mean_on_cpu_tensors_two_ranks(tmp_path):
    """The 4 MB flat gradient exchange over gloo: every rank ends with the mean of the ranks' buffers."""
    import torch.multiprocessing as mp
    from test_parallel_cpu import _free_port
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_allreduce_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    want = (torch.arange(1017906, dtype=torch.float32) * 1e-6 + 0.5 * (1.0 + 3.0))
    assert torch.equal(a, b) and torch.allclose(a, want)


def _allreduce_worker(rank, world, port, out):
    import torch.distributed as dist
    sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..")))
    from isr2_amd.train import allreduce_mean_
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(1017906, dtype=torch.float32) * 1e-6 + (1.0 if rank == 0 else 3.0)
    allreduce_mean_(flat, world)
    torch.save(flat, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_checkpoint_loads_through_the_reference_checkpoint_manager(tmp_path):
    if not os.path.exists(os.path.join(REF, "src", "utils", "checkpoint_manager.py")):
        pytest.skip("reference tree not present (build container only)")
    import contextlib
    import io
    from make_golden_train import build_cached_mode_model, HP
    from isr2_amd import train as TR
    with contextlib.redirect_stdout(io.StringIO()):
        model, sd, _ = build_cached_mode_model()
        from src.utils.checkpoint_manager import CheckpointManager, EMAModel
    full = model.state_dict()                                        # complete, in the reference's own key order
    names = TR.trainable_names(full)
    assert len(names) == 222
    g = torch.Generator().manual_seed(0)
    params = {k: full[k] + 0.01 * torch.randn(full[k].shape, generator=g) for k in names}
    m = {k: 0.01 * torch.randn(full[k].shape, generator=g) for k in names}
    v = {k: 1e-4 * torch.rand(full[k].shape, generator=g) for k in names}
    model_sd = type(full)((k, params.get(k, t)) for k, t in full.items())
    ema = {k: model_sd[k] * 0.5 for k in model_sd if TR.is_parameter_key(k)}
    hp = dict(TR.HP)
    ck = TR.build_checkpoint(model_sd, m, v, ema, step=7, hp=hp, epoch=3, metrics={"psnr": 30.05, "loss": 0.02}, lr_now=1.2e-4,
                             scheduler_state={"T_0": 50, "T_i": 50, "T_mult": 2, "eta_min": 5e-8, "T_cur": 3, "base_lrs": [1.5e-4], "last_epoch": 3,
                                              "_step_count": 4, "_get_lr_called_within_step": False, "_last_lr": [1.2e-4]})
    path = str(tmp_path / "checkpoint_epoch0003.pth")
    TR._save_atomic(ck, path)
    assert not os.path.exists(str(tmp_path / "checkpoint_epoch0003.tmp"))
    # --- the reference side: exactly what train.py:943-966 does on resume
    with contextlib.redirect_stdout(io.StringIO()):
        model2, _, _ = build_cached_mode_model()
        opt = torch.optim.AdamW(model2.parameters(), lr=HP["lr"], betas=HP["betas"], weight_decay=HP["weight_decay"], eps=HP["eps"])
        sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=50, T_mult=2, eta_min=5e-8)
        ema2 = EMAModel(model2, decay=0.999)
        mgr = CheckpointManager(str(tmp_path / "mgr"))
        got = mgr.load_checkpoint(path, model2, opt, sched, load_optimizer=True, device="cpu")
        ema2.load_state_dict(got["ema_state_dict"])
    assert got["epoch"] == 3 and got["metrics"]["psnr"] == 30.05
    named = dict(model2.named_parameters())
    for k in names:
        assert torch.equal(named[k].data, params[k]), k
        st = opt.state[named[k]]
        assert torch.equal(st["exp_avg"], m[k]) and torch.equal(st["exp_avg_sq"], v[k]) and float(st["step"]) == 7.0, k
    dead = [k for k in named if k not in names]
    assert len(dead) == 20 and all(named[k] not in opt.state or not opt.state[named[k]] for k in dead)
    assert ema2.decay == hp["ema_decay"] and len(ema2.shadow) == 242
    assert torch.equal(ema2.shadow[names[5]], ema[names[5]])
    assert opt.param_groups[0]["lr"] == 1.2e-4 and sched.T_cur == 3
    for p in model2.parameters():                                     # the restored optimizer is usable
        p.grad = torch.zeros_like(p) + 1e-3
    opt.step()
    sched.step()


@pytest.mark.gpu
def test_checkpoint_round_trip_resumes_bit_exactly(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from train_inputs import make_train_batch
    from isr2_amd.train import FusionTrainer
    from isr2_amd.weights import synth_state_dict
    d = {k: torch.from_numpy(v) for k, v in make_train_batch(77, 2, 16, 16).items()}
    outs = {k: d["out_" + k] for k in ("hat", "dat", "nafnet")}
    feats = {k: d["feat_" + k] for k in ("hat", "dat", "nafnet")}
    sd = synth_state_dict(1234, parts=("fusion", "collab"))
    a = FusionTrainer(sd, "cuda:0", dropout=0.1, seed=5)
    for _ in range(2):
        a.step(d["lr"], d["hr"], outs, feats)
    path = a.save_checkpoint(str(tmp_path / "ck.pth"), epoch=1, metrics={"psnr": 20.0})
    a.step(d["lr"], d["hr"], outs, feats)
    ck = torch.load(path, weights_only=True)
    assert sorted(ck) == ["ema_state_dict", "epoch", "ff_trainer", "metrics", "model_state_dict", "optimizer_state_dict", "timestamp"]
    assert len(ck["optimizer_state_dict"]["state"]) == 222 and len(ck["ema_state_dict"]["shadow"]) == 222   # the synthetic dict has no dead parameters
    assert int(ck["model_state_dict"]["cross_band_attn.lka_block.norm1.num_batches_tracked"]) == 18
    b = FusionTrainer(synth_state_dict(99, parts=("fusion", "collab")), "cuda:0", dropout=0.0, seed=0)
    b.load_checkpoint(path)
    assert b.step_count == 2 and b.dropout == 0.1 and b.seed == 5
    b.step(d["lr"], d["hr"], outs, feats)
    assert torch.equal(a.P, b.P) and torch.equal(a.EMA, b.EMA) and torch.equal(a.M, b.M) and torch.equal(a.V, b.V)
    for k in a.buffers:
        assert torch.equal(a.buffers[k], b.buffers[k]), k
    assert a.nbt == b.nbt


def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist
    sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..")))
    sys.path.insert(0, GOLD)
    from train_inputs import make_train_batch
    from isr2_amd.train import FusionTrainer, distributed_step
    from isr2_amd.weights import synth_state_dict
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = {k: torch.from_numpy(v) for k, v in make_train_batch(78, 4, 16, 16).items()}
    sl = slice(2 * rank, 2 * rank + 2)                               # this rank's shard of the global batch of 4
    outs = {k: d["out_" + k][sl] for k in ("hat", "dat", "nafnet")}
    feats = {k: d["feat_" + k][sl] for k in ("hat", "dat", "nafnet")}
    tr = FusionTrainer(synth_state_dict(1234, parts=("fusion", "collab")), "cuda:0", dropout=0.0)
    distributed_step(tr, d["lr"][sl], d["hr"][sl], outs, feats, world)
    torch.cuda.synchronize()
    torch.save({"P": tr.P.cpu(), "G": tr.G.cpu()}, os.path.join(out, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_step_equals_step_on_averaged_shard_gradients(tmp_path):
    """one flat all-reduce per optimizer step.  Two ranks (gloo, sharing the one GPU of the box; on the
    8-GPU node the same code runs over RCCL) against a single process that averages the two shard gradients by hand."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    from test_parallel_cpu import _free_port
    from train_inputs import make_train_batch
    from isr2_amd.train import FusionTrainer
    from isr2_amd.weights import synth_state_dict
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["P"], r1["P"]) and torch.equal(r0["G"], r1["G"])       # ranks stay in lock-step
    d = {k: torch.from_numpy(v) for k, v in make_train_batch(78, 4, 16, 16).items()}
    tr = FusionTrainer(synth_state_dict(1234, parts=("fusion", "collab")), "cuda:0", dropout=0.0)
    gs = []
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        t2 = FusionTrainer(synth_state_dict(1234, parts=("fusion", "collab")), "cuda:0", dropout=0.0)
        t2.forward_backward(d["lr"][sl], d["hr"][sl], {k: d["out_" + k][sl] for k in ("hat", "dat", "nafnet")},
                            {k: d["feat_" + k][sl] for k in ("hat", "dat", "nafnet")})
        gs.append(t2.G.clone())
    tr.G.copy_((gs[0] + gs[1]) * 0.5)
    tr.optimizer_step()
    torch.cuda.synchronize()
    assert torch.equal(tr.G.cpu(), r0["G"]) and torch.equal(tr.P.cpu(), r0["P"])
