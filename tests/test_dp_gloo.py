"""Data-parallel path on CPU: world_size 2 over gloo, mock kernel backend.
Contract (dp.py): G ranks x local batch == one rank on the concatenated batch (LayerNorm
encoder, since BatchNorm statistics are per replica by design), replicas stay identical."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
DIMS = dict(B=4, N=23, T=5, V=13, U=16, E=10)


def _worker(rank, world, port, kind, q):
    sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import masters_thesis_amd.ops as ops
    from masters_thesis_amd import dp
    from masters_thesis_amd.optimizers import Adam
    from mock_backend import MockBackend
    from helpers import synth_batch
    ops.set_backend(MockBackend())
    model = _make(kind, seed=100 + rank)           # different init per rank: broadcast must fix it
    model.compile(Adam(1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    dp.attach(model)
    rng = np.random.default_rng(7)
    out = []
    for step in range(2):
        data, tgt = _global_batch(rng, world)
        sl = slice(rank * DIMS["B"], (rank + 1) * DIMS["B"])
        local = tuple(a[sl] for a in data)
        m = model.train_step((local, tgt[sl]))
        out.append({k: float(v) for k, v in dp.allreduce_metrics(m).items()})
    out.append(type(model.grad_sync).__name__)
    q.put((rank, {k: v for k, v in model.get_weights_dict().items()}, out))
    dist.barrier()
    dist.destroy_process_group()


def _make(kind, seed, **kw):
    d = DIMS
    if kind == "nic":
        from masters_thesis_amd.nic import NIC
        return NIC(d["N"], d["U"], d["E"], d["V"], d["T"], 0.0, 0.0, 0.0, 0.01, 3e-5, 1e-5, norm="layer", device="cpu",
                   seed=seed, **kw)
    from masters_thesis_amd.lc_nic import NIC
    from helpers import tiny_groups
    g = (tiny_groups(d["N"], 4, np.random.default_rng(3)), [16] * 4)
    return NIC(g, d["U"], 512, d["E"], 6, d["V"], d["T"], 0, 0, 0, 0, 0, 0, 0.01, 0.001, 3e-5, 1e-5, norm="layer",
               device="cpu", seed=seed, **kw)


def _global_batch(rng, world):
    from helpers import synth_batch
    d = DIMS
    return synth_batch(world * d["B"], d["N"], d["T"], d["V"], d["U"], rng)


@pytest.mark.parametrize("kind", ["nic", "lcnic"])
def test_dp2_equals_single_process(kind):
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(world):
        r, w, out = q.get(timeout=120)
        res[r] = (w, out)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # both model families run their pipelined (bucket-per-launch-group) schedule, not the generic fallback
    assert res[0][1][-1] == {"nic": "PipelinedDenseSync", "lcnic": "PipelinedAttentionSync"}[kind]
    # replicas identical
    for k in res[0][0]:
        assert np.array_equal(res[0][0][k], res[1][0][k]), k
    # single process on the concatenated batch (rank 0's initial weights = seed 100)
    sys.path.insert(0, HERE)
    import masters_thesis_amd.ops as ops
    from masters_thesis_amd.optimizers import Adam
    from mock_backend import MockBackend
    old = ops._backend
    ops.set_backend(MockBackend())
    try:
        ref = _make(kind, seed=100)
        ref.compile(Adam(1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
        rng = np.random.default_rng(7)
        for step in range(2):
            data, tgt = _global_batch(rng, world)
            m = ref.train_step((data, tgt)).as_floats()
            assert abs(m["loss"] - res[0][1][step]["loss"]) < 1e-5
        if True:                     # dropout-free: weights must match the single-process run
            for k, v in ref.get_weights_dict().items():
                if k == "attention/V/bias":
                    continue
                assert np.allclose(res[0][0][k], v, rtol=1e-4, atol=2e-6), k
    finally:
        ops.set_backend(old)


def _worker_syncbn(rank, world, port, kind, q):
    sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import masters_thesis_amd.ops as ops
    from masters_thesis_amd import dp
    from masters_thesis_amd.optimizers import Adam
    from mock_backend import MockBackend
    ops.set_backend(MockBackend())
    model = _make_bn(kind, seed=100 + rank)
    model.compile(Adam(1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    dp.attach(model, sync_bn=True)
    rng = np.random.default_rng(7)
    losses = []
    for step in range(2):
        data, tgt = _global_batch(rng, world)
        sl = slice(rank * DIMS["B"], (rank + 1) * DIMS["B"])
        m = model.train_step((tuple(a[sl] for a in data), tgt[sl]))
        losses.append(float(dp.allreduce_metrics(m)["loss"]))
    q.put((rank, model.get_weights_dict(), losses, type(model.grad_sync).__name__))
    dist.barrier()
    dist.destroy_process_group()


def _make_bn(kind, seed):
    d = DIMS
    if kind == "nic":
        from masters_thesis_amd.nic import NIC
        return NIC(d["N"], d["U"], d["E"], d["V"], d["T"], 0.0, 0.0, 0.0, 0.01, 3e-5, 1e-5, norm="batch", device="cpu",
                   seed=seed)
    from masters_thesis_amd.lc_nic import NIC
    from helpers import tiny_groups
    g = (tiny_groups(d["N"], 4, np.random.default_rng(3)), [16] * 4)
    return NIC(g, d["U"], 512, d["E"], 6, d["V"], d["T"], 0, 0, 0, 0, 0, 0, 0.01, 0.001, 3e-5, 1e-5, norm="batch",
               device="cpu", seed=seed)


@pytest.mark.parametrize("kind", ["nic", "lcnic"])
def test_dp2_sync_batchnorm_equals_single_process_on_the_global_batch(kind):
    """dp.attach(sync_bn=True): BatchNorm statistics over the global batch (all-gathered chunk partials in the forward,
    all-reduced sums in the backward) -- two replicas on half batches train like ONE process on the whole batch, moving
    statistics included, with the BatchNorm encoders of both model families."""
    world, port = 2, 33500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_syncbn, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, w, losses, sched = q.get(timeout=120)
        res[r] = (w, losses, sched)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][2] == "function" or "Pipelined" not in res[0][2]           # generic schedule
    for k in res[0][0]:
        assert np.array_equal(res[0][0][k], res[1][0][k]), k                    # replicas identical, moving statistics too
    sys.path.insert(0, HERE)
    import masters_thesis_amd.ops as ops
    from masters_thesis_amd.optimizers import Adam
    from mock_backend import MockBackend
    old = ops._backend
    ops.set_backend(MockBackend())
    try:
        ref = _make_bn(kind, seed=100)
        ref.compile(Adam(1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
        rng = np.random.default_rng(7)
        for step in range(2):
            data, tgt = _global_batch(rng, world)
            m = ref.train_step((data, tgt)).as_floats()
            assert abs(m["loss"] - res[0][1][step]) < 1e-5, (step, m["loss"], res[0][1][step])
        for k, v in ref.get_weights_dict().items():
            if k == "attention/V/bias":
                continue
            assert np.allclose(res[0][0][k], v, rtol=1e-4, atol=5e-6), (k, np.abs(res[0][0][k] - v).max())
    finally:
        ops.set_backend(old)


def _worker_bn(rank, world, port, tmp, q):
    sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import masters_thesis_amd.ops as ops
    from masters_thesis_amd import dp
    from masters_thesis_amd.callbacks import ModelCheckpoint
    from masters_thesis_amd.nic import NIC
    from masters_thesis_amd.optimizers import Adam
    from mock_backend import MockBackend
    ops.set_backend(MockBackend())
    d = DIMS
    model = NIC(d["N"], d["U"], d["E"], d["V"], d["T"], 0.0, 0.0, 0.0, 0.01, 3e-5, 1e-5, norm="batch", device="cpu",
                seed=100 + rank)
    model.compile(Adam(1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    dp.attach(model)
    data, tgt = _global_batch(np.random.default_rng(9), world)
    sl = slice(rank * d["B"], (rank + 1) * d["B"])
    model.train_step((tuple(a[sl] for a in data), tgt[sl]))
    local = model.get_weight("batch_norm/moving_mean")
    ck = ModelCheckpoint(os.path.join(tmp, "w.npz"), save_best_only=False)
    ck.set_model(model)
    ck.on_epoch_end(0, {"val_loss": 1.0 + rank})
    q.put((rank, local, model.get_weight("batch_norm/moving_mean"), model.get_weight("lstm/kernel")))
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_batchnorm_stats_are_per_replica_and_averaged_at_checkpoint(tmp_path):
    world, port = 2, 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bn, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, local, avg, w = q.get(timeout=120)
        res[r] = (local, avg, w)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert not np.allclose(res[0][0], res[1][0])                      # each replica saw its own batch
    want = (res[0][0] + res[1][0]) / 2
    assert np.allclose(res[0][1], want, atol=1e-7) and np.allclose(res[1][1], want, atol=1e-7)
    assert np.array_equal(res[0][2], res[1][2])                       # trainables stay identical
    with np.load(os.path.join(str(tmp_path), "w.npz")) as z:          # written once, by rank 0
        assert np.allclose(z["batch_norm__moving_mean"], want, atol=1e-7)


def _fit_worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import masters_thesis_amd.ops as ops
    from masters_thesis_amd import dp
    from masters_thesis_amd.optimizers import Adam
    from masters_thesis_amd.callbacks import Callback
    from mock_backend import MockBackend
    ops.set_backend(MockBackend())
    model = _make("nic", seed=100 + rank)
    model.compile(Adam(1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    dp.attach(model)
    rng = np.random.default_rng(50 + rank)           # rank-local batches: rank-local losses
    d = DIMS
    from helpers import synth_batch
    batches = [synth_batch(d["B"], d["N"], d["T"], d["V"], d["U"], rng) for _ in range(2)]
    seen = []

    class StopOnRank1(Callback):                     # a rank-local decision (what EarlyStopping on a rank-local val_loss is)
        def on_epoch_end(self, epoch, logs=None):
            seen.append(dict(logs))
            if rank == 1 and epoch == 1:
                self.model.stop_training = True
    hist = model.fit(batches, epochs=5, callbacks=[StopOnRank1()], verbose=0)
    q.put((rank, len(hist["loss"]), seen))
    dist.barrier()
    dist.destroy_process_group()


def test_fit_logs_and_stop_flag_agree_across_ranks():
    """fit() under data parallel: the epoch logs the callbacks see are the mean over the ranks (identical everywhere) and
    a stop_training raised on ONE rank ends the loop on every rank in the same epoch (no rank is left in a collective)."""
    world, port = 2, 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fit_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, n, seen = q.get(timeout=120)
        res[r] = (n, seen)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] == 2, (res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert a.keys() == b.keys() and all(abs(a[k] - b[k]) < 1e-12 for k in a)


def _worker_shard(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import masters_thesis_amd.ops as ops
    from masters_thesis_amd import dp
    from masters_thesis_amd.optimizers import Adam
    from mock_backend import MockBackend
    from helpers import synth_batch
    ops.set_backend(MockBackend())
    model = _make_n24(seed=100 + rank)
    model.compile(Adam(1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    dp.attach(model, shard_encoder=True)
    rng = np.random.default_rng(7)
    d = DIMS
    out = []
    for step in range(3):
        data, tgt = synth_batch(world * d["B"], 24, d["T"], d["V"], d["U"], rng)
        sl = slice(rank * d["B"], (rank + 1) * d["B"])
        m = model.train_step((tuple(a[sl] for a in data), tgt[sl]))
        out.append({k: float(v) for k, v in dp.allreduce_metrics(m).items()})
    sharded = model.grad_sync._shard_ok(model)
    q.put((rank, model.get_weights_dict(), out, sharded))
    dist.barrier()
    dist.destroy_process_group()


def _make_n24(seed):
    d = DIMS
    from masters_thesis_amd.nic import NIC
    return NIC(24, d["U"], d["E"], d["V"], d["T"], 0.0, 0.0, 0.0, 0.01, 3e-5, 1e-5, norm="layer", device="cpu", seed=seed)


def test_dp2_row_sharded_encoder_update_equals_single_process():
    """dp.attach(shard_encoder=True): each of two ranks forms HALF the rows of the encoder kernel's gradient from the gathered
    operands, the clip norm is the all-reduced pair of shard norms, Adam runs on the shard, the updated rows are all-gathered --
    three steps (the moments of a shard live on its owner only) train like ONE process on the concatenated batch, replicas
    identical, L2 metric (sum theta^2 of the whole kernel) included."""
    world, port = 2, 35500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_shard, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, w, out, sharded = q.get(timeout=120)
        res[r] = (w, out, sharded)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][2] and res[1][2], "the sharded path did not apply"
    for k in res[0][0]:
        assert np.array_equal(res[0][0][k], res[1][0][k]), k
    sys.path.insert(0, HERE)
    import masters_thesis_amd.ops as ops
    from masters_thesis_amd.optimizers import Adam
    from mock_backend import MockBackend
    from helpers import synth_batch
    old = ops._backend
    ops.set_backend(MockBackend())
    try:
        ref = _make_n24(seed=100)
        ref.compile(Adam(1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
        rng = np.random.default_rng(7)
        d = DIMS
        for step in range(3):
            data, tgt = synth_batch(world * d["B"], 24, d["T"], d["V"], d["U"], rng)
            m = ref.train_step((data, tgt)).as_floats()
            assert abs(m["loss"] - res[0][1][step]["loss"]) < 1e-5
            assert abs(m["L2"] - res[0][1][step]["L2"]) < 1e-6 * max(1.0, abs(m["L2"])), (step, m["L2"], res[0][1][step]["L2"])
        for k, v in ref.get_weights_dict().items():
            assert np.allclose(res[0][0][k], v, rtol=1e-4, atol=2e-6), (k, np.abs(res[0][0][k] - v).max())
    finally:
        ops.set_backend(old)
