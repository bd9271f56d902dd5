"""Model-level parity on the GPU: NIC (BASELINE config 2) through the real HIP kernels against
the float64 oracle -- per-token probabilities/logit-derived loss within 1e-4 relative, identical
greedy captions, gradients and post-Adam weights over several steps (hipGraph replay included)."""
import os

import numpy as np
import pytest
import torch

from oracle import models as M
from helpers import synth_batch

pytestmark = pytest.mark.gpu


def build(rng, rates, dims, norm="batch", use_graph=True):
    from masters_thesis_amd.nic import NIC
    B, N, T, V, U, E = dims
    model = NIC(N, U, E, V, T, rates[0], rates[1], rates[2], 0.01, 3e-5, 1e-5, norm=norm, seed=11, use_graph=use_graph)
    orc = M.NICDense(N, U, E, V, T, rates[0], rates[1], rates[2], 0.01, 3e-5, 1e-5, norm=norm).init_params(rng)
    for k, v in orc.p.items():
        model.set_weight(k, v)
    return model, orc


DIMS = [(3, 37, 4, 11, 16, 6), (8, 2000, 15, 501, 64, 64)]


@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("rates,norm", [((0, 0, 0), "batch"), ((0.1, 0.2, 0.2), "batch"), ((0, 0, 0), "layer")])
def test_train_parity(dims, rates, norm):
    from masters_thesis_amd.optimizers import Adam
    rng = np.random.default_rng(31)
    B, N, T, V, U, E = dims
    model, orc = build(rng, rates, dims, norm)
    model.compile(Adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=0.1)
    noisy_el = {}
    for step in range(4):                      # eager, capture, replay, replay
        data, tgt = synth_batch(B, N, T, V, U, rng, zero_first=(step == 2))
        res, grads, probs = orc.train_step(data, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        got = model.train_step((data, tgt)).as_floats()
        assert abs(got["loss"] - res["loss"]) <= 1e-4 * abs(res["loss"]), (step, got, res)
        assert abs(got["accuracy"] - res["accuracy"]) < 1e-6
        assert abs(got["L2"] - res["L2"]) <= 1e-4 * abs(res["L2"])
        for k, v in orc.p.items():
            w = model.get_weight(k)
            # Adam's first steps move every weight by ~lr regardless of gradient scale, so compare
            # against the update size, not the weight size.  Elements whose true gradient is zero (an
            # encoder bias whose rows all sit on one side of the LeakyReLU: BatchNorm' sums to 0 over the
            # batch) carry float32 rounding noise ~1e-10 that Adam's g/(|g|+eps) turns into a fraction of lr.
            tol = 2e-2 * 1e-3 + 1e-4 * np.abs(v).max()
            if k in grads and grads[k] is not None:
                noisy = noisy_el.setdefault(k, np.zeros(v.shape, bool))
                noisy |= np.abs(grads[k]) < 1e-8
                tol = tol + 1e-3 * (step + 1) * noisy
            assert (np.abs(w - v) <= tol).all(), (step, k, np.abs(w - v).max())


@pytest.mark.parametrize("dims", DIMS)
def test_train_parity_with_adaptive_gradient_clipping(dims):
    """gradients = agc.adaptive_clip_grad(...) before apply_gradients (agc.py:20-38, lc_NIC.py:388) through
    tnt_agc_f32 / tnt_colsq_f32: post-AGC gradients and post-Adam weights against the oracle, graph replay included."""
    from masters_thesis_amd.optimizers import Adam
    rng = np.random.default_rng(34)
    B, N, T, V, U, E = dims
    model, orc = build(rng, (0, 0.2, 0.2), dims)
    model.compile(Adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    model.enable_agc(0.02, 1e-3)
    orc.agc = (0.02, 1e-3)
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=0.1)
    lam = {"dense_img/kernel": 0.01, "lstm/kernel": 3e-5, "time_distributed_softmax/kernel": 3e-5}
    for step in range(3):
        data, tgt = synth_batch(B, N, T, V, U, rng)
        w0 = {k: v.copy() for k, v in orc.p.items()}
        probs, cache = orc.forward(data, True, M.DropCtx(seed=11, step=step, training=True))
        raw, _ = orc.backward(probs, cache, tgt)
        res, grads, _ = orc.train_step(data, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        got = model.train_step((data, tgt)).as_floats()
        assert abs(got["loss"] - res["loss"]) <= 1e-4 * abs(res["loss"])
        changed = 0
        for k in orc.TRAINABLE:
            g = model.get_gradient(k) + 2 * lam.get(k, 0.0) * w0[k]
            assert np.abs(g - grads[k]).max() <= 2e-4 * np.abs(grads[k]).max() + 1e-9, (step, k)
            changed += int(np.abs(raw[k] - grads[k]).max() > 1e-12)
        assert changed > 0
        for k, v in orc.p.items():
            w = model.get_weight(k)
            assert np.abs(w - v).max() <= 2e-2 * 1e-3 + 1e-4 * np.abs(v).max() + 1e-3 * (step + 1) * float((np.abs(grads.get(k, np.ones(1))) < 1e-8).any()), (step, k)


@pytest.mark.parametrize("dims", DIMS)
def test_forward_gradients_greedy(dims):
    from masters_thesis_amd.optimizers import Adam
    rng = np.random.default_rng(32)
    B, N, T, V, U, E = dims
    model, orc = build(rng, (0, 0, 0), dims, use_graph=False)
    model.compile(Adam(1e-4, clipnorm=None))
    data, tgt = synth_batch(B, N, T, V, U, rng)
    probs, cache = orc.forward(data, False)
    p = model(data, training=False).cpu().numpy()
    # per-token parity on the logits: compare log-probabilities (= logits - logsumexp)
    assert np.abs(np.log(p) - np.log(probs)).max() <= 1e-4 * np.abs(cache["logits"]).max()
    w0 = {k: v.copy() for k, v in orc.p.items()}
    probs, cache = orc.forward(data, True, M.DropCtx(training=True))
    grads, _ = orc.backward(probs, cache, tgt)
    model.train_step((data, tgt))
    lam = {"dense_img/kernel": 0.01, "lstm/kernel": 3e-5, "time_distributed_softmax/kernel": 3e-5}
    for k in orc.TRAINABLE:
        g = model.get_gradient(k) + 2 * lam.get(k, 0.0) * w0[k]
        assert np.abs(g - grads[k]).max() <= 1e-4 * np.abs(grads[k]).max() + 1e-9, k
    # greedy decode: identical captions (argmax ids) and probabilities within tolerance
    for k, v in w0.items():
        model.set_weight(k, v)
    z = np.zeros((B, U), np.float32)
    want = orc_greedy = M.NICDense.greedy_predict
    orc.p = w0
    want = orc.greedy_predict(data[0], z, z, np.ones(B, np.int64), T)
    got = model.greedy_predict(data[0], z, z, np.ones(B, np.int64), T, U)
    assert np.array_equal(got.argmax(-1), want.argmax(-1))
    assert np.abs(got - want).max() <= 1e-4


def test_full_size_properties():
    """BASELINE config-2 size (B=64, N=20000, U=E=512, V=5001, T=15): size-independent checks --
    probabilities sum to 1, loss starts near ln(V), a few Adam steps on one batch reduce the loss,
    moving statistics move, graph replay == eager."""
    from masters_thesis_amd.nic import NIC
    from masters_thesis_amd.optimizers import Adam
    rng = np.random.default_rng(33)
    B, N, T, V, U = 64, 20000, 15, 5001, 512
    data, tgt = synth_batch(B, N, T, V, U, rng, min_len=7)
    losses = {}
    for use_graph in (False, True):
        model = NIC(N, U, 512, V, T, 0, 0.0, 0.0, 0.01, 3e-5, 1e-5, seed=5, use_graph=use_graph)
        model.compile(Adam(1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
        p = model(data, training=False)
        assert torch.allclose(p.sum(-1), torch.ones_like(p.sum(-1)), atol=1e-5)
        ls = [model.train_step((data, tgt)).as_floats()["loss"] for _ in range(6)]
        assert abs(ls[0] - np.log(V)) < 0.5
        assert ls[-1] < ls[0]
        losses[use_graph] = ls
        assert np.abs(model.get_weight("batch_norm/moving_mean")).max() > 0
    assert np.allclose(losses[False], losses[True], rtol=1e-6)


def _twin_models(rates=(0.1, 0.2, 0.2), dims=(8, 2000, 15, 501, 64, 64)):
    from masters_thesis_amd.optimizers import Adam
    rng = np.random.default_rng(77)
    a, orc = build(rng, rates, dims)
    b, _ = build(np.random.default_rng(77), rates, dims)
    for k, v in orc.p.items():
        b.set_weight(k, v)
    for m in (a, b):
        m.compile(Adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    return a, b, dims


def test_launch_plan_replay_equals_graph_replay():
    """ModelBase._run_planned (the step re-issued from recorded C-ABI calls, used by the data-parallel schedule)
    against the captured hipGraph: same launches, so weights, optimizer state and metrics stay bit-identical over
    eager -> record/capture -> replay -> replay steps, dropout included."""
    a, b, (B, N, T, V, U, E) = _twin_models()
    a.plan_step = False                      # a: the captured hipGraph; b: the default since round 3, the recorded launch plan
    b._run_captured = b._run_planned
    rng = np.random.default_rng(5)
    for step in range(5):
        data, tgt = synth_batch(B, N, T, V, U, rng)
        ra, rb = a.train_step((data, tgt)).as_floats(), b.train_step((data, tgt)).as_floats()
        assert ra == rb, (step, ra, rb)
    plans = [v for v in b._graphs.values() if isinstance(v, tuple)]
    assert plans and len(plans[0][1]) >= 10          # the step really was replayed from a recorded plan
    assert not any(isinstance(v, tuple) for v in a._graphs.values())
    torch.cuda.synchronize()
    assert torch.equal(a.arena.theta, b.arena.theta)
    assert torch.equal(a.opt_m, b.opt_m) and torch.equal(a.opt_v, b.opt_v)


def test_pipelined_dp_schedule_world1_equals_single_gpu_step():
    """The data-parallel schedule of config 2 (dp.PipelinedDenseSync: six segments, collectives between them, the
    encoder weight gradient from gathered operands, the update in three arena slices) at world size 1 over RCCL must
    train exactly like the single-GPU step."""
    import socket
    import torch.distributed as dist
    from masters_thesis_amd import dp
    a, b, (B, N, T, V, U, E) = _twin_models()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        dp.attach(b, 1)
        assert isinstance(b.grad_sync, dp.PipelinedDenseSync)
        rng = np.random.default_rng(6)
        lr = 1e-3
        for step in range(5):
            data, tgt = synth_batch(B, N, T, V, U, rng)
            ra, rb = a.train_step((data, tgt)).as_floats(), b.train_step((data, tgt)).as_floats()
            for k in ra:
                assert abs(ra[k] - rb[k]) <= 1e-6 * max(1.0, abs(ra[k])), (step, k, ra, rb)
            if step == 0:
                # from identical weights the two schedules must produce BIT-IDENTICAL gradients (same kernels, same order);
                # what may differ is the last bit of a per-variable squared norm: the single-GPU step sums the span partials in
                # its finalize launch, the schedule per arena slice (seg_sqnorm) -- two summation trees over the same numbers
                torch.cuda.synchronize()
                for k, e in a.arena.entries.items():
                    ga, gb = a.arena.grad[e.off:e.off + e.size], b.arena.grad[e.off:e.off + e.size]
                    assert torch.equal(ga, gb), k
                    if e.seg == a.emb_seg:      # the Embedding's clip norm is the IndexedSlices one, filed in sq_override (the
                        sa, sb = float(a.arena.sq_override[e.seg]), float(b.arena.sq_override[e.seg])   # span norm is not taken)
                    else:
                        sa, sb = float(a.arena.sq[e.seg]), float(b.arena.sq[e.seg])
                    assert abs(sa - sb) <= 4e-7 * abs(sa), (k, sa, sb)
        torch.cuda.synchronize()
        # ... and that last bit of a clip factor is all that separates the weights: within 1e-3 of ONE Adam step after five
        # (Adam's m / sqrt(v) turns a 1e-7 relative change of a gradient into up to ~1e-4 of a step on elements whose
        # successive gradients nearly cancel; measured 3.7e-7 = 3.7e-4 lr on the Embedding, everything else <= 1.2e-7)
        d = (a.arena.theta - b.arena.theta).abs().max().item()
        assert d <= 1e-3 * lr, d
    finally:
        dist.destroy_process_group()


def test_stream_policy_is_a_cache_hint_only():
    """TNT_STREAM_NT (non-temporal optimizer moments, read once per process by the library): three training steps of the
    config-2 shaped model with the policy off and on, each in its own process, leave bit-identical weights."""
    import subprocess, sys
    code = ("import sys, os, hashlib, numpy as np, torch; sys.path.insert(0, '.'); sys.path.insert(0, 'tests');"
            "from masters_thesis_amd.nic import NIC; from masters_thesis_amd.optimizers import Adam; from helpers import synth_batch;"
            "m = NIC(2048, 512, 512, 301, 6, 0.0, 0.2, 0.2, 0.01, 3e-5, 1e-5, device='cuda', seed=3);"
            "m.compile(Adam(1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=0.1)); rng = np.random.default_rng(1);"
            "[m.train_step(synth_batch(16, 2048, 6, 301, 512, rng)) for _ in range(3)]; torch.cuda.synchronize();"
            "print('H', hashlib.sha1(m.arena.theta.cpu().numpy().tobytes()).hexdigest(),"
            " hashlib.sha1(m.opt_v.cpu().numpy().tobytes()).hexdigest())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for nt in ("0", "1"):
        env = dict(os.environ, TNT_STREAM_NT=nt)
        p = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        out.append([ln for ln in p.stdout.splitlines() if ln.startswith("H ")][-1])
    assert out[0] == out[1], out


def test_row_sharded_encoder_update_world1_equals_single_gpu_step():
    """dp.attach(shard_encoder=True) at world size 1 over RCCL on the HIP kernels: the shard is the whole kernel, so the schedule
    (shard product on gemm3, shard norm -> 8-byte all-reduce, everything else through the fused update behind variable 0, shard
    Adam, all-gather of the rows) must train like the single-GPU step; world size 2 against one process: tests/test_dp_gloo.py."""
    import socket
    import torch.distributed as dist
    from masters_thesis_amd import dp
    a, b, (B, N, T, V, U, E) = _twin_models()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        dp.attach(b, 1, shard_encoder=True)
        assert isinstance(b.grad_sync, dp.PipelinedDenseSync) and b.grad_sync._shard_ok(b)
        rng = np.random.default_rng(6)
        lr = 1e-3
        for step in range(5):
            data, tgt = synth_batch(B, N, T, V, U, rng)
            ra, rb = a.train_step((data, tgt)).as_floats(), b.train_step((data, tgt)).as_floats()
            for k in ra:
                assert abs(ra[k] - rb[k]) <= 1e-6 * max(1.0, abs(ra[k])), (step, k, ra, rb)
        torch.cuda.synchronize()
        d = (a.arena.theta - b.arena.theta).abs().max().item()
        assert d <= 1e-3 * lr, d
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("betas_dtype", ["float32", "float16"])
def test_pinned_prefetcher_feeds_the_step(betas_dtype):
    """data.PinnedPrefetcher (pinned double-buffered H2D on a side stream; optionally float16 betas on the wire, widened
    by tnt_stage_batch_h16) against feeding the same host batches directly: identical training, batch after batch."""
    from masters_thesis_amd.data import SyntheticGenerator, PinnedPrefetcher
    a, b, (B, N, T, V, U, E) = _twin_models(rates=(0.0, 0.2, 0.2))
    gen = SyntheticGenerator(6, B, N, U, T, V, seed=3)
    pre = PinnedPrefetcher(SyntheticGenerator(6, B, N, U, T, V, seed=3), "cuda", betas_dtype=betas_dtype)
    for i in range(len(gen)):
        (x, cap, a0, c0), tgt = gen[i]
        if betas_dtype == "float16":
            x = x.astype(np.float16).astype(np.float32)          # what the wire format keeps of the betas
        ra = a.train_step(((x, cap, a0, c0), tgt)).as_floats()
        item = pre[i]
        assert item[0][0].is_cuda and item[0][0].dtype == (torch.float16 if betas_dtype == "float16" else torch.float32)
        rb = b.train_step(item).as_floats()
        assert ra == rb, (i, ra, rb)
    torch.cuda.synchronize()
    assert torch.equal(a.arena.theta, b.arena.theta)


def test_pinned_prefetcher_without_host_syncs():
    """The prefetcher's point is a host that runs ahead of the device: 12 steps over 6 wide batches x 2 epochs with
    NO host read in between (metrics left on the device), against the same batches fed one by one with a
    synchronisation after each.  Bit-identical weights: no batch was torn by a reused pinned or device slot."""
    from masters_thesis_amd.data import SyntheticGenerator, PinnedPrefetcher
    a, b, (B, N, T, V, U, E) = _twin_models(rates=(0.0, 0.2, 0.2))
    gen = SyntheticGenerator(6, B, N, U, T, V, seed=4)
    pre = PinnedPrefetcher(SyntheticGenerator(6, B, N, U, T, V, seed=4), "cuda")
    for epoch in range(2):
        for i in range(len(gen)):
            a.train_step(gen[i])
            torch.cuda.synchronize()
        for i in range(len(pre)):
            b.train_step(pre[i])                      # no .as_floats(), no synchronize
        pre.on_epoch_end()
    torch.cuda.synchronize()
    assert torch.equal(a.arena.theta, b.arena.theta)


def test_full_cortex_width_through_the_prefetcher():
    """the reference's generator yields betas of width 327 684 (data_generator_guse.py:129-171; 84 MB per
    batch of 64 in float32).  The region-wise model over that width (360 regions that together reference a subset of
    the cortex, as the Glasser groups do) trained through PinnedPrefetcher with float32 and with float16 on-wire betas
    and through compact_groups (only the referenced voxels cross PCIe): same losses as feeding the host batch directly."""
    from masters_thesis_amd.data import PinnedPrefetcher
    from masters_thesis_amd.lc_nic import NIC, synthetic_groups, compact_groups
    from masters_thesis_amd.optimizers import Adam
    N, R, D, B, T, V, U = 327684, 360, 32, 64, 15, 5001, 512
    rng = np.random.default_rng(12)
    groups = synthetic_groups(62756, R, D, seed=42)                        # the visual-cortex subset (62 756 voxels) ...
    perm = np.sort(rng.choice(N, 62756, replace=False))
    groups = ([perm[np.asarray(g)] for g in groups[0]], groups[1])         # ... scattered over the full cortex

    class Gen:
        def __len__(self): return 3
        def on_epoch_end(self): pass
        def __getitem__(self, i):
            r = np.random.default_rng(100 + i)
            x = r.standard_normal((B, N)).astype(np.float32)
            data, tgt = synth_batch(B, 4, T, V, U, r)
            return ((x, data[1], data[2], data[3]), tgt)
    gen = Gen()
    mk = lambda g: NIC(g, U, 512, 512, 32, V, T, 0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.01, 0.001, 3e-5, 1e-5, seed=5)
    a, b = mk(groups), mk(groups)
    for m in (a, b):
        m.compile(Adam(1e-4, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    pre = PinnedPrefetcher(gen, "cuda")
    la = [a.train_step(gen[i]).as_floats()["loss"] for i in range(3)]
    lb = [b.train_step(pre[i]).as_floats()["loss"] for i in range(3)]
    assert la == lb and np.isfinite(la).all()
    # compact_groups: the same function on the referenced columns only (62 756 of 327 684: 16 MB instead of 84 MB per batch)
    cols, cg = compact_groups(groups)
    c = mk(cg)
    c.compile(Adam(1e-4, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    lc = []
    for i in range(3):
        (x, cap, a0, c0), tgt = gen[i]
        lc.append(c.train_step(((np.ascontiguousarray(x[:, cols]), cap, a0, c0), tgt)).as_floats()["loss"])
    assert np.allclose(lc, la, rtol=1e-5)
