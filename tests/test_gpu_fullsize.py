"""BASELINE.json's full sizes (config 2: N=20000, U=E=512, V=5001, T=15, B=64; config 3: 360 regions x 32 + attention)
on the GPU: one eval forward against the float64 oracle (it finishes in seconds at this size), and size-independent
properties of the training step -- bit-exact run-to-run determinism (no atomics anywhere on the path), hipGraph replay ==
eager launches, probabilities on the simplex, loss going down on a fixed batch, weights round trip, and the edge shapes
B = 1 / T = 1."""
import numpy as np
import pytest
import torch

from oracle import models as M
from oracle import ops as O

pytestmark = pytest.mark.gpu

B, T, V, U, E, N = 64, 15, 5001, 512, 512, 20000


def synth(rng, b=B, t=T):
    x = rng.standard_normal((b, N)).astype(np.float32)
    cap = np.zeros((b, t), np.int32)
    for i in range(b):
        L = int(rng.integers(1, max(2, t - 1)))
        cap[i, 0] = 1
        cap[i, 1:1 + L] = rng.integers(3, V, size=min(L, t - 1))
        if 1 + L < t:
            cap[i, 1 + L] = 2
    tgt = np.zeros_like(cap)
    tgt[:, :-1] = cap[:, 1:]
    z = np.zeros((b, U), np.float32)
    return (x, cap, z, z.copy()), tgt


RATES = {"dense": (0.0, 0.2, 0.2), "attention": (0.0, 0.2, 0.2, 0.2, 0.2, 0.2)}       # AttemptFour/config.yaml:36-41


def make(kind, seed=42, rates=None, **kw):
    from masters_thesis_amd.optimizers import Adam
    rates = RATES[kind] if rates is None else rates
    if kind == "dense":
        from masters_thesis_amd.nic import NIC
        m = NIC(N, U, E, V, T, *rates, 0.01, 0.00003, 0.00001, seed=seed, **kw)
    else:
        from masters_thesis_amd.lc_nic import NIC, synthetic_groups
        m = NIC(synthetic_groups(N, 360, 32, seed=42), U, 512, E, 32, V, T, *rates, 0.01, 0.001,
                0.00003, 0.00001, seed=seed, **kw)
    m.compile(Adam(learning_rate=1e-4, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    return m


def make_oracle(kind, rates=None):
    rates = RATES[kind] if rates is None else rates
    if kind == "dense":
        return M.NICDense(N, U, E, V, T, *rates, 0.01, 0.00003, 0.00001)
    from masters_thesis_amd.lc_nic import synthetic_groups
    return M.LcNIC(synthetic_groups(N, 360, 32, seed=42), U, 512, E, 32, V, T, *rates, 0.01, 0.001, 0.00003, 0.00001)


@pytest.mark.parametrize("kind", ["dense", "attention"])
def test_eval_forward_matches_oracle_at_full_size(kind):
    rng = np.random.default_rng(7)
    model = make(kind)
    data, tgt = synth(rng)
    if kind == "dense":
        orc = M.NICDense(N, U, E, V, T, 0.0, 0.2, 0.2, 0.01, 0.00003, 0.00001)
    else:
        from masters_thesis_amd.lc_nic import synthetic_groups
        orc = M.LcNIC(synthetic_groups(N, 360, 32, seed=42), U, 512, E, 32, V, T, 0.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.01, 0.001,
                      0.00003, 0.00001)
    orc.p = {k: v.astype(np.float64) for k, v in model.get_weights_dict().items()}
    out = model(data, training=False)
    want = orc.forward(data, False)[0]
    p = (out[0] if isinstance(out, tuple) else out).cpu().numpy()
    probs = want[0] if isinstance(want, tuple) else want
    assert np.abs(p.sum(-1) - 1).max() < 1e-5 and np.isfinite(p).all()
    # 1e-4 on the logits == 1e-4 absolute on log-probabilities
    assert np.abs(np.log(p) - np.log(probs)).max() <= 1e-4 * max(1.0, np.abs(np.log(probs)).max())
    if isinstance(out, tuple):
        assert np.abs(out[1].cpu().numpy() - want[1]).max() <= 1e-5


@pytest.mark.parametrize("kind", ["dense", "attention"])
def test_training_step_properties_at_full_size(kind, tmp_path):
    rng = np.random.default_rng(8)
    data, tgt = synth(rng)
    runs = []
    for use_graph in (True, True, False):            # two captured runs + one eager run, same seed
        model = make(kind, use_graph=use_graph)
        hist = [model.train_step((data, tgt)).as_floats() for _ in range(5)]
        runs.append((hist, {k: v.copy() for k, v in model.get_weights_dict().items()}, model))
    (h0, w0, m0), (h1, w1, _), (h2, w2, _) = runs
    assert h0 == h1 and all(np.array_equal(w0[k], w1[k]) for k in w0), "run-to-run determinism"
    assert h0 == h2 and all(np.array_equal(w0[k], w2[k]) for k in w0), "hipGraph replay == eager launches"
    assert all(np.isfinite(list(h.values())).all() for h in h0)
    # loss goes down on a fixed batch (dropout on, so compare a later average with the first steps)
    more = [m0.train_step((data, tgt)).as_floats()["loss"] for _ in range(60)]
    assert np.mean(more[-10:]) < np.mean([h["loss"] for h in h0[:3]])
    # weights round trip -> identical inference
    path = str(tmp_path / "w.npz")
    m0.save_weights(path)
    m2 = make(kind, seed=1)
    m2.load_weights(path, by_name=True, skip_mismatch=True)
    a, b = m0(data, training=False), m2(data, training=False)
    a, b = (a[0], b[0]) if isinstance(a, tuple) else (a, b)
    assert torch.equal(a, b)


@pytest.mark.parametrize("kind", ["dense", "attention"])
@pytest.mark.parametrize("b,t", [(1, 15), (64, 1), (1, 1), (130, 3)])
def test_edge_shapes(kind, b, t):
    rng = np.random.default_rng(9)
    model = make(kind)
    data, tgt = synth(rng, b, t)
    r = model.train_step((data, tgt)).as_floats()
    assert np.isfinite(list(r.values())).all()
    out = model(data, training=False)
    p = (out[0] if isinstance(out, tuple) else out).cpu().numpy()
    assert p.shape == (b, t, V) and np.abs(p.sum(-1) - 1).max() < 1e-5
    z = np.zeros((b, U), np.float32)
    g = model.greedy_predict(data[0], z, z, np.ones(b, np.int64), 4, U, None)
    words = g[0] if isinstance(g, tuple) else g.argmax(-1)
    assert np.asarray(words).size == b * 4


def test_lc_seq_fwd_chain_equals_step_kernels():
    """tnt_lc_seq_fwd_f32 (the T attention -> LSTM steps of config 3 as one persistent launch, the default) against the per-step
    kernels on the same model: dropout on (stored keep masks, context input dropout), captured training steps -- same
    losses, attention metric and weights up to float32 rounding of the gate math, error word 0; and inference outputs."""
    rng = np.random.default_rng(11)
    data, tgt = synth(rng)
    a, b = make("attention"), make("attention")
    b.use_lc_seq = False
    ha = [a.train_step((data, tgt)).as_floats() for _ in range(4)]
    hb = [b.train_step((data, tgt)).as_floats() for _ in range(4)]
    if not a._lc_seq_ok():
        pytest.skip("persistent chain kernel not supported on this device")
    assert not b._lc_seq_ok()
    a.check_device_errors()
    for x, y in zip(ha, hb):
        for k in x:
            assert abs(x[k] - y[k]) <= 2e-5 * max(1.0, abs(y[k])), (k, x, y)
    wa, wb = a.get_weights_dict(), b.get_weights_dict()
    for k in wa:
        d = np.abs(wa[k] - wb[k])
        assert (d > 3e-5).mean() <= 5e-3, (k, d.max())          # see the LeakyReLU-kink note in the test below
    pa, aa = a(data, training=False)
    pb, ab = b(data, training=False)
    assert (pa - pb).abs().max().item() <= 2e-4 and (aa - ab).abs().max().item() <= 2e-5


def test_persistent_lstm_forward_trains_like_the_step_kernels():
    """Config 2 with the persistent sequence kernel (tnt_lstm_seq_fwd_f32: one launch for the T+1 dependent LSTM steps)
    against the same model on the per-step kernels: same losses and weights over captured training steps (the two
    kernels differ by float32 rounding in the gate math only), and no barrier timeout."""
    rng = np.random.default_rng(9)
    data, tgt = synth(rng)
    a, b = make("dense"), make("dense")
    b.use_seq_lstm = False
    ha = [a.train_step((data, tgt)).as_floats() for _ in range(6)]
    hb = [b.train_step((data, tgt)).as_floats() for _ in range(6)]
    if not a._seq_lstm:
        pytest.skip("persistent LSTM kernel not supported on this device")
    assert not b._seq_lstm
    a.check_device_errors()
    for x, y in zip(ha, hb):
        assert abs(x["loss"] - y["loss"]) <= 2e-5 * abs(y["loss"]), (x, y)
        assert abs(x["accuracy"] - y["accuracy"]) <= 2.0 / (B * T)
    wa, wb = a.get_weights_dict(), b.get_weights_dict()
    for k in wa:
        # Adam moves every weight by ~lr per step whatever the gradient scale, so rounding noise shows up as a small
        # fraction of the 6 * lr = 6e-4 the weights travelled.  One exception is legitimate: an encoder pre-activation
        # within rounding of 0 can take the other LeakyReLU slope in one of the two models (1 of 32768 elements in a
        # step, seen at step 2 with this seed), which changes that unit's bias and its one kernel column (0.2 % of the
        # tensor) by a real amount -- so the bound is on all but 0.5 % of each tensor.
        d = np.abs(wa[k] - wb[k])
        assert (d > 3e-5).mean() <= 5e-3, (k, d.max(), (d > 3e-5).mean())
    # falling back at run time (what bench.py does after a barrier timeout): graphs are re-captured on the step kernels
    a.disable_seq_lstm()
    for _ in range(3):
        x, y = a.train_step((data, tgt)).as_floats(), b.train_step((data, tgt)).as_floats()
        assert abs(x["loss"] - y["loss"]) <= 5e-5 * abs(y["loss"]), (x, y)
    assert not a._seq_lstm
    a.check_device_errors()


def _sync_oracle(model, orc, opt, step):
    """The oracle starts the step from exactly the model's state: weights, BatchNorm statistics, Adam moments, t."""
    orc.p = {k: v.astype(np.float64) for k, v in model.get_weights_dict().items()}
    if step > 0:
        for k in opt.m:
            opt.m[k] = model.get_optimizer_slot(k, "m").astype(np.float64)
            opt.v[k] = model.get_optimizer_slot(k, "v").astype(np.float64)
    opt.t = step


def _kink_flips(kind, model, w0, data, step, rates):
    """LeakyReLU elements whose float32 pre-activation has the other sign than the float64 oracle's (|pre| ~ 1e-8: a handful
    of the 1.7 M pre-activations of a step land that close to zero every few steps).  The derivative jumps from 1 to 0.2
    there, so the two sides legitimately back-propagate different values through that one element.
    -> ({oracle pre-activation shape: boolean mask of the flipped elements, in the oracle's own layout}, count)."""
    orc2 = make_oracle(kind, rates)
    orc2.p = w0
    _, cache = orc2.forward(data, training=True, drop=M.DropCtx(seed=model.seed, step=step, training=True))
    if kind == "dense":
        pairs = [(model.enc_pre, cache["pre"], None)]
    else:      # the model keeps the head's pre-activation time-major, the oracle batch-major
        pairs = [(model.Ppre, cache["Ppre"], None), (model.ipre, cache["ipre"], (0, 1)), (model.enc_pre, cache["enc"]["pre"], None)]
    masks, flips = {}, 0
    for mine, ref, swap in pairs:
        ref = np.asarray(ref)
        view = np.swapaxes(ref, *swap) if swap else ref
        a = mine.detach().cpu().numpy().reshape(-1)
        assert a.size == view.size
        bad = (a.reshape(view.shape) > 0) != (view > 0)
        assert np.abs(view[bad]).max(initial=0.0) < 1e-6, "pre-activation sign differs away from zero"
        if bad.any():
            native = np.ascontiguousarray(np.swapaxes(bad, *swap) if swap else bad)
            assert native.shape == ref.shape and ref.shape not in masks
            masks[ref.shape] = native
            flips += int(bad.sum())
    return masks, flips


@pytest.mark.parametrize("kind", ["dense", "attention"])
@pytest.mark.parametrize("dropout", [True, False])
def test_train_step_matches_oracle_at_full_size(kind, dropout):
    """ONE training step at BASELINE size against the float64 oracle, three times over (eager launch sequence, the
    captured hipGraph's first replay, a later replay), each from the model's own state: loss / accuracy / L2
    [/ attention], EVERY gradient (NIC.py:248-249, lc_NIC.py:386-387), the BatchNorm moving statistics and the
    post-Adam weights (main.py:97, lc_NIC.py:389).  This is the only place where the kernels that exist only at full
    size -- the persistent LSTM forward (U == 512), lstm_bwd_lds (4U % 1024 == 0), the one-round head GEMM, the skinny
    encoder dW, the in-kernel-reduced gradient GEMMs -- run inside one oracle-checked step.
    A gradient outside the 1e-4 bound is accepted only if the step is shown to contain a LeakyReLU kink flip (_kink_flips)
    AND the oracle re-run with those elements on the model's side of the kink matches every tensor at 1e-4.
    The weights are checked twice: loosely against the oracle's own update (Adam turns a 1e-4 gradient error on a
    near-zero element into a fraction of lr), and tightly (1e-3 of one update) against the float64 Adam formulas
    applied to the gradients the model itself produced, which pins clip-by-norm + Adam + the IndexedSlices norm."""
    rng = np.random.default_rng(21)
    rates = None if dropout else tuple(0.0 for _ in RATES[kind])
    model = make(kind, rates=rates)
    orc = make_oracle(kind, rates)
    orc.p = {k: v.astype(np.float64) for k, v in model.get_weights_dict().items()}
    names = [k for k in orc.p if "moving_" not in k]
    lam = {k: model.arena.entries[k].l2 for k in names}
    lr = 1e-4
    opt = M.AdamState({k: orc.p[k] for k in names}, lr=lr, b1=0.9, b2=0.98, eps=1e-8, clipnorm=0.1)
    for step in range(3):
        data, tgt = synth(rng)
        _sync_oracle(model, orc, opt, step)
        w0 = {k: v.copy() for k, v in orc.p.items()}
        m0 = {k: v.copy() for k, v in opt.m.items()}
        v0 = {k: v.copy() for k, v in opt.v.items()}
        res, grads, _ = orc.train_step(data, tgt, opt, M.DropCtx(seed=model.seed, step=step, training=True))
        got = model.train_step((data, tgt)).as_floats()
        model.check_device_errors()
        for k in res:
            if k == "lr":
                continue
            tol = 1e-6 if k == "accuracy" else 1e-4 * abs(res[k]) + 1e-7
            assert abs(got[k] - res[k]) <= tol, (step, k, got[k], res[k])
        gm = {k: model.get_gradient(k).astype(np.float64) + 2 * lam[k] * w0[k] for k in names}

        def grad_errors(ref):
            out = {}
            for k in names:
                if ref.get(k) is None or k == "attention/V/bias":
                    continue
                scale = np.abs(ref[k]).max()
                err = np.abs(gm[k] - ref[k]).max()
                if err > 1e-4 * scale + 1e-10:
                    out[k] = (err, scale)
            return out
        if "attention/V/bias" in gm:               # softmax is shift-invariant: the true gradient is 0
            assert np.abs(gm["attention/V/bias"]).max() < 1e-5
        bad = grad_errors(grads)
        if bad:
            # A gradient outside 1e-4 is accepted on ONE condition: the step contains LeakyReLU elements whose float32
            # pre-activation sits on the other side of the kink (|pre| < 1e-6, checked), and the float64 oracle RE-RUN from the
            # same state with exactly those elements forced onto the model's side (oracle.ops.KINK_FLIPS) agrees with every
            # gradient tensor at the usual 1e-4 -- no tensor gets a looser bound, flipped step or not.
            masks, kinks = _kink_flips(kind, model, w0, data, step, rates)
            assert 0 < kinks <= 3, (step, bad, kinks)
            orc.p = {k: v.copy() for k, v in w0.items()}
            opt.m, opt.v, opt.t = {k: v.copy() for k, v in m0.items()}, {k: v.copy() for k, v in v0.items()}, step
            O.KINK_FLIPS.update(masks)
            try:
                res2, grads, _ = orc.train_step(data, tgt, opt, M.DropCtx(seed=model.seed, step=step, training=True))
            finally:
                O.KINK_FLIPS.clear()
            for k in res:
                if k != "lr":
                    assert abs(res2[k] - res[k]) <= 1e-7 * abs(res[k]) + 1e-10, (step, k)    # the forward value does not move
            bad = grad_errors(grads)
            assert not bad, (step, "after forcing the kink flips", kinks, bad)
        # BatchNorm moving statistics
        for k in orc.p:
            if "moving_" in k:
                w = model.get_weight(k)
                assert np.abs(w - orc.p[k]).max() <= 1e-5 * max(1.0, np.abs(orc.p[k]).max()), (step, k)
        # (a) against the oracle's own update
        for k in names:
            if k == "attention/V/bias":
                continue
            w, v = model.get_weight(k), orc.p[k]
            noisy = np.abs(grads[k]) < 1e-2 * np.abs(grads[k]).max()
            tol = 2e-2 * lr + 1e-6 * np.abs(v).max() + 2.2 * lr * noisy
            assert (np.abs(w - v) <= tol).all(), (step, k, np.abs(w - v).max())
        # (b) the float64 Adam formulas on the model's own gradients
        sparse = orc.last_sparse
        p2 = {k: w0[k].copy() for k in names}
        opt2 = M.AdamState(p2, lr=lr, b1=0.9, b2=0.98, eps=1e-8, clipnorm=0.1)
        opt2.m, opt2.v, opt2.t = m0, v0, step
        opt2.apply(p2, gm, sparse)
        for k in names:
            if k == "attention/V/bias":
                continue
            w = model.get_weight(k)
            tol = 1e-3 * lr + 3e-7 * np.abs(p2[k])
            assert (np.abs(w - p2[k]) <= tol).all(), (step, k, np.abs(w - p2[k]).max())


@pytest.mark.parametrize("kind", ["dense", "attention"])
def test_test_step_matches_oracle_at_full_size(kind):
    """NIC.test_step (NIC.py:254-299) / lc_NIC.test_step (lc_NIC.py:410-459) at BASELINE size: inference-mode forward
    (BatchNorm on the moving statistics, no dropout), loss / accuracy / L2 [/ attention] against the oracle --
    eager, captured, replayed -- after two training steps have moved the weights and the moving statistics."""
    rng = np.random.default_rng(22)
    model = make(kind)
    orc = make_oracle(kind)
    for _ in range(2):
        model.train_step(synth(rng))
    for rep in range(3):
        data, tgt = synth(rng)
        orc.p = {k: v.astype(np.float64) for k, v in model.get_weights_dict().items()}
        want = orc.test_step(data, tgt)[0]
        got = model.test_step((data, tgt)).as_floats()
        assert set(got) == set(want), (sorted(got), sorted(want))
        for k in want:
            tol = 1e-6 if k == "accuracy" else 1e-4 * abs(want[k]) + 1e-7
            assert abs(got[k] - want[k]) <= tol, (rep, k, got[k], want[k])
    model.check_device_errors()


def test_guard_trip_skips_the_update_and_falls_back():
    """Fault injection (the persistent LSTM kernel's error word pre-set, as a barrier timeout would leave it): the
    step's metrics raise DeviceGuardError, the optimizer kernels have skipped -- weights, Adam moments, t and the
    dropout stream are bit-identical to before the step -- the model has fallen back to the per-step kernels, and
    running the step again gives what a model that never used the persistent kernel gives."""
    from masters_thesis_amd.model_base import DeviceGuardError
    rng = np.random.default_rng(23)
    data, tgt = synth(rng)
    a, b = make("dense"), make("dense")
    b.use_seq_lstm = False
    for _ in range(3):                       # eager, capture, replay
        a.train_step((data, tgt)).as_floats(); b.train_step((data, tgt)).as_floats()
    if not a._seq_lstm:
        pytest.skip("persistent LSTM kernel not supported on this device")
    torch.cuda.synchronize()
    before = (a.arena.theta.clone(), a.opt_m.clone(), a.opt_v.clone(), a.adam_t.clone(), a.drop_step.clone())
    a.seq_sync[1024] = 1
    res = a.train_step((data, tgt))
    with pytest.raises(DeviceGuardError):
        res.as_floats()
    after = (a.arena.theta, a.opt_m, a.opt_v, a.adam_t, a.drop_step)
    assert all(torch.equal(x, y) for x, y in zip(before, after)), "a guarded step must leave the model untouched"
    assert not a._seq_lstm and int(a.seq_sync[1024]) == 0
    ra, rb = a.train_step((data, tgt)).as_floats(), b.train_step((data, tgt)).as_floats()
    assert abs(ra["loss"] - rb["loss"]) <= 5e-5 * abs(rb["loss"]), (ra, rb)
    # fit() redoes the step by itself
    c = make("dense")
    c.train_step((data, tgt)).as_floats()
    c.seq_sync[1024] = 1
    hist = c.fit([(data, tgt)] * 2, epochs=1, verbose=0)
    assert np.isfinite(hist["loss"][0]) and not c._seq_lstm
    # inference paths check after their own host read and re-run on the per-step kernels
    d = make("dense")
    p0 = d(data, training=False)
    d.seq_sync[1024] = 1
    p1 = d(data, training=False)
    assert not d._seq_lstm and (p0 - p1).abs().max().item() <= 1e-5


def test_guard_trip_of_the_attention_chains_skips_the_update_and_falls_back():
    """The same fault injection for the attention model, whose forward AND backward T-step chains are persistent launches
    (tnt_lc_seq_fwd_f32 / tnt_lc_seq_bwd_f32) sharing the dense model's sync state and guard: the step's metrics raise, the
    update was skipped, the model continues on the per-step attention / LSTM launches and matches a model that never
    used the chain kernels."""
    from masters_thesis_amd.model_base import DeviceGuardError
    rng = np.random.default_rng(29)
    data, tgt = synth(rng)
    a, b = make("attention"), make("attention")
    b.use_lc_seq = False
    for _ in range(3):
        a.train_step((data, tgt)).as_floats(); b.train_step((data, tgt)).as_floats()
    if not a._lc_seq_ok():
        pytest.skip("persistent chain kernels not supported on this device")
    torch.cuda.synchronize()
    before = (a.arena.theta.clone(), a.opt_m.clone(), a.opt_v.clone(), a.adam_t.clone(), a.drop_step.clone())
    a.seq_sync[1024] = 1
    res = a.train_step((data, tgt))
    with pytest.raises(DeviceGuardError):
        res.as_floats()
    after = (a.arena.theta, a.opt_m, a.opt_v, a.adam_t, a.drop_step)
    assert all(torch.equal(x, y) for x, y in zip(before, after)), "a guarded step must leave the model untouched"
    assert not a._lc_seq_ok() and int(a.seq_sync[1024]) == 0
    ra, rb = a.train_step((data, tgt)).as_floats(), b.train_step((data, tgt)).as_floats()
    for k in ("loss", "attention"):
        assert abs(ra[k] - rb[k]) <= 5e-5 * max(1.0, abs(rb[k])), (k, ra, rb)


@pytest.mark.parametrize("R,D,A,Bq", [(400, 32, 32, 20), (100, 32, 48, 64), (360, 32, 32, 70)])
def test_attention_chain_kernels_off_the_benchmark_shape(R, D, A, Bq):
    """The one-launch chains away from config 3's shape: R = 400 (the backward chain's LDS limit sends it to the per-step
    launches while the forward chain runs), attention width 48 (the 16-lanes-per-row variants), a ragged batch of 70
    (16-row blocks, the last one partly empty) -- each against the same model on the per-step launches, dropout on."""
    from masters_thesis_amd.lc_nic import NIC, synthetic_groups
    from masters_thesis_amd.optimizers import Adam

    def build():
        m = NIC(synthetic_groups(6000, R, D, seed=1), U, 512, E, A, V, T, 0.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.01, 0.001, 0.00003,
                0.00001, seed=42)
        m.compile(Adam(learning_rate=1e-4, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
        return m
    rng = np.random.default_rng(R + A)
    (x, cap, z, c), tgt = synth(rng, b=Bq)
    data = (x[:, :6000].copy(), cap, z, c)
    a, b = build(), build()
    b.use_lc_seq = False
    ha = [a.train_step((data, tgt)).as_floats() for _ in range(3)]
    hb = [b.train_step((data, tgt)).as_floats() for _ in range(3)]
    if not a._lc_seq_ok():
        pytest.skip("persistent chain kernels not supported on this device")
    a.check_device_errors()
    for s_, (p, q) in enumerate(zip(ha, hb)):
        for k in ("loss", "accuracy", "attention"):
            assert abs(p[k] - q[k]) <= 3e-5 * max(1.0, abs(q[k])), (s_, k, p, q)
    wa, wb = a.get_weights_dict(), b.get_weights_dict()
    for k in wa:
        d = np.abs(wa[k] - wb[k])
        assert (d > 3e-5).mean() <= 5e-3, (k, d.max())
