"""BASELINE.json's full sizes (config 2: N=20000, U=E=512, V=5001, T=15, B=64; config 3: 360 regions x 32 + attention)
on the GPU: one eval forward against the float64 oracle (it finishes in seconds at this size), and size-independent
properties of the training step -- bit-exact run-to-run determinism (no atomics anywhere on the path), hipGraph replay ==
eager launches, probabilities on the simplex, loss going down on a fixed batch, weights round trip, and the edge shapes
B = 1 / T = 1."""
import numpy as np
import pytest
import torch

from oracle import models as M

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


def make(kind, seed=42, **kw):
    from masters_thesis_amd.optimizers import Adam
    if kind == "dense":
        from masters_thesis_amd.nic import NIC
        m = NIC(N, U, E, V, T, 0.0, 0.2, 0.2, 0.01, 0.00003, 0.00001, seed=seed, **kw)
    else:
        from masters_thesis_amd.lc_nic import NIC, synthetic_groups
        m = NIC(synthetic_groups(N, 360, 32, seed=42), U, 512, E, 32, V, T, 0.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.01, 0.001,
                0.00003, 0.00001, seed=seed, **kw)
    m.compile(Adam(learning_rate=1e-4, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    return m


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
        # fraction of the 6 * lr = 6e-4 the weights travelled
        assert np.abs(wa[k] - wb[k]).max() <= 3e-5, (k, np.abs(wa[k] - wb[k]).max())
    # falling back at run time (what bench.py does after a barrier timeout): graphs are re-captured on the step kernels
    a.disable_seq_lstm()
    for _ in range(3):
        x, y = a.train_step((data, tgt)).as_floats(), b.train_step((data, tgt)).as_floats()
        assert abs(x["loss"] - y["loss"]) <= 5e-5 * abs(y["loss"]), (x, y)
    assert not a._seq_lstm
    a.check_device_errors()
