"""Model-level parity on the GPU: fully-connected mode (lc_NIC.call_fc / greedy_predict_fc) through
the HIP kernels against oracle/models_fc.py (float64)."""
import numpy as np
import pytest

from oracle import models as M
from oracle import models_fc as MF
from helpers import synth_batch

pytestmark = pytest.mark.gpu

L2 = {"dense_in/kernel": 0.01, "lstm/kernel": 3e-5, "time_distributed_nonlinear/kernel": 1e-5,
      "time_distributed_softmax/kernel": 1e-5}


def build(rng, rates, dims, use_graph=True):
    from masters_thesis_amd.fc_nic import NICfc
    B, N, T, V, U, E = dims
    args = (N, U, E, E, V, T) + tuple(rates) + (0.01, 3e-5, 1e-5)
    model = NICfc(*args, seed=11, use_graph=use_graph)
    orc = MF.FcNIC(*args).init_params(rng)
    for k, v in orc.p.items():
        model.set_weight(k, v)
    return model, orc


DIMS = [(3, 37, 4, 11, 16, 8), (8, 2000, 15, 501, 64, 64)]


@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("rates", [(0, 0, 0, 0, 0), (0.1, 0.2, 0.1, 0.2, 0.3)])
def test_train_parity(dims, rates):
    from masters_thesis_amd.optimizers import Adam
    rng = np.random.default_rng(71)
    B, N, T, V, U, E = dims
    model, orc = build(rng, rates, dims)
    model.compile(Adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=0.1)
    for step in range(4):                      # eager, capture, replay, replay
        data, tgt = synth_batch(B, N, T, V, U, rng, zero_first=(step == 2))
        res, grads, probs = orc.train_step(data, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        got = model.train_step((data, tgt)).as_floats()
        assert abs(got["loss"] - res["loss"]) <= 1e-4 * abs(res["loss"]), (step, got, res)
        assert abs(got["accuracy"] - res["accuracy"]) < 1e-6
        assert abs(got["L2"] - res["L2"]) <= 1e-4 * abs(res["L2"])
        for k, v in orc.p.items():
            w = model.get_weight(k)
            assert np.abs(w - v).max() <= 2e-2 * 1e-3 + 1e-4 * np.abs(v).max(), (step, k, np.abs(w - v).max())


@pytest.mark.parametrize("dims", DIMS)
def test_forward_gradients_greedy(dims):
    from masters_thesis_amd.optimizers import Adam
    rng = np.random.default_rng(72)
    B, N, T, V, U, E = dims
    model, orc = build(rng, (0, 0, 0, 0, 0), dims, use_graph=False)
    model.compile(Adam(1e-4, clipnorm=None))
    data, tgt = synth_batch(B, N, T, V, U, rng)
    probs, cache = orc.forward(data, False)
    p, attn = model(data, training=False)
    assert attn is None
    assert np.abs(np.log(p.cpu().numpy()) - np.log(probs)).max() <= 1e-4 * np.abs(cache["logits"]).max()
    w0 = {k: v.copy() for k, v in orc.p.items()}
    z = np.zeros((B, U), np.float32)
    want = orc.greedy_predict(data[0], z, z, np.ones(B, np.int64), T)
    got = model.greedy_predict(data[0], z, z, np.ones(B, np.int64), T, U)
    assert got.shape == (T, B, 1) and np.array_equal(got, want)
    probs, cache = orc.forward(data, True, M.DropCtx(training=True))
    grads, _ = orc.backward(probs, cache, tgt)
    model.train_step((data, tgt))
    for k in orc.TRAINABLE:
        g = model.get_gradient(k) + 2 * L2.get(k, 0.0) * w0[k]
        want_g = grads[k] if grads[k] is not None else np.zeros_like(w0[k])
        assert np.abs(g - want_g).max() <= 1e-4 * np.abs(want_g).max() + 1e-9, k


def test_full_width_forward_and_persistent_lstm():
    """call_fc at the BASELINE widths (B = 64, U = E = 512, V = 5001, T = 15; N shortened to 5000): eval probabilities
    against the float64 oracle, and -- where the device supports it -- the persistent sequence kernel
    (tnt_lstm_seq_fwd_f32, T masked steps in one launch) against the per-step kernels over captured training steps."""
    from masters_thesis_amd.optimizers import Adam
    dims = (64, 5000, 15, 5001, 512, 512)
    B, N, T, V, U, E = dims
    rng = np.random.default_rng(73)
    a, orc = build(rng, (0, 0, 0, 0, 0), dims)
    b, _ = build(np.random.default_rng(73), (0, 0, 0, 0, 0), dims)
    for k, v in orc.p.items():
        b.set_weight(k, v)
    b.use_seq_lstm = False
    for m in (a, b):
        m.compile(Adam(learning_rate=1e-4, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    data, tgt = synth_batch(B, N, T, V, U, rng)
    probs, cache = orc.forward(data, False)
    p, _ = a(data, training=False)
    assert np.abs(np.log(p.cpu().numpy()) - np.log(probs)).max() <= 1e-4 * np.abs(cache["logits"]).max()
    for step in range(4):
        ra, rb = a.train_step((data, tgt)).as_floats(), b.train_step((data, tgt)).as_floats()
        assert abs(ra["loss"] - rb["loss"]) <= 2e-5 * abs(rb["loss"]), (step, ra, rb)
    a.check_device_errors()
    assert not b._seq_lstm
    if a._seq_lstm:                       # rounding-level differences only: Adam moves a weight by ~lr per step whatever the
        for k in orc.p:                   # gradient's size, so a near-zero gradient element whose rounding differs shows up as
            d = np.abs(a.get_weight(k) - b.get_weight(k))      # a fraction of 4 lr on that element (same rule as the other
            assert (d > 3e-5).mean() <= 5e-3 and d.max() <= 5e-4, (k, d.max())      # chain-vs-step tests)
