"""Host orchestration of NIC (config 2) against the model-level oracle, on CPU through the
mock backend (tests/mock_backend.py): launch order, buffer wiring, gate-interleaved / padded
layouts, dropout stream wiring, masked LSTM, clip + Adam plumbing, greedy decode."""
import numpy as np
import pytest
import torch

import masters_thesis_amd.ops as ops
from masters_thesis_amd.nic import NIC
from masters_thesis_amd.optimizers import Adam, SGD, CategoricalCrossentropy
from oracle import models as M
from helpers import synth_batch
from mock_backend import MockBackend


@pytest.fixture(autouse=True)
def mock_backend():
    old = ops._backend
    ops.set_backend(MockBackend())
    yield
    ops.set_backend(old)


def make_pair(rng, rates, norm="batch", B=5, N=23, T=6, V=13, U=16, E=10, seed=11):
    model = NIC(N, U, E, V, T, rates[0], rates[1], rates[2], 0.01, 3e-5, 1e-5, norm=norm, device="cpu", seed=seed)
    orc = M.NICDense(N, U, E, V, T, rates[0], rates[1], rates[2], 0.01, 3e-5, 1e-5, norm=norm).init_params(rng)
    for k, v in orc.p.items():
        model.set_weight(k, v)
        assert np.allclose(model.get_weight(k), v, atol=1e-6)
    return model, orc


def onehot(ids, V):
    oh = np.zeros(ids.shape + (V,), np.float32)
    np.put_along_axis(oh, ids[..., None], 1.0, -1)
    return oh


@pytest.mark.parametrize("rates,norm,zero_first", [((0, 0, 0), "batch", False), ((0.1, 0.2, 0.2), "batch", True),
                                                   ((0, 0.2, 0), "layer", False)])
def test_train_steps_match_oracle(rates, norm, zero_first):
    rng = np.random.default_rng(21)
    B, N, T, V, U = 5, 23, 6, 13, 16
    model, orc = make_pair(rng, rates, norm)
    model.compile(Adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1),
                  CategoricalCrossentropy(from_logits=False, reduction="none"))
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=0.1)
    for step in range(3):
        data, tgt = synth_batch(B, N, T, V, U, rng, zero_first=zero_first)
        res, grads, probs = orc.train_step(data, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        target = onehot(tgt, V) if step != 1 else tgt          # both target forms
        got = model.train_step((data, target)).as_floats()
        assert abs(got["loss"] - res["loss"]) < 2e-5 * max(1, abs(res["loss"]))
        assert abs(got["accuracy"] - res["accuracy"]) < 1e-6
        assert abs(got["L2"] - res["L2"]) < 1e-5 * max(1, abs(res["L2"]))
        for k in orc.TRAINABLE:
            lam = {"dense_img/kernel": 0.01, "lstm/kernel": 3e-5, "time_distributed_softmax/kernel": 3e-5}.get(k, 0.0)
            # gradient buffers hold the data gradient; the oracle's includes 2*lambda*W (pre-update W)
            g = model.get_gradient(k)
            assert g.shape == grads[k].shape
        for k, v in orc.p.items():
            w = model.get_weight(k)
            assert np.allclose(w, v, rtol=2e-4, atol=2e-6), (step, k, np.abs(w - v).max())
    assert model.optimizer.iterations == 3


@pytest.mark.parametrize("N", [23, 32])        # 32: the norm comes from the Gram by-products of the forward (N % 16 == 0)
def test_fused_encoder_update_matches_oracle(N):
    """E = 512: the single-process step consumes the encoder kernel's gradient inside the optimizer launches
    (dense_dw_sqnorm + dense_dw_adam, no dW buffer write).  Weights after 3 steps against the oracle, the same model with
    the gradient written out, and the on-request gradient."""
    rng = np.random.default_rng(23)
    B, T, V, U, E = 5, 4, 13, 16, 512
    model, orc = make_pair(rng, (0, 0.2, 0.2), B=B, N=N, T=T, V=V, U=U, E=E)
    plain, _ = make_pair(np.random.default_rng(23), (0, 0.2, 0.2), B=B, N=N, T=T, V=V, U=U, E=E)
    plain.fuse_enc_update = False
    for m in (model, plain):
        m.compile(Adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=0.1)
    for step in range(3):
        data, tgt = synth_batch(B, N, T, V, U, rng)
        w0 = orc.p["dense_img/kernel"].copy()
        res, grads, _ = orc.train_step(data, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        got, ref = model.train_step((data, tgt)).as_floats(), plain.train_step((data, tgt)).as_floats()
        assert model._enc_grad_stale is not None and plain._enc_grad_stale is None          # the fused path ran
        assert (getattr(model, "_enc_gram", None) is not None) == (N % 16 == 0)
        assert abs(got["loss"] - res["loss"]) < 2e-5 * max(1, abs(res["loss"]))
        assert abs(got["L2"] - ref["L2"]) < 1e-6 * max(1, abs(ref["L2"]))
        g = model.get_gradient("dense_img/kernel") + 2 * 0.01 * w0
        if step == 0:       # later steps: the two weight sets have drifted by rounding, the gradients with them
            assert np.allclose(g, grads["dense_img/kernel"], rtol=1e-4, atol=1e-6 * np.abs(g).max() + 1e-9)
        assert np.allclose(g, plain.get_gradient("dense_img/kernel") + 2 * 0.01 * w0, rtol=1e-5, atol=1e-8)
        for k, v in orc.p.items():
            w = model.get_weight(k)
            # the encoder bias gradient is ~1e-8 here (BatchNorm behind it removes most of it): at Adam's epsilon scale,
            # where float32 rounding of the column sums moves the update by percents of lr
            assert np.allclose(w, v, rtol=2e-4, atol=2e-5 if k == "dense_img/bias" else 2e-6), (step, k, np.abs(w - v).max())
            assert np.allclose(w, plain.get_weight(k), rtol=1e-5, atol=1e-7), (step, k)


def test_gradients_match_oracle():
    rng = np.random.default_rng(22)
    B, N, T, V, U = 4, 19, 5, 11, 16
    model, orc = make_pair(rng, (0, 0, 0), B=B, N=N, T=T, V=V, U=U)
    model.compile(Adam(1e-4, clipnorm=None))
    w0 = {k: v.copy() for k, v in orc.p.items()}
    data, tgt = synth_batch(B, N, T, V, U, rng)
    probs, cache = orc.forward(data, True, M.DropCtx(training=True))
    grads, _ = orc.backward(probs, cache, tgt)
    model.train_step((data, tgt))
    lam = {"dense_img/kernel": 0.01, "lstm/kernel": 3e-5, "time_distributed_softmax/kernel": 3e-5}
    for k in orc.TRAINABLE:
        g = model.get_gradient(k) + 2 * lam.get(k, 0.0) * w0[k]
        assert np.allclose(g, grads[k], rtol=1e-4, atol=1e-6 * np.abs(grads[k]).max() + 1e-9), k


def test_test_step_call_and_greedy():
    rng = np.random.default_rng(23)
    B, N, T, V, U = 4, 19, 5, 11, 16
    model, orc = make_pair(rng, (0.1, 0.2, 0.2), B=B, N=N, T=T, V=V, U=U)
    model.compile(SGD(learning_rate=0.01, momentum=0.9))
    data, tgt = synth_batch(B, N, T, V, U, rng)
    res, probs = orc.test_step(data, tgt)
    got = model.test_step((data, onehot(tgt, V))).as_floats()
    assert abs(got["loss"] - res["loss"]) < 2e-5 and abs(got["accuracy"] - res["accuracy"]) < 1e-6
    p = model(data, training=False)
    assert tuple(p.shape) == (B, T, V)
    assert np.allclose(p.numpy(), probs, rtol=1e-4, atol=1e-6)
    # greedy; force id 0 for one sample so the masked-step branch is exercised
    orc.p["time_distributed_softmax/bias"][0] = 3.0
    model.set_weight("time_distributed_softmax/bias", orc.p["time_distributed_softmax/bias"])
    z = np.zeros((B, U), np.float32)
    want = orc.greedy_predict(data[0], z, z, np.ones(B, np.int64), T)
    gotp = model.greedy_predict(data[0], z, z, np.ones(B, np.int64), T, U)
    assert gotp.shape == want.shape == (T, B, 1, V)
    assert np.array_equal(gotp.argmax(-1), want.argmax(-1))
    assert np.allclose(gotp, want, rtol=1e-4, atol=1e-6)


def test_sgd_and_weights_io(tmp_path):
    rng = np.random.default_rng(24)
    B, N, T, V, U = 4, 19, 5, 11, 16
    model, orc = make_pair(rng, (0, 0, 0), B=B, N=N, T=T, V=V, U=U)
    model.compile(SGD(learning_rate=0.01, momentum=0.9))
    data, tgt = synth_batch(B, N, T, V, U, rng)
    probs, cache = orc.forward(data, True, M.DropCtx(training=True))
    grads, _ = orc.backward(probs, cache, tgt)
    model.train_step((data, tgt))
    for k in orc.TRAINABLE:
        assert np.allclose(model.get_weight(k), orc.p[k] - 0.01 * grads[k], rtol=1e-4, atol=1e-6), k
    path = str(tmp_path / "w.npz")
    model.save_weights(path)
    m2 = NIC(N, U, 10, V, T, 0, 0, 0, 0.01, 3e-5, 1e-5, device="cpu")
    m2.load_weights(path, by_name=True, skip_mismatch=True)
    for k in model.keras_shapes:
        assert np.array_equal(m2.get_weight(k), model.get_weight(k)), k
    lw = model.get_layer("lstm").get_weights()
    assert [w.shape for w in lw] == [(10, 64), (16, 64), (64,)]
    m2.get_layer("lstm").set_weights([w * 2 for w in lw])
    assert np.allclose(m2.get_weight("lstm/kernel"), 2 * lw[0])
    with pytest.raises(ValueError):
        model.get_layer("nope")


def test_device_resident_batch_is_staged_in_one_launch():
    """torch tensors on the model's device in the staged dtypes take tnt_stage_batch_f32; results equal the
    general per-tensor path (numpy inputs)."""
    rng = np.random.default_rng(25)
    B, N, T, V, U = 4, 19, 5, 11, 16
    m1, orc = make_pair(rng, (0, 0, 0), B=B, N=N, T=T, V=V, U=U)
    m2, _ = make_pair(np.random.default_rng(25), (0, 0, 0), B=B, N=N, T=T, V=V, U=U)
    for m in (m1, m2):
        m.compile(Adam(1e-3, clipnorm=0.1))
    data, tgt = synth_batch(B, N, T, V, U, rng)
    calls = []
    be = ops.backend()
    orig = be.stage_batch
    be.stage_batch = lambda *a: (calls.append(1), orig(*a))[1]
    r1 = m1.train_step((data, tgt)).as_floats()                                     # numpy -> general path
    assert not calls
    tdata = tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in data)
    r2 = m2.train_step((tdata, torch.from_numpy(tgt.astype(np.int32)))).as_floats()  # tensors -> one launch
    assert calls == [1]
    assert r1 == r2
    for k in m1.keras_shapes:
        assert np.array_equal(m1.get_weight(k), m2.get_weight(k)), k


def test_half_precision_betas_on_the_wire():
    """float16 betas (data.PinnedPrefetcher(betas_dtype="float16")) are widened by the staging launch: the step equals
    the step on the float32 image of the same half values."""
    rng = np.random.default_rng(26)
    B, N, T, V, U = 4, 20, 5, 11, 16
    m1, _ = make_pair(rng, (0, 0, 0), B=B, N=N, T=T, V=V, U=U)
    m2, _ = make_pair(np.random.default_rng(26), (0, 0, 0), B=B, N=N, T=T, V=V, U=U)
    for m in (m1, m2):
        m.compile(Adam(1e-3, clipnorm=0.1))
    data, tgt = synth_batch(B, N, T, V, U, rng)
    xh = torch.from_numpy(np.ascontiguousarray(data[0])).half()
    rest = tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in data[1:])
    tt = torch.from_numpy(tgt.astype(np.int32))
    r1 = m1.train_step(((xh,) + rest, tt)).as_floats()
    r2 = m2.train_step(((xh.float(),) + rest, tt)).as_floats()
    assert r1 == r2
    for k in m1.keras_shapes:
        assert np.array_equal(m1.get_weight(k), m2.get_weight(k)), k


def test_adaptive_gradient_clipping_matches_oracle():
    """model.enable_agc == agc.adaptive_clip_grad before apply_gradients (agc.py:20-38, call site lc_NIC.py:388):
    unit-wise norms per output column, the Embedding through its un-deduplicated IndexedSlices rows."""
    rng = np.random.default_rng(29)
    B, N, T, V, U = 5, 23, 6, 13, 16
    model, orc = make_pair(rng, (0, 0.2, 0.2))
    model.compile(Adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    model.enable_agc(0.02, 1e-3)
    orc.agc = (0.02, 1e-3)
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=0.1)
    lam = {"dense_img/kernel": 0.01, "lstm/kernel": 3e-5, "time_distributed_softmax/kernel": 3e-5}
    for step in range(3):
        data, tgt = synth_batch(B, N, T, V, U, rng)
        w0 = {k: v.copy() for k, v in orc.p.items()}
        raw, _ = orc.backward(*(lambda pc: (pc[0], pc[1], tgt))(orc.forward(data, True, M.DropCtx(seed=11, step=step, training=True))))
        res, grads, _ = orc.train_step(data, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        model.train_step((data, tgt)).as_floats()
        changed = 0
        for k in orc.TRAINABLE:
            g = model.get_gradient(k) + 2 * lam.get(k, 0.0) * w0[k]
            assert np.allclose(g, grads[k], rtol=2e-4, atol=1e-5 * np.abs(grads[k]).max() + 1e-9), (step, k)
            changed += int(not np.allclose(raw[k], grads[k], rtol=1e-9, atol=0))
        assert 0 < changed, "the clip never triggered: the test exercises nothing"
        for k, v in orc.p.items():
            assert np.allclose(model.get_weight(k), v, rtol=2e-4, atol=2e-6), (step, k)


def test_metrics_ring_keeps_each_steps_values_until_its_row_is_reused():
    """The fused step files its metrics vector in a ring (tnt_adam_ring_f32) instead of cloning it behind
    every step: Metrics objects of earlier steps keep THEIR values while later steps run, test_step still clones, and a
    Metrics object read after its row was reused says so instead of returning another step's numbers."""
    rng = np.random.default_rng(3)
    B, N, T, V, U = 5, 23, 6, 13, 16
    model, orc = make_pair(rng, (0, 0, 0))
    model.METRIC_RING = 4
    model.compile(Adam(learning_rate=1e-2, clipnorm=0.1), CategoricalCrossentropy(from_logits=False, reduction="none"))
    ref, _ = make_pair(np.random.default_rng(3), (0, 0, 0))
    ref.metric_ring = False
    ref.compile(Adam(learning_rate=1e-2, clipnorm=0.1), CategoricalCrossentropy(from_logits=False, reduction="none"))
    for k, v in model.get_weights_dict().items():
        ref.set_weight(k, v)
    held, want = [], []
    for step in range(6):
        data, tgt = synth_batch(B, N, T, V, U, rng)
        held.append(model.train_step((data, tgt)))
        want.append(ref.train_step((data, tgt)).as_floats())
        if step == 2:
            t1, t2 = model.test_step((data, tgt)).as_floats(), ref.test_step((data, tgt)).as_floats()
            assert t1 == t2
    assert model.met_ring.shape == (4, model.met.numel() + 1) and ref.__dict__.get("met_ring") is None
    for step in (2, 3, 4, 5):                # the four newest rows are intact, whatever ran since
        assert held[step].as_floats() == want[step]
    for step in (0, 1):                      # rows reused by steps 4 and 5
        with pytest.raises(RuntimeError, match="ring row has been reused"):
            held[step].as_floats()
