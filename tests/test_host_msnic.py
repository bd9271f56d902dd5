"""Multi-subject model (ms2_NIC generalised to S subjects): oracle pins (finite differences) and
host orchestration against the oracle on CPU through the mock backend."""
import numpy as np
import pytest

import masters_thesis_amd.ops as ops
from masters_thesis_amd.ms_nic import NIC
from masters_thesis_amd.optimizers import Adam
from oracle import models as M
from oracle.models_ms import MsLcNIC
from helpers import synth_batch, tiny_groups
from mock_backend import MockBackend


@pytest.fixture(autouse=True)
def mock_backend():
    old = ops._backend
    ops.set_backend(MockBackend())
    yield
    ops.set_backend(old)


D = dict(Bs=3, N=41, R=4, D=16, A=5, U=16, Et=10, V=13, T=4)


def make(rng, rates, S, norm="batch"):
    g = (tiny_groups(D["N"], D["R"], rng), [D["D"]] * D["R"])
    args = (g, D["U"], 512, D["Et"], D["A"], D["V"], D["T"], *rates, 0.01, 0.001, 3e-5, 1e-5)
    orc = MsLcNIC(*args, n_subjects=S, norm=norm).init_params(rng)
    model = NIC(*args, n_subjects=S, norm=norm, device="cpu", seed=11)
    for k, v in orc.p.items():
        model.set_weight(k, v)
    return model, orc


def test_oracle_ms_backward_fd():
    rng = np.random.default_rng(60)
    _, orc = make(rng, (0.1, 0.2, 0.2, 0.2, 0.2, 0.2), 2)
    data, tgt = synth_batch(2 * D["Bs"], D["N"], D["T"], D["V"], D["U"], rng, dtype=np.float64)
    drop = M.DropCtx(seed=3, step=0, training=True)
    (probs, attn), cache = orc.forward(data, True, drop)
    grads, _ = orc.backward(probs, cache, tgt)

    def loss():
        (p, a), _ = orc.forward(data, True, drop)
        return orc.metrics_ms(p, a, tgt)["loss"] + orc.l2_loss()
    for k in ["dense_in_1/2/kernel", "dense_in_0/1/bias", "input_bn_1/gamma", "attention/W1/kernel", "lstm/bias"]:
        w = orc.p[k]
        for idx in [tuple(rng.integers(0, n) for n in w.shape) for _ in range(3)]:
            old = w[idx]
            w[idx] = old + 1e-6; lp = loss()
            w[idx] = old - 1e-6; lm = loss()
            w[idx] = old
            assert abs((lp - lm) / 2e-6 - grads[k][idx]) < 1e-6 * max(1, abs(grads[k][idx])), (k, idx)


@pytest.mark.parametrize("S,rates", [(2, (0,) * 6), (2, (0.1, 0.2, 0.2, 0.2, 0.2, 0.2)), (3, (0, 0.2, 0, 0, 0, 0))])
def test_train_and_test_step_match_oracle(S, rates):
    rng = np.random.default_rng(61)
    model, orc = make(rng, rates, S)
    model.compile(Adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=0.1)
    B = S * D["Bs"]
    step = 0
    for it in range(2):
        data, tgt = synth_batch(B, D["N"], D["T"], D["V"], D["U"], rng)
        res, _, _ = orc.train_step(data, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        got = model.train_step((data, tgt)).as_floats()
        step += 1
        assert set(got) == set(res)
        for k in res:
            assert abs(got[k] - res[k]) < 3e-5 * max(1, abs(res[k])), (it, k, got[k], res[k])
        for k, v in orc.p.items():
            atol = 3e-3 * (it + 1) if k == "attention/V/bias" else 3e-6
            assert np.allclose(model.get_weight(k), v, rtol=2e-4, atol=atol), (it, k)
    # test_step: sub-calls run in training mode (dropout on, BN batch statistics) -- quirk kept
    data, tgt = synth_batch(B, D["N"], D["T"], D["V"], D["U"], rng)
    res, _ = orc.test_step(data, tgt, M.DropCtx(seed=11, step=step, training=True))
    got = model.test_step((data, tgt)).as_floats()
    for k in res:
        assert abs(got[k] - res[k]) < 3e-5 * max(1, abs(res[k])), (k, got[k], res[k])
    outs = model((data[0], data[1], data[2], data[3]))
    assert len(outs) == 2 * S and tuple(outs[0].shape) == (D["Bs"], D["T"], D["V"])
