"""ThinkAndTell / ShowAndTell caption generators: host orchestration vs the oracle (mock backend)."""
import numpy as np
import pytest

import masters_thesis_amd.ops as ops
from masters_thesis_amd import think_and_tell as TT, show_and_tell as SAT
from masters_thesis_amd.optimizers import Adam, SGD
from oracle import models as M
from oracle.models_tt import CaptionGeneratorTT
from mock_backend import MockBackend


@pytest.fixture(autouse=True)
def mock_backend():
    old = ops._backend
    ops.set_backend(MockBackend())
    yield
    ops.set_backend(old)


def batch(rng, B, N, T, V):
    x = rng.standard_normal((B, N)).astype(np.float32)
    tgt = rng.integers(1, V, (B, T)).astype(np.int32)
    for b in range(B):
        tgt[b, rng.integers(2, T + 1):] = 0
    return x, tgt


def make(rng, sat, l2=0.01, drop=0.0, B=4, N=19, E=10, U=16, V=13, T=5):
    orc = CaptionGeneratorTT(N, E, U, V, T, l2_reg=l2, dropout=drop, show_and_tell=sat).init_params(rng)
    if sat:
        model = SAT.CaptionGenerator(SAT.Encoder(E), SAT.Decoder(E, U, V), None, T, device="cpu", seed=11)
    else:
        model = TT.CaptionGenerator(TT.Encoder(E, l2, "glorot_uniform", drop), TT.Decoder(E, U, V, l2, "glorot_uniform", drop),
                                    None, T, device="cpu", seed=11)
    return model, orc, (B, N, T, V)


@pytest.mark.parametrize("sat,drop", [(False, 0.0), (False, 0.3), (True, 0.0)])
def test_train_test_match_oracle(sat, drop):
    rng = np.random.default_rng(81)
    model, orc, (B, N, T, V) = make(rng, sat, drop=drop)
    model.compile(Adam(learning_rate=1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=None))
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=None)
    for step in range(3):
        x, tgt = batch(rng, B, N, T, V)
        if step == 0:
            model._stage(x, tgt)                      # creates the parameters (N known at first batch)
            for k, v in orc.p.items():
                model.set_weight(k, v)
        res, grads = orc.train_step(x, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        data = (x, tgt) if sat else (x, None, tgt)
        got = model.train_step(data).as_floats()
        assert set(got) == set(res)
        for k in res:
            assert abs(got[k] - res[k]) < 3e-5 * max(1, abs(res[k])), (step, k, got[k], res[k])
        for k, v in orc.p.items():
            assert np.allclose(model.get_weight(k), v, rtol=2e-4, atol=3e-6), (step, k)
    x, tgt = batch(rng, B, N, T, V)
    res = orc.test_step(x, tgt)
    got = model.test_step((x, tgt) if sat else (x, None, tgt)).as_floats()
    for k in res:
        assert abs(got[k] - res[k]) < 3e-5 * max(1, abs(res[k])), k
    lg = model((x, tgt))
    assert tuple(lg.shape) == (B, T + 1, V)
    assert np.allclose(lg.numpy(), orc.forward(x, tgt, False)[0], rtol=1e-4, atol=1e-5)


def test_sam_step_matches_oracle():
    rng = np.random.default_rng(82)
    model, orc, (B, N, T, V) = make(rng, False, drop=0.2)
    model.compile(SGD(learning_rate=0.05, momentum=0.9))
    x, tgt = batch(rng, B, N, T, V)
    model._stage(x, tgt)
    for k, v in orc.p.items():
        model.set_weight(k, v)

    class SgdState:
        def __init__(s, p):
            s.mom = {k: np.zeros_like(v) for k, v in p.items()}

        def apply(s, p, g, sparse=None):
            from oracle import ops as O
            for k in g:
                p[k], s.mom[k] = O.sgd_momentum_update(p[k], s.mom[k], g[k], 0.05, 0.9)
    opt = SgdState(orc.p)
    for step in range(2):
        x, tgt = batch(rng, B, N, T, V)
        res, _ = orc.train_step_sam(x, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        got = model.train_step_SAM((x, None, tgt)).as_floats()
        for k in res:
            assert abs(got[k] - res[k]) < 5e-5 * max(1, abs(res[k])), (step, k, got[k], res[k])
        for k, v in orc.p.items():
            assert np.allclose(model.get_weight(k), v, rtol=3e-4, atol=5e-6), (step, k)


def test_simple_eval_samples_from_the_logits():
    """evaluate.simple_eval (ThinkAndTell/evaluate.py:261-284): teacher-forced forward + one categorical draw per position."""
    from masters_thesis_amd.evaluate import simple_eval
    from oracle import ops as O
    rng = np.random.default_rng(85)
    model, orc, (B, N, T, V) = make(rng, False)
    x, tgt = batch(rng, B, N, T, V)
    model._stage(x, tgt)
    for k, v in orc.p.items():
        model.set_weight(k, v)
    ids, caps = simple_eval(model, x, tgt, None, temperature=0.9, sample_step=2)
    logits, _ = orc.forward(x, tgt, False)
    want, margin = O.sample_rows(logits.reshape(B * (T + 1), V), 0.9, True, model.seed, M.S_SAMPLE, 2)
    assert caps is None and ids.shape == (B, T + 1)
    bad = ids.reshape(-1) != want
    assert np.all(margin[bad] < 1e-5)
