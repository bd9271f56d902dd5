"""ThinkAndTell/att_model.py generator (GRU decoder): oracle self-check (finite differences) and host
orchestration vs the oracle on the mock backend."""
import numpy as np
import pytest

import masters_thesis_amd.ops as ops
from masters_thesis_amd import think_and_tell_att as ATT
from masters_thesis_amd.optimizers import Adam
from oracle import models as M
from oracle import ops as O
from oracle.models_att import CaptionGeneratorAtt, TRAINABLE
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


def test_gru_cell_matches_torch_and_backward_is_the_gradient():
    """keras GRU v2 (reset_after=True) == torch.nn.GRUCell with gate order (r, z, n) permuted to (z, r, h)."""
    import torch
    rng = np.random.default_rng(5)
    B, E, U = 3, 5, 4
    x, h = rng.standard_normal((B, E)), rng.standard_normal((B, U))
    W, Uk, b = rng.standard_normal((E, 3 * U)), rng.standard_normal((U, 3 * U)), rng.standard_normal((2, 3 * U))
    h2, cache = O.gru_step_fwd(x @ W + b[0], h, Uk, b[1])
    cell = torch.nn.GRUCell(E, U).double()
    perm = lambda a: np.concatenate([a[..., U:2 * U], a[..., :U], a[..., 2 * U:]], axis=-1)     # (z,r,h) -> (r,z,n)
    with torch.no_grad():
        cell.weight_ih.copy_(torch.tensor(perm(W).T)); cell.weight_hh.copy_(torch.tensor(perm(Uk).T))
        cell.bias_ih.copy_(torch.tensor(perm(b[0]))); cell.bias_hh.copy_(torch.tensor(perm(b[1])))
    ht = torch.tensor(h, requires_grad=True)
    xt = torch.tensor(x, requires_grad=True)
    out = cell(xt, ht)
    assert np.allclose(out.detach().numpy(), h2, atol=1e-12)
    dout = rng.standard_normal((B, U))
    out.backward(torch.tensor(dout))
    dxz, drec, dh = O.gru_step_bwd(dout, cache, Uk)
    assert np.allclose(dh, ht.grad.numpy(), atol=1e-12)
    assert np.allclose(dxz @ W.T, xt.grad.numpy(), atol=1e-12)
    assert np.allclose(perm(x.T @ dxz), cell.weight_ih.grad.numpy().T, atol=1e-12)
    assert np.allclose(perm(h.T @ drec), cell.weight_hh.grad.numpy().T, atol=1e-12)
    assert np.allclose(perm(drec.sum(0)), cell.bias_hh.grad.numpy(), atol=1e-12)


def test_oracle_gradient_by_finite_differences():
    rng = np.random.default_rng(6)
    B, N, E, U, V, T = 3, 7, 5, 4, 9, 5
    orc = CaptionGeneratorAtt(N, E, U, V, T, l2_reg=0.01, dropout=0.0).init_params(rng)
    x, tgt = batch(rng, B, N, T, V)
    x = x.astype(np.float64)
    logits, cache = orc.forward(x, tgt, True)
    g = orc.backward(logits, cache)
    f = lambda: orc.loss(orc.forward(x, tgt, True)[0], tgt) + orc.l2_loss()
    for k in TRAINABLE:
        w = orc.p[k]
        for _ in range(4):
            idx = tuple(rng.integers(0, s) for s in w.shape)
            old = w[idx]
            w[idx] = old + 1e-6; fp = f()
            w[idx] = old - 1e-6; fm = f()
            w[idx] = old
            assert abs((fp - fm) / 2e-6 - g[k][idx]) < 1e-6 + 1e-5 * abs(g[k][idx]), (k, idx)


@pytest.mark.parametrize("drop,clip", [(0.0, None), (0.3, 0.1)])
def test_train_test_match_oracle(drop, clip):
    rng = np.random.default_rng(82)
    B, N, E, U, V, T = 4, 19, 12, 16, 13, 6
    orc = CaptionGeneratorAtt(N, E, U, V, T, l2_reg=0.01, dropout=drop).init_params(rng)
    model = ATT.CaptionGenerator(ATT.Encoder(E, 0.01, "glorot_uniform", drop), ATT.Decoder(E, U, V, 0.01, "glorot_uniform", drop),
                                 None, T, device="cpu", seed=11)
    model.compile(Adam(learning_rate=1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=clip))
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=clip)
    for step in range(3):
        x, tgt = batch(rng, B, N, T, V)
        if step == 0:
            model._stage(x, tgt)
            for k, v in orc.p.items():
                model.set_weight(k, v)
                assert np.allclose(model.get_weight(k), v, atol=1e-6)
        res, grads = orc.train_step(x, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        got = model.train_step((x, None, tgt)).as_floats()
        assert set(got) == set(res)
        for k in res:
            assert abs(got[k] - res[k]) < 3e-5 * max(1, abs(res[k])), (step, k, got[k], res[k])
        for k, v in orc.p.items():
            assert np.allclose(model.get_weight(k), v, rtol=2e-4, atol=3e-6), (step, k)
    x, tgt = batch(rng, B, N, T, V)
    res = orc.test_step(x, tgt)
    got = model.test_step((x, None, tgt)).as_floats()
    for k in res:
        assert abs(got[k] - res[k]) < 3e-5 * max(1, abs(res[k])), k
    logits = model((x, None, tgt), training=False)
    want, _ = orc.forward(x, tgt, False)
    assert tuple(logits.shape) == (B, T + 1, V) and np.allclose(logits.numpy(), want, rtol=1e-4, atol=1e-5)
    with pytest.raises(NotImplementedError):
        model.train_step_SAM((x, None, tgt))
