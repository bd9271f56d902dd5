"""Pins for the CPU oracle (PARITY UNPINNED by the reference -- SURVEY 8c):
closed forms, finite differences, and an independent torch-autograd composition."""
import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import models as M
from oracle.philox import keep_mask, philox4x32_10
from helpers import tiny_groups, synth_batch


def test_cce_closed_form_temp_py():
    # AttemptFour/temp.py:73-90: -ln 0.6 and -ln 0.93
    x1 = np.array([[0.1, 0.05, 0.05, 0.2, 0.6]])
    x2 = np.array([[0.01, 0.02, 0.03, 0.01, 0.93]])
    y = np.array([4])
    assert abs(O.cce_from_probs(x1, y)[0] - 0.5108256) < 1e-6
    assert abs(O.cce_from_probs(x2, y)[0] - 0.0725707) < 1e-6


def test_cce_clip_path():
    p = np.array([[1 - 1e-9, 1e-9]])
    assert abs(O.cce_from_probs(p, np.array([1]))[0] + np.log(1e-7)) < 1e-9
    # clip active -> zero gradient
    g = O.cce_softmax_bwd(p, np.array([1]), np.array([1.0]))
    assert np.all(g == 0)


def test_argmax_tie_first():
    p = np.array([[0.3, 0.3, 0.2, 0.2]])
    assert O.accuracy(p, np.array([0])) == 1.0
    assert O.accuracy(p, np.array([1])) == 0.0


def test_philox_known_answer():
    # Random123 KAT: philox4x32-10, counter = key = 0
    r = philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(v[0]) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = philox4x32_10([0xffffffff], [0xffffffff], [0xffffffff], [0xffffffff], 0xffffffff, 0xffffffff)
    assert [int(v[0]) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = philox4x32_10([0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344], 0xa4093822, 0x299f31d0)
    assert [int(v[0]) for v in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_keep_mask_rate():
    m = keep_mask((200, 500), 0.2, seed=42, site=3, step=7)
    assert abs(m.mean() - 0.8) < 0.005
    assert (keep_mask((200, 500), 0.2, 42, 3, 7) == m).all()
    assert (keep_mask((200, 500), 0.2, 42, 3, 8) != m).any()


def _fd(fn, x, eps=1e-6):
    g = np.zeros_like(x)
    it = np.nditer(x, flags=['multi_index'])
    for _ in it:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps
        fp = fn()
        x[i] = old - eps
        fm = fn()
        x[i] = old
        g[i] = (fp - fm) / (2 * eps)
    return g


def test_batchnorm_bwd_fd():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 4, 5))
    gam, bet = rng.standard_normal(5), rng.standard_normal(5)
    w = rng.standard_normal((3, 4, 5))

    def f():
        return (O.batchnorm_fwd(x, gam, bet, np.zeros(5), np.ones(5), True)[0] * w).sum()
    y, cache, mm, mv = O.batchnorm_fwd(x, gam, bet, np.zeros(5), np.ones(5), True)
    dx, dg, db = O.batchnorm_bwd(w, gam, cache)
    assert np.allclose(dx, _fd(f, x), atol=1e-6)
    assert np.allclose(dg, _fd(f, gam), atol=1e-6)
    assert np.allclose(mm, 0.01 * x.mean(axis=(0, 1)))
    assert np.allclose(mv, 0.99 + 0.01 * x.var(axis=(0, 1)))


def test_attention_bwd_fd():
    rng = np.random.default_rng(1)
    B, R, D, A, U = 2, 4, 3, 3, 5
    F = rng.standard_normal((B, R, D)); h = rng.standard_normal((B, U))
    W1 = rng.standard_normal((D, A)); b1 = rng.standard_normal(A)
    W2 = rng.standard_normal((U, A)); b2 = rng.standard_normal(A)
    v = rng.standard_normal((A, 1)); bv = rng.standard_normal(1)
    w = rng.standard_normal((B, D))
    keep = rng.random((B, R, A)) > 0.3

    def f():
        P, _ = O.attention_proj_fwd(F, W1, b1)
        (ctx, _, _), _ = O.attention_step_fwd(h, F, P, W2, b2, v, bv, keep, 0.3)
        return (ctx * w).sum()
    P, Ppre = O.attention_proj_fwd(F, W1, b1)
    (ctx, al, sd), cache = O.attention_step_fwd(h, F, P, W2, b2, v, bv, keep, 0.3)
    dh, dF, dsum, dW2, db2, dv, dbv = O.attention_step_bwd(w, F, W2, v, cache)
    dF1, dW1, db1 = O.dense_bwd(F, W1, Ppre, dsum, O.ACT_LEAKY)
    assert np.allclose(dh, _fd(f, h), atol=1e-6)
    assert np.allclose(dF + dF1, _fd(f, F), atol=1e-6)
    assert np.allclose(dW1, _fd(f, W1), atol=1e-6)
    assert np.allclose(dW2, _fd(f, W2), atol=1e-6)
    assert np.allclose(dv, _fd(f, v), atol=1e-6)
    assert np.allclose(dbv, _fd(f, bv), atol=1e-6)
    assert np.allclose(al.sum(axis=1), 1)


# ----------------------------------------------------------------- torch autograd cross-check
def _t(a, req=True):
    return torch.tensor(a, dtype=torch.float64, requires_grad=req)


def _leaky(x):
    return torch.where(x > 0, x, 0.2 * x)


def _lstm_t(xz, h, c, U_):
    Un = h.shape[1]
    z = xz + h @ U_
    i, f = torch.sigmoid(z[:, :Un]), torch.sigmoid(z[:, Un:2 * Un])
    g, o = torch.tanh(z[:, 2 * Un:3 * Un]), torch.sigmoid(z[:, 3 * Un:])
    c2 = f * c + i * g
    return o * torch.tanh(c2), c2


def _cce_t(probs, y):
    q = probs / probs.sum(-1, keepdim=True)
    q = torch.clamp(q, 1e-7, 1 - 1e-7)
    return -torch.log(torch.gather(q, -1, y[..., None])[..., 0])


def _bn_t(x, gam, bet):
    red = tuple(range(x.dim() - 1))
    mean = x.mean(dim=red)
    var = ((x - mean) ** 2).mean(dim=red)
    return (x - mean) / torch.sqrt(var + 1e-3) * gam + bet


def _drop_t(x, keep, rate):
    if keep is None:
        return x
    return torch.where(torch.tensor(keep), x / (1 - float(np.float32(rate))), torch.zeros_like(x))


def torch_nicdense_loss(model, data, y_ids, drop):
    """Independent torch restatement of NIC.py:100-145 + the loss of NIC.py:233-246."""
    x, ids, a0, c0 = data
    P = {k: _t(v, 'moving' not in k) for k, v in model.p.items()}
    B, T = ids.shape
    xd = _drop_t(_t(x, False), drop.mask(x.shape, model.r_in, M.S_IN), model.r_in)
    y = _leaky(xd @ P['dense_img/kernel'] + P['dense_img/bias'])
    yd = _drop_t(y, drop.mask(y.shape, model.r_feat, M.S_FEAT), model.r_feat)
    f = _bn_t(yd, P['batch_norm/gamma'], P['batch_norm/beta'])
    emb = P['emb_text/embeddings'][torch.tensor(ids, dtype=torch.long)]
    Wl, Ul, bl = P['lstm/kernel'], P['lstm/recurrent_kernel'], P['lstm/bias']
    fd = _drop_t(f[:, None, :], drop.mask((B, 1, model.E), model.r_lstm, M.S_LSTM_IN), model.r_lstm)[:, 0]
    a, c = _lstm_t(fd @ Wl + bl, _t(a0, False), _t(c0, False), Ul)
    embd = _drop_t(emb, drop.mask((B, T, model.E), model.r_lstm, M.S_LSTM_IN + 1), model.r_lstm)
    out_prev = torch.zeros(B, model.U, dtype=torch.float64)
    outs = []
    for t in range(T):
        h2, c2 = _lstm_t(embd[:, t] @ Wl + bl, a, c, Ul)
        m = torch.tensor(ids[:, t] != 0)[:, None]
        a, c = torch.where(m, h2, a), torch.where(m, c2, c)
        out_prev = torch.where(m, h2, out_prev)
        outs.append(out_prev)
    A = torch.stack(outs, 1)
    probs = torch.softmax(A @ P['time_distributed_softmax/kernel'] + P['time_distributed_softmax/bias'], -1)
    yt = torch.tensor(y_ids, dtype=torch.long)
    ce = sum(_cce_t(probs[:, t], yt[:, t]).mean() for t in range(T)) / T
    l2 = (model.l2_in * (P['dense_img/kernel'] ** 2).sum() + model.l2_lstm * (Wl ** 2).sum()
          + model.l2_out * (P['time_distributed_softmax/kernel'] ** 2).sum())
    return ce + l2, probs, P


@pytest.mark.parametrize("rates,zero_first", [((0, 0, 0), False), ((0.1, 0.2, 0.2), True)])
def test_nicdense_grads_vs_torch(rates, zero_first):
    rng = np.random.default_rng(3)
    B, N, T, V, U, E = 4, 19, 6, 11, 7, 5
    model = M.NICDense(N, U, E, V, T, rates[0], rates[1], rates[2], 0.01, 3e-5, 1e-5).init_params(rng)
    data, tgt = synth_batch(B, N, T, V, U, rng, dtype=np.float64, zero_first=zero_first)
    drop = M.DropCtx(seed=5, step=1, training=True)
    probs, cache = model.forward(data, True, drop)
    grads, sparse = model.backward(probs, cache, tgt)
    loss, probs_t, P = torch_nicdense_loss(model, data, tgt, drop)
    loss.backward()
    assert np.allclose(probs, probs_t.detach().numpy(), atol=1e-12)
    for k in model.TRAINABLE:
        assert np.allclose(grads[k], P[k].grad.numpy(), rtol=1e-8, atol=1e-12), k
    ce, acc = model.metrics(probs, tgt)
    assert abs(float(ce + model.l2_loss()) - float(loss)) < 1e-12


def torch_lcnic_loss(model, data, y_ids, drop):
    """Independent torch restatement of lc_NIC.py:223-263 (W1(features) recomputed per step,
    exactly as the reference does) + the loss of lc_NIC.py:370-383."""
    x, ids, a0, c0 = data
    P = {k: _t(v, 'moving' not in k) for k, v in model.p.items()}
    B, T = ids.shape
    R, D, A = model.R, model.D, model.A
    xd = _drop_t(_t(x, False), drop.mask(x.shape, model.r_in, M.S_IN), model.r_in)
    ys = [_leaky(xd[:, torch.tensor(g)] @ P[f'dense_in/{r}/kernel'] + P[f'dense_in/{r}/bias'])
          for r, g in enumerate(model.groups)]
    y = torch.stack(ys, 0).permute(1, 0, 2)
    bn = _bn_t(y, P['input_bn/gamma'], P['input_bn/beta'])
    F = _drop_t(bn, drop.mask((B, R, D), model.r_feat, M.S_FEAT), model.r_feat)
    emb = P['emb_text/embeddings'][torch.tensor(ids, dtype=torch.long)]
    text = _drop_t(emb, drop.mask(tuple(emb.shape), model.r_text, M.S_TEXT), model.r_text)
    a, c = _t(a0, False), _t(c0, False)
    outs, alphas = [], []
    for i in range(T):
        hid = a[:, None, :]
        s = torch.tanh(_leaky(F @ P['attention/W1/kernel'] + P['attention/W1/bias'])
                       + _leaky(hid @ P['attention/W2/kernel'] + P['attention/W2/bias']))
        s = _drop_t(s, drop.mask((B, R, A), model.r_attn, M.S_ATTN + i), model.r_attn)
        score = s @ P['attention/V/kernel'] + P['attention/V/bias']
        w = torch.softmax(score, dim=1)
        ctx = (w * F).sum(1)
        sample = torch.cat([ctx, text[:, i]], -1)
        sample = _drop_t(sample[:, None, :], drop.mask((B, 1, sample.shape[1]), model.r_lstm, M.S_LSTM_IN + i),
                         model.r_lstm)[:, 0]
        a, c = _lstm_t(sample @ P['lstm/kernel'] + P['lstm/bias'], a, c, P['lstm/recurrent_kernel'])
        outs.append(_drop_t(a, drop.mask((B, model.U), model.r_lstm, M.S_LSTM_OUT + i), model.r_lstm))
        alphas.append(w)
    Hs = torch.stack(outs, 1)
    inter = _leaky(Hs @ P['time_distributed_nonlinear/kernel'] + P['time_distributed_nonlinear/bias'])
    inter = _drop_t(inter, drop.mask(tuple(inter.shape), model.r_out, M.S_OUT), model.r_out)
    probs = torch.softmax(inter @ P['time_distributed_softmax/kernel'] + P['time_distributed_softmax/bias'], -1)
    yt = torch.tensor(y_ids, dtype=torch.long)
    ce = sum(_cce_t(probs[:, t], yt[:, t]).mean() for t in range(T)) / T
    l2 = sum(model.l2_in * (P[f'dense_in/{r}/kernel'] ** 2).sum() for r in range(R))
    l2 = l2 + model.l2_attn * ((P['attention/W1/kernel'] ** 2).sum() + (P['attention/W2/kernel'] ** 2).sum())
    l2 = l2 + model.l2_lstm * (P['lstm/kernel'] ** 2).sum()
    l2 = l2 + model.l2_out * ((P['time_distributed_nonlinear/kernel'] ** 2).sum()
                              + (P['time_distributed_softmax/kernel'] ** 2).sum())
    return ce + l2, probs, torch.stack(alphas, 0), P


@pytest.mark.parametrize("rates", [(0,) * 6, (0.1, 0.2, 0.2, 0.2, 0.2, 0.2)])
def test_lcnic_grads_vs_torch(rates):
    rng = np.random.default_rng(4)
    B, N, R, D, A, U, Et, V, T = 3, 37, 4, 5, 3, 8, 6, 11, 4
    groups = tiny_groups(N, R, rng)
    model = M.LcNIC((groups, [D] * R), U, 512, Et, A, V, T, *rates, 0.01, 0.001, 3e-5, 1e-5).init_params(rng)
    data, tgt = synth_batch(B, N, T, V, U, rng, dtype=np.float64)
    drop = M.DropCtx(seed=9, step=2, training=True)
    (probs, attn), cache = model.forward(data, True, drop)
    grads, sparse = model.backward(probs, cache, tgt)
    loss, probs_t, al_t, P = torch_lcnic_loss(model, data, tgt, drop)
    loss.backward()
    assert np.allclose(probs, probs_t.detach().numpy(), atol=1e-12)
    assert np.allclose(attn, al_t.detach().numpy(), atol=1e-12)
    for k in model.trainable():
        assert np.allclose(grads[k], P[k].grad.numpy(), rtol=1e-8, atol=1e-12), k


def test_adam_clip_and_sparse_norm():
    rng = np.random.default_rng(5)
    p = {'w': rng.standard_normal((4, 3)), 'e': rng.standard_normal((5, 2))}
    opt = M.AdamState(p, clipnorm=0.1)
    g = {'w': rng.standard_normal((4, 3)), 'e': rng.standard_normal((5, 2))}
    p0 = {k: v.copy() for k, v in p.items()}
    opt.apply(p, g, {'e': 10.0})
    # first Adam step: m/(sqrt(v)+eps) = g_c*(1-b1) / (|g_c|*sqrt(1-b2) + eps); lr_t = lr*sqrt(1-b2)/(1-b1)
    gc = g['w'] * 0.1 / max(np.linalg.norm(g['w']), 0.1)
    exp = p0['w'] - 1e-4 * np.sqrt(1 - 0.98) / (1 - 0.9) * (0.1 * gc) / (np.sqrt(0.02 * gc * gc) + 1e-8)
    assert np.allclose(p['w'], exp, rtol=1e-12)
    ge = g['e'] * 0.1 / 10.0
    exp = p0['e'] - 1e-4 * np.sqrt(1 - 0.98) / (1 - 0.9) * (0.1 * ge) / (np.sqrt(0.02 * ge * ge) + 1e-8)
    assert np.allclose(p['e'], exp, rtol=1e-12)


def test_greedy_shapes_and_masking():
    rng = np.random.default_rng(6)
    B, N, T, V, U, E = 3, 13, 5, 9, 6, 4
    model = M.NICDense(N, U, E, V, T, 0, 0, 0, 0.01, 3e-5, 1e-5).init_params(rng)
    # force id 0 to be the argmax for everything -> later steps are masked
    model.p['time_distributed_softmax/bias'][0] = 50.0
    x = rng.standard_normal((B, N))
    z = np.zeros((B, U))
    out = model.greedy_predict(x, z, z, np.ones(B, np.int64), T)
    assert out.shape == (T, B, 1, V)
    # masked steps: whole = zeros -> probs = softmax(bias)
    sb = O.softmax(model.p['time_distributed_softmax/bias'])
    assert np.allclose(out[2, :, 0], sb)


def test_kink_flip_hook_equals_the_other_side_of_the_kink():
    """oracle.ops.KINK_FLIPS (the checker-side hook tests/test_gpu_fullsize.py uses when a float32 pre-activation lands on
    the other side of LeakyReLU's kink): forcing an element with |pre| ~ 1e-9 onto the other slope gives exactly the gradients
    of the same layer with that pre-activation nudged across zero; elements away from zero are never touched."""
    from oracle import ops as O
    rng = np.random.default_rng(5)
    x = rng.standard_normal((6, 9))
    K = rng.standard_normal((9, 4))
    b = rng.standard_normal(4)
    pre0 = x @ K + b
    b[2] += 1e-9 - pre0[3, 2]                              # pre[3, 2] = +1e-9
    dy = rng.standard_normal((6, 4))
    y, pre = O.dense_fwd(x, K, b, O.ACT_LEAKY)
    assert 0 < pre[3, 2] < 1e-8
    base = O.dense_bwd(x, K, pre, dy, O.ACT_LEAKY)
    pre_neg = pre.copy(); pre_neg[3, 2] = -1e-9
    want = O.dense_bwd(x, K, pre_neg, dy, O.ACT_LEAKY)
    mask = np.zeros(pre.shape, bool); mask[3, 2] = True
    O.KINK_FLIPS[pre.shape] = mask
    try:
        got = O.dense_bwd(x, K, pre, dy, O.ACT_LEAKY)
        far = np.zeros(pre.shape, bool); far[0, 0] = True  # |pre| is O(1) there: the hook must refuse the whole mask
        O.KINK_FLIPS[pre.shape] = far
        untouched = O.dense_bwd(x, K, pre, dy, O.ACT_LEAKY)
    finally:
        O.KINK_FLIPS.clear()
    for g, w, b0 in zip(got, want, base):
        assert np.array_equal(g, w)
    assert not np.array_equal(got[1], base[1])
    for u, b0 in zip(untouched, base):
        assert np.array_equal(u, b0)
