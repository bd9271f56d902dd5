"""Op-level CPU oracle (numpy; dtype follows the inputs, use float64 for truth).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED.

Each function restates one reference op with Keras/TF semantics (SURVEY.md
section 9); the reference line each one follows is cited in its docstring
(paths relative to /root/reference).  Forward functions return what the
backward needs; backward functions are explicit formulas (verified against
finite differences and torch autograd in tests/).
"""
import numpy as np

ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_TANH = 0, 1, 2, 3
BN_EPS = 1e-3        # keras BatchNormalization / LayerNormalization default epsilon
BN_MOMENTUM = 0.99   # keras BatchNormalization default momentum
CCE_EPS = 1e-7       # keras backend.epsilon()


# ---------------------------------------------------------------- activations
# Checker-side hook (tests/test_gpu_fullsize.py): LeakyReLU elements whose pre-activation is within rounding of zero and
# whose float32 twin on the device landed on the other side of the kink.  {pre.shape: boolean mask}; a masked element
# takes the OTHER slope, and only if every masked pre-activation is indeed within 1e-6 of zero.  Empty outside that test.
KINK_FLIPS = {}


def _leaky_positive(pre):
    pos = pre > 0
    m = KINK_FLIPS.get(pre.shape)
    if m is not None and m.any() and (np.abs(pre[m]) < 1e-6).all():
        pos = pos ^ m
    return pos


def act_fwd(pre, act, slope=0.2):
    """LeakyReLU(0.2) instance used as activation: AttemptFour/Model/lc_NIC.py:87,98,142."""
    if act == ACT_NONE:
        return pre
    if act == ACT_LEAKY:
        return np.where(_leaky_positive(pre), pre, pre * slope)
    if act == ACT_RELU:
        return np.maximum(pre, 0)
    if act == ACT_TANH:
        return np.tanh(pre)
    raise ValueError(act)


def act_bwd(pre, dy, act, slope=0.2):
    if act == ACT_NONE:
        return dy
    if act == ACT_LEAKY:
        return np.where(_leaky_positive(pre), dy, dy * slope)
    if act == ACT_RELU:
        return np.where(pre > 0, dy, 0 * dy)
    if act == ACT_TANH:
        t = np.tanh(pre)
        return dy * (1 - t * t)
    raise ValueError(act)


# ---------------------------------------------------------------------- dense
def dense_fwd(x, K, b, act=ACT_NONE, slope=0.2):
    """keras Dense: y = act(x @ K + b), K is (in, out).  layers.py:33, attention.py:21-23,
    lc_NIC.py:140-157, NIC.py:64-69,92-96.  Returns (y, pre)."""
    pre = x @ K
    if b is not None:
        pre = pre + b
    return act_fwd(pre, act, slope), pre


def dense_bwd(x, K, pre, dy, act=ACT_NONE, slope=0.2, need_dx=True):
    dpre = act_bwd(pre, dy, act, slope)
    x2 = x.reshape(-1, x.shape[-1])
    d2 = dpre.reshape(-1, dpre.shape[-1])
    dK = x2.T @ d2
    db = d2.sum(axis=0)
    dx = (dpre @ K.T) if need_dx else None
    return dx, dK, db


# -------------------------------------------------------------------- dropout
def dropout_fwd(x, keep, rate):
    """keras Dropout in training: inverted dropout (lc_NIC.py:51-55).  ``keep`` is a
    boolean mask (oracle/philox.keep_mask); rate 0 or keep None = identity."""
    if keep is None or rate <= 0.0:
        return x
    scale = x.dtype.type(1.0) / (x.dtype.type(1.0) - x.dtype.type(np.float32(rate)))
    return np.where(keep, x * scale, x.dtype.type(0))


def dropout_bwd(dy, keep, rate):
    return dropout_fwd(dy, keep, rate)


# ------------------------------------------------------------- normalisation
def batchnorm_fwd(x, gamma, beta, mov_mean, mov_var, training, eps=BN_EPS, momentum=BN_MOMENTUM):
    """keras BatchNormalization (non-fused path, axis=-1): layers.py:40,50; NIC.py:62,128;
    fullyConnected.py:18,24.  Training: biased batch statistics over every axis but the
    last; moving <- moving*momentum + batch*(1-momentum).  Returns (y, cache, new_mm, new_mv)."""
    red = tuple(range(x.ndim - 1))
    if training:
        mean = x.mean(axis=red)
        var = ((x - mean) ** 2).mean(axis=red)
        new_mm = mov_mean * momentum + mean * (1 - momentum)
        new_mv = mov_var * momentum + var * (1 - momentum)
    else:
        mean, var = mov_mean, mov_var
        new_mm, new_mv = mov_mean, mov_var
    inv = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * inv
    y = xhat * gamma + beta
    return y, (xhat, inv, training), new_mm, new_mv


def batchnorm_bwd(dy, gamma, cache):
    xhat, inv, training = cache
    red = tuple(range(dy.ndim - 1))
    dgamma = (dy * xhat).sum(axis=red)
    dbeta = dy.sum(axis=red)
    dxhat = dy * gamma
    if training:
        n = dy.size // dy.shape[-1]
        dx = inv / n * (n * dxhat - dxhat.sum(axis=red) - xhat * (dxhat * xhat).sum(axis=red))
    else:
        dx = dxhat * inv
    return dx, dgamma, dbeta


def layernorm_fwd(x, gamma, beta, eps=BN_EPS):
    """keras LayerNormalization(axis=-1): the commented alternative at layers.py:41 and the
    encoder norm BASELINE.json names.  Returns (y, cache)."""
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    inv = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * inv
    return xhat * gamma + beta, (xhat, inv)


def layernorm_bwd(dy, gamma, cache):
    xhat, inv = cache
    red = tuple(range(dy.ndim - 1))
    dgamma = (dy * xhat).sum(axis=red)
    dbeta = dy.sum(axis=red)
    dxhat = dy * gamma
    n = dy.shape[-1]
    dx = inv / n * (n * dxhat - dxhat.sum(axis=-1, keepdims=True)
                    - xhat * (dxhat * xhat).sum(axis=-1, keepdims=True))
    return dx, dgamma, dbeta


# ------------------------------------------------------------ locally dense
def locally_dense_fwd(x, groups, Ws, bs, slope=0.2):
    """layers.LocallyDense.call (layers.py:43-48) up to the stack/transpose:
    y[:, r, :] = LeakyReLU(x[:, idx_r] @ W_r + b_r).  Groups are ragged and may overlap
    (layers.py:13).  Returns (y (B,R,D), pre (B,R,D))."""
    B = x.shape[0]
    R = len(groups)
    D = Ws[0].shape[1]
    pre = np.empty((B, R, D), dtype=x.dtype)
    for r, idx in enumerate(groups):
        pre[:, r, :] = x[:, idx] @ Ws[r] + bs[r]
    return act_fwd(pre, ACT_LEAKY, slope), pre


def locally_dense_bwd(x, groups, pre, dy, slope=0.2):
    dpre = act_bwd(pre, dy, ACT_LEAKY, slope)
    dWs, dbs = [], []
    for r, idx in enumerate(groups):
        dWs.append(x[:, idx].T @ dpre[:, r, :])
        dbs.append(dpre[:, r, :].sum(axis=0))
    return dWs, dbs


# ------------------------------------------------------------------ embedding
def embedding_fwd(table, ids):
    """keras Embedding(mask_zero=True) is a plain row gather (lc_NIC.py:105-112,233)."""
    return table[ids]


def embedding_bwd_rows(dy, ids):
    """IndexedSlices gradient: one row per (b,t) occurrence, un-merged."""
    return dy.reshape(-1, dy.shape[-1]), ids.reshape(-1)


def embedding_bwd_dense(dy, ids, vocab):
    rows, flat = embedding_bwd_rows(dy, ids)
    g = np.zeros((vocab, rows.shape[1]), dtype=dy.dtype)
    np.add.at(g, flat, rows)
    return g


# ------------------------------------------------------------------ attention
def softmax(e, axis=-1):
    m = e.max(axis=axis, keepdims=True)
    ex = np.exp(e - m)
    return ex / ex.sum(axis=axis, keepdims=True)


def attention_proj_fwd(F, W1, b1, slope=0.2):
    """Loop-invariant half of attention.Attention.call (attention.py:32): W1(features).
    The reference recomputes it every timestep (lc_NIC.py:246); hoisted here."""
    return dense_fwd(F, W1, b1, ACT_LEAKY, slope)


def attention_step_fwd(h, F, P, W2, b2, v, bv, keep=None, rate=0.0, slope=0.2):
    """attention.Attention.call (attention.py:25-44) given P = W1(features).
    Returns ((ctx, alpha, s_dropped), cache)."""
    q, qpre = dense_fwd(h, W2, b2, ACT_LEAKY, slope)           # (B,A)
    s = np.tanh(P + q[:, None, :])                              # (B,R,A)
    sd = dropout_fwd(s, keep, rate)
    e = sd @ v[:, 0] + bv[0]                                    # (B,R)
    alpha = softmax(e, axis=1)
    ctx = (alpha[:, :, None] * F).sum(axis=1)                   # (B,D)
    return (ctx, alpha, sd), (h, qpre, s, sd, alpha, keep, rate)


def attention_step_bwd(dctx, F, W2, v, cache, slope=0.2, dalpha_ext=None):
    """Returns dh, dF(step part, excluding the W1 path), dPsum (B,R,A), dW2, db2, dv, dbv.
    dalpha_ext (B,R): a gradient that reaches the attention weights directly (lc_NIC.train_step_sam's MSE term)."""
    h, qpre, s, sd, alpha, keep, rate = cache
    dalpha = (dctx[:, None, :] * F).sum(axis=2)                 # (B,R)
    if dalpha_ext is not None:
        dalpha = dalpha + dalpha_ext
    dF = alpha[:, :, None] * dctx[:, None, :]
    de = alpha * (dalpha - (alpha * dalpha).sum(axis=1, keepdims=True))
    dv = (sd * de[:, :, None]).sum(axis=(0, 1))[:, None]
    dbv = np.array([de.sum()], dtype=de.dtype)
    dsd = de[:, :, None] * v[:, 0]
    ds = dropout_bwd(dsd, keep, rate)
    dsum = ds * (1 - s * s)                                     # grad wrt (P + q)
    dq = dsum.sum(axis=1)
    dqpre = act_bwd(qpre, dq, ACT_LEAKY, slope)
    dW2 = h.T @ dqpre
    db2 = dqpre.sum(axis=0)
    dh = dqpre @ W2.T
    return dh, dF, dsum, dW2, db2, dv, dbv


# ----------------------------------------------------------------------- LSTM
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_step_fwd(xz, h, c, U):
    """One keras LSTM (v2) cell step; gate order i,f,c~,o (lc_NIC.py:118-124,255).
    ``xz`` = x @ kernel + bias, already computed.  Returns (h', c', cache)."""
    Un = h.shape[1]
    z = xz + h @ U
    i = sigmoid(z[:, :Un])
    f = sigmoid(z[:, Un:2 * Un])
    g = np.tanh(z[:, 2 * Un:3 * Un])
    o = sigmoid(z[:, 3 * Un:])
    c2 = f * c + i * g
    tc = np.tanh(c2)
    h2 = o * tc
    return h2, c2, (i, f, g, o, c, tc, h)


def lstm_step_bwd(dh2, dc2, cache, U):
    """Returns (dz (B,4U), dh_prev, dc_prev)."""
    i, f, g, o, c, tc, h = cache
    do = dh2 * tc
    dc = dc2 + dh2 * o * (1 - tc * tc)
    di = dc * g
    df = dc * c
    dg = dc * i
    dz = np.concatenate([di * i * (1 - i), df * f * (1 - f), dg * (1 - g * g), do * o * (1 - o)], axis=1)
    return dz, dz @ U.T, dc * f


# ----------------------------------------------------------- softmax + losses
def ln_lstm_step_fwd(x, h, c, W, U, b, gk, bk, gr, br, gs, bs, eps=BN_EPS):
    """tensorflow_addons.rnn.LayerNormLSTMCell.call (tfa 0.15; the use_layer_norm branch of lc_NIC.py:126-136):
        z = LN_kernel(x @ W) + LN_recurrent(h @ U) + b;  i, f, c~, o = split(z)
        c' = LN_state(sigmoid(f) * c + sigmoid(i) * tanh(c~));  h' = sigmoid(o) * tanh(c')
    three keras LayerNormalization(epsilon = norm_epsilon = 1e-3) layers over the last axis; no dropout inside the
    cell.  The NORMALISED c' is the state that is carried.  Restated from the published tfa source (the package is
    not installable here): parity unpinned.  Returns (h', c', cache)."""
    Un = h.shape[1]
    zk, ck = layernorm_fwd(x @ W, gk, bk, eps)
    zr, cr = layernorm_fwd(h @ U, gr, br, eps)
    z = zk + zr + b
    i, f = sigmoid(z[:, :Un]), sigmoid(z[:, Un:2 * Un])
    g, o = np.tanh(z[:, 2 * Un:3 * Un]), sigmoid(z[:, 3 * Un:])
    c_raw = f * c + i * g
    cn, cs = layernorm_fwd(c_raw, gs, bs, eps)
    tc = np.tanh(cn)
    return o * tc, cn, (x, h, c, i, f, g, o, tc, ck, cr, cs)


def ln_lstm_step_bwd(dh2, dcn, cache, W, U, gk, gr, gs):
    """Returns (dx, dh_prev, dc_prev, grads) with grads = dict(W, U, b, gk, bk, gr, br, gs, bs)."""
    x, h, c, i, f, g, o, tc, ck, cr, cs = cache
    do = dh2 * tc
    dcn_t = dcn + dh2 * o * (1 - tc * tc)
    dc_raw, dgs, dbs = layernorm_bwd(dcn_t, gs, cs)
    dz = np.concatenate([dc_raw * g * i * (1 - i), dc_raw * c * f * (1 - f), dc_raw * i * (1 - g * g),
                         do * o * (1 - o)], axis=1)
    dzk, dgk, dbk = layernorm_bwd(dz, gk, ck)
    dzr, dgr, dbr = layernorm_bwd(dz, gr, cr)
    grads = dict(W=x.T @ dzk, U=h.T @ dzr, b=dz.sum(axis=0), gk=dgk, bk=dbk, gr=dgr, br=dbr, gs=dgs, bs=dbs)
    return dzk @ W.T, dzr @ U.T, dc_raw * f, grads


def gru_step_fwd(xz, h, Uk, br):
    """keras GRU cell, reset_after=True (the TF2 default; ThinkAndTell/att_model.py:84-93): xz = x@W + b_i (B,3U)
    in keras gate order [z, r, h]; Uk (U,3U); br (3U,) the recurrent bias."""
    U = h.shape[1]
    rec = h @ Uk + br
    z = sigmoid(xz[:, :U] + rec[:, :U])
    r = sigmoid(xz[:, U:2 * U] + rec[:, U:2 * U])
    hh = np.tanh(xz[:, 2 * U:] + r * rec[:, 2 * U:])
    h2 = z * h + (1 - z) * hh
    return h2, (z, r, hh, rec[:, 2 * U:], h)


def gru_step_bwd(dh2, cache, Uk):
    """returns (dxz (B,3U), drec (B,3U), dh_prev)."""
    z, r, hh, rech, h = cache
    dz = dh2 * (h - hh)
    dah = dh2 * (1 - z) * (1 - hh * hh)
    daz = dz * z * (1 - z)
    dar = dah * rech * r * (1 - r)
    dxz = np.concatenate([daz, dar, dah], axis=1)
    drec = np.concatenate([daz, dar, dah * r], axis=1)
    return dxz, drec, dh2 * z + drec @ Uk.T


def cce_from_probs(p, y_ids, eps=CCE_EPS):
    """keras CategoricalCrossentropy(from_logits=False, reduction='none') on a one-hot target
    (main.py:107-110): p <- p/sum(p); clip(p, eps, 1-eps); -log p[y].
    Returns per-sample losses (B,)."""
    q = p / p.sum(axis=-1, keepdims=True)
    q = np.clip(q, eps, 1 - eps)
    return -np.log(np.take_along_axis(q, y_ids[..., None], axis=-1)[..., 0])


def cce_softmax_bwd(p, y_ids, dl, eps=CCE_EPS):
    """Gradient wrt the *logits* of softmax->normalise->clip->-log, per-sample upstream dl (B,).
    Zero where the clip is active (clip_by_value passes no gradient outside its range)."""
    S = p.sum(axis=-1, keepdims=True)
    q = p / S
    qy = np.take_along_axis(q, y_ids[..., None], axis=-1)
    active = ((qy >= eps) & (qy <= 1 - eps)).astype(p.dtype)
    onehot = np.zeros_like(p)
    np.put_along_axis(onehot, y_ids[..., None], 1.0, axis=-1)
    py = np.take_along_axis(p, y_ids[..., None], axis=-1)
    # dL/dp_j = -(1/q_y) * (delta_jy / S - p_y / S^2)
    dp = -(active / np.maximum(qy, 1e-300)) * (onehot / S - py / (S * S)) * dl[..., None]
    return p * (dp - (dp * p).sum(axis=-1, keepdims=True))


def sparse_cce_from_logits(logits, y_ids):
    """tf SparseCategoricalCrossentropy(from_logits=True, reduction='none')
    (ThinkAndTell/train.py:262-263).  Returns (loss (B,), probs)."""
    p = softmax(logits, axis=-1)
    m = logits.max(axis=-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(logits - m).sum(axis=-1))
    return lse - np.take_along_axis(logits, y_ids[..., None], axis=-1)[..., 0], p


def accuracy(p, y_ids):
    """lc_NIC.accuracy_calculation (lc_NIC.py:468-486): argmax ties -> first index."""
    return (p.argmax(axis=-1) == y_ids).astype(p.dtype).mean()


# ------------------------------------------------------------------ optimizer
def clip_by_norm(g, clipnorm):
    """tf.clip_by_norm as applied per variable by keras clipnorm:
    g * c / max(||g||, c)."""
    n = np.sqrt((g * g).sum())
    return g * clipnorm / np.maximum(n, clipnorm)


def unitwise_norm(x):
    """agc.unitwise_norm (AttemptFour/Model/agc.py:6-18): scalars / vectors -> one norm; rank 2, 3 -> over axis 0
    (one unit per output column), keepdims."""
    x = np.asarray(x)
    if x.ndim <= 1:
        return np.sqrt((x * x).sum())
    if x.ndim in (2, 3):
        return np.sqrt((x * x).sum(axis=0, keepdims=True))
    raise ValueError("unitwise_norm: rank must be <= 3 here")


def adaptive_clip_grad(param, grad, clip_factor=0.01, eps=1e-3):
    """agc.adaptive_clip_grad for ONE (parameter, gradient) pair (agc.py:20-38).  ``grad`` may be the values of an
    IndexedSlices (rows gathered from an Embedding): its unit norms are then taken over those rows (agc.py:25-30)."""
    p_norm = unitwise_norm(param)
    max_norm = np.maximum(p_norm, eps) * clip_factor
    grad_norm = unitwise_norm(grad)
    clipped = grad * (max_norm / np.maximum(grad_norm, 1e-6))
    return np.where(grad_norm < max_norm, grad, clipped)


def adam_update(theta, m, v, g, t, lr=1e-4, b1=0.9, b2=0.98, eps=1e-8):
    """keras OptimizerV2 Adam (main.py:97): epsilon outside the bias correction;
    t starts at 1.  Returns (theta, m, v)."""
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = m + (g - m) * (1 - b1)
    v = v + (g * g - v) * (1 - b2)
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v


def sgd_momentum_update(theta, mom, g, lr, momentum=0.9):
    """keras SGD(momentum=0.9, nesterov=False) (main.py:100-102)."""
    mom = momentum * mom - lr * g
    return theta + mom, mom


def sample_rows(x, temperature, from_logits, seed, site, step):
    """Categorical sampling per row, inverse CDF with one Philox uniform per row -- the definition of
    tnt_sample_rows_f32 (stands for tf.random.categorical(logits / temperature, 1),
    ThinkAndTell/evaluate.py:223,278; lc_NIC.sample_choice lc_NIC.py:571-575 passes log(probs)).
    Returns (ids, margin): margin[r] = distance of u*sum from the nearest CDF edge relative to sum
    (a float32 implementation may legitimately differ where the margin is ~1e-6)."""
    from .philox import uniform24
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(divide='ignore'):
        l = x if from_logits else np.log(x)
    w = np.exp((l - l.max(axis=-1, keepdims=True)) / temperature)
    cdf = np.cumsum(w, axis=-1)
    u = uniform24(x.shape[0], int(seed), int(site), int(step)).astype(np.float64)
    target = u * cdf[:, -1]
    ids = np.array([int(np.searchsorted(cdf[r], target[r], side='right')) for r in range(x.shape[0])])
    ids = np.minimum(ids, x.shape[1] - 1)
    margin = np.abs(cdf - target[:, None]).min(axis=-1) / cdf[:, -1]
    return ids, margin
