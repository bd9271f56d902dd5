"""ThinkAndTell/att_model.py caption generator (GRU decoder) -- CPU oracle.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED.

Restates (paths relative to /root/reference/ThinkAndTell):
  * ``Encoder``  att_model.py:31-52   Dense(E, relu, L2 on the kernel), Dropout only when training
  * ``Decoder``  att_model.py:61-129  Embedding (no mask, L2 on the table) -> the feature prepended as token 0 ->
                 GRU over T+1 steps from zero state (keras GRU v2: reset_after=True, bias (2, 3U), L2 on kernel
                 and recurrent kernel) -> training: Dropout -> fc1 Dense(U, relu) -> Dropout (the SAME Dropout
                 layer called twice = two independent masks); inference: fc1 -> fc2 Dense(V, relu) = "logits".
                 (``BahdanauAttention``, att_model.py:11-29, is an empty stub and is not called.)
  * ``CaptionGenerator.train_step / test_step / loss_function``  att_model.py:228-321: masked sparse CE from logits,
                 target[:, i] paired with predictions[:, i-1] for i = 1..T-1, sum / T, + encoder and decoder L2.
The Embedding gets a dense regulariser gradient next to its IndexedSlices, so tape.gradient returns a dense tensor
and clipnorm uses the ordinary norm (no sparse-norm quirk here).
"""
import numpy as np

from . import ops as O
from .models import DropCtx, _l2, S_FEAT, S_OUT

TRAINABLE = ['fc_embedding/kernel', 'fc_embedding/bias', 'embedding/embeddings', 'gru/kernel', 'gru/recurrent_kernel',
             'gru/bias', 'fc1/kernel', 'fc1/bias', 'fc_vocab/kernel', 'fc_vocab/bias']


class CaptionGeneratorAtt:
    def __init__(self, input_size, embedding_dim, units, vocab_size, max_length, l2_reg=0.0, dropout=0.0):
        self.N, self.E, self.U, self.V, self.T = input_size, embedding_dim, units, vocab_size, max_length
        self.l2, self.rate = l2_reg, dropout
        self.p = {}

    def init_params(self, rng, dtype=np.float64):
        N, E, U, V = self.N, self.E, self.U, self.V
        p = self.p
        p['fc_embedding/kernel'] = (rng.standard_normal((N, E)) / np.sqrt(N)).astype(dtype)
        p['fc_embedding/bias'] = (0.05 * rng.standard_normal(E)).astype(dtype)
        p['embedding/embeddings'] = rng.uniform(-0.05, 0.05, (V, E)).astype(dtype)
        p['gru/kernel'] = (rng.standard_normal((E, 3 * U)) / np.sqrt(E)).astype(dtype)
        p['gru/recurrent_kernel'] = (rng.standard_normal((U, 3 * U)) / np.sqrt(U)).astype(dtype)
        p['gru/bias'] = (0.05 * rng.standard_normal((2, 3 * U))).astype(dtype)
        p['fc1/kernel'] = (rng.standard_normal((U, U)) / np.sqrt(U)).astype(dtype)
        p['fc1/bias'] = (0.05 * rng.standard_normal(U)).astype(dtype)
        p['fc_vocab/kernel'] = (rng.standard_normal((U, V)) / np.sqrt(U)).astype(dtype)
        p['fc_vocab/bias'] = (0.05 * rng.standard_normal(V)).astype(dtype)
        return self

    REG = ('fc_embedding/kernel', 'embedding/embeddings', 'gru/kernel', 'gru/recurrent_kernel')

    def l2_loss(self):
        return sum(_l2(self.l2, self.p[k]) for k in self.REG)

    def forward(self, x, target, training=False, drop=None):
        p = self.p
        dt = p['gru/kernel'].dtype
        x = x.astype(dt)
        B, T = target.shape
        drop = drop or DropCtx(training=training)
        feat, epre = O.dense_fwd(x, p['fc_embedding/kernel'], p['fc_embedding/bias'], O.ACT_RELU)     # :49-52
        k_f = drop.mask(feat.shape, self.rate, S_FEAT)
        feat_d = O.dropout_fwd(feat, k_f, self.rate)
        emb = O.embedding_fwd(p['embedding/embeddings'], target)                                     # :108
        xin = np.concatenate([feat_d[:, None, :], emb], axis=1)                                        # :112
        W, Uk, b = p['gru/kernel'], p['gru/recurrent_kernel'], p['gru/bias']
        h = np.zeros((B, self.U), dt)
        outs, caches = [], []
        for t in range(T + 1):                                                                         # :118
            h, ch = O.gru_step_fwd(xin[:, t] @ W + b[0], h, Uk, b[1])
            outs.append(h); caches.append(ch)
        Hs = np.stack(outs, axis=1)
        k1 = drop.mask(Hs.shape, self.rate, S_OUT)
        Hd = O.dropout_fwd(Hs, k1, self.rate)                                                          # :121-122
        mid, mpre = O.dense_fwd(Hd, p['fc1/kernel'], p['fc1/bias'], O.ACT_RELU)
        k2 = drop.mask(mid.shape, self.rate, S_OUT + 1)
        mid_d = O.dropout_fwd(mid, k2, self.rate)
        logits, lpre = O.dense_fwd(mid_d, p['fc_vocab/kernel'], p['fc_vocab/bias'], O.ACT_RELU)        # :127
        cache = dict(x=x, epre=epre, k_f=k_f, xin=xin, caches=caches, Hd=Hd, k1=k1, mpre=mpre, mid_d=mid_d, k2=k2,
                     lpre=lpre, target=target)
        return logits, cache

    def loss(self, logits, target):
        """sum_{i=1}^{T-1} mean_b( SCCE(target[:, i], logits[:, i-1]) * [target[:, i] != 0] ) / T  (:257-262,306-321)."""
        B, T = target.shape
        tot = 0.0
        for i in range(1, T):
            l, _ = O.sparse_cce_from_logits(logits[:, i - 1], target[:, i])
            tot = tot + (l * (target[:, i] != 0)).mean()
        return tot / T

    def backward(self, logits, cache):
        p = self.p
        target = cache['target']
        B, T = target.shape
        U = self.U
        dlog = np.zeros_like(logits)
        for i in range(1, T):
            pr = O.softmax(logits[:, i - 1])
            oh = np.zeros_like(pr); np.put_along_axis(oh, target[:, i][:, None], 1.0, 1)
            dlog[:, i - 1] = (pr - oh) * (target[:, i] != 0)[:, None] / (B * T)
        g = {}
        dmid_d, g['fc_vocab/kernel'], g['fc_vocab/bias'] = O.dense_bwd(cache['mid_d'], p['fc_vocab/kernel'], cache['lpre'],
                                                                         dlog, O.ACT_RELU)
        dmid = O.dropout_bwd(dmid_d, cache['k2'], self.rate)
        dHd, g['fc1/kernel'], g['fc1/bias'] = O.dense_bwd(cache['Hd'], p['fc1/kernel'], cache['mpre'], dmid, O.ACT_RELU)
        dHs = O.dropout_bwd(dHd, cache['k1'], self.rate)
        W, Uk = p['gru/kernel'], p['gru/recurrent_kernel']
        dW, dUk, db = np.zeros_like(W), np.zeros_like(Uk), np.zeros_like(p['gru/bias'])
        dxin = np.zeros_like(cache['xin'])
        dh = np.zeros((B, U), logits.dtype)
        for t in reversed(range(T + 1)):
            dxz, drec, dh = O.gru_step_bwd(dh + dHs[:, t], cache['caches'][t], Uk)
            dW += cache['xin'][:, t].T @ dxz
            dUk += cache['caches'][t][4].T @ drec
            db[0] += dxz.sum(0); db[1] += drec.sum(0)
            dxin[:, t] = dxz @ W.T
        lam = self.l2
        g['gru/kernel'] = dW + 2 * lam * W
        g['gru/recurrent_kernel'] = dUk + 2 * lam * Uk
        g['gru/bias'] = db
        g['embedding/embeddings'] = (O.embedding_bwd_dense(dxin[:, 1:], target, self.V)
                                     + 2 * lam * p['embedding/embeddings'])
        dfeat = O.dropout_bwd(dxin[:, 0], cache['k_f'], self.rate)
        _, dK, dbe = O.dense_bwd(cache['x'], p['fc_embedding/kernel'], cache['epre'], dfeat, O.ACT_RELU, need_dx=False)
        g['fc_embedding/kernel'] = dK + 2 * lam * p['fc_embedding/kernel']
        g['fc_embedding/bias'] = dbe
        return g

    def train_step(self, x, target, opt, drop=None):
        drop = drop or DropCtx(training=True)
        logits, cache = self.forward(x, target, True, drop)
        scce = self.loss(logits, target)
        l2 = self.l2_loss()
        grads = self.backward(logits, cache)
        opt.apply(self.p, grads, None)
        return {'scce': scce, 'L2': l2, 'loss': scce + l2}, grads

    def test_step(self, x, target):
        logits, _ = self.forward(x, target, False)
        scce = self.loss(logits, target)
        l2 = self.l2_loss()
        return {'scce': scce, 'L2': l2, 'loss': scce + l2}
