"""ThinkAndTell / ShowAndTell caption generators -- CPU oracle.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED.

``CaptionGeneratorTT`` restates ThinkAndTell/model.py: Encoder (10-33: Dense(tanh), L2 on kernel
and bias, dropout after the Dense only when training), Decoder (42-114: Embedding without mask,
the feature prepended as token 0, one LSTM call over T+1 steps from zero state, L2 on kernel
and recurrent kernel, dropout on the LSTM output when training, TimeDistributed Dense(V, relu)),
CaptionGenerator.train_step / test_step / loss_function (241-334: masked sparse CE from logits,
target[:, i] paired with predictions[:, i], summed over i, divided by T, plus L2) and
train_step_SAM (166-233).
``show_and_tell=True`` switches to ShowAndTell/model.py (config 1): relu encoder without
dropout/L2 (10-20), linear fc1 -> fc2 head (37-38,59-63), loss over i = 1..T-1 and the gradient
of the UN-normalised sum (154-161).  Its LSTM is called with the Embedding mask (51-57); the
mask is applied as a per-sample sequence length over the T+1 inputs, which is what the cuDNN
path of keras LSTM does with a mask.
"""
import numpy as np

from . import ops as O
from .models import DropCtx, _l2, S_FEAT, S_OUT


class CaptionGeneratorTT:
    def __init__(self, input_size, embedding_dim, units, vocab_size, max_length, l2_reg=0.0, dropout=0.0,
                 show_and_tell=False):
        self.N, self.E, self.U, self.V, self.T = input_size, embedding_dim, units, vocab_size, max_length
        self.l2, self.rate, self.sat = l2_reg, dropout, show_and_tell
        self.p = {}

    def init_params(self, rng, dtype=np.float64):
        N, E, U, V = self.N, self.E, self.U, self.V
        p = self.p
        p['fc_embedding/kernel'] = (rng.standard_normal((N, E)) / np.sqrt(N)).astype(dtype)
        p['fc_embedding/bias'] = (0.05 * rng.standard_normal(E)).astype(dtype)
        p['embedding/embeddings'] = rng.uniform(-0.05, 0.05, (V, E)).astype(dtype)
        p['lstm/kernel'] = (rng.standard_normal((E, 4 * U)) / np.sqrt(E)).astype(dtype)
        p['lstm/recurrent_kernel'] = (rng.standard_normal((U, 4 * U)) / np.sqrt(U)).astype(dtype)
        b = 0.01 * rng.standard_normal(4 * U); b[U:2 * U] += 1.0
        p['lstm/bias'] = b.astype(dtype)
        if self.sat:
            p['fc1/kernel'] = (rng.standard_normal((U, U)) / np.sqrt(U)).astype(dtype)
            p['fc1/bias'] = (0.01 * rng.standard_normal(U)).astype(dtype)
        p['fc_vocab/kernel'] = (rng.standard_normal((U, V)) / np.sqrt(U)).astype(dtype)
        p['fc_vocab/bias'] = (0.05 * rng.standard_normal(V)).astype(dtype)
        return self

    def trainable(self):
        return list(self.p)

    def l2_loss(self):
        if self.sat or self.l2 == 0:
            return 0.0
        p = self.p
        return sum(_l2(self.l2, p[k]) for k in ('fc_embedding/kernel', 'fc_embedding/bias', 'lstm/kernel',
                                                'lstm/recurrent_kernel'))

    def forward(self, x, target, training=False, drop=None):
        p = self.p
        dt = p['lstm/kernel'].dtype
        x = x.astype(dt)
        B, T = target.shape
        drop = drop or DropCtx(training=training)
        act = O.ACT_RELU if self.sat else O.ACT_TANH
        feat, epre = O.dense_fwd(x, p['fc_embedding/kernel'], p['fc_embedding/bias'], act)
        k_f = drop.mask(feat.shape, 0.0 if self.sat else self.rate, S_FEAT)
        feat_d = O.dropout_fwd(feat, k_f, self.rate)
        emb = O.embedding_fwd(p['embedding/embeddings'], target)
        xin = np.concatenate([feat_d[:, None, :], emb], axis=1)              # (B, T+1, E)
        lens = (target != 0).sum(axis=1) if self.sat else np.full(B, T + 1)
        h = np.zeros((B, self.U), dt); c = np.zeros((B, self.U), dt)
        outs, caches, masks = [], [], []
        Wl, Ul, bl = p['lstm/kernel'], p['lstm/recurrent_kernel'], p['lstm/bias']
        for t in range(T + 1):
            h2, c2, ch = O.lstm_step_fwd(xin[:, t] @ Wl + bl, h, c, Ul)
            m = (t < lens)[:, None]
            h, c = np.where(m, h2, h), np.where(m, c2, c)
            outs.append(np.where(m, h2, 0.0))                                 # cuDNN: zeros past the length
            caches.append(ch); masks.append(m)
        Hs = np.stack(outs, axis=1)                                           # (B, T+1, U)
        k_o = drop.mask(Hs.shape, 0.0 if self.sat else self.rate, S_OUT)
        Hd = O.dropout_fwd(Hs, k_o, self.rate)
        if self.sat:
            mid, _ = O.dense_fwd(Hd, p['fc1/kernel'], p['fc1/bias'])
            logits, lpre = O.dense_fwd(mid, p['fc_vocab/kernel'], p['fc_vocab/bias'])
        else:
            mid = Hd
            logits, lpre = O.dense_fwd(Hd, p['fc_vocab/kernel'], p['fc_vocab/bias'], O.ACT_RELU)
        cache = dict(x=x, epre=epre, k_f=k_f, xin=xin, caches=caches, masks=masks, Hd=Hd, mid=mid, lpre=lpre,
                     k_o=k_o, target=target)
        return logits, cache

    def _positions(self, T):
        return range(1, T) if self.sat else range(0, T)

    def loss(self, logits, target):
        """sum_i mean_b( SCCE(target[:,i], logits[:,i]) * [target[:,i] != 0] )  (and the same / T)."""
        B, T = target.shape
        tot = 0.0
        for i in self._positions(T):
            l, _ = O.sparse_cce_from_logits(logits[:, i], target[:, i])
            tot = tot + (l * (target[:, i] != 0)).mean()
        return tot, tot / T

    def backward(self, logits, cache, grad_of_sum=False):
        p = self.p
        target = cache['target']
        B, T = target.shape
        scale = (1.0 if (self.sat or grad_of_sum) else 1.0 / T) / B
        dlog = np.zeros_like(logits)
        for i in self._positions(T):
            pr = O.softmax(logits[:, i])
            oh = np.zeros_like(pr); np.put_along_axis(oh, target[:, i][:, None], 1.0, 1)
            dlog[:, i] = (pr - oh) * (target[:, i] != 0)[:, None] * scale
        g = {}
        if self.sat:
            dmid, g['fc_vocab/kernel'], g['fc_vocab/bias'] = O.dense_bwd(cache['mid'], p['fc_vocab/kernel'], cache['lpre'], dlog)
            dHd, g['fc1/kernel'], g['fc1/bias'] = O.dense_bwd(cache['Hd'], p['fc1/kernel'], None, dmid)
        else:
            dHd, g['fc_vocab/kernel'], g['fc_vocab/bias'] = O.dense_bwd(cache['Hd'], p['fc_vocab/kernel'], cache['lpre'],
                                                                          dlog, O.ACT_RELU)
        dHs = O.dropout_bwd(dHd, cache['k_o'], self.rate)
        Wl, Ul = p['lstm/kernel'], p['lstm/recurrent_kernel']
        dWl, dUl, dbl = np.zeros_like(Wl), np.zeros_like(Ul), np.zeros_like(p['lstm/bias'])
        dxin = np.zeros_like(cache['xin'])
        dh = np.zeros((B, self.U), logits.dtype); dc = np.zeros((B, self.U), logits.dtype)
        for t in reversed(range(T + 1)):
            m = cache['masks'][t]
            dh2 = np.where(m, dh + dHs[:, t], 0)
            dc2 = np.where(m, dc, 0)
            dz, dhp, dcp = O.lstm_step_bwd(dh2, dc2, cache['caches'][t], Ul)
            dWl += cache['xin'][:, t].T @ dz
            dUl += cache['caches'][t][6].T @ dz
            dbl += dz.sum(0)
            dxin[:, t] = dz @ Wl.T
            dh = np.where(m, 0, dh) + dhp
            dc = np.where(m, 0, dc) + dcp
        lam = 0.0 if self.sat else self.l2
        g['lstm/kernel'] = dWl + 2 * lam * Wl
        g['lstm/recurrent_kernel'] = dUl + 2 * lam * Ul
        g['lstm/bias'] = dbl
        g['embedding/embeddings'] = O.embedding_bwd_dense(dxin[:, 1:], target, self.V)
        rows = dxin[:, 1:].reshape(-1, self.E)
        dfeat = O.dropout_bwd(dxin[:, 0], cache['k_f'], self.rate)
        act = O.ACT_RELU if self.sat else O.ACT_TANH
        _, dK, db = O.dense_bwd(cache['x'], p['fc_embedding/kernel'], cache['epre'], dfeat, act, need_dx=False)
        g['fc_embedding/kernel'] = dK + 2 * lam * p['fc_embedding/kernel']
        g['fc_embedding/bias'] = db + 2 * lam * p['fc_embedding/bias']
        return g, {'embedding/embeddings': np.sqrt((rows * rows).sum())}

    def train_step(self, x, target, opt, drop=None):
        """CaptionGenerator.train_step (ThinkAndTell/model.py:241-290; ShowAndTell/model.py:125-164)."""
        drop = drop or DropCtx(training=True)
        logits, cache = self.forward(x, target, True, drop)
        tot, scce = self.loss(logits, target)
        l2 = self.l2_loss()
        grads, sparse = self.backward(logits, cache)
        opt.apply(self.p, grads, sparse)
        if self.sat:
            return {'loss': tot, 'norm loss': scce}, grads
        return {'scce': scce, 'L2': l2, 'loss': scce + l2}, grads

    def test_step(self, x, target):
        logits, _ = self.forward(x, target, False)
        tot, scce = self.loss(logits, target)
        if self.sat:
            return {'loss': tot, 'norm loss': scce}
        l2 = self.l2_loss()
        return {'scce': scce, 'L2': l2, 'loss': scce + l2}

    def train_step_sam(self, x, target, opt, drop=None, rho=0.05):
        """CaptionGenerator.train_step_SAM (ThinkAndTell/model.py:166-233)."""
        drop = drop or DropCtx(training=True)
        logits, cache = self.forward(x, target, True, drop)
        g1, _ = self.backward(logits, cache)
        norm = np.sqrt(sum((g * g).sum() for g in g1.values()))
        scale = rho / (norm + 1e-12)
        e_ws = {k: g * scale for k, g in g1.items()}
        for k, e in e_ws.items():
            self.p[k] = self.p[k] + e
        logits2, cache2 = self.forward(x, target, True, drop)
        _, scce = self.loss(logits2, target)
        l2 = self.l2_loss()
        g2, sparse = self.backward(logits2, cache2)
        for k, e in e_ws.items():
            self.p[k] = self.p[k] - e
        opt.apply(self.p, g2, sparse)
        return {'scce': scce, 'L2': l2, 'loss': scce + l2}, g2
