"""CPU oracle of the fully-connected mode of AttemptFour/Model/lc_NIC.py.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED.

Restates (paths relative to /root/reference/AttemptFour/Model):
  * ``lc_NIC.call_fc``            lc_NIC.py:298-323
  * ``FullyConnected.call``       fullyConnected.py:20-27  (Dense -> BatchNorm -> Dropout; the
                                  construction the reference keeps commented at lc_NIC.py:60-67)
  * ``lc_NIC.greedy_predict_fc``  lc_NIC.py:511-542
with the loss / metrics / update of lc_NIC.train_step (lc_NIC.py:328-408).

Behaviour kept exactly as written in the reference:
  * call_fc runs the LSTM on the encoded feature but throws the resulting state away
    (lc_NIC.py:317) and starts the text LSTM from (a0, c0) (lc_NIC.py:318): the prediction does not
    depend on the betas.  The encoder therefore receives no data gradient -- its kernel only the
    L2 term, bias / gamma / beta none at all (tape.gradient -> None, skipped by apply_gradients) --
    but its BatchNorm still updates the moving statistics in training.
  * greedy_predict_fc DOES use the feature state (lc_NIC.py:520) and freezes the whole batch once
    every sample has emitted id 0 (lc_NIC.py:526-527).
  * call_fc returns (output, None); the attention-coverage metric of train_step cannot be computed
    from None, so the step dictionary here carries loss / L2 / accuracy only.
"""
import numpy as np
from . import ops as O
from .models import DropCtx, _l2, S_IN, S_FEAT, S_TEXT, S_OUT, S_LSTM_IN, S_LSTM_OUT


class FcNIC:
    H = 256                                                       # lc_NIC.py:141

    TRAINABLE = ['dense_in/kernel', 'dense_in/bias', 'dense_in_bn/gamma', 'dense_in_bn/beta',
                 'emb_text/embeddings', 'lstm/kernel', 'lstm/recurrent_kernel', 'lstm/bias',
                 'time_distributed_nonlinear/kernel', 'time_distributed_nonlinear/bias',
                 'time_distributed_softmax/kernel', 'time_distributed_softmax/bias']

    def __init__(self, input_size, units, embedding_features, embedding_text, vocab_size, max_length,
                 dropout_input, dropout_features, dropout_text, dropout_lstm, dropout_out, input_reg, lstm_reg,
                 output_reg):
        self.N, self.U, self.Ef, self.Et, self.V, self.T = (input_size, units, embedding_features, embedding_text,
                                                            vocab_size, max_length)
        self.r_in, self.r_feat, self.r_text = dropout_input, dropout_features, dropout_text
        self.r_lstm, self.r_out = dropout_lstm, dropout_out
        self.l2_in, self.l2_lstm, self.l2_out = input_reg, lstm_reg, output_reg
        self.p = {}

    def init_params(self, rng, dtype=np.float64):
        N, U, Ef, Et, V, H = self.N, self.U, self.Ef, self.Et, self.V, self.H
        p = self.p
        p['dense_in/kernel'] = (rng.standard_normal((N, Ef)) * np.sqrt(2.0 / N)).astype(dtype)
        p['dense_in/bias'] = (0.01 * rng.standard_normal(Ef)).astype(dtype)
        p['dense_in_bn/gamma'] = (1 + 0.1 * rng.standard_normal(Ef)).astype(dtype)
        p['dense_in_bn/beta'] = (0.1 * rng.standard_normal(Ef)).astype(dtype)
        p['dense_in_bn/moving_mean'] = (0.1 * rng.standard_normal(Ef)).astype(dtype)
        p['dense_in_bn/moving_variance'] = (1 + 0.1 * rng.random(Ef)).astype(dtype)
        p['emb_text/embeddings'] = rng.uniform(-0.08, 0.08, (V, Et)).astype(dtype)
        lim = np.sqrt(6.0 / (Et + 4 * U))
        p['lstm/kernel'] = rng.uniform(-lim, lim, (Et, 4 * U)).astype(dtype)
        p['lstm/recurrent_kernel'] = (rng.standard_normal((U, 4 * U)) / np.sqrt(U)).astype(dtype)
        b = 0.01 * rng.standard_normal(4 * U)
        b[U:2 * U] += 1.0
        p['lstm/bias'] = b.astype(dtype)
        p['time_distributed_nonlinear/kernel'] = (rng.standard_normal((U, H)) * np.sqrt(2.0 / (U + H))).astype(dtype)
        p['time_distributed_nonlinear/bias'] = (0.01 * rng.standard_normal(H)).astype(dtype)
        p['time_distributed_softmax/kernel'] = (rng.standard_normal((H, V)) * np.sqrt(2.0 / (H + V))).astype(dtype)
        p['time_distributed_softmax/bias'] = (0.01 * rng.standard_normal(V)).astype(dtype)
        return self

    # ---- FullyConnected.call (fullyConnected.py:20-27)
    def encode(self, x, training, drop):
        p = self.p
        k_in = drop.mask(x.shape, self.r_in, S_IN)
        xd = O.dropout_fwd(x, k_in, self.r_in)                                        # lc_NIC.py:301
        y, _ = O.dense_fwd(xd, p['dense_in/kernel'], p['dense_in/bias'], O.ACT_LEAKY)  # dense2, LeakyReLU(0.2)
        f, _, mm, mv = O.batchnorm_fwd(y, p['dense_in_bn/gamma'], p['dense_in_bn/beta'], p['dense_in_bn/moving_mean'],
                                       p['dense_in_bn/moving_variance'], training)
        k_f = drop.mask(f.shape, self.r_feat, S_FEAT)
        return O.dropout_fwd(f, k_f, self.r_feat), mm, mv

    # ---- call_fc (lc_NIC.py:298-323)
    def forward(self, data, training=False, drop=None):
        x, ids, a0, c0 = data
        p = self.p
        dt = p['lstm/kernel'].dtype
        B, T, U = x.shape[0], ids.shape[1], self.U
        drop = drop or DropCtx(training=training)
        _, mm, mv = self.encode(x.astype(dt), training, drop)          # result unused (:317), BN stats are not
        emb = O.embedding_fwd(p['emb_text/embeddings'], ids)           # :307
        mask = ids != 0
        k_t = drop.mask(emb.shape, self.r_text, S_TEXT)
        text = O.dropout_fwd(emb, k_t, self.r_text)                    # :308
        k_l = drop.mask(emb.shape, self.r_lstm, S_LSTM_IN + 1)         # LSTM(dropout=...) on its input
        text_d = O.dropout_fwd(text, k_l, self.r_lstm)
        Wl, Ul, bl = p['lstm/kernel'], p['lstm/recurrent_kernel'], p['lstm/bias']
        a, c = a0.astype(dt), c0.astype(dt)
        outs, caches = [], []
        out_prev = np.zeros((B, U), dt)
        for t in range(T):                                             # :318, masked by the Embedding mask
            h2, c2, ch = O.lstm_step_fwd(text_d[:, t] @ Wl + bl, a, c, Ul)
            m = mask[:, t][:, None]
            a = np.where(m, h2, a)
            c = np.where(m, c2, c)
            out_prev = np.where(m, h2, out_prev)
            outs.append(out_prev)
            caches.append(ch)
        A = np.stack(outs, axis=1)
        k_a = drop.mask(A.shape, self.r_lstm, S_LSTM_OUT)
        A_d = O.dropout_fwd(A, k_a, self.r_lstm)                       # :321
        inter, ipre = O.dense_fwd(A_d, p['time_distributed_nonlinear/kernel'], p['time_distributed_nonlinear/bias'],
                                  O.ACT_LEAKY)
        k_o = drop.mask(inter.shape, self.r_out, S_OUT)
        inter_d = O.dropout_fwd(inter, k_o, self.r_out)                # :322
        logits = inter_d @ p['time_distributed_softmax/kernel'] + p['time_distributed_softmax/bias']
        probs = O.softmax(logits, axis=-1)                             # :323
        cache = dict(k_t=k_t, k_l=k_l, text_d=text_d, mask=mask, caches=caches, A_d=A_d, k_a=k_a, ipre=ipre, k_o=k_o,
                     inter_d=inter_d, ids=ids, new_mm=mm, new_mv=mv, logits=logits)
        return probs, cache

    def l2_loss(self):
        p = self.p
        return (_l2(self.l2_in, p['dense_in/kernel']) + _l2(self.l2_lstm, p['lstm/kernel'])
                + _l2(self.l2_out, p['time_distributed_nonlinear/kernel'])
                + _l2(self.l2_out, p['time_distributed_softmax/kernel']))

    def metrics(self, probs, y_ids):
        T = y_ids.shape[1]
        ce = sum(O.cce_from_probs(probs[:, t], y_ids[:, t]).mean() for t in range(T)) / T   # lc_NIC.py:370-376
        acc = sum(O.accuracy(probs[:, t], y_ids[:, t]) for t in range(T)) / T
        return ce, acc

    def backward(self, probs, cache, y_ids):
        p = self.p
        B, T = y_ids.shape
        U, H = self.U, self.H
        g = {}
        dl = np.full((B, T), 1.0 / (B * T), probs.dtype)
        dlogits = O.cce_softmax_bwd(probs, y_ids, dl)
        Wo, Wi = p['time_distributed_softmax/kernel'], p['time_distributed_nonlinear/kernel']
        g['time_distributed_softmax/kernel'] = (cache['inter_d'].reshape(-1, H).T @ dlogits.reshape(B * T, -1)
                                                + 2 * self.l2_out * Wo)
        g['time_distributed_softmax/bias'] = dlogits.sum(axis=(0, 1))
        dinter = O.dropout_bwd(dlogits @ Wo.T, cache['k_o'], self.r_out)
        dipre = O.act_bwd(cache['ipre'], dinter, O.ACT_LEAKY)
        g['time_distributed_nonlinear/kernel'] = cache['A_d'].reshape(-1, U).T @ dipre.reshape(B * T, -1) + 2 * self.l2_out * Wi
        g['time_distributed_nonlinear/bias'] = dipre.sum(axis=(0, 1))
        dA = O.dropout_bwd(dipre @ Wi.T, cache['k_a'], self.r_lstm)
        Wl, Ul = p['lstm/kernel'], p['lstm/recurrent_kernel']
        dWl, dUl, dbl = np.zeros_like(Wl), np.zeros_like(Ul), np.zeros_like(p['lstm/bias'])
        dtext_d = np.zeros_like(cache['text_d'])
        mask = cache['mask']
        da = np.zeros((B, U), probs.dtype)
        dc = np.zeros((B, U), probs.dtype)
        dout = np.zeros((B, U), probs.dtype)
        for t in reversed(range(T)):
            m = mask[:, t][:, None]
            dout = dout + dA[:, t]
            dh2 = np.where(m, da + dout, 0)
            dc2 = np.where(m, dc, 0)
            dz, dh_prev, dc_prev = O.lstm_step_bwd(dh2, dc2, cache['caches'][t], Ul)
            h_prev = cache['caches'][t][6]
            dWl += cache['text_d'][:, t].T @ dz
            dUl += h_prev.T @ dz
            dbl += dz.sum(axis=0)
            dtext_d[:, t] = dz @ Wl.T
            da = np.where(m, 0, da) + dh_prev
            dc = np.where(m, 0, dc) + dc_prev
            dout = np.where(m, 0, dout)
        g['lstm/kernel'] = dWl + 2 * self.l2_lstm * Wl
        g['lstm/recurrent_kernel'] = dUl
        g['lstm/bias'] = dbl
        demb = O.dropout_bwd(O.dropout_bwd(dtext_d, cache['k_l'], self.r_lstm), cache['k_t'], self.r_text)
        rows, _ = O.embedding_bwd_rows(demb, cache['ids'])
        g['emb_text/embeddings'] = O.embedding_bwd_dense(demb, cache['ids'], self.V)
        # encoder: unconnected to the loss except through its kernel regulariser
        g['dense_in/kernel'] = 2 * self.l2_in * p['dense_in/kernel']
        g['dense_in/bias'] = g['dense_in_bn/gamma'] = g['dense_in_bn/beta'] = None
        return g, {'emb_text/embeddings': np.sqrt((rows * rows).sum())}

    def train_step(self, data, y_ids, opt, drop=None):
        drop = drop or DropCtx(training=True)
        probs, cache = self.forward(data, training=True, drop=drop)
        ce, acc = self.metrics(probs, y_ids)
        l2 = self.l2_loss()
        grads, sparse = self.backward(probs, cache, y_ids)
        opt.apply(self.p, grads, sparse)
        self.p['dense_in_bn/moving_mean'] = cache['new_mm']
        self.p['dense_in_bn/moving_variance'] = cache['new_mv']
        return {'loss': ce, 'L2': l2, 'accuracy': acc, 'lr': opt.lr}, grads, probs

    def test_step(self, data, y_ids):
        probs, _ = self.forward(data, training=False)
        ce, acc = self.metrics(probs, y_ids)
        return {'loss': ce, 'L2': self.l2_loss(), 'accuracy': acc}, probs

    # ---- greedy_predict_fc (lc_NIC.py:511-542): returns ids (max_len, B, 1)
    def greedy_predict(self, x, a0, c0, start_seq, max_len):
        p = self.p
        dt = p['lstm/kernel'].dtype
        f, _, _ = self.encode(x.astype(dt), False, DropCtx(training=False))
        Wf = p['lstm/kernel']
        assert self.Ef == self.Et, "the feature is fed through the text LSTM kernel (:520)"
        Ul, bl = p['lstm/recurrent_kernel'], p['lstm/bias']
        a, c, _ = O.lstm_step_fwd(f @ Wf + bl, a0.astype(dt), c0.astype(dt), Ul)       # :520
        word = np.asarray(start_seq).reshape(-1)
        B = word.shape[0]
        m = np.ones((B, 1), bool)                   # first step: mask lost through the Lambda expand (:515-516)
        frozen = False
        outs = []
        for _ in range(max_len):
            if frozen:
                outs.append(np.zeros((B, 1), np.int64))                                 # :526-527
                continue
            e = p['emb_text/embeddings'][word]
            h2, c2, _ = O.lstm_step_fwd(e @ Wf + bl, a, c, Ul)                          # :529
            a = np.where(m, h2, a)
            c = np.where(m, c2, c)
            inter, _ = O.dense_fwd(a, p['time_distributed_nonlinear/kernel'], p['time_distributed_nonlinear/bias'],
                                   O.ACT_LEAKY)
            probs = O.softmax(inter @ p['time_distributed_softmax/kernel'] + p['time_distributed_softmax/bias'])
            word = probs.argmax(axis=-1)                                                 # :535
            outs.append(word[:, None].astype(np.int64))
            m = (word != 0)[:, None]
            frozen = bool(np.all(word == 0))
        return np.stack(outs, axis=0)
