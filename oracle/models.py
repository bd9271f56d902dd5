"""Model-level CPU oracle: forward, explicit backward, train/test step, greedy decode.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED.

Two model forms, both restated from the reference (paths relative to /root/reference):
  * ``NICDense``  -- AttemptFour/Model/NIC.py:22-299   (BASELINE config 2: dense voxel encoder
                     + BatchNorm, feature as LSTM step 0, masked text LSTM, softmax head)
  * ``LcNIC``     -- AttemptFour/Model/lc_NIC.py:42-638 + layers.py + attention.py
                     (BASELINE config 3: region-wise encoder + additive attention + LSTM)
Parameters live in a dict keyed by keras layer/weight names, in keras layouts
(kernel = (in, out); LSTM gates i,f,c~,o concatenated on the last axis).
"""
import numpy as np
from . import ops as O
from .philox import keep_mask

# dropout site ids (shared with the product's host code; see csrc/tnt_rng.h)
S_IN, S_FEAT, S_TEXT, S_OUT = 1, 2, 3, 5
S_ATTN, S_LSTM_IN, S_LSTM_OUT = 16, 48, 80
S_SAMPLE = 112          # + decode position: the categorical-sampling stream of sample_predict
S_DEEP = 6              # + layer index (< 10): feature dropout after deep layer i of the depth-n encoder


class DropCtx:
    """Dropout stream context: masks = philox(seed, site, step)."""

    def __init__(self, seed=0, step=0, training=False):
        self.seed, self.step, self.training = seed, step, training

    def mask(self, shape, rate, site):
        if not self.training or rate <= 0.0:
            return None
        return keep_mask(shape, rate, self.seed, site, self.step)


def _l2(lam, W):
    return lam * (W * W).sum()


def apply_agc(model, grads, sparse, emb='emb_text/embeddings'):
    """gradients = agc.adaptive_clip_grad(trainable_variables, gradients, clip_factor, eps) -- the commented-out call
    of lc_NIC.py:388 (agc.py:20-38), enabled by ``model.agc = (clip_factor, eps)``.  The Embedding gradient is an
    IndexedSlices there: its rows are clipped with unit norms taken over the un-deduplicated rows, and the norm that
    clip-by-norm uses afterwards is the norm of those clipped rows."""
    if not getattr(model, 'agc', None):
        return grads, sparse
    cf, eps = model.agc
    out = {}
    sparse = dict(sparse or {})
    for k, g in grads.items():
        if g is None:
            out[k] = None
        elif k == emb and getattr(model, 'last_emb_rows', None) is not None:
            rows, flat = model.last_emb_rows
            rows2 = O.adaptive_clip_grad(model.p[k], rows, cf, eps)
            dense = np.zeros_like(model.p[k])
            np.add.at(dense, flat, rows2)
            out[k] = dense
            sparse[k] = np.sqrt((rows2 * rows2).sum())
        else:
            out[k] = O.adaptive_clip_grad(model.p[k], g, cf, eps)
    return out, sparse


class AdamState:
    """keras Adam(lr, b1, b2, eps, clipnorm) -- main.py:97.  clip=None disables clipping
    (TF<=2.3 behaviour, SURVEY 9.9 version hazard)."""

    def __init__(self, params, lr=1e-4, b1=0.9, b2=0.98, eps=1e-8, clipnorm=0.1):
        self.lr, self.b1, self.b2, self.eps, self.clipnorm = lr, b1, b2, eps, clipnorm
        self.t = 0
        self.m = {k: np.zeros_like(v) for k, v in params.items()}
        self.v = {k: np.zeros_like(v) for k, v in params.items()}

    def apply(self, params, grads, sparse_norms=None):
        """grads: dict name -> dense gradient.  sparse_norms: dict name -> l2 norm to use for
        clipping instead of the dense gradient's (the IndexedSlices quirk)."""
        self.t += 1
        for k, g in grads.items():
            if g is None:
                continue
            if self.clipnorm is not None:
                n = np.sqrt((g * g).sum())
                if sparse_norms is not None and k in sparse_norms:
                    n = sparse_norms[k]
                g = g * self.clipnorm / np.maximum(n, self.clipnorm)
            params[k], self.m[k], self.v[k] = O.adam_update(
                params[k], self.m[k], self.v[k], g, self.t, self.lr, self.b1, self.b2, self.eps)


# =========================================================================== NIC.py
class NICDense:
    """AttemptFour/Model/NIC.py.  ctor args follow NIC.py:22."""

    TRAINABLE = ['dense_img/kernel', 'dense_img/bias', 'batch_norm/gamma', 'batch_norm/beta',
                 'emb_text/embeddings', 'lstm/kernel', 'lstm/recurrent_kernel', 'lstm/bias',
                 'time_distributed_softmax/kernel', 'time_distributed_softmax/bias']

    def __init__(self, input_size, units, embedding_dim, vocab_size, max_length, dropout_input,
                 dropout, dropout_lstm, input_reg, lstm_reg, output_reg, norm='batch'):
        self.N, self.U, self.E, self.V, self.T = input_size, units, embedding_dim, vocab_size, max_length
        self.r_in, self.r_feat, self.r_lstm = dropout_input, dropout, dropout_lstm
        # NIC.py:55 -- the output layer's L2 uses lstm_reg (quirk kept)
        self.l2_in, self.l2_lstm, self.l2_out = input_reg, lstm_reg, lstm_reg
        self.norm = norm
        self.p = {}

    def init_params(self, rng, dtype=np.float64):
        N, U, E, V = self.N, self.U, self.E, self.V
        p = self.p
        p['dense_img/kernel'] = (rng.standard_normal((N, E)) * np.sqrt(2.0 / (N + E))).astype(dtype)
        p['dense_img/bias'] = (0.01 * rng.standard_normal(E)).astype(dtype)
        p['batch_norm/gamma'] = (1 + 0.1 * rng.standard_normal(E)).astype(dtype)
        p['batch_norm/beta'] = (0.1 * rng.standard_normal(E)).astype(dtype)
        p['batch_norm/moving_mean'] = np.zeros(E, dtype)
        p['batch_norm/moving_variance'] = np.ones(E, dtype)
        p['emb_text/embeddings'] = rng.uniform(-0.05, 0.05, (V, E)).astype(dtype)
        lim = np.sqrt(6.0 / (E + 4 * U))
        p['lstm/kernel'] = rng.uniform(-lim, lim, (E, 4 * U)).astype(dtype)
        p['lstm/recurrent_kernel'] = (rng.standard_normal((U, 4 * U)) / np.sqrt(U)).astype(dtype)
        b = 0.01 * rng.standard_normal(4 * U)
        b[U:2 * U] += 1.0
        p['lstm/bias'] = b.astype(dtype)
        p['time_distributed_softmax/kernel'] = (rng.standard_normal((U, V)) * np.sqrt(2.0 / (U + V))).astype(dtype)
        p['time_distributed_softmax/bias'] = (0.01 * rng.standard_normal(V)).astype(dtype)
        return self

    # ---- forward (NIC.py:100-145)
    def forward(self, data, training=False, drop=None):
        x, ids, a0, c0 = data
        p = self.p
        dt = p['dense_img/kernel'].dtype
        x = x.astype(dt)
        B, T, U = x.shape[0], ids.shape[1], self.U
        drop = drop or DropCtx(training=training)
        k_in = drop.mask(x.shape, self.r_in, S_IN)
        xd = O.dropout_fwd(x, k_in, self.r_in)                                   # NIC.py:122
        y, pre = O.dense_fwd(xd, p['dense_img/kernel'], p['dense_img/bias'], O.ACT_LEAKY)  # :125
        k_feat = drop.mask(y.shape, self.r_feat, S_FEAT)
        yd = O.dropout_fwd(y, k_feat, self.r_feat)                                # :126
        if self.norm == 'batch':                                                  # :127-128
            f, bn_cache, mm, mv = O.batchnorm_fwd(yd, p['batch_norm/gamma'], p['batch_norm/beta'],
                                                  p['batch_norm/moving_mean'],
                                                  p['batch_norm/moving_variance'], training)
        else:
            f, bn_cache = O.layernorm_fwd(yd, p['batch_norm/gamma'], p['batch_norm/beta'])
            mm, mv = p['batch_norm/moving_mean'], p['batch_norm/moving_variance']
        emb = O.embedding_fwd(p['emb_text/embeddings'], ids)                      # :131
        mask = ids != 0                                                           # mask_zero
        Wl, Ul, bl = p['lstm/kernel'], p['lstm/recurrent_kernel'], p['lstm/bias']
        # lstm call 1: the feature as a single unmasked timestep (:138)
        k_l0 = drop.mask((B, 1, self.E), self.r_lstm, S_LSTM_IN + 0)
        f_d = O.dropout_fwd(f[:, None, :], k_l0, self.r_lstm)[:, 0, :]
        a, c, cache0 = O.lstm_step_fwd(f_d @ Wl + bl, a0.astype(dt), c0.astype(dt), Ul)
        # lstm call 2: the text, masked by the embedding mask (:140)
        k_l1 = drop.mask((B, T, self.E), self.r_lstm, S_LSTM_IN + 1)
        emb_d = O.dropout_fwd(emb, k_l1, self.r_lstm)
        outs, caches = [], []
        out_prev = np.zeros((B, U), dt)
        for t in range(T):
            h2, c2, ch = O.lstm_step_fwd(emb_d[:, t] @ Wl + bl, a, c, Ul)
            m = mask[:, t][:, None]
            a = np.where(m, h2, a)
            c = np.where(m, c2, c)
            out_prev = np.where(m, h2, out_prev)
            outs.append(out_prev)
            caches.append(ch)
        A = np.stack(outs, axis=1)                                                # (B,T,U)
        logits = A @ p['time_distributed_softmax/kernel'] + p['time_distributed_softmax/bias']
        probs = O.softmax(logits, axis=-1)                                        # :143
        cache = dict(xd=xd, pre=pre, k_feat=k_feat, bn=bn_cache, f=f, f_d=f_d, k_l0=k_l0, k_l1=k_l1,
                     emb_d=emb_d, mask=mask, cache0=cache0, caches=caches, A=A, ids=ids,
                     logits=logits, new_mm=mm, new_mv=mv)
        return probs, cache

    def l2_loss(self):
        p = self.p
        return (_l2(self.l2_in, p['dense_img/kernel']) + _l2(self.l2_lstm, p['lstm/kernel'])
                + _l2(self.l2_out, p['time_distributed_softmax/kernel']))

    def metrics(self, probs, y_ids):
        """NIC.py:233-243: per-timestep mean-over-batch CE and accuracy, summed, / T."""
        T = y_ids.shape[1]
        ce = sum(O.cce_from_probs(probs[:, t], y_ids[:, t]).mean() for t in range(T)) / T
        acc = sum(O.accuracy(probs[:, t], y_ids[:, t]) for t in range(T)) / T
        return ce, acc

    # ---- backward of (CE + L2) wrt every trainable (tape.gradient, NIC.py:248-249)
    def backward(self, probs, cache, y_ids):
        p = self.p
        B, T = y_ids.shape
        U = self.U
        g = {}
        dl = np.full((B, T), 1.0 / (B * T), probs.dtype)
        dlogits = O.cce_softmax_bwd(probs, y_ids, dl)
        A = cache['A']
        Wo = p['time_distributed_softmax/kernel']
        g['time_distributed_softmax/kernel'] = A.reshape(-1, U).T @ dlogits.reshape(B * T, -1) + 2 * self.l2_out * Wo
        g['time_distributed_softmax/bias'] = dlogits.sum(axis=(0, 1))
        dA = dlogits @ Wo.T                                                        # (B,T,U)
        Wl, Ul = p['lstm/kernel'], p['lstm/recurrent_kernel']
        dWl = np.zeros_like(Wl)
        dUl = np.zeros_like(Ul)
        dbl = np.zeros_like(p['lstm/bias'])
        demb_d = np.zeros_like(cache['emb_d'])
        mask = cache['mask']
        da = np.zeros((B, U), probs.dtype)
        dc = np.zeros((B, U), probs.dtype)
        dout = np.zeros((B, U), probs.dtype)    # grad flowing into out_prev chain
        for t in reversed(range(T)):
            m = mask[:, t][:, None]
            dout = dout + dA[:, t]
            # forward: a=where(m,h2,a_prev); c=where(m,c2,c_prev); out=where(m,h2,out_prev)
            dh2 = np.where(m, da + dout, 0)
            dc2 = np.where(m, dc, 0)
            dz, dh_prev, dc_prev = O.lstm_step_bwd(dh2, dc2, cache['caches'][t], Ul)
            h_prev = cache['caches'][t][6]
            dWl += cache['emb_d'][:, t].T @ dz
            dUl += h_prev.T @ dz
            dbl += dz.sum(axis=0)
            demb_d[:, t] = dz @ Wl.T
            da = np.where(m, 0, da) + dh_prev
            dc = np.where(m, 0, dc) + dc_prev
            dout = np.where(m, 0, dout)
        # lstm call 1 (feature step)
        dz, _, _ = O.lstm_step_bwd(da, dc, cache['cache0'], Ul)
        h_prev0 = cache['cache0'][6]
        dWl += cache['f_d'].T @ dz
        dUl += h_prev0.T @ dz
        dbl += dz.sum(axis=0)
        df_d = dz @ Wl.T
        g['lstm/kernel'] = dWl + 2 * self.l2_lstm * Wl
        g['lstm/recurrent_kernel'] = dUl
        g['lstm/bias'] = dbl
        demb = O.dropout_bwd(demb_d, cache['k_l1'], self.r_lstm)
        rows, flat = O.embedding_bwd_rows(demb, cache['ids'])
        self.last_emb_rows = (rows, flat)
        g['emb_text/embeddings'] = O.embedding_bwd_dense(demb, cache['ids'], self.V)
        sparse_norm = np.sqrt((rows * rows).sum())
        df = O.dropout_bwd(df_d[:, None, :], cache['k_l0'], self.r_lstm)[:, 0, :]
        if self.norm == 'batch':
            dyd, dgam, dbet = O.batchnorm_bwd(df, p['batch_norm/gamma'], cache['bn'])
        else:
            dyd, dgam, dbet = O.layernorm_bwd(df, p['batch_norm/gamma'], cache['bn'])
        g['batch_norm/gamma'], g['batch_norm/beta'] = dgam, dbet
        dy = O.dropout_bwd(dyd, cache['k_feat'], self.r_feat)
        _, dK, db = O.dense_bwd(cache['xd'], p['dense_img/kernel'], cache['pre'], dy, O.ACT_LEAKY, need_dx=False)
        g['dense_img/kernel'] = dK + 2 * self.l2_in * p['dense_img/kernel']
        g['dense_img/bias'] = db
        return g, {'emb_text/embeddings': sparse_norm}

    def train_step(self, data, y_ids, opt, drop=None):
        """NIC.train_step (NIC.py:198-252)."""
        drop = drop or DropCtx(training=True)
        probs, cache = self.forward(data, training=True, drop=drop)
        ce, acc = self.metrics(probs, y_ids)
        l2 = self.l2_loss()
        grads, sparse = self.backward(probs, cache, y_ids)
        grads, sparse = apply_agc(self, grads, sparse)
        self.last_sparse = sparse          # IndexedSlices norms used by clip-by-norm (for tests that re-apply Adam)
        opt.apply(self.p, grads, sparse)
        self.p['batch_norm/moving_mean'] = cache['new_mm']
        self.p['batch_norm/moving_variance'] = cache['new_mv']
        return {'loss': ce, 'L2': l2, 'accuracy': acc}, grads, probs

    def test_step(self, data, y_ids):
        """NIC.test_step (NIC.py:254-299)."""
        probs, _ = self.forward(data, training=False)
        ce, acc = self.metrics(probs, y_ids)
        return {'loss': ce, 'L2': self.l2_loss(), 'accuracy': acc}, probs

    def greedy_predict(self, x, a0, c0, start_seq, max_len):
        """NIC.greedy_predict (NIC.py:148-195), inference mode.  Returns probs (max_len,B,1,V).
        A predicted id 0 masks the next LSTM step (state carried, step output = zeros)."""
        p = self.p
        dt = p['dense_img/kernel'].dtype
        y, _ = O.dense_fwd(x.astype(dt), p['dense_img/kernel'], p['dense_img/bias'], O.ACT_LEAKY)
        if self.norm == 'batch':
            f, _, _, _ = O.batchnorm_fwd(y, p['batch_norm/gamma'], p['batch_norm/beta'],
                                         p['batch_norm/moving_mean'], p['batch_norm/moving_variance'], False)
        else:
            f, _ = O.layernorm_fwd(y, p['batch_norm/gamma'], p['batch_norm/beta'])
        Wl, Ul, bl = p['lstm/kernel'], p['lstm/recurrent_kernel'], p['lstm/bias']
        a, c, _ = O.lstm_step_fwd(f @ Wl + bl, a0.astype(dt), c0.astype(dt), Ul)
        word = np.asarray(start_seq).reshape(-1)
        m = np.ones((word.shape[0], 1), bool)      # first call: mask lost through Lambda expand
        outs = []
        for _ in range(max_len):
            e = p['emb_text/embeddings'][word]
            h2, c2, _ = O.lstm_step_fwd(e @ Wl + bl, a, c, Ul)
            whole = np.where(m, h2, 0)
            a = np.where(m, h2, a)
            c = np.where(m, c2, c)
            probs = O.softmax(whole @ p['time_distributed_softmax/kernel'] + p['time_distributed_softmax/bias'])
            outs.append(probs[:, None, :])
            word = probs.argmax(axis=-1)
            m = (word != 0)[:, None]
        return np.stack(outs, axis=0)


# ======================================================================== lc_NIC.py
class LcNIC:
    """AttemptFour/Model/lc_NIC.py (attention path).  ctor args follow lc_NIC.py:42;
    ``groups`` = (list_of_index_arrays, list_of_out_dims) as load_avg_betas.get_groups returns."""

    H = 256  # hard-coded dense_inter width, lc_NIC.py:141

    def __init__(self, groups, units, embedding_features, embedding_text, attn_units, vocab_size,
                 max_length, dropout_input, dropout_features, dropout_text, dropout_attn,
                 dropout_lstm, dropout_out, input_reg, attn_reg, lstm_reg, output_reg, norm='batch', depth=0, use_layer_norm=False):
        self.use_layer_norm = bool(use_layer_norm)      # lc_NIC.py:115,126-136: tfa LayerNormLSTMCell as the decoder cell
        self.depth = int(depth)       # deep_layers.LocallyDense(depth=n): n more per-region Dense + BN + Dropout stages
        self.groups = [np.asarray(gi, dtype=np.int64) for gi in groups[0]]
        self.D = int(groups[1][0])
        assert all(int(d) == self.D for d in groups[1])
        self.R = len(self.groups)
        self.U, self.Et, self.A, self.V, self.T = units, embedding_text, attn_units, vocab_size, max_length
        self.r_in, self.r_feat, self.r_text = dropout_input, dropout_features, dropout_text
        self.r_attn, self.r_lstm, self.r_out = dropout_attn, dropout_lstm, dropout_out
        self.l2_in, self.l2_attn, self.l2_lstm, self.l2_out = input_reg, attn_reg, lstm_reg, output_reg
        self.norm = norm
        self.p = {}

    def trainable(self):
        return [k for k in self.p if 'moving_' not in k]

    def init_params(self, rng, dtype=np.float64):
        p = self.p
        D, A, U, Et, V, H = self.D, self.A, self.U, self.Et, self.V, self.H
        for r, idx in enumerate(self.groups):
            p[f'dense_in/{r}/kernel'] = (rng.standard_normal((len(idx), D)) * np.sqrt(2.0 / len(idx))).astype(dtype)
            p[f'dense_in/{r}/bias'] = (0.01 * rng.standard_normal(D)).astype(dtype)
        p['input_bn/gamma'] = (1 + 0.1 * rng.standard_normal(D)).astype(dtype)
        p['input_bn/beta'] = (0.1 * rng.standard_normal(D)).astype(dtype)
        p['input_bn/moving_mean'] = np.zeros(D, dtype)
        p['input_bn/moving_variance'] = np.ones(D, dtype)
        for i in range(self.depth):                                                    # deep_layers.py:42-51
            for r in range(self.R):
                p[f'dense_in/deep{i}/{r}/kernel'] = (rng.standard_normal((D, D)) * np.sqrt(2.0 / D)).astype(dtype)
                p[f'dense_in/deep{i}/{r}/bias'] = (0.01 * rng.standard_normal(D)).astype(dtype)
            p[f'input_bn/deep{i}/gamma'] = (1 + 0.1 * rng.standard_normal(D)).astype(dtype)
            p[f'input_bn/deep{i}/beta'] = (0.1 * rng.standard_normal(D)).astype(dtype)
            p[f'input_bn/deep{i}/moving_mean'] = np.zeros(D, dtype)
            p[f'input_bn/deep{i}/moving_variance'] = np.ones(D, dtype)
        p['attention/W1/kernel'] = (rng.standard_normal((D, A)) * np.sqrt(2.0 / D)).astype(dtype)
        p['attention/W1/bias'] = (0.01 * rng.standard_normal(A)).astype(dtype)
        p['attention/W2/kernel'] = (rng.standard_normal((U, A)) * np.sqrt(2.0 / U)).astype(dtype)
        p['attention/W2/bias'] = (0.01 * rng.standard_normal(A)).astype(dtype)
        p['attention/V/kernel'] = rng.uniform(-1, 1, (A, 1)).astype(dtype) * np.sqrt(6.0 / (A + 1))
        p['attention/V/bias'] = (0.01 * rng.standard_normal(1)).astype(dtype)
        p['emb_text/embeddings'] = rng.uniform(-0.08, 0.08, (V, Et)).astype(dtype)
        lim = np.sqrt(6.0 / (D + Et + 4 * U))
        p['lstm/kernel'] = rng.uniform(-lim, lim, (D + Et, 4 * U)).astype(dtype)
        p['lstm/recurrent_kernel'] = (rng.standard_normal((U, 4 * U)) / np.sqrt(U)).astype(dtype)
        b = 0.01 * rng.standard_normal(4 * U)
        b[U:2 * U] += 1.0
        p['lstm/bias'] = b.astype(dtype)
        if self.use_layer_norm:
            for nm, n in (('kernel_norm', 4 * U), ('recurrent_norm', 4 * U), ('state_norm', U)):
                p[f'lstm/{nm}/gamma'] = (1 + 0.1 * rng.standard_normal(n)).astype(dtype)
                p[f'lstm/{nm}/beta'] = (0.1 * rng.standard_normal(n)).astype(dtype)
        p['time_distributed_nonlinear/kernel'] = (rng.standard_normal((U, H)) * np.sqrt(2.0 / (U + H))).astype(dtype)
        p['time_distributed_nonlinear/bias'] = (0.01 * rng.standard_normal(H)).astype(dtype)
        p['time_distributed_softmax/kernel'] = (rng.standard_normal((H, V)) * np.sqrt(2.0 / (H + V))).astype(dtype)
        p['time_distributed_softmax/bias'] = (0.01 * rng.standard_normal(V)).astype(dtype)
        return self

    def _encode(self, x, training, drop):
        """dropout_input -> layers.LocallyDense.call (layers.py:43-53)."""
        p = self.p
        k_in = drop.mask(x.shape, self.r_in, S_IN)
        xd = O.dropout_fwd(x, k_in, self.r_in)                                     # lc_NIC.py:227
        Ws = [p[f'dense_in/{r}/kernel'] for r in range(self.R)]
        bs = [p[f'dense_in/{r}/bias'] for r in range(self.R)]
        y, pre = O.locally_dense_fwd(xd, self.groups, Ws, bs)                      # layers.py:45-48
        if self.norm == 'batch':
            bn, bn_cache, mm, mv = O.batchnorm_fwd(y, p['input_bn/gamma'], p['input_bn/beta'],
                                                   p['input_bn/moving_mean'],
                                                   p['input_bn/moving_variance'], training)  # :49
        else:
            bn, bn_cache = O.layernorm_fwd(y, p['input_bn/gamma'], p['input_bn/beta'])
            mm, mv = p['input_bn/moving_mean'], p['input_bn/moving_variance']
        k_feat = drop.mask(bn.shape, self.r_feat, S_FEAT)
        F = O.dropout_fwd(bn, k_feat, self.r_feat)                                  # layers.py:51
        deep = []
        B, R, D = F.shape
        for i in range(self.depth):                                                 # deep_layers.one_layer (:53-59)
            Wd = [p[f'dense_in/deep{i}/{r}/kernel'] for r in range(R)]
            bd = [p[f'dense_in/deep{i}/{r}/bias'] for r in range(R)]
            idg = [np.arange(r * D, (r + 1) * D) for r in range(R)]                 # layer(x[:, region, :])
            xin = F.reshape(B, R * D)
            y2, pre2 = O.locally_dense_fwd(xin, idg, Wd, bd)
            if self.norm == 'batch':
                bn2, c2, mm2, mv2 = O.batchnorm_fwd(y2, p[f'input_bn/deep{i}/gamma'], p[f'input_bn/deep{i}/beta'],
                                                     p[f'input_bn/deep{i}/moving_mean'],
                                                     p[f'input_bn/deep{i}/moving_variance'], training)
            else:
                bn2, c2 = O.layernorm_fwd(y2, p[f'input_bn/deep{i}/gamma'], p[f'input_bn/deep{i}/beta'])
                mm2, mv2 = p[f'input_bn/deep{i}/moving_mean'], p[f'input_bn/deep{i}/moving_variance']
            k2 = drop.mask(bn2.shape, self.r_feat, S_DEEP + i)
            F = O.dropout_fwd(bn2, k2, self.r_feat)
            deep.append(dict(xin=xin, idg=idg, pre=pre2, bn=c2, k=k2, new_mm=mm2, new_mv=mv2))
        return F, dict(xd=xd, pre=pre, bn=bn_cache, k_feat=k_feat, new_mm=mm, new_mv=mv, deep=deep)

    # ---- lc_NIC.call_attention (lc_NIC.py:223-263)
    def forward(self, data, training=False, drop=None):
        x, ids, a0, c0 = data
        p = self.p
        dt = p['lstm/kernel'].dtype
        x = x.astype(dt)
        B, T = ids.shape
        drop = drop or DropCtx(training=training)
        F, enc = self._encode(x, training, drop)
        out, cache = self._decode_fwd(F, ids, a0, c0, training, drop)
        cache['enc'] = enc
        return out, cache

    def _decode_fwd(self, F, ids, a0, c0, training, drop):
        """Everything of call_attention after the encoder (lc_NIC.py:233-263)."""
        p = self.p
        dt = p['lstm/kernel'].dtype
        B, T = ids.shape
        emb = O.embedding_fwd(p['emb_text/embeddings'], ids)
        k_text = drop.mask(emb.shape, self.r_text, S_TEXT)
        text = O.dropout_fwd(emb, k_text, self.r_text)                              # :233
        P, Ppre = O.attention_proj_fwd(F, p['attention/W1/kernel'], p['attention/W1/bias'])
        a, c = a0.astype(dt), c0.astype(dt)
        Wl, Ul, bl = p['lstm/kernel'], p['lstm/recurrent_kernel'], p['lstm/bias']
        steps, outs, alphas = [], [], []
        for i in range(T):                                                          # :244-256
            k_at = drop.mask((B, self.R, self.A), self.r_attn, S_ATTN + i)
            (ctx, alpha, _), acache = O.attention_step_fwd(
                a, F, P, p['attention/W2/kernel'], p['attention/W2/bias'],
                p['attention/V/kernel'], p['attention/V/bias'], k_at, self.r_attn)
            sample = np.concatenate([ctx, text[:, i]], axis=1)                      # :253
            if self.use_layer_norm:          # LayerNormLSTMCell(units, kernel_regularizer): no dropout inside the cell
                k_li, sample_d = None, sample
                a, c, lcache = O.ln_lstm_step_fwd(sample, a, c, Wl, Ul, bl, *[p[f'lstm/{n}/{w}'] for n in
                                                  ('kernel_norm', 'recurrent_norm', 'state_norm') for w in ('gamma', 'beta')])
            else:
                k_li = drop.mask((B, 1, sample.shape[1]), self.r_lstm, S_LSTM_IN + i)
                sample_d = O.dropout_fwd(sample[:, None, :], k_li, self.r_lstm)[:, 0]
                a, c, lcache = O.lstm_step_fwd(sample_d @ Wl + bl, a, c, Ul)        # :255
            k_lo = drop.mask((B, self.U), self.r_lstm, S_LSTM_OUT + i)
            outs.append(O.dropout_fwd(a, k_lo, self.r_lstm))                        # :256
            alphas.append(alpha)
            steps.append(dict(acache=acache, sample_d=sample_d, k_li=k_li, lcache=lcache, k_lo=k_lo))
        Hs = np.stack(outs, axis=1)                                                 # (B,T,U)
        inter, ipre = O.dense_fwd(Hs, p['time_distributed_nonlinear/kernel'],
                                  p['time_distributed_nonlinear/bias'], O.ACT_LEAKY)
        k_out = drop.mask(inter.shape, self.r_out, S_OUT)
        inter_d = O.dropout_fwd(inter, k_out, self.r_out)
        logits = inter_d @ p['time_distributed_softmax/kernel'] + p['time_distributed_softmax/bias']
        probs = O.softmax(logits, axis=-1)                                          # :261
        attn = np.stack(alphas, axis=0)[..., None]                                  # (T,B,R,1) :263
        cache = dict(F=F, P=P, Ppre=Ppre, k_text=k_text, steps=steps, Hs=Hs, ipre=ipre,
                     k_out=k_out, inter_d=inter_d, ids=ids, logits=logits)
        return (probs, attn), cache

    def _cell(self, sample, a, c):
        """one inference step of the decoder cell: keras LSTM, or tfa LayerNormLSTMCell with use_layer_norm"""
        p = self.p
        if self.use_layer_norm:
            h2, c2, _ = O.ln_lstm_step_fwd(sample, a, c, p['lstm/kernel'], p['lstm/recurrent_kernel'], p['lstm/bias'],
                                           *[p[f'lstm/{n}/{w}'] for n in ('kernel_norm', 'recurrent_norm', 'state_norm')
                                             for w in ('gamma', 'beta')])
            return h2, c2
        h2, c2, _ = O.lstm_step_fwd(sample @ p['lstm/kernel'] + p['lstm/bias'], a, c, p['lstm/recurrent_kernel'])
        return h2, c2

    def l2_loss(self):
        p = self.p
        s = sum(_l2(self.l2_in, p[f'dense_in/{r}/kernel']) for r in range(self.R))
        s += sum(_l2(self.l2_in, p[f'dense_in/deep{i}/{r}/kernel']) for i in range(self.depth) for r in range(self.R))
        s += _l2(self.l2_attn, p['attention/W1/kernel']) + _l2(self.l2_attn, p['attention/W2/kernel'])
        s += _l2(self.l2_lstm, p['lstm/kernel'])
        s += _l2(self.l2_out, p['time_distributed_nonlinear/kernel'])
        s += _l2(self.l2_out, p['time_distributed_softmax/kernel'])
        return s

    def metrics(self, probs, attn, y_ids):
        """lc_NIC.py:365-376: CE/accuracy as NIC.py; attention metric = MSE(1, sum over the
        *batch* axis of alpha) (shape quirk kept)."""
        T = y_ids.shape[1]
        ce = sum(O.cce_from_probs(probs[:, t], y_ids[:, t]).mean() for t in range(T)) / T
        acc = sum(O.accuracy(probs[:, t], y_ids[:, t]) for t in range(T)) / T
        across = attn[..., 0].sum(axis=1)                                           # (T,R)
        attn_loss = ((1.0 - across) ** 2).mean()
        return ce, acc, attn_loss

    def backward(self, probs, cache, y_ids):
        g, sparse, dF = self._decode_bwd(probs, cache, y_ids)
        self._encode_bwd(dF, cache['enc'], g)
        return g, sparse

    def _decode_bwd(self, probs, cache, y_ids):
        p = self.p
        B, T = y_ids.shape
        U, D = self.U, self.D
        g = {}
        dl = np.full((B, T), 1.0 / (B * T), probs.dtype)
        dlogits = O.cce_softmax_bwd(probs, y_ids, dl)
        Wo, Wi = p['time_distributed_softmax/kernel'], p['time_distributed_nonlinear/kernel']
        g['time_distributed_softmax/kernel'] = (cache['inter_d'].reshape(B * T, -1).T
                                                @ dlogits.reshape(B * T, -1) + 2 * self.l2_out * Wo)
        g['time_distributed_softmax/bias'] = dlogits.sum(axis=(0, 1))
        dinter = O.dropout_bwd(dlogits @ Wo.T, cache['k_out'], self.r_out)
        dHs, dWi, dbi = O.dense_bwd(cache['Hs'], Wi, cache['ipre'], dinter, O.ACT_LEAKY)
        g['time_distributed_nonlinear/kernel'] = dWi + 2 * self.l2_out * Wi
        g['time_distributed_nonlinear/bias'] = dbi
        Wl, Ul = p['lstm/kernel'], p['lstm/recurrent_kernel']
        W2, v = p['attention/W2/kernel'], p['attention/V/kernel']
        dWl, dUl, dbl = np.zeros_like(Wl), np.zeros_like(Ul), np.zeros_like(p['lstm/bias'])
        dW2, db2 = np.zeros_like(W2), np.zeros_like(p['attention/W2/bias'])
        dv, dbv = np.zeros_like(v), np.zeros_like(p['attention/V/bias'])
        F = cache['F']
        dF = np.zeros_like(F)
        dP = np.zeros_like(cache['P'])
        dtext = np.zeros((B, T, self.Et), probs.dtype)
        da = np.zeros((B, U), probs.dtype)
        dc = np.zeros((B, U), probs.dtype)
        dln = {}
        for i in reversed(range(T)):
            st = cache['steps'][i]
            dh2 = da + O.dropout_bwd(dHs[:, i], st['k_lo'], self.r_lstm)
            if self.use_layer_norm:
                dsample, dh_prev, dc, gl = O.ln_lstm_step_bwd(dh2, dc, st['lcache'], Wl, Ul, p['lstm/kernel_norm/gamma'],
                                                              p['lstm/recurrent_norm/gamma'], p['lstm/state_norm/gamma'])
                dWl += gl['W']; dUl += gl['U']; dbl += gl['b']
                for nm, kk in (('kernel_norm', 'k'), ('recurrent_norm', 'r'), ('state_norm', 's')):
                    dln[f'lstm/{nm}/gamma'] = dln.get(f'lstm/{nm}/gamma', 0) + gl['g' + kk]
                    dln[f'lstm/{nm}/beta'] = dln.get(f'lstm/{nm}/beta', 0) + gl['b' + kk]
            else:
                dz, dh_prev, dc = O.lstm_step_bwd(dh2, dc, st['lcache'], Ul)
                h_prev = st['lcache'][6]
                dWl += st['sample_d'].T @ dz
                dUl += h_prev.T @ dz
                dbl += dz.sum(axis=0)
                dsample = O.dropout_bwd((dz @ Wl.T)[:, None, :], st['k_li'], self.r_lstm)[:, 0]
            dctx, dtext[:, i] = dsample[:, :D], dsample[:, D:]
            am = getattr(self, '_alpha_mse', 0.0)       # train_step_sam, first pass: d/dalpha of mean (1 - alpha)^2
            dh_att, dF_i, dsum, dW2_i, db2_i, dv_i, dbv_i = O.attention_step_bwd(
                dctx, F, W2, v, st['acache'], dalpha_ext=(am * (st['acache'][4] - 1.0)) if am else None)
            dF += dF_i
            dP += dsum
            dW2 += dW2_i
            db2 += db2_i
            dv += dv_i
            dbv += dbv_i
            da = dh_prev + dh_att
        g['lstm/kernel'] = dWl + 2 * self.l2_lstm * Wl
        g['lstm/recurrent_kernel'], g['lstm/bias'] = dUl, dbl
        g.update(dln)
        g['attention/W2/kernel'] = dW2 + 2 * self.l2_attn * W2
        g['attention/W2/bias'] = db2
        g['attention/V/kernel'], g['attention/V/bias'] = dv, dbv
        W1 = p['attention/W1/kernel']
        dF1, dW1, db1 = O.dense_bwd(F, W1, cache['Ppre'], dP, O.ACT_LEAKY)
        dF += dF1
        g['attention/W1/kernel'] = dW1 + 2 * self.l2_attn * W1
        g['attention/W1/bias'] = db1
        demb = O.dropout_bwd(dtext, cache['k_text'], self.r_text)
        rows, flat_ = O.embedding_bwd_rows(demb, cache['ids'])
        self.last_emb_rows = (rows, flat_)
        g['emb_text/embeddings'] = O.embedding_bwd_dense(demb, cache['ids'], self.V)
        sparse = {'emb_text/embeddings': np.sqrt((rows * rows).sum())}
        return g, sparse, dF

    def _encode_bwd(self, dF, enc, g, prefix=''):
        p = self.p
        B, R, D = dF.shape
        for i in reversed(range(self.depth)):
            dc = enc['deep'][i]
            d2 = O.dropout_bwd(dF, dc['k'], self.r_feat)
            if self.norm == 'batch':
                dy2, dgam, dbet = O.batchnorm_bwd(d2, p[f'input_bn/deep{i}/gamma'], dc['bn'])
            else:
                dy2, dgam, dbet = O.layernorm_bwd(d2, p[f'input_bn/deep{i}/gamma'], dc['bn'])
            g[f'input_bn/deep{i}/gamma'], g[f'input_bn/deep{i}/beta'] = dgam, dbet
            dWs, dbs = O.locally_dense_bwd(dc['xin'], dc['idg'], dc['pre'], dy2)
            dpre = O.act_bwd(dc['pre'], dy2, O.ACT_LEAKY, 0.2)
            dF = np.empty_like(dF)
            for r in range(R):
                W = p[f'dense_in/deep{i}/{r}/kernel']
                g[f'dense_in/deep{i}/{r}/kernel'] = dWs[r] + 2 * self.l2_in * W
                g[f'dense_in/deep{i}/{r}/bias'] = dbs[r]
                dF[:, r, :] = dpre[:, r, :] @ W.T
        dbn = O.dropout_bwd(dF, enc['k_feat'], self.r_feat)
        if self.norm == 'batch':
            dy, dgam, dbet = O.batchnorm_bwd(dbn, p['input_bn/gamma'], enc['bn'])
        else:
            dy, dgam, dbet = O.layernorm_bwd(dbn, p['input_bn/gamma'], enc['bn'])
        g['input_bn/gamma'], g['input_bn/beta'] = dgam, dbet
        dWs, dbs = O.locally_dense_bwd(enc['xd'], self.groups, enc['pre'], dy)
        for r in range(self.R):
            g[f'dense_in/{r}/kernel'] = dWs[r] + 2 * self.l2_in * p[f'dense_in/{r}/kernel']
            g[f'dense_in/{r}/bias'] = dbs[r]

    def train_step(self, data, y_ids, opt, drop=None):
        """lc_NIC.train_step (lc_NIC.py:328-408)."""
        drop = drop or DropCtx(training=True)
        (probs, attn), cache = self.forward(data, training=True, drop=drop)
        ce, acc, al = self.metrics(probs, attn, y_ids)
        l2 = self.l2_loss()
        grads, sparse = self.backward(probs, cache, y_ids)
        grads, sparse = apply_agc(self, grads, sparse)
        self.last_sparse = sparse          # IndexedSlices norms used by clip-by-norm (for tests that re-apply Adam)
        opt.apply(self.p, grads, sparse)
        self.p['input_bn/moving_mean'] = cache['enc']['new_mm']
        self.p['input_bn/moving_variance'] = cache['enc']['new_mv']
        for i, dc in enumerate(cache['enc'].get('deep', [])):
            self.p[f'input_bn/deep{i}/moving_mean'], self.p[f'input_bn/deep{i}/moving_variance'] = dc['new_mm'], dc['new_mv']
        return {'loss': ce, 'L2': l2, 'accuracy': acc, 'attention': al, 'lr': opt.lr}, grads, (probs, attn)

    def train_step_sam(self, data, y_ids, opt, drop=None, rho=0.05):
        """lc_NIC.train_step_sam (lc_NIC.py:713-838): gradient of CE + L2 + MSE(ones, attention_scores) -> ascent step
        e_w = g * rho / (global_norm(g) + 1e-12), the Embedding's IndexedSlices counted by its un-deduplicated values
        (:768-786) -> gradient of CE + L2 at the perturbed weights (:800-830) -> restore -> apply_gradients (:833-836).
        Returns the second pass's CE, L2 (at the perturbed weights), accuracy and 'attention' = MSE(ones, alpha) (:838).
        Both passes draw the same dropout masks (TF would draw fresh ones; this library's stream is keyed by the step)."""
        drop = drop or DropCtx(training=True)
        (probs, attn), cache = self.forward(data, training=True, drop=drop)
        self._alpha_mse = 2.0 / attn.size
        g1, sp1 = self.backward(probs, cache, y_ids)
        self._alpha_mse = 0.0
        sq = sum(((sp1[k] ** 2) if k in sp1 else (g * g).sum()) for k, g in g1.items())
        scale = rho / (np.sqrt(sq) + 1e-12)
        e_ws = {k: g * scale for k, g in g1.items()}
        for k, e in e_ws.items():
            self.p[k] = self.p[k] + e
        bn1 = cache['enc']
        self.p['input_bn/moving_mean'], self.p['input_bn/moving_variance'] = bn1['new_mm'], bn1['new_mv']
        for i, dc in enumerate(bn1.get('deep', [])):
            self.p[f'input_bn/deep{i}/moving_mean'], self.p[f'input_bn/deep{i}/moving_variance'] = dc['new_mm'], dc['new_mv']
        (probs, attn), cache = self.forward(data, training=True, drop=drop)
        ce, acc, _ = self.metrics(probs, attn, y_ids)
        l2 = self.l2_loss()
        al = ((1.0 - attn) ** 2).mean()
        g2, sparse = self.backward(probs, cache, y_ids)
        for k, e in e_ws.items():
            self.p[k] = self.p[k] - e
        opt.apply(self.p, g2, sparse)
        self.p['input_bn/moving_mean'] = cache['enc']['new_mm']
        self.p['input_bn/moving_variance'] = cache['enc']['new_mv']
        for i, dc in enumerate(cache['enc'].get('deep', [])):
            self.p[f'input_bn/deep{i}/moving_mean'], self.p[f'input_bn/deep{i}/moving_variance'] = dc['new_mm'], dc['new_mv']
        return {'loss': ce, 'L2': l2, 'accuracy': acc, 'attention': al, 'lr': opt.lr}, g2

    def test_step(self, data, y_ids):
        """lc_NIC.test_step (lc_NIC.py:410-459)."""
        (probs, attn), _ = self.forward(data, training=False)
        ce, acc, al = self.metrics(probs, attn, y_ids)
        return {'loss': ce, 'L2': self.l2_loss(), 'accuracy': acc, 'attention': al}, (probs, attn)

    def greedy_predict(self, x, a0, c0, start_seq, max_len, sampler=None):
        """lc_NIC.greedy_predict_attention (lc_NIC.py:577-638), training=False.
        Returns (words (B,T,1) int64, probs (B,T,V), alpha (T,B,R,1), s (T,B,R,A)).
        sampler(probs, i) -> ids replaces the argmax (lc_NIC.sample_choice, lc_NIC.py:571-575)."""
        p = self.p
        dt = p['lstm/kernel'].dtype
        drop = DropCtx(training=False)
        F, _ = self._encode(x.astype(dt), False, drop)
        P, _ = O.attention_proj_fwd(F, p['attention/W1/kernel'], p['attention/W1/bias'])
        a, c = a0.astype(dt), c0.astype(dt)
        word = np.asarray(start_seq).reshape(-1)
        words, raws, alphas, ss = [], [], [], []
        for _ in range(max_len):
            text = p['emb_text/embeddings'][word]
            (ctx, alpha, s), _ = O.attention_step_fwd(a, F, P, p['attention/W2/kernel'], p['attention/W2/bias'],
                                                      p['attention/V/kernel'], p['attention/V/bias'])
            sample = np.concatenate([ctx, text], axis=1)
            a, c = self._cell(sample, a, c)
            inter, _ = O.dense_fwd(a, p['time_distributed_nonlinear/kernel'],
                                   p['time_distributed_nonlinear/bias'], O.ACT_LEAKY)
            probs = O.softmax(inter @ p['time_distributed_softmax/kernel'] + p['time_distributed_softmax/bias'])
            word = probs.argmax(axis=-1) if sampler is None else sampler(probs, len(words))
            words.append(word[:, None])
            raws.append(probs)
            alphas.append(alpha[..., None])
            ss.append(s)
        return (np.stack(words, axis=1).astype(np.int64), np.stack(raws, axis=1),
                np.stack(alphas, axis=0), np.stack(ss, axis=0))

    def beam_search(self, x, a0, c0, start_seq, max_len, k=5, end_id=-1):
        """Log-probability beam search of width k with the greedy decoder's step (lc_NIC.py:596-632).  The reference
        only sketches beam search (lc_NIC.py:640-692; ThinkAndTell/evaluate.py:203-228); this is the definition the
        product implements (include/tnt_hip.h: tnt_beam_topk_f32).  Returns (sequences (B,k,max_len), scores (B,k),
        margin (B,) = the smallest gap between the k-th kept and the best dropped candidate over all steps)."""
        p = self.p
        dt = p['lstm/kernel'].dtype
        B = x.shape[0]
        F, _ = self._encode(x.astype(dt), False, DropCtx(training=False))
        P, _ = O.attention_proj_fwd(F, p['attention/W1/kernel'], p['attention/W1/bias'])
        rep = lambda t: np.repeat(t, k, axis=0)
        F, P = rep(F), rep(P)
        a, c = rep(a0.astype(dt)), rep(c0.astype(dt))
        word = rep(np.asarray(start_seq).reshape(-1))
        score = np.zeros((B, k)); score[:, 1:] = -1e30
        fin = np.zeros((B, k), bool)
        seqs = np.zeros((B, k, 0), np.int64)
        margin = np.full(B, np.inf)
        V = self.V
        for _ in range(max_len):
            text = p['emb_text/embeddings'][word]
            (ctx, alpha, s), _ = O.attention_step_fwd(a, F, P, p['attention/W2/kernel'], p['attention/W2/bias'],
                                                      p['attention/V/kernel'], p['attention/V/bias'])
            a, c = self._cell(np.concatenate([ctx, text], axis=1), a, c)
            inter, _ = O.dense_fwd(a, p['time_distributed_nonlinear/kernel'], p['time_distributed_nonlinear/bias'], O.ACT_LEAKY)
            probs = O.softmax(inter @ p['time_distributed_softmax/kernel'] + p['time_distributed_softmax/bias'])
            cand = score[:, :, None] + np.log(np.maximum(probs, 1e-30)).reshape(B, k, V)
            frozen = np.full((B, k, V), -np.inf); frozen[:, :, 0] = score
            cand = np.where(fin[:, :, None], frozen, cand).reshape(B, k * V)
            order = np.argsort(-cand, axis=1, kind='stable')          # ties: lower flat index first
            top = order[:, :k]
            best = np.take_along_axis(cand, top, axis=1)
            nxt = np.take_along_axis(cand, order[:, k:k + 1], axis=1)[:, 0]
            margin = np.minimum(margin, np.min(best[:, :-1] - best[:, 1:], axis=1) if k > 1 else np.inf)
            margin = np.minimum(margin, best[:, -1] - nxt)
            pj, tv = top // V, top % V
            rows = (np.arange(B)[:, None] * k + pj).reshape(-1)
            a, c = a[rows], c[rows]
            seqs = np.concatenate([np.take_along_axis(seqs, pj[:, :, None], axis=1) if seqs.shape[2] else seqs,
                                   tv[:, :, None]], axis=2)
            fin = np.take_along_axis(fin, pj, axis=1) | (tv == end_id)
            score = best
            word = tv.reshape(-1)
        return seqs, score, margin
