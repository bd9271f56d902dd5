"""Multi-subject oracle: AttemptFour/Model/ms2_NIC.py generalised from 2 to S subjects.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED.

ms2_NIC.call (ms2_NIC.py:177-205) halves every input, runs ``call_attention_A`` /
``call_attention_B`` (207-291) -- one ``layers.LocallyDense`` encoder per subject, shared
embedding / attention / LSTM / head -- and ``train_step`` (294-374) averages the per-subject
cross-entropies.  Restated here as: S encoders on S equal batch slices, the shared decoder on
the re-concatenated features (rows are independent given the features, so this equals S
separate decoder calls), metrics per subject.  Quirks kept: the encoder output passes through
the feature Dropout twice (ms2_NIC.py:214 + layers.py:51), and the sub-calls always run with
training=True, also under test_step (ms2_NIC.py:195,202,419,426).
"""
import numpy as np

from . import ops as O
from .models import LcNIC, DropCtx, S_IN, S_FEAT, _l2

SUBJ_SITE = 1000      # dropout site offset per subject
S_FEAT2 = 4           # second application of the feature dropout


class MsLcNIC(LcNIC):
    def __init__(self, groups, *args, n_subjects=2, **kw):
        super().__init__(groups, *args, **kw)
        self.S = n_subjects

    def init_params(self, rng, dtype=np.float64):
        super().init_params(rng, dtype)
        p = self.p
        shared = {k: v for k, v in p.items() if not (k.startswith('dense_in/') or k.startswith('input_bn/'))}
        enc = {k: v for k, v in p.items() if k not in shared}
        self.p = dict(shared)
        for s in range(self.S):
            for k, v in enc.items():
                head, rest = k.split('/', 1)
                w = v if 'moving' in k else v * (1 + 0.05 * rng.standard_normal(v.shape)) + 0.01 * rng.standard_normal(v.shape)
                self.p[f'{head}_{s}/{rest}'] = w.astype(dtype)
        return self

    def trainable(self):
        return [k for k in self.p if 'moving_' not in k]

    def _encode_s(self, x, s, training, drop):
        p, pre_ = self.p, f'dense_in_{s}'
        off = SUBJ_SITE * (s + 1)
        k_in = drop.mask(x.shape, self.r_in, S_IN + off)
        xd = O.dropout_fwd(x, k_in, self.r_in)
        Ws = [p[f'{pre_}/{r}/kernel'] for r in range(self.R)]
        bs = [p[f'{pre_}/{r}/bias'] for r in range(self.R)]
        y, pre = O.locally_dense_fwd(xd, self.groups, Ws, bs)
        bnp = f'input_bn_{s}'
        if self.norm == 'batch':
            bn, bn_cache, mm, mv = O.batchnorm_fwd(y, p[f'{bnp}/gamma'], p[f'{bnp}/beta'], p[f'{bnp}/moving_mean'],
                                                   p[f'{bnp}/moving_variance'], training)
        else:
            bn, bn_cache = O.layernorm_fwd(y, p[f'{bnp}/gamma'], p[f'{bnp}/beta'])
            mm, mv = p[f'{bnp}/moving_mean'], p[f'{bnp}/moving_variance']
        k1 = drop.mask(bn.shape, self.r_feat, S_FEAT + off)
        f1 = O.dropout_fwd(bn, k1, self.r_feat)                       # layers.py:51
        k2 = drop.mask(bn.shape, self.r_feat, S_FEAT2 + off)
        F = O.dropout_fwd(f1, k2, self.r_feat)                        # ms2_NIC.py:214
        return F, dict(xd=xd, pre=pre, bn=bn_cache, k1=k1, k2=k2, new_mm=mm, new_mv=mv)

    def forward(self, data, training=True, drop=None):
        x, ids, a0, c0 = data
        dt = self.p['lstm/kernel'].dtype
        x = x.astype(dt)
        S = self.S
        Bs = x.shape[0] // S
        drop = drop or DropCtx(training=training)
        Fs, encs = [], []
        for s in range(S):
            F, enc = self._encode_s(x[s * Bs:(s + 1) * Bs], s, training, drop)
            Fs.append(F); encs.append(enc)
        out, cache = self._decode_fwd(np.concatenate(Fs, axis=0), ids, a0, c0, training, drop)
        cache['encs'] = encs
        return out, cache

    def l2_loss(self):
        p = self.p
        s = sum(_l2(self.l2_in, p[f'dense_in_{q}/{r}/kernel']) for q in range(self.S) for r in range(self.R))
        s += _l2(self.l2_attn, p['attention/W1/kernel']) + _l2(self.l2_attn, p['attention/W2/kernel'])
        s += _l2(self.l2_lstm, p['lstm/kernel'])
        s += _l2(self.l2_out, p['time_distributed_nonlinear/kernel']) + _l2(self.l2_out, p['time_distributed_softmax/kernel'])
        return s

    def metrics_ms(self, probs, attn, y_ids):
        S = self.S
        Bs = y_ids.shape[0] // S
        out = {}
        ces = []
        for s in range(S):
            sl = slice(s * Bs, (s + 1) * Bs)
            ce, acc, al = LcNIC.metrics(self, probs[sl], attn[:, sl], y_ids[sl])
            tag = chr(ord('A') + s)
            out[f'loss{tag}'], out[f'accuracy{tag}'], out[f'attention{tag}'] = ce, acc, al
            ces.append(ce)
        out['loss'] = sum(ces) / S
        return out

    def backward(self, probs, cache, y_ids):
        g, sparse, dF = self._decode_bwd(probs, cache, y_ids)
        p = self.p
        S = self.S
        Bs = y_ids.shape[0] // S
        for s in range(S):
            enc = cache['encs'][s]
            d = dF[s * Bs:(s + 1) * Bs]
            d = O.dropout_bwd(d, enc['k2'], self.r_feat)
            d = O.dropout_bwd(d, enc['k1'], self.r_feat)
            bnp = f'input_bn_{s}'
            if self.norm == 'batch':
                dy, dgam, dbet = O.batchnorm_bwd(d, p[f'{bnp}/gamma'], enc['bn'])
            else:
                dy, dgam, dbet = O.layernorm_bwd(d, p[f'{bnp}/gamma'], enc['bn'])
            g[f'{bnp}/gamma'], g[f'{bnp}/beta'] = dgam, dbet
            dWs, dbs = O.locally_dense_bwd(enc['xd'], self.groups, enc['pre'], dy)
            for r in range(self.R):
                k = f'dense_in_{s}/{r}/kernel'
                g[k] = dWs[r] + 2 * self.l2_in * p[k]
                g[f'dense_in_{s}/{r}/bias'] = dbs[r]
        return g, sparse

    def _commit_stats(self, cache):
        for s in range(self.S):
            self.p[f'input_bn_{s}/moving_mean'] = cache['encs'][s]['new_mm']
            self.p[f'input_bn_{s}/moving_variance'] = cache['encs'][s]['new_mv']

    def train_step(self, data, y_ids, opt, drop=None):
        """ms2_NIC.train_step (ms2_NIC.py:294-374)."""
        drop = drop or DropCtx(training=True)
        (probs, attn), cache = self.forward(data, True, drop)
        m = self.metrics_ms(probs, attn, y_ids)
        m['L2'] = self.l2_loss()
        grads, sparse = self.backward(probs, cache, y_ids)
        opt.apply(self.p, grads, sparse)
        self._commit_stats(cache)
        m['lr'] = opt.lr
        return m, grads, (probs, attn)

    def test_step(self, data, y_ids, drop=None):
        """ms2_NIC.test_step (ms2_NIC.py:376-465): sub-calls run with training=True (quirk)."""
        drop = drop or DropCtx(training=True)
        (probs, attn), cache = self.forward(data, True, drop)
        m = self.metrics_ms(probs, attn, y_ids)
        m['L2'] = self.l2_loss()
        self._commit_stats(cache)
        return m, (probs, attn)
