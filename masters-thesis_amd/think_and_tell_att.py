"""ThinkAndTell/att_model.py caption generator (GRU decoder) on the HIP kernels.

Drop-in for ``ThinkAndTell/att_model.py``: ``Encoder(embedding_dim, l2_reg, init_method, dropout)`` (31-52),
``Decoder(embedding_dim, units, vocab_size, l2_reg, init_method, dropout, use_stateful=False)`` (61-129: the GRU
path, :118), ``CaptionGenerator(encoder, decoder, tokenizer, max_length)`` with ``train_step`` (228-277) and
``test_step`` (279-304).  Data tuple ``(img_tensor, _, target)``, target = int ids (B, T).
Restated semantics and quirks: oracle/models_att.py.  ``BahdanauAttention`` (11-29) is an empty stub in the
reference and is never called.

Launch plan: Dense(relu) GEMM -> [dropout] -> embedding gather -> one input-projection GEMM for all T+1 steps
(input bias fused) -> T+1 fused GRU steps (tnt_gru_step_fwd_f32) -> [dropout] -> fc1 GEMM (relu) -> [dropout] ->
vocabulary GEMM (relu) -> fused masked sparse-CE-from-logits on rows 0..T-2 against target[:, 1:] -> mirror-image
backward (tnt_gru_step_bwd_f32 yields the input-side and the recurrent-side gate gradients) -> clip + Adam/SGD.
"""
from collections import OrderedDict

import numpy as np
import torch

from .arena import ParamArena
from .model_base import Metrics, S_FEAT, S_OUT
from .ops import ACT_RELU
from .think_and_tell import CaptionGenerator as _CaptionGeneratorLSTM, _r4


def interleave3(w, U):
    """keras [.., 3U] gate blocks (z, r, h) -> interleaved [.., U, 4] with a zero fourth slot."""
    lead = w.shape[:-1]
    out = np.zeros(lead + (U, 4), np.float32)
    out[..., :3] = np.moveaxis(np.asarray(w, np.float32).reshape(*lead, 3, U), -2, -1)
    return out


def deinterleave3(w):
    lead, U = w.shape[:-2], w.shape[-2]
    return np.ascontiguousarray(np.moveaxis(w[..., :3], -1, -2)).reshape(*lead, 3 * U)


class Encoder:
    """ThinkAndTell/att_model.py:31-52 descriptor (weights live in the CaptionGenerator's arena)."""

    def __init__(self, embedding_dim, l2_reg=0.0, init_method="glorot_uniform", dropout=0.0):
        self.embedding_dim, self.l2, self.init_method, self.dropout = int(embedding_dim), float(l2_reg), init_method, float(dropout)


class Decoder:
    """ThinkAndTell/att_model.py:61-129 descriptor."""

    def __init__(self, embedding_dim, units, vocab_size, l2_reg=0.0, init_method="glorot_uniform", dropout=0.0,
                 use_stateful=False):
        self.embedding_dim, self.units, self.vocab_size = int(embedding_dim), int(units), int(vocab_size)
        self.l2, self.init_method, self.dropout = float(l2_reg), init_method, float(dropout)
        self.show_and_tell = False


class CaptionGenerator(_CaptionGeneratorLSTM):
    GRU = ("gru/kernel", "gru/recurrent_kernel", "gru/bias")

    def __init__(self, encoder, decoder, tokenizer=None, max_length=15, **kw):
        super().__init__(encoder, decoder, tokenizer, max_length, **kw)
        self.sat = False

    # ------------------------------------------------------------------ parameters
    def _create(self, N):
        self.N, self.ldx = int(N), _r4(int(N))
        N, E, U, V, ldV = self.N, self.E, self.U, self.V, self.ldV
        le, ld = self.encoder.l2, self.decoder.l2
        self.keras_shapes = OrderedDict([
            ("fc_embedding/kernel", (N, E)), ("fc_embedding/bias", (E,)), ("embedding/embeddings", (V, E)),
            ("gru/kernel", (E, 3 * U)), ("gru/recurrent_kernel", (U, 3 * U)), ("gru/bias", (2, 3 * U)),
            ("fc1/kernel", (U, U)), ("fc1/bias", (U,)), ("fc_vocab/kernel", (U, V)), ("fc_vocab/bias", (V,))])
        self.layers_spec = OrderedDict([("fc_embedding", ["kernel", "bias"]), ("embedding", ["embeddings"]),
                                        ("gru", ["kernel", "recurrent_kernel", "bias"]), ("fc1", ["kernel", "bias"]),
                                        ("fc_vocab", ["kernel", "bias"])])
        a = self.arena = ParamArena(self.device)
        a.add("fc_embedding/kernel", (N, E), le); a.add("fc_embedding/bias", (E,))
        a.add("embedding/embeddings", (V, E), ld)                 # embeddings_regularizer, att_model.py:70
        a.add("gru/kernel", (E, U, 4), ld); a.add("gru/recurrent_kernel", (U, U, 4), ld); a.add("gru/bias", (2, U, 4))
        a.add("fc1/kernel", (U, U)); a.add("fc1/bias", (U,))
        a.add("fc_vocab/kernel", (U, ldV)); a.add("fc_vocab/bias", (ldV,))
        a.finalize()
        rng = np.random.default_rng(self.seed)
        glorot = lambda shp: rng.uniform(-1, 1, shp) * np.sqrt(6.0 / (shp[0] + shp[1]))
        self.set_weight("fc_embedding/kernel", glorot((N, E)))
        self.set_weight("embedding/embeddings", rng.uniform(-0.05, 0.05, (V, E)))
        self.set_weight("gru/kernel", glorot((E, 3 * U)))
        q = np.concatenate([np.linalg.qr(rng.standard_normal((U, U)))[0] for _ in range(3)], axis=1)   # orthogonal
        self.set_weight("gru/recurrent_kernel", q)
        self.set_weight("fc1/kernel", glorot((U, U)))
        self.set_weight("fc_vocab/kernel", glorot((U, V)))
        if self.optimizer is not None:
            self._init_optimizer_state()

    def set_weight(self, name, arr):
        arr = np.asarray(arr, dtype=np.float32)
        assert tuple(arr.shape) == tuple(self.keras_shapes[name]), (name, arr.shape, self.keras_shapes[name])
        if name in self.GRU:
            dst = self.arena.p(name)
            dst.copy_(torch.from_numpy(np.ascontiguousarray(interleave3(arr, self.U))).view(dst.shape))
            return
        super().set_weight(name, arr)

    def _unpack(self, name, t):
        if name in self.GRU:
            return deinterleave3(t.detach().cpu().numpy())
        return super()._unpack(name, t)

    # ------------------------------------------------------------------ buffers
    def _build(self, B, T):
        if self._shape == (B, T):
            return
        f = self._f
        N, E, U, V, ldV = self.N, self.E, self.U, self.V, self.ldV
        R1 = (T + 1) * B
        self.x = f(B, self.ldx)
        self.cap = torch.zeros(B, T, dtype=torch.int32, device=self.device)
        self.tgt = torch.zeros(R1, dtype=torch.int32, device=self.device)
        self.enc_pre = f(B, E)
        self.Xin, self.XZ = f(R1, E), f(R1, U, 4)
        self.Hs = f(T + 2, B, U)
        self.Cs = self.Hs                                   # (unused; the parent's _stage zeroes Cs[0])
        self.lenmask = torch.ones(B, T + 1, dtype=torch.int32, device=self.device)
        self.gates = f(T + 1, B, U, 4)
        self.Hd = f(R1, U) if self.r_dec > 0 else None
        self.mid, self.mpre = f(R1, U), f(R1, U)
        self.mid_d = f(R1, U) if self.r_dec > 0 else self.mid
        self.logits, self.dlogits = f(R1, ldV), f(R1, ldV)
        self.loss_row = f(R1)
        self.met = f(8)
        self.dmid, self.dOut = f(R1, U), f(R1, U)
        self.dXZ, self.dREC = f(R1, U, 4), f(R1, U, 4)
        self.dh_pass = f(B, U)
        self.dXin = f(R1, E)
        self.dpre = f(B, E)
        self.ew = None
        nch = self.be.bn_nchunk(R1)
        self.work = f(max(E, 4 * U, ldV) * (2 * nch + 1))
        self._alloc_splitk([(B, E, N), (R1, U, V), (R1, E, 4 * U), (R1, U, U), (U, 4 * U, R1)])
        self._shape = (B, T)
        self._graphs = {}
        if self.optimizer is not None and getattr(self, "opt_m", None) is None:
            self._init_optimizer_state()
        self.built = True

    def _stage(self, img, target):
        B, T = super()._stage(img, target)
        if T > 1:        # target[:, i] pairs with predictions[:, i-1] (att_model.py:258-259): shift by one
            tgt = self.cap
            self.tgt[:(T - 1) * B].view(T - 1, B).copy_(tgt[:, 1:].t())
        return B, T

    # ------------------------------------------------------------------ forward / backward
    def _forward(self, B, T, training):
        be, a = self.be, self.arena
        N, E, U, V, ldV = self.N, self.E, self.U, self.V, self.ldV
        R1 = (T + 1) * B
        sd, ds = self.seed, self.drop_step
        self.gemm_sk(self.x, a.p("fc_embedding/kernel"), self.Xin, B, E, N, self.ldx, E, E, bias=a.p("fc_embedding/bias"),
                     pre=self.enc_pre, act=ACT_RELU)                                                   # :49-52
        if training and self.r_enc > 0:
            be.dropout(self.Xin, self.Xin, B, E, E, 0, E, 0, self.r_enc, sd, S_FEAT, 0, ds)
        be.embedding_fwd(a.p("embedding/embeddings"), self.cap, self.Xin[B:], B, T, E, E, V)           # :108-112
        bias = a.p("gru/bias")
        self.gemm_sk(self.Xin, a.p("gru/kernel"), self.XZ, R1, 4 * U, E, E, 4 * U, 4 * U, bias=bias[0])
        Uk = a.p("gru/recurrent_kernel")
        for t in range(T + 1):                                                                          # :118
            be.gru_step_fwd(self.XZ[t * B:(t + 1) * B], self.Hs[t], Uk, bias[1], self.Hs[t + 1], self.gates[t], B, U)
        hs = self.Hs[1:].view(R1, U)
        drop = training and self.r_dec > 0
        if drop:                                                                                        # :121-122
            be.dropout(hs, self.Hd, R1, U, U, B, U, 0, self.r_dec, sd, S_OUT, 0, ds)
            hs = self.Hd
        self._hd_used = hs
        self.gemm_sk(hs, a.p("fc1/kernel"), self.mid, R1, U, U, U, U, U, bias=a.p("fc1/bias"), pre=self.mpre, act=ACT_RELU)
        mid = self.mid
        if drop:
            be.dropout(self.mid, self.mid_d, R1, U, U, B, U, 0, self.r_dec, sd, S_OUT + 1, 0, ds)
            mid = self.mid_d
        self._mid_used = mid
        self.gemm_sk(mid, a.p("fc_vocab/kernel"), self.logits, R1, V, U, U, ldV, ldV, bias=a.p("fc_vocab/bias"),
                     act=ACT_RELU)                                                                      # :127

    def _loss(self, B, T, want_grad, grad_scale):
        """masked sparse CE from logits, target[:, i] <-> predictions[:, i-1], i = 1..T-1, / T (:257-262,306-321)."""
        be = self.be
        n = (T - 1) * B
        if want_grad:
            self.dlogits.zero_()
        if n > 0:
            be.softmax_cce(self.logits, self.tgt, None, self.loss_row, None, self.dlogits if want_grad else None, n,
                           self.V, self.ldV, grad_scale, from_logits=True, mask_zero=True)
        be.sum(self.loss_row, self.met[1:2], n, 1.0 / (B * T))

    def _backward(self, B, T):
        be, a = self.be, self.arena
        N, E, U, V, ldV = self.N, self.E, self.U, self.V, self.ldV
        R1 = (T + 1) * B
        sd, ds = self.seed, self.drop_step
        hd, mid = self._hd_used, self._mid_used
        be.act_bwd(self.logits, self.dlogits, self.dlogits, R1 * ldV, ACT_RELU, 0.0)      # relu'(pre) == (out > 0)
        self.gemm_sk(mid, self.dlogits, a.g("fc_vocab/kernel"), U, V, R1, U, ldV, ldV, transA=True)
        be.colsum(self.dlogits, a.g("fc_vocab/bias"), R1, V, ldV, self.work)
        self.gemm_sk(self.dlogits, a.p("fc_vocab/kernel"), self.dmid, R1, U, V, ldV, ldV, U, transB=True)
        if self.r_dec > 0:
            be.dropout(self.dmid, self.dmid, R1, U, U, B, U, 0, self.r_dec, sd, S_OUT + 1, 0, ds)
        be.act_bwd(self.mpre, self.dmid, self.dmid, R1 * U, ACT_RELU, 0.0)
        self.gemm_sk(hd, self.dmid, a.g("fc1/kernel"), U, U, R1, U, U, U, transA=True)
        be.colsum(self.dmid, a.g("fc1/bias"), R1, U, U, self.work)
        self.gemm_sk(self.dmid, a.p("fc1/kernel"), self.dOut, R1, U, U, U, U, U, transB=True)
        if self.r_dec > 0:
            be.dropout(self.dOut, self.dOut, R1, U, U, B, U, 0, self.r_dec, sd, S_OUT, 0, ds)
        Uk = a.p("gru/recurrent_kernel")
        for t in range(T, -1, -1):
            first = t == T
            be.gru_step_bwd(None if first else self.dREC[(t + 1) * B:(t + 2) * B], Uk, None if first else self.dh_pass,
                            self.dOut[t * B:(t + 1) * B], self.gates[t], self.Hs[t], self.dXZ[t * B:(t + 1) * B],
                            self.dREC[t * B:(t + 1) * B], self.dh_pass, B, U)
        gb = a.g("gru/bias")
        self.gemm_sk(self.Hs, self.dREC, a.g("gru/recurrent_kernel"), U, 4 * U, R1, U, 4 * U, 4 * U, transA=True)
        self.gemm_sk(self.Xin, self.dXZ, a.g("gru/kernel"), E, 4 * U, R1, E, 4 * U, 4 * U, transA=True)
        be.colsum(self.dXZ, gb[0], R1, 4 * U, 4 * U, self.work)
        be.colsum(self.dREC, gb[1], R1, 4 * U, 4 * U, self.work)
        self.gemm_sk(self.dXZ, a.p("gru/kernel"), self.dXin, R1, E, 4 * U, 4 * U, 4 * U, E, transB=True)
        # dense regulariser gradient beside the IndexedSlices -> tape.gradient is dense: ordinary clip norm
        be.embedding_bwd(self.dXin[B:], self.cap, a.g("embedding/embeddings"), None, None, B, T, E, E, V)
        if self.r_enc > 0:
            be.dropout(self.dXin, self.dXin, B, E, E, 0, E, 0, self.r_enc, sd, S_FEAT, 0, ds)
        be.act_bwd(self.enc_pre, self.dXin, self.dpre, B * E, ACT_RELU, 0.0)
        be.colsum(self.dpre, a.g("fc_embedding/bias"), B, E, E, self.work)
        if B <= 64 and E % 16 == 0:
            be.dense_dw_skinny(self.x, self.dpre, a.g("fc_embedding/kernel"), N, E, B, self.ldx)
        else:
            self.gemm_sk(self.x, self.dpre, a.g("fc_embedding/kernel"), N, E, B, self.ldx, E, E, transA=True)

    def _grad_scale(self, B, T):
        return 1.0 / (T * B * self.dp_world)

    def _result(self):
        m = self._met_snapshot()
        return Metrics(scce=m[1], L2=m[2], loss=m[1] + m[2])

    def train_step_SAM(self, data, rho=0.05):
        # att_model.py:153-220 records its second tape BEFORE it perturbs the weights (assign_add comes after the
        # `with` block), so what it applies is the plain gradient of a second forward pass; not rebuilt.
        raise NotImplementedError("att_model.CaptionGenerator.train_step_SAM is not rebuilt; "
                                  "the sharpness-aware step is available on the model.py generator")
