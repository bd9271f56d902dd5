"""ThinkAndTell / ShowAndTell caption generators on the HIP kernels.

Drop-in for ``ThinkAndTell/model.py`` -- ``Encoder(embedding_dim, l2_reg, init_method, dropout)``
(model.py:10-33), ``Decoder(embedding_dim, units, vocab_size, l2_reg, init_method, dropout)``
(42-114), ``CaptionGenerator(encoder, decoder, tokenizer, max_length)`` with ``train_step``
(241-290), ``test_step`` (292-317), ``train_step_SAM`` (166-233) -- and, with
``show_and_tell=True`` descriptors from ``show_and_tell.py``, for ``ShowAndTell/model.py``
(config 1).  Data tuples as in the reference: ``(img_tensor, _, target)`` resp.
``(img_tensor, target)``, target = int ids (B, T).

Launch plan: Dense(tanh|relu) GEMM -> [dropout] -> embedding gather -> one input-projection GEMM
for all T+1 steps -> T+1 fused LSTM steps -> [dropout] -> vocab GEMM(s) (ReLU epilogue) ->
fused sparse-CE-from-logits with the zero-target mask -> mirror-image backward -> clip + Adam/SGD.
The GRU decoder of ThinkAndTell/att_model.py (84-93,118) lives in think_and_tell_att.py.
"""
from collections import OrderedDict

import numpy as np
import torch

from .arena import ParamArena
from .model_base import ModelBase, Metrics, interleave_gates, deinterleave_gates, S_FEAT, S_OUT
from .ops import ACT_RELU, ACT_TANH


def _r4(n):
    return (n + 3) // 4 * 4


class Encoder:
    """ThinkAndTell/model.py:10-33 descriptor (weights live in the CaptionGenerator's arena)."""

    def __init__(self, embedding_dim, l2_reg=0.0, init_method="glorot_uniform", dropout=0.0):
        self.embedding_dim, self.l2, self.init_method, self.dropout = int(embedding_dim), float(l2_reg), init_method, float(dropout)
        self.show_and_tell = False


class Decoder:
    """ThinkAndTell/model.py:42-114 descriptor."""

    def __init__(self, embedding_dim, units, vocab_size, l2_reg=0.0, init_method="glorot_uniform", dropout=0.0,
                 use_stateful=False):
        self.embedding_dim, self.units, self.vocab_size = int(embedding_dim), int(units), int(vocab_size)
        self.l2, self.init_method, self.dropout = float(l2_reg), init_method, float(dropout)
        self.show_and_tell = False


class CaptionGenerator(ModelBase):
    def __init__(self, encoder, decoder, tokenizer=None, max_length=15, **kw):
        super().__init__(**kw)
        self.encoder, self.decoder, self.tokenizer, self.max_length = encoder, decoder, tokenizer, int(max_length)
        self.sat = bool(getattr(decoder, "show_and_tell", False))
        self.E, self.U, self.V = decoder.embedding_dim, decoder.units, decoder.vocab_size
        assert encoder.embedding_dim == self.E, "encoder and decoder embedding_dim differ"
        if self.U % 16:
            raise ValueError("units must be a multiple of 16 (LSTM step kernel tile)")
        self.r_enc = 0.0 if self.sat else encoder.dropout
        self.r_dec = 0.0 if self.sat else decoder.dropout
        self.ldV = _r4(self.V)
        self.N = None
        self.drop_step = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._shape = None

    # ------------------------------------------------------------------ parameters (built at first batch: N unknown before)
    def _create(self, N):
        self.N, self.ldx = int(N), _r4(int(N))
        N, E, U, V, ldV = self.N, self.E, self.U, self.V, self.ldV
        le, ld = (0.0, 0.0) if self.sat else (self.encoder.l2, self.decoder.l2)
        ks = OrderedDict([("fc_embedding/kernel", (N, E)), ("fc_embedding/bias", (E,)),
                          ("embedding/embeddings", (V, E)),
                          ("lstm/kernel", (E, 4 * U)), ("lstm/recurrent_kernel", (U, 4 * U)), ("lstm/bias", (4 * U,))])
        ls = OrderedDict([("fc_embedding", ["kernel", "bias"]), ("embedding", ["embeddings"]),
                          ("lstm", ["kernel", "recurrent_kernel", "bias"])])
        if self.sat:
            ks["fc1/kernel"], ks["fc1/bias"] = (U, U), (U,)
            ls["fc1"] = ["kernel", "bias"]
        ks["fc_vocab/kernel"], ks["fc_vocab/bias"] = (U, V), (V,)
        ls["fc_vocab"] = ["kernel", "bias"]
        self.keras_shapes, self.layers_spec = ks, ls
        a = self.arena = ParamArena(self.device)
        a.add("fc_embedding/kernel", (N, E), le); a.add("fc_embedding/bias", (E,), le)
        a.add("embedding/embeddings", (V, E))
        a.add("lstm/kernel", (E, U, 4), ld); a.add("lstm/recurrent_kernel", (U, U, 4), ld); a.add("lstm/bias", (U, 4))
        if self.sat:
            a.add("fc1/kernel", (U, U)); a.add("fc1/bias", (U,))
        a.add("fc_vocab/kernel", (U, ldV)); a.add("fc_vocab/bias", (ldV,))
        a.finalize()
        rng = np.random.default_rng(self.seed)
        glorot = lambda shp: rng.uniform(-1, 1, shp) * np.sqrt(6.0 / (shp[0] + shp[1]))
        self.set_weight("fc_embedding/kernel", glorot((N, E)))
        self.set_weight("embedding/embeddings", rng.uniform(-0.05, 0.05, (V, E)))
        self.set_weight("lstm/kernel", glorot((E, 4 * U)))
        self.set_weight("lstm/recurrent_kernel", glorot((U, 4 * U)))
        b = np.zeros(4 * U); b[U:2 * U] = 1.0
        self.set_weight("lstm/bias", b)
        if self.sat:
            self.set_weight("fc1/kernel", glorot((U, U)))
        self.set_weight("fc_vocab/kernel", glorot((U, V)))
        if self.optimizer is not None:
            self._init_optimizer_state()

    def set_weight(self, name, arr):
        arr = np.asarray(arr, dtype=np.float32)
        assert tuple(arr.shape) == tuple(self.keras_shapes[name]), (name, arr.shape, self.keras_shapes[name])
        dst = self.arena.p(name)
        if name.startswith("lstm/"):
            arr = interleave_gates(arr, self.U)
        elif name == "fc_vocab/kernel":
            pad = np.zeros((self.U, self.ldV), np.float32); pad[:, :self.V] = arr; arr = pad
        elif name == "fc_vocab/bias":
            pad = np.zeros(self.ldV, np.float32); pad[:self.V] = arr; arr = pad
        dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).view(dst.shape))

    def _unpack(self, name, t):
        arr = t.detach().cpu().numpy()
        if name.startswith("lstm/"):
            return deinterleave_gates(arr)
        if name == "fc_vocab/kernel":
            return np.ascontiguousarray(arr[:, :self.V])
        if name == "fc_vocab/bias":
            return np.ascontiguousarray(arr[:self.V])
        return arr.copy()

    def get_weight(self, name):
        return self._unpack(name, self.arena.p(name))

    def get_gradient(self, name):
        return self._unpack(name, self.arena.g(name))

    def state_tensors(self):
        return []

    # ------------------------------------------------------------------ buffers
    def _build(self, B, T):
        if self._shape == (B, T):
            return
        f = self._f
        N, E, U, V, ldV = self.N, self.E, self.U, self.V, self.ldV
        R1 = (T + 1) * B
        self.x = f(B, self.ldx)
        self.cap = torch.zeros(B, T, dtype=torch.int32, device=self.device)
        self.tgt = torch.zeros(R1, dtype=torch.int32, device=self.device)           # time-major, last block unused
        self.lenmask = torch.ones(B, T + 1, dtype=torch.int32, device=self.device)
        self.enc_pre = f(B, E)
        self.Xin, self.XZ = f(R1, E), f(R1, U, 4)
        self.Hs, self.Cs = f(T + 2, B, U), f(T + 2, B, U)
        self.gates = f(T + 1, B, U, 4)
        self.Out = f(R1, U)
        self._init_seq_lstm(B, U)
        self.Hd = f(R1, U) if self.r_dec > 0 else self.Out
        self.mid = f(R1, U) if self.sat else None
        self.logits, self.dlogits = f(R1, ldV), f(R1, ldV)
        self.loss_row = f(R1)
        self.met = f(8)
        self.dmid = f(R1, U) if self.sat else None
        self.dOut = f(R1, U)
        self.dZ = f(R1, U, 4)
        self.da_pass, self.dc = f(B, U), f(B, U)
        self.dXin = f(R1, E)
        self.dpre = f(B, E)
        self.ew = None
        nch = self.be.bn_nchunk(R1)
        self.work = f(max(E, 4 * U, ldV) * (2 * nch + 1))
        self.rowsq = f(B * T)
        self._alloc_splitk([(B, E, N), (R1, U, V), (R1, E, 4 * U)])
        self.emb_seg = self.arena.entries["embedding/embeddings"].seg
        self._shape = (B, T)
        self._graphs = {}
        if self.optimizer is not None and getattr(self, "opt_m", None) is None:
            self._init_optimizer_state()
        self.built = True

    def _stage(self, img, target):
        tgt = self._to_dev(target, torch.int32)
        B, T = tgt.shape
        xs = self._to_dev(img, torch.float32).reshape(B, -1)
        if self.N is None:
            self._create(xs.shape[1])
        self._build(B, T)
        self.x[:, :self.N].copy_(xs)
        self.cap.copy_(tgt)
        self.tgt[:T * B].view(T, B).copy_(tgt.t())
        if self.sat:      # Embedding mask -> per-sample sequence length over the T+1 LSTM inputs (cuDNN semantics)
            lens = (tgt != 0).sum(dim=1, keepdim=True)
            self.lenmask.copy_((torch.arange(T + 1, device=self.device)[None, :] < lens).to(torch.int32))
        self.Hs[0].zero_(); self.Cs[0].zero_()
        return B, T

    # ------------------------------------------------------------------ forward / backward
    def _forward(self, B, T, training):
        be, a = self.be, self.arena
        N, E, U, V, ldV = self.N, self.E, self.U, self.V, self.ldV
        R1 = (T + 1) * B
        sd, ds = self.seed, self.drop_step
        self.gemm_sk(self.x, a.p("fc_embedding/kernel"), self.Xin, B, E, N, self.ldx, E, E, bias=a.p("fc_embedding/bias"),
                     pre=self.enc_pre, act=ACT_RELU if self.sat else ACT_TANH)
        if training and self.r_enc > 0:
            be.dropout(self.Xin, self.Xin, B, E, E, 0, E, 0, self.r_enc, sd, S_FEAT, 0, ds)
        be.embedding_fwd(a.p("embedding/embeddings"), self.cap, self.Xin[B:], B, T, E, E, V)
        self.gemm_sk(self.Xin, a.p("lstm/kernel"), self.XZ, R1, 4 * U, E, E, 4 * U, 4 * U, bias=a.p("lstm/bias"))
        Ur = a.p("lstm/recurrent_kernel")
        mask = self.lenmask if self.sat else None
        if self._seq_lstm and mask is None:     # unmasked sequence: one persistent launch (tnt_lstm_seq_fwd_f32); the masked
            # ShowAndTell form zeroes masked outputs instead of repeating them and stays on the step kernel
            be.lstm_seq_fwd(self.XZ, self.Hs, self.Cs, Ur, None, None, 0, 0, self.Out, self.gates, T + 1, B, U, self.seq_sync,
                            self._guard_out())
        else:
            for t in range(T + 1):
                be.lstm_step_fwd(self.XZ[t * B:(t + 1) * B], self.Hs[t], self.Cs[t], Ur, None, None, 0, mask, T + 1, t, None,
                                 self.Hs[t + 1], self.Cs[t + 1], self.Out[t * B:(t + 1) * B], self.gates[t], B, U)
        hd = self.Out
        if training and self.r_dec > 0:
            be.dropout(self.Out, self.Hd, R1, U, U, B, U, 0, self.r_dec, sd, S_OUT, 0, ds)
            hd = self.Hd
        self._hd_used = hd
        if self.sat:
            self.gemm_sk(hd, a.p("fc1/kernel"), self.mid, R1, U, U, U, U, U, bias=a.p("fc1/bias"))
            self.gemm_sk(self.mid, a.p("fc_vocab/kernel"), self.logits, R1, V, U, U, ldV, ldV, bias=a.p("fc_vocab/bias"))
        else:
            self.gemm_sk(hd, a.p("fc_vocab/kernel"), self.logits, R1, V, U, U, ldV, ldV, bias=a.p("fc_vocab/bias"),
                    act=ACT_RELU)

    def _loss(self, B, T, want_grad, grad_scale):
        """masked sparse CE from logits, target[:, i] <-> predictions[:, i] (model.py:271-272,319-334)."""
        be = self.be
        i0 = 1 if self.sat else 0
        n = (T - i0) * B
        lg, tg = self.logits[i0 * B:], self.tgt[i0 * B:]
        if want_grad:
            self.dlogits.zero_()
        be.softmax_cce(lg, tg, None, self.loss_row, None, self.dlogits[i0 * B:] if want_grad else None, n, self.V,
                       self.ldV, grad_scale, from_logits=True, mask_zero=True)
        be.sum(self.loss_row, self.met[0:1], n, 1.0 / B)              # sum_i mean_b
        be.sum(self.loss_row, self.met[1:2], n, 1.0 / (B * T))        # ... / T

    def _backward(self, B, T):
        be, a = self.be, self.arena
        N, E, U, V, ldV = self.N, self.E, self.U, self.V, self.ldV
        R1 = (T + 1) * B
        sd, ds = self.seed, self.drop_step
        hd = self._hd_used
        if self.sat:
            self.gemm_sk(self.mid, self.dlogits, a.g("fc_vocab/kernel"), U, V, R1, U, ldV, ldV, transA=True)
            be.colsum(self.dlogits, a.g("fc_vocab/bias"), R1, V, ldV, self.work)
            self.gemm_sk(self.dlogits, a.p("fc_vocab/kernel"), self.dmid, R1, U, V, ldV, ldV, U, transB=True)
            self.gemm_sk(hd, self.dmid, a.g("fc1/kernel"), U, U, R1, U, U, U, transA=True)
            be.colsum(self.dmid, a.g("fc1/bias"), R1, U, U, self.work)
            self.gemm_sk(self.dmid, a.p("fc1/kernel"), self.dOut, R1, U, U, U, U, U, transB=True)
        else:
            # ReLU backward: relu'(pre) == (output > 0), so the stored output serves as "pre"
            be.act_bwd(self.logits, self.dlogits, self.dlogits, R1 * ldV, ACT_RELU, 0.0)
            self.gemm_sk(hd, self.dlogits, a.g("fc_vocab/kernel"), U, V, R1, U, ldV, ldV, transA=True)
            be.colsum(self.dlogits, a.g("fc_vocab/bias"), R1, V, ldV, self.work)
            self.gemm_sk(self.dlogits, a.p("fc_vocab/kernel"), self.dOut, R1, U, V, ldV, ldV, U, transB=True)
        if self.r_dec > 0:
            be.dropout(self.dOut, self.dOut, R1, U, U, B, U, 0, self.r_dec, sd, S_OUT, 0, ds)
        Ur = a.p("lstm/recurrent_kernel")
        mask = self.lenmask if self.sat else None
        seqb = self._seq_lstm and mask is None and self.seq_xch is not None
        if seqb:       # BPTT as one persistent launch (unmasked decoder), see nic.NIC._bwd_seq_lstm
            be.lstm_seq_bwd(Ur, self.dOut, None, 0, 0, self.gates, self.Cs, self.dZ, self.seq_xch, T + 1, B, U, self.seq_sync,
                            self._guard_out())
        else:
            for t in range(T, -1, -1):
                first = t == T
                be.lstm_step_bwd(None if first else self.dZ[(t + 1) * B:(t + 2) * B], Ur, None if first else self.da_pass,
                                 None, None if first else self.dc, None, self.dOut[t * B:(t + 1) * B], mask, T + 1, t,
                                 self.gates[t], self.Cs[t + 1], self.Cs[t], self.dZ[t * B:(t + 1) * B], self.da_pass,
                                 self.dc, None, B, U)
        self.gemm_sk(self.Hs, self.dZ, a.g("lstm/recurrent_kernel"), U, 4 * U, R1, U, 4 * U, 4 * U, transA=True)
        self.gemm_sk(self.Xin, self.dZ, a.g("lstm/kernel"), E, 4 * U, R1, E, 4 * U, 4 * U, transA=True)
        be.colsum(self.dZ, a.g("lstm/bias"), R1, 4 * U, 4 * U, self.work)
        self.gemm_sk(self.dZ, a.p("lstm/kernel"), self.dXin, R1, E, 4 * U, 4 * U, 4 * U, E, transB=True)
        sqo = a.sq_override[self.emb_seg:self.emb_seg + 1]
        be.embedding_bwd(self.dXin[B:], self.cap, a.g("embedding/embeddings"), sqo, self.rowsq, B, T, E, E, V)
        if self.r_enc > 0:
            be.dropout(self.dXin, self.dXin, B, E, E, 0, E, 0, self.r_enc, sd, S_FEAT, 0, ds)
        be.act_bwd(self.enc_pre, self.dXin, self.dpre, B * E, ACT_RELU if self.sat else ACT_TANH, 0.0)
        be.colsum(self.dpre, a.g("fc_embedding/bias"), B, E, E, self.work)
        if B <= 64 and E % 16 == 0:
            be.dense_dw_skinny(self.x, self.dpre, a.g("fc_embedding/kernel"), N, E, B, self.ldx)
        else:
            self.gemm_sk(self.x, self.dpre, a.g("fc_embedding/kernel"), N, E, B, self.ldx, E, E, transA=True)

    # ------------------------------------------------------------------ steps
    def _unpack_batch(self, data):
        if len(data) == 3:
            return data[0], data[2]
        return data[0], data[1]

    def _grad_scale(self, B, T):
        # ThinkAndTell differentiates scce = sum/T (model.py:274-282); ShowAndTell differentiates the
        # un-normalised sum (ShowAndTell/model.py:154-161)
        return (1.0 if self.sat else 1.0 / T) / (B * self.dp_world)

    def _result(self):
        m = self._met_snapshot()
        if self.sat:
            return Metrics({"loss": m[0], "norm loss": m[1]})
        out = Metrics(scce=m[1], L2=m[2], loss=m[1] + m[2])
        return out.guarded(self, m[self.GUARD]) if self._seq_lstm else out

    def train_step(self, data):
        if self.optimizer is None:
            raise RuntimeError("compile() the model before train_step")
        img, target = self._unpack_batch(data)
        B, T = self._stage(img, target)
        self._sync_lr()

        def fb():
            self._forward(B, T, True)
            self._loss(B, T, True, self._grad_scale(B, T))
            self._backward(B, T)

        def up():
            self._norms_and_l2(self.met[2:3])
            self._apply_optimizer()
        if self.grad_sync is None:
            self._run_captured(("train", B, T), lambda: (fb(), up()))
        else:
            self._run_captured(("train_fb", B, T), fb)
            self.grad_sync(self)
            self._run_captured(("train_up", B, T), up)
        self.optimizer.iterations += 1
        return self._result()

    def train_step_SAM(self, data, rho=0.05):
        """Sharpness-aware step (ThinkAndTell/model.py:166-233): ascent by rho*g/||g||, second gradient
        at the perturbed weights, restore, apply.  Both passes use the same dropout masks."""
        if self.optimizer is None:
            raise RuntimeError("compile() the model before train_step_SAM")
        img, target = self._unpack_batch(data)
        B, T = self._stage(img, target)
        self._sync_lr()
        be, a, sp = self.be, self.arena, self.arena.spans
        if self.ew is None:
            self.ew = torch.zeros_like(a.theta)

        def run():
            gs = self._grad_scale(B, T)
            self._forward(B, T, True); self._loss(B, T, True, gs); self._backward(B, T)
            self._norms_and_l2(None)
            be.sam(a.theta, a.grad, self.ew, sp.span_seg, sp.span_off, sp.span_len, a.seg_l2, a.sq, a.nseg, sp.nspan,
                   rho, 0)
            self._forward(B, T, True); self._loss(B, T, True, gs); self._backward(B, T)
            self._norms_and_l2(self.met[2:3])       # L2 metric at the perturbed weights, as the reference reports it
            be.sam(a.theta, a.grad, self.ew, sp.span_seg, sp.span_off, sp.span_len, a.seg_l2, a.sq, a.nseg, sp.nspan,
                   rho, 1)
            self._norms_and_l2(None)                # clip norms of the SAM gradient at the restored weights
            self._apply_optimizer()
        self._run_captured(("sam", B, T), run)
        self.optimizer.iterations += 1
        return self._result()

    def test_step(self, data):
        img, target = self._unpack_batch(data)
        B, T = self._stage(img, target)

        def run():
            self._forward(B, T, False)
            self._loss(B, T, False, 0.0)
            self._norms_and_l2(self.met[2:3])
        self._run_captured(("test", B, T), run)
        return self._result()

    def __call__(self, data, training=False):
        """decoder((target, features)) after the encoder: returns the (B, T+1, V) "logits" (model.py:84-114)."""
        img, target = self._unpack_batch(data) if isinstance(data, (tuple, list)) and len(data) >= 2 else data
        B, T = self._stage(img, target)

        def run():
            self._forward(B, T, training)
            return self.logits.view(T + 1, B, self.ldV)[:, :, :self.V].permute(1, 0, 2).contiguous()
        return self._guarded(run)
