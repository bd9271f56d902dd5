"""NIC -- dense voxel encoder + LSTM decoder (BASELINE config 2).

Drop-in for ``AttemptFour/Model/NIC.py`` (class NIC, lines 19-325): same constructor
arguments (NIC.py:22), ``call`` (100-145), ``greedy_predict`` (148-195), ``train_step``
(198-252), ``test_step`` (254-299).  The step is a fixed sequence of HIP kernel launches
over static buffers (captured into a hipGraph after warm-up):

  forward : [dropout] -> split-K GEMM (B x N x E, bias+LeakyReLU) -> [dropout] -> BatchNorm
            -> embedding gather -> one GEMM for every timestep's input projection
            -> T+1 fused LSTM step kernels -> vocab GEMM -> fused softmax/CE/accuracy/dlogits
  backward: vocab dW/dX GEMMs -> T+1 fused LSTM backward steps -> dU, dW, dX GEMMs
            -> embedding scatter -> BatchNorm/LeakyReLU backward -> encoder dW GEMM
  update  : [all-reduce] -> per-variable norms -> clip + Adam over the flat arena.
"""
from collections import OrderedDict

import numpy as np
import torch

from .arena import ParamArena
from .model_base import (ModelBase, Metrics, interleave_gates, deinterleave_gates, S_IN, S_FEAT, S_LSTM_IN, BN_EPS,
                         BN_MOMENTUM)
from .ops import ACT_LEAKY

ENC_SPLITS = 16      # K splits of the streaming encoder forward: 16 column groups x 16 splits = one workgroup per CU


def _r4(n):
    return (n + 3) // 4 * 4


class NIC(ModelBase):
    def __init__(self, input_size, units, embedding_dim, vocab_size, max_length, dropout_input, dropout,
                 dropout_lstm, input_reg, lstm_reg, output_reg, norm="batch", **kw):
        super().__init__(**kw)
        self.N, self.U, self.E, self.V, self.max_length = int(input_size), int(units), int(embedding_dim), int(vocab_size), int(max_length)
        self.r_in, self.r_feat, self.r_lstm = float(dropout_input), float(dropout), float(dropout_lstm)
        # NIC.py:53-55: the output layer is regularised with lstm_reg (quirk kept)
        self.l2_in, self.l2_lstm, self.l2_out = float(input_reg), float(lstm_reg), float(lstm_reg)
        assert norm in ("batch", "layer")
        self.norm = norm
        if self.U % 16:
            raise ValueError("units must be a multiple of 16 (LSTM step kernel tile)")
        N, U, E, V = self.N, self.U, self.E, self.V
        self.ldx, self.ldV = _r4(N), _r4(V)
        self.layers_spec = OrderedDict([
            ("dense_img", ["kernel", "bias"]),
            ("batch_norm", ["gamma", "beta", "moving_mean", "moving_variance"]),
            ("emb_text", ["embeddings"]),
            ("lstm", ["kernel", "recurrent_kernel", "bias"]),
            ("time_distributed_softmax", ["kernel", "bias"])])
        self.keras_shapes = OrderedDict([
            ("dense_img/kernel", (N, E)), ("dense_img/bias", (E,)),
            ("batch_norm/gamma", (E,)), ("batch_norm/beta", (E,)),
            ("batch_norm/moving_mean", (E,)), ("batch_norm/moving_variance", (E,)),
            ("emb_text/embeddings", (V, E)),
            ("lstm/kernel", (E, 4 * U)), ("lstm/recurrent_kernel", (U, 4 * U)), ("lstm/bias", (4 * U,)),
            ("time_distributed_softmax/kernel", (U, V)), ("time_distributed_softmax/bias", (V,))])
        a = self.arena = ParamArena(self.device)
        a.add("dense_img/kernel", (N, E), self.l2_in)
        a.add("dense_img/bias", (E,))
        a.add("batch_norm/gamma", (E,))
        a.add("batch_norm/beta", (E,))
        a.add("emb_text/embeddings", (V, E))
        a.add("lstm/kernel", (E, U, 4), self.l2_lstm)
        a.add("lstm/recurrent_kernel", (U, U, 4))
        a.add("lstm/bias", (U, 4))
        a.add("time_distributed_softmax/kernel", (U, self.ldV), self.l2_out)
        a.add("time_distributed_softmax/bias", (self.ldV,))
        a.finalize()
        self.mov_mean, self.mov_var = self._f(E), torch.ones(E, dtype=torch.float32, device=self.device)
        self.drop_step = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._init_weights(np.random.default_rng(self.seed))
        self._shape = None

    # ------------------------------------------------------------------ weights
    def _init_weights(self, rng):
        """Initialisers of NIC.py:64-98 (GlorotNormal kernels, keras defaults elsewhere); values only
        matter for benchmarks -- parity tests inject weights."""
        N, U, E, V = self.N, self.U, self.E, self.V
        tn = lambda shape, std: np.clip(rng.standard_normal(shape), -2, 2) * std / 0.8796
        self.set_weight("dense_img/kernel", tn((N, E), np.sqrt(2.0 / (N + E))))
        self.set_weight("emb_text/embeddings", rng.uniform(-0.05, 0.05, (V, E)))
        lim = np.sqrt(6.0 / (E + 4 * U))
        self.set_weight("lstm/kernel", rng.uniform(-lim, lim, (E, 4 * U)))
        q = np.concatenate([np.linalg.qr(rng.standard_normal((U, U)))[0] for _ in range(4)], axis=1)
        self.set_weight("lstm/recurrent_kernel", q)
        b = np.zeros(4 * U); b[U:2 * U] = 1.0          # unit_forget_bias
        self.set_weight("lstm/bias", b)
        self.set_weight("time_distributed_softmax/kernel", tn((U, V), np.sqrt(2.0 / (U + V))))
        self.set_weight("batch_norm/gamma", np.ones(E))

    def set_weight(self, name, arr):
        arr = np.asarray(arr, dtype=np.float32)
        assert tuple(arr.shape) == tuple(self.keras_shapes[name]), (name, arr.shape, self.keras_shapes[name])
        if name == "batch_norm/moving_mean":
            self.mov_mean.copy_(torch.from_numpy(arr)); return
        if name == "batch_norm/moving_variance":
            self.mov_var.copy_(torch.from_numpy(arr)); return
        dst = self.arena.p(name)
        if name.startswith("lstm/"):
            arr = interleave_gates(arr, self.U)
        elif name == "time_distributed_softmax/kernel":
            pad = np.zeros((self.U, self.ldV), np.float32); pad[:, :self.V] = arr; arr = pad
        elif name == "time_distributed_softmax/bias":
            pad = np.zeros(self.ldV, np.float32); pad[:self.V] = arr; arr = pad
        dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).view(dst.shape))

    def _unpack(self, name, t):
        arr = t.detach().cpu().numpy()
        if name.startswith("lstm/"):
            return deinterleave_gates(arr)
        if name == "time_distributed_softmax/kernel":
            return np.ascontiguousarray(arr[:, :self.V])
        if name == "time_distributed_softmax/bias":
            return np.ascontiguousarray(arr[:self.V])
        return arr.copy()

    def get_weight(self, name):
        if name == "batch_norm/moving_mean":
            return self.mov_mean.cpu().numpy().copy()
        if name == "batch_norm/moving_variance":
            return self.mov_var.cpu().numpy().copy()
        return self._unpack(name, self.arena.p(name))

    def get_gradient(self, name):
        """Last computed gradient of a trainable (keras layout, *without* the L2 term, which the
        optimizer kernels add on the fly)."""
        st = self.__dict__.get("_enc_grad_stale")
        if name == "dense_img/kernel" and st is not None:
            # the fused step consumed X^T dpre inside the optimizer launches without writing it: its operands are still
            # in place, materialise it on request
            x, dpre, rows = st
            self.be.dense_dw_skinny(x, dpre, self.arena.g(name), self.N, self.E, rows, self.ldx)
            self._enc_grad_stale = None
        return self._unpack(name, self.arena.g(name))

    def _enc_update_fused(self, rows):
        """True when the encoder kernel's gradient is consumed inside the optimizer launches instead of being written
        out (tnt_dense_dw_sqnorm_f32 / tnt_dense_dw_adam_f32): the single-process fused step with Adam, no AGC."""
        opt = self.optimizer
        return bool(self.__dict__.get("_defer_sum2") and self.dp_world == 1 and getattr(self, "fuse_enc_update", True)
                    and opt is not None and opt.kind == "adam" and not self.__dict__.get("agc") and rows <= 64
                    and self.E % 512 == 0 and hasattr(self.be, "dense_dw_adam") and hasattr(self.be, "step_finalize")
                    and self.arena.entries["dense_img/kernel"].seg == 0
                    # one norm-partial slot per workgroup inside the variable's own span slots
                    and min((self.N + 15) // 16, 256) * (self.E // 512) <= self.arena.spans.first_host[1])

    def state_tensors(self):
        """Non-trainable device state (BatchNorm moving statistics)."""
        return [self.mov_mean, self.mov_var]

    @property
    def losses(self):
        """[lambda*||W||^2 ...] as self.losses (NIC.py:242-243)."""
        a = self.arena
        self._norms_and_l2(self.met[2:3])
        return [a.seg_l2[e.seg] * a.wsq[e.seg] for e in a.entries.values() if e.l2 > 0]

    # ------------------------------------------------------------------ buffers
    def _build(self, B, T):
        if self._shape == (B, T):
            return
        f = self._f
        N, U, E, V, ldV = self.N, self.U, self.E, self.V, self.ldV
        R1 = (T + 1) * B
        self.x = f(B, self.ldx)
        self.xd = f(B, self.ldx) if self.r_in > 0 else self.x
        self.cap = torch.zeros(B, T, dtype=torch.int32, device=self.device)
        self.tgt = torch.zeros(T * B, dtype=torch.int32, device=self.device)
        self.enc_pre, self.enc_y = f(B, E), f(B, E)
        # K-split partials of the streaming encoder forward (tnt_dense_fwd_stream_f32); None -> generic split-K GEMM
        ok = self.norm == "batch" and B <= 256 and E % 32 == 0 and N % 4 == 0 and hasattr(self.be, "dense_fwd_stream")
        self.enc_part = f(ENC_SPLITS * B * E) if ok and getattr(self, "stream_encoder", True) else None
        self.enc_gx = self.enc_w2 = None        # Gram by-products of the forward (allocated on first use, outside captures)
        self.enc_yd = f(B, E) if self.r_feat > 0 else self.enc_y
        self.xhat = f(B, E)
        self.inv_std = f(max(B, E))
        self.Xin = f(R1, E)
        self.Xin_d = f(R1, E) if self.r_lstm > 0 else self.Xin
        self.XZ = f(R1, U, 4)
        self.Hs, self.Cs = f(T + 2, B, U), f(T + 2, B, U)
        self.gates = f(T + 1, B, U, 4)
        self.Out = f(T, B, U)
        self._init_seq_lstm(B, U)        # persistent sequence kernel: opt-in per shape and per device
        self.logits = f(T * B, ldV)
        self.loss_row, self.corr_row = f(T * B), f(T * B)
        self.met = f(8)
        self.dOut = f(T * B, U)
        self.dZ = f(R1, U, 4)
        self.da_pass, self.dc, self.dout = f(B, U), f(B, U), f(B, U)
        self.dXin = f(R1, E)
        self.dyd, self.dpre = f(B, E), f(B, E)
        be = self.be
        self._alloc_splitk([(B, E, N), (T * B, U, V), (R1, E, 4 * U), (R1, 4 * U, E), (T * B, V, U)])
        nch = max(be.bn_nchunk(B), be.bn_nchunk(R1), be.bn_nchunk(T * B))
        self.work = f(max(E, 4 * U, ldV) * (2 * nch + 1))
        self.work2 = f(max(E, 4 * U, ldV) * (2 * nch + 1))
        self.rowsq = f(B * T)
        self.emb_seg = self.arena.entries["emb_text/embeddings"].seg
        self._shape = (B, T)
        self._graphs = {}
        if self.optimizer is not None and getattr(self, "opt_m", None) is None:
            self._init_optimizer_state()
        self.built = True

    def _stage_inputs(self, data):
        x, cap, a0, c0 = data
        cap_t = self._to_dev(cap, torch.int32)
        B, T = cap_t.shape
        self._build(B, T)
        xs = self._to_dev(x, torch.float32)
        assert xs.shape == (B, self.N), f"betas shape {tuple(xs.shape)} != {(B, self.N)}"
        self.x[:, :self.N].copy_(xs)
        self.cap.copy_(cap_t)
        self.Hs[0].copy_(self._to_dev(a0, torch.float32))
        self.Cs[0].copy_(self._to_dev(c0, torch.float32))
        return B, T

    def _fused_tail(self, B):
        """one-launch encoder tail (tnt_enc_tail_*): BatchNorm encoder, batch <= 256 rows, E % 4 == 0"""
        return (self.norm == "batch" and B <= 256 and self.E % 4 == 0 and getattr(self, "fuse_tail", True)
                and not self._sync_bn_on())        # synchronised BatchNorm: the statistics leave the kernel for a collective

    # ------------------------------------------------------------------ forward
    def _forward(self, B, T, training):
        be, a = self.be, self.arena
        N, U, E, V, ldV = self.N, self.U, self.E, self.V, self.ldV
        R1 = (T + 1) * B
        sd, ds = self.seed, self.drop_step
        x = self.x
        if training and self.r_in > 0:                                              # NIC.py:122
            be.dropout(self.x, self.xd, B, N, self.ldx, 0, N, 0, self.r_in, sd, S_IN, 0, ds)
            x = self.xd
        drop_l = training and self.r_lstm > 0
        xin = self.Xin_d if drop_l else self.Xin
        fused = self._fused_tail(B)
        stream = fused and self.enc_part is not None
        emb_ride = False
        if stream:      # :125-128 + the feature step's LSTM input dropout: the streaming product's K-split partials are
            #             summed (+ bias, LeakyReLU) by the tail kernel, which holds whole columns for the batch statistics
            self._enc_gram = None
            if (training and self._enc_update_fused(B) and getattr(self, "gram_norm", True) and N % 16 == 0 and E >= 512
                    and hasattr(be, "dense_gram_norm")):
                # the optimizer step will take this kernel's gradient X^T dpre without writing it: leave X X^T and
                # sum W^2 behind, from which (with dpre) its clip-by-norm factor follows (tnt_dense_gram_norm_f32)
                if self.enc_gx is None:
                    self.enc_gx, self.enc_w2 = self._f(ENC_SPLITS * 64 * 64), self._f(ENC_SPLITS * (E // 32))
                be.dense_fwd_stream_gram(x, a.p("dense_img/kernel"), self.enc_part, self.enc_gx, self.enc_w2, B, E, N,
                                         self.ldx, E, ENC_SPLITS)
                self._enc_gram = (self.enc_pre, a.p("dense_img/bias"), self.enc_gx, ENC_SPLITS, self.enc_w2,
                                  ENC_SPLITS * (E // 32))
            else:
                be.dense_fwd_stream(x, a.p("dense_img/kernel"), self.enc_part, B, E, N, self.ldx, E, ENC_SPLITS)
            emb_ride = E % 4 == 0 and hasattr(be, "enc_tail_fwd_sk_emb") and getattr(self, "emb_ride", True)
            targs = (self.enc_part, ENC_SPLITS, a.p("dense_img/bias"), self.enc_pre, 0.2, a.p("batch_norm/gamma"),
                     a.p("batch_norm/beta"), self.mov_mean, self.mov_var, xin, self.xhat, self.inv_std, B, E, E, training,
                     BN_EPS, BN_MOMENTUM, self.r_feat if training else 0.0, self.r_lstm if training else 0.0, sd, S_FEAT,
                     S_LSTM_IN + 0, ds)
            if emb_ride:    # ... with the Embedding gather (+ the text call's input dropout) of the rows behind the features
                be.enc_tail_fwd_sk_emb(*targs, a.p("emb_text/embeddings"), self.cap, xin[B:], B, T, V,
                                       self.r_lstm if drop_l else 0.0, S_LSTM_IN + 1)
            else:
                be.enc_tail_fwd_sk(*targs)
        else:
            self.gemm_sk(x, a.p("dense_img/kernel"), self.enc_y, B, E, N, self.ldx, E, E, bias=a.p("dense_img/bias"),
                         pre=self.enc_pre, act=ACT_LEAKY, slope=0.2)                          # :125
        y = self.enc_y
        if stream:
            pass
        elif fused:     # :126-128 + the feature step's LSTM input dropout in one launch
            be.enc_tail_fwd(y, a.p("batch_norm/gamma"), a.p("batch_norm/beta"), self.mov_mean, self.mov_var, xin,
                            self.xhat, self.inv_std, B, E, E, training, BN_EPS, BN_MOMENTUM,
                            self.r_feat if training else 0.0, self.r_lstm if training else 0.0, sd, S_FEAT,
                            S_LSTM_IN + 0, ds)
        else:
            if training and self.r_feat > 0:                                            # :126
                be.dropout(self.enc_y, self.enc_yd, B, E, E, 0, E, 0, self.r_feat, sd, S_FEAT, 0, ds)
                y = self.enc_yd
            if self.norm == "batch":                                                    # :127-128
                self._bn_fwd(y, a.p("batch_norm/gamma"), a.p("batch_norm/beta"), self.mov_mean, self.mov_var,
                             self.Xin, self.xhat, self.inv_std, B, E, E, training, self.work)
            else:
                be.layernorm_fwd(y, a.p("batch_norm/gamma"), a.p("batch_norm/beta"), self.Xin, self.xhat,
                                 self.inv_std, B, E, E, BN_EPS)
        if drop_l and not fused:
            be.dropout(self.Xin, self.Xin_d, B, E, E, 0, E, 0, self.r_lstm, sd, S_LSTM_IN + 0, 0, ds)
        if stream and emb_ride:
            pass                        # gathered by the tail launch above
        elif drop_l and E % 4 == 0:     # :131 + the text call's LSTM(dropout=...) input mask in one launch
            be.embedding_fwd_drop(a.p("emb_text/embeddings"), self.cap, None, self.Xin_d[B:], B, T, E, E, V, self.r_lstm,
                                  sd, S_LSTM_IN + 1, 0, ds)
        else:
            be.embedding_fwd(a.p("emb_text/embeddings"), self.cap, self.Xin[B:], B, T, E, E, V)   # :131
            if drop_l:       # LSTM(dropout=...) masks the layer input, one mask per call
                be.dropout(self.Xin[B:], self.Xin_d[B:], T * B, E, E, B, E, 0, self.r_lstm, sd, S_LSTM_IN + 1, 0, ds)
        self._xin_used = xin
        # input projection of all T+1 steps as ONE epilogue-free GEMM; the LSTM bias is added inside the step kernel
        if getattr(self, "fused_xproj", False) and hasattr(be, "gemm_fused") and be.gemm_fused_cfg(R1, 4 * U, E) > 0:
            be.gemm_fused(xin, a.p("lstm/kernel"), self.XZ, R1, 4 * U, E, E, 4 * U, 4 * U)
        else:
            self.gemm_sk(xin, a.p("lstm/kernel"), self.XZ, R1, 4 * U, E, E, 4 * U, 4 * U)
        Ur, bl = a.p("lstm/recurrent_kernel"), a.p("lstm/bias")
        if self._seq_lstm:
            # both LSTM calls (NIC.py:138,140) as ONE persistent launch: the T+1 dependent steps pay an XCD-local barrier
            # each instead of a kernel launch, the recurrent weights stay in VGPRs (tnt_lstm_seq_fwd_f32)
            be.lstm_seq_fwd(self.XZ, self.Hs, self.Cs, Ur, bl, self.cap, T, 1, self.Out, self.gates, T + 1, B, U,
                            self.seq_sync, self._guard_out())
        else:
            # lstm call 1: the feature, one unmasked step (NIC.py:138)
            be.lstm_step_fwd(self.XZ[:B], self.Hs[0], self.Cs[0], Ur, None, None, 0, None, 0, 0, None, self.Hs[1],
                             self.Cs[1], None, self.gates[0], B, U, xz_bias=bl)
            # lstm call 2: the text, masked by the Embedding mask (NIC.py:140)
            for t in range(1, T + 1):
                be.lstm_step_fwd(self.XZ[t * B:(t + 1) * B], self.Hs[t], self.Cs[t], Ur, None, None, 0, self.cap, T,
                                 t - 1, self.Out[t - 2] if t > 1 else None, self.Hs[t + 1], self.Cs[t + 1],
                                 self.Out[t - 1], self.gates[t], B, U, xz_bias=bl)
        self.gemm_sk(self.Out, a.p("time_distributed_softmax/kernel"), self.logits, T * B, V, U, U, ldV, ldV,
                bias=a.p("time_distributed_softmax/bias"))                         # NIC.py:143

    def _loss_metrics(self, B, T, want_grad):
        """softmax + per-timestep mean CE / accuracy summed over T and divided by T
        (NIC.py:233-240) == sum over all (b,t) / (B*T)."""
        be = self.be
        n = T * B
        if want_grad:
            be.softmax_cce(self.logits, self.tgt, None, self.loss_row, self.corr_row, self.logits, n, self.V,
                           self.ldV, 1.0 / (n * self.dp_world))
        else:
            be.softmax_cce(self.logits, self.tgt, self.logits, self.loss_row, self.corr_row, None, n, self.V,
                           self.ldV, 0.0)
        self._sum2(self.loss_row, self.met[0:1], self.corr_row, self.met[1:2], n, 1.0 / n)

    # ------------------------------------------------------------------ backward
    def _backward(self, B, T):
        self._bwd_head(B, T, join=False)       # dW of the head keeps running beside the BPTT chain
        self._bwd_seq(B, T, join=False)
        self._bwd_enc(B, T)
        self.join()

    def _bwd_head(self, B, T, join=True):
        """vocabulary head: dW, db, dX (gradients ready first -> first all-reduce bucket under DP)."""
        be, a = self.be, self.arena
        U, V, ldV = self.U, self.V, self.ldV
        dlog = self.logits
        Wo = a.p("time_distributed_softmax/kernel")
        dw = dict(A=self.Out, B=dlog, C=a.g("time_distributed_softmax/kernel"), M=U, N=V, K=T * B, lda=U, ldb=ldV, ldc=ldV,
                  transA=True, colsum=a.g("time_distributed_softmax/bias"))
        dx = dict(A=dlog, B=Wo, C=self.dOut, M=T * B, N=U, K=V, lda=ldV, ldb=ldV, ldc=U, transB=True)
        if getattr(self, "g3_riders", True) and self.gemm3_pair(dw, dx):
            # kernel gradient (with the bias gradient as a rider on the dlogits tiles it streams anyway) and input gradient,
            # the two independent readers of dlogits, in ONE launch
            if join:
                self.join()
            return
        if getattr(self, "g3_riders", True) and self.gemm3(self.Out, dlog, a.g("time_distributed_softmax/kernel"), U, V, T * B,
                                                        U, ldV, ldV, transA=True, colsum=a.g("time_distributed_softmax/bias")):
            pass        # dW and the bias gradient: one launch
        elif getattr(self, "fused_head_grads", False) and hasattr(be, "gemm_fused") and \
                be.gemm_fused_cfg(U, V, T * B, True, False, 1) > 0:
            be.gemm_fused(self.Out, dlog, a.g("time_distributed_softmax/kernel"), U, V, T * B, U, ldV, ldV, transA=True,
                          colsum=a.g("time_distributed_softmax/bias"))
        else:
            with self.side(0 if getattr(self, "side_head", True) else -1):
                self.gemm_sk(self.Out, dlog, a.g("time_distributed_softmax/kernel"), U, V, T * B, U, ldV, ldV, transA=True,
                             ws=1)
                if self.__dict__.get("_defer_sum2") and hasattr(be, "colsum2") and T * B <= 2048:
                    # single-process step: the head-bias column sums share a launch with the LSTM-bias ones behind the
                    # BPTT chain (dlogits stays in place until the next forward)
                    self._colsum_deferred = (dlog, a.g("time_distributed_softmax/bias"), T * B, V, ldV)
                else:
                    be.colsum(dlog, a.g("time_distributed_softmax/bias"), T * B, V, ldV, self.work2)
        self.gemm_sk(dlog, Wo, self.dOut, T * B, U, V, ldV, ldV, U, transB=True)
        if join:
            self.join()

    def _bwd_seq(self, B, T, join=True):
        """BPTT, LSTM / embedding / BatchNorm gradients, down to dpre of the encoder Dense."""
        self._bwd_seq_lstm(B, T)
        self._bwd_seq_front(B, T)
        if join:
            self.join()

    def _bwd_seq_lstm(self, B, T):
        """BPTT over the T+1 LSTM steps and the three LSTM parameter gradients."""
        be, a = self.be, self.arena
        N, U, E, V, ldV = self.N, self.U, self.E, self.V, self.ldV
        R1 = (T + 1) * B
        sd, ds = self.seed, self.drop_step
        Ur = a.p("lstm/recurrent_kernel")
        dOut = self.dOut.view(T, B, U)
        if self._seq_lstm and self.seq_xch is not None:
            # the T+1 dependent backward steps as ONE persistent launch (tnt_lstm_seq_bwd_f32): weights stationary, the
            # recurrent product pushed as partial tiles through the XCD's L2
            be.lstm_seq_bwd(Ur, dOut, self.cap, T, 1, self.gates, self.Cs, self.dZ, self.seq_xch, T + 1, B, U, self.seq_sync,
                            self._guard_out())
        else:
            for t in range(T, 0, -1):
                first = t == T
                be.lstm_step_bwd(None if first else self.dZ[(t + 1) * B:(t + 2) * B], Ur,
                                 None if first else self.da_pass, None, None if first else self.dc,
                                 None if first else self.dout, dOut[t - 1], self.cap, T, t - 1, self.gates[t],
                                 self.Cs[t + 1], self.Cs[t], self.dZ[t * B:(t + 1) * B], self.da_pass, self.dc, self.dout,
                                 B, U)
            be.lstm_step_bwd(self.dZ[B:2 * B], Ur, self.da_pass, None, self.dc, None, None, None, 0, 0, self.gates[0],
                             self.Cs[1], self.Cs[0], self.dZ[:B], None, None, None, B, U)
        xin = self._xin_used
        self._dxin_done = False
        if getattr(self, "g3_riders", True) and E == U:
            dw = dict(A=xin, B=self.dZ, C=a.g("lstm/kernel"), M=E, N=4 * U, K=R1, lda=E, ldb=4 * U, ldc=4 * U, transA=True,
                      colsum=a.g("lstm/bias"), A2=self.Hs, C2=a.g("lstm/recurrent_kernel"))
            dx = dict(A=self.dZ, B=a.p("lstm/kernel"), C=self.dXin, M=R1, N=E, K=4 * U, lda=4 * U, ldb=4 * U, ldc=E, transB=True)
            if self.gemm3_pair(dw, dx):
                # the LSTM's three parameter gradients AND its input gradient (the four readers of dZ) in ONE launch;
                # _bwd_seq_front finds dXin done
                self._dxin_done = True
                d = self.__dict__.pop("_colsum_deferred", None)
                if d is not None:
                    be.colsum(*d, self.work2)
                return
        if getattr(self, "g3_riders", True) and E == U and self.gemm3(
                xin, self.dZ, a.g("lstm/kernel"), E, 4 * U, R1, E, 4 * U, 4 * U, transA=True, colsum=a.g("lstm/bias"),
                A2=self.Hs, C2=a.g("lstm/recurrent_kernel")):
            # kernel, recurrent-kernel and bias gradients in ONE launch: the two products share dZ, the column sums of dZ
            # ride on the fragments the first tile row holds anyway
            d = self.__dict__.pop("_colsum_deferred", None)
            if d is not None:
                be.colsum(*d, self.work2)
            return
        if getattr(self, "fused_lstm_grads", False) and E == U and hasattr(be, "gemm_fused") and \
                be.gemm_fused_cfg(U, 4 * U, R1, True, False, 2) > 0:
            # kernel, recurrent-kernel and bias gradients in ONE launch of the one-round GEMM family: the two products
            # share dZ, the bias gradient (column sums of dZ) rides on the tiles the first row of workgroups loads anyway
            be.gemm_fused(xin, self.dZ, a.g("lstm/kernel"), E, 4 * U, R1, E, 4 * U, 4 * U, transA=True,
                          colsum=a.g("lstm/bias"), A2=self.Hs, C2=a.g("lstm/recurrent_kernel"))
            d = self.__dict__.pop("_colsum_deferred", None)
            if d is not None:
                be.colsum(*d, self.work2)
            return
        with self.side(1):
            self.gemm_sk(self.Hs, self.dZ, a.g("lstm/recurrent_kernel"), U, 4 * U, R1, U, 4 * U, 4 * U, transA=True, ws=2)
        with self.side(0):
            self.gemm_sk(xin, self.dZ, a.g("lstm/kernel"), E, 4 * U, R1, E, 4 * U, 4 * U, transA=True, ws=1)
            d = self.__dict__.pop("_colsum_deferred", None)
            if d is not None and R1 <= 2048:
                be.colsum2(*d, self.dZ, a.g("lstm/bias"), R1, 4 * U, 4 * U)
            else:
                if d is not None:
                    be.colsum(*d, self.work2)
                be.colsum(self.dZ, a.g("lstm/bias"), R1, 4 * U, 4 * U, self.work2)

    def _bwd_seq_front(self, B, T):
        """LSTM input gradient -> embedding rows, BatchNorm, encoder activation, encoder bias."""
        be, a = self.be, self.arena
        N, U, E, V, ldV = self.N, self.U, self.E, self.V, self.ldV
        R1 = (T + 1) * B
        sd, ds = self.seed, self.drop_step
        if not self.__dict__.pop("_dxin_done", False):
            self.gemm_sk(self.dZ, a.p("lstm/kernel"), self.dXin, R1, E, 4 * U, 4 * U, 4 * U, E, transB=True)
        fused = self._fused_tail(B)
        if self.r_lstm > 0:
            if not fused:
                be.dropout(self.dXin, self.dXin, B, E, E, 0, E, 0, self.r_lstm, sd, S_LSTM_IN + 0, 0, ds)
        # The text call's LSTM-input dropout' can ride on the sparse Embedding backward (mask applied to the rows as they are
        # read; not when AGC needs the dropped-out rows themselves).  Off by default: measured 0.5761 -> 0.5845 ms/step --
        # the Philox calls land on the few workgroups that own heavily duplicated ids, on the step's critical path, and cost
        # more there than the 4 us launch they save.
        fold = self.r_lstm > 0 and self._emb_sparse_ok(E, E) and not self.__dict__.get("agc") and getattr(self, "fold_emb_drop", False)
        # ... else it shares a launch with the encoder tail backward (both read the dXin the GEMM above just wrote, different
        # rows), which then runs in front of the Embedding backward
        ride = (self.r_lstm > 0 and not fold and fused and E % 4 == 0 and hasattr(be, "enc_tail_bwd_drop")
                and getattr(self, "tail_drop_ride", True))
        if ride:
            be.enc_tail_bwd_drop(self.dXin, self.xhat, a.p("batch_norm/gamma"), self.inv_std, self.enc_pre, self.dpre,
                                 a.g("batch_norm/gamma"), a.g("batch_norm/beta"), a.g("dense_img/bias"), B, E, E,
                                 self.r_feat, self.r_lstm, 0.2, sd, S_FEAT, S_LSTM_IN + 0, ds,
                                 self.dXin[B:], T * B, E, E, B, E, 0, self.r_lstm, S_LSTM_IN + 1)
        elif self.r_lstm > 0 and not fold:
            be.dropout(self.dXin[B:], self.dXin[B:], T * B, E, E, B, E, 0, self.r_lstm, sd, S_LSTM_IN + 1, 0, ds)
        self._emb_rows = (self.dXin[B:], T * B, E, E, "emb_text/embeddings")
        # id 0 is the Embedding's mask (mask_zero, NIC.py:77) and the text LSTM honours it: a masked step hands no gradient to
        # its input row, so the rows of id 0 are zero (BPTT leaves dz = 0 there and dXin = dz W^T)
        self._embedding_bwd(self.dXin[B:], self.cap, "emb_text/embeddings", B, T, E, E, V,
                            drop=(self.r_lstm, sd, S_LSTM_IN + 1, ds) if fold else None, zero_id=0)
        if ride:
            return
        if fused:       # dropout' -> BatchNorm' -> dropout' -> LeakyReLU' -> dpre, encoder bias gradient: one launch
            be.enc_tail_bwd(self.dXin, self.xhat, a.p("batch_norm/gamma"), self.inv_std, self.enc_pre, self.dpre,
                            a.g("batch_norm/gamma"), a.g("batch_norm/beta"), a.g("dense_img/bias"), B, E, E,
                            self.r_feat, self.r_lstm, 0.2, sd, S_FEAT, S_LSTM_IN + 0, ds)
            return
        if self.norm == "batch":
            self._bn_bwd(self.dXin, self.xhat, a.p("batch_norm/gamma"), self.inv_std, self.dyd,
                         a.g("batch_norm/gamma"), a.g("batch_norm/beta"), B, E, E, self.work)
        else:
            be.layernorm_bwd(self.dXin, self.xhat, a.p("batch_norm/gamma"), self.inv_std, self.dyd,
                             a.g("batch_norm/gamma"), a.g("batch_norm/beta"), B, E, E, self.work)
        if self.r_feat > 0:
            be.dropout(self.dyd, self.dyd, B, E, E, 0, E, 0, self.r_feat, sd, S_FEAT, 0, ds)
        be.act_bwd(self.enc_pre, self.dyd, self.dpre, B * E, ACT_LEAKY, 0.2)
        be.colsum(self.dpre, a.g("dense_img/bias"), B, E, E, self.work)

    def _bwd_enc(self, B, T, x_all=None, dpre_all=None):
        """encoder kernel gradient dW = X^T dpre.  Under DP the operands of all ranks are passed
        (all-gathered: 5 MB of betas + 128 KB per rank) and every rank computes the full global-batch
        gradient itself, instead of all-reducing the 41 MB result."""
        be, a = self.be, self.arena
        x = self.xd if self.r_in > 0 else self.x
        rows = B
        if x_all is not None:
            x, dpre, rows = x_all, dpre_all, x_all.shape[0]
        else:
            dpre = self.dpre
        self._enc_last_fused = None
        if x_all is None and self._enc_update_fused(rows):
            self._enc_fused = ("dense_img/kernel", x, dpre, rows, self.N, self.E, self.ldx,
                               self.__dict__.get("_enc_gram"))                                    # -> _update_fused
            self._enc_last_fused = (x, dpre, rows)       # train_step marks the gradient buffer stale after every (re)play
            return
        if rows <= 64 and self.E % 16 == 0 and getattr(self, "skinny_dw", True):
            be.dense_dw_skinny(x, dpre, a.g("dense_img/kernel"), self.N, self.E, rows, self.ldx)
        else:
            self.gemm_sk(x, dpre, a.g("dense_img/kernel"), self.N, self.E, rows, self.ldx, self.E, self.E, transA=True)

    # ---- row-sharded update of the encoder kernel under data parallel (dp.PipelinedDenseSync(shard_encoder=True)): rank r forms
    # rows [r0, r0 + nr) of dW = X_all^T dpre_all from the gathered operands (N / G rows: the single-process product's work
    # instead of G times it), the variable's clip norm is one 8-byte all-reduce of the shards' (sum g^2, sum theta^2), Adam runs
    # on the shard (1 / G of the 246 MB the update of this variable moves) and the updated rows are all-gathered.
    def _enc_shard_state(self, r0, nr):
        st = self.__dict__.get("_enc_shard")
        if st is None or st["key"] != (r0, nr):
            from .arena import build_spans
            a, e = self.arena, self.arena.entries["dense_img/kernel"]
            assert e.seg == 0 and e.off == 0
            sp = build_spans([e.off + r0 * self.E], [nr * self.E], self.device)
            st = self._enc_shard = dict(key=(r0, nr), sp=sp, scratch=self._f(2 * sp.nspan), send=self._f(nr * self.E))
            a.partial[:2 * a.spans.first_host[1]].zero_()          # the variable's slots: only [0:2] carries data from here on
        return st

    def _enc_shard_grad(self, x_all, dpre_all, r0, nr):
        """rows [r0, r0 + nr) of the encoder kernel's global-batch gradient + the shard's norm pair -> arena.partial[0:2]"""
        be, a = self.be, self.arena
        st = self._enc_shard_state(r0, nr)
        sp = st["sp"]
        rows = x_all.shape[0]
        self.gemm_sk(x_all[:, r0:], dpre_all, a.g("dense_img/kernel")[r0:r0 + nr], nr, self.E, rows, self.ldx, self.E, self.E,
                     transA=True)
        be.seg_sqnorm(a.theta, a.grad, sp.span_seg, sp.span_off, sp.span_len, sp.seg_first, a.seg_l2, st["scratch"],
                      a.partial[0:1], a.partial[1:2], None, sp.nspan, 1)

    def _enc_shard_adam(self, r0, nr):
        """clip (norm of the WHOLE variable: arena.partial[0], all-reduced) + Adam on the shard; the updated rows -> send buffer"""
        be, a, opt = self.be, self.arena, self.optimizer
        st = self._enc_shard_state(r0, nr)
        sp = st["sp"]
        clip = opt.clipnorm if opt.clipnorm is not None else 0.0
        be.adam(a.theta, self.opt_m, self.opt_v, a.grad, sp.span_seg, sp.span_off, sp.span_len, a.seg_l2, a.partial[0:1],
                a.sq_override, sp.nspan, 0.0, self.lr_t_dev, opt.beta_1, opt.beta_2, opt.epsilon, clip, guard=self._guard_word())
        return st["send"]

    # ------------------------------------------------------------------ steps
    def _train_graph(self, B, T):
        self._forward(B, T, True)
        self._loss_metrics(B, T, True)
        self._backward(B, T)

    def _update_graph(self):
        self._apply_agc()
        self._norms_and_l2(self.met[2:3])
        self._apply_optimizer()

    def _train_and_update_graph(self, B, T):
        """the single-process step as one launch sequence: the loss / accuracy totals ride in the step-finalize launch"""
        if not getattr(self, "fused_update", True):          # A/B switch (tools/ab_attr.py): the unfused launch sequence
            self._train_graph(B, T)
            self._update_graph()
            return
        self._defer_sum2 = True
        try:
            self._train_graph(B, T)
        finally:
            self._defer_sum2 = False
        self._update_fused(self.met[2:3])

    def train_step(self, data):
        """NIC.train_step (NIC.py:198-252): data = ((betas, cap, a0, c0), target)."""
        if self.optimizer is None:
            raise RuntimeError("compile() the model before train_step")
        B, T = self._stage_batch(data[0], data[1], self.N)
        self._sync_lr()
        self._enc_grad_stale = None
        ring = False
        if self.grad_sync is None:
            # The step is replayed as a recorded launch plan (ModelBase._run_planned), not as a hipGraph: with 15 (dense) / 32
            # (attention) launches a step the host re-issues them in ~15 % of the step's time and every launch starts ~0.2-0.4 us
            # earlier than as a graph node (0.4650 -> 0.4588 and 0.5666 -> 0.5626 ms/step, tools/probe/plan_bench.py: separate
            # models, repeated, spread 0.0005).  ``plan_step = False`` restores the graph.  A plan re-issues backend launches
            # only, so it is used where the step is nothing else: the sparse Embedding backward (the dense form hands its ids on
            # with a tensor copy, which a graph captures and a plan would drop).
            plan = (getattr(self, "plan_step", True) and self.E % 4 == 0 and getattr(self, "sparse_emb_bwd", True)
                    and hasattr(self.be, "embedding_bwd_sparse"))
            run = self._run_planned if plan else self._run_captured
            ring = self._run_step(run, ("train", B, T), lambda: self._train_and_update_graph(B, T))
            self._enc_grad_stale = self.__dict__.get("_enc_last_fused")
        elif getattr(self.grad_sync, "pipelined", False):
            self.grad_sync.step(self, B, T)
        else:       # data parallel: forward+backward | all-reduce of the flat gradient | update
            self._run_captured(("train_fb", B, T), lambda: self._train_graph(B, T))
            self.grad_sync(self)
            self._run_captured(("train_up", B, T), self._update_graph)
        self.optimizer.iterations += 1
        m = self._met_snapshot(ring)
        return self._metrics_from(m, loss=0, L2=2, accuracy=1)

    def test_step(self, data):
        """NIC.test_step (NIC.py:254-299)."""
        B, T = self._stage_batch(data[0], data[1], self.N)

        def run():
            self._forward(B, T, False)
            self._loss_metrics(B, T, False)
            self._norms_and_l2(self.met[2:3])
        self._run_captured(("test", B, T), run)
        m = self._met_snapshot()
        return self._metrics_from(m, loss=0, L2=2, accuracy=1)

    def __call__(self, data, training=False):
        """NIC.call (NIC.py:100-145): returns probabilities (B, T, V)."""
        B, T = self._stage_inputs(data)

        def run():
            self._forward(B, T, training)
            self.be.softmax_cce(self.logits, None, self.logits, None, None, None, T * B, self.V, self.ldV, 0.0)
            return self.logits.view(T, B, self.ldV)[:, :, :self.V].permute(1, 0, 2).contiguous()
        return self._guarded(run)

    call = __call__

    def greedy_predict(self, img_input, a0, c0, start_seq, max_len, units=None, tokenizer=None):
        """NIC.greedy_predict (NIC.py:148-195), inference mode; returns np.ndarray (max_len, B, 1, V).
        A predicted id 0 masks the following LSTM step exactly as the keras Embedding mask does."""
        be, a = self.be, self.arena
        start = self._to_dev(np.asarray(start_seq).reshape(-1), torch.int32)
        B = start.shape[0]
        cap = torch.zeros(B, 1, dtype=torch.int32, device=self.device)
        self._stage_inputs((img_input, cap, a0, c0))
        N, U, E, V, ldV = self.N, self.U, self.E, self.V, self.ldV
        # static decode buffers per (B, max_len): the whole loop is one captured hipGraph
        key = (B, max_len)
        bufs = self.__dict__.setdefault("_dec_bufs", {})
        if key not in bufs:
            bufs[key] = (torch.zeros(B, 1, dtype=torch.int32, device=self.device),
                         torch.zeros(B, 1, dtype=torch.int32, device=self.device),
                         torch.zeros(max_len, B, ldV, dtype=torch.float32, device=self.device))
        start_buf, words, probs_all = bufs[key]
        start_buf.copy_(start.view(B, 1))

        def run():
            self.gemm_sk(self.x, a.p("dense_img/kernel"), self.enc_y, B, E, N, self.ldx, E, E, bias=a.p("dense_img/bias"),
                         pre=self.enc_pre, act=ACT_LEAKY, slope=0.2)
            if self.norm == "batch":
                be.batchnorm_fwd(self.enc_y, a.p("batch_norm/gamma"), a.p("batch_norm/beta"), self.mov_mean, self.mov_var,
                                 self.Xin, self.xhat, self.inv_std, B, E, E, False, BN_EPS, BN_MOMENTUM, self.work)
            else:
                be.layernorm_fwd(self.enc_y, a.p("batch_norm/gamma"), a.p("batch_norm/beta"), self.Xin, self.xhat,
                                 self.inv_std, B, E, E, BN_EPS)
            Wl, bl, Ur = a.p("lstm/kernel"), a.p("lstm/bias"), a.p("lstm/recurrent_kernel")
            xz, emb = self.XZ[:B], self.Xin[B:2 * B]
            h = [self.Hs[0], self.Hs[1]]
            c = [self.Cs[0], self.Cs[1]]
            self.gemm_sk(self.Xin, Wl, xz, B, 4 * U, E, E, 4 * U, 4 * U, bias=bl)
            be.lstm_step_fwd(xz, h[0], c[0], Ur, None, None, 0, None, 0, 0, None, h[1], c[1], None, self.gates[0], B, U)
            cur = 1
            out = self.Out[0]
            for i in range(max_len):
                tok = start_buf if i == 0 else words
                be.embedding_fwd(a.p("emb_text/embeddings"), tok, emb, B, 1, E, E, V)
                self.gemm_sk(emb, Wl, xz, B, 4 * U, E, E, 4 * U, 4 * U, bias=bl)
                be.lstm_step_fwd(xz, h[cur], c[cur], Ur, None, None, 0, words if i > 0 else None, 1, 0, None,
                                 h[1 - cur], c[1 - cur], out, self.gates[0], B, U)
                cur = 1 - cur
                self.gemm_sk(out, a.p("time_distributed_softmax/kernel"), probs_all[i], B, V, U, U, ldV, ldV,
                             bias=a.p("time_distributed_softmax/bias"))
                be.softmax_cce(probs_all[i], None, probs_all[i], None, None, None, B, V, ldV, 0.0)
                be.argmax_rows(probs_all[i], words, B, V, ldV)
        self._run_captured(("greedy",) + key, run)
        return probs_all[:, :, :V].cpu().numpy()[:, :, None, :]
