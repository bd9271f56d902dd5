"""Fully-connected mode of the AttemptFour model: ``lc_NIC.call_fc`` + ``greedy_predict_fc`` over a
``FullyConnected`` voxel encoder.

Reference (paths under AttemptFour/Model): call_fc lc_NIC.py:298-323, greedy_predict_fc
lc_NIC.py:511-542, FullyConnected fullyConnected.py:8-27 (Dense -> BatchNorm -> Dropout; its
construction is the block kept commented at lc_NIC.py:60-67), dispatch lc_NIC.py:163-167 /
507-509 (the reference switches modes by editing those two methods; here it is a class).

Kept as written in the reference (see oracle/models_fc.py for the restatement):
  * call_fc discards the LSTM state computed from the feature (:317) and starts the text LSTM from
    (a0, c0) (:318), so the prediction is independent of the betas and the encoder gets no data
    gradient.  The training graph therefore only runs the encoder far enough to update the
    BatchNorm moving statistics; bias/gamma/beta keep zero gradient (their Adam slots stay 0,
    which is what skipping a None gradient amounts to), the kernel sees its L2 term only.
  * greedy_predict_fc does start from the feature state (:520).
  * the step dictionary has no 'attention' entry (call_fc returns None for the scores).
"""
from collections import OrderedDict

import numpy as np
import torch

from .arena import ParamArena
from .model_base import (Metrics, S_IN, S_TEXT, S_OUT, S_LSTM_IN, S_LSTM_OUT, BN_EPS, BN_MOMENTUM)
from .nic import NIC as _DenseNIC, _r4
from .ops import ACT_LEAKY


class NICfc(_DenseNIC):
    H = 256                                                       # lc_NIC.py:141
    BN = "dense_in_bn"

    def __init__(self, input_size, units, embedding_features, embedding_text, vocab_size, max_length, dropout_input,
                 dropout_features, dropout_text, dropout_lstm, dropout_out, input_reg, lstm_reg, output_reg, **kw):
        super(_DenseNIC, self).__init__(**kw)
        self.N, self.U, self.Ef, self.E, self.V = int(input_size), int(units), int(embedding_features), int(embedding_text), int(vocab_size)
        self.max_length = int(max_length)
        self.r_in, self.r_feat, self.r_text = float(dropout_input), float(dropout_features), float(dropout_text)
        self.r_lstm, self.r_out = float(dropout_lstm), float(dropout_out)
        self.l2_in, self.l2_lstm, self.l2_out = float(input_reg), float(lstm_reg), float(output_reg)
        self.norm = "batch"
        if self.U % 16:
            raise ValueError("units must be a multiple of 16 (LSTM step kernel tile)")
        N, U, Ef, E, V, H = self.N, self.U, self.Ef, self.E, self.V, self.H
        self.ldx, self.ldV = _r4(N), _r4(V)
        bn = self.BN
        self.layers_spec = OrderedDict([
            ("dense_in", ["kernel", "bias"]),
            (bn, ["gamma", "beta", "moving_mean", "moving_variance"]),
            ("emb_text", ["embeddings"]),
            ("lstm", ["kernel", "recurrent_kernel", "bias"]),
            ("time_distributed_nonlinear", ["kernel", "bias"]),
            ("time_distributed_softmax", ["kernel", "bias"])])
        self.keras_shapes = OrderedDict([
            ("dense_in/kernel", (N, Ef)), ("dense_in/bias", (Ef,)),
            (f"{bn}/gamma", (Ef,)), (f"{bn}/beta", (Ef,)), (f"{bn}/moving_mean", (Ef,)), (f"{bn}/moving_variance", (Ef,)),
            ("emb_text/embeddings", (V, E)),
            ("lstm/kernel", (E, 4 * U)), ("lstm/recurrent_kernel", (U, 4 * U)), ("lstm/bias", (4 * U,)),
            ("time_distributed_nonlinear/kernel", (U, H)), ("time_distributed_nonlinear/bias", (H,)),
            ("time_distributed_softmax/kernel", (H, V)), ("time_distributed_softmax/bias", (V,))])
        a = self.arena = ParamArena(self.device)
        a.add("dense_in/kernel", (N, Ef), self.l2_in)
        a.add("dense_in/bias", (Ef,))
        a.add(f"{bn}/gamma", (Ef,))
        a.add(f"{bn}/beta", (Ef,))
        a.add("emb_text/embeddings", (V, E))
        a.add("lstm/kernel", (E, U, 4), self.l2_lstm)
        a.add("lstm/recurrent_kernel", (U, U, 4))
        a.add("lstm/bias", (U, 4))
        a.add("time_distributed_nonlinear/kernel", (U, H), self.l2_out)
        a.add("time_distributed_nonlinear/bias", (H,))
        a.add("time_distributed_softmax/kernel", (H, self.ldV), self.l2_out)
        a.add("time_distributed_softmax/bias", (self.ldV,))
        a.finalize()
        self.mov_mean, self.mov_var = self._f(Ef), torch.ones(Ef, dtype=torch.float32, device=self.device)
        self.drop_step = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._init_weights(np.random.default_rng(self.seed))
        self._shape = None

    # ------------------------------------------------------------------ weights
    def _init_weights(self, rng):
        """Initialisers of lc_NIC.py:60-67,105-159 (he-style encoder, RandomUniform(+-0.08) embedding,
        keras LSTM defaults, GlorotNormal head)."""
        N, U, Ef, E, V, H = self.N, self.U, self.Ef, self.E, self.V, self.H
        tn = lambda shape, std: np.clip(rng.standard_normal(shape), -2, 2) * std / 0.8796
        self.set_weight("dense_in/kernel", tn((N, Ef), np.sqrt(2.0 / (N + Ef))))
        self.set_weight("emb_text/embeddings", rng.uniform(-0.08, 0.08, (V, E)))
        lim = np.sqrt(6.0 / (E + 4 * U))
        self.set_weight("lstm/kernel", rng.uniform(-lim, lim, (E, 4 * U)))
        q = np.concatenate([np.linalg.qr(rng.standard_normal((U, U)))[0] for _ in range(4)], axis=1)
        self.set_weight("lstm/recurrent_kernel", q)
        b = np.zeros(4 * U); b[U:2 * U] = 1.0
        self.set_weight("lstm/bias", b)
        self.set_weight("time_distributed_nonlinear/kernel", tn((U, H), np.sqrt(2.0 / (U + H))))
        self.set_weight("time_distributed_softmax/kernel", tn((H, V), np.sqrt(2.0 / (H + V))))
        self.set_weight(f"{self.BN}/gamma", np.ones(Ef))

    def set_weight(self, name, arr):
        arr = np.asarray(arr, dtype=np.float32)
        assert tuple(arr.shape) == tuple(self.keras_shapes[name]), (name, arr.shape, self.keras_shapes[name])
        if name == f"{self.BN}/moving_mean":
            self.mov_mean.copy_(torch.from_numpy(arr)); return
        if name == f"{self.BN}/moving_variance":
            self.mov_var.copy_(torch.from_numpy(arr)); return
        if name == "time_distributed_softmax/kernel":
            pad = np.zeros((self.H, self.ldV), np.float32); pad[:, :self.V] = arr
            self.arena.p(name).copy_(torch.from_numpy(pad)); return
        super().set_weight(name, arr)

    def get_weight(self, name):
        if name == f"{self.BN}/moving_mean":
            return self.mov_mean.cpu().numpy().copy()
        if name == f"{self.BN}/moving_variance":
            return self.mov_var.cpu().numpy().copy()
        return self._unpack(name, self.arena.p(name))

    # ------------------------------------------------------------------ buffers
    def _build(self, B, T):
        if self._shape == (B, T):
            return
        f = self._f
        N, U, Ef, E, V, H, ldV = self.N, self.U, self.Ef, self.E, self.V, self.H, self.ldV
        n = T * B
        self.x = f(B, self.ldx)
        self.xd = f(B, self.ldx) if self.r_in > 0 else self.x
        self.cap = torch.zeros(B, T, dtype=torch.int32, device=self.device)
        self.tgt = torch.zeros(n, dtype=torch.int32, device=self.device)
        self.enc_pre, self.enc_y, self.feat, self.xhat = f(B, Ef), f(B, Ef), f(B, Ef), f(B, Ef)
        self.inv_std = f(max(B, Ef))
        self.text = f(n, E)
        self.XZ = f(n + B, U, 4)
        self.Hs, self.Cs = f(T + 2, B, U), f(T + 2, B, U)
        self.gates = f(T + 1, B, U, 4)
        self.Out = f(T, B, U)
        self._init_seq_lstm(B, U)
        self.Out_d = f(T, B, U) if self.r_lstm > 0 else self.Out
        self.inter, self.ipre = f(n, H), f(n, H)
        self.inter_d = f(n, H) if self.r_out > 0 else self.inter
        self.logits = f(n, ldV)
        self.loss_row, self.corr_row = f(n), f(n)
        self.met = f(8)
        self.dinter, self.dOut = f(n, H), f(n, U)
        self.dZ = f(n, U, 4)
        self.da_pass, self.dc, self.dout = f(B, U), f(B, U), f(B, U)
        self.dtext = f(n, E)
        self._alloc_splitk([(B, Ef, N), (n, H, U), (n, V, H), (n, 4 * U, E), (n, E, 4 * U), (H, V, n), (U, 4 * U, n)])
        nch = max(self.be.bn_nchunk(B), self.be.bn_nchunk(n))
        self.work = f(max(Ef, E, 4 * U, ldV) * (2 * nch + 1))
        self.rowsq = f(n)
        self.emb_seg = self.arena.entries["emb_text/embeddings"].seg
        self._shape = (B, T)
        self._graphs = {}
        if self.optimizer is not None and getattr(self, "opt_m", None) is None:
            self._init_optimizer_state()
        self.built = True

    # ------------------------------------------------------------------ forward
    def _encode(self, B, training):
        """FullyConnected.call (fullyConnected.py:20-27) after dropout_input (lc_NIC.py:301)."""
        be, a = self.be, self.arena
        x = self.x
        if training and self.r_in > 0:
            be.dropout(self.x, self.xd, B, self.N, self.ldx, 0, self.N, 0, self.r_in, self.seed, S_IN, 0, self.drop_step)
            x = self.xd
        self.gemm_sk(x, a.p("dense_in/kernel"), self.enc_y, B, self.Ef, self.N, self.ldx, self.Ef, self.Ef,
                     bias=a.p("dense_in/bias"), pre=self.enc_pre, act=ACT_LEAKY, slope=0.2)
        be.batchnorm_fwd(self.enc_y, a.p(f"{self.BN}/gamma"), a.p(f"{self.BN}/beta"), self.mov_mean, self.mov_var,
                         self.feat, self.xhat, self.inv_std, B, self.Ef, self.Ef, training, BN_EPS, BN_MOMENTUM, self.work)

    def _forward(self, B, T, training):
        be, a = self.be, self.arena
        U, E, V, H, ldV = self.U, self.E, self.V, self.H, self.ldV
        n = T * B
        sd, ds = self.seed, self.drop_step
        if training:
            self._encode(B, True)          # output unused (lc_NIC.py:317); the BN moving statistics are not
        be.embedding_fwd(a.p("emb_text/embeddings"), self.cap, self.text, B, T, E, E, V)            # :307
        if training and self.r_text > 0:                                                            # :308
            be.dropout(self.text, self.text, n, E, E, B, E, 0, self.r_text, sd, S_TEXT, 0, ds)
        if training and self.r_lstm > 0:                                                            # LSTM(dropout=)
            be.dropout(self.text, self.text, n, E, E, B, E, 0, self.r_lstm, sd, S_LSTM_IN + 1, 0, ds)
        self.gemm_sk(self.text, a.p("lstm/kernel"), self.XZ, n, 4 * U, E, E, 4 * U, 4 * U)   # bias: in the step kernel
        Ur, bl = a.p("lstm/recurrent_kernel"), a.p("lstm/bias")
        if self._seq_lstm:       # the T masked steps (:318) as one persistent launch, see nic.NIC._forward
            be.lstm_seq_fwd(self.XZ, self.Hs, self.Cs, Ur, bl, self.cap, T, 0, self.Out, self.gates, T, B, U, self.seq_sync,
                            self._guard_out())
        else:
            for t in range(T):                                                                      # :318
                be.lstm_step_fwd(self.XZ[t * B:(t + 1) * B], self.Hs[t], self.Cs[t], Ur, None, None, 0, self.cap, T, t,
                                 self.Out[t - 1] if t > 0 else None, self.Hs[t + 1], self.Cs[t + 1], self.Out[t],
                                 self.gates[t], B, U, xz_bias=bl)
        out = self.Out
        if training and self.r_lstm > 0:                                                            # :321
            be.dropout(self.Out, self.Out_d, n, U, U, B, U, 0, self.r_lstm, sd, S_LSTM_OUT, 0, ds)
            out = self.Out_d
        self._out_used = out
        self.gemm_sk(out, a.p("time_distributed_nonlinear/kernel"), self.inter, n, H, U, U, H, H,
                     bias=a.p("time_distributed_nonlinear/bias"), pre=self.ipre, act=ACT_LEAKY, slope=0.2)
        inter = self.inter
        if training and self.r_out > 0:                                                             # :322
            be.dropout(self.inter, self.inter_d, n, H, H, B, H, 0, self.r_out, sd, S_OUT, 0, ds)
            inter = self.inter_d
        self._inter_used = inter
        self.gemm_sk(inter, a.p("time_distributed_softmax/kernel"), self.logits, n, V, H, H, ldV, ldV,
                     bias=a.p("time_distributed_softmax/bias"))                                    # :323

    # ------------------------------------------------------------------ backward
    def _backward(self, B, T):
        be, a = self.be, self.arena
        U, E, V, H, ldV = self.U, self.E, self.V, self.H, self.ldV
        n = T * B
        sd, ds = self.seed, self.drop_step
        dlog, inter, out = self.logits, self._inter_used, self._out_used
        self.gemm_sk(inter, dlog, a.g("time_distributed_softmax/kernel"), H, V, n, H, ldV, ldV, transA=True)
        be.colsum(dlog, a.g("time_distributed_softmax/bias"), n, V, ldV, self.work)
        self.gemm_sk(dlog, a.p("time_distributed_softmax/kernel"), self.dinter, n, H, V, ldV, ldV, H, transB=True)
        if self.r_out > 0:
            be.dropout(self.dinter, self.dinter, n, H, H, B, H, 0, self.r_out, sd, S_OUT, 0, ds)
        be.act_bwd(self.ipre, self.dinter, self.dinter, n * H, ACT_LEAKY, 0.2)
        self.gemm_sk(out, self.dinter, a.g("time_distributed_nonlinear/kernel"), U, H, n, U, H, H, transA=True)
        be.colsum(self.dinter, a.g("time_distributed_nonlinear/bias"), n, H, H, self.work)
        self.gemm_sk(self.dinter, a.p("time_distributed_nonlinear/kernel"), self.dOut, n, U, H, H, H, U, transB=True)
        if self.r_lstm > 0:
            be.dropout(self.dOut, self.dOut, n, U, U, B, U, 0, self.r_lstm, sd, S_LSTM_OUT, 0, ds)
        Ur = a.p("lstm/recurrent_kernel")
        dOut = self.dOut.view(T, B, U)
        seqb = self._seq_lstm and self.seq_xch is not None
        if seqb:       # BPTT as one persistent launch, see nic.NIC._bwd_seq_lstm
            be.lstm_seq_bwd(Ur, dOut, self.cap, T, 0, self.gates, self.Cs, self.dZ, self.seq_xch, T, B, U, self.seq_sync,
                            self._guard_out())
        else:
            for t in range(T - 1, -1, -1):
                last = t == T - 1
                be.lstm_step_bwd(None if last else self.dZ[(t + 1) * B:(t + 2) * B], Ur, None if last else self.da_pass,
                                 None, None if last else self.dc, None if last else self.dout, dOut[t], self.cap, T, t,
                                 self.gates[t], self.Cs[t + 1], self.Cs[t], self.dZ[t * B:(t + 1) * B], self.da_pass,
                                 self.dc, self.dout, B, U)
        hprev = self.Hs[:T].view(n, U)
        self.gemm_sk(hprev, self.dZ, a.g("lstm/recurrent_kernel"), U, 4 * U, n, U, 4 * U, 4 * U, transA=True)
        self.gemm_sk(self.text, self.dZ, a.g("lstm/kernel"), E, 4 * U, n, E, 4 * U, 4 * U, transA=True)
        be.colsum(self.dZ, a.g("lstm/bias"), n, 4 * U, 4 * U, self.work)
        self.gemm_sk(self.dZ, a.p("lstm/kernel"), self.dtext, n, E, 4 * U, 4 * U, 4 * U, E, transB=True)
        if self.r_lstm > 0:
            be.dropout(self.dtext, self.dtext, n, E, E, B, E, 0, self.r_lstm, sd, S_LSTM_IN + 1, 0, ds)
        if self.r_text > 0:
            be.dropout(self.dtext, self.dtext, n, E, E, B, E, 0, self.r_text, sd, S_TEXT, 0, ds)
        sqo = a.sq_override[self.emb_seg:self.emb_seg + 1]
        be.embedding_bwd(self.dtext, self.cap, a.g("emb_text/embeddings"), sqo, self.rowsq, B, T, E, E, V)
        # encoder gradients stay zero (arena.grad is zero-initialised and never written for them)

    # ------------------------------------------------------------------ steps
    def _train_graph(self, B, T):
        self._forward(B, T, True)
        self._loss_metrics(B, T, True)
        self._backward(B, T)

    def train_step(self, data):
        """lc_NIC.train_step (lc_NIC.py:328-408) over call_fc: {loss, L2, accuracy, lr}."""
        if self.optimizer is None:
            raise RuntimeError("compile() the model before train_step")
        B, T = self._stage_batch(data[0], data[1], self.N)
        self._sync_lr()
        ring = False
        if self.grad_sync is None:
            ring = self._run_step(self._run_captured, ("train", B, T), lambda: self._train_and_update_graph(B, T))
        else:
            self._run_captured(("train_fb", B, T), lambda: self._train_graph(B, T))
            self.grad_sync(self)
            self._run_captured(("train_up", B, T), self._update_graph)
        self.optimizer.iterations += 1
        m = self._met_snapshot(ring)
        return self._metrics_from(m, loss=0, L2=2, accuracy=1, lr=self.lr_dev.clone()[0])

    def __call__(self, data, training=False):
        """lc_NIC.call_fc (lc_NIC.py:298-323): returns (probabilities (B,T,V), None)."""
        B, T = self._stage_inputs(data)

        def run():
            self._forward(B, T, training)
            self.be.softmax_cce(self.logits, None, self.logits, None, None, None, T * B, self.V, self.ldV, 0.0)
            return self.logits.view(T, B, self.ldV)[:, :, :self.V].permute(1, 0, 2).contiguous()
        return self._guarded(run), None

    call = call_fc = __call__

    def greedy_predict(self, img_input, a0, c0, start_seq, max_len, units=None, tokenizer=None, training=False):
        """lc_NIC.greedy_predict_fc (lc_NIC.py:511-542): returns ids np.ndarray (max_len, B, 1) int64.
        The reference stops stepping once every sample has emitted 0 (:526-527); stepping on gives the
        same ids (a masked sample carries its state, so it re-emits 0), which lets the loop stay on the
        device without a host sync per word."""
        be, a = self.be, self.arena
        if self.Ef != self.E:
            raise ValueError("greedy_predict_fc feeds the feature through the text LSTM kernel: "
                             "embedding_features must equal embedding_text")
        start = self._to_dev(np.asarray(start_seq).reshape(-1), torch.int32)
        B = start.shape[0]
        self._stage_inputs((img_input, torch.zeros(B, max(1, max_len), dtype=torch.int32), a0, c0))
        U, E, V, H, ldV = self.U, self.E, self.V, self.H, self.ldV
        self._encode(B, False)
        Wl, bl, Ur = a.p("lstm/kernel"), a.p("lstm/bias"), a.p("lstm/recurrent_kernel")
        xz, emb = self.XZ[:B], self.text[:B]
        h, c = [self.Hs[0], self.Hs[1]], [self.Cs[0], self.Cs[1]]
        self.gemm_sk(self.feat, Wl, xz, B, 4 * U, E, E, 4 * U, 4 * U, bias=bl)
        be.lstm_step_fwd(xz, h[0], c[0], Ur, None, None, 0, None, 0, 0, None, h[1], c[1], None, self.gates[0], B, U)   # :520
        cur = 1
        words = start.clone().view(B, 1)
        ids = torch.zeros(max_len, B, dtype=torch.int32, device=self.device)
        probs = self.logits[:B]
        for i in range(max_len):
            be.embedding_fwd(a.p("emb_text/embeddings"), words, emb, B, 1, E, E, V)
            self.gemm_sk(emb, Wl, xz, B, 4 * U, E, E, 4 * U, 4 * U, bias=bl)
            be.lstm_step_fwd(xz, h[cur], c[cur], Ur, None, None, 0, words if i > 0 else None, 1, 0, None, h[1 - cur],
                             c[1 - cur], None, self.gates[0], B, U)                                   # :529
            cur = 1 - cur
            self.gemm_sk(h[cur], a.p("time_distributed_nonlinear/kernel"), self.inter[:B], B, H, U, U, H, H,
                         bias=a.p("time_distributed_nonlinear/bias"), act=ACT_LEAKY, slope=0.2)       # :531
            self.gemm_sk(self.inter[:B], a.p("time_distributed_softmax/kernel"), probs, B, V, H, H, ldV, ldV,
                         bias=a.p("time_distributed_softmax/bias"))                                  # :532
            be.softmax_cce(probs, None, probs, None, None, None, B, V, ldV, 0.0)
            be.argmax_rows(probs, ids[i], B, V, ldV)                                                  # :535
            words = ids[i].view(B, 1)
        return ids.cpu().numpy().astype(np.int64)[:, :, None]

    greedy_predict_fc = greedy_predict
