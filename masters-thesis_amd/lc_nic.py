"""lc_NIC -- region-wise encoder + additive attention + LSTM decoder (BASELINE config 3).

Drop-in for ``AttemptFour/Model/lc_NIC.py`` (class NIC, lines 36-838): same 17 positional
constructor arguments (lc_NIC.py:42; call site main.py:113-132), ``call`` =
``call_attention`` (223-263), ``train_step`` (328-408), ``test_step`` (410-459),
``greedy_predict`` = ``greedy_predict_attention`` (577-638).  ``groups`` is the
``(list_of_index_arrays, list_of_out_dims)`` pair of load_avg_betas.get_groups
(load_avg_betas.py:96-114).

Launch plan of one train step (hipGraph-captured):
  encoder : [dropout] -> ONE locally-dense launch (R regions) -> BatchNorm over (B,R) -> [dropout]
  hoisted : embedding gather -> [dropouts] -> text half of the LSTM input projection for all T
            steps in one GEMM; P = LeakyReLU(W1 F + b1) once (the reference recomputes it T times)
  T steps : fused attention step kernel -> fused LSTM step kernel (recurrent + context matmul + gates)
  head    : [dropout] -> Dense(256, LeakyReLU) GEMM -> [dropout] -> vocab GEMM -> softmax/CE/dlogits
  backward: mirror image; the T-step chain is lstm_step_bwd -> (dZ Wc^T) GEMM -> attention_step_bwd
  update  : [all-reduce] -> per-variable norms -> clip + Adam over the flat arena
"""
from collections import OrderedDict

import numpy as np
import torch

from .arena import ParamArena
from .model_base import (ModelBase, Metrics, interleave_gates, deinterleave_gates, S_IN, S_FEAT, S_TEXT, S_OUT, S_ATTN,
                         S_LSTM_IN, S_LSTM_OUT, S_SAMPLE, BN_EPS, BN_MOMENTUM)

SUBJ_SITE = 1000      # dropout-site offset per subject (multi-subject model)
S_FEAT2 = 4           # second application of the feature dropout (ms2_NIC.py:214)
S_DEEP = 6            # + i: feature dropout behind deep stage i of the depth-n encoder (deep_layers.py:58)
from .ops import ACT_LEAKY


def _r4(n):
    return (n + 3) // 4 * 4


def synthetic_groups(n_voxels, n_regions, out_dim, seed=42, overlap=0.0):
    """Pseudo-Glasser parcellation for benchmarks: a partition of the voxel axis into
    ``n_regions`` ragged groups (sizes log-normal, min 8, mean ~ n_voxels/n_regions), optionally
    with a fraction of extra overlapping indices.  Returns the reference's ``groups`` pair."""
    rng = np.random.default_rng(seed)
    w = rng.lognormal(0.0, 0.6, n_regions)
    sizes = np.maximum(8, np.floor(w / w.sum() * (n_voxels - 8 * n_regions)).astype(np.int64) + 8)
    while sizes.sum() > n_voxels:
        sizes[np.argmax(sizes)] -= 1
    sizes[np.argmin(sizes)] += n_voxels - sizes.sum()
    perm = rng.permutation(n_voxels)
    cuts = np.cumsum(sizes)[:-1]
    groups = [np.sort(g) for g in np.split(perm, cuts)]
    if overlap > 0:
        groups = [np.unique(np.concatenate([g, rng.choice(n_voxels, max(1, int(overlap * len(g))))])) for g in groups]
    return groups, [out_dim] * n_regions


def compact_groups(groups):
    """Input-pipeline helper for full-cortex widths: the region-wise encoder only ever reads the voxel
    columns its groups reference (after ``select_groups``, main.py:115, a good part of the 327 684-wide betas is
    unused).  Returns ``(used, groups2)``: ``used`` = sorted unique referenced columns, ``groups2`` = the same groups
    re-indexed into ``betas[:, used]``.  A generator that keeps / loads only ``betas[:, used]`` ships that many
    fewer bytes over PCIe per batch; the model built on ``groups2`` computes exactly the same function."""
    in_groups, out_groups = groups
    used = np.unique(np.concatenate([np.asarray(g, dtype=np.int64) for g in in_groups]))
    remap = np.full(int(used.max()) + 1, -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    return used, ([remap[np.asarray(g, dtype=np.int64)] for g in in_groups], list(out_groups))


class NIC(ModelBase):
    H = 256     # dense_inter width, hard-coded at lc_NIC.py:141
    PIECE = 64  # voxels per encoder workgroup (split mode)

    def __init__(self, groups, units, embedding_features, embedding_text, attn_units, vocab_size, max_length,
                 dropout_input, dropout_features, dropout_text, dropout_attn, dropout_lstm, dropout_out, input_reg,
                 attn_reg, lstm_reg, output_reg, norm="batch", n_subjects=1, depth=0, use_layer_norm=False, **kw):
        super().__init__(**kw)
        # use_layer_norm: the decoder cell is tensorflow_addons' LayerNormLSTMCell (lc_NIC.py:115,126-136; hard-wired off in
        # the reference): LayerNorm on x W, on h U and on the new cell state, no dropout inside the cell
        self.use_layer_norm = bool(use_layer_norm)
        # depth > 0: deep_layers.LocallyDense(groups, dropout, depth=n) (AttemptFour/Model/deep_layers.py:15-75) in place
        # of layers.LocallyDense -- n more stages of {per-region Dense(D -> D), BatchNorm, Dropout} behind the first one
        self.depth = int(depth)
        if self.depth and int(n_subjects) != 1:
            raise ValueError("the depth-n encoder is a single-subject variant")
        if not 0 <= self.depth < 10:
            raise ValueError("encoder depth must be in 0..9")
        # n_subjects > 1: AttemptFour/Model/ms2_NIC.py generalised to S subjects -- one region-wise
        # encoder (+ its own BatchNorm) per subject on S equal batch slices, shared decoder.
        self.S = int(n_subjects)
        self.enc_name = (lambda q: "dense_in") if self.S == 1 else (lambda q: f"dense_in_{q}")
        self.bn_name = (lambda q: "input_bn") if self.S == 1 else (lambda q: f"input_bn_{q}")
        in_groups, out_groups = groups
        assert len(in_groups) == len(out_groups), "Input groups don't match ouput groups"   # layers.py:30
        self.groups = [np.asarray(g, dtype=np.int64) for g in in_groups]
        self.R = len(self.groups)
        self.D = int(out_groups[0])
        if any(int(d) != self.D for d in out_groups):
            raise ValueError("all regions must project to the same width (they are stacked, layers.py:47-48)")
        self.U, self.Et, self.A, self.V, self.max_length = int(units), int(embedding_text), int(attn_units), int(vocab_size), int(max_length)
        self.embedding_features = embedding_features
        self.r_in, self.r_feat, self.r_text = float(dropout_input), float(dropout_features), float(dropout_text)
        self.r_attn, self.r_lstm, self.r_out = float(dropout_attn), float(dropout_lstm), float(dropout_out)
        self.l2_in, self.l2_attn, self.l2_lstm, self.l2_out = float(input_reg), float(attn_reg), float(lstm_reg), float(output_reg)
        assert norm in ("batch", "layer")
        self.norm = norm
        if self.U % 16 or self.D % 16 or self.D > 64 or self.A > 64:
            raise ValueError("kernel tiles need units % 16 == 0, group_size % 16 == 0 and <= 64, attn_units <= 64")
        R, D, A, U, Et, V, H = self.R, self.D, self.A, self.U, self.Et, self.V, self.H
        self.ldV = _r4(V)
        self.n_in = int(max(int(g.max()) for g in self.groups if len(g)) + 1)
        self.goff_host = np.concatenate([[0], np.cumsum([len(g) for g in self.groups])]).astype(np.int32)
        self.goff = torch.tensor(self.goff_host, dtype=torch.int32, device=self.device)
        # pieces of at most PIECE voxels per workgroup of the encoder kernels: one large region (Glasser regions span
        # 8..400+ voxels) would otherwise set the kernel time (tnt_locally_dense_*_split_f32)
        vg, vr, vf, rf = [0], [], [], [0]
        for r in range(self.R):
            g0, g1 = int(self.goff_host[r]), int(self.goff_host[r + 1])
            k = g0
            while True:
                k2 = min(g1, k + self.PIECE)
                vg.append(k2); vr.append(r); vf.append(1 if k == g0 else 0)
                k = k2
                if k >= g1:
                    break
            rf.append(len(vr))
        ti = lambda a: torch.tensor(a, dtype=torch.int32, device=self.device)
        self.vgoff, self.vreg, self.vfirst, self.rfirst, self.NV = ti(vg), ti(vr), ti(vf), ti(rf), len(vr)
        self.idx = torch.tensor(np.concatenate(self.groups).astype(np.int32), dtype=torch.int32, device=self.device)
        ls = OrderedDict()
        ks = OrderedDict()
        for q in range(self.S):
            en, bn = self.enc_name(q), self.bn_name(q)
            for r, g in enumerate(self.groups):
                ls[f"{en}/{r}"] = ["kernel", "bias"]
                ks[f"{en}/{r}/kernel"] = (len(g), D)
                ks[f"{en}/{r}/bias"] = (D,)
            ls[bn] = ["gamma", "beta", "moving_mean", "moving_variance"]
            for w in ls[bn]:
                ks[f"{bn}/{w}"] = (D,)
        for i in range(self.depth):
            for r in range(self.R):
                ls[f"dense_in/deep{i}/{r}"] = ["kernel", "bias"]
                ks[f"dense_in/deep{i}/{r}/kernel"], ks[f"dense_in/deep{i}/{r}/bias"] = (D, D), (D,)
            ls[f"input_bn/deep{i}"] = ["gamma", "beta", "moving_mean", "moving_variance"]
            for w in ls[f"input_bn/deep{i}"]:
                ks[f"input_bn/deep{i}/{w}"] = (D,)
        for nm, shp in (("attention/W1", (D, A)), ("attention/W2", (U, A)), ("attention/V", (A, 1))):
            ls[nm] = ["kernel", "bias"]
            ks[f"{nm}/kernel"] = shp
            ks[f"{nm}/bias"] = (shp[1],)
        ls["emb_text"] = ["embeddings"]
        ks["emb_text/embeddings"] = (V, Et)
        ls["lstm"] = ["kernel", "recurrent_kernel", "bias"]
        ks["lstm/kernel"], ks["lstm/recurrent_kernel"], ks["lstm/bias"] = (D + Et, 4 * U), (U, 4 * U), (4 * U,)
        if self.use_layer_norm:
            for nm, n in (("kernel_norm", 4 * U), ("recurrent_norm", 4 * U), ("state_norm", U)):
                ls[f"lstm/{nm}"] = ["gamma", "beta"]
                ks[f"lstm/{nm}/gamma"], ks[f"lstm/{nm}/beta"] = (n,), (n,)
        ls["time_distributed_nonlinear"] = ["kernel", "bias"]
        ks["time_distributed_nonlinear/kernel"], ks["time_distributed_nonlinear/bias"] = (U, H), (H,)
        ls["time_distributed_softmax"] = ["kernel", "bias"]
        ks["time_distributed_softmax/kernel"], ks["time_distributed_softmax/bias"] = (H, V), (V,)
        self.layers_spec, self.keras_shapes = ls, ks

        a = self.arena = ParamArena(self.device)
        enc_off = []
        for q in range(self.S):
            en, bn = self.enc_name(q), self.bn_name(q)
            for r, g in enumerate(self.groups):                  # contiguous CSR concatenation
                a.add(f"{en}/{r}/kernel", (len(g), D), self.l2_in, align=4)
            w_off = a.entries[f"{en}/0/kernel"].off
            a.total = (a.total + 63) // 64 * 64
            for r in range(R):
                a.add(f"{en}/{r}/bias", (D,), align=4)
            enc_off.append((w_off, a.entries[f"{en}/0/bias"].off))
            a.total = (a.total + 63) // 64 * 64
            a.add(f"{bn}/gamma", (D,)); a.add(f"{bn}/beta", (D,))
        deep_off = []
        for i in range(self.depth):
            for r in range(R):
                a.add(f"dense_in/deep{i}/{r}/kernel", (D, D), self.l2_in, align=4)
            w_off = a.entries[f"dense_in/deep{i}/0/kernel"].off
            a.total = (a.total + 63) // 64 * 64
            for r in range(R):
                a.add(f"dense_in/deep{i}/{r}/bias", (D,), align=4)
            deep_off.append((w_off, a.entries[f"dense_in/deep{i}/0/bias"].off))
            a.total = (a.total + 63) // 64 * 64
            a.add(f"input_bn/deep{i}/gamma", (D,)); a.add(f"input_bn/deep{i}/beta", (D,))
        a.add("attention/W1/kernel", (D, A), self.l2_attn); a.add("attention/W1/bias", (A,))
        a.add("attention/W2/kernel", (U, A), self.l2_attn); a.add("attention/W2/bias", (A,))
        a.add("attention/V/kernel", (A,)); a.add("attention/V/bias", (1,))
        a.add("emb_text/embeddings", (V, Et))
        a.add("lstm/kernel", (D + Et, U, 4), self.l2_lstm)
        a.add("lstm/recurrent_kernel", (U, U, 4)); a.add("lstm/bias", (U, 4))
        if self.use_layer_norm:
            for nm in ("kernel_norm", "recurrent_norm"):
                a.add(f"lstm/{nm}/gamma", (U, 4)); a.add(f"lstm/{nm}/beta", (U, 4))
            a.add("lstm/state_norm/gamma", (U,)); a.add("lstm/state_norm/beta", (U,))
        a.add("time_distributed_nonlinear/kernel", (U, H), self.l2_out); a.add("time_distributed_nonlinear/bias", (H,))
        a.add("time_distributed_softmax/kernel", (H, self.ldV), self.l2_out)
        a.add("time_distributed_softmax/bias", (self.ldV,))
        a.finalize()
        nW = int(self.goff_host[-1]) * D
        self.encW = [a.theta[w:w + nW] for w, _ in enc_off]
        self.encWg = [a.grad[w:w + nW] for w, _ in enc_off]
        self.encB = [a.theta[b:b + R * D] for _, b in enc_off]
        self.encBg = [a.grad[b:b + R * D] for _, b in enc_off]
        self.deepW = [a.theta[w:w + R * D * D] for w, _ in deep_off]
        self.deepWg = [a.grad[w:w + R * D * D] for w, _ in deep_off]
        self.deepB = [a.theta[b:b + R * D] for _, b in deep_off]
        self.deepBg = [a.grad[b:b + R * D] for _, b in deep_off]
        self.deep_idx = torch.arange(R * D, dtype=torch.int32, device=self.device)
        self.deep_goff = torch.arange(0, (R + 1) * D, D, dtype=torch.int32, device=self.device)
        # moving statistics: one pair per subject encoder, then one pair per deep stage
        self.mov_mean = [self._f(D) for _ in range(self.S + self.depth)]
        self.mov_var = [torch.ones(D, dtype=torch.float32, device=self.device) for _ in range(self.S + self.depth)]
        self.drop_step = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._init_weights(np.random.default_rng(self.seed))
        self._shape = None

    # ------------------------------------------------------------------ weights
    def _init_weights(self, rng):
        """Initialisers of lc_NIC.py:84-159 / attention.py:21-23."""
        D, A, U, Et, V, H = self.D, self.A, self.U, self.Et, self.V, self.H
        tn = lambda shape, std: np.clip(rng.standard_normal(shape), -2, 2) * std / 0.8796
        for q in range(self.S):
            for r, g in enumerate(self.groups):
                self.set_weight(f"{self.enc_name(q)}/{r}/kernel", tn((len(g), D), np.sqrt(2.0 / max(len(g), 1))))  # he_normal
            self.set_weight(f"{self.bn_name(q)}/gamma", np.ones(D))
        for i in range(self.depth):
            for r in range(self.R):
                self.set_weight(f"dense_in/deep{i}/{r}/kernel", tn((D, D), np.sqrt(2.0 / D)))
            self.set_weight(f"input_bn/deep{i}/gamma", np.ones(D))
        self.set_weight("attention/W1/kernel", tn((D, A), np.sqrt(2.0 / D)))
        self.set_weight("attention/W2/kernel", tn((U, A), np.sqrt(2.0 / U)))
        lim = np.sqrt(6.0 / (A + 1))
        self.set_weight("attention/V/kernel", rng.uniform(-lim, lim, (A, 1)))
        self.set_weight("emb_text/embeddings", rng.uniform(-0.08, 0.08, (V, Et)))
        lim = np.sqrt(6.0 / (D + Et + 4 * U))
        self.set_weight("lstm/kernel", rng.uniform(-lim, lim, (D + Et, 4 * U)))
        q = np.concatenate([np.linalg.qr(rng.standard_normal((U, U)))[0] for _ in range(4)], axis=1)
        self.set_weight("lstm/recurrent_kernel", q)
        b = np.zeros(4 * U); b[U:2 * U] = 1.0
        self.set_weight("lstm/bias", b)
        if self.use_layer_norm:
            for nm, n in (("kernel_norm", 4 * U), ("recurrent_norm", 4 * U), ("state_norm", U)):
                self.set_weight(f"lstm/{nm}/gamma", np.ones(n))
        self.set_weight("time_distributed_nonlinear/kernel", tn((U, H), np.sqrt(2.0 / (U + H))))
        self.set_weight("time_distributed_softmax/kernel", tn((H, V), np.sqrt(2.0 / (H + V))))

    def set_weight(self, name, arr):
        arr = np.asarray(arr, dtype=np.float32)
        assert tuple(arr.shape) == tuple(self.keras_shapes[name]), (name, arr.shape, self.keras_shapes[name])
        for q, bn in self._bn_slots():
            if name == f"{bn}/moving_mean":
                self.mov_mean[q].copy_(torch.from_numpy(arr)); return
            if name == f"{bn}/moving_variance":
                self.mov_var[q].copy_(torch.from_numpy(arr)); return
        dst = self.arena.p(name)
        if name.startswith("lstm/") and arr.shape[-1] == 4 * self.U:
            arr = interleave_gates(arr, self.U)
        elif name == "time_distributed_softmax/kernel":
            pad = np.zeros((self.H, self.ldV), np.float32); pad[:, :self.V] = arr; arr = pad
        elif name == "time_distributed_softmax/bias":
            pad = np.zeros(self.ldV, np.float32); pad[:self.V] = arr; arr = pad
        dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).view(dst.shape))

    def _unpack(self, name, t):
        arr = t.detach().cpu().numpy()
        if name.startswith("lstm/") and not name.startswith("lstm/state_norm"):
            return deinterleave_gates(arr)
        if name == "time_distributed_softmax/kernel":
            return np.ascontiguousarray(arr[:, :self.V])
        if name == "time_distributed_softmax/bias":
            return np.ascontiguousarray(arr[:self.V])
        return arr.reshape(self.keras_shapes[name]).copy()

    def _bn_slots(self):
        """(slot of mov_mean / mov_var, keras layer name) of every BatchNormalization of the encoder"""
        return [(q, self.bn_name(q)) for q in range(self.S)] + [(self.S + i, f"input_bn/deep{i}") for i in range(self.depth)]

    def get_weight(self, name):
        for q, bn in self._bn_slots():
            if name == f"{bn}/moving_mean":
                return self.mov_mean[q].cpu().numpy().copy()
            if name == f"{bn}/moving_variance":
                return self.mov_var[q].cpu().numpy().copy()
        return self._unpack(name, self.arena.p(name))

    def get_gradient(self, name):
        return self._unpack(name, self.arena.g(name))

    def state_tensors(self):
        return list(self.mov_mean) + list(self.mov_var)

    @property
    def losses(self):
        a = self.arena
        self._norms_and_l2(self.met[2:3])
        return [a.seg_l2[e.seg] * a.wsq[e.seg] for e in a.entries.values() if e.l2 > 0]

    # ------------------------------------------------------------------ buffers
    def _build(self, B, T):
        if self._shape == (B, T):
            return
        if B % self.S:
            raise ValueError("batch must split into n_subjects equal slices")
        f = self._f
        R, D, A, U, Et, V, H, ldV = self.R, self.D, self.A, self.U, self.Et, self.V, self.H, self.ldV
        n = T * B
        self.ldx = _r4(self.n_in)
        self.x = f(B, self.ldx)
        self.xd = f(B, self.ldx) if self.r_in > 0 else self.x
        self.cap = torch.zeros(B, T, dtype=torch.int32, device=self.device)
        self.tgt = torch.zeros(n, dtype=torch.int32, device=self.device)
        self.enc_pre, self.enc_y = f(B, R, D), f(B, R, D)
        self.enc_part = f(self.NV, 64, D)                    # piece partials of the split encoder forward
        self.dctx_part = f(U // 16, B, D)                    # per-unit-block context-gradient partials of a BPTT step
        # voxel-major copy of the betas for the encoder's gather (a voxel's batch values contiguous); not with input
        # dropout, whose mask is defined on the batch-major tensor
        self.xT = f(self.n_in, (B + 3) // 4 * 4) if (self.r_in == 0 and self.ldx == self.n_in) else None
        self.xhat, self.inv_std = f(B * R, D), f(self.S, max(B * R, D))
        # depth-n encoder: stage i reads Fs[i] and writes Fs[i + 1]; the attention reads the last one
        self.Fs = [f(B, R, D) for _ in range(self.depth + 1)]
        self.F = self.Fs[-1]
        self.deep_pre = [f(B, R, D) for _ in range(self.depth)]
        self.deep_y = [f(B, R, D) for _ in range(self.depth)]
        self.deep_xhat = [f(B * R, D) for _ in range(self.depth)]
        self.deep_istd = [f(max(B * R, D)) for _ in range(self.depth)]
        self.dFd = f(B, R, D) if self.depth else None
        self.text = f(n, Et)
        self.XZ = f(n, U, 4)
        self.P, self.Ppre = f(B * R, A), f(B * R, A)
        self.Hs, self.Cs = f(T + 1, B, U), f(T + 1, B, U)
        self.gates = f(T, B, U, 4)
        self.qpre, self.alpha = f(T, B, A), f(T, B, R)
        self.ctx, self.ctx_d = f(T, B, D), f(T, B, D)
        self.Hd = f(n, U) if self.r_lstm > 0 else None
        if self.use_layer_norm:       # LayerNormLSTMCell: normalised projections, LayerNorm caches, per-step gradients
            self.ZK, self.ZR, self.ZRn = f(n, U, 4), f(n, U, 4), f(n, U, 4)
            self.xh_k, self.xh_r = f(n, 4 * U), f(n, 4 * U)
            self.is_k, self.is_r, self.is_s = f(T, max(B, 4 * U)), f(T, max(B, 4 * U)), f(T, B)
            self.chat, self.dcnt = f(n, U), f(n, U)
            self.dZK, self.dZR = f(n, U, 4), f(n, U, 4)
            self.dh_rec = f(B, U)
        self.ipre, self.inter = f(n, H), f(n, H)
        self.inter_d = f(n, H) if self.r_out > 0 else self.inter
        self.logits = f(n, ldV)
        self.loss_row, self.corr_row = f(n), f(n)
        self.met = f(8 + 4 * self.S)
        self._init_seq_lstm(B, U)        # device census + sync state of the persistent chain kernels (tnt_lc_seq_{fwd,bwd}_f32)
        self.lc_xch = None                # exchange space of the persistent backward chain
        if self.__dict__.get("_seq_lstm") and hasattr(self.be, "lc_seq_bwd") and B <= 128:
            self.lc_xch = f(self.be.lc_seq_bwd_work_floats(B, U))
        if self.__dict__.get("_seq_lstm") and hasattr(self.be, "lc_seq_fwd") and B <= 128:
            self.lc_fwd_work = f(self.be.lc_seq_fwd_work_floats(B))
        self.colB = f(2 * B)
        # backward
        self.dinter, self.dHs = f(n, H), f(n, U)
        self.dZ = f(n, U, 4)
        self.dc, self.dh_att = f(B, U), f(B, U)
        self.dctx = f(B, D)
        # the three accumulators of the attention backward sit in one allocation: one fill zeroes them per step
        pad4 = lambda v: -(-v // 4) * 4                      # every piece starts 16-byte aligned
        oF, ovb = pad4(B * R * A), pad4(B * R * A) + pad4(B * R * D)
        self.datt = f(ovb + pad4(B * (A + 1)))
        self.dP, self.dF = self.datt[:B * R * A].view(B * R, A), self.datt[oF:oF + B * R * D].view(B, R, D)
        self.dvb, self.dqpre = self.datt[ovb:ovb + B * (A + 1)].view(B, A + 1), f(T, B, A)
        # attention-dropout keep bits of all T timesteps (4 per byte), produced by ONE chip-wide launch per step instead
        # of Philox inside every per-timestep kernel on the serial chain (2 us forward + 2.4 us backward per timestep)
        self._keep_stored = self.r_attn > 0 and A % 4 == 0
        self.att_keep = (torch.zeros(T, B * R * A // 4, dtype=torch.uint8, device=self.device)
                         if self._keep_stored else None)
        self.dtext = f(n, Et)
        self.dbn = f(B * R, D)
        nch = max(self.be.bn_nchunk(B * R), self.be.bn_nchunk(n))
        self.work = f(max(D, A, 4 * U, ldV, H) * (2 * nch + 1))
        self.rowsq = f(n)
        self.metric_part = f(T * ((self.R + 63) // 64))      # partial sums of the attention metric (tnt_attention_metric_parts)
        self._alloc_splitk([(n, H, V), (n, Et, 4 * U), (n, U, H), (D, 4 * U, n), (U, A, n), (D, A, B * R)])
        self.emb_seg = self.arena.entries["emb_text/embeddings"].seg
        self._shape = (B, T)
        self._graphs = {}
        if self.optimizer is not None and getattr(self, "opt_m", None) is None:
            self._init_optimizer_state()
        self.built = True

    def _stage_inputs(self, data):
        x, cap, a0, c0 = data[:4]
        cap_t = self._to_dev(cap, torch.int32)
        if cap_t.dim() == 1:
            cap_t = cap_t.view(-1, 1)
        B, T = cap_t.shape
        self._build(B, T)
        xs = self._to_dev(x, torch.float32)
        assert xs.shape[0] == B and xs.shape[1] >= self.n_in, f"betas shape {tuple(xs.shape)}"
        self.x[:, :self.n_in].copy_(xs[:, :self.n_in])
        if self.xT is not None:
            self.xT[:, :B].copy_(self.x[:, :self.n_in].t())
        self.cap.copy_(cap_t)
        self.Hs[0].copy_(self._to_dev(a0, torch.float32))
        self.Cs[0].copy_(self._to_dev(c0, torch.float32))
        return B, T

    # ------------------------------------------------------------------ forward
    def _encode(self, B, training):
        """dropout_input -> layers.LocallyDense.call (lc_NIC.py:227-230; layers.py:43-53)."""
        be, a = self.be, self.arena
        R, D, S = self.R, self.D, self.S
        sd, ds = self.seed, self.drop_step
        Bs = B // S
        for q in range(S):                       # one encoder per subject on its batch slice (ms2_NIC.py:181-203)
            off = SUBJ_SITE * (q + 1) if S > 1 else 0
            r0, r1 = q * Bs, (q + 1) * Bs
            x = self.x[r0:r1]
            if training and self.r_in > 0:
                be.dropout(x, self.xd[r0:r1], Bs, self.n_in, self.ldx, 0, self.n_in, 0, self.r_in, sd, S_IN + off, 0, ds)
                x = self.xd[r0:r1]
            if self.NV > R and hasattr(be, "locally_dense_fwd_split") and getattr(self, "split_encoder", True):
                vm = self.xT is not None and getattr(self, "voxel_major", True)
                be.locally_dense_fwd_split(self.xT[:, r0:r1] if vm else x, self.xT.shape[1] if vm else self.ldx, self.idx,
                                           self.vgoff, self.vreg, self.rfirst, self.NV, self.encW[q], self.encB[q],
                                           self.enc_pre[r0:r1], self.enc_y[r0:r1], self.enc_part, Bs, R, D, 0.2,
                                           voxel_major=vm)
            else:
                be.locally_dense_fwd(x, self.ldx, self.idx, self.goff, self.encW[q], self.encB[q],
                                     self.enc_pre[r0:r1], self.enc_y[r0:r1], Bs, R, D, 0.2)
            bn = self.bn_name(q)
            Fq, xh = self.Fs[0][r0:r1], self.xhat[r0 * R:r1 * R]
            dropped = False
            if self.norm == "batch":                 # the feature Dropout (layers.py:51) rides in the BN apply pass
                drop = (self.r_feat, sd, S_FEAT + off, ds) if training and self.r_feat > 0 else None
                dropped = self._bn_fwd(self.enc_y[r0:r1], a.p(f"{bn}/gamma"), a.p(f"{bn}/beta"), self.mov_mean[q],
                                       self.mov_var[q], Fq, xh, self.inv_std[q], Bs * R, D, D, training, self.work, drop=drop)
            else:
                be.layernorm_fwd(self.enc_y[r0:r1], a.p(f"{bn}/gamma"), a.p(f"{bn}/beta"), Fq, xh, self.inv_std[q],
                                 Bs * R, D, D, BN_EPS)
            if training and self.r_feat > 0:
                if not dropped:
                    be.dropout(Fq, Fq, Bs * R, D, D, 0, D, 0, self.r_feat, sd, S_FEAT + off, 0, ds)        # layers.py:51
                if S > 1:                        # ms2_NIC.py:214,257: the feature Dropout is applied a second time
                    be.dropout(Fq, Fq, Bs * R, D, D, 0, D, 0, self.r_feat, sd, S_FEAT2 + off, 0, ds)
        for i in range(self.depth):              # deep_layers.one_layer (deep_layers.py:53-59): Dense_r(x[:, r, :]), BN, Dropout
            xin, out = self.Fs[i], self.Fs[i + 1]
            be.locally_dense_fwd(xin, R * D, self.deep_idx, self.deep_goff, self.deepW[i], self.deepB[i], self.deep_pre[i],
                                 self.deep_y[i], B, R, D, 0.2)
            bn, q = f"input_bn/deep{i}", self.S + i
            dropped = False
            if self.norm == "batch":
                drop = (self.r_feat, sd, S_DEEP + i, ds) if training and self.r_feat > 0 else None
                dropped = self._bn_fwd(self.deep_y[i], a.p(f"{bn}/gamma"), a.p(f"{bn}/beta"), self.mov_mean[q], self.mov_var[q],
                                       out, self.deep_xhat[i], self.deep_istd[i], B * R, D, D, training, self.work, drop=drop)
            else:
                be.layernorm_fwd(self.deep_y[i], a.p(f"{bn}/gamma"), a.p(f"{bn}/beta"), out, self.deep_xhat[i],
                                 self.deep_istd[i], B * R, D, D, BN_EPS)
            if training and self.r_feat > 0 and not dropped:
                be.dropout(out, out, B * R, D, D, 0, D, 0, self.r_feat, sd, S_DEEP + i, 0, ds)
        self.gemm_sk(self.F, a.p("attention/W1/kernel"), self.P, B * R, self.A, D, D, self.A, self.A,
                bias=a.p("attention/W1/bias"), pre=self.Ppre, act=ACT_LEAKY, slope=0.2)      # attention.py:32 (hoisted)

    def _lc_seq_ok(self):
        """the persistent forward-chain kernel applies (tnt_lc_seq_fwd_f32's shape limits; the device census was taken by
        _init_seq_lstm in _build)"""
        # Default on (use_lc_seq = False selects the two per-step launches).  MI355X, B = 64, R = 360, U = 512: the chain as one
        # launch of role-specialised workgroups (16 attention + 16 LSTM workgroups per XCD, step-invariant operands resident,
        # the recurrent product overlapped with the attention) takes config 3 from 1.010 to 0.920 ms/step
        # (tools/ab_attr.py attention use_lc_seq=False).  An earlier form that ran both phases on the same workgroups one
        # after the other was slower than the per-step launches (16.3 vs 15.6 us per step): attention and LSTM register sets
        # together spilled 58 VGPRs and half of each group idled through the attention phase.
        return bool(self.__dict__.get("_seq_lstm") and getattr(self, "use_lc_seq", True) and not self.use_layer_norm
                    and hasattr(self.be, "lc_seq_fwd") and self.R <= (512 if max(self.A, self.D) <= 32 else 384)
                    and self.A % 4 == 0 and self.D % 4 == 0 and self.A <= 64 and self.D <= 64)

    def _decode_step(self, i, B, training, s_out=None, xz_bias=None):
        """attention -> concat -> one LSTM step (lc_NIC.py:246-255)."""
        be, a = self.be, self.arena
        R, D, A, U, Et = self.R, self.D, self.A, self.U, self.Et
        Wl = a.p("lstm/kernel")
        r_in = self.r_lstm if (training and not self.use_layer_norm) else 0.0      # the LayerNorm cell has no input dropout
        be.attention_step_fwd(self.Hs[i], self.F, self.P, a.p("attention/W2/kernel"), a.p("attention/W2/bias"),
                              a.p("attention/V/kernel"), a.p("attention/V/bias"), self.qpre[i], self.alpha[i],
                              self.ctx[i], self.ctx_d[i], s_out, B, R, D, A, U, 0.2,
                              self.r_attn if training else 0.0, r_in, D + Et, self.seed,
                              S_ATTN + i, S_LSTM_IN + i, 0, self.drop_step,
                              keep4=self.att_keep[i] if (training and self._keep_stored) else None)
        if self.use_layer_norm:
            # LayerNormLSTMCell.call: z = LN_kernel([ctx, text] W) + LN_recurrent(h U) + b, state LayerNorm inside the cell
            rows = slice(i * B, (i + 1) * B)
            xz = self.XZ[rows]                       # text part of x W (all T steps in one GEMM, no bias) ...
            self.gemm_sk(self.ctx_d[i], Wl[:D], xz, B, 4 * U, D, D, 4 * U, 4 * U, accumulate=True)       # ... + ctx W_ctx
            be.layernorm_fwd(xz, a.p("lstm/kernel_norm/gamma"), a.p("lstm/kernel_norm/beta"), self.ZK[rows], self.xh_k[rows],
                             self.is_k[i], B, 4 * U, 4 * U, BN_EPS)
            self.gemm_sk(self.Hs[i], a.p("lstm/recurrent_kernel"), self.ZR[rows], B, 4 * U, U, U, 4 * U, 4 * U)
            be.layernorm_fwd(self.ZR[rows], a.p("lstm/recurrent_norm/gamma"), a.p("lstm/recurrent_norm/beta"), self.ZRn[rows],
                             self.xh_r[rows], self.is_r[i], B, 4 * U, 4 * U, BN_EPS)
            be.ln_lstm_cell_fwd(self.ZK[rows], self.ZRn[rows], a.p("lstm/bias"), self.Cs[i], a.p("lstm/state_norm/gamma"),
                                a.p("lstm/state_norm/beta"), self.gates[i], self.chat[rows], self.is_s[i], self.Cs[i + 1],
                                self.Hs[i + 1], B, U, BN_EPS)
            return
        be.lstm_step_fwd(self.XZ[i * B:(i + 1) * B], self.Hs[i], self.Cs[i], a.p("lstm/recurrent_kernel"),
                         self.ctx_d[i], Wl[:D], D, None, 0, 0, None, self.Hs[i + 1], self.Cs[i + 1], None,
                         self.gates[i], B, U, xz_bias=xz_bias)

    def _forward(self, B, T, training):
        be, a = self.be, self.arena
        R, D, A, U, Et, V, H, ldV = self.R, self.D, self.A, self.U, self.Et, self.V, self.H, self.ldV
        n = T * B
        sd, ds = self.seed, self.drop_step
        self._encode(B, training)
        lstm_in_mask = training and self.r_lstm > 0 and not self.use_layer_norm
        if training and self.r_text > 0 and Et % 4 == 0:     # lc_NIC.py:233 + its Dropout in one launch
            # ... and the text half of the per-call LSTM input mask over (B, 1, D + Et) behind it, when the layout allows
            ride = lstm_in_mask and D % 4 == 0 and getattr(self, "fused_text_masks", True)
            be.embedding_fwd_drop(a.p("emb_text/embeddings"), self.cap, None, self.text, B, T, Et, Et, V, self.r_text,
                                  sd, S_TEXT, 0, ds, mask2=(self.r_lstm, S_LSTM_IN, D + Et, D) if ride else None)
            lstm_in_mask = lstm_in_mask and not ride
        else:
            be.embedding_fwd(a.p("emb_text/embeddings"), self.cap, self.text, B, T, Et, Et, V)       # lc_NIC.py:233
            if training and self.r_text > 0:
                be.dropout(self.text, self.text, n, Et, Et, B, Et, 0, self.r_text, sd, S_TEXT, 0, ds)
        if lstm_in_mask:                                                   # text half of the per-call LSTM input mask over (B,1,D+Et)
            be.dropout(self.text, self.text, n, Et, Et, 0, D + Et, D, self.r_lstm, sd, S_LSTM_IN, 0, ds,
                       rows_per_site=B)
        Wl = a.p("lstm/kernel")
        # text half of the input projection for all T steps: one epilogue-free GEMM; bias added in the step kernel
        self.gemm_sk(self.text, Wl[D:], self.XZ, n, 4 * U, Et, Et, 4 * U, 4 * U)
        if training and self._keep_stored and not self.__dict__.get("_masks_staged"):
            # (a training step has them generated with its batch staging, outside the captured sequence: _stage_mask_job)
            be.dropout_mask4(self.att_keep, B * R * A, T, self.r_attn, sd, S_ATTN, 0, ds)
        if self._lc_seq_ok():
            # the T attention -> LSTM steps as ONE persistent launch (tnt_lc_seq_fwd_f32), XCD-local data-polling hand-offs
            r_in = self.r_lstm if training else 0.0
            keep = self.att_keep if (training and self._keep_stored) else None
            # (with the Dropout behind the LSTM, :256, as a rider of the same launch: Hd leaves the chain with hs)
            self._out_dropped = bool(training and self.r_lstm > 0 and (int(getattr(self, "fuse_out_drop", 3)) & 1))
            be.lc_seq_fwd(self.F, self.P, a.p("attention/W2/kernel"), a.p("attention/W2/bias"), a.p("attention/V/kernel"),
                          a.p("attention/V/bias"), self.qpre, self.alpha, self.ctx, self.ctx_d, keep,
                          B * R * A // 4 if keep is not None else 0, self.XZ, Wl[:D], a.p("lstm/recurrent_kernel"),
                          a.p("lstm/bias"), self.Hs, self.Cs, self.gates, T, B, R, D, A, U, 0.2,
                          self.r_attn if training else 0.0, r_in, D + Et, self.seed, S_ATTN, S_LSTM_IN, self.drop_step,
                          self.lc_fwd_work, self.seq_sync, self._guard_out(),
                          out_drop=(self.Hd, self.r_lstm, S_LSTM_OUT) if self._out_dropped else None)
        else:
            self._out_dropped = False
            for i in range(T):                                                                  # :244-256
                self._decode_step(i, B, training, xz_bias=a.p("lstm/bias"))
        hs = self.Hs[1:].view(n, U)
        if self._out_dropped:
            hs = self.Hd
        elif training and self.r_lstm > 0:                                                      # :256
            be.dropout(hs, self.Hd, n, U, U, 0, U, 0, self.r_lstm, sd, S_LSTM_OUT, 0, ds, rows_per_site=B)
            hs = self.Hd
        self._hs_used = hs
        self.gemm_sk(hs, a.p("time_distributed_nonlinear/kernel"), self.inter, n, H, U, U, H, H,
                bias=a.p("time_distributed_nonlinear/bias"), pre=self.ipre, act=ACT_LEAKY, slope=0.2)
        inter = self.inter
        self._metric_parts_ready = False
        if training and self.r_out > 0:
            npart = be.attention_metric_parts(T, self.R) if hasattr(be, "dropout_metric") else 0
            if (self.S == 1 and self.__dict__.get("_defer_sum2") and getattr(self, "fuse_metric_rider", True) and H % 4 == 0
                    and 0 < npart <= self.metric_part.numel()):
                # the attention metric's partials (:365-367) ride in this launch: alpha is complete behind the chain
                be.dropout_metric(self.inter, self.inter_d, n, H, H, B, H, 0, self.r_out, sd, S_OUT, 0, ds, self.alpha,
                                  self.metric_part, T, B, self.R)
                self._metric_parts_ready = True
            else:
                be.dropout(self.inter, self.inter_d, n, H, H, B, H, 0, self.r_out, sd, S_OUT, 0, ds)
            inter = self.inter_d
        self._inter_used = inter
        self.gemm_sk(inter, a.p("time_distributed_softmax/kernel"), self.logits, n, V, H, H, ldV, ldV,
                bias=a.p("time_distributed_softmax/bias"))                                     # :261

    def _loss_metrics(self, B, T, want_grad):
        be = self.be
        n = T * B
        if want_grad:
            be.softmax_cce(self.logits, self.tgt, None, self.loss_row, self.corr_row, self.logits, n, self.V, self.ldV,
                           1.0 / (n * self.dp_world))
        else:
            be.softmax_cce(self.logits, self.tgt, self.logits, self.loss_row, self.corr_row, None, n, self.V, self.ldV,
                           0.0)
        self._sum2(self.loss_row, self.met[0:1], self.corr_row, self.met[1:2], n, 1.0 / n)
        if self.S == 1:
            npart = be.attention_metric_parts(T, self.R) if hasattr(be, "attention_metric_parts") else 0
            if self.__dict__.get("_defer_sum2") and want_grad and 0 < npart <= self.metric_part.numel():
                # fused single-process step: partials only, their total rides in the step-finalize launch
                if not self.__dict__.get("_metric_parts_ready"):
                    be.attention_metric(self.alpha, None, self.metric_part, T, B, self.R)
                self._metric_deferred = (self.metric_part, self.met[3:4], npart, 1.0 / (T * self.R))
            else:
                be.attention_metric(self.alpha, self.met[3:4], self.metric_part, T, B, self.R)              # :365-367
        else:       # per-subject loss / accuracy / attention metric (ms2_NIC.py:324-372)
            S, Bs = self.S, B // self.S
            be.colsum(self.loss_row, self.colB[:B], T, B, B, self.work)
            be.colsum(self.corr_row, self.colB[B:], T, B, B, self.work)
            for q in range(S):
                k = 8 + 4 * q
                be.sum(self.colB[q * Bs:], self.met[k:k + 1], Bs, 1.0 / (Bs * T))
                be.sum(self.colB[B + q * Bs:], self.met[k + 1:k + 2], Bs, 1.0 / (Bs * T))
                be.attention_metric(self.alpha.view(-1)[q * Bs * self.R:], self.met[k + 2:k + 3], self.metric_part, T, Bs,
                                    self.R, B * self.R)

    # ------------------------------------------------------------------ backward
    def _backward(self, B, T):
        """tape.gradient (lc_NIC.py:386-387) as four launch groups, in the order the gradients become final -- the
        data-parallel schedule (dp.PipelinedAttentionSync) issues one all-reduce bucket after each."""
        self._bwd_head(B, T)
        self._bwd_chain(B, T)
        # the text branch (input Dropout', sparse Embedding scatter) and the front branch (attention parameters, BatchNorm,
        # region-wise encoder) only share the chain's outputs: with `branch_streams` they are two parallel branches of the step
        with self.side(0 if getattr(self, "branch_streams", False) else -1):
            self._bwd_emb(B, T)
        self._bwd_front(B, T)
        self.join()

    def _bwd_head(self, B, T):
        """vocabulary head: gradients of time_distributed_softmax / time_distributed_nonlinear, dHs."""
        be, a = self.be, self.arena
        U, V, H, ldV = self.U, self.V, self.H, self.ldV
        n = T * B
        sd, ds = self.seed, self.drop_step
        dlog, inter, hs = self.logits, self._inter_used, self._hs_used
        dhs_done = False
        # kernel gradient and input gradient of the softmax layer, the two independent readers of dlogits: ONE launch
        if not (getattr(self, "g3_riders", True) and self.gemm3_pair(
                dict(A=inter, B=dlog, C=a.g("time_distributed_softmax/kernel"), M=H, N=V, K=n, lda=H, ldb=ldV, ldc=ldV, transA=True),
                dict(A=dlog, B=a.p("time_distributed_softmax/kernel"), C=self.dinter, M=n, N=H, K=V, lda=ldV, ldb=ldV, ldc=H,
                     transB=True))):
            self.gemm_sk(inter, dlog, a.g("time_distributed_softmax/kernel"), H, V, n, H, ldV, ldV, transA=True)
            self.gemm_sk(dlog, a.p("time_distributed_softmax/kernel"), self.dinter, n, H, V, ldV, ldV, H, transB=True)
        if getattr(self, "fused_head_tail", True) and hasattr(be, "bias_act_drop_bwd") and n <= 2048 and H % 4 == 0:
            # dropout' + LeakyReLU' + the nonlinear layer's bias gradient in one pass over dinter, with the softmax layer's
            # bias gradient (column sums of dlogits) riding in the same launch: 1 launch instead of 4
            be.bias_act_drop_bwd(self.dinter, self.ipre, self.dinter, a.g("time_distributed_nonlinear/bias"), n, H, H,
                                 ACT_LEAKY, 0.2, B, H, 0, self.r_out, sd, S_OUT, ds,
                                 extra=(dlog, a.g("time_distributed_softmax/bias"), n, V, ldV))
            # kernel gradient and input gradient of the nonlinear layer, the two readers of dinter: ONE launch
            if getattr(self, "g3_riders", True) and self.gemm3_pair(
                    dict(A=hs, B=self.dinter, C=a.g("time_distributed_nonlinear/kernel"), M=U, N=H, K=n, lda=U, ldb=H, ldc=H,
                         transA=True, small=True),
                    dict(A=self.dinter, B=a.p("time_distributed_nonlinear/kernel"), C=self.dHs, M=n, N=U, K=H, lda=H, ldb=H, ldc=U,
                         transB=True, small=True)):
                dhs_done = True
            else:
                self.gemm_sk(hs, self.dinter, a.g("time_distributed_nonlinear/kernel"), U, H, n, U, H, H, transA=True)
        else:
            be.colsum(dlog, a.g("time_distributed_softmax/bias"), n, V, ldV, self.work)
            if self.r_out > 0:
                be.dropout(self.dinter, self.dinter, n, H, H, B, H, 0, self.r_out, sd, S_OUT, 0, ds)
            be.act_bwd(self.ipre, self.dinter, self.dinter, n * H, ACT_LEAKY, 0.2)
            self.gemm_sk(hs, self.dinter, a.g("time_distributed_nonlinear/kernel"), U, H, n, U, H, H, transA=True)
            be.colsum(self.dinter, a.g("time_distributed_nonlinear/bias"), n, H, H, self.work)
        if not dhs_done:
            self.gemm_sk(self.dinter, a.p("time_distributed_nonlinear/kernel"), self.dHs, n, U, H, H, H, U, transB=True)
        self._dout_masked = not (self.r_lstm > 0 and self._lc_seq_bwd_ok() and (int(getattr(self, "fuse_out_drop", 3)) & 2))
        if self.r_lstm > 0 and self._dout_masked:           # (else Dropout' rides in the backward chain: tnt_lc_seq_bwd_drop_f32)
            be.dropout(self.dHs, self.dHs, n, U, U, 0, U, 0, self.r_lstm, sd, S_LSTM_OUT, 0, ds, rows_per_site=B)

    def _lc_seq_bwd_ok(self):
        """the persistent backward-chain kernel applies (tnt_lc_seq_bwd_f32: the forward chain's shape limits)."""
        return bool(not self.use_layer_norm and self._lc_seq_ok() and self.__dict__.get("lc_xch") is not None
                    and getattr(self, "use_lc_seq_bwd", True))

    def _bwd_chain(self, B, T):
        """the T-step chain (LSTM step backward -> attention step backward) and the LSTM parameter gradients."""
        be, a = self.be, self.arena
        R, D, A, U, Et = self.R, self.D, self.A, self.U, self.Et
        n = T * B
        sd, ds = self.seed, self.drop_step
        # no zero fill of dP / dF / dvb: the first executed step (i = T-1) overwrites them (fresh)
        Wl, Ur = a.p("lstm/kernel"), a.p("lstm/recurrent_kernel")
        W2, v = a.p("attention/W2/kernel"), a.p("attention/V/kernel")
        if self.use_layer_norm:
            return self._bwd_chain_ln(B, T)
        if self._lc_seq_bwd_ok():
            # the T LSTM-backward -> attention-backward steps as ONE persistent launch (tnt_lc_seq_bwd_f32): role-specialised
            # workgroups per XCD, dP / dF / dvb accumulated on chip and written once
            keep = self.att_keep if self._keep_stored else None
            be.lc_seq_bwd(self.F, self.P, W2, v, self.qpre, self.alpha, keep, B * R * A // 4 if keep is not None else 0,
                          self.dP, self.dF, self.dvb, self.dqpre, Ur, Wl[:D], self.dHs, self.gates, self.Cs, self.dZ,
                          self.lc_xch, T, B, R, D, A, U, 0.2, self.r_attn, self.r_lstm, D + Et, sd, S_ATTN, S_LSTM_IN, ds,
                          self._alpha_mse, self.seq_sync, self._guard_out(),
                          out_drop=None if self.__dict__.get("_dout_masked", True) else (self.r_lstm, S_LSTM_OUT))
        else:
            self._bwd_chain_steps(B, T, Wl, Ur, W2, v)
        hprev = self.Hs[:T].view(n, U)
        gWl = a.g("lstm/kernel")
        self._dtext_done = False
        if getattr(self, "g3_riders", True) and Et == U and self.gemm3_pair(
                dict(A=hprev, B=self.dZ, C=a.g("lstm/recurrent_kernel"), M=U, N=4 * U, K=n, lda=U, ldb=4 * U, ldc=4 * U, transA=True,
                     colsum=a.g("lstm/bias"), A2=self.text, C2=gWl[D:]),
                dict(A=self.dZ, B=Wl[D:], C=self.dtext, M=n, N=Et, K=4 * U, lda=4 * U, ldb=4 * U, ldc=Et, transB=True)):
            # the four readers of dZ that fill the chip -- recurrent-kernel gradient, the text rows of the kernel gradient (+ the
            # bias gradient as a rider) and the text-input gradient (_bwd_emb finds it done) -- in ONE launch
            self._dtext_done = True
            # ... and the two small kernel gradients that only need the chain's outputs -- the context rows of the LSTM kernel
            # and the attention's W2 -- in one launch behind it
            self._dw2_done = bool(self.gemm3_pair(
                dict(A=self.ctx_d, B=self.dZ, C=gWl[:D], M=D, N=4 * U, K=n, lda=D, ldb=4 * U, ldc=4 * U, transA=True, small=True),
                dict(A=hprev, B=self.dqpre, C=a.g("attention/W2/kernel"), M=U, N=A, K=n, lda=U, ldb=A, ldc=A, transA=True,
                     small=True)))
            if not self._dw2_done:
                self.gemm_sk(self.ctx_d, self.dZ, gWl[:D], D, 4 * U, n, D, 4 * U, 4 * U, transA=True)
            return
        if getattr(self, "g3_riders", True) and Et == U and self.gemm3(
                hprev, self.dZ, a.g("lstm/recurrent_kernel"), U, 4 * U, n, U, 4 * U, 4 * U, transA=True,
                colsum=a.g("lstm/bias"), A2=self.text, C2=gWl[D:]):
            # recurrent-kernel gradient, the text rows of the kernel gradient and the bias gradient in ONE launch: the two
            # products share dZ, its column sums ride on the fragments the first tile row holds anyway
            self.gemm_sk(self.ctx_d, self.dZ, gWl[:D], D, 4 * U, n, D, 4 * U, 4 * U, transA=True)
            return
        self.gemm_sk(hprev, self.dZ, a.g("lstm/recurrent_kernel"), U, 4 * U, n, U, 4 * U, 4 * U, transA=True)
        self.gemm_sk(self.text, self.dZ, gWl[D:], Et, 4 * U, n, Et, 4 * U, 4 * U, transA=True)
        self.gemm_sk(self.ctx_d, self.dZ, gWl[:D], D, 4 * U, n, D, 4 * U, 4 * U, transA=True)
        be.colsum(self.dZ, a.g("lstm/bias"), n, 4 * U, 4 * U, self.work)

    def _bwd_chain_steps(self, B, T, Wl, Ur, W2, v):
        """the chain as 2 T per-step launches"""
        be = self.be
        R, D, A, U, Et = self.R, self.D, self.A, self.U, self.Et
        sd, ds = self.seed, self.drop_step
        for i in range(T - 1, -1, -1):
            last = i == T - 1
            # dctx_i = dZ_i @ Wc^T: every LSTM-backward workgroup leaves the partial of its 16 units, the attention
            # backward sums the U/16 parts (instead of running the whole product per sample on its own critical path)
            parts = (U // 16) * D <= 1024 and getattr(self, "ctx_parts", True)
            be.lstm_step_bwd(None if last else self.dZ[(i + 1) * B:(i + 2) * B], Ur, None,
                             None if last else self.dh_att, None if last else self.dc, None,
                             self.dHs[i * B:(i + 1) * B], None, 0, 0, self.gates[i], self.Cs[i + 1], self.Cs[i],
                             self.dZ[i * B:(i + 1) * B], None, self.dc, None, B, U,
                             Wc=Wl[:D] if parts else None, D=D, dctx_part=self.dctx_part if parts else None)
            if parts:
                be.attention_step_bwd(None, self.F, self.P, W2, v, self.qpre[i], self.alpha[i], self.dP, self.dF,
                                      self.dvb, self.dqpre[i], self.dh_att, B, R, D, A, U, 0.2, self.r_attn, self.r_lstm,
                                      D + Et, sd, S_ATTN + i, S_LSTM_IN + i, 0, ds, dctx_part=self.dctx_part,
                                      nparts=U // 16, keep4=self.att_keep[i] if self._keep_stored else None,
                                      alpha_mse=self._alpha_mse, fresh=last)
            else:
                be.attention_step_bwd(None, self.F, self.P, W2, v, self.qpre[i], self.alpha[i], self.dP, self.dF,
                                      self.dvb, self.dqpre[i], self.dh_att, B, R, D, A, U, 0.2, self.r_attn, self.r_lstm,
                                      D + Et, sd, S_ATTN + i, S_LSTM_IN + i, 0, ds, dz=self.dZ[i * B:(i + 1) * B],
                                      Wc=Wl[:D], keep4=self.att_keep[i] if self._keep_stored else None,
                                      alpha_mse=self._alpha_mse, fresh=last)

    def _bwd_chain_ln(self, B, T):
        """The T-step chain with the LayerNormLSTMCell (reverse of _decode_step's LayerNorm branch).  Per step: cell
        backward (gate math + state LayerNorm) -> input gradients of the two 4U-wide LayerNorms -> dh of the previous
        step through U^T -> attention step backward (context gradient from dZK).  The parameter gradients of the three
        LayerNorms and of W / U / b are batched over all T steps behind the chain."""
        be, a = self.be, self.arena
        R, D, A, U, Et = self.R, self.D, self.A, self.U, self.Et
        n = T * B
        sd, ds = self.seed, self.drop_step
        Wl, Ur = a.p("lstm/kernel"), a.p("lstm/recurrent_kernel")
        W2, v = a.p("attention/W2/kernel"), a.p("attention/V/kernel")
        gk, gr, gs = a.p("lstm/kernel_norm/gamma"), a.p("lstm/recurrent_norm/gamma"), a.p("lstm/state_norm/gamma")
        for i in range(T - 1, -1, -1):
            last = i == T - 1
            rows = slice(i * B, (i + 1) * B)
            be.ln_lstm_cell_bwd(self.dHs[rows], None if last else self.dh_att, None if last else self.dh_rec,
                                None if last else self.dc, self.gates[i], self.Cs[i], self.Cs[i + 1], self.chat[rows],
                                self.is_s[i], gs, self.dZ[rows], self.dc, self.dcnt[rows], B, U)
            be.layernorm_bwd(self.dZ[rows], self.xh_k[rows], gk, self.is_k[i], self.dZK[rows], None, None, B, 4 * U, 4 * U, None)
            be.layernorm_bwd(self.dZ[rows], self.xh_r[rows], gr, self.is_r[i], self.dZR[rows], None, None, B, 4 * U, 4 * U, None)
            self.gemm_sk(self.dZR[rows], Ur, self.dh_rec, B, U, 4 * U, 4 * U, 4 * U, U, transB=True)
            be.attention_step_bwd(None, self.F, self.P, W2, v, self.qpre[i], self.alpha[i], self.dP, self.dF,
                                  self.dvb, self.dqpre[i], self.dh_att, B, R, D, A, U, 0.2, self.r_attn, 0.0,
                                  D + Et, sd, S_ATTN + i, S_LSTM_IN + i, 0, ds, dz=self.dZK[rows],
                                  Wc=Wl[:D], keep4=self.att_keep[i] if self._keep_stored else None,
                                  alpha_mse=self._alpha_mse, fresh=last)
        hprev = self.Hs[:T].view(n, U)
        gWl = a.g("lstm/kernel")
        self.gemm_sk(hprev, self.dZR, a.g("lstm/recurrent_kernel"), U, 4 * U, n, U, 4 * U, 4 * U, transA=True)
        self.gemm_sk(self.text, self.dZK, gWl[D:], Et, 4 * U, n, Et, 4 * U, 4 * U, transA=True)
        self.gemm_sk(self.ctx_d, self.dZK, gWl[:D], D, 4 * U, n, D, 4 * U, 4 * U, transA=True)
        # gamma / beta of the three LayerNorms, summed over all T*B rows: dgamma = sum dy * xhat, dbeta = sum dy
        be.layernorm_bwd(self.dZ, self.xh_k, gk, self.is_k, None, a.g("lstm/kernel_norm/gamma"), a.g("lstm/kernel_norm/beta"),
                         n, 4 * U, 4 * U, self.work)
        be.layernorm_bwd(self.dZ, self.xh_r, gr, self.is_r, None, a.g("lstm/recurrent_norm/gamma"),
                         a.g("lstm/recurrent_norm/beta"), n, 4 * U, 4 * U, self.work)
        be.layernorm_bwd(self.dcnt, self.chat, gs, self.is_s, None, a.g("lstm/state_norm/gamma"), a.g("lstm/state_norm/beta"),
                         n, U, U, self.work)
        be.colsum(self.dZ, a.g("lstm/bias"), n, 4 * U, 4 * U, self.work)

    def _bwd_emb(self, B, T):
        """text branch: dtext = dZ Wl_text^T, its dropouts, the embedding scatter (+ IndexedSlices norm)."""
        be, a = self.be, self.arena
        D, U, Et, V = self.D, self.U, self.Et, self.V
        n = T * B
        sd, ds = self.seed, self.drop_step
        Wl = a.p("lstm/kernel")
        if not self.__dict__.pop("_dtext_done", False):
            self.gemm_sk(self.dZK if self.use_layer_norm else self.dZ, Wl[D:], self.dtext, n, Et, 4 * U, 4 * U, 4 * U, Et, transB=True)
        lstm_in = self.r_lstm > 0 and not self.use_layer_norm
        if (lstm_in and self.r_text > 0 and Et % 4 == 0 and D % 4 == 0 and hasattr(be, "dropout2")
                and getattr(self, "fused_text_masks", True)):
            # the LSTM input mask and the Embedding Dropout of the text rows in one pass
            be.dropout2(self.dtext, self.dtext, n, Et, Et, (0, D + Et, D, B, self.r_lstm, S_LSTM_IN),
                        (B, Et, 0, 0, self.r_text, S_TEXT), sd, 0, ds)
        else:
            if lstm_in:
                be.dropout(self.dtext, self.dtext, n, Et, Et, 0, D + Et, D, self.r_lstm, sd, S_LSTM_IN, 0, ds,
                           rows_per_site=B)
            if self.r_text > 0:
                be.dropout(self.dtext, self.dtext, n, Et, Et, B, Et, 0, self.r_text, sd, S_TEXT, 0, ds)
        self._emb_rows = (self.dtext, n, Et, Et, "emb_text/embeddings")
        self._embedding_bwd(self.dtext, self.cap, "emb_text/embeddings", B, T, Et, Et, V)

    def _bwd_front(self, B, T):
        """attention parameters, then BatchNorm / region-wise encoder backward."""
        be, a = self.be, self.arena
        R, D, A, U = self.R, self.D, self.A, self.U
        n = T * B
        sd, ds = self.seed, self.drop_step
        hprev = self.Hs[:T].view(n, U)
        # attention parameters
        if not self.__dict__.pop("_dw2_done", False):
            self.gemm_sk(hprev, self.dqpre, a.g("attention/W2/kernel"), U, A, n, U, A, A, transA=True)
        jobs = [(self.dqpre, a.g("attention/W2/bias"), n, A, A), (self.dvb, a.g("attention/V/kernel"), B, A, A + 1),
                (self.dvb.view(-1)[A:], a.g("attention/V/bias"), B, 1, A + 1)]
        if hasattr(be, "colsum_multi") and n <= 2048 and B <= 2048:
            be.colsum_multi(jobs)                     # the three small bias-type gradients of the attention layer: one launch
        else:
            for x, out, rows, C, ld in jobs:
                be.colsum(x, out, rows, C, ld, self.work)
        fdrop = None                                   # (rate, seed, site, step_dev) of the Dropout' folded into the dF pass
        if getattr(self, "fused_att_front", True) and hasattr(be, "attention_front_bwd") and D == 32 and A == 32:
            # LeakyReLU' + bias gradient + W1 gradient + the dF contribution of the hoisted Dense in two launches instead of five
            fb = self.__dict__.get("_att_fb")
            if fb is None or fb[1] != (B * R, D, A):
                if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("attention front-backward scratch must be built outside a graph capture")
                fb = self._att_fb = (self._f(be.attention_front_bwd_parts(B * R, D, A)), (B * R, D, A))
            # dF is finished by this launch: the backward of the LAST feature Dropout (one site over all B*R rows: the
            # deepest stage's, or the single-subject encoder's) rides in its dF pass
            if self.r_feat > 0 and getattr(self, "fused_bn_drop", True) and (self.depth > 0 or self.S == 1):
                fdrop = (self.r_feat, sd, S_DEEP + self.depth - 1 if self.depth > 0 else S_FEAT, ds)
            be.attention_front_bwd(self.Ppre, self.dP, self.F, a.p("attention/W1/kernel"), self.dF,
                                   a.g("attention/W1/kernel"), a.g("attention/W1/bias"), fb[0], B * R, D, A, 0.2, drop=fdrop)
        else:
            be.act_bwd(self.Ppre, self.dP, self.dP, B * R * A, ACT_LEAKY, 0.2)
            self.gemm_sk(self.F, self.dP, a.g("attention/W1/kernel"), D, A, B * R, D, A, A, transA=True)
            be.colsum(self.dP, a.g("attention/W1/bias"), B * R, A, A, self.work)
            self.gemm_sk(self.dP, a.p("attention/W1/kernel"), self.dF, B * R, D, A, A, A, D, transB=True, accumulate=True)
        # encoder
        S = self.S
        Bs = B // S
        dF_enc = self.dF
        for i in range(self.depth - 1, -1, -1):          # the deep stages, last first; dF_enc: gradient wrt Fs[i + 1]
            bn = f"input_bn/deep{i}"
            if self.r_feat > 0 and not (fdrop is not None and i == self.depth - 1):
                be.dropout(dF_enc, dF_enc, B * R, D, D, 0, D, 0, self.r_feat, sd, S_DEEP + i, 0, ds)
            acted = False
            if self.norm == "batch":
                acted = self._bn_bwd(dF_enc, self.deep_xhat[i], a.p(f"{bn}/gamma"), self.deep_istd[i], self.dbn,
                                     a.g(f"{bn}/gamma"), a.g(f"{bn}/beta"), B * R, D, D, self.work, act_pre=self.deep_pre[i])
            else:
                be.layernorm_bwd(dF_enc, self.deep_xhat[i], a.p(f"{bn}/gamma"), self.deep_istd[i], self.dbn,
                                 a.g(f"{bn}/gamma"), a.g(f"{bn}/beta"), B * R, D, D, self.work)
            if not acted:
                be.act_bwd(self.deep_pre[i], self.dbn, self.dbn, B * R * D, ACT_LEAKY, 0.2)
            be.locally_dense_bwd(self.Fs[i], R * D, self.deep_idx, self.deep_goff, self.dbn, self.deepWg[i], self.deepBg[i],
                                 B, R, D)
            be.block_dense_dx(self.dbn, self.deepW[i], self.dFd, B, R, D, D)
            dF_enc = self.dFd
        for q in range(S):
            off = SUBJ_SITE * (q + 1) if S > 1 else 0
            r0, r1 = q * Bs, (q + 1) * Bs
            dFq, xh, dbn = dF_enc[r0:r1], self.xhat[r0 * R:r1 * R], self.dbn[r0 * R:r1 * R]
            if self.r_feat > 0:
                if S > 1:
                    be.dropout(dFq, dFq, Bs * R, D, D, 0, D, 0, self.r_feat, sd, S_FEAT2 + off, 0, ds)
                if not (fdrop is not None and self.depth == 0):
                    be.dropout(dFq, dFq, Bs * R, D, D, 0, D, 0, self.r_feat, sd, S_FEAT + off, 0, ds)
            bn = self.bn_name(q)
            acted = False
            if self.norm == "batch":
                acted = self._bn_bwd(dFq, xh, a.p(f"{bn}/gamma"), self.inv_std[q], dbn, a.g(f"{bn}/gamma"),
                                     a.g(f"{bn}/beta"), Bs * R, D, D, self.work, act_pre=self.enc_pre[r0:r1])
            else:
                be.layernorm_bwd(dFq, xh, a.p(f"{bn}/gamma"), self.inv_std[q], dbn, a.g(f"{bn}/gamma"),
                                 a.g(f"{bn}/beta"), Bs * R, D, D, self.work)
            if not acted:
                be.act_bwd(self.enc_pre[r0:r1], dbn, dbn, Bs * R * D, ACT_LEAKY, 0.2)
            x = (self.xd if self.r_in > 0 else self.x)[r0:r1]
            if self.NV > R and hasattr(be, "locally_dense_bwd_split") and getattr(self, "split_encoder", True):
                vm = self.xT is not None and getattr(self, "voxel_major", True)
                be.locally_dense_bwd_split(self.xT[:, r0:r1] if vm else x, self.xT.shape[1] if vm else self.ldx, self.idx,
                                           self.vgoff, self.vreg, self.vfirst, self.NV, dbn, self.encWg[q], self.encBg[q],
                                           Bs, R, D, voxel_major=vm)
            else:
                be.locally_dense_bwd(x, self.ldx, self.idx, self.goff, dbn, self.encWg[q], self.encBg[q], Bs, R, D)

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

    def _stage_mask_job(self):
        if not (self._keep_stored and getattr(self, "stage_masks", True)):
            return None
        B, T = self._shape
        return (self.att_keep, B * self.R * self.A, T, self.r_attn, self.seed, S_ATTN, self.drop_step)

    def _metrics(self, with_lr, ring=False):
        m = self._met_snapshot(ring)
        if self.S == 1:
            out = Metrics(loss=m[0], L2=m[2], accuracy=m[1], attention=m[3])
        else:       # keys of ms2_NIC.train_step's return dict (ms2_NIC.py:366-374), A, B, C ... per subject
            out = Metrics(loss=m[0], L2=m[2])
            for q in range(self.S):
                tag, k = chr(ord("A") + q), 8 + 4 * q
                out[f"loss{tag}"], out[f"accuracy{tag}"], out[f"attention{tag}"] = m[k], m[k + 1], m[k + 2]
        if with_lr:
            out["lr"] = torch.tensor(self._lr_host, dtype=torch.float32)      # host-set (ModelBase._sync_lr): no device copy
        out._ring = self.__dict__.get("_last_ring")
        return out.guarded(self, m[self.GUARD]) if self.__dict__.get("_seq_lstm") else out

    def train_step(self, data):
        """lc_NIC.train_step (lc_NIC.py:328-408): returns {loss, L2, accuracy, attention, lr}."""
        if self.optimizer is None:
            raise RuntimeError("compile() the model before train_step")
        B, T = self._stage_batch(data[0], data[1], self.n_in, masks=True)
        self._masks_staged = self._stage_mask_job() is not None
        self._sync_lr()
        ring = False
        if self.grad_sync is None:
            # The step is replayed as a recorded launch plan (ModelBase._run_planned), not as a hipGraph: with 15 (dense) / 32
            # (attention) launches a step the host re-issues them in ~15 % of the step's time and every launch starts ~0.2-0.4 us
            # earlier than as a graph node (0.4650 -> 0.4588 and 0.5666 -> 0.5626 ms/step, tools/probe/plan_bench.py: separate
            # models, repeated, spread 0.0005).  ``plan_step = False`` restores the graph.  A plan re-issues backend launches
            # only, so it is used where the step is nothing else: the sparse Embedding backward (the dense form hands its ids on
            # with a tensor copy, which a graph captures and a plan would drop).
            plan = (getattr(self, "plan_step", True) and self.Et % 4 == 0 and getattr(self, "sparse_emb_bwd", True)
                    and hasattr(self.be, "embedding_bwd_sparse"))
            run = self._run_planned if plan else self._run_captured
            ring = self._run_step(run, ("train", B, T), lambda: self._train_and_update_graph(B, T))
        elif getattr(self.grad_sync, "pipelined", False):
            self.grad_sync.step(self, B, T)
        else:
            self._run_captured(("train_fb", B, T), lambda: self._train_graph(B, T))
            self.grad_sync(self)
            self._run_captured(("train_up", B, T), self._update_graph)
        self.optimizer.iterations += 1
        return self._metrics(True, ring)

    _alpha_mse = 0.0        # coefficient of the attention-MSE gradient in the backward chain (train_step_sam's first pass)

    def train_step_sam(self, data, rho=0.05):
        """lc_NIC.train_step_sam (lc_NIC.py:713-838): sharpness-aware step.  Pass 1 differentiates
        CE + L2 + MSE(ones, attention_scores) (:751-766), the weights move by e_w = g * rho / (global_norm(g) + 1e-12)
        (:768-786; the Embedding's IndexedSlices enter the norm by their un-deduplicated values), pass 2 differentiates
        CE + L2 there (:800-830), the weights are restored and the optimizer applies the second gradient (:833-836).
        Returns {loss, L2, accuracy, attention = MSE(ones, alpha), lr} of the second pass (:838).  Both passes draw the
        same dropout masks.  One captured launch sequence: two forward / backward passes and the update."""
        if self.optimizer is None:
            raise RuntimeError("compile() the model before train_step_sam")
        if self.S != 1:
            raise NotImplementedError("train_step_sam is a single-subject step (lc_NIC.py)")
        B, T = self._stage_batch(data[0], data[1], self.n_in, masks=True)
        self._masks_staged = self._stage_mask_job() is not None
        self._sync_lr()
        be, a, sp = self.be, self.arena, self.arena.spans
        if self.__dict__.get("ew") is None:
            self.ew = torch.zeros_like(a.theta)
        n_alpha = T * B * self.R

        def run():
            self._forward(B, T, True)
            self._loss_metrics(B, T, True)
            self._alpha_mse = 2.0 / n_alpha
            try:
                self._backward(B, T)
            finally:
                self._alpha_mse = 0.0
            self._norms_and_l2(None)
            be.sam(a.theta, a.grad, self.ew, sp.span_seg, sp.span_off, sp.span_len, a.seg_l2, a.sq, a.nseg, sp.nspan, rho, 0,
                   sq_override=a.sq_override)
            self._forward(B, T, True)
            self._loss_metrics(B, T, True)
            be.sqdiff_mean(self.alpha, self.met[3:4], n_alpha, 1.0)
            self._backward(B, T)
            self._norms_and_l2(self.met[2:3])          # L2 as the reference reports it: at the perturbed weights
            be.sam(a.theta, a.grad, self.ew, sp.span_seg, sp.span_off, sp.span_len, a.seg_l2, a.sq, a.nseg, sp.nspan, rho, 1)
            self._apply_agc()
            self._norms_and_l2(None)                   # clip norms of the second gradient at the restored weights
            self._apply_optimizer()
        self._run_captured(("sam", B, T, float(rho)), run)
        self.optimizer.iterations += 1
        return self._metrics(True)

    def test_step(self, data):
        """lc_NIC.test_step (lc_NIC.py:410-459)."""
        B, T = self._stage_batch(data[0], data[1], self.n_in)
        self._masks_staged = False

        # ms2_NIC.test_step calls its sub-models with training=True (ms2_NIC.py:419,426) -- quirk kept
        train_flag = self.S > 1

        def run():
            self._forward(B, T, train_flag)
            self._loss_metrics(B, T, False)
            self._norms_and_l2(self.met[2:3])
            if train_flag:
                self.be.step_tick(None, self.drop_step, None, None, 0.0, 0.0)
        self._run_captured(("test", B, T), run)
        return self._metrics(False)

    def __call__(self, data, training=False):
        """lc_NIC.call -> call_attention (lc_NIC.py:163-164,223-263):
        returns (probabilities (B,T,V), attention scores (T,B,R,1))."""
        B, T = self._stage_inputs(data)
        self._masks_staged = False

        def run():
            self._forward(B, T, training)
            self.be.softmax_cce(self.logits, None, self.logits, None, None, None, T * B, self.V, self.ldV, 0.0)
            probs = self.logits.view(T, B, self.ldV)[:, :, :self.V].permute(1, 0, 2).contiguous()
            return probs, self.alpha.clone().unsqueeze(-1)
        return self._guarded(run)

    call = call_attention = __call__

    def sample_predict(self, img_input, a0, c0, start_seq, max_len, units=None, tokenizer=None, temperature=1.0,
                       sample_step=0):
        """greedy_predict_attention with the argmax replaced by lc_NIC.sample_choice (lc_NIC.py:571-575:
        tf.random.categorical(log(probs), 1)); ``temperature`` as in ThinkAndTell/evaluate.py:223.  TF's
        sampler cannot be reproduced; the draw is the Philox stream (seed, S_SAMPLE + position, sample_step),
        restated by oracle.ops.sample_rows.  Same return tuple as greedy_predict."""
        return self.greedy_predict(img_input, a0, c0, start_seq, max_len, units, tokenizer,
                                   _sample=(float(temperature), int(sample_step)))

    def greedy_predict(self, img_input, a0, c0, start_seq, max_len, units=None, tokenizer=None, training=False,
                       _sample=None, return_s=True):
        """lc_NIC.greedy_predict -> greedy_predict_attention (lc_NIC.py:507-508,577-638).
        Returns (words (B,max_len,1) int64, probs (B,max_len,V), alpha (max_len,B,R,1), s (max_len,B,R,A))
        as numpy arrays; the whole decode runs on the device with no per-step host sync."""
        assert training is False, "training is set to True"                                  # lc_NIC.py:591
        be, a = self.be, self.arena
        start = self._to_dev(np.asarray(start_seq).reshape(-1), torch.int32)
        B = start.shape[0]
        self._stage_inputs((img_input, torch.zeros(B, max_len, dtype=torch.int32), a0, c0))
        R, D, A, U, Et, V, H, ldV = self.R, self.D, self.A, self.U, self.Et, self.V, self.H, self.ldV
        Wl = a.p("lstm/kernel")
        # decode buffers are static per (B, max_len, return_s) so that the whole loop can be one captured hipGraph
        # (encoder + max_len x {embed, project, attention, LSTM step, head, softmax, argmax}: ~8 launches per token)
        key = (B, max_len, bool(return_s))
        bufs = self.__dict__.setdefault("_dec_bufs", {})
        if key not in bufs:
            bufs[key] = (torch.zeros(B, 1, dtype=torch.int32, device=self.device),
                         torch.zeros(max_len, B, ldV, dtype=torch.float32, device=self.device),
                         torch.zeros(max_len, B, R, A, dtype=torch.float32, device=self.device) if return_s else None,
                         torch.zeros(max_len, B, dtype=torch.int32, device=self.device))
        start_buf, probs, s_all, ids = bufs[key]
        start_buf.copy_(start.view(B, 1))

        def run():
            self._encode(B, False)
            words = start_buf
            for i in range(max_len):
                text = self.text[i * B:(i + 1) * B]
                be.embedding_fwd(a.p("emb_text/embeddings"), words, text, B, 1, Et, Et, V)        # :596,632
                self.gemm_sk(text, Wl[D:], self.XZ[i * B:(i + 1) * B], B, 4 * U, Et, Et, 4 * U, 4 * U,
                             bias=None if self.use_layer_norm else a.p("lstm/bias"))      # LN cell: bias behind the norms
                self._decode_step(i, B, False, s_all[i] if return_s else None)
                self.gemm_sk(self.Hs[i + 1], a.p("time_distributed_nonlinear/kernel"), self.inter[:B], B, H, U, U, H, H,
                             bias=a.p("time_distributed_nonlinear/bias"), act=ACT_LEAKY, slope=0.2)     # :621
                self.gemm_sk(self.inter[:B], a.p("time_distributed_softmax/kernel"), probs[i], B, V, H, H, ldV, ldV,
                             bias=a.p("time_distributed_softmax/bias"))                                # :623
                be.softmax_cce(probs[i], None, probs[i], None, None, None, B, V, ldV, 0.0)
                if _sample is None:
                    be.argmax_rows(probs[i], ids[i], B, V, ldV)                                    # :627
                else:
                    be.sample_rows(probs[i], ids[i], B, V, ldV, _sample[0], False, self.seed, S_SAMPLE + i, _sample[1])
                words = ids[i].view(B, 1)
        if _sample is None:
            self._run_captured(("greedy",) + key, run)
        else:                      # the sampling stream step is a launch argument: not captured
            run()
        out_words = ids.t().contiguous().cpu().numpy().astype(np.int64)[:, :, None]
        out_probs = probs[:, :, :V].permute(1, 0, 2).contiguous().cpu().numpy()
        # s (the dropout-free tanh activations, 44 MB at the BASELINE shape) is part of the reference's return tuple but
        # only analysis code reads it: return_s=False skips its device-to-host copy (eval_model does)
        return out_words, out_probs, self.alpha[:max_len].cpu().numpy()[..., None], (s_all.cpu().numpy() if return_s else None)

    greedy_predict_attention = greedy_predict

    def beam_search(self, img_input, a0, c0, start_seq, max_len, beam_width=5, end_id=-1, units=None, tokenizer=None):
        """Beam search over the attention decoder.  The reference only sketches it (lc_NIC.beam_search / _beam_search,
        lc_NIC.py:640-692, recurse without returning; ThinkAndTell/evaluate.py:203-228 stops after one expansion), so
        the definition is this library's: standard log-probability beam search of width ``beam_width`` with the greedy
        decoder's step (lc_NIC.py:596-632), no length normalisation; a beam that emits ``end_id`` (the tokenizer's
        '<end>' index; -1 = never) is finished and pads with 0.  Returns (sequences (B, k, max_len) int64, best first;
        scores (B, k) float32 = sum of log-probabilities).  The whole search runs on the device: per token one
        expansion launch (tnt_beam_topk_f32) and row gathers of the LSTM state by parent beam; the paths are
        back-tracked on the host at the end.  Restated by oracle.models.LcNIC.beam_search."""
        be, a = self.be, self.arena
        k = int(beam_width)
        start = np.asarray(start_seq).reshape(-1)
        B = start.shape[0]
        Bk = B * k
        rep = lambda t: np.repeat(np.asarray(t), k, axis=0)
        x = img_input.cpu().numpy() if isinstance(img_input, torch.Tensor) else np.asarray(img_input)
        self._stage_inputs((rep(x), torch.zeros(Bk, max_len, dtype=torch.int32), rep(np.asarray(a0)), rep(np.asarray(c0))))
        R, D, A, U, Et, V, H, ldV = self.R, self.D, self.A, self.U, self.Et, self.V, self.H, self.ldV
        Wl = a.p("lstm/kernel")
        dev, i32 = self.device, torch.int32
        words0 = torch.as_tensor(rep(start).astype(np.int32)).to(dev).view(Bk, 1)
        score = [torch.zeros(Bk, device=dev), torch.zeros(Bk, device=dev)]
        score[0].view(B, k)[:, 1:] = -1e30            # step 0: the k beams of a sample are copies, only beam 0 counts
        fin = [torch.zeros(Bk, dtype=i32, device=dev), torch.zeros(Bk, dtype=i32, device=dev)]
        parents = torch.zeros(max_len, Bk, dtype=i32, device=dev)
        tokens = torch.zeros(max_len, Bk, dtype=i32, device=dev)
        probs = self.logits[:Bk]
        hg, cg = self._f(Bk, U), self._f(Bk, U)
        self._encode(Bk, False)
        words = words0
        for i in range(max_len):
            text = self.text[i * Bk:(i + 1) * Bk]
            be.embedding_fwd(a.p("emb_text/embeddings"), words, text, Bk, 1, Et, Et, V)
            self.gemm_sk(text, Wl[D:], self.XZ[i * Bk:(i + 1) * Bk], Bk, 4 * U, Et, Et, 4 * U, 4 * U,
                         bias=None if self.use_layer_norm else a.p("lstm/bias"))
            self._decode_step(i, Bk, False, None)
            self.gemm_sk(self.Hs[i + 1], a.p("time_distributed_nonlinear/kernel"), self.inter[:Bk], Bk, H, U, U, H, H,
                         bias=a.p("time_distributed_nonlinear/bias"), act=ACT_LEAKY, slope=0.2)
            self.gemm_sk(self.inter[:Bk], a.p("time_distributed_softmax/kernel"), probs, Bk, V, H, H, ldV, ldV,
                         bias=a.p("time_distributed_softmax/bias"))
            be.softmax_cce(probs, None, probs, None, None, None, Bk, V, ldV, 0.0)
            cur, nxt = i & 1, (i & 1) ^ 1
            be.beam_topk(probs, score[cur], fin[cur], B, V, ldV, k, int(end_id), score[nxt], parents[i], tokens[i], fin[nxt])
            # the surviving beams continue from their parents' LSTM state (row gather by parent)
            be.embedding_fwd(self.Hs[i + 1], parents[i].view(Bk, 1), hg, Bk, 1, U, U, Bk)
            be.embedding_fwd(self.Cs[i + 1], parents[i].view(Bk, 1), cg, Bk, 1, U, U, Bk)
            self.Hs[i + 1].copy_(hg)
            self.Cs[i + 1].copy_(cg)
            words = tokens[i].view(Bk, 1)
        final = score[max_len & 1].cpu().numpy().reshape(B, k)
        par, tok = parents.cpu().numpy(), tokens.cpu().numpy()
        seqs = np.zeros((B, k, max_len), np.int64)
        for b in range(B):
            for r in range(k):
                row = b * k + r
                for i in range(max_len - 1, -1, -1):
                    seqs[b, r, i] = tok[i, row]
                    row = par[i, row]
        return seqs, final
