"""Keras-style model surface shared by the NIC variants.

Mirrors what AttemptFour/main.py and eval.py call on a ``tf.keras.Model``
(SURVEY.md 8b): ``compile``, ``__call__``, ``fit``, ``train_step``, ``test_step``,
``save_weights`` / ``load_weights(by_name, skip_mismatch)``, ``summary``,
``get_layer(name).get_weights/set_weights``, ``optimizer.lr``, ``trainable_variables``,
``losses`` -- on top of the HIP kernel backend.  Host code here only sequences kernel
launches and owns buffers; it contains no arithmetic of the hot path.
"""
import time
from collections import OrderedDict
from contextlib import contextmanager

import numpy as np
import torch

from . import ops
from .optimizers import Adam

# dropout site ids of the Philox stream (shared with oracle/models.py)
S_IN, S_FEAT, S_TEXT, S_OUT = 1, 2, 3, 5
S_ATTN, S_LSTM_IN, S_LSTM_OUT = 16, 48, 80
S_SAMPLE = 112          # + decode position: categorical-sampling stream of sample_predict
BN_EPS, BN_MOMENTUM = 1e-3, 0.99


def interleave_gates(w, U):
    """keras [.., 4U] (i,f,c~,o blocks) -> kernel layout [.., U, 4]."""
    w = np.asarray(w)
    return np.ascontiguousarray(np.moveaxis(w.reshape(*w.shape[:-1], 4, U), -2, -1))


def deinterleave_gates(w):
    """kernel layout [.., U, 4] -> keras [.., 4U]."""
    w = np.asarray(w)
    return np.ascontiguousarray(np.moveaxis(w, -1, -2)).reshape(*w.shape[:-2], -1)


class _LayerView:
    """What ``model.get_layer(name)`` returns: get_weights/set_weights in keras layouts
    (main.py:161-162 warm-starts 'lstm' and 'time_distributed_softmax' this way)."""

    def __init__(self, model, name, weight_names):
        self.model, self.name, self.weight_names = model, name, weight_names

    def get_weights(self):
        return [self.model.get_weight(f"{self.name}/{w}") for w in self.weight_names]

    def set_weights(self, weights):
        assert len(weights) == len(self.weight_names), "weight list length mismatch"
        for w, arr in zip(self.weight_names, weights):
            self.model.set_weight(f"{self.name}/{w}", arr)


class DeviceGuardError(RuntimeError):
    """A device-side guard tripped (today: the barrier / census guard of the persistent LSTM kernel).  The step that
    carried it left the model state untouched (the optimizer kernels skip on the error word) and the model has already
    fallen back to the per-step kernels, so the caller may simply run the step again -- ``fit`` does."""


class Metrics(dict):
    """train_step/test_step result: values are 0-d device tensors (no host sync until read).  The step's device guard
    word (ModelBase.GUARD slot of ``met``, copied in the same clone as the metrics) rides along: reading the metrics
    raises DeviceGuardError when it is set, so an invalid step cannot go unnoticed -- and costs no extra sync."""
    _guard = _model = _ring = None

    def guarded(self, model, word):
        self._model, self._guard = model, word
        return self

    def as_floats(self):
        out = {k: float(v) for k, v in self.items()}
        if self._ring is not None:        # values are views of a metrics-ring row (ModelBase._met_snapshot): still this step's?
            row, tag = self._ring
            if int(float(row[-1])) != tag:
                raise RuntimeError(f"these metrics were read more than {ModelBase.METRIC_RING} training steps after their step: "
                                   "the ring row has been reused (read metrics earlier, or set model.metric_ring = False)")
        if self._guard is not None and float(self._guard) != 0.0:
            self._model._on_guard_trip(int(float(self._guard)))
        return out


class ModelBase:
    GUARD = 7       # slot of ``met`` that carries the device guard word of the step (see Metrics)
    METRIC_RING = 1024      # rows of the metrics ring: a step's Metrics stay readable for this many further training steps
    # subclasses fill: self.layers_spec = OrderedDict(layer -> [weight names]),
    # self.keras_shapes = {full name: keras shape}
    def __init__(self, device=None, seed=42, use_graph=True, grad_sync=None):
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.seed = int(seed)
        self.use_graph = bool(use_graph)
        self.grad_sync = grad_sync          # callable(model) -> None: data-parallel all-reduce hook (dp.py)
        self.dp_world = int(getattr(grad_sync, "world", 1)) if grad_sync is not None else 1
        self.optimizer = None
        self.loss = None
        self.built = False
        self.stop_training = False
        self._graphs = {}
        self._lr_host = None

    # ------------------------------------------------------------------ keras surface
    def compile(self, optimizer=None, loss=None, *metrics, run_eagerly=True, **kw):
        """model.compile(optimizer, loss_object, run_eagerly=True) -- main.py:134."""
        self.optimizer = optimizer if optimizer is not None else Adam()
        self.loss = loss
        if self.built:
            self._init_optimizer_state()

    @property
    def be(self):
        return ops.backend()

    def _f(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def _init_optimizer_state(self):
        a = self.arena
        self.opt_m = torch.zeros_like(a.theta)
        self.opt_v = torch.zeros_like(a.theta) if self.optimizer.kind == "adam" else None
        self.adam_t = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.lr_dev = torch.tensor([self.optimizer.lr], dtype=torch.float32, device=self.device)
        self.lr_t_dev = self._f(1)
        self._lr_host = self.optimizer.lr
        self._graphs = {}

    def _sync_lr(self):
        if self.optimizer.lr != self._lr_host:
            self.lr_dev.fill_(self.optimizer.lr)
            self._lr_host = self.optimizer.lr

    def _apply_optimizer(self):
        """per-variable clipnorm + Adam/SGD (optimizer.apply_gradients, lc_NIC.py:389)."""
        be, a, sp, opt = self.be, self.arena, self.arena.spans, self.optimizer
        clip = opt.clipnorm if opt.clipnorm is not None else 0.0
        gd = self._guard_word()
        if opt.kind == "adam":
            be.step_tick(self.adam_t, self.drop_step, self.lr_dev, self.lr_t_dev, opt.beta_1, opt.beta_2, guard=gd)
            be.adam(a.theta, self.opt_m, self.opt_v, a.grad, sp.span_seg, sp.span_off, sp.span_len, a.seg_l2, a.sq,
                    a.sq_override, sp.nspan, 0.0, self.lr_t_dev, opt.beta_1, opt.beta_2, opt.epsilon, clip, guard=gd)
        else:
            be.step_tick(self.adam_t, self.drop_step, self.lr_dev, None, 0.0, 0.0, guard=gd)
            be.sgd(a.theta, self.opt_m, a.grad, sp.span_seg, sp.span_off, sp.span_len, a.seg_l2, a.sq, a.sq_override,
                   sp.nspan, 0.0, self.lr_dev, opt.momentum, clip, guard=gd)

    def _update_fused(self, l2_out, skip_first=False):
        """single-process update: [AGC] -> span norms -> ONE finalize launch (per-variable norms, L2 metric, the step's
        loss / accuracy totals if the model deferred them, step tick) -> clip + Adam / SGD.  Replaces the five dependent
        launches seg_finalize, l2_total, sum2, step_tick of the unfused sequence (each ~4.6 us inside the graph).
        ``skip_first`` (dp.PipelinedDenseSync with a row-sharded encoder kernel): variable 0 is updated by the caller, its
        (sum g^2, sum theta^2) pair sits in partial[0:2] with the variable's other slots zero -- the finalize work files it
        like any other, the norm / update launches here start behind it."""
        be, a, sp, opt = self.be, self.arena, self.arena.spans, self.optimizer
        self._apply_agc()
        d = self.__dict__.pop("_sum2_deferred", None)
        if not hasattr(be, "step_finalize"):
            if d is not None:
                be.sum2(*d)
            md = self.__dict__.pop("_metric_deferred", None)
            if md is not None:
                be.sum(md[0], md[1], md[2], md[3])
            self._norms_and_l2(l2_out)
            self._apply_optimizer()
            return
        clip = opt.clipnorm if opt.clipnorm is not None else 0.0
        gd = self._guard_word()
        adam = opt.kind == "adam"
        enc = self.__dict__.pop("_enc_fused", None)
        # no finalize launch (tnt_adam_fin_f32): the norm launch leaves lr_t, the update launches sum the clip norms they need
        # from the span partials themselves, one extra workgroup of the Adam launch files the scalars, the counters tick at its end
        fin = adam and hasattr(be, "adam_fin") and getattr(self, "fused_finalize", True)
        lr_job = (self.adam_t, self.lr_dev, self.lr_t_dev, opt.beta_1, opt.beta_2) if fin else None
        # the norm launch reads only what the update will consume (tnt_span_norm, csrc/tnt_fin.h): no gradient pass for a
        # variable whose clip norm is supplied through sq_override, no theta pass where there is no regulariser
        skip = a.sq_override if (fin and not self.__dict__.get("agc") and getattr(self, "norm_skip", True)) else None
        s1, rest_done = 0, False
        if skip_first:
            if enc is not None or not adam:
                raise RuntimeError("skip_first needs Adam and an encoder gradient that the caller handles")
            s1 = sp.first_host[1]
        if enc is not None:
            # The dense encoder kernel (segment 0, 59 % of config 2's parameters) never has its gradient written: one
            # pass of the skinny product leaves its norm partials in the variable's span slots, a second one applies
            # clip + Adam to the strips as they leave the MFMAs (24 bytes per parameter instead of 40).
            name, x, dpre, rows, N, E, ldx, gram = enc
            e = a.entries[name]
            s1 = sp.first_host[e.seg + 1]
            if gram is not None and 4 * rows + gram[5] <= s1:
                # norm from the forward's by-products: ||X^T D||^2 = sum (X X^T) o (D D^T), no pass over the gradient
                pre, bias, gx, nsplit, w2, nw2 = gram
                # ... with the span norms of every other variable riding in the same launch
                be.dense_gram_norm(dpre, pre, bias, gx, nsplit, w2, nw2, e.l2, a.partial, s1, rows, E,
                                   spans=(a.theta, a.grad, sp.span_seg[s1:], sp.span_off[s1:], sp.span_len[s1:], a.seg_l2,
                                          a.partial[2 * s1:], sp.nspan - s1), lr_job=lr_job, **({"skip": skip} if skip is not None else {}))
                rest_done = True
            else:
                be.dense_dw_sqnorm(x, dpre, a.p(name), e.l2, a.partial, s1, N, E, rows, ldx)
        if not rest_done:
            if fin and sp.nspan - s1 > 0:
                be.span_sqnorm_lr(a.theta, a.grad, sp.span_seg[s1:], sp.span_off[s1:], sp.span_len[s1:], a.seg_l2, a.partial[2 * s1:],
                                  sp.nspan - s1, *lr_job, **({"skip": skip} if skip is not None else {}))
            else:
                fin = False
                be.span_sqnorm(a.theta, a.grad, sp.span_seg[s1:], sp.span_off[s1:], sp.span_len[s1:], a.seg_l2, a.partial[2 * s1:],
                               sp.nspan - s1)
        kw = dict(x0=d[0], out0=d[1], x1=d[2], out1=d[3], n=d[4], scale=d[5]) if d is not None else {}
        md = self.__dict__.pop("_metric_deferred", None)
        if md is not None:
            kw.update(x2=md[0], out2=md[1], n2=md[2], scale2=md[3])
        ef = self.__dict__.pop("_emb_finalize", None)
        if ef is not None:       # sparse embedding backward: sum its norm partials, hand this step's ids on as prev_ids
            parts, sqo, nparts, ids, prev, nids = ef
            kw.update(ids_src=ids, ids_dst=prev, n_ids=nids)
            if sqo is not None:
                kw.update(extra_part=parts, extra=sqo, n_extra=nparts)
        if fin:
            if enc is not None:        # the encoder kernel first: it reads the step counter's lr_t like the Adam launch, which ticks
                sl = slice(e.off, e.off + e.size)
                be.dense_dw_adam_fin(x, dpre, a.theta[sl], self.opt_m[sl], self.opt_v[sl], e.l2, a.partial, sp.first_host[e.seg],
                                     sp.first_host[e.seg + 1], a.sq_override[e.seg:e.seg + 1], self.lr_t_dev, opt.beta_1,
                                     opt.beta_2, opt.epsilon, clip, N, E, rows, ldx, guard=gd)
            if "extra_part" in kw:
                kw["extra_seg"] = self.emb_seg
            arrive = self.__dict__.get("_fin_arrive")
            if arrive is None:
                arrive = self._fin_arrive = torch.zeros(16, dtype=torch.int32, device=self.device)
            # one descriptor per distinct argument set, alive as long as the model (recorded launch plans re-issue the call)
            ck = tuple((k, v.data_ptr() if torch.is_tensor(v) else v) for k, v in sorted(kw.items())) + (l2_out.data_ptr(), gd is not None)
            descs = self.__dict__.setdefault("_fin_descs", {})
            if ck not in descs:
                descs[ck] = be.finalize_desc(a.partial, sp.seg_first, a.seg_l2, a.sq, a.wsq, l2_out, a.nseg, arrive,
                                             adam_t=self.adam_t, drop_step=self.drop_step, lr=self.lr_dev, lr_t=self.lr_t_dev,
                                             beta1=opt.beta_1, beta2=opt.beta_2, guard=gd, **kw)
            be.adam_fin(a.theta, self.opt_m, self.opt_v, a.grad, sp.span_seg[s1:], sp.span_off[s1:], sp.span_len[s1:],
                        a.sq_override, sp.nspan - s1, opt.epsilon, clip, descs[ck], **self._ring_args())
            return
        be.step_finalize(a.partial, sp.seg_first, a.seg_l2, a.sq, a.wsq, l2_out, a.nseg, adam_t=self.adam_t,
                         drop_step=self.drop_step, lr=self.lr_dev, lr_t=self.lr_t_dev if adam else None,
                         beta1=opt.beta_1 if adam else 0.0, beta2=opt.beta_2 if adam else 0.0, guard=gd, **kw)
        if adam:
            be.adam(a.theta, self.opt_m, self.opt_v, a.grad, sp.span_seg[s1:], sp.span_off[s1:], sp.span_len[s1:], a.seg_l2,
                    a.sq, a.sq_override, sp.nspan - s1, 0.0, self.lr_t_dev, opt.beta_1, opt.beta_2, opt.epsilon, clip,
                    guard=gd, **(self._ring_args() if sp.nspan - s1 > 0 else {}))
            if enc is not None:
                sl = slice(e.off, e.off + e.size)
                be.dense_dw_adam(x, dpre, a.theta[sl], self.opt_m[sl], self.opt_v[sl], e.l2, a.sq[e.seg:e.seg + 1],
                                 a.sq_override[e.seg:e.seg + 1], self.lr_t_dev, opt.beta_1, opt.beta_2, opt.epsilon, clip,
                                 N, E, rows, ldx, guard=gd)
        else:
            be.sgd(a.theta, self.opt_m, a.grad, sp.span_seg, sp.span_off, sp.span_len, a.seg_l2, a.sq, a.sq_override,
                   sp.nspan, 0.0, self.lr_dev, opt.momentum, clip, guard=gd)

    # ------------------------------------------------------------------ BatchNorm, optionally synchronised across replicas
    def _sync_bn_on(self):
        """Synchronised BatchNorm (``sync_bn = True`` on a data-parallel model): batch statistics over the GLOBAL batch, so
        G replicas x local batch train exactly like one process on the concatenated batch.  The reference has no
        counterpart (it trains on one device); per-replica statistics remain the default.  The collectives sit inside
        the forward / backward pass, so dp.attach puts such a model on the generic (non-pipelined, eager) schedule."""
        return bool(getattr(self, "sync_bn", False) and self.dp_world > 1)

    def _bn_fwd(self, x, gamma, beta, mov_mean, mov_var, y, xhat, inv_std, rows, C, ldy, training, work, drop=None):
        """drop = (rate, seed, site, step_dev), training only: the Dropout over y that follows the normalisation, fused into
        the apply pass where the backend offers it (returns True when it was applied)"""
        be = self.be
        if not (training and self._sync_bn_on()):
            if drop is not None and drop[0] > 0 and getattr(self, "fused_bn_drop", True):
                be.batchnorm_fwd(x, gamma, beta, mov_mean, mov_var, y, xhat, inv_std, rows, C, ldy, training, BN_EPS,
                                 BN_MOMENTUM, work, drop=drop)
                return True
            be.batchnorm_fwd(x, gamma, beta, mov_mean, mov_var, y, xhat, inv_std, rows, C, ldy, training, BN_EPS, BN_MOMENTUM,
                             work)
            return False
        import torch.distributed as dist
        n = be.bn_nchunk(rows) * 2 * C
        part, allp = work[C:C + n], self._bn_scratch(self.dp_world * n)
        be.batchnorm_stats(x, rows, C, part)
        dist.all_gather(list(allp.view(self.dp_world, n).unbind(0)), part.contiguous())
        be.batchnorm_apply_stats(allp, self.dp_world, x, gamma, beta, mov_mean, mov_var, y, xhat, inv_std, rows, C, ldy, BN_EPS,
                                 BN_MOMENTUM, work)

    def _bn_bwd(self, dy, xhat, gamma, inv_std, dx, dgamma, dbeta, rows, C, lddy, work, act_pre=None, slope=0.2):
        """act_pre: LeakyReLU' of the activation in front of the normalisation folded into the dx pass (returns True when
        it was applied)"""
        be = self.be
        if not self._sync_bn_on():
            if act_pre is not None and getattr(self, "fused_bn_drop", True):
                be.batchnorm_bwd(dy, xhat, gamma, inv_std, dx, dgamma, dbeta, rows, C, lddy, True, work, act_pre=act_pre,
                                 slope=slope)
                return True
            be.batchnorm_bwd(dy, xhat, gamma, inv_std, dx, dgamma, dbeta, rows, C, lddy, True, work)
            return False
        import torch.distributed as dist
        # local sums stay in the gradient buffers (the replicas' gradients are averaged as usual); dx needs the global sums
        be.batchnorm_bwd(dy, xhat, gamma, inv_std, None, dgamma, dbeta, rows, C, lddy, True, work)
        sums = self._bn_scratch(2 * C)
        sums[:C].copy_(dgamma.reshape(-1)[:C]); sums[C:2 * C].copy_(dbeta.reshape(-1)[:C])
        dist.all_reduce(sums[:2 * C], op=dist.ReduceOp.SUM)
        be.batchnorm_dx(dy, lddy, xhat, gamma, inv_std, sums[:C], sums[C:2 * C], dx, rows, C, rows * self.dp_world)

    def _bn_scratch(self, n):
        buf = self.__dict__.get("_bn_buf")
        if buf is None or buf.numel() < n:
            buf = self._bn_buf = self._f(n)
        return buf[:n]

    def _emb_sparse_ok(self, E, ldd):
        """the sparse Embedding backward runs (single-process fused step)"""
        return bool(self.__dict__.get("_defer_sum2") and self.dp_world == 1 and getattr(self, "sparse_emb_bwd", True)
                    and hasattr(self.be, "embedding_bwd_sparse") and E % 4 == 0 and ldd % 4 == 0)

    def _embedding_bwd(self, drows, ids, name, B, T, E, ldd, V, drop=None, zero_id=-1):
        """Embedding scatter + IndexedSlices norm.  Inside the fused single-process step (``_defer_sum2``) the sparse
        form runs: no table-wide zero fill (rows touched by the previous step only are cleaned through ``prev_ids``),
        the norm as per-block partials that the step-finalize launch sums.  Anywhere else (data parallel: the all-reduce
        writes rows of other ranks' tokens; SAM; eager paths) the dense form runs and hands its ids on as prev_ids, so
        the invariant "dtable is zero outside the rows of prev_ids" holds whichever form runs next."""
        be, a = self.be, self.arena
        seg = a.entries[name].seg
        sqo = a.sq_override[seg:seg + 1]
        st = self.__dict__.get("_emb_state")
        if st is None or st[0] != (B, T, name):
            if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("embedding backward state must be built outside a graph capture (run one eager step)")
            nparts = be.embedding_bwd_parts(B, T, E) if hasattr(be, "embedding_bwd_parts") else 1
            st = self._emb_state = ((B, T, name), torch.full((B * T,), -1, dtype=torch.int32, device=self.device),
                                    self._f(nparts), nparts)
            a.g(name).zero_()
        _, prev, parts, nparts = st
        sparse = self._emb_sparse_ok(E, ldd)
        assert drop is None or sparse, "the input-dropout mask can only ride on the sparse form"
        if sparse:
            kw = dict(drop_rate=drop[0], drop_seed=drop[1], drop_site=drop[2], drop_step_dev=drop[3]) if drop else {}
            if zero_id >= 0 and getattr(self, "emb_skip_masked", True):
                kw["zero_id"] = zero_id       # rows of the mask id carry no gradient (the caller's guarantee): not read
            be.embedding_bwd_sparse(drows, ids, prev, a.g(name), parts, B, T, E, ldd, V, **kw)
            self._emb_finalize = (parts, None if self.__dict__.get("agc") else sqo, nparts, ids, prev, B * T)
        else:
            be.embedding_bwd(drows, ids, a.g(name), sqo, self.rowsq, B, T, E, ldd, V)
            prev.copy_(ids.reshape(-1))

    def _sum2(self, x0, out0, x1, out1, n, scale):
        """loss / accuracy totals: deferred into the step-finalize launch when a fused update follows in the same
        launch sequence (``_defer_sum2`` set by train_step), else their own launch"""
        if self.__dict__.get("_defer_sum2"):
            self._sum2_deferred = (x0, out0, x1, out1, n, scale)
        else:
            self.be.sum2(x0, out0, x1, out1, n, scale)

    def _tick(self):
        opt, gd = self.optimizer, self._guard_word()
        if opt.kind == "adam":
            self.be.step_tick(self.adam_t, self.drop_step, self.lr_dev, self.lr_t_dev, opt.beta_1, opt.beta_2, guard=gd)
        else:
            self.be.step_tick(self.adam_t, self.drop_step, self.lr_dev, None, 0.0, 0.0, guard=gd)

    def _update_slice(self, sl):
        """norms + clip + optimizer on one contiguous range of variables (arena.seg_slice); the caller
        runs _tick() once before the first slice and l2_total after the last."""
        be, a, opt = self.be, self.arena, self.optimizer
        clip = opt.clipnorm if opt.clipnorm is not None else 0.0
        be.seg_sqnorm(a.theta, a.grad, sl.span_seg, sl.span_off, sl.span_len, sl.seg_first, a.seg_l2, sl.partial,
                      sl.sq, sl.wsq, None, sl.nspan, sl.nseg)
        gd = self._guard_word()
        if opt.kind == "adam":
            be.adam(a.theta, self.opt_m, self.opt_v, a.grad, sl.span_seg, sl.span_off, sl.span_len, a.seg_l2, a.sq,
                    a.sq_override, sl.nspan, 0.0, self.lr_t_dev, opt.beta_1, opt.beta_2, opt.epsilon, clip, guard=gd)
        else:
            be.sgd(a.theta, self.opt_m, a.grad, sl.span_seg, sl.span_off, sl.span_len, a.seg_l2, a.sq, a.sq_override,
                   sl.nspan, 0.0, self.lr_dev, opt.momentum, clip, guard=gd)

    @staticmethod
    def pick_splitk(M, N, K):
        """Split-K factor (power of two) so that a GEMM launches ~1024 workgroups of 64x64 tiles
        (4 per CU), calibrated with tools/gemm_bench.py: head dX (120 tiles) -> 8, dXin (128) -> 8,
        dU (256) -> 4, head dW (632) -> 2, encoder forward (8 tiles, K = 20000) -> 64."""
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        if K < 256:
            return 1
        sk = 1
        while sk * 2 * tiles <= 1280 and K // (sk * 2) >= 128 and sk < 64:
            sk *= 2
        return sk

    def _alloc_splitk(self, shapes):
        """Workspace for the split-K GEMMs of this model: shapes = [(M, N, K), ...]."""
        need = max([self.pick_splitk(*s) * s[0] * s[1] for s in shapes] + [1])
        self.skwork = self._f(need)

    def _g3_space(self, floats):
        """Split-K exchange space of tnt_gemm3_f32: ONE armed work buffer and ONE error word per model, shared by every
        launch (launches of a model run one after the other on its stream, and every launch leaves the buffer armed).
        Grown during eager passes only; growing invalidates captured graphs."""
        d = self.__dict__
        w, sy = d.get("_g3_work"), d.get("_g3_sync")
        if w is None or w.numel() < floats or sy is None:
            if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("gemm3 split-K workspace too small inside a graph capture (run one eager step first)")
            if w is None or w.numel() < floats:
                w = d["_g3_work"] = self._f((max(floats, 4) + 3) // 4 * 4)
                self.be.gemm3_work_arm(w)
            if sy is None:
                sy = d["_g3_sync"] = torch.zeros(64, dtype=torch.int32, device=self.device)
            self._graphs = {}
        return w, sy

    def _g3_plan(self, M, N, K, transA, transB, batch, colsum):
        key = (M, N, K, bool(transA), bool(transB), batch, bool(colsum))
        plans = self.__dict__.setdefault("_g3_plans", {})
        if key not in plans:
            force = getattr(self, "g3_force", {}).get(key[:6])          # tools / tests: {(M, N, K, tA, tB, batch): (tile, splitk)}
            plans[key] = force or self.be.gemm3_plan(M, N, K, transA, transB, batch, allow_split=not colsum)
        return plans[key]

    def gemm3(self, A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, colsum=None, A2=None, C2=None):
        """One product (or two sharing B) on the hand-written FP32-MFMA family of csrc/gemm3.hip, tile and K split from the
        library's cost model (tnt_gemm3_plan; cached per shape).  True = call issued."""
        be = self.be
        if not getattr(self, "use_gemm3", True) or not hasattr(be, "gemm3") or (transA and transB):
            return False
        if lda % 4 or ldb % 4 or ldc % 4:
            return False
        batch = 2 if A2 is not None else 1
        tile, sk = self._g3_plan(M, N, K, transA, transB, batch, colsum is not None)
        work, sync = self._g3_space(be.gemm3_work_floats(M, N, tile, sk, batch)) if sk > 1 else (None, None)
        be.gemm3(A, B, C, M, N, K, lda, ldb, ldc, transA=transA, transB=transB, bias=bias, colsum=colsum, A2=A2, C2=C2,
                 tile=tile, splitk=sk, work=work, sync=sync)
        return True

    def gemm3_pair(self, p, q):
        """Two INDEPENDENT products in one launch (tnt_gemm3_pair_f32): p, q = dicts of gemm3's arguments.  True = issued;
        False = this pair has no co-launch form (the caller issues the two products itself)."""
        be = self.be
        if (not getattr(self, "use_gemm3", True) or not getattr(self, "g3_pairs", True) or not hasattr(be, "gemm3_pair")):
            return False
        descs, need = [], 0
        for d in (p, q):
            if d["lda"] % 4 or d["ldb"] % 4 or d["ldc"] % 4:
                return False
            batch = 2 if d.get("A2") is not None else 1
            tile, sk = self._g3_plan(d["M"], d["N"], d["K"], d.get("transA", False), d.get("transB", False), batch,
                                     d.get("colsum") is not None)
            if d.get("small"):            # a small product rides along: the 32-deep tile of the SAME shape has a pair form
                tile = {9: 7, 10: 5}.get(tile, tile)      # (same tile count, so the planned split stays valid)
            wf = (be.gemm3_work_floats(d["M"], d["N"], tile, sk, batch) + 3) // 4 * 4 if sk > 1 else 0
            descs.append((d, tile, sk, need, wf))
            need += wf
        (d1, t1, _, _, _), (d2, t2, _, _, _) = descs
        if not be.gemm3_pair_supported(t1, d1.get("transA", False), d1.get("transB", False), t2, d2.get("transA", False),
                                       d2.get("transB", False)):
            return False
        work, sync = self._g3_space(need) if need else (None, None)
        # the descriptors are passed by address and recorded launch plans re-issue the call later: one descriptor pair per
        # distinct argument set, kept for the life of the model (same operands -> same objects)
        ck = tuple((tuple((k, v.data_ptr() if torch.is_tensor(v) else v) for k, v in sorted(d.items()) if k != "small"), tile, sk, off)
                   for d, tile, sk, off, _ in descs) + (work.data_ptr() if need else 0,)
        keep = self.__dict__.setdefault("_g3_descs", {})
        if ck in keep:
            be.gemm3_pair(*keep[ck])
            return True
        out = []
        for d, tile, sk, off, wf in descs:
            out.append(be.gemm3_desc(d["A"], d["B"], d["C"], d["M"], d["N"], d["K"], d["lda"], d["ldb"], d["ldc"],
                                     transA=d.get("transA", False), transB=d.get("transB", False), bias=d.get("bias"),
                                     colsum=d.get("colsum"), A2=d.get("A2"), C2=d.get("C2"), tile=tile, splitk=sk,
                                     work=work[off:off + wf] if wf else None, sync=sync if wf else None))
        keep[ck] = (out[0], out[1])
        be.gemm3_pair(out[0], out[1])
        return True

    def _route_lt(self, A, B, C, M, N, K, lda, ldb, ldc, ws, kw):
        """A/B tool since round 3 (``use_gemm3 = False``): hipBLASLt (tnt_gemm_lt_f32) for the vocabulary-sized GEMMs, i.e. the
        head forward and its two gradients; rocBLAS for the LSTM-sized ones (gemm_sk below).  True = call issued."""
        be = self.be
        if (not getattr(self, "use_lt", True) or not hasattr(be, "gemm_lt") or kw.get("pre") is not None or kw.get("act", 0)
                or kw.get("accumulate") or max(N, K) < 4096 or 2.0 * M * N * K < getattr(self, "lt_min_flops", 1e9)):
            return False
        tA, tB = bool(kw.get("transA", False)), bool(kw.get("transB", False))
        if tA and tB:
            return False
        be.gemm_lt(A, B, C, M, N, K, lda, ldb, ldc, transA=tA, transB=tB, bias=kw.get("bias"))
        return True

    def gemm_sk(self, A, B, C, M, N, K, lda, ldb, ldc, ws=0, **kw):
        """GEMM with the calibrated split-K choice.  ``ws`` selects the split-K workspace (one per
        concurrent branch, see ``side``).  Workspaces grow on demand during eager (warm-up) passes;
        growing one invalidates captured graphs, which are then re-captured."""
        # default since round 3: every product without an activation epilogue that is large enough to fill the chip runs on
        # the hand-written family (csrc/gemm3.hip); the vendor libraries remain as A/B tools (use_gemm3 = False)
        if (kw.get("pre") is None and kw.get("act", 0) == 0 and not kw.get("accumulate") and 2.0 * M * N * K >= getattr(self, "g3_min_flops", 3e7)
                and self.gemm3(A, B, C, M, N, K, lda, ldb, ldc, transA=kw.get("transA", False), transB=kw.get("transB", False),
                               bias=kw.get("bias"))):
            return
        if not getattr(self, "use_gemm3", True) and self._route_lt(A, B, C, M, N, K, lda, ldb, ldc, ws, kw):
            return
        plain = kw.get("bias") is None and kw.get("pre") is None and kw.get("act", 0) == 0
        # One shape family where the vendor's pick is poor: NT with a narrow output and a very long K (config 3's
        # head dX = dlogits[960x5001] @ Wo^T[5001x256]: 59 us = 42 TF, against 38 us for the tiled kernel with
        # split-K 16; tools/c3_head_grad_probe.py).  At N = 512 the two are level and the library stays.
        blas_poor = kw.get("transB", False) and not kw.get("transA", False) and N <= 256 and K >= 16 * N
        if (plain and not blas_poor and getattr(self, "use_blas", True) and not getattr(self, "use_gemm3", True)
                and hasattr(self.be, "gemm_blas")):
            # no fused epilogue (weight / input gradients): the vendor's stream-K sgemm (tnt_gemm_blas_f32) needs no
            # split-K pass + reduce launch on these skinny-output / long-K shapes
            self.be.gemm_blas(A, B, C, M, N, K, lda, ldb, ldc, transA=kw.get("transA", False),
                              transB=kw.get("transB", False), accumulate=kw.get("accumulate", False))
            return
        sk = self.pick_splitk(M, N, K)
        if sk > 1:
            pool = self.__dict__.setdefault("_skw", {})
            buf = self.skwork if ws == 0 else pool.get(ws)
            if buf is None or sk * M * N > buf.numel():
                if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("split-K workspace too small inside a graph capture")
                buf = self._f(sk * M * N)
                if ws == 0:
                    self.skwork = buf
                else:
                    pool[ws] = buf
                self._graphs = {}
            self.be.gemm(A, B, C, M, N, K, lda, ldb, ldc, splitk=sk, work=buf, **kw)
        else:
            self.be.gemm(A, B, C, M, N, K, lda, ldb, ldc, **kw)

    # ---- intra-step concurrency: independent gradient GEMMs run on side streams (captured as parallel
    # graph branches) next to the latency-bound BPTT chain, which leaves most CUs idle.
    @contextmanager
    def side(self, i):
        # Measured on MI355X (tools/side_bench.py): 0.868 ms/step without, 0.94-1.07 ms with side branches --
        # the concurrent GEMM workgroups crowd out the LDS-heavy BPTT step kernels.  Off by default.
        if i < 0 or self.device.type != "cuda" or not getattr(self, "use_side_streams", False):
            yield
            return
        streams = self.__dict__.setdefault("_side_streams", {})
        if i not in streams:
            streams[i] = torch.cuda.Stream(device=self.device)
        s = streams[i]
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            yield
        self.__dict__.setdefault("_side_used", set()).add(i)

    def join(self):
        used = self.__dict__.get("_side_used")
        if not used:
            return
        main = torch.cuda.current_stream()
        for i in sorted(used):
            main.wait_stream(self._side_streams[i])
        used.clear()

    # ------------------------------------------------------------------ adaptive gradient clipping (agc.py)
    def enable_agc(self, clip_factor=0.01, eps=1e-3):
        """gradients = agc.adaptive_clip_grad(trainable_variables, gradients, clip_factor, eps) before
        optimizer.apply_gradients -- the (commented-out) call of lc_NIC.py:388.  clip_factor=None switches it off."""
        if clip_factor is not None and int(self.__dict__.get("dp_world", 1) or 1) > 1:
            raise NotImplementedError("adaptive gradient clipping is not supported under data parallel (dp.attach)")
        self.agc = None if clip_factor is None else (float(clip_factor), float(eps))
        self._graphs = {}

    def _emb_row_grads(self):
        """(row-gradient matrix of the Embedding [n][E], n, E, ld, arena name) or None: the IndexedSlices values whose
        un-deduplicated column norms agc.py:25-30 uses.  Set by the models that have an Embedding."""
        return self.__dict__.get("_emb_rows")

    def _apply_agc(self):
        if not self.__dict__.get("agc"):
            return
        from .arena import AgcTable
        a, er = self.arena, self._emb_row_grads()
        tab = self.__dict__.get("_agc_tab")
        if tab is None:
            shapes = {n: self.keras_shapes[n] for n in a.entries}
            tab = self._agc_tab = AgcTable(a, shapes, er[4] if er else None)
            self._agc_colsq = self._f(er[2]) if er else None
        if er is not None:
            x, n, E, ld, name = er
            self.be.colsq(x, self._agc_colsq, n, E, ld)
            seg = a.entries[name].seg
            self.be.agc(a.theta, a.grad, tab, self._agc_colsq, a.sq_override[seg:seg + 1], *self.agc)
        else:
            self.be.agc(a.theta, a.grad, tab, None, None, *self.agc)

    def _norms_and_l2(self, l2_out):
        a, sp = self.arena, self.arena.spans
        self.be.seg_sqnorm(a.theta, a.grad, sp.span_seg, sp.span_off, sp.span_len, sp.seg_first, a.seg_l2, a.partial,
                           a.sq, a.wsq, l2_out, sp.nspan, a.nseg)

    # ------------------------------------------------------------------ weights
    @property
    def trainable_variables(self):
        return [(n, self.get_weight(n)) for n in self.trainable_names()]

    def trainable_names(self):
        return [n for n in self.keras_shapes if "moving_" not in n]

    def get_weights_dict(self):
        return OrderedDict((n, self.get_weight(n)) for n in self.keras_shapes)

    def set_weights_dict(self, d, strict=True):
        for n, v in d.items():
            if n in self.keras_shapes:
                self.set_weight(n, v)
            elif strict:
                raise KeyError(n)

    def get_optimizer_slot(self, name, slot):
        """Adam first ('m') / second ('v') moment (SGD: 'm' = momentum) of one trainable, in the keras layout."""
        buf = self.opt_m if slot == "m" else self.opt_v
        return self._unpack(name, self.arena.slot(buf, name))

    def get_layer(self, name):
        if name not in self.layers_spec:
            raise ValueError(f"No such layer: {name}")
        return _LayerView(self, name, self.layers_spec[name])

    def save_weights(self, path):
        """ModelCheckpoint(save_weights_only=True) target (main.py:168-190).  ``*.h5`` / ``*.hdf5``: a Keras weight file
        (layer_names / weight_names attributes, one float32 dataset per weight, keras layouts), written by the
        pure-Python h5lite module -- readable by h5py / Keras ``load_weights(by_name=True)`` wherever the layer names
        agree.  Anything else: ``.npz`` with the same names and layouts."""
        path = str(path)
        if path.endswith((".h5", ".hdf5")):
            from . import h5lite
            layers = OrderedDict((layer, [(f"{layer}/{w}:0", self.get_weight(f"{layer}/{w}")) for w in ws])
                                 for layer, ws in self.layers_spec.items())
            h5lite.write_keras_weights(path, layers)
            return
        arrs = {k.replace("/", "__"): v for k, v in self.get_weights_dict().items()}
        with open(path, "wb") as f:
            np.savez(f, **arrs)

    def load_weights(self, path, by_name=True, skip_mismatch=False, name_map=None):
        """model.load_weights(path, by_name=True, skip_mismatch=True) -- eval.py:140.  Reads ``.npz`` (this library) and
        Keras ``.h5`` weight files (h5lite: contiguous float datasets, old-style groups -- what Keras / h5py write).
        Weights are matched BY NAME: ``<layer>/<weight>`` of ``model.layers_spec``; ``name_map`` ({file layer name: model
        layer name}) renames layers of a file whose auto-generated Keras names differ (the reference's checkpoints are
        not available here, so their exact layer names are unpinned).  Unknown layers are ignored (by_name semantics);
        a shape mismatch raises unless ``skip_mismatch``."""
        path = str(path)
        name_map = name_map or {}
        items = []
        if path.endswith((".h5", ".hdf5")):
            from . import h5lite
            _, layers = h5lite.read_keras_weights(path)
            for ln, ws in layers.items():
                tgt = name_map.get(ln, ln)
                for wn, arr in ws:
                    w = wn.split("/")[-1].split(":")[0]
                    items.append((f"{tgt}/{w}", arr))
        else:
            with np.load(path, allow_pickle=False) as z:
                items = [(name_map.get(k.replace("__", "/"), k.replace("__", "/")), z[k]) for k in z.files]
        for n, arr in items:
            if n not in self.keras_shapes:
                continue
            if tuple(arr.shape) != tuple(self.keras_shapes[n]):
                if skip_mismatch:
                    continue
                raise ValueError(f"shape mismatch for {n}: {arr.shape} vs {self.keras_shapes[n]}")
            self.set_weight(n, arr)

    def count_params(self):
        return int(sum(np.prod(s) for s in self.keras_shapes.values()))

    def summary(self, print_fn=print):
        print_fn(f'Model: "{type(self).__name__}"')
        for layer, ws in self.layers_spec.items():
            n = sum(int(np.prod(self.keras_shapes[f"{layer}/{w}"])) for w in ws)
            print_fn(f"  {layer:36s} {n:>12,d}")
        print_fn(f"Total params: {self.count_params():,d}")

    # ------------------------------------------------------------------ input staging
    def _to_dev(self, a, dtype):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype, non_blocking=True)
        return torch.as_tensor(np.asarray(a), dtype=dtype).to(self.device, non_blocking=True)

    def _stage_mask_job(self):
        """(out, n, nsites, rate, seed, site0, step_dev) of the dropout masks a training step wants generated with its
        batch staging (attention model: the stored attention-dropout masks of all T steps), or None"""
        return None

    def _stage_batch(self, inputs, target, n_cols, masks=False):
        """(inputs, target) -> static buffers.  A batch that already sits on the model's device in the staged
        dtypes (float32 -- or float16 "on-wire" -- betas, float32 states, int32 ids, contiguous) goes through ONE launch
        (tnt_stage_batch_f32 / _h16);
        anything else (numpy, one-hot targets, other dtypes) takes the general per-tensor path."""
        x, cap, a0, c0 = inputs[:4]
        ts = [x, cap, a0, c0] + ([target] if target is not None else [])
        dev = self.device
        same = lambda t: t.device.type == dev.type and (t.device.index or 0) == (dev.index or 0)
        ok = all(isinstance(t, torch.Tensor) and same(t) and t.is_contiguous() for t in ts)
        ok = ok and x.dtype in (torch.float32, torch.float16) and a0.dtype == c0.dtype == torch.float32
        ok = ok and cap.dtype == torch.int32
        ok = ok and x.dim() == 2 and cap.dim() == 2 and x.shape == (cap.shape[0], n_cols)
        ok = ok and (target is None or (target.dtype == torch.int32 and target.shape == cap.shape))
        if not ok:
            B, T = self._stage_inputs(inputs)
            if target is not None:
                self._stage_target(target, B, T)
            mk = self._stage_mask_job() if masks else None
            if mk is not None:
                self.be.dropout_mask4(mk[0], mk[1], mk[2], mk[3], mk[4], mk[5], 0, mk[6])
            return B, T
        B, T = cap.shape
        self._build(B, T)
        assert a0.shape == c0.shape == (B, self.U), f"state shape {tuple(a0.shape)} != {(B, self.U)}"
        xT = getattr(self, "xT", None)       # voxel-major copy for the region-wise encoder, written in the same launch
        kw = {}
        mk = self._stage_mask_job() if masks else None
        if mk is not None:                   # the step's dropout masks ride in the staging launch
            kw["masks"] = mk
        if xT is not None:
            self.be.stage_batch(x, self.x, cap, self.cap, target, self.tgt, a0, self.Hs[0], c0, self.Cs[0], B, T, n_cols,
                                self.ldx, self.U, xT, xT.shape[1], **kw)
        else:
            self.be.stage_batch(x, self.x, cap, self.cap, target, self.tgt, a0, self.Hs[0], c0, self.Cs[0], B, T, n_cols,
                                self.ldx, self.U, **kw)
        return B, T

    def _stage_target(self, target, B, T):
        """target: one-hot (B,T,V) float (to_categorical, data_generator_guse.py:163) or int ids (B,T).
        Fills self.tgt (time-major int32 ids)."""
        if isinstance(target, np.ndarray) and target.ndim == 2 or (isinstance(target, torch.Tensor) and target.dim() == 2):
            t = self._to_dev(target, torch.int32)
            self.tgt.view(T, B).copy_(t.t())
        else:
            oh = self._to_dev(target, torch.float32).contiguous()
            assert oh.shape == (B, T, self.V), f"target shape {tuple(oh.shape)}"
            self.be.onehot_argmax(oh, self.tgt, B, T, self.V)

    # ------------------------------------------------------------------ device guard (persistent LSTM kernel)
    def _guard_word(self):
        """The error word of the persistent kernel's sync state (uint32, device) while that kernel is in use, else None.
        The optimizer kernels take it as ``guard`` and leave the model untouched when it is set."""
        sync = self.__dict__.get("seq_sync")
        return sync[1024:1025] if (sync is not None and self.__dict__.get("_seq_lstm")) else None

    def _guard_out(self):
        """Where the persistent kernel reports its error code for the host: slot GUARD of the metrics buffer."""
        return self.met[self.GUARD:self.GUARD + 1]

    def _run_step(self, run, key, fn):
        """``run(key, fn)`` (_run_captured / _run_planned) for a training step; returns whether that step's finalize launch
        filed the metrics vector in the ring (known when ``fn`` actually runs -- eager or under capture --, remembered per
        key for the replays)."""
        graphs = self.use_graph and self.device.type == "cuda"
        st = self._graphs.get(key) if graphs else None
        executed = (not graphs) or st is None or (isinstance(st, str) and st == "warm")
        self._ring_hit = False
        run(key, fn)
        keys = self.__dict__.setdefault("_ring_keys", set())
        if executed or self._ring_hit:
            (keys.add if self._ring_hit else keys.discard)(key)
        return key in keys

    def _met_snapshot(self, ring=False):
        """This step's metrics vector: the row the step's Adam launch copied it to (``ring``: fused single-process step, no
        device copy behind the step -- a 5 us launch on a 0.55 ms step), else a clone of ``met``."""
        if ring:
            t = self._ring_host
            self._ring_host = t + 1
            row = self.met_ring[t % self.METRIC_RING]
            self._last_ring = (row, t & 0xFFFFFF)
            return row
        self._last_ring = None
        return self.met.clone()

    def _ring_args(self):
        """keyword arguments that make the Adam launch of the fused step file the metrics vector in the ring"""
        if not getattr(self, "metric_ring", True) or self.__dict__.get("met") is None or self.met.numel() > 62:
            return {}
        if self.__dict__.get("met_ring") is None or self.met_ring.shape[1] != self.met.numel() + 1:
            if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                return {}
            self.met_ring = self._f(self.METRIC_RING, self.met.numel() + 1)
            self.ring_t = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._ring_host = 0
            self._graphs = {}
        self._ring_hit = True
        return dict(met=self.met, ring=self.met_ring, ring_t=self.ring_t)

    def _metrics_from(self, m, **slots):
        """Metrics from a snapshot ``m`` of the metrics buffer (_met_snapshot); the guard word of the same snapshot rides along."""
        out = Metrics((k, m[i] if isinstance(i, int) else i) for k, i in slots.items())
        out._ring = self.__dict__.get("_last_ring")
        return out.guarded(self, m[self.GUARD]) if self.__dict__.get("_seq_lstm") else out

    def _on_guard_trip(self, code):
        self.disable_seq_lstm()
        what = {1: "a barrier timed out", 2: "a launch did not place 32 workgroups on every XCD"}.get(code, "unknown")
        raise DeviceGuardError(
            f"persistent LSTM kernel: device guard tripped (code {code}: {what}).  The results of that step are invalid; "
            "its optimizer update was skipped, so weights, moments and step counters are unchanged.  NOT unchanged: the "
            "BatchNorm moving mean / variance, which the forward pass of the faulted step already advanced (a retried step "
            "applies that 1 % moving-average update a second time), and under data parallel the trip is per rank -- the other "
            "replicas have applied their update, so re-broadcast the parameters (dp.broadcast_parameters) before going on.  "
            "The model now uses the per-step LSTM kernels: run the step again.")

    def _guarded(self, fn):
        """Inference paths: run ``fn`` (which ends in a host read anyway), check the guard word, and on a trip fall back
        to the per-step kernels and run it once more -- inference mutates no model state."""
        out = fn()
        if self._guard_word() is not None and float(self.met[self.GUARD]) != 0.0:
            self.disable_seq_lstm()
            out = fn()
        return out

    def check_device_errors(self):
        """Raises DeviceGuardError if the persistent LSTM kernel's error word is set (synchronises).  train_step /
        test_step results carry the same check with them (Metrics.as_floats), the inference paths check after their
        own host read; this is the explicit form for loops that never read a metric."""
        sync = self.__dict__.get("seq_sync")
        if sync is not None and int(sync[1024].item()) != 0:
            self._on_guard_trip(int(sync[1024].item()))

    def _init_seq_lstm(self, B, U):
        """Opt in to the persistent sequence-forward kernel for this (B, U) on this device.  Probes once per process
        (tnt_lstm_seq_supported synchronises), so it is called from _build, outside any capture."""
        self._seq_lstm = bool(getattr(self, "use_seq_lstm", True) and hasattr(self.be, "lstm_seq_supported")
                              and self.be.lstm_seq_supported(B, U))
        if self._seq_lstm and self.__dict__.get("seq_sync") is None:
            self.seq_sync = torch.zeros(1025, dtype=torch.int32, device=self.device)     # re-armed by the kernel itself
        elif not self._seq_lstm and "seq_sync" not in self.__dict__:
            self.seq_sync = None
        if self._seq_lstm and hasattr(self.be, "lstm_seq_bwd") and getattr(self, "use_seq_lstm_bwd", True):
            n = self.be.lstm_seq_bwd_work_floats(B, U)       # exchange buffer of the persistent BPTT chain
            if self.__dict__.get("seq_xch") is None or self.seq_xch.numel() < n:
                self.seq_xch = self._f(n)
        else:
            self.seq_xch = None

    def disable_seq_lstm(self):
        """Back to the per-step LSTM kernels (after a guard trip of the persistent one, or by choice): zeroes the sync
        state and the guard slot, drops captured graphs / launch plans; the step buffers are reused as they are."""
        self.use_seq_lstm = False
        sync = self.__dict__.get("seq_sync")
        if sync is not None:
            sync.zero_()
        if self.__dict__.get("met") is not None:
            self.met[self.GUARD] = 0
        self._seq_lstm = False
        self._graphs = {}

    # ------------------------------------------------------------------ graph capture
    def _run_captured(self, key, fn):
        """Run ``fn`` (a fixed launch sequence over static buffers) through a hipGraph:
        first call eager (warm-up), second call captures, later calls replay."""
        if not (self.use_graph and self.device.type == "cuda"):
            fn()
            return
        st = self._graphs.get(key)
        if st is None:
            fn()
            self._graphs[key] = "warm"
        elif st == "warm":
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may query events while we capture (data parallel)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
            self._graphs[key] = g
            g.replay()
        else:
            st.replay()

    def _run_planned(self, key, fn):
        """Like _run_captured, but the segment is replayed as a recorded list of C-ABI launches instead of a
        hipGraph: first call eager, second call records every backend call ``fn`` makes (HipBackend._call), later
        calls re-issue those bound calls.  The device sees plain kernel launches -- no graph-launch gap (~15-20 us
        per hipGraphLaunch) -- and the host skips the Python argument plumbing (~9 -> ~4 us per launch).  Same
        contract as capture: ``fn`` may only make backend launches on static buffers, on the current stream."""
        if not (self.use_graph and self.device.type == "cuda"):
            fn()
            return
        st = self._graphs.get(key)
        if st is None:
            fn()
            self._graphs[key] = "warm"
            return
        stream = self.be._s()
        if st == "warm" or st[0] != stream:
            self.be._rec = rec = []
            try:
                fn()
            finally:
                self.be._rec = None
            self._graphs[key] = (stream, rec)
            return
        for f, name, args in st[1]:
            if f(*args) != 0:
                raise RuntimeError(f"{name} failed while replaying launch plan {key}")

    def _dp_mean_logs(self, logs):
        """epoch logs averaged over the data-parallel ranks (same keys on every rank, sorted)"""
        import torch.distributed as dist
        keys = sorted(logs)
        vals = _dp_reduce([logs[k] for k in keys], dist.ReduceOp.SUM)
        return {k: v / self.dp_world for k, v in zip(keys, vals)}

    def _dp_any(self, flag):
        import torch.distributed as dist
        return _dp_reduce([1.0 if flag else 0.0], dist.ReduceOp.MAX)[0] > 0

    # ------------------------------------------------------------------ fit loop
    def fit(self, x=None, epochs=1, steps_per_epoch=None, batch_size=None, callbacks=None, validation_data=None,
            validation_steps=None, initial_epoch=0, verbose=1, keras_last_batch_logs=False, **kw):
        """model.fit(generator, epochs, steps_per_epoch, batch_size, callbacks, validation_data,
        validation_steps, initial_epoch) -- main.py:269-281.  Honours the keras callback protocol
        (on_train_begin, on_epoch_begin, on_train_batch_end, on_test_batch_end, on_epoch_end,
        on_train_end; Callbacks/EpochLoss.py:21-52).
        Epoch logs: the MEAN of the per-batch logs (what keras' compiled metrics report).  The reference's models
        override train_step / test_step and return plain tensors, for which Keras 2.4 hands ``on_epoch_end`` the
        LAST batch's values -- ``keras_last_batch_logs=True`` reproduces that (it matters to
        ModelCheckpoint(save_best_only) / EarlyStopping on ``val_loss``); the mean is the default because it is
        what those callbacks are meant to see.
        Data parallel: the epoch logs are averaged over the ranks and ``stop_training`` is OR-ed before the
        callbacks' decision takes effect, so every rank leaves the loop in the same epoch."""
        callbacks = list(callbacks or [])
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
            else:
                cb.model = self
        history = {}
        _call(callbacks, "on_train_begin", {})
        self.stop_training = False
        for epoch in range(initial_epoch, epochs):
            _call(callbacks, "on_epoch_begin", epoch, {})
            n = len(x) if steps_per_epoch is None else steps_per_epoch
            sums, last, t0 = {}, {}, time.time()
            for b in range(n):
                _call(callbacks, "on_train_batch_begin", b, {})
                try:
                    logs = self.train_step(x[b]).as_floats()
                except DeviceGuardError:        # the step left the model untouched and the fallback is in place: redo it
                    logs = self.train_step(x[b]).as_floats()
                for k, v in logs.items():
                    sums[k] = sums.get(k, 0.0) + v
                last = logs
                _call(callbacks, "on_train_batch_end", b, logs)
            elogs = dict(last) if keras_last_batch_logs else {k: v / max(n, 1) for k, v in sums.items()}
            if validation_data is not None:
                nv = len(validation_data) if validation_steps is None else validation_steps
                vs, vlast = {}, {}
                for b in range(nv):
                    try:
                        logs = self.test_step(validation_data[b]).as_floats()
                    except DeviceGuardError:
                        logs = self.test_step(validation_data[b]).as_floats()
                    for k, v in logs.items():
                        vs[k] = vs.get(k, 0.0) + v
                    vlast = logs
                    _call(callbacks, "on_test_batch_end", b, logs)
                elogs.update({f"val_{k}": v for k, v in vlast.items()} if keras_last_batch_logs else
                             {f"val_{k}": v / max(nv, 1) for k, v in vs.items()})
            if self.dp_world > 1:
                elogs = self._dp_mean_logs(elogs)
            if verbose:
                print(f"epoch {epoch + 1}/{epochs} - {time.time() - t0:.1f}s - " +
                      " - ".join(f"{k}: {v:.4f}" for k, v in elogs.items()))
            for k, v in elogs.items():
                history.setdefault(k, []).append(v)
            self.check_device_errors()
            _call(callbacks, "on_epoch_end", epoch, elogs)
            if hasattr(x, "on_epoch_end"):
                x.on_epoch_end()
            if self.dp_world > 1:
                self.stop_training = self._dp_any(self.stop_training)
            if self.stop_training:
                break
        _call(callbacks, "on_train_end", {})
        return history


def _dp_reduce(values, op):
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(values, dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=op)
    return t.cpu().tolist()


def _call(callbacks, name, *args):
    for cb in callbacks:
        fn = getattr(cb, name, None)
        if fn is not None:
            fn(*args)
