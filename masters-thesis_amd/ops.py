"""Tensor-level wrappers over the C ABI (include/tnt_hip.h).

Each method takes torch device tensors (or views into the flat arenas), passes their
``data_ptr()`` and the current torch stream to the kernel library, and returns nothing:
all outputs are written into caller-owned buffers.  PyTorch is plumbing here (memory,
streams); every arithmetic op of the hot path is a HIP kernel.

``backend()`` returns the active backend.  The only product backend is ``HipBackend``;
tests may install a CPU stand-in with ``set_backend`` to exercise the host
orchestration without a GPU (tests/mock_backend.py -- never used by the product).
"""
import torch

from . import _lib

ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_TANH = 0, 1, 2, 3


def _p(t):
    return None if t is None else t.data_ptr()


import ctypes as _ct


class _G3Desc(_ct.Structure):
    _fields_ = ([(n, _ct.c_void_p) for n in ("A", "B", "C", "bias", "colsum", "A2", "C2")]
                + [(n, _ct.c_int32) for n in ("M", "N", "K", "lda", "ldb", "ldc", "transA", "transB", "tile", "splitk")]
                + [(n, _ct.c_void_p) for n in ("work", "sync")])


class _FinDesc(_ct.Structure):
    """tnt_finalize_desc (include/tnt_hip.h)"""
    _fields_ = [("partial", _ct.c_void_p), ("seg_first", _ct.c_void_p), ("seg_l2", _ct.c_void_p), ("sq", _ct.c_void_p),
                ("wsq", _ct.c_void_p), ("l2_out", _ct.c_void_p), ("nseg", _ct.c_int32),
                ("x0", _ct.c_void_p), ("out0", _ct.c_void_p), ("x1", _ct.c_void_p), ("out1", _ct.c_void_p), ("n", _ct.c_int32),
                ("scale", _ct.c_float),
                ("extra_part", _ct.c_void_p), ("extra", _ct.c_void_p), ("n_extra", _ct.c_int32), ("extra_seg", _ct.c_int32),
                ("ids_src", _ct.c_void_p), ("ids_dst", _ct.c_void_p), ("n_ids", _ct.c_int32),
                ("x2", _ct.c_void_p), ("out2", _ct.c_void_p), ("n2", _ct.c_int32), ("scale2", _ct.c_float),
                ("adam_t", _ct.c_void_p), ("drop_step", _ct.c_void_p), ("lr", _ct.c_void_p), ("lr_t", _ct.c_void_p),
                ("beta1", _ct.c_float), ("beta2", _ct.c_float), ("guard", _ct.c_void_p), ("arrive", _ct.c_void_p)]


class HipBackend:
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.KernelLibraryError("no GPU visible: the HIP backend needs an MI355X (there is no CPU fallback)")

    _rec = None     # while a launch plan is being recorded: list of (c function, name, argument tuple)

    def _call(self, fn, name, *args):
        """Every C-ABI launch goes through here: checks the return code and, while a plan is being recorded
        (ModelBase._run_planned), keeps the bound call so the same launch can be re-issued without the Python
        argument plumbing."""
        if self._rec is not None:
            self._rec.append((fn, name, args))
        rc = fn(*args)
        if rc != 0:
            raise _lib.KernelLibraryError(f"{name} failed with code {rc}")

    @staticmethod
    def _s():
        # raw handle of torch's current stream (also the capturing stream inside a hipGraph capture); the Stream-object
        # route costs ~8 us per launch, which is most of an eager decode step
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())

    def bn_nchunk(self, rows):
        return self.lib.tnt_bn_nchunk(rows)

    def gemm(self, A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, pre=None,
             act=ACT_NONE, slope=0.2, accumulate=False, splitk=1, work=None):
        self._call(self.lib.tnt_gemm_f32, "tnt_gemm_f32", _p(A), _p(B), _p(C), _p(bias), _p(pre), M, N, K, lda, ldb, ldc,
                                         int(transA), int(transB), act, slope, int(accumulate), splitk,
                                         _p(work), self._s())

    def gemm_fused_cfg(self, M, N, K, transA=False, transB=False, batch=1):
        """configuration tnt_gemm_fused_f32 would pick for this shape (0 = none: use gemm with split-K)"""
        return int(self.lib.tnt_gemm_fused_cfg(M, N, K, int(transA), int(transB), batch))

    def gemm_fused(self, A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, colsum=None, A2=None,
                   C2=None, cfg=0):
        """one-round GEMM without an activation epilogue; riders: column sums of B (bias gradient), a second product"""
        self._call(self.lib.tnt_gemm_fused_f32, "tnt_gemm_fused_f32", _p(A), _p(B), _p(C), _p(bias), _p(colsum), _p(A2), _p(C2),
                   M, N, K, lda, ldb, ldc, int(transA), int(transB), cfg, self._s())

    def gemm3(self, A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, colsum=None, A2=None, C2=None,
              tile=1, splitk=1, work=None, sync=None):
        """the round-3 FP32-MFMA family (LDS-DMA staged, b128 fragments): C = op(A) op(B) (+ bias); riders: column sums of B,
        a second product sharing B; splitk > 1 reduces the K splits inside the launch (work / sync: gemm3_work_floats /
        gemm3_sync_words)"""
        self._call(self.lib.tnt_gemm3_f32, "tnt_gemm3_f32", _p(A), _p(B), _p(C), _p(bias), _p(colsum), _p(A2), _p(C2),
                   M, N, K, lda, ldb, ldc, int(transA), int(transB), tile, splitk, _p(work), _p(sync), self._s())

    def gemm3_plan(self, M, N, K, transA=False, transB=False, batch=1, allow_split=True):
        """(tile, splitk) of the library's cost model for this shape"""
        import ctypes
        t, s = ctypes.c_int32(0), ctypes.c_int32(1)
        rc = self.lib.tnt_gemm3_plan(M, N, K, int(transA), int(transB), batch, int(allow_split), ctypes.byref(t), ctypes.byref(s))
        if rc != 0:
            raise RuntimeError(f"tnt_gemm3_plan failed: {rc}")
        return int(t.value), int(s.value)

    def gemm3_work_floats(self, M, N, tile, splitk, batch=1):
        return int(self.lib.tnt_gemm3_work_floats(M, N, tile, splitk, batch))

    def gemm3_sync_words(self, M, N, tile, batch=1):
        return int(self.lib.tnt_gemm3_sync_words(M, N, tile, batch))

    def gemm3_pair(self, d1, d2):
        """two independent products in one launch; d1, d2 = gemm3_desc(...)"""
        import ctypes
        self._call(self.lib.tnt_gemm3_pair_f32, "tnt_gemm3_pair_f32", ctypes.addressof(d1), ctypes.addressof(d2), self._s())

    def gemm3_pair_supported(self, tile1, tA1, tB1, tile2, tA2, tB2):
        return bool(self.lib.tnt_gemm3_pair_supported(tile1, int(tA1), int(tB1), tile2, int(tA2), int(tB2)))

    @staticmethod
    def gemm3_desc(A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, colsum=None, A2=None, C2=None, tile=1,
                   splitk=1, work=None, sync=None):
        """tnt_gemm3_desc (include/tnt_hip.h); the returned object keeps the tensors alive"""
        d = _G3Desc(_p(A), _p(B), _p(C), _p(bias), _p(colsum), _p(A2), _p(C2), M, N, K, lda, ldb, ldc, int(transA), int(transB),
                    tile, splitk, _p(work), _p(sync))
        d._keep = (A, B, C, bias, colsum, A2, C2, work, sync)
        return d

    def gemm3_work_arm(self, work):
        """fill a split-K exchange buffer with the "not written yet" pattern (once; every launch leaves it armed)"""
        self._call(self.lib.tnt_gemm3_work_arm, "tnt_gemm3_work_arm", _p(work), work.numel(), self._s())

    def gemm_tile(self, A, B, C, M, N, K, lda, ldb, ldc, bm, bn, transA=False, transB=False, bias=None, pre=None,
                  act=ACT_NONE, slope=0.2, accumulate=False, splitk=1, work=None):
        """tnt_gemm_f32 with the workgroup tile forced: (64|128, 64|128) = the tiled kernel, (160, 128) = the
        one-round kernel (tests and tools/ only)."""
        self._call(self.lib.tnt_gemm_f32_tile, "tnt_gemm_f32_tile", _p(A), _p(B), _p(C), _p(bias), _p(pre), M, N, K, lda, ldb, ldc,
                                              int(transA), int(transB), act, slope, int(accumulate), splitk,
                                              _p(work), bm, bn, self._s())

    def dropout(self, x, y, rows, cols, ld, tmajor_B, lwidth, lcol0, rate, seed, site, step, step_dev=None,
                rows_per_site=0):
        self._call(self.lib.tnt_dropout_f32, "tnt_dropout_f32", _p(x), _p(y), rows, cols, ld, tmajor_B, lwidth, lcol0, rows_per_site,
                                            rate, seed, site, step, _p(step_dev), self._s())

    def dropout_metric(self, x, y, rows, cols, ld, tmajor_B, lwidth, lcol0, rate, seed, site, step, step_dev, alpha, partial,
                       T, B, R, rows_per_site=0):
        """dropout + the attention metric's partials (attention_metric(out=None)) in one launch"""
        self._call(self.lib.tnt_dropout_metric_f32, "tnt_dropout_metric_f32", _p(x), _p(y), rows, cols, ld, tmajor_B, lwidth,
                   lcol0, rows_per_site, rate, seed, site, step, _p(step_dev), _p(alpha), _p(partial), T, B, R, self._s())

    def dropout2(self, x, y, rows, cols, ld, mask_a, mask_b, seed, step, step_dev=None):
        """two masks in one pass; mask_* = (tmajor_B, lwidth, lcol0, rows_per_site, rate, site)"""
        self._call(self.lib.tnt_dropout2_f32, "tnt_dropout2_f32", _p(x), _p(y), rows, cols, ld, *[v for m in (mask_a, mask_b)
                   for v in (int(m[0]), int(m[1]), int(m[2]), int(m[3]), float(m[4]), int(m[5]))], int(seed), int(step),
                   _p(step_dev), self._s())

    def dropout_mask4(self, out, n, nsites, rate, seed, site0, step, step_dev=None):
        """uint8 keep-masks (4 elements per byte) of ``nsites`` consecutive dropout sites, no data pass."""
        self._call(self.lib.tnt_dropout_mask4_u8, "tnt_dropout_mask4_u8", _p(out), n, nsites, rate, int(seed), int(site0), int(step), _p(step_dev),
                   self._s())

    def bias_act_drop_bwd(self, dy, pre, dx, dbias, rows, cols, ld, act, slope, tmajor_B, lwidth, lcol0, rate, seed, site,
                          step_dev=None, extra=None):
        """dx = dropout'(dy) * act'(pre), dbias = column sums of dx; ``extra`` = (x1, out1, rows1, C1, ld1) second colsum job"""
        x1, out1, rows1, C1, ld1 = extra if extra is not None else (None, None, 0, 0, 0)
        self._call(self.lib.tnt_bias_act_drop_bwd_f32, "tnt_bias_act_drop_bwd_f32", _p(dy), _p(pre), _p(dx), _p(dbias), rows, cols,
                   ld, act, slope, tmajor_B, lwidth, lcol0, rate, int(seed), int(site), _p(step_dev), _p(x1), _p(out1), rows1,
                   C1, ld1, self._s())

    def act_bwd(self, pre, dy, dx, n, act, slope=0.2):
        self._call(self.lib.tnt_act_bwd_f32, "tnt_act_bwd_f32", _p(pre), _p(dy), _p(dx), n, act, slope, self._s())

    def batchnorm_fwd(self, x, gamma, beta, mov_mean, mov_var, y, xhat, inv_std, rows, C, ldy, training, eps,
                      momentum, work, drop=None):
        """drop = (rate, seed, site, step_dev): the Dropout behind the normalisation applied to y in the same pass"""
        if drop is not None and drop[0] > 0:
            rate, seed, site, step_dev = drop
            self._call(self.lib.tnt_batchnorm_fwd_drop_f32, "tnt_batchnorm_fwd_drop_f32", _p(x), _p(gamma), _p(beta), _p(mov_mean),
                       _p(mov_var), _p(y), _p(xhat), _p(inv_std), rows, C, ldy, int(training), eps, momentum, _p(work),
                       float(rate), int(seed), int(site), _p(step_dev), self._s())
            return
        self._call(self.lib.tnt_batchnorm_fwd_f32, "tnt_batchnorm_fwd_f32", _p(x), _p(gamma), _p(beta), _p(mov_mean), _p(mov_var), _p(y),
                                                  _p(xhat), _p(inv_std), rows, C, ldy, int(training), eps, momentum,
                                                  _p(work), self._s())

    def batchnorm_bwd(self, dy, xhat, gamma, inv_std, dx, dgamma, dbeta, rows, C, lddy, training, work, act_pre=None,
                      slope=0.0):
        """act_pre: dx is further multiplied by LeakyReLU'(act_pre, slope) (the activation in front of the normalisation)"""
        if act_pre is not None:
            self._call(self.lib.tnt_batchnorm_bwd_act_f32, "tnt_batchnorm_bwd_act_f32", _p(dy), _p(xhat), _p(gamma), _p(inv_std),
                       _p(dx), _p(dgamma), _p(dbeta), rows, C, lddy, int(training), _p(work), _p(act_pre), float(slope), self._s())
            return
        self._call(self.lib.tnt_batchnorm_bwd_f32, "tnt_batchnorm_bwd_f32", _p(dy), _p(xhat), _p(gamma), _p(inv_std), _p(dx), _p(dgamma),
                                                  _p(dbeta), rows, C, lddy, int(training), _p(work), self._s())

    def batchnorm_stats(self, x, rows, C, part):
        self._call(self.lib.tnt_batchnorm_stats_f32, "tnt_batchnorm_stats_f32", _p(x), rows, C, _p(part), self._s())

    def batchnorm_apply_stats(self, part_all, nrep, x, gamma, beta, mov_mean, mov_var, y, xhat, inv_std, rows, C, ldy, eps,
                              momentum, mean_work):
        self._call(self.lib.tnt_batchnorm_apply_stats_f32, "tnt_batchnorm_apply_stats_f32", _p(part_all), nrep, _p(x), _p(gamma),
                   _p(beta), _p(mov_mean), _p(mov_var), _p(y), _p(xhat), _p(inv_std), rows, C, ldy, eps, momentum, _p(mean_work),
                   self._s())

    def batchnorm_dx(self, dy, lddy, xhat, gamma, inv_std, dgamma_sum, dbeta_sum, dx, rows, C, n_total):
        self._call(self.lib.tnt_batchnorm_dx_f32, "tnt_batchnorm_dx_f32", _p(dy), lddy, _p(xhat), _p(gamma), _p(inv_std),
                   _p(dgamma_sum), _p(dbeta_sum), _p(dx), rows, C, n_total, self._s())

    def layernorm_fwd(self, x, gamma, beta, y, xhat, inv_std, rows, C, ldy, eps):
        self._call(self.lib.tnt_layernorm_fwd_f32, "tnt_layernorm_fwd_f32", _p(x), _p(gamma), _p(beta), _p(y), _p(xhat), _p(inv_std), rows, C,
                                                  ldy, eps, self._s())

    def layernorm_bwd(self, dy, xhat, gamma, inv_std, dx, dgamma, dbeta, rows, C, lddy, work):
        self._call(self.lib.tnt_layernorm_bwd_f32, "tnt_layernorm_bwd_f32", _p(dy), _p(xhat), _p(gamma), _p(inv_std), _p(dx), _p(dgamma),
                                                  _p(dbeta), rows, C, lddy, _p(work), self._s())

    def colsum_multi(self, jobs):
        """jobs: up to four (x, out, rows, C, ld) column-sum jobs of <= 2048 rows each, one launch"""
        args = []
        for k in range(4):
            x, out, rows, C, ld = jobs[k] if k < len(jobs) else (None, None, 0, 0, 0)
            args += [_p(x), _p(out), rows, C, ld]
        self._call(self.lib.tnt_colsum4_f32, "tnt_colsum4_f32", *args, self._s())

    def colsum2(self, x0, out0, rows0, C0, ld0, x1, out1, rows1, C1, ld1):
        self._call(self.lib.tnt_colsum2_f32, "tnt_colsum2_f32", _p(x0), _p(out0), rows0, C0, ld0, _p(x1), _p(out1), rows1, C1, ld1,
                   self._s())

    def colsum(self, x, out, rows, C, ld, work):
        self._call(self.lib.tnt_colsum_f32, "tnt_colsum_f32", _p(x), _p(out), rows, C, ld, _p(work), self._s())

    def embedding_fwd(self, table, ids, out, B, T, E, ldo, V):
        self._call(self.lib.tnt_embedding_fwd_f32, "tnt_embedding_fwd_f32", _p(table), _p(ids), _p(out), B, T, E, ldo, V, self._s())

    def embedding_fwd_drop(self, table, ids, out, out_drop, B, T, E, ldo, V, rate, seed, site, step, step_dev=None,
                           mask2=None):
        """mask2 = (rate2, site2, lwidth2, lcol0_2): the per-timestep LSTM input mask applied behind the Embedding Dropout"""
        if mask2 is not None and mask2[0] > 0:
            self._call(self.lib.tnt_embedding_fwd_drop2_f32, "tnt_embedding_fwd_drop2_f32", _p(table), _p(ids), _p(out), _p(out_drop),
                       B, T, E, ldo, V, rate, int(seed), int(site), int(step), _p(step_dev), float(mask2[0]), int(mask2[1]),
                       int(mask2[2]), int(mask2[3]), self._s())
            return
        self._call(self.lib.tnt_embedding_fwd_drop_f32, "tnt_embedding_fwd_drop_f32", _p(table), _p(ids), _p(out), _p(out_drop), B, T, E, ldo, V, rate,
                                                       int(seed), int(site), int(step), _p(step_dev), self._s())

    def embedding_bwd(self, drows, ids, dtable, sq_norm, rowsq_work, B, T, E, ldd, V):
        self._call(self.lib.tnt_embedding_bwd_f32, "tnt_embedding_bwd_f32", _p(drows), _p(ids), _p(dtable), _p(sq_norm), _p(rowsq_work), B, T,
                                                  E, ldd, V, self._s())

    def lstm_step_fwd(self, xz, h_prev, c_prev, Ur, ctx, Wc, D, mask_ids, mask_T, mask_t, out_prev, h, c, out,
                      gates, B, U, xz_bias=None):
        self._call(self.lib.tnt_lstm_step_fwd_f32, "tnt_lstm_step_fwd_f32", _p(xz), _p(h_prev), _p(c_prev), _p(Ur), _p(ctx), _p(Wc), D,
                                                  _p(mask_ids), mask_T, mask_t, _p(out_prev), _p(h), _p(c), _p(out),
                                                  _p(gates), B, U, _p(xz_bias), self._s())

    def lstm_seq_supported(self, B, U):
        """1 when the persistent sequence kernel can run here (U == 512, 256 CUs, 32 workgroups per XCD); probes once."""
        return bool(self.lib.tnt_lstm_seq_supported(int(B), int(U)))

    def lstm_seq_fwd(self, xz, hs, cs, Ur, xz_bias, mask_ids, mask_T, mask_s0, out, gates, S, B, U, sync, guard_out=None):
        self._call(self.lib.tnt_lstm_seq_fwd_f32, "tnt_lstm_seq_fwd_f32", _p(xz), _p(hs), _p(cs), _p(Ur), _p(xz_bias),
                   _p(mask_ids), mask_T, mask_s0, _p(out), _p(gates), S, B, U, _p(sync), _p(guard_out), self._s())

    def ln_lstm_cell_fwd(self, zk, zr, bias, c_prev, gamma_s, beta_s, gates, chat, istd, c, h, B, U, eps):
        self._call(self.lib.tnt_ln_lstm_cell_fwd_f32, "tnt_ln_lstm_cell_fwd_f32", _p(zk), _p(zr), _p(bias), _p(c_prev), _p(gamma_s),
                   _p(beta_s), _p(gates), _p(chat), _p(istd), _p(c), _p(h), B, U, eps, self._s())

    def ln_lstm_cell_bwd(self, dh_a, dh_b, dh_c, dcn_in, gates, c_prev, c, chat, istd, gamma_s, dz, dc_prev, dcnt, B, U):
        self._call(self.lib.tnt_ln_lstm_cell_bwd_f32, "tnt_ln_lstm_cell_bwd_f32", _p(dh_a), _p(dh_b), _p(dh_c), _p(dcn_in), _p(gates),
                   _p(c_prev), _p(c), _p(chat), _p(istd), _p(gamma_s), _p(dz), _p(dc_prev), _p(dcnt), B, U, self._s())

    def lstm_seq_bwd_work_floats(self, B, U):
        return int(self.lib.tnt_lstm_seq_bwd_work_floats(int(B), int(U)))

    def lstm_seq_bwd(self, Ur, dout_seq, mask_ids, mask_T, mask_s0, gates, cs, dz, work, S, B, U, sync, guard_out=None):
        """BPTT over the S steps of one sequence in ONE persistent launch (tnt_lstm_seq_bwd_f32)."""
        self._call(self.lib.tnt_lstm_seq_bwd_f32, "tnt_lstm_seq_bwd_f32", _p(Ur), _p(dout_seq), _p(mask_ids), mask_T, mask_s0,
                   _p(gates), _p(cs), _p(dz), _p(work), work.numel(), S, B, U, _p(sync), _p(guard_out), self._s())

    def lstm_step_bwd(self, dz_next, Ur, da_pass_in, dh_ext, dc_in, dout_in, dout_t, mask_ids, mask_T, mask_t,
                      gates, c, c_prev, dz, da_pass_out, dc_out, dout_out, B, U, Wc=None, D=0, dctx_part=None):
        self._call(self.lib.tnt_lstm_step_bwd_f32, "tnt_lstm_step_bwd_f32", _p(dz_next), _p(Ur), _p(da_pass_in), _p(dh_ext), _p(dc_in),
                                                  _p(dout_in), _p(dout_t), _p(mask_ids), mask_T, mask_t, _p(gates),
                                                  _p(c), _p(c_prev), _p(dz), _p(da_pass_out), _p(dc_out),
                                                  _p(dout_out), B, U, _p(Wc), D, _p(dctx_part), self._s())

    def gru_step_fwd(self, xz, h_prev, Uk, br, h, gates, B, U):
        self._call(self.lib.tnt_gru_step_fwd_f32, "tnt_gru_step_fwd_f32", _p(xz), _p(h_prev), _p(Uk), _p(br), _p(h), _p(gates), B, U, self._s())

    def gru_step_bwd(self, drec_next, Uk, dh_pass_in, dh_ext, gates, h_prev, dxz, drec, dh_pass_out, B, U):
        self._call(self.lib.tnt_gru_step_bwd_f32, "tnt_gru_step_bwd_f32", _p(drec_next), _p(Uk), _p(dh_pass_in), _p(dh_ext), _p(gates), _p(h_prev),
                                                 _p(dxz), _p(drec), _p(dh_pass_out), B, U, self._s())

    def softmax_cce(self, logits, target, probs, loss_row, correct_row, dlogits, rows, V, ld, gscale,
                    from_logits=False, mask_zero=False):
        self._call(self.lib.tnt_softmax_cce_f32, "tnt_softmax_cce_f32", _p(logits), _p(target), _p(probs), _p(loss_row), _p(correct_row),
                                                _p(dlogits), rows, V, ld, gscale, int(from_logits), int(mask_zero),
                                                self._s())

    def onehot_argmax(self, onehot, ids_tmajor, B, T, V):
        self._call(self.lib.tnt_onehot_argmax_f32, "tnt_onehot_argmax_f32", _p(onehot), _p(ids_tmajor), B, T, V, self._s())

    def argmax_rows(self, x, out, rows, V, ld):
        self._call(self.lib.tnt_argmax_rows_f32, "tnt_argmax_rows_f32", _p(x), _p(out), rows, V, ld, self._s())

    def enc_tail_fwd(self, y, gamma, beta, mov_mean, mov_var, out, xhat, inv_std, rows, C, ldo, training, eps, momentum,
                     r_feat, r_lstm, seed, site_feat, site_lstm, step_dev=None):
        self._call(self.lib.tnt_enc_tail_fwd_f32, "tnt_enc_tail_fwd_f32", _p(y), _p(gamma), _p(beta), _p(mov_mean), _p(mov_var), _p(out), _p(xhat),
                                                 _p(inv_std), rows, C, ldo, int(training), eps, momentum, r_feat, r_lstm,
                                                 int(seed), int(site_feat), int(site_lstm), _p(step_dev), self._s())

    def enc_tail_bwd_drop(self, dout, xhat, gamma, inv_std, pre, dpre, dgamma, dbeta, dbias, rows, C, ldo, r_feat, r_lstm,
                          slope, seed, site_feat, site_lstm, step_dev, drop_x, drop_rows, drop_cols, drop_ld, drop_tmajor_B,
                          drop_lwidth, drop_lcol0, drop_rate, drop_site):
        self._call(self.lib.tnt_enc_tail_bwd_drop_f32, "tnt_enc_tail_bwd_drop_f32", _p(dout), _p(xhat), _p(gamma), _p(inv_std),
                   _p(pre), _p(dpre), _p(dgamma), _p(dbeta), _p(dbias), rows, C, ldo, r_feat, r_lstm, slope, int(seed),
                   int(site_feat), int(site_lstm), _p(step_dev), _p(drop_x), drop_rows, drop_cols, drop_ld, drop_tmajor_B,
                   drop_lwidth, drop_lcol0, drop_rate, int(drop_site), self._s())

    def enc_tail_bwd(self, dout, xhat, gamma, inv_std, pre, dpre, dgamma, dbeta, dbias, rows, C, ldo, r_feat, r_lstm,
                     slope, seed, site_feat, site_lstm, step_dev=None):
        self._call(self.lib.tnt_enc_tail_bwd_f32, "tnt_enc_tail_bwd_f32", _p(dout), _p(xhat), _p(gamma), _p(inv_std), _p(pre), _p(dpre), _p(dgamma),
                                                 _p(dbeta), _p(dbias), rows, C, ldo, r_feat, r_lstm, slope, int(seed),
                                                 int(site_feat), int(site_lstm), _p(step_dev), self._s())

    def dense_fwd_stream(self, x, w, part, B, E, K, ldx, ldw, nsplit):
        self._call(self.lib.tnt_dense_fwd_stream_f32, "tnt_dense_fwd_stream_f32", _p(x), _p(w), _p(part), B, E, K, ldx, ldw,
                   nsplit, self._s())

    def dense_fwd_stream_gram(self, x, w, part, gx_part, w2_part, B, E, K, ldx, ldw, nsplit):
        self._call(self.lib.tnt_dense_fwd_stream_gram_f32, "tnt_dense_fwd_stream_gram_f32", _p(x), _p(w), _p(part), _p(gx_part),
                   _p(w2_part), B, E, K, ldx, ldw, nsplit, self._s())

    def dense_gram_norm(self, dpre, pre, bias, gx_part, nsplit, w2_part, nw2, l2, partial, nslot, Bk, E, spans=None, lr_job=None,
                        skip=None):
        """``spans`` = (theta, grad, span_seg, span_off, span_len, seg_l2, span_partial, nspan): the span norms of the other
        variables in the same launch (tnt_dense_gram_norm_spans_f32); ``lr_job`` = (adam_t, lr, lr_t, beta1, beta2): Adam's
        step size for the update that follows, written by the same launch (tnt_dense_gram_norm_spans_lr_f32); ``skip`` = the
        sq_override table: variables whose clip norm is supplied there get no pass over their gradient (with lr_job only)"""
        if spans is not None and lr_job is not None:
            th, gr, sseg, soff, slen, sl2, spart, nspan = spans
            at, lr, lrt, b1, b2 = lr_job
            self._call(self.lib.tnt_dense_gram_norm_spans_lr_f32, "tnt_dense_gram_norm_spans_lr_f32", _p(dpre), _p(pre), _p(bias),
                       _p(gx_part), nsplit, _p(w2_part), nw2, l2, _p(partial), nslot, Bk, E, _p(th), _p(gr), _p(sseg), _p(soff),
                       _p(slen), _p(sl2), _p(spart), nspan, _p(at), _p(lr), _p(lrt), b1, b2, _p(skip), self._s())
            return
        if spans is None:
            self._call(self.lib.tnt_dense_gram_norm_f32, "tnt_dense_gram_norm_f32", _p(dpre), _p(pre), _p(bias), _p(gx_part), nsplit,
                       _p(w2_part), nw2, l2, _p(partial), nslot, Bk, E, self._s())
            return
        th, gr, sseg, soff, slen, sl2, spart, nspan = spans
        self._call(self.lib.tnt_dense_gram_norm_spans_f32, "tnt_dense_gram_norm_spans_f32", _p(dpre), _p(pre), _p(bias), _p(gx_part),
                   nsplit, _p(w2_part), nw2, l2, _p(partial), nslot, Bk, E, _p(th), _p(gr), _p(sseg), _p(soff), _p(slen), _p(sl2),
                   _p(spart), nspan, self._s())

    def enc_tail_fwd_sk_emb(self, part, nsplit, bias, pre, slope, gamma, beta, mov_mean, mov_var, out, xhat, inv_std, rows, C,
                            ldo, training, eps, momentum, r_feat, r_lstm, seed, site_feat, site_lstm, step_dev, emb_table,
                            emb_ids, emb_out, emb_B, emb_T, emb_V, emb_rate, emb_site):
        self._call(self.lib.tnt_enc_tail_fwd_sk_emb_f32, "tnt_enc_tail_fwd_sk_emb_f32", _p(part), nsplit, _p(bias), _p(pre), slope,
                   _p(gamma), _p(beta), _p(mov_mean), _p(mov_var), _p(out), _p(xhat), _p(inv_std), rows, C, ldo,
                   int(training), eps, momentum, r_feat, r_lstm, int(seed), int(site_feat), int(site_lstm), _p(step_dev),
                   _p(emb_table), _p(emb_ids), _p(emb_out), emb_B, emb_T, emb_V, emb_rate, int(emb_site), self._s())

    def enc_tail_fwd_sk(self, part, nsplit, bias, pre, slope, gamma, beta, mov_mean, mov_var, out, xhat, inv_std, rows, C,
                        ldo, training, eps, momentum, r_feat, r_lstm, seed, site_feat, site_lstm, step_dev=None):
        self._call(self.lib.tnt_enc_tail_fwd_sk_f32, "tnt_enc_tail_fwd_sk_f32", _p(part), nsplit, _p(bias), _p(pre), slope,
                   _p(gamma), _p(beta), _p(mov_mean), _p(mov_var), _p(out), _p(xhat), _p(inv_std), rows, C, ldo,
                   int(training), eps, momentum, r_feat, r_lstm, int(seed), int(site_feat), int(site_lstm), _p(step_dev),
                   self._s())

    def dense_dw_sqnorm(self, x, dpre, theta, l2, partial, nslot, N, E, Bk, ldx):
        self._call(self.lib.tnt_dense_dw_sqnorm_f32, "tnt_dense_dw_sqnorm_f32", _p(x), _p(dpre), _p(theta), l2, _p(partial),
                   nslot, N, E, Bk, ldx, self._s())

    def dense_dw_adam(self, x, dpre, theta, m, v, l2, sq, sq_override, lr_t_dev, beta1, beta2, eps, clipnorm, N, E, Bk, ldx,
                      guard=None):
        self._call(self.lib.tnt_dense_dw_adam_f32, "tnt_dense_dw_adam_f32", _p(x), _p(dpre), _p(theta), _p(m), _p(v), l2,
                   _p(sq), _p(sq_override), _p(lr_t_dev), beta1, beta2, eps, clipnorm, _p(guard), N, E, Bk, ldx, self._s())

    def dense_dw_skinny(self, x, dpre, dw, N, E, Bk, ldx):
        self._call(self.lib.tnt_dense_dw_skinny_f32, "tnt_dense_dw_skinny_f32", _p(x), _p(dpre), _p(dw), N, E, Bk, ldx, self._s())

    def lc_seq_fwd(self, F, P, W2, b2, v, bv, qpre, alpha, ctx, ctx_d, keep4, keep_stride, xz, Wc, Ur, xz_bias, hs, cs, gates, T,
                   B, R, D, A, U, slope, rate_attn, rate_in, in_lwidth, seed, site_attn0, site_in0, step_dev, work, sync,
                   guard_out=None, out_drop=None):
        """out_drop = (hd, rate, site0): the Dropout behind the LSTM rides in the chain (tnt_lc_seq_fwd_drop_f32);
        work: lc_seq_fwd_work_floats(B) floats"""
        if out_drop is not None:
            hd, rate_out, site_out0 = out_drop
            self._call(self.lib.tnt_lc_seq_fwd_drop_f32, "tnt_lc_seq_fwd_drop_f32", _p(F), _p(P), _p(W2), _p(b2), _p(v), _p(bv),
                       _p(qpre), _p(alpha), _p(ctx), _p(ctx_d), _p(keep4), int(keep_stride), _p(xz), _p(Wc), _p(Ur), _p(xz_bias),
                       _p(hs), _p(cs), _p(gates), T, B, R, D, A, U, slope, rate_attn, rate_in, in_lwidth, int(seed),
                       int(site_attn0), int(site_in0), _p(step_dev), _p(hd), float(rate_out), int(site_out0), _p(work), _p(sync),
                       _p(guard_out), self._s())
            return
        self._call(self.lib.tnt_lc_seq_fwd_f32, "tnt_lc_seq_fwd_f32", _p(F), _p(P), _p(W2), _p(b2), _p(v), _p(bv), _p(qpre), _p(alpha),
                   _p(ctx), _p(ctx_d), _p(keep4), int(keep_stride), _p(xz), _p(Wc), _p(Ur), _p(xz_bias), _p(hs), _p(cs), _p(gates),
                   T, B, R, D, A, U, slope, rate_attn, rate_in, in_lwidth, int(seed), int(site_attn0), int(site_in0),
                   _p(step_dev), _p(work), _p(sync), _p(guard_out), self._s())

    def lc_seq_fwd_work_floats(self, B):
        return int(self.lib.tnt_lc_seq_fwd_work_floats(int(B)))

    def lc_seq_bwd_work_floats(self, B, U):
        return int(self.lib.tnt_lc_seq_bwd_work_floats(int(B), int(U)))

    def lc_seq_bwd(self, F, P, W2, v, qpre, alpha, keep4, keep_stride, dP, dF, dvb, dqpre, Ur, Wc, dout, gates, cs, dz, work, T,
                   B, R, D, A, U, slope, rate_attn, rate_in, in_lwidth, seed, site_attn0, site_in0, step_dev, alpha_mse, sync,
                   guard_out=None, out_drop=None):
        """out_drop = (rate, site0): dout is the gradient of the DROPPED outputs (tnt_lc_seq_bwd_drop_f32)."""
        if out_drop is not None:
            rate_out, site_out0 = out_drop
            self._call(self.lib.tnt_lc_seq_bwd_drop_f32, "tnt_lc_seq_bwd_drop_f32", _p(F), _p(P), _p(W2), _p(v), _p(qpre),
                       _p(alpha), _p(keep4), int(keep_stride), _p(dP), _p(dF), _p(dvb), _p(dqpre), _p(Ur), _p(Wc), _p(dout),
                       _p(gates), _p(cs), _p(dz), _p(work), T, B, R, D, A, U, slope, rate_attn, rate_in, in_lwidth, int(seed),
                       int(site_attn0), int(site_in0), _p(step_dev), float(alpha_mse), float(rate_out), int(site_out0),
                       _p(sync), _p(guard_out), self._s())
            return
        self._call(self.lib.tnt_lc_seq_bwd_f32, "tnt_lc_seq_bwd_f32", _p(F), _p(P), _p(W2), _p(v), _p(qpre), _p(alpha), _p(keep4),
                   int(keep_stride), _p(dP), _p(dF), _p(dvb), _p(dqpre), _p(Ur), _p(Wc), _p(dout), _p(gates), _p(cs), _p(dz),
                   _p(work), T, B, R, D, A, U, slope, rate_attn, rate_in, in_lwidth, int(seed), int(site_attn0), int(site_in0),
                   _p(step_dev), float(alpha_mse), _p(sync), _p(guard_out), self._s())

    def attention_front_bwd_parts(self, rows, D, A):
        return int(self.lib.tnt_attention_front_bwd_parts(rows, D, A))

    def attention_front_bwd(self, Ppre, dP, F, W1, dF, dW1, db1, part, rows, D, A, slope=0.2, drop=None):
        """drop = (rate, seed, site, step_dev): Dropout' of the feature Dropout applied to the finished dF in the same pass"""
        if drop is not None and drop[0] > 0:
            rate, seed, site, step_dev = drop
            self._call(self.lib.tnt_attention_front_bwd_drop_f32, "tnt_attention_front_bwd_drop_f32", _p(Ppre), _p(dP), _p(F),
                       _p(W1), _p(dF), _p(dW1), _p(db1), _p(part), rows, D, A, slope, float(rate), int(seed), int(site),
                       _p(step_dev), self._s())
            return
        self._call(self.lib.tnt_attention_front_bwd_f32, "tnt_attention_front_bwd_f32", _p(Ppre), _p(dP), _p(F), _p(W1), _p(dF),
                   _p(dW1), _p(db1), _p(part), rows, D, A, slope, self._s())

    def gemm_lt(self, A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None):
        self._call(self.lib.tnt_gemm_lt_f32, "tnt_gemm_lt_f32", _p(A), _p(B), _p(C), _p(bias), M, N, K, lda, ldb, ldc, int(transA),
                   int(transB), self._s())

    def gemm_blas(self, A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, accumulate=False):
        self._call(self.lib.tnt_gemm_blas_f32, "tnt_gemm_blas_f32", _p(A), _p(B), _p(C), M, N, K, lda, ldb, ldc, int(transA), int(transB),
                                              int(accumulate), self._s())

    def locally_dense_fwd_split(self, x, ldx, idx, vgoff, vreg, rfirst, NV, W, bias, pre, y, partial, B, R, D, slope=0.2,
                                voxel_major=False):
        self._call(self.lib.tnt_locally_dense_fwd_split_f32, "tnt_locally_dense_fwd_split_f32", _p(x), ldx, _p(idx), _p(vgoff), _p(vreg), _p(rfirst), NV, _p(W),
                                                            _p(bias), _p(pre), _p(y), _p(partial), B, R, D, slope,
                                                            int(voxel_major), self._s())

    def locally_dense_bwd_split(self, x, ldx, idx, vgoff, vreg, vfirst, NV, dpre, dW, db, B, R, D, voxel_major=False):
        self._call(self.lib.tnt_locally_dense_bwd_split_f32, "tnt_locally_dense_bwd_split_f32", _p(x), ldx, _p(idx), _p(vgoff), _p(vreg), _p(vfirst), NV,
                                                            _p(dpre), _p(dW), _p(db), B, R, D, int(voxel_major), self._s())

    def sum2(self, x0, out0, x1, out1, n, scale):
        self._call(self.lib.tnt_sum2_f32, "tnt_sum2_f32", _p(x0), _p(out0), _p(x1), _p(out1), n, scale, self._s())

    def stage_batch(self, x, x_dst, cap, cap_dst, tgt, tgt_tmajor, a0, h0, c0, c0_dst, B, T, N, ldx, U, xT_dst=None, ldt=0,
                    masks=None):
        """x: float32 betas, or float16 ("fp16 on-wire": widened to float by the staging kernel).
        masks = (out, n, nsites, rate, seed, site0, step_dev): dropout_mask4's job in the same launch (float32 betas)."""
        half = x.dtype == torch.float16
        if masks is not None and not half:
            out, n, nsites, rate, seed, site0, step_dev = masks
            self._call(self.lib.tnt_stage_batch_masks_f32, "tnt_stage_batch_masks_f32", _p(x), _p(x_dst), _p(cap), _p(cap_dst),
                       _p(tgt), _p(tgt_tmajor), _p(a0), _p(h0), _p(c0), _p(c0_dst), B, T, N, ldx, U, _p(xT_dst), ldt, _p(out),
                       int(n), int(nsites), float(rate), int(seed), int(site0), _p(step_dev), self._s())
            return
        if masks is not None:
            self.dropout_mask4(masks[0], masks[1], masks[2], masks[3], masks[4], masks[5], 0, masks[6])
        fn, name = (self.lib.tnt_stage_batch_h16, "tnt_stage_batch_h16") if half else \
                   (self.lib.tnt_stage_batch_f32, "tnt_stage_batch_f32")
        self._call(fn, name, _p(x), _p(x_dst), _p(cap), _p(cap_dst), _p(tgt), _p(tgt_tmajor), _p(a0),
                                                _p(h0), _p(c0), _p(c0_dst), B, T, N, ldx, U, _p(xT_dst), ldt, self._s())

    def sample_rows(self, x, out, rows, V, ld, temperature, from_logits, seed, site, step, step_dev=None):
        self._call(self.lib.tnt_sample_rows_f32, "tnt_sample_rows_f32", _p(x), _p(out), rows, V, ld, float(temperature), int(from_logits),
                                                int(seed), int(site), int(step), _p(step_dev), self._s())

    def sum(self, x, out, n, scale):
        self._call(self.lib.tnt_sum_f32, "tnt_sum_f32", _p(x), _p(out), n, scale, self._s())

    def seg_sqnorm(self, theta, grad, span_seg, span_off, span_len, seg_first, seg_l2, partial, sq, wsq, l2_out,
                   nspan, nseg):
        self._call(self.lib.tnt_seg_sqnorm_f32, "tnt_seg_sqnorm_f32", _p(theta), _p(grad), _p(span_seg), _p(span_off), _p(span_len),
                                               _p(seg_first), _p(seg_l2), _p(partial), _p(sq), _p(wsq), _p(l2_out),
                                               nspan, nseg, self._s())

    def l2_total(self, wsq, seg_l2, nseg, out):
        self._call(self.lib.tnt_l2_total_f32, "tnt_l2_total_f32", _p(wsq), _p(seg_l2), nseg, _p(out), self._s())

    def adam(self, theta, m, v, grad, span_seg, span_off, span_len, seg_l2, sq, sq_override, nspan, lr_t, lr_t_dev,
             beta1, beta2, eps, clipnorm, guard=None, met=None, ring=None, ring_t=None):
        if ring is not None:       # ... and the step's metrics vector filed in the metrics ring by the same launch
            self._call(self.lib.tnt_adam_ring_f32, "tnt_adam_ring_f32", _p(theta), _p(m), _p(v), _p(grad), _p(span_seg), _p(span_off),
                       _p(span_len), _p(seg_l2), _p(sq), _p(sq_override), nspan, lr_t, _p(lr_t_dev), beta1, beta2, eps, clipnorm,
                       _p(guard), _p(met), met.numel(), _p(ring), ring.shape[0], _p(ring_t), self._s())
            return
        self._call(self.lib.tnt_adam_f32, "tnt_adam_f32", _p(theta), _p(m), _p(v), _p(grad), _p(span_seg), _p(span_off), _p(span_len),
                                         _p(seg_l2), _p(sq), _p(sq_override), nspan, lr_t, _p(lr_t_dev), beta1,
                                         beta2, eps, clipnorm, _p(guard), self._s())

    def sgd(self, theta, mom, grad, span_seg, span_off, span_len, seg_l2, sq, sq_override, nspan, lr, lr_dev,
            momentum, clipnorm, guard=None):
        self._call(self.lib.tnt_sgd_f32, "tnt_sgd_f32", _p(theta), _p(mom), _p(grad), _p(span_seg), _p(span_off), _p(span_len),
                                        _p(seg_l2), _p(sq), _p(sq_override), nspan, lr, _p(lr_dev), momentum,
                                        clipnorm, _p(guard), self._s())

    def sam(self, theta, grad, ew, span_seg, span_off, span_len, seg_l2, sq, nseg, nspan, rho, mode, sq_override=None):
        self._call(self.lib.tnt_sam_f32, "tnt_sam_f32", _p(theta), _p(grad), _p(ew), _p(span_seg), _p(span_off), _p(span_len),
                                        _p(seg_l2), _p(sq), _p(sq_override), nseg, nspan, rho, mode, self._s())

    def sqdiff_mean(self, x, out, n, c):
        self._call(self.lib.tnt_sqdiff_mean_f32, "tnt_sqdiff_mean_f32", _p(x), _p(out), n, float(c), self._s())

    def span_sqnorm(self, theta, grad, span_seg, span_off, span_len, seg_l2, partial, nspan):
        self._call(self.lib.tnt_span_sqnorm_f32, "tnt_span_sqnorm_f32", _p(theta), _p(grad), _p(span_seg), _p(span_off), _p(span_len),
                   _p(seg_l2), _p(partial), nspan, self._s())

    @staticmethod
    def finalize_desc(partial, seg_first, seg_l2, sq, wsq, l2_out, nseg, arrive, x0=None, out0=None, x1=None, out1=None, n=0,
                      scale=1.0, extra_part=None, extra=None, n_extra=0, extra_seg=-1, ids_src=None, ids_dst=None, n_ids=0,
                      adam_t=None, drop_step=None, lr=None, lr_t=None, beta1=0.0, beta2=0.0, guard=None, x2=None, out2=None,
                      n2=0, scale2=1.0):
        """tnt_finalize_desc for adam_fin; the returned object keeps the tensors alive"""
        d = _FinDesc(_p(partial), _p(seg_first), _p(seg_l2), _p(sq), _p(wsq), _p(l2_out), nseg, _p(x0), _p(out0), _p(x1), _p(out1), n,
                     scale, _p(extra_part), _p(extra), n_extra, extra_seg, _p(ids_src), _p(ids_dst), n_ids, _p(x2), _p(out2), n2,
                     scale2, _p(adam_t), _p(drop_step), _p(lr), _p(lr_t), beta1, beta2, _p(guard), _p(arrive))
        d._keep = (partial, seg_first, seg_l2, sq, wsq, l2_out, x0, out0, x1, out1, extra_part, extra, ids_src, ids_dst, x2, out2,
                   adam_t, drop_step, lr, lr_t, guard, arrive)
        return d

    def adam_fin(self, theta, m, v, grad, span_seg, span_off, span_len, sq_override, nspan, eps, clipnorm, fin, met=None, ring=None,
                 ring_t=None):
        """clip + Adam with the step's scalar tail inside the launch (fin = finalize_desc(...)); the counters tick at its end"""
        self._call(self.lib.tnt_adam_fin_f32, "tnt_adam_fin_f32", _p(theta), _p(m), _p(v), _p(grad), _p(span_seg), _p(span_off),
                   _p(span_len), _p(sq_override), nspan, eps, clipnorm, _ct.addressof(fin), _p(met), met.numel() if met is not None else 0,
                   _p(ring), ring.shape[0] if ring is not None else 0, _p(ring_t), self._s())

    def span_sqnorm_lr(self, theta, grad, span_seg, span_off, span_len, seg_l2, partial, nspan, adam_t, lr, lr_t, beta1, beta2,
                       skip=None):
        """``skip`` = the sq_override table (or None): see tnt_span_sqnorm_lr_f32"""
        self._call(self.lib.tnt_span_sqnorm_lr_f32, "tnt_span_sqnorm_lr_f32", _p(theta), _p(grad), _p(span_seg), _p(span_off),
                   _p(span_len), _p(seg_l2), _p(partial), nspan, _p(adam_t), _p(lr), _p(lr_t), beta1, beta2, _p(skip), self._s())

    def dense_dw_adam_fin(self, x, dpre, theta, m, v, l2, partial, k0, k1, sq_override, lr_t_dev, beta1, beta2, eps, clipnorm, N, E,
                          Bk, ldx, guard=None):
        self._call(self.lib.tnt_dense_dw_adam_fin_f32, "tnt_dense_dw_adam_fin_f32", _p(x), _p(dpre), _p(theta), _p(m), _p(v), l2,
                   _p(partial), k0, k1, _p(sq_override), _p(lr_t_dev), beta1, beta2, eps, clipnorm, _p(guard), N, E, Bk, ldx, self._s())

    def step_finalize(self, partial, seg_first, seg_l2, sq, wsq, l2_out, nseg, x0=None, out0=None, x1=None, out1=None, n=0,
                      scale=1.0, extra_part=None, extra=None, n_extra=0, ids_src=None, ids_dst=None, n_ids=0, adam_t=None,
                      drop_step=None, lr=None, lr_t=None, beta1=0.0, beta2=0.0, guard=None, x2=None, out2=None, n2=0,
                      scale2=1.0):
        self._call(self.lib.tnt_step_finalize_f32, "tnt_step_finalize_f32", _p(partial), _p(seg_first), _p(seg_l2), _p(sq), _p(wsq),
                   _p(l2_out), nseg, _p(x0), _p(out0), _p(x1), _p(out1), n, scale, _p(extra_part), _p(extra), n_extra,
                   _p(ids_src), _p(ids_dst), n_ids, _p(x2), _p(out2), n2, scale2, _p(adam_t), _p(drop_step), _p(lr), _p(lr_t), beta1, beta2, _p(guard),
                   self._s())

    def embedding_bwd_parts(self, B, T, E):
        return int(self.lib.tnt_embedding_bwd_parts(B, T, E))

    def embedding_bwd_sparse(self, drows, ids, prev_ids, dtable, sq_part, B, T, E, ldd, V, drop_rate=0.0, drop_seed=0,
                             drop_site=0, drop_step_dev=None, zero_id=-1):
        """``zero_id`` >= 0: the caller guarantees that the rows of that id are zero (mask_zero + masked LSTM); they are not read"""
        self._call(self.lib.tnt_embedding_bwd_sparse_f32, "tnt_embedding_bwd_sparse_f32", _p(drows), _p(ids), _p(prev_ids), _p(dtable),
                   _p(sq_part), B, T, E, ldd, V, drop_rate, int(drop_seed), int(drop_site), _p(drop_step_dev), int(zero_id), self._s())

    def agc(self, theta, grad, tab, gsq_cols=None, sq_out=None, clip_factor=0.01, eps=1e-3):
        """unit-wise adaptive gradient clipping over the arena; ``tab`` = arena.AgcTable"""
        emb = tab.emb if gsq_cols is not None else None
        self._call(self.lib.tnt_agc_f32, "tnt_agc_f32", _p(theta), _p(grad), _p(tab.var_off), _p(tab.var_ld), _p(tab.var_lam),
                   _p(tab.item), _p(tab.cb_first), tab.nitem, _p(tab.partial), _p(gsq_cols), emb[0] if emb else -1,
                   emb[1] if emb else 0, emb[2] if emb else 0, _p(tab.sq_part), _p(sq_out), clip_factor, eps, self._s())

    def colsq(self, x, out, rows, cols, ld):
        self._call(self.lib.tnt_colsq_f32, "tnt_colsq_f32", _p(x), _p(out), rows, cols, ld, self._s())

    def beam_topk(self, probs, score_in, fin_in, B, V, ld, k, end_id, score_out, parent, token, fin_out):
        self._call(self.lib.tnt_beam_topk_f32, "tnt_beam_topk_f32", _p(probs), _p(score_in), _p(fin_in), B, V, ld, k, end_id,
                   _p(score_out), _p(parent), _p(token), _p(fin_out), self._s())

    def step_tick(self, adam_t, drop_step, lr, lr_t, beta1, beta2, guard=None):
        self._call(self.lib.tnt_step_tick, "tnt_step_tick", _p(adam_t), _p(drop_step), _p(lr), _p(lr_t), beta1, beta2, _p(guard),
                   self._s())

    def block_dense_dx(self, dpre, W, dx, B, R, Din, Dout):
        self._call(self.lib.tnt_block_dense_dx_f32, "tnt_block_dense_dx_f32", _p(dpre), _p(W), _p(dx), B, R, Din, Dout, self._s())

    def locally_dense_fwd(self, x, ldx, idx, goff, W, bias, pre, y, B, R, D, slope=0.2):
        self._call(self.lib.tnt_locally_dense_fwd_f32, "tnt_locally_dense_fwd_f32", _p(x), ldx, _p(idx), _p(goff), _p(W), _p(bias), _p(pre), _p(y),
                                                      B, R, D, slope, self._s())

    def locally_dense_bwd(self, x, ldx, idx, goff, dpre, dW, db, B, R, D):
        self._call(self.lib.tnt_locally_dense_bwd_f32, "tnt_locally_dense_bwd_f32", _p(x), ldx, _p(idx), _p(goff), _p(dpre), _p(dW), _p(db), B, R,
                                                      D, self._s())

    def attention_step_fwd(self, h, F, P, W2, b2, v, bv, qpre, alpha, ctx, ctx_d, s_out, B, R, D, A, U, slope,
                           rate_attn, rate_in, in_lwidth, seed, site_attn, site_in, step, step_dev=None, keep4=None):
        self._call(self.lib.tnt_attention_step_fwd_f32, "tnt_attention_step_fwd_f32", _p(h), _p(F), _p(P), _p(W2), _p(b2), _p(v), _p(bv), _p(qpre),
                                                       _p(alpha), _p(ctx), _p(ctx_d), _p(s_out), B, R, D, A, U, slope,
                                                       rate_attn, rate_in, in_lwidth, seed, site_attn, site_in, step,
                                                       _p(step_dev), _p(keep4), self._s())

    def attention_step_bwd(self, dctx_d, F, P, W2, v, qpre, alpha, dP, dF, dvb, dqpre, dh, B, R, D, A, U, slope,
                           rate_attn, rate_in, in_lwidth, seed, site_attn, site_in, step, step_dev=None, dz=None,
                           Wc=None, dctx_part=None, nparts=0, keep4=None, alpha_mse=0.0, fresh=False):
        self._call(self.lib.tnt_attention_step_bwd_f32, "tnt_attention_step_bwd_f32", _p(dctx_d), _p(F), _p(P), _p(W2), _p(v), _p(qpre), _p(alpha),
                                                       _p(dP), _p(dF), _p(dvb), _p(dqpre), _p(dh), B, R, D, A, U,
                                                       slope, rate_attn, rate_in, in_lwidth, seed, site_attn, site_in,
                                                       step, _p(step_dev), _p(dz), _p(Wc), _p(dctx_part), nparts,
                                                       _p(keep4), float(alpha_mse), int(bool(fresh)), self._s())

    def attention_metric_parts(self, T, R):
        return int(self.lib.tnt_attention_metric_parts(T, R))

    def attention_metric(self, alpha, out, work, T, B, R, tstride=0):
        self._call(self.lib.tnt_attention_metric_f32, "tnt_attention_metric_f32", _p(alpha), _p(out), _p(work), T, B, R, tstride, self._s())


_backend = None


def backend():
    """The active backend; created on first use.  Raises (loudly) if the HIP library or the
    GPU is missing -- there is no silent fallback."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def set_backend(b):
    """TEST HOOK ONLY: install a stand-in backend (tests/mock_backend.py)."""
    global _backend
    _backend = b
