/*
 * tnt_hip.h -- C ABI of the MI355X (gfx950) kernel library for the fMRI->caption
 * training hot path ("Think and Tell", seang123/Masters-Thesis).
 *
 * The reference has no FFI: its arithmetic is TensorFlow/Keras called from Python
 * (SURVEY.md 8b).  Each entry point below replaces the Keras op(s) at the cited
 * reference lines (paths relative to the reference repo root).  The reference-side
 * binding a maintainer would add is a ctypes stub; see INTEGRATION.md.
 *
 * Conventions
 *   - all tensors float32 unless noted; token ids int32; plain device pointers;
 *     the caller owns every buffer and workspace; no allocation, no global state
 *     (one documented exception: the two vendor-library A/B entry points tnt_gemm_blas_f32 /
 *     tnt_gemm_lt_f32 keep a process-wide rocBLAS / hipBLASLt handle and per-shape plans behind a
 *     mutex, csrc/blas.hip; no default training path calls them);
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work;
 *   - return 0 on success, a negative hipError_t on launch failure, -1000-k for
 *     an argument the kernels do not support (k = argument position);
 *   - matrices are row-major with an explicit leading dimension (in floats);
 *   - sequence activations are time-major: row = t*B + b;
 *   - LSTM tensors use the gate-interleaved axis order [.., U, 4] (i,f,c~,o
 *     innermost) instead of keras' [.., 4U] blocks; the host layer converts in
 *     get_weights/set_weights;
 *   - dropout masks come from the Philox4x32-10 stream of csrc/tnt_rng.h:
 *     element e of the *logical* tensor, (seed, site, step) identify the mask.
 */
#ifndef TNT_HIP_H
#define TNT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNT_ACT_NONE 0
#define TNT_ACT_LEAKY 1 /* x>0 ? x : slope*x   (LeakyReLU(0.2): lc_NIC.py:87,98,142) */
#define TNT_ACT_RELU 2
#define TNT_ACT_TANH 3

int32_t tnt_version(void);

/* ---- GEMM: C[M,N] (+)= act(op(A)[M,K] * op(B)[K,N] + bias[N]) -------------------
 * transA=0: A is [M][lda]; transA=1: A is stored [K][lda] (A^T), i.e. m contiguous.
 * transB=0: B is [K][ldb]; transB=1: B is stored [N][ldb] (B^T), i.e. k contiguous.
 * Replaces keras Dense / TimeDistributed(Dense) forward and the matmuls of
 * tape.gradient: layers.py:33, attention.py:21-23, lc_NIC.py:140-157,261,386-387,
 * NIC.py:64-69,92-96,143,248-249.  `pre` (nullable) receives the pre-activation.
 * accumulate=1: C += result (bias/act must then be none).
 * splitk>1: `work` must hold splitk*M*N floats; partials are reduced by a second
 * launch that applies bias/act (fixed order: bitwise reproducible). */
int32_t tnt_gemm_f32(const float* A, const float* B, float* C, const float* bias, float* pre,
                     int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc,
                     int32_t transA, int32_t transB, int32_t act, float slope,
                     int32_t accumulate, int32_t splitk, float* work, void* stream);
/* Plain library GEMM (rocBLAS sgemm, exact f32, atomics disabled => bitwise reproducible), same operand conventions as
 * tnt_gemm_f32 without the epilogue arguments: C = op(A) op(B) (+ C if accumulate).  For the matmuls that carry no
 * fused epilogue (weight / input gradients); the fused ones use tnt_gemm_f32.  The first call creates the process-wide
 * rocBLAS handle, and the first call of each shape loads its kernel -- make those outside a stream capture. */
int32_t tnt_gemm_blas_f32(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K,
                          int32_t lda, int32_t ldb, int32_t ldc, int32_t transA, int32_t transB,
                          int32_t accumulate, void* stream);

/* The same product (no accumulate) through hipBLASLt in full FP32 (HIPBLAS_COMPUTE_32F), optional bias [N] added to every
 * row of C (the Dense layers' bias, NIC.py:143).  The algorithm is the heuristic's first workspace-free candidate for the
 * shape (reproducible); TNT_LT_TUNE=1 in the environment times the candidates at the first un-captured call instead. */
int32_t tnt_gemm_lt_f32(const float* A, const float* B, float* C, const float* bias, int32_t M, int32_t N, int32_t K,
                        int32_t lda, int32_t ldb, int32_t ldc, int32_t transA, int32_t transB, void* stream);

/* One-round GEMM family for the products that carry no activation epilogue -- the weight / input gradients that
 * tape.gradient derives from the Dense / LSTM layers (lc_NIC.py:386-387, NIC.py:248-249) and the LSTM input projection
 * (NIC.py:138-140): C = op(A) op(B) (+ bias[N]), operand conventions of tnt_gemm_f32.  The workgroup tile is picked so
 * that the whole output is ONE round of at most 256 workgroups (one per CU); FP32 MFMA 16x16x4, exact f32, a fixed
 * summation order (bitwise reproducible).  Fused riders:
 *   colsum (nullable; transA=1, transB=0 only): colsum[n] = sum_k B[k][n] -- the bias gradient of the layer whose
 *     kernel gradient this product is (B = dY), computed from the B tiles the first row of workgroups loads anyway;
 *   A2 / C2 (both or neither): a second product C2 = op(A2) op(B) with the same dims and strides in the same launch
 *     (the LSTM's kernel and recurrent-kernel gradients share dZ).
 * cfg = 0: automatic; > 0: force configuration `cfg` of the table in csrc/gemm.hip (tools/gemm_cfg_scan.py).
 * Returns a negative code when no one-round configuration exists for the shape (the caller then uses tnt_gemm_f32). */
int32_t tnt_gemm_fused_f32(const float* A, const float* B, float* C, const float* bias, float* colsum,
                           const float* A2, float* C2, int32_t M, int32_t N, int32_t K, int32_t lda,
                           int32_t ldb, int32_t ldc, int32_t transA, int32_t transB, int32_t cfg, void* stream);
/* the configuration tnt_gemm_fused_f32 would pick (0 = none) */
int32_t tnt_gemm_fused_cfg(int32_t M, int32_t N, int32_t K, int32_t transA, int32_t transB, int32_t batch);

/* ---- gemm3 (round 3): the hand-written FP32-MFMA family that carries the step's large products by default --
 * the vocabulary head forward (NIC.py:143, lc_NIC.py:261) and, under tape.gradient (NIC.py:248-249, lc_NIC.py:386-387),
 * its kernel / input gradients, the LSTM input projection (NIC.py:138-140) and the LSTM kernel / recurrent-kernel / input
 * gradients.  C = op(A) op(B) (+ bias[N]); operand conventions of tnt_gemm_f32; A, B, C 16-byte aligned, lda / ldb / ldc
 * multiples of 4; a K-contiguous operand (A with transA = 0, B with transB = 1) whose K is not a multiple of 4 must have
 * zeros in the pad columns [K, roundup4(K)) (the layout contract of every buffer of this library).  Exact f32
 * (v_mfma_f32_16x16x4_f32), fixed summation order: bitwise reproducible.  cfg selects the workgroup tile (table in
 * csrc/gemm3.hip; tools/gemm3_scan.py). */
int32_t tnt_gemm3_f32(const float* A, const float* B, float* C, const float* bias, float* colsum, const float* A2,
                      float* C2, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t transA,
                      int32_t transB, int32_t tile, int32_t splitk, float* work, uint32_t* sync, void* stream);
/* Riders (all nullable): bias [N] added to every row; colsum [N] = sum_k B[k][n] (transA = 1, transB = 0, splitk = 1 only:
 * B = dY, so this is the bias gradient of the layer whose kernel gradient the product is); A2 / C2 (both or neither): a
 * second product C2 = op(A2) op(B) with the same dims and strides in the same launch (the LSTM's kernel and
 * recurrent-kernel gradients share dZ).
 * splitk > 1: K is split over `splitk` workgroups per output tile, which reduce inside the launch (fixed order).  `work`:
 * tnt_gemm3_work_floats(M, N, tile, splitk, batch) floats, ARMED once with tnt_gemm3_work_arm (every word = the "not
 * written yet" pattern) -- every launch leaves it armed, so launches of any shape may share one buffer as long as they run
 * one after the other; `sync`: tnt_gemm3_sync_words(...) uint32 words (one: an error flag, zero before the first use;
 * nonzero afterwards means a workgroup gave up waiting for a peer -- the launch did not fit the device in one round -- and
 * C is invalid).
 * batch = 2 with A2 / C2, else 1. */
/* The (tile, splitk) the library's cost model picks for a shape (batch = 2 for a dual launch; allow_split = 0 restricts
 * the choice to splitk = 1, which the colsum rider needs). */
int32_t tnt_gemm3_plan(int32_t M, int32_t N, int32_t K, int32_t transA, int32_t transB, int32_t batch,
                       int32_t allow_split, int32_t* tile, int32_t* splitk);
int32_t tnt_gemm3_work_floats(int32_t M, int32_t N, int32_t tile, int32_t splitk, int32_t batch);
int32_t tnt_gemm3_sync_words(int32_t M, int32_t N, int32_t tile, int32_t batch);
int32_t tnt_gemm3_work_arm(float* work, int64_t floats, void* stream);

/* Two INDEPENDENT products in one launch (the second one's workgroups start on the CUs the first one's tail leaves idle;
 * a launch boundary costs a 25-50 us GEMM 4-6 us of ramp-up and drain): fields = the arguments of tnt_gemm3_f32.  The
 * backward pass has such pairs back to back: the head's kernel gradient + input gradient (both read dlogits; NIC.py:143
 * under tape.gradient), the LSTM's kernel gradients + input gradient (both read dZ; NIC.py:138-140).  Supported
 * combinations: tnt_gemm3_pair_supported (else TNT_BADARG: issue two tnt_gemm3_f32 calls).  Two split products need
 * distinct `work` buffers. */
typedef struct tnt_gemm3_desc {
  const float* A; const float* B; float* C; const float* bias; float* colsum; const float* A2; float* C2;
  int32_t M, N, K, lda, ldb, ldc, transA, transB, tile, splitk;
  float* work; uint32_t* sync;
} tnt_gemm3_desc;
int32_t tnt_gemm3_pair_supported(int32_t tile1, int32_t transA1, int32_t transB1, int32_t tile2, int32_t transA2,
                                 int32_t transB2);
int32_t tnt_gemm3_pair_f32(const tnt_gemm3_desc* p, const tnt_gemm3_desc* q, void* stream);

/* ---- the optimizer step without a finalize launch (single-process step; optimizer.apply_gradients with clipnorm + Adam,
 * main.py:97, lc_NIC.py:389).  tnt_step_finalize_f32 -- per-variable norms from the span partials, the step's scalar totals,
 * the counter tick -- was one dependent launch in front of the update: 12 us of a 500 us step.  Here
 *   tnt_span_sqnorm_lr_f32 / tnt_dense_gram_norm_spans_lr_f32: the norm launches, with one thread also writing Adam's step
 *       size lr_t = lr sqrt(1-b2^t)/(1-b1^t), t = *adam_t + 1, for the update that follows.  `sq_override` (may be NULL:
 *       everything is read): the per-variable table of externally supplied clip norms the update launches take; a variable
 *       with sq_override[seg] >= 0 (the Embedding: IndexedSlices norm, lc_NIC.py:386-389) gets no pass over its gradient,
 *       and sum theta^2 -- consumed only as lambda * sum theta^2 of the L2 metric -- is left 0 where seg_l2[seg] == 0;
 *   tnt_dense_dw_adam_fin_f32: tnt_dense_dw_adam_f32 with the kernel's clip norm summed from partial[2k], k0 <= k < k1, by
 *       the launch itself;
 *   tnt_adam_fin_f32: tnt_adam_ring_f32 over spans whose variables' clip norms are summed from `fin->partial` by every
 *       workgroup that needs one (same order everywhere); ONE extra workgroup files everything tnt_step_finalize_f32 wrote
 *       (fields of tnt_finalize_desc = its arguments; extra_seg = the variable whose norm is sum(extra_part), -1: none;
 *       `arrive`: one zeroed uint32, left zero) beside the update; *adam_t and *drop_step advance when the LAST workgroup
 *       of the launch is done (unless *guard != 0).  Issue order: norm launch, tnt_dense_dw_adam_fin_f32, tnt_adam_fin_f32. */
typedef struct tnt_finalize_desc {
  const float* partial; const int32_t* seg_first; const float* seg_l2; float* sq; float* wsq; float* l2_out; int32_t nseg;
  const float* x0; float* out0; const float* x1; float* out1; int32_t n; float scale;
  const float* extra_part; float* extra; int32_t n_extra; int32_t extra_seg;
  const int32_t* ids_src; int32_t* ids_dst; int32_t n_ids;
  const float* x2; float* out2; int32_t n2; float scale2;
  int64_t* adam_t; uint32_t* drop_step; const float* lr; float* lr_t; float beta1, beta2; const uint32_t* guard;
  uint32_t* arrive;
} tnt_finalize_desc;
int32_t tnt_span_sqnorm_lr_f32(const float* theta, const float* grad, const int32_t* span_seg, const int64_t* span_off,
                               const int32_t* span_len, const float* seg_l2, float* partial, int32_t nspan,
                               const int64_t* adam_t, const float* lr, float* lr_t, float beta1, float beta2,
                               const float* sq_override, void* stream);
int32_t tnt_dense_gram_norm_spans_lr_f32(const float* dpre, const float* pre, const float* bias, const float* gx_part,
                                         int32_t nsplit, const float* w2_part, int32_t nw2, float l2, float* partial,
                                         int32_t nslot, int32_t Bk, int32_t E, const float* theta, const float* grad,
                                         const int32_t* span_seg, const int64_t* span_off, const int32_t* span_len,
                                         const float* seg_l2, float* span_partial, int32_t nspan, const int64_t* adam_t,
                                         const float* lr, float* lr_t, float beta1, float beta2, const float* sq_override,
                                         void* stream);
int32_t tnt_dense_dw_adam_fin_f32(const float* x, const float* dpre, float* theta, float* m, float* v, float l2,
                                  const float* partial, int32_t k0, int32_t k1, const float* sq_override,
                                  const float* lr_t_dev, float beta1, float beta2, float eps, float clipnorm,
                                  const uint32_t* guard, int32_t N, int32_t E, int32_t Bk, int32_t ldx, void* stream);
int32_t tnt_adam_fin_f32(float* theta, float* m, float* v, const float* grad, const int32_t* span_seg,
                         const int64_t* span_off, const int32_t* span_len, const float* sq_override, int32_t nspan, float eps,
                         float clipnorm, const tnt_finalize_desc* fin, const float* met, int32_t nmet, float* ring,
                         int32_t ring_rows, uint32_t* ring_t, void* stream);

/* tuning entry point: tnt_gemm_f32 with the workgroup tile forced to bm x bn (each 64 or 128; anything else =
 * the library's own choice).  Used by tools/gemm_bench.py / gemm_scan.py to calibrate the tile heuristic. */
int32_t tnt_gemm_f32_tile(const float* A, const float* B, float* C, const float* bias, float* pre,
                          int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc,
                          int32_t transA, int32_t transB, int32_t act, float slope,
                          int32_t accumulate, int32_t splitk, float* work, int32_t bm, int32_t bn,
                          void* stream);

/* ---- dropout (keras Dropout, inverted; lc_NIC.py:51-55,94; LSTM dropout= :122) ---
 * y[r][c] = keep ? x[r][c]/(1-rate) : 0 for r<rows, c<cols (ld = row stride of x,y).
 * logical element index e = lrow*lwidth + lcol0 + c, with
 *   lrow = r                      if tmajor_B == 0
 *   lrow = (r % B)*T + r / B      if tmajor_B == B > 0 (buffer time-major, logical (B,T,..)),
 * where T = rows / B.  In-place (y == x) allowed.  Backward = same call on dy.
 * rows_per_site > 0 (requires tmajor_B == 0): the buffer is a stack of independent
 * (rows_per_site x cols) tensors, block k using site + k and local rows -- one launch for
 * the T per-timestep masks of lc_NIC.py:255-256.
 * The stream step is step + *step_dev (step_dev nullable): a device-resident counter
 * lets a captured hipGraph replay with a fresh mask each time (see tnt_step_tick). */
/* Keep-masks only (no data pass): out[nsites][n/4] bytes, bit j of byte g of block k = element 4g+j of site site0+k is
 * kept -- the same Philox stream (seed, site, step + *step_dev, element) as tnt_dropout_f32, n % 4 == 0.  One launch
 * produces the attention-dropout masks of all T timesteps (attention.py:36, sites S_ATTN + t) off the serial chain;
 * tnt_attention_step_{fwd,bwd}_f32 take them through `keep4`. */
int32_t tnt_dropout_mask4_u8(uint8_t* out, int64_t n, int32_t nsites, float rate, uint64_t seed, uint32_t site0,
                             uint32_t step, const uint32_t* step_dev, void* stream);
int32_t tnt_dropout_f32(const float* x, float* y, int32_t rows, int32_t cols, int32_t ld,
                        int32_t tmajor_B, int32_t lwidth, int32_t lcol0, int32_t rows_per_site,
                        float rate, uint64_t seed, uint32_t site, uint32_t step,
                        const uint32_t* step_dev, void* stream);
/* tnt_dropout_f32 (vector form only: cols, ld, lwidth, lcol0 % 4 == 0, 16-byte aligned x / y) with the partials of
 * tnt_attention_metric_f32(out = NULL) as extra workgroups of the same launch: partial[tnt_attention_metric_parts(T, R)]
 * from alpha [T][B][R] (lc_NIC.py:365-367). */
int32_t tnt_dropout_metric_f32(const float* x, float* y, int32_t rows, int32_t cols, int32_t ld, int32_t tmajor_B,
                               int32_t lwidth, int32_t lcol0, int32_t rows_per_site, float rate, uint64_t seed,
                               uint32_t site, uint32_t step, const uint32_t* step_dev, const float* alpha,
                               float* partial, int32_t T, int32_t B, int32_t R, void* stream);
/* y = mask_b(mask_a(x)): two Dropout masks over the same matrix in one launch, each with its own logical layout, rate and
 * site (arguments as in tnt_dropout_f32, suffix _a / _b), same seed and step -- bit-identical to the two tnt_dropout_f32
 * calls it replaces (the LSTM input mask and the Embedding Dropout of the text rows in the backward pass,
 * lc_NIC.py:233-234,255).  Both rates in (0, 1); cols, ld, lwidth_*, lcol0_* % 4 == 0, x / y 16-byte aligned. */
int32_t tnt_dropout2_f32(const float* x, float* y, int32_t rows, int32_t cols, int32_t ld, int32_t tmajor_B_a,
                         int32_t lwidth_a, int32_t lcol0_a, int32_t rows_per_site_a, float rate_a, uint32_t site_a,
                         int32_t tmajor_B_b, int32_t lwidth_b, int32_t lcol0_b, int32_t rows_per_site_b, float rate_b,
                         uint32_t site_b, uint64_t seed, uint32_t step, const uint32_t* step_dev, void* stream);

/* ---- activation backward: dx = dy * act'(pre)  ---------------------------------- */
int32_t tnt_act_bwd_f32(const float* pre, const float* dy, float* dx, int64_t n, int32_t act,
                        float slope, void* stream);
/* Backward of y = Dropout(act(pre)) of a layer applied to rows <= 2048 rows, up to its bias gradient, in one launch:
 * dx = dropout'(dy) * act'(pre) (dx may alias dy), dbias[c] = sum_r dx[r][c].  The dropout stream / logical layout
 * arguments are those of tnt_dropout_f32 for the forward's mask (rate 0: no dropout).  cols, ld, lwidth, lcol0 % 4 == 0,
 * 16-byte aligned.  Optional second job in the same launch: out1 = column sums of x1 [rows1 <= 2048][C1] (x1 NULL: none).
 * (TimeDistributed(Dense) + Dropout of the caption head, lc_NIC.py:271-275, and its bias gradients.) */
int32_t tnt_bias_act_drop_bwd_f32(const float* dy, const float* pre, float* dx, float* dbias, int32_t rows, int32_t cols,
                                  int32_t ld, int32_t act, float slope, int32_t tmajor_B, int32_t lwidth, int32_t lcol0,
                                  float rate, uint64_t seed, uint32_t site, const uint32_t* step_dev, const float* x1,
                                  float* out1, int32_t rows1, int32_t C1, int32_t ld1, void* stream);

/* ---- BatchNormalization, axis=-1, non-fused keras semantics ---------------------
 * (layers.py:40,50; NIC.py:62,128; fullyConnected.py:18,24)
 * x is [rows][C] (ld = C).  training=1: biased batch statistics over rows;
 * moving <- moving*momentum + batch*(1-momentum) (updated in place).
 * Outputs: y, xhat (saved for backward), inv_std[C].  work: C*(2*nchunk+1) floats
 * (nchunk = tnt_bn_nchunk(rows)), same size for the backward and for LayerNorm bwd. */
int32_t tnt_bn_nchunk(int32_t rows);
int32_t tnt_batchnorm_fwd_f32(const float* x, const float* gamma, const float* beta,
                              float* mov_mean, float* mov_var, float* y, float* xhat,
                              float* inv_std, int32_t rows, int32_t C, int32_t ldy,
                              int32_t training, float eps, float momentum, float* work,
                              void* stream);
/* ... with the keras Dropout that follows the normalisation (layers.py:50-51) applied to y in the same pass: element
 * r*C + c of stream (seed, site, *step_dev); xhat stays the un-dropped normalised value.  rate 0 = tnt_batchnorm_fwd_f32. */
int32_t tnt_batchnorm_fwd_drop_f32(const float* x, const float* gamma, const float* beta,
                                   float* mov_mean, float* mov_var, float* y, float* xhat,
                                   float* inv_std, int32_t rows, int32_t C, int32_t ldy,
                                   int32_t training, float eps, float momentum, float* work,
                                   float rate, uint64_t seed, uint32_t site, const uint32_t* step_dev,
                                   void* stream);
/* dx (nullable), dgamma[C], dbeta[C] from dy (row stride lddy), xhat, inv_std. */
int32_t tnt_batchnorm_bwd_f32(const float* dy, const float* xhat, const float* gamma,
                              const float* inv_std, float* dx, float* dgamma, float* dbeta,
                              int32_t rows, int32_t C, int32_t lddy, int32_t training,
                              float* work, void* stream);
/* ... with dx further multiplied by LeakyReLU'(act_pre [rows][C], slope): the backward of the activation in front of the
 * normalisation (layers.py:48-50) in the same pass.  act_pre NULL = tnt_batchnorm_bwd_f32. */
int32_t tnt_batchnorm_bwd_act_f32(const float* dy, const float* xhat, const float* gamma,
                              const float* inv_std, float* dx, float* dgamma, float* dbeta,
                              int32_t rows, int32_t C, int32_t lddy, int32_t training,
                              float* work, const float* act_pre, float slope, void* stream);

/* Synchronised BatchNorm (data parallel, opt-in; the reference trains on one device, so this is what makes G replicas x
 * local batch equal ONE process on the concatenated batch when the encoder uses BatchNorm): the pieces of the two
 * functions above around the caller's collectives.
 *   stats        part = this replica's row-chunk partials, tnt_bn_nchunk(rows) * 2 * C floats;
 *   apply_stats  part_all = nrep replicas' partials back to back (all-gathered, every replica `rows` rows): statistics over
 *                nrep * rows rows, moving statistics updated, y / xhat / inv_std written; mean_work: C floats of scratch;
 *   dx           input gradient from dgamma_sum / dbeta_sum taken over all n_total rows (all-reduced local sums of
 *                tnt_batchnorm_bwd_f32 called with dx = NULL). */
int32_t tnt_batchnorm_stats_f32(const float* x, int32_t rows, int32_t C, float* part, void* stream);
int32_t tnt_batchnorm_apply_stats_f32(const float* part_all, int32_t nrep, const float* x, const float* gamma,
                                      const float* beta, float* mov_mean, float* mov_var, float* y, float* xhat,
                                      float* inv_std, int32_t rows, int32_t C, int32_t ldy, float eps, float momentum,
                                      float* mean_work, void* stream);
int32_t tnt_batchnorm_dx_f32(const float* dy, int32_t lddy, const float* xhat, const float* gamma, const float* inv_std,
                             const float* dgamma_sum, const float* dbeta_sum, float* dx, int32_t rows, int32_t C,
                             int32_t n_total, void* stream);

/* Split mode of the two entry points above: the launch runs over PIECES of at most `piece` voxels of a region's index
 * list, so that one large region does not set the kernel time (regions span 8..400+ voxels; measured 39 -> ~12 us
 * at 360 regions of 18..240 voxels).  Tables (host-built from goff): vgoff[NV+1] CSR range of each piece, vreg[NV] its
 * region, vfirst[NV] 1 for a region's first piece, rfirst[R+1] first piece of each region.  Forward = piece partials
 * (partial: NV*64*D floats) + a combine launch; results equal the unsplit ones up to f32 summation order.
 * x_voxel_major = 1: x is stored [n_voxels][ldx] (ldx >= B, ldx % 4 == 0): a voxel's batch values are contiguous, the
 * gather reads whole 256-byte rows and every byte of the betas is fetched once per piece -- the layout to stage at
 * full-cortex width, where the batch-major gather re-fetches each 128-byte line once per region that shares it. */
int32_t tnt_locally_dense_fwd_split_f32(const float* x, int32_t ldx, const int32_t* idx, const int32_t* vgoff,
                                        const int32_t* vreg, const int32_t* rfirst, int32_t NV,
                                        const float* W, const float* bias, float* pre, float* y,
                                        float* partial, int32_t B, int32_t R, int32_t D, float slope,
                                        int32_t x_voxel_major, void* stream);
int32_t tnt_locally_dense_bwd_split_f32(const float* x, int32_t ldx, const int32_t* idx, const int32_t* vgoff,
                                        const int32_t* vreg, const int32_t* vfirst, int32_t NV,
                                        const float* dpre, float* dW, float* db, int32_t B, int32_t R,
                                        int32_t D, int32_t x_voxel_major, void* stream);

/* ---- forward of the dense voxel encoder at small batch, as K-split partials ---------------------------
 * part[s][B][E] = x[B][K slice s] @ w[K slice s][E], s < nsplit; x [B][ldx] (K % 4 == 0, ldx % 4 == 0), w [K][ldw]
 * keras (in, out) kernel, E % 32 == 0, nsplit <= 64, all 16-byte aligned.  (NIC.py:64-69,125; ThinkAndTell
 * model.py:22-33.)  The sum over s (+ bias + LeakyReLU) is taken by tnt_enc_tail_fwd_sk_f32. */
int32_t tnt_dense_fwd_stream_f32(const float* x, const float* w, float* part, int32_t B, int32_t E, int32_t K,
                                 int32_t ldx, int32_t ldw, int32_t nsplit, void* stream);

/* Training variant with the two by-products the optimizer step needs to clip this kernel's gradient without forming
 * it (one row block: B <= 64, K % 16 == 0, E >= 512): gx_part [nsplit][64][64] K-split partials of x x^T (rows/columns
 * >= B undefined), w2_part [nsplit * E / 32] parts of sum w^2. */
int32_t tnt_dense_fwd_stream_gram_f32(const float* x, const float* w, float* part, float* gx_part, float* w2_part,
                                      int32_t B, int32_t E, int32_t K, int32_t ldx, int32_t ldw, int32_t nsplit,
                                      void* stream);
/* ... and the finish, after the backward pass produced dpre [Bk][E] (gradient w.r.t. the pre-activation pre [Bk][E]
 * = x w + bias): for g = x^T dpre (NIC.py:248-249) the clipnorm input sum (g + 2 l2 w)^2 = ||g||^2 + 4 l2 <g, w> +
 * 4 l2^2 ||w||^2 with ||g||^2 = sum (x x^T) o (dpre dpre^T) and <g, w> = sum dpre o (pre - bias), written as span
 * partials (partial[2k] parts of the norm, partial[2k+1] parts of sum w^2, every k < nslot written; nslot >= 4 Bk + nw2)
 * in the layout tnt_span_sqnorm_f32 / tnt_step_finalize_f32 use for this variable. */
int32_t tnt_dense_gram_norm_f32(const float* dpre, const float* pre, const float* bias, const float* gx_part,
                                int32_t nsplit, const float* w2_part, int32_t nw2, float l2, float* partial,
                                int32_t nslot, int32_t Bk, int32_t E, void* stream);
/* ... with tnt_span_sqnorm_f32's job for the other variables (same arguments: the span tables and partial buffer of the
 * nspan spans that are NOT this variable's) riding in the same launch */
int32_t tnt_dense_gram_norm_spans_f32(const float* dpre, const float* pre, const float* bias, const float* gx_part,
                                      int32_t nsplit, const float* w2_part, int32_t nw2, float l2, float* partial,
                                      int32_t nslot, int32_t Bk, int32_t E, const float* theta, const float* grad,
                                      const int32_t* span_seg, const int64_t* span_off, const int32_t* span_len,
                                      const float* seg_l2, float* span_partial, int32_t nspan, void* stream);

/* ---- weight gradient of the dense voxel encoder at small batch:  dw[N][E] = x^T @ dpre --------------
 * x [Bk][ldx] (the betas, Bk <= 64 rows), dpre [Bk][E] (E % 32 == 0, E <= 512, 16-byte aligned), dw [N][E].
 * (NIC.py:64-69,248-249; ThinkAndTell model.py:22-33.)  Same result as tnt_gemm_f32(transA=1) up to f32
 * summation order; a persistent, write-bound kernel instead of 2504 two-chunk tiles. */
int32_t tnt_dense_dw_skinny_f32(const float* x, const float* dpre, float* dw, int32_t N, int32_t E,
                                int32_t Bk, int32_t ldx, void* stream);

/* ---- the same weight gradient, consumed in place by the single-process optimizer step (clipnorm + Adam of
 * optimizer.apply_gradients, NIC.py:250-251, main.py:97) instead of being written out: E % 512 == 0, Bk <= 64.
 * sqnorm: partial[2k] = part of sum (g + 2 l2 theta)^2, partial[2k+1] = part of sum theta^2, k < nslot -- the span-partial
 *         layout of tnt_span_sqnorm_f32 for this variable's nslot >= 256 * E / 512 (or N/16 * E / 512) spans; every slot
 *         is written.
 * adam:   theta, m, v of the variable ([N][E]) updated exactly as tnt_adam_f32 does with grad = x^T dpre, clip factor
 *         from sq[0] (sq_override[0] >= 0 wins), lr_t from *lr_t_dev; skipped when *guard != 0. */
int32_t tnt_dense_dw_sqnorm_f32(const float* x, const float* dpre, const float* theta, float l2, float* partial,
                                int32_t nslot, int32_t N, int32_t E, int32_t Bk, int32_t ldx, void* stream);
int32_t tnt_dense_dw_adam_f32(const float* x, const float* dpre, float* theta, float* m, float* v, float l2,
                              const float* sq, const float* sq_override, const float* lr_t_dev, float beta1,
                              float beta2, float eps, float clipnorm, const uint32_t* guard, int32_t N, int32_t E,
                              int32_t Bk, int32_t ldx, void* stream);

/* ---- fused tail of the dense voxel encoder at small batch (rows <= 256, C % 4 == 0, 16-byte aligned
 * operands), NIC.py:126-128,138 ------------------------------------------------------------------
 * fwd: out = lstm_in_dropout( BatchNorm( feature_dropout(y) ) ), training statistics over the rows (moving
 *      statistics updated), or the moving statistics when training=0 (dropouts off).  Saves xhat, inv_std.
 *      Dropout element index = r*C + c in streams (seed, site_feat | site_lstm, *step_dev).
 * bwd: dout (gradient w.r.t. out, row stride ldo) -> lstm-in dropout' -> BatchNorm' (dgamma, dbeta) ->
 *      feature dropout' -> LeakyReLU'(pre, slope) -> dpre [rows][C]; dbias[c] = sum_r dpre.
 * One launch each, replacing 5 / 7 dependent launches of the generic entry points above (same arithmetic). */
int32_t tnt_enc_tail_fwd_f32(const float* y, const float* gamma, const float* beta, float* mov_mean,
                             float* mov_var, float* out, float* xhat, float* inv_std, int32_t rows,
                             int32_t C, int32_t ldo, int32_t training, float eps, float momentum,
                             float r_feat, float r_lstm, uint64_t seed, uint32_t site_feat,
                             uint32_t site_lstm, const uint32_t* step_dev, void* stream);
/* fwd from the K-split partials of tnt_dense_fwd_stream_f32: y = LeakyReLU(sum_s part[s] + bias, slope) in split
 * order (the Dense layer's output, NIC.py:125; pre-activation stored to pre [rows][C] for the backward), then as above. */
int32_t tnt_enc_tail_fwd_sk_f32(const float* part, int32_t nsplit, const float* bias, float* pre, float slope,
                                const float* gamma, const float* beta, float* mov_mean, float* mov_var, float* out,
                                float* xhat, float* inv_std, int32_t rows, int32_t C, int32_t ldo, int32_t training,
                                float eps, float momentum, float r_feat, float r_lstm, uint64_t seed,
                                uint32_t site_feat, uint32_t site_lstm, const uint32_t* step_dev, void* stream);
/* ... with the Embedding gather + input dropout of the text rows (tnt_embedding_fwd_drop_f32 with out = NULL: table
 * [emb_V][C], ids [emb_B][emb_T], emb_out rows t*B + b with row stride ldo, stream (seed, emb_site, *step_dev), rate 0 =
 * plain gather) riding in the same launch: both feed the same LSTM input buffer (NIC.py:131,138-140) */
int32_t tnt_enc_tail_fwd_sk_emb_f32(const float* part, int32_t nsplit, const float* bias, float* pre, float slope,
                                    const float* gamma, const float* beta, float* mov_mean, float* mov_var, float* out,
                                    float* xhat, float* inv_std, int32_t rows, int32_t C, int32_t ldo, int32_t training,
                                    float eps, float momentum, float r_feat, float r_lstm, uint64_t seed,
                                    uint32_t site_feat, uint32_t site_lstm, const uint32_t* step_dev,
                                    const float* emb_table, const int32_t* emb_ids, float* emb_out, int32_t emb_B,
                                    int32_t emb_T, int32_t emb_V, float emb_rate, uint32_t emb_site, void* stream);
int32_t tnt_enc_tail_bwd_f32(const float* dout, const float* xhat, const float* gamma, const float* inv_std,
                             const float* pre, float* dpre, float* dgamma, float* dbeta, float* dbias,
                             int32_t rows, int32_t C, int32_t ldo, float r_feat, float r_lstm, float slope,
                             uint64_t seed, uint32_t site_feat, uint32_t site_lstm,
                             const uint32_t* step_dev, void* stream);
/* bwd with an independent in-place dropout' job in the same launch: drop_x [drop_rows][drop_cols] (row stride drop_ld) gets
 * the keep mask of stream (seed, drop_site, *step_dev) in the logical layout arguments of tnt_dropout_f32 -- the text
 * call's LSTM-input dropout over the other rows of the same gradient buffer (NIC.py:131,140).  All % 4 == 0. */
int32_t tnt_enc_tail_bwd_drop_f32(const float* dout, const float* xhat, const float* gamma, const float* inv_std,
                                  const float* pre, float* dpre, float* dgamma, float* dbeta, float* dbias, int32_t rows,
                                  int32_t C, int32_t ldo, float r_feat, float r_lstm, float slope, uint64_t seed,
                                  uint32_t site_feat, uint32_t site_lstm, const uint32_t* step_dev, float* drop_x,
                                  int32_t drop_rows, int32_t drop_cols, int32_t drop_ld, int32_t drop_tmajor_B,
                                  int32_t drop_lwidth, int32_t drop_lcol0, float drop_rate, uint32_t drop_site,
                                  void* stream);

/* ---- LayerNormalization(axis=-1) (layers.py:41 alternative; BASELINE north_star) - */
int32_t tnt_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                              float* xhat, float* inv_std, int32_t rows, int32_t C,
                              int32_t ldy, float eps, void* stream);
int32_t tnt_layernorm_bwd_f32(const float* dy, const float* xhat, const float* gamma,
                              const float* inv_std, float* dx, float* dgamma, float* dbeta,
                              int32_t rows, int32_t C, int32_t lddy, float* work, void* stream);

/* ---- column sums: out[c] = sum_r x[r][c]  (bias gradients of every Dense) -------
 * work: C*tnt_bn_nchunk(rows) floats. */
int32_t tnt_colsum_f32(const float* x, float* out, int32_t rows, int32_t C, int32_t ld,
                       float* work, void* stream);
/* two / up to four independent matrices of <= 2048 rows each in one launch (same results as separate tnt_colsum_f32
 * calls; in tnt_colsum4_f32 a job with x == NULL is skipped) */
int32_t tnt_colsum4_f32(const float* x0, float* out0, int32_t rows0, int32_t C0, int32_t ld0, const float* x1, float* out1,
                        int32_t rows1, int32_t C1, int32_t ld1, const float* x2, float* out2, int32_t rows2, int32_t C2,
                        int32_t ld2, const float* x3, float* out3, int32_t rows3, int32_t C3, int32_t ld3, void* stream);
int32_t tnt_colsum2_f32(const float* x0, float* out0, int32_t rows0, int32_t C0, int32_t ld0, const float* x1,
                        float* out1, int32_t rows1, int32_t C1, int32_t ld1, void* stream);

/* ---- Embedding (lc_NIC.py:105-112,233; NIC.py:75-79,131) ------------------------
 * fwd: out[(t*B+b)][:] = table[ids[b*T+t]][:]   (ids is the keras (B,T) int32 array)
 * bwd: dtable[v][:] = sum over (b,t) with ids[b*T+t]==v of drows[(t*B+b)][:] (every row
 *      of dtable is written: unreferenced rows are zeroed by the call itself);
 *      sq_norm[0] = sum of squares of the un-merged rows (IndexedSlices clipnorm
 *      quirk), overwritten; rowsq_work: B*T floats.
 *      Deterministic (no atomics): one wave per vocabulary row sums its matches in
 *      (b,t) order and writes every row of dtable (no pre-zeroing needed). */
int32_t tnt_embedding_fwd_f32(const float* table, const int32_t* ids, float* out, int32_t B,
                              int32_t T, int32_t E, int32_t ldo, int32_t V, void* stream);
/* fwd + keras Dropout over the logical (B,T,E) tensor in the same pass (NIC.py:131,140): out_drop gets the
 * masked rows (stream (seed, site, step + *step_dev), element (b*T+t)*E + j); out (nullable) the plain ones.
 * E % 4 == 0, ldo % 4 == 0, 16-byte aligned. */
int32_t tnt_embedding_fwd_drop_f32(const float* table, const int32_t* ids, float* out, float* out_drop,
                                   int32_t B, int32_t T, int32_t E, int32_t ldo, int32_t V, float rate,
                                   uint64_t seed, uint32_t site, uint32_t step, const uint32_t* step_dev,
                                   void* stream);
/* ... followed, when rate2 > 0, by a second mask in the same pass: the LSTM layer's per-call input dropout, whose call of
 * timestep t draws one mask over its (B, lwidth2) input from site2 + t; the text rows are its columns lcol0_2 .. lcol0_2 + E
 * (lc_NIC.py:255).  Bit-identical to tnt_embedding_fwd_drop_f32 + tnt_dropout_f32(rows_per_site = B).
 * lwidth2, lcol0_2 % 4 == 0. */
int32_t tnt_embedding_fwd_drop2_f32(const float* table, const int32_t* ids, float* out, float* out_drop,
                                    int32_t B, int32_t T, int32_t E, int32_t ldo, int32_t V, float rate,
                                    uint64_t seed, uint32_t site, uint32_t step, const uint32_t* step_dev,
                                    float rate2, uint32_t site2, int32_t lwidth2, int32_t lcol0_2, void* stream);
/* Single-process form of tnt_embedding_bwd_f32 without the table-wide zero fill and the row-norm launches: the first
 * occurrence of an id sums its rows into dtable[id] (deterministic, as above) and leaves its share of the IndexedSlices
 * squared norm in sq_part[k * ny + y] (ny = ceil(E / 256); tnt_embedding_bwd_parts(B, T, E) floats; their sum is the
 * norm of the un-deduplicated rows); rows that only the PREVIOUS step touched (prev_ids [B*T], -1 = none) are zeroed.
 * Contract: dtable starts zeroed and is written by nothing else between calls (no gradient all-reduce), prev_ids holds
 * the ids of the previous call (tnt_step_finalize_f32 copies them).  E % 4 == 0, 16-byte aligned rows.
 * drop_rate > 0: drows is the gradient w.r.t. the dropped-out Embedding output (the LSTM layer's input dropout of the
 * text call, NIC.py:131,140); the keep mask tnt_embedding_fwd_drop_f32 applied (element (b*T + t) * E + e of stream
 * (drop_seed, drop_site, *drop_step_dev)) is applied to the rows as they are read.
 * zero_id >= 0 (-1: none): the caller guarantees that every row of drows whose id is zero_id is zero -- Embedding(mask_zero=True)
 * in front of an LSTM that honours the mask (NIC.py:77,131,140: a masked step passes no gradient to its input); those rows
 * (~40 % of a padded caption batch, all of ONE id) are not read, dtable[zero_id] is written as zeros. */
int32_t tnt_embedding_bwd_parts(int32_t B, int32_t T, int32_t E);
int32_t tnt_embedding_bwd_sparse_f32(const float* drows, const int32_t* ids, const int32_t* prev_ids, float* dtable,
                                     float* sq_part, int32_t B, int32_t T, int32_t E, int32_t ldd, int32_t V,
                                     float drop_rate, uint64_t drop_seed, uint32_t drop_site,
                                     const uint32_t* drop_step_dev, int32_t zero_id, void* stream);
int32_t tnt_embedding_bwd_f32(const float* drows, const int32_t* ids, float* dtable,
                              float* sq_norm, float* rowsq_work, int32_t B, int32_t T,
                              int32_t E, int32_t ldd, int32_t V, void* stream);

/* ---- LSTM cell step, keras LSTM v2 semantics (lc_NIC.py:118-124,255; NIC.py:82-88,
 * 138-140), gate-interleaved layouts.
 * fwd:  z = xz[B][U][4] + h_prev[B][U] @ Ur[U][U][4] (+ ctx[B][D] @ Wc[D][U][4])
 *       i,f,o = sigmoid, g = tanh; c = f*c_prev + i*g; h = o*tanh(c)
 *       mask (nullable, int32 ids[B*T], column `mask_t`): rows with id==0 keep
 *       (h_prev,c_prev) and repeat out_prev (zeros if out_prev is null).
 *       Saves gates[B][U][4] (post-activation).  out (nullable) receives the
 *       sequence output of this step.  xz_bias (nullable, [U][4]) is added to xz inside the kernel, so that
 *       the input projection of all timesteps can be an epilogue-free GEMM. */
int32_t tnt_lstm_step_fwd_f32(const float* xz, const float* h_prev, const float* c_prev,
                              const float* Ur, const float* ctx, const float* Wc, int32_t D,
                              const int32_t* mask_ids, int32_t mask_T, int32_t mask_t,
                              const float* out_prev, float* h, float* c, float* out,
                              float* gates, int32_t B, int32_t U, const float* xz_bias, void* stream);
/* ---- LayerNormLSTMCell (tensorflow_addons 0.15 rnn.LayerNormLSTMCell; the use_layer_norm branch of lc_NIC.py:126-136):
 *   z = LN_kernel(x W) + LN_recurrent(h U) + b;  c' = LN_state(sig(f) c + sig(i) tanh(c~));  h' = sig(o) tanh(c')
 * The two 4U-wide LayerNorms are tnt_layernorm_fwd/bwd_f32 launches (dgamma = dbeta = NULL there: input gradient only)
 * around the matmuls; these two entry points are the cell: gate math + the state LayerNorm over U, and the reverse.
 * Gate-interleaved tensors [B][U][4] (zk, zr, gates, dz; bias [U][4]); chat = the normalised pre-affine state, istd [B].
 * bwd: dh = dh_a + dh_b + dh_c (each nullable), dcn_in (nullable) = gradient wrt the normalised state carried to the
 * next step; writes dz, dc_prev (gradient wrt the previous step's normalised state; may alias dcn_in) and dcnt = the
 * total gradient at the normalised state (its products with chat / its column sums over all steps are the state
 * norm's gamma / beta gradients).  U <= 4096. */
int32_t tnt_ln_lstm_cell_fwd_f32(const float* zk, const float* zr, const float* bias, const float* c_prev,
                                 const float* gamma_s, const float* beta_s, float* gates, float* chat,
                                 float* istd, float* c, float* h, int32_t B, int32_t U, float eps, void* stream);
int32_t tnt_ln_lstm_cell_bwd_f32(const float* dh_a, const float* dh_b, const float* dh_c, const float* dcn_in,
                                 const float* gates, const float* c_prev, const float* c, const float* chat,
                                 const float* istd, const float* gamma_s, float* dz, float* dc_prev,
                                 float* dcnt, int32_t B, int32_t U, void* stream);
/* Persistent form of the masked sequence forward of NIC.py:138-140: ONE launch runs the S dependent steps
 * (step s reads hs[s], cs[s], xz[s] and writes hs[s+1], cs[s+1], gates[s]; steps s >= mask_s0 write the sequence
 * output out[s - mask_s0] (out nullable) and, if mask_ids[B][mask_T] is given, are masked by its column s - mask_s0;
 * pass mask_s0 = S for neither).  hs/cs: [S+1][B][U] with slab 0 = the initial state; xz: [S][B][U][4]; gates: [S][B][U][4];
 * sync: 1025 uint32 of scratch owned by the caller, zeroed before first use and after an error, never in between (the
 * kernel re-arms its own counters when it ends, so a captured launch replays without any reset node; protocol in
 * csrc/tnt_seq_sync.h).  Word 1024 is the sticky error word: 1 = a barrier timed out, 2 = a launch did not place 32
 * workgroups on each XCD; results are invalid from then on.  guard_out (nullable): one float the kernel sets to the error
 * code when it sees the error word set (never cleared by the device), so the caller can carry the check along with
 * the metrics it reads anyway; tnt_step_tick / tnt_adam_f32 / tnt_sgd_f32 take the error word as `guard` and leave the
 * model state untouched when it is set.  S <= 64.  Same arithmetic as S calls of tnt_lstm_step_fwd_f32.  A step pays one XCD-local barrier instead of a
 * dependent kernel launch and the recurrent weights stay in VGPRs; needs U == 512, B <= 128 and a 256-CU device on
 * which a 256-workgroup launch places 32 workgroups on each of the 8 XCDs: tnt_lstm_seq_supported() tests exactly that
 * (one synchronising probe launch per process, so call it outside any graph capture) and returns 1 or 0. */
int32_t tnt_lstm_seq_supported(int32_t B, int32_t U);
int32_t tnt_lstm_seq_fwd_f32(const float* xz, float* hs, float* cs, const float* Ur, const float* xz_bias,
                             const int32_t* mask_ids, int32_t mask_T, int32_t mask_s0, float* out,
                             float* gates, int32_t S, int32_t B, int32_t U, uint32_t* sync, float* guard_out, void* stream);
/* Persistent form of the BPTT chain of the same sequence (the S calls of tnt_lstm_step_bwd_f32 that nic.NIC makes, in
 * ONE launch): step s = S-1 .. 0 reads gates[s], cs[s+1], cs[s] and writes dz[s] ([S][B][U][4]); steps s >= mask_s0
 * add dout_seq[s - mask_s0] (gradient of the sequence output, nullable) and are masked by column s - mask_s0 of
 * mask_ids (nullable) exactly like the per-step kernel (masked row: dz = 0, the carried da / dc / dout pass through);
 * the carried output gradient is dropped below mask_s0 (NIC.py:138: the feature step's output is not part of the
 * sequence).  Weights stay in registers, the recurrent product is "pushed" as partial tiles through `work`
 * (tnt_lstm_seq_bwd_work_floats(B, U) floats, 16-byte aligned) inside one XCD per 16-row block; sync / guard_out and the
 * device requirements as tnt_lstm_seq_fwd_f32 (same sync buffer, launches on one stream).  Deterministic. */
int32_t tnt_lstm_seq_bwd_work_floats(int32_t B, int32_t U);
int32_t tnt_lstm_seq_bwd_f32(const float* Ur, const float* dout_seq, const int32_t* mask_ids, int32_t mask_T,
                             int32_t mask_s0, const float* gates, const float* cs, float* dz, float* work,
                             int64_t work_floats, int32_t S, int32_t B, int32_t U, uint32_t* sync,
                             float* guard_out, void* stream);
/* bwd of one step, fused with the recurrent matmul of the step after it:
 *   da = da_pass_in + dh_ext + (dz_next ? dz_next[B][U][4] @ Ur^T : 0)
 *   dout = dout_in + dout_t ; masked rows pass (da, dc, dout) through, dz = 0
 *   unmasked rows: dh = da + dout, standard LSTM cell backward -> dz[B][U][4],
 *   dc_out = dc*f, da_pass_out = 0, dout_out = 0.
 * Any of dz_next, da_pass_in, dh_ext, dc_in, dout_in, dout_t, mask_ids may be null.
 * dctx_part (nullable; attention model): [U/16][B][D] partial context gradients of this step,
 *   dctx_part[ub][b][d] = sum over the 16 units of block ub and the 4 gates of dz[b][u][g] * Wc[d][u][g]
 *   (Wc = the context rows of the LSTM kernel, [D][U][4], D <= 64); tnt_attention_step_bwd_f32 sums the parts. */
int32_t tnt_lstm_step_bwd_f32(const float* dz_next, const float* Ur, const float* da_pass_in,
                              const float* dh_ext, const float* dc_in, const float* dout_in,
                              const float* dout_t, const int32_t* mask_ids, int32_t mask_T,
                              int32_t mask_t, const float* gates, const float* c,
                              const float* c_prev, float* dz, float* da_pass_out,
                              float* dc_out, float* dout_out, int32_t B, int32_t U,
                              const float* Wc, int32_t D, float* dctx_part, void* stream);

/* ---- GRU cell step, keras GRU v2 semantics (reset_after=True): the decoder of ThinkAndTell/att_model.py
 * (att_model.py:84-93,118).  Gate-interleaved layouts [..][U][4] with slots (z, r, h, 0 pad).
 * fwd:  rec = h_prev[B][U] @ Uk[U][U][4] + br[U][4];  z = sigmoid(xz_z + rec_z), r = sigmoid(xz_r + rec_r),
 *       hh = tanh(xz_h + r*rec_h);  h = z*h_prev + (1-z)*hh.   xz[B][U][4] = x @ W + b_i (one GEMM for all steps).
 *       Saves gates[B][U][4] = (z, r, hh, rec_h).
 * bwd:  dh = dh_ext + dh_pass_in + (drec_next ? drec_next[B][U][4] @ Uk^T : 0);
 *       dxz = (da_z, da_r, da_h, 0)  [input side: kernel / input-bias gradients, dX],
 *       drec = (da_z, da_r, da_h*r, 0)  [recurrent side: recurrent-kernel / recurrent-bias gradients, and the
 *       matmul the call of the step before applies];  dh_pass_out = dh*z.  Nullable: drec_next, dh_pass_in, dh_ext,
 *       dh_pass_out. */
int32_t tnt_gru_step_fwd_f32(const float* xz, const float* h_prev, const float* Uk, const float* br, float* h,
                             float* gates, int32_t B, int32_t U, void* stream);
int32_t tnt_gru_step_bwd_f32(const float* drec_next, const float* Uk, const float* dh_pass_in,
                             const float* dh_ext, const float* gates, const float* h_prev, float* dxz,
                             float* drec, float* dh_pass_out, int32_t B, int32_t U, void* stream);

/* ---- softmax + CategoricalCrossentropy(from_logits=False) + accuracy ------------
 * (lc_NIC.py:153,370-376,461-486; NIC.py:93,234-240; main.py:107-110)
 * logits [rows][ld], V valid columns.  target ids int32[rows] (argmax of the one-hot).
 * probs (nullable, may alias logits): softmax.  loss_row/correct_row [rows]:
 * -log(clip(p_y/sum p, 1e-7, 1-1e-7)) and (argmax p == y).  dlogits (nullable, may
 * alias logits): (p - onehot)*gscale, zero rows where the clip is active.
 * from_logits=1: tf SparseCategoricalCrossentropy(from_logits=True) (ThinkAndTell/train.py:262-263):
 * loss = logsumexp - x_y, no clipping.  mask_zero=1: rows whose target id is 0 give zero loss
 * and zero gradient (CaptionGenerator.loss_function, ThinkAndTell/model.py:319-334). */
int32_t tnt_softmax_cce_f32(const float* logits, const int32_t* target, float* probs,
                            float* loss_row, float* correct_row, float* dlogits,
                            int32_t rows, int32_t V, int32_t ld, float gscale,
                            int32_t from_logits, int32_t mask_zero, void* stream);
/* target ids from a dense one-hot (B,T,V) float array: ids[t*B+b] = argmax_v. */
int32_t tnt_onehot_argmax_f32(const float* onehot, int32_t* ids_tmajor, int32_t B, int32_t T,
                              int32_t V, void* stream);
/* row argmax (first max wins), out int32[rows]. */
/* Beam-search expansion (beam search is only sketched in the reference: lc_NIC.py:640-692,
 * ThinkAndTell/evaluate.py:203-228).  Rows b*k .. b*k+k-1 of probs [B*k][ld] are the k beams of sample b; candidate
 * (beam j, token v) scores score_in[j] + log(max(p, 1e-30)); a finished beam (fin_in != 0: it has emitted end_id)
 * only continues with token 0 at its own score.  The k best become the new beams (ties: lower j*V + v):
 * score_out, parent (global row of the extended beam), token, fin_out -- all [B*k].  k <= 16. */
int32_t tnt_beam_topk_f32(const float* probs, const float* score_in, const int32_t* fin_in, int32_t B,
                          int32_t V, int32_t ld, int32_t k, int32_t end_id, float* score_out,
                          int32_t* parent, int32_t* token, int32_t* fin_out, void* stream);
int32_t tnt_argmax_rows_f32(const float* x, int32_t* out, int32_t rows, int32_t V, int32_t ld,
                            void* stream);
/* out[0] = scale * sum_i x[i]  (fixed-order, one workgroup). */
/* Categorical sampling per row (tf.random.categorical(logits / temperature, 1): ThinkAndTell/evaluate.py:223,278;
 * lc_NIC.sample_choice lc_NIC.py:571-575 samples from log(probs)).  x: logits (from_logits=1) or probabilities.
 * out[row] = first j with  w_0+..+w_j > u*sum(w),  w_j = exp((l_j - max l)/temperature),  u = the Philox
 * uniform of element `row` in stream (seed, site, step + *step_dev).  Deterministic; restated by the oracle. */
int32_t tnt_sample_rows_f32(const float* x, int32_t* out, int32_t rows, int32_t V, int32_t ld,
                            float temperature, int32_t from_logits, uint64_t seed, uint32_t site,
                            uint32_t step, const uint32_t* step_dev, void* stream);
/* out[0] = mean_i (c - x[i])^2 (MeanSquaredError against a constant target, lc_NIC.py:813-814) */
int32_t tnt_sqdiff_mean_f32(const float* x, float* out, int64_t n, float c, void* stream);
int32_t tnt_sum_f32(const float* x, float* out, int32_t n, float scale, void* stream);
/* two of them in one launch: out0[0] = scale*sum(x0[0..n)), out1[0] = scale*sum(x1[0..n))  (loss + accuracy) */
int32_t tnt_sum2_f32(const float* x0, float* out0, const float* x1, float* out1, int32_t n, float scale,
                     void* stream);
/* Input staging of one device-resident batch in one launch (the generator tuple of
 * data_generator_guse.py:156-171 -> the static buffers of the captured step): x (B,N) -> x_dst (B,ldx);
 * cap (B,T) int32 -> cap_dst; tgt (B,T) int32 ids (nullable) -> tgt_tmajor (T,B); a0, c0 (B,U) -> h0, c0_dst.
 * xT_dst (nullable): additionally the voxel-major copy xT[N][ldt] (ldt >= B) the region-wise encoder gathers from. */
int32_t tnt_stage_batch_f32(const float* x, float* x_dst, const int32_t* cap, int32_t* cap_dst,
                            const int32_t* tgt, int32_t* tgt_tmajor, const float* a0, float* h0,
                            const float* c0, float* c0_dst, int32_t B, int32_t T, int32_t N, int32_t ldx,
                            int32_t U, float* xT_dst, int32_t ldt, void* stream);
/* tnt_stage_batch_f32 with tnt_dropout_mask4_u8's job riding in the same launch: keep_out [keep_sites][keep_n / 4] bytes, the
 * keep masks of sites keep_site0 .. of stream (keep_seed, *keep_step_dev) -- the attention-dropout masks of the training
 * step this batch feeds (attention.py:36).  The Philox-bound mask blocks and the memory-bound copies share the chip; the
 * captured step then starts without its own mask launch. */
int32_t tnt_stage_batch_masks_f32(const float* x, float* x_dst, const int32_t* cap, int32_t* cap_dst,
                            const int32_t* tgt, int32_t* tgt_tmajor, const float* a0, float* h0,
                            const float* c0, float* c0_dst, int32_t B, int32_t T, int32_t N, int32_t ldx,
                            int32_t U, float* xT_dst, int32_t ldt, uint8_t* keep_out, int64_t keep_n,
                                  int32_t keep_sites, float keep_rate, uint64_t keep_seed, uint32_t keep_site0,
                                  const uint32_t* keep_step_dev, void* stream);
/* Same staging for betas that crossed PCIe as IEEE half (x_half: (B,N) uint16 bit patterns, 8-byte aligned when
 * N % 4 == 0): widened to float in the same pass.  Opt-in "fp16 on-wire" input of data.PinnedPrefetcher
 * (at full-cortex width, N = 327 684, the 84 MB float batch is what bounds the step). */
int32_t tnt_stage_batch_h16(const uint16_t* x_half, float* x_dst, const int32_t* cap, int32_t* cap_dst,
                            const int32_t* tgt, int32_t* tgt_tmajor, const float* a0, float* h0,
                            const float* c0, float* c0_dst, int32_t B, int32_t T, int32_t N, int32_t ldx,
                            int32_t U, float* xT_dst, int32_t ldt, void* stream);

/* ---- optimizer: per-variable clipnorm + Adam / SGD over a flat parameter arena --
 * (main.py:97,100-102; lc_NIC.py:389).  The arena is cut by the host into
 * spans of one variable ("segment") each: span_seg[k], span_off[k] (arena offset in
 * floats, multiple of 4), span_len[k]; seg_first[s..s+1] = spans of segment s;
 * seg_l2[s] = L2 lambda (the gradient gets + 2*lambda*theta, lc_NIC.py:47-50).
 * sqnorm: sq[s] = sum (g + 2 lambda theta)^2, wsq[s] = sum theta^2 (L2 metric);
 * partial: 2*nspan floats; l2_out (nullable) = sum_s lambda_s*wsq[s] (the 'L2' metric,
 * tf.add_n(self.losses), lc_NIC.py:379).  If sq_override[s] >= 0 it replaces sq[s] for clipping
 * (Embedding IndexedSlices norm).  clipnorm <= 0 disables clipping.
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is read from lr_t_dev when non-null. */
int32_t tnt_seg_sqnorm_f32(const float* theta, const float* grad, const int32_t* span_seg,
                           const int64_t* span_off, const int32_t* span_len,
                           const int32_t* seg_first, const float* seg_l2, float* partial,
                           float* sq, float* wsq, float* l2_out, int32_t nspan, int32_t nseg,
                           void* stream);
/* The scalar tail of a training step in ONE launch (each piece used to be its own dependent launch):
 * span partials (tnt_span_sqnorm_f32: the first kernel of tnt_seg_sqnorm_f32 alone) -> sq / wsq per variable;
 * l2_out = sum_s seg_l2[s]*wsq[s]; out0 = scale*sum x0[0..n), out1 = scale*sum x1[0..n) (loss / accuracy totals,
 * lc_NIC.py:370-376; x1 nullable); extra[0] = sum extra_part[0..n_extra) (the Embedding's IndexedSlices squared norm
 * from the per-block partials of the scatter); ids_dst[0..n_ids) = ids_src (this step's token ids become the
 * prev_ids of tnt_embedding_bwd_sparse_f32); out2 = scale2*sum x2[0..n2) (one more scaled total: the per-timestep partials
 * of tnt_attention_metric_f32 called with out = NULL); then the step state of tnt_step_tick advances (same arguments,
 * same guard rule).  Every piece is optional: nseg = 0, n = 0, n_extra = 0, n2 = 0, null state pointers.  Fixed summation
 * order. */
int32_t tnt_span_sqnorm_f32(const float* theta, const float* grad, const int32_t* span_seg,
                            const int64_t* span_off, const int32_t* span_len, const float* seg_l2,
                            float* partial, int32_t nspan, void* stream);
int32_t tnt_step_finalize_f32(const float* partial, const int32_t* seg_first, const float* seg_l2, float* sq,
                              float* wsq, float* l2_out, int32_t nseg, const float* x0, float* out0,
                              const float* x1, float* out1, int32_t n, float scale, const float* extra_part,
                              float* extra, int32_t n_extra, const int32_t* ids_src, int32_t* ids_dst,
                              int32_t n_ids, const float* x2, float* out2, int32_t n2, float scale2,
                              int64_t* adam_t, uint32_t* drop_step, const float* lr,
                              float* lr_t, float beta1, float beta2, const uint32_t* guard, void* stream);
/* out[0] = sum_s seg_l2[s]*wsq[s]; for callers that run tnt_seg_sqnorm_f32 on slices of the span
 * table (offset pointers, slice-local seg_first; the pipelined data-parallel update) and need the
 * total afterwards. */
int32_t tnt_l2_total_f32(const float* wsq, const float* seg_l2, int32_t nseg, float* out, void* stream);
int32_t tnt_adam_f32(float* theta, float* m, float* v, const float* grad,
                     const int32_t* span_seg, const int64_t* span_off, const int32_t* span_len,
                     const float* seg_l2, const float* sq, const float* sq_override,
                     int32_t nspan, float lr_t, const float* lr_t_dev, float beta1, float beta2,
                     float eps, float clipnorm, const uint32_t* guard, void* stream);
/* tnt_adam_f32 that also files the step's metrics vector in a ring (one wave of the launch, before the guard check): met
 * [nmet <= 62] is copied to row (*ring_t % ring_rows) of ring [ring_rows][nmet + 1], column nmet = the launch number *ring_t
 * (low 24 bits, as a float), and *ring_t advances -- the host reads that row when it wants the step's metrics instead of
 * copying `met` behind every step (a 5 us launch on a 0.55 ms step).  ring NULL = tnt_adam_f32. */
int32_t tnt_adam_ring_f32(float* theta, float* m, float* v, const float* grad,
                     const int32_t* span_seg, const int64_t* span_off, const int32_t* span_len,
                     const float* seg_l2, const float* sq, const float* sq_override,
                     int32_t nspan, float lr_t, const float* lr_t_dev, float beta1, float beta2,
                     float eps, float clipnorm, const uint32_t* guard, const float* met, int32_t nmet, float* ring,
                          int32_t ring_rows, uint32_t* ring_t, void* stream);
int32_t tnt_sgd_f32(float* theta, float* mom, const float* grad, const int32_t* span_seg,
                    const int64_t* span_off, const int32_t* span_len, const float* seg_l2,
                    const float* sq, const float* sq_override, int32_t nspan, float lr,
                    const float* lr_dev, float momentum, float clipnorm, const uint32_t* guard, void* stream);
/* ---- adaptive gradient clipping, unit-wise (AttemptFour/Model/agc.py:20-38; call site lc_NIC.py:388) ----
 * Applied in place to the flat gradient arena before the norms / clip-by-norm / Adam: per variable v (table entry:
 * arena offset var_off, row stride var_ld, L2 lambda var_lam) and per unit = output column of a keras (in, out) kernel
 * (vectors: one unit):  g <- g * max(||theta_u||, eps) * clip_factor / max(||g_u||, 1e-6)  where ||g_u|| is not below
 * that bound; g = arena gradient + 2 lambda theta (what tape.gradient returns), written back minus the L2 term.
 * item [nitem][6] = (var, c0, ncols <= 64, r0, r1, column block id), cb_first [ncb+1] = first item of each column
 * block (items of a block are consecutive); partial: nitem*128 floats.  Embedding (IndexedSlices, agc.py:25-30):
 * gsq_cols[E] = column sums of squares of the un-deduplicated row gradients (tnt_colsq_f32) replaces the dense
 * gradient's for variable gsq_var, whose column blocks are gsq_cb0 .. gsq_cb0+gsq_ncb-1; sq_out[0] receives the squared
 * norm of the clipped rows (for clip-by-norm's IndexedSlices norm), sq_part: gsq_ncb floats of scratch. */
int32_t tnt_agc_f32(const float* theta, float* grad, const int64_t* var_off, const int32_t* var_ld,
                    const float* var_lam, const int32_t* item, const int32_t* cb_first, int32_t nitem,
                    float* partial, const float* gsq_cols, int32_t gsq_var, int32_t gsq_cb0, int32_t gsq_ncb,
                    float* sq_part, float* sq_out, float clip_factor, float eps, void* stream);
/* out[c] = sum_r x[r][c]^2 */
int32_t tnt_colsq_f32(const float* x, float* out, int32_t rows, int32_t cols, int32_t ld, void* stream);
/* Sharpness-aware minimisation helper (CaptionGenerator.train_step_SAM, ThinkAndTell/model.py:166-233;
 * lc_NIC.train_step_sam, lc_NIC.py:713-838).  mode 0: e_w = (g + 2 lambda theta) * rho/(||g||+1e-12)
 * with ||g||^2 = sum_s sq[s] (from tnt_seg_sqnorm_f32; sq_override[s] >= 0 replaces sq[s]: tf.linalg.global_norm takes an
 * IndexedSlices gradient by its un-deduplicated values, nullable); theta += e_w; e_w stored.  mode 1: theta -= e_w, and
 * grad += 2 lambda e_w: the gradient the tape took at theta + e_w carries the L2 term of the PERTURBED weights, while the
 * optimizer kernels add 2 lambda theta at the restored ones. */
int32_t tnt_sam_f32(float* theta, float* grad, float* ew, const int32_t* span_seg,
                    const int64_t* span_off, const int32_t* span_len, const float* seg_l2,
                    const float* sq, const float* sq_override, int32_t nseg, int32_t nspan, float rho, int32_t mode, void* stream);
/* device-resident step state, advanced inside the (captured) step:
 * adam_t += 1; lr_t = lr[0]*sqrt(1-b2^t)/(1-b1^t); drop_step += 1.  Pointers nullable.
 * guard (nullable): a device error word (tnt_lstm_seq_fwd_f32); when it is non-zero nothing is advanced, and
 * tnt_adam_f32 / tnt_sgd_f32 given the same word skip their update: a step whose forward pass was invalid leaves
 * weights, moments, t and the dropout stream exactly as they were, so the caller can redo it. */
int32_t tnt_step_tick(int64_t* adam_t, uint32_t* drop_step, const float* lr, float* lr_t,
                      float beta1, float beta2, const uint32_t* guard, void* stream);

/* ---- region-wise encoder: layers.LocallyDense.call (layers.py:43-48) ------------
 * CSR groups: idx[goff[r] .. goff[r+1]) are the voxel columns of group r; W is the
 * concatenation of the per-group kernels, [goff[R]][D]; bias [R][D].
 * Any B (processed in row blocks of 64, in order), D % 16 == 0, D <= 64.
 * fwd: pre/y[b][r][:] = LeakyReLU(x[b][idx_r] @ W_r + b_r)    (y,pre: [B][R][D])
 * bwd: dW_r = x[:,idx_r]^T @ dpre[:,r,:]; db_r = sum_b dpre[b][r][:]. */
/* Input gradient of a stack of per-region Dense layers -- the deeper stages of deep_layers.LocallyDense
 * (AttemptFour/Model/deep_layers.py:53-59), whose forward and weight gradients are tnt_locally_dense_*_f32 on the
 * identity groups {r*D .. (r+1)*D-1}:  dx[b][r][k] = sum_n dpre[b][r][n] * W[r][k][n];  dpre [B][R][Dout],
 * W [R][Din][Dout], dx [B][R][Din]; Din, Dout <= 64. */
int32_t tnt_block_dense_dx_f32(const float* dpre, const float* W, float* dx, int32_t B, int32_t R,
                               int32_t Din, int32_t Dout, void* stream);
int32_t tnt_locally_dense_fwd_f32(const float* x, int32_t ldx, const int32_t* idx,
                                  const int32_t* goff, const float* W, const float* bias,
                                  float* pre, float* y, int32_t B, int32_t R, int32_t D,
                                  float slope, void* stream);
int32_t tnt_locally_dense_bwd_f32(const float* x, int32_t ldx, const int32_t* idx,
                                  const int32_t* goff, const float* dpre, float* dW, float* db,
                                  int32_t B, int32_t R, int32_t D, void* stream);

/* ---- additive attention step: attention.Attention.call (attention.py:25-44) -----
 * P = LeakyReLU(F @ W1 + b1) is loop-invariant and computed once with tnt_gemm_f32.
 * fwd: q = LeakyReLU(h @ W2 + b2); s = tanh(P + q); dropout(s); e = s.v + bv;
 *      alpha = softmax_R(e); ctx = sum_R alpha*F; ctx_d = LSTM-input dropout of ctx.
 *      Saves qpre[B][A], alpha[B][R]. s_out (nullable) [B][R][A] post-dropout.
 *      keep4 (nullable, fwd and bwd): this step's [B*R*A/4] bytes of tnt_dropout_mask4_u8 for (seed, site_attn, step);
 *      when given (A % 4 == 0) the kernels read the bits instead of running Philox -- bit-identical results. */
int32_t tnt_attention_step_fwd_f32(const float* h, const float* F, const float* P,
                                   const float* W2, const float* b2, const float* v,
                                   const float* bv, float* qpre, float* alpha, float* ctx,
                                   float* ctx_d, float* s_out, int32_t B, int32_t R, int32_t D,
                                   int32_t A, int32_t U, float slope, float rate_attn,
                                   float rate_in, int32_t in_lwidth, uint64_t seed,
                                   uint32_t site_attn, uint32_t site_in, uint32_t step,
                                   const uint32_t* step_dev, const uint8_t* keep4, void* stream);
/* bwd: given dctx_d (grad wrt the dropped ctx), accumulates dP[B][R][A] += , dF[B][R][D] +=,
 * dvb[B][A+1] += (per-sample partials of dV and dbV), writes dqpre[B][A] and
 * dh[B][U] = dqpre @ W2^T.  If dz != NULL the context gradient is computed in-kernel as
 * dctx_d[b][d] = sum_n dz[b][n] * Wc[d][n] (dz: this step's LSTM dz [B][4U]; Wc: the context
 * rows of the LSTM kernel [D][4U]) and the dctx_d argument is ignored.  If dctx_part is non-null ([nparts][B][D], written
 * by tnt_lstm_step_bwd_f32; nparts*D <= 1024) the context gradient is the sum of the parts and dctx_d / dz are ignored. */
/* alpha_mse_coef: dalpha[b][r] += alpha_mse_coef * (alpha[b][r] - 1) before the softmax backward -- the gradient of
 * c * sum (1 - alpha)^2 with alpha_mse_coef = 2c (lc_NIC.train_step_sam's attention term, lc_NIC.py:751-752); 0 = none.
 * fresh != 0: dP / dF / dvb are overwritten with this step's terms instead of added to (the first executed step of a
 * chain, so the caller needs no zero fill of the three accumulators). */
int32_t tnt_attention_step_bwd_f32(const float* dctx_d, const float* F, const float* P,
                                   const float* W2, const float* v, const float* qpre,
                                   const float* alpha, float* dP, float* dF, float* dvb,
                                   float* dqpre, float* dh, int32_t B, int32_t R, int32_t D,
                                   int32_t A, int32_t U, float slope, float rate_attn,
                                   float rate_in, int32_t in_lwidth, uint64_t seed,
                                   uint32_t site_attn, uint32_t site_in, uint32_t step,
                                   const uint32_t* step_dev, const float* dz, const float* Wc,
                                   const float* dctx_part, int32_t nparts, const uint8_t* keep4,
                                   float alpha_mse_coef, int32_t fresh, void* stream);

/* ---- the forward chain of the attention captioner as ONE persistent launch (lc_NIC.py:244-256): for i < T:
 * tnt_attention_step_fwd_f32(h = hs[i], ..., qpre[i], alpha[i], ctx[i], ctx_d[i], keep4 + i * keep_stride, sites + i) then
 * tnt_lstm_step_fwd_f32(xz[i], hs[i], cs[i], Ur, ctx_d[i], Wc, ... -> hs[i+1], cs[i+1], gates[i]) with xz_bias.
 * hs / cs [T+1][B][U] (slab 0 = initial state), xz [T][B][U][4], qpre [T][B][A], alpha [T][B][R], ctx / ctx_d [T][B][D],
 * gates [T][B][U][4].  U == 512, B <= 128, R <= 512, A % 4 == D % 4 == 0, A, D <= 64, the device census of
 * tnt_lstm_seq_supported.  work: tnt_lc_seq_fwd_work_floats(B) floats of exchange space (contents irrelevant: the partial
 * query sums the LSTM workgroups hand to the attention workgroups).  sync / guard_out: as tnt_lstm_seq_fwd_f32 (same state
 * words). */
int32_t tnt_lc_seq_fwd_work_floats(int32_t B);
int32_t tnt_lc_seq_fwd_f32(const float* F, const float* P, const float* W2, const float* b2, const float* v,
                           const float* bv, float* qpre, float* alpha, float* ctx, float* ctx_d, const uint8_t* keep4,
                           int64_t keep_stride, const float* xz, const float* Wc, const float* Ur, const float* xz_bias,
                           float* hs, float* cs, float* gates, int32_t T, int32_t B, int32_t R, int32_t D, int32_t A,
                           int32_t U, float slope, float rate_attn, float rate_in, int32_t in_lwidth, uint64_t seed,
                           uint32_t site_attn0, uint32_t site_in0, const uint32_t* step_dev, float* work,
                           uint32_t* sync, float* guard_out, void* stream);
/* the same launch with the Dropout behind the LSTM (lc_NIC.py:256) as a rider: hd [T][B][U] (nullable) receives
 * tnt_dropout_f32(hs[1:], rows_per_site = B, site = site_out0 + i for step i) as the states leave the chain. */
int32_t tnt_lc_seq_fwd_drop_f32(const float* F, const float* P, const float* W2, const float* b2, const float* v,
                                const float* bv, float* qpre, float* alpha, float* ctx, float* ctx_d,
                                const uint8_t* keep4, int64_t keep_stride, const float* xz, const float* Wc,
                                const float* Ur, const float* xz_bias, float* hs, float* cs, float* gates, int32_t T,
                                int32_t B, int32_t R, int32_t D, int32_t A, int32_t U, float slope, float rate_attn,
                                float rate_in, int32_t in_lwidth, uint64_t seed, uint32_t site_attn0,
                                uint32_t site_in0, const uint32_t* step_dev, float* hd, float rate_out,
                                uint32_t site_out0, float* work, uint32_t* sync, float* guard_out, void* stream);

/* ---- the backward chain of the attention captioner as ONE persistent launch (tape.gradient through lc_NIC.py:244-256):
 * for i = T-1 .. 0: tnt_lstm_step_bwd_f32(dz_next = dz[i+1], dh_ext = the attention's query gradient of step i+1,
 * dout_t = dout[i], gates[i], cs[i+1], cs[i] -> dz[i], context-gradient parts) then tnt_attention_step_bwd_f32(parts,
 * qpre[i], alpha[i], keep4 + i * keep_stride, sites + i -> dqpre[i], query gradient), with dP [B][R][A], dF [B][R][D] and
 * dvb [B][A+1] accumulated on chip over the T steps and WRITTEN once (no zero fill by the caller).
 * dout [T][B][U], gates / dz [T][B][U][4], cs [T+1][B][U], qpre / dqpre [T][B][A], alpha [T][B][R]; work:
 * tnt_lc_seq_bwd_work_floats(B, U) floats of exchange space (contents irrelevant).  Shape limits, sync and guard_out as
 * tnt_lc_seq_fwd_f32 (a sample's P, F, dP and dF rows live in registers: without spills for R <= 384 with A, D <= 32 and
 * for R <= 192 otherwise). */
int32_t tnt_lc_seq_bwd_work_floats(int32_t B, int32_t U);
int32_t tnt_lc_seq_bwd_f32(const float* F, const float* P, const float* W2, const float* v, const float* qpre,
                           const float* alpha, const uint8_t* keep4, int64_t keep_stride, float* dP, float* dF,
                           float* dvb, float* dqpre, const float* Ur, const float* Wc, const float* dout,
                           const float* gates, const float* cs, float* dz, float* work, int32_t T, int32_t B,
                           int32_t R, int32_t D, int32_t A, int32_t U, float slope, float rate_attn, float rate_in,
                           int32_t in_lwidth, uint64_t seed, uint32_t site_attn0, uint32_t site_in0,
                           const uint32_t* step_dev, float alpha_mse_coef, uint32_t* sync, float* guard_out,
                           void* stream);
/* the same launch with that Dropout's backward as a rider: dout is the gradient w.r.t. the DROPPED outputs and meets the
 * keep mask (rate_out, site_out0 + i for step i; the bits of tnt_lc_seq_fwd_drop_f32) as the chain reads it. */
int32_t tnt_lc_seq_bwd_drop_f32(const float* F, const float* P, const float* W2, const float* v, const float* qpre,
                                const float* alpha, const uint8_t* keep4, int64_t keep_stride, float* dP, float* dF,
                                float* dvb, float* dqpre, const float* Ur, const float* Wc, const float* dout,
                                const float* gates, const float* cs, float* dz, float* work, int32_t T, int32_t B,
                                int32_t R, int32_t D, int32_t A, int32_t U, float slope, float rate_attn,
                                float rate_in, int32_t in_lwidth, uint64_t seed, uint32_t site_attn0,
                                uint32_t site_in0, const uint32_t* step_dev, float alpha_mse_coef, float rate_out,
                                uint32_t site_out0, uint32_t* sync, float* guard_out, void* stream);

/* ---- the attention layer's hoisted first Dense (P = LeakyReLU(F W1 + b1), attention.py:32) backward, behind the chain:
 * with dP [rows][A] the score gradient accumulated over the T steps and Ppre its pre-activation:
 *   g = dP * LeakyReLU'(Ppre, slope);  db1 = column sums of g;  dW1 [D][A] = F^T g;  dF [rows][D] += g W1^T
 * (lc_NIC.py:386-387) in two launches; g is not stored.  part: tnt_attention_front_bwd_parts(rows, D, A) floats of
 * scratch.  D = A = 32 (the reference's attention width; other sizes: TNT_BADARG, use the generic entry points).
 * Deterministic (partials summed in row-chunk order). */
int32_t tnt_attention_front_bwd_parts(int32_t rows, int32_t D, int32_t A);
int32_t tnt_attention_front_bwd_f32(const float* Ppre, const float* dP, const float* F, const float* W1, float* dF,
                                    float* dW1, float* db1, float* part, int32_t rows, int32_t D, int32_t A,
                                    float slope, void* stream);
/* ... and, with drop_rate > 0, the backward of the feature Dropout that produced F (layers.py:51) applied to the finished
 * dF in the same pass: dF = keep ? (dF + g W1^T) / (1 - rate) : 0, element row*D + d of stream (drop_seed, drop_site,
 * *drop_step_dev). */
int32_t tnt_attention_front_bwd_drop_f32(const float* Ppre, const float* dP, const float* F, const float* W1, float* dF,
                                         float* dW1, float* db1, float* part, int32_t rows, int32_t D, int32_t A,
                                         float slope, float drop_rate, uint64_t drop_seed, uint32_t drop_site,
                                         const uint32_t* drop_step_dev, void* stream);

/* attention "coverage" metric (lc_NIC.py:365-367): mean over (T,R) of
 * (1 - sum_b alpha[t][b][r])^2.  alpha is [T][B][R].  work: tnt_attention_metric_parts(T, R) floats of partial sums;
 * out == NULL: only the partials are written, and the metric is their sum times 1 / (T R) (taken by
 * tnt_step_finalize_f32's x2 job inside the fused training step). */
int32_t tnt_attention_metric_parts(int32_t T, int32_t R);
int32_t tnt_attention_metric_f32(const float* alpha, float* out, float* work,
                                 int32_t T, int32_t B, int32_t R,
                                 int64_t tstride /* floats between timesteps; 0 = B*R */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TNT_HIP_H */
