// HBM-bound row/column kernels: dropout, activation backward, BatchNorm, LayerNorm,
// column sums, scalar sum.  Reference ops: keras Dropout (lc_NIC.py:51-55,94),
// BatchNormalization (layers.py:40,50; NIC.py:62,128; fullyConnected.py:18,24),
// LayerNormalization (layers.py:41), bias gradients of every Dense.
// All reductions are fixed-order (no float atomics) so results are bitwise reproducible.
#include "tnt_common.h"
#include <hip/hip_fp16.h>
#include "tnt_rng.h"

namespace {

// column reductions run in two levels: <= MAX_CHUNKS row chunks -> per-column finalize
constexpr int MAX_CHUNKS = 128;
inline int chunk_count(int rows) { int n = (rows + 63) / 64; return n > MAX_CHUNKS ? MAX_CHUNKS : (n < 1 ? 1 : n); }
inline int chunk_rows(int rows) { const int n = chunk_count(rows); return (rows + n - 1) / n; }

// ------------------------------------------------------------------------- dropout
struct DropArgs {
  const float* x; float* y; int rows, cols, ld, tB, lwidth, lcol0, rows_per_site;
  float rate, scale; uint64_t seed; uint32_t site, step; const uint32_t* step_dev;
};

// logical row / site of buffer row r
__device__ __forceinline__ void drop_row(const DropArgs& a, int r, int T, long& lrow, uint32_t& site) {
  site = a.site;
  if (a.rows_per_site > 0) { site += (uint32_t)(r / a.rows_per_site); r = r % a.rows_per_site; }
  lrow = a.tB > 0 ? (long)(r % a.tB) * T + r / a.tB : r;
}

__global__ __launch_bounds__(256) void dropout_kernel(DropArgs a) {
  const uint32_t step = a.step + (a.step_dev ? a.step_dev[0] : 0u);
  const long total = (long)a.rows * a.cols;
  const int T = a.tB > 0 ? a.rows / a.tB : 0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / a.cols), c = (int)(e % a.cols);
    long lrow; uint32_t site;
    drop_row(a, r, T, lrow, site);
    const uint64_t le = (uint64_t)lrow * (uint64_t)a.lwidth + (uint64_t)(a.lcol0 + c);
    const bool k = tnt_keep(le, a.rate, a.seed, site, step);
    const long o = (long)r * a.ld + c;
    a.y[o] = k ? a.x[o] * a.scale : 0.f;
  }
}

// 4 consecutive columns per thread: one Philox call yields the 4 decisions (cols, ld, lwidth, lcol0 % 4 == 0)
__global__ __launch_bounds__(256) void dropout4_kernel(DropArgs a) {
  const uint32_t step = a.step + (a.step_dev ? a.step_dev[0] : 0u);
  const int c4n = a.cols >> 2;
  const long total = (long)a.rows * c4n;
  const int T = a.tB > 0 ? a.rows / a.tB : 0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / c4n), c = (int)(e % c4n) * 4;
    long lrow; uint32_t site;
    drop_row(a, r, T, lrow, site);
    const uint64_t le = (uint64_t)lrow * (uint64_t)a.lwidth + (uint64_t)(a.lcol0 + c);
    bool k[4];
    tnt_keep4(le, a.rate, a.seed, site, step, k);
    const long o = (long)r * a.ld + c;
    const float4 v = *reinterpret_cast<const float4*>(a.x + o);
    float4 w;
    w.x = k[0] ? v.x * a.scale : 0.f; w.y = k[1] ? v.y * a.scale : 0.f;
    w.z = k[2] ? v.z * a.scale : 0.f; w.w = k[3] ? v.w * a.scale : 0.f;
    *reinterpret_cast<float4*>(a.y + o) = w;
  }
}

// two masks in one pass (y = mask_b(mask_a(x)), the bits of two tnt_dropout_f32 launches): same matrix, two logical layouts
__global__ __launch_bounds__(256) void dropout4x2_kernel(DropArgs a, DropArgs b) {
  const uint32_t step = a.step + (a.step_dev ? a.step_dev[0] : 0u);
  const int c4n = a.cols >> 2;
  const long total = (long)a.rows * c4n;
  const int Ta = a.tB > 0 ? a.rows / a.tB : 0, Tb = b.tB > 0 ? b.rows / b.tB : 0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / c4n), c = (int)(e % c4n) * 4;
    long lrow; uint32_t site;
    bool ka[4], kb[4];
    drop_row(a, r, Ta, lrow, site);
    tnt_keep4((uint64_t)lrow * (uint64_t)a.lwidth + (uint64_t)(a.lcol0 + c), a.rate, a.seed, site, step, ka);
    drop_row(b, r, Tb, lrow, site);
    tnt_keep4((uint64_t)lrow * (uint64_t)b.lwidth + (uint64_t)(b.lcol0 + c), b.rate, b.seed, site, step, kb);
    const long o = (long)r * a.ld + c;
    const float4 v = *reinterpret_cast<const float4*>(a.x + o);
    float4 w;
    w.x = ka[0] ? v.x * a.scale : 0.f; w.y = ka[1] ? v.y * a.scale : 0.f;
    w.z = ka[2] ? v.z * a.scale : 0.f; w.w = ka[3] ? v.w * a.scale : 0.f;
    w.x = kb[0] ? w.x * b.scale : 0.f; w.y = kb[1] ? w.y * b.scale : 0.f;
    w.z = kb[2] ? w.z * b.scale : 0.f; w.w = kb[3] ? w.w * b.scale : 0.f;
    *reinterpret_cast<float4*>(a.y + o) = w;
  }
}

// Keep-masks only, one byte per 4 consecutive logical elements (bit j = element 4g + j kept), for `nsites`
// consecutive sites of n4 groups each.  Philox4x32-10 is 40 quarter-rate integer multiplies per call: ~2 us inside each
// of the per-timestep attention kernels, which run on B of the 256 CUs on the serial chain, but ~10 us for all T
// timesteps at once when spread over the whole chip (and the backward reuses the bytes).
__global__ __launch_bounds__(256) void dropout_mask4_kernel(uint8_t* out, long n4, long total, float rate, uint64_t seed,
                                                            uint32_t site0, uint32_t step, const uint32_t* step_dev) {
  const uint32_t st = step + (step_dev ? step_dev[0] : 0u);
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const long k = g / n4, gi = g - k * n4;
    bool kp[4];
    tnt_keep4((uint64_t)gi * 4u, rate, seed, site0 + (uint32_t)k, st, kp);
    out[g] = (uint8_t)((kp[0] ? 1 : 0) | (kp[1] ? 2 : 0) | (kp[2] ? 4 : 0) | (kp[3] ? 8 : 0));
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* pre, const float* dy, float* dx, long n, int act,
                                                      float slope) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
    dx[e] = tnt_act_grad(pre[e], dy[e], act, slope);
}

// --------------------------------------------------------------- column partials
// grid (ceil(C/CL), nchunk), block 256 = CL column lanes x (256/CL) row lanes.
// MODE 0: Welford (mean, M2) of x          -> work[chunk][0][c]=mean, [1][c]=M2
// MODE 1: sums of a and a*b                -> work[chunk][0][c]=sum a, [1][c]=sum a*b
// MODE 2: sum of a only                    -> work[chunk][0][c]
template <int MODE, int CL>
__global__ __launch_bounds__(256) void col_partial_kernel(const float* a, int lda, const float* b, int ldb, int rows,
                                                          int C, int crows, float* work) {
  constexpr int RL = 256 / CL;
  __shared__ float s0[RL][CL], s1[RL][CL], sn[RL][CL];
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int c = blockIdx.x * CL + cl;
  const int chunk = blockIdx.y;
  const int r0 = chunk * crows, r1 = min(rows, r0 + crows);
  float v0 = 0.f, v1 = 0.f, n = 0.f;
  if (c < C) {
    // MODE 0: sums of (x - K) and (x - K)^2 with K = the thread's first row (shifted sums: no catastrophic cancellation
    // in S2 - S1^2 / n, and no per-element division on a serial chain as in the textbook Welford update)
    const float K = (MODE == 0 && r0 + rl < r1) ? a[(long)(r0 + rl) * lda + c] : 0.f;
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += RL) {
      const float x = a[(long)r * lda + c];
      if (MODE == 0) {
        const float d = x - K;
        n += 1.f; v0 += d; v1 += d * d;
      } else if (MODE == 1) {
        v0 += x;
        v1 += x * b[(long)r * ldb + c];
      } else {
        v0 += x;
      }
    }
  }
  if (MODE == 0 && n > 0.f) {          // (count, mean, M2) of this thread's rows from the shifted sums
    const float K = a[(long)(r0 + rl) * lda + c], m = v0 / n;
    v1 = v1 - v0 * m;                  // sum (x - mean)^2 = S2 - S1^2 / n
    v0 = K + m;
  }
  s0[rl][cl] = v0; s1[rl][cl] = v1; sn[rl][cl] = n;
  __syncthreads();
  if (rl == 0 && c < C) {
    if (MODE == 0) {
      float nn = sn[0][cl], mean = s0[0][cl], m2 = s1[0][cl];
      for (int j = 1; j < RL; ++j) {
        const float nj = sn[j][cl];
        if (nj > 0.f) {
          const float d = s0[j][cl] - mean, nt = nn + nj;
          mean += d * nj / nt;
          m2 += s1[j][cl] + d * d * nn * nj / nt;
          nn = nt;
        }
      }
      work[((long)chunk * 2 + 0) * C + c] = mean;
      work[((long)chunk * 2 + 1) * C + c] = m2;
    } else {
      float t0 = 0.f, t1 = 0.f;
      for (int j = 0; j < RL; ++j) { t0 += s0[j][cl]; t1 += s1[j][cl]; }
      if (MODE == 1) {
        work[((long)chunk * 2 + 0) * C + c] = t0;
        work[((long)chunk * 2 + 1) * C + c] = t1;
      } else {
        work[(long)chunk * C + c] = t0;
      }
    }
  }
}

template <int MODE>
void launch_col_partial(const float* a, int lda, const float* b, int ldb, int rows, int C, float* work, hipStream_t s) {
  const int nchunk = chunk_count(rows), crows = chunk_rows(rows);
  if (C <= 32)
    hipLaunchKernelGGL((col_partial_kernel<MODE, 32>), dim3((C + 31) / 32, nchunk), dim3(256), 0, s, a, lda, b, ldb, rows,
                       C, crows, work);
  else
    hipLaunchKernelGGL((col_partial_kernel<MODE, 64>), dim3((C + 63) / 64, nchunk), dim3(256), 0, s, a, lda, b, ldb, rows,
                       C, crows, work);
}

// BN statistics finalize (training), one wave per column, lanes over the chunks: mean = sum n_j m_j / N, then
// M2 = sum (M2_j + n_j (m_j - mean)^2); fixed order (lane-strided partial sums + the shuffle tree of tnt_wave_sum).
// nrep > 1 (synchronised BatchNorm): `work` holds nrep replicas' chunk partials back to back (all-gathered), each of `rows`
// rows: the statistics are those of the nrep * rows rows.
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float* work, int rows, int C, int nchunk1, int crows,
                                                         float eps, float momentum, float* mov_mean, float* mov_var,
                                                         float* mean_out, float* inv_std, int nrep) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c >= C) return;
  const int nchunk = nchunk1 * nrep;
  float nn = 0.f, ms = 0.f;
  for (int k = lane; k < nchunk; k += 64) {
    const int k1 = k % nchunk1;
    const float nj = (float)(min(rows, (k1 + 1) * crows) - k1 * crows);
    if (nj <= 0.f) continue;
    nn += nj;
    ms += nj * work[((long)k * 2 + 0) * C + c];
  }
  nn = tnt_wave_sum(nn); ms = tnt_wave_sum(ms);
  const float mean = ms / nn;
  float m2 = 0.f;
  for (int k = lane; k < nchunk; k += 64) {
    const int k1 = k % nchunk1;
    const float nj = (float)(min(rows, (k1 + 1) * crows) - k1 * crows);
    if (nj <= 0.f) continue;
    const float d = work[((long)k * 2 + 0) * C + c] - mean;
    m2 += work[((long)k * 2 + 1) * C + c] + nj * d * d;
  }
  m2 = tnt_wave_sum(m2);
  if (lane != 0) return;
  const float var = m2 / nn;
  mean_out[c] = mean;
  inv_std[c] = 1.f / sqrtf(var + eps);
  mov_mean[c] = mov_mean[c] * momentum + mean * (1.f - momentum);
  mov_var[c] = mov_var[c] * momentum + var * (1.f - momentum);
}

__global__ void bn_infer_prep_kernel(const float* mov_mean, const float* mov_var, int C, float eps, float* mean_out,
                                     float* inv_std) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  mean_out[c] = mov_mean[c];
  inv_std[c] = 1.f / sqrtf(mov_var[c] + eps);
}

// y = BN(x) (and xhat for the backward); rate > 0: the keras Dropout behind the normalisation (layers.py:50-51) in the same
// pass -- y = keep ? BN(x) / (1 - rate) : 0, element r * C + c of stream (seed, site, step)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* x, const float* mean, const float* inv_std,
                                                       const float* gamma, const float* beta, float* y, float* xhat,
                                                       int rows, int C, int ldy, float rate, uint64_t seed, uint32_t site,
                                                       const uint32_t* step_dev) {
  const long total = (long)rows * C;
  const uint32_t step = step_dev ? step_dev[0] : 0u;
  const float scale = 1.f / (1.f - rate);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / C), c = (int)(e % C);
    const float xh = (x[e] - mean[c]) * inv_std[c];
    xhat[e] = xh;
    float v = xh * gamma[c] + beta[c];
    if (rate > 0.f) v = tnt_keep((uint64_t)e, rate, seed, site, step) ? v * scale : 0.f;
    y[(long)r * ldy + c] = v;
  }
}

// 4 consecutive columns per thread (C, ldy % 4 == 0, 16-byte aligned): one Philox call per float4
__global__ __launch_bounds__(256) void bn_apply4_kernel(const float* x, const float* mean, const float* inv_std,
                                                        const float* gamma, const float* beta, float* y, float* xhat,
                                                        int rows, int C, int ldy, float rate, uint64_t seed, uint32_t site,
                                                        const uint32_t* step_dev) {
  const int c4n = C >> 2;
  const long total = (long)rows * c4n;
  const uint32_t step = step_dev ? step_dev[0] : 0u;
  const float scale = 1.f / (1.f - rate);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / c4n), c = (int)(e % c4n) * 4;
    const float4 xv = *reinterpret_cast<const float4*>(x + e * 4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(inv_std + c);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
    const float4 xh = make_float4((xv.x - mu.x) * is.x, (xv.y - mu.y) * is.y, (xv.z - mu.z) * is.z, (xv.w - mu.w) * is.w);
    *reinterpret_cast<float4*>(xhat + e * 4) = xh;
    float4 v = make_float4(xh.x * ga.x + be.x, xh.y * ga.y + be.y, xh.z * ga.z + be.z, xh.w * ga.w + be.w);
    if (rate > 0.f) {
      bool k[4];
      tnt_keep4((uint64_t)e * 4, rate, seed, site, step, k);
      v.x = k[0] ? v.x * scale : 0.f; v.y = k[1] ? v.y * scale : 0.f;
      v.z = k[2] ? v.z * scale : 0.f; v.w = k[3] ? v.w * scale : 0.f;
    }
    *reinterpret_cast<float4*>(y + (long)r * ldy + c) = v;
  }
}

// sums over chunks: out0[c] = sum_k work[k][0][c] (and out1 from slot 1 if TWO); one wave per column, lanes over chunks
template <bool TWO>
__global__ __launch_bounds__(64) void col_finalize_kernel(const float* work, int C, int nchunk, float* out0, float* out1) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c >= C) return;
  float t0 = 0.f, t1 = 0.f;
  for (int k = lane; k < nchunk; k += 64) {
    if (TWO) {
      t0 += work[((long)k * 2 + 0) * C + c];
      t1 += work[((long)k * 2 + 1) * C + c];
    } else {
      t0 += work[(long)k * C + c];
    }
  }
  t0 = tnt_wave_sum(t0);
  if (TWO) t1 = tnt_wave_sum(t1);
  if (lane == 0) {
    out0[c] = t0;
    if (TWO) out1[c] = t1;
  }
}

// dx = gamma*inv/n * (n*dy - dbeta - xhat*dgamma)   (training)  |  gamma*inv*dy (inference)
// act_pre != NULL: dx is further multiplied by LeakyReLU'(act_pre) -- the activation in front of the normalisation
// (layers.py:48-50), whose backward would otherwise be one more elementwise launch over the same matrix
__global__ __launch_bounds__(256) void bn_dx_kernel(const float* dy, int lddy, const float* xhat, const float* gamma,
                                                    const float* inv_std, const float* dgamma, const float* dbeta,
                                                    float* dx, int rows, int C, int training, int n_total,
                                                    const float* act_pre, float slope) {
  const long total = (long)rows * C;
  const float n = (float)n_total;       // rows of the whole (possibly cross-replica) batch the sums were taken over
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int r = (int)(e / C), c = (int)(e % C);
    const float g = dy[(long)r * lddy + c];
    const float k = gamma[c] * inv_std[c];
    float d = training ? k / n * (n * g - dbeta[c] - xhat[e] * dgamma[c]) : k * g;
    if (act_pre) d = tnt_act_grad(act_pre[e], d, 1, slope);
    dx[e] = d;
  }
}

// ----------------------------------------------------------------------- LayerNorm
// one wave per row
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* x, const float* gamma, const float* beta, float* y,
                                                     float* xhat, float* inv_std, int rows, int C, int ldy, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = tnt_wave_sum(s) / C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
  const float inv = 1.f / sqrtf(tnt_wave_sum(q) / C + eps);
  if (lane == 0) inv_std[row] = inv;
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - mean) * inv;
    xhat[(long)row * C + c] = xh;
    y[(long)row * ldy + c] = xh * gamma[c] + beta[c];
  }
}

__global__ __launch_bounds__(256) void ln_dx_kernel(const float* dy, int lddy, const float* xhat, const float* gamma,
                                                    const float* inv_std, float* dx, int rows, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s0 = 0.f, s1 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = dy[(long)row * lddy + c] * gamma[c];
    s0 += d;
    s1 += d * xhat[(long)row * C + c];
  }
  s0 = tnt_wave_sum(s0); s1 = tnt_wave_sum(s1);
  const float k = inv_std[row] / C;
  for (int c = lane; c < C; c += 64) {
    const float d = dy[(long)row * lddy + c] * gamma[c];
    dx[(long)row * C + c] = k * (C * d - s0 - xhat[(long)row * C + c] * s1);
  }
}

__global__ __launch_bounds__(1024) void sum_kernel(const float* x, float* out, int n, float scale) {
  __shared__ float sw[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += x[i];
  s = tnt_wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sw[w];
    out[0] = t * scale;
  }
}

// two independent sums in one launch (loss and accuracy of a step): block b reduces x[b] into out[b]
__global__ __launch_bounds__(1024) void sum2_kernel(const float* x0, float* out0, const float* x1, float* out1, int n,
                                                    float scale) {
  __shared__ float sw[16];
  const float* x = blockIdx.x == 0 ? x0 : x1;
  float* out = blockIdx.x == 0 ? out0 : out1;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += x[i];
  s = tnt_wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sw[w];
    out[0] = t * scale;
  }
}

// One launch for the per-step input staging of a batch that is already on the device
// (data_generator_guse.py:156-171 tuple -> the static buffers of the captured step).
// XT = float, or __half when the betas crossed PCIe as IEEE half ("fp16 on-wire", SURVEY 8f rank 1: at full-cortex
// width the 84 MB float batch is what bounds the step); the widening to float happens here, in the same pass.
template <typename XT>
struct StageArgs {
  const XT* x; float* xd; const int* cap; int* capd; const int* tgt; int* tgtd;
  const float* a0; float* h0; const float* c0; float* c0d;
  int B, T, N, ldx, U;
  float* xT; int ldt; int ncopy;      // optional voxel-major copy xT[N][ldt] (blocks >= ncopy transpose 64x64 tiles)
  // optional riding job (blocks >= nstage): the keep masks of tnt_dropout_mask4_u8 for the step that this batch feeds
  uint8_t* mk_out; long mk_n4, mk_total; float mk_rate; uint64_t mk_seed; uint32_t mk_site0; const uint32_t* mk_step_dev;
  int nstage;
};
__device__ __forceinline__ float stage_ld(const float* p) { return *p; }
__device__ __forceinline__ float stage_ld(const __half* p) { return __half2float(*p); }
__device__ __forceinline__ float4 stage_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 stage_ld4(const __half* p) {          // 8-byte aligned: 4 halves
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  const __half2 lo = *reinterpret_cast<const __half2*>(&u.x), hi = *reinterpret_cast<const __half2*>(&u.y);
  const float2 a = __half22float2(lo), b = __half22float2(hi);
  return make_float4(a.x, a.y, b.x, b.y);
}
template <typename XT>
__global__ __launch_bounds__(256) void stage_batch_kernel(StageArgs<XT> a) {
  if ((int)blockIdx.x >= a.nstage) {          // Philox-bound, the copies around it memory-bound: they share the chip
    const uint32_t st = a.mk_step_dev ? a.mk_step_dev[0] : 0u;
    const long nb = gridDim.x - a.nstage;
    for (long g = (long)(blockIdx.x - a.nstage) * 256 + threadIdx.x; g < a.mk_total; g += nb * 256) {
      const long k = g / a.mk_n4, gi = g - k * a.mk_n4;
      bool kp[4];
      tnt_keep4((uint64_t)gi * 4u, a.mk_rate, a.mk_seed, a.mk_site0 + (uint32_t)k, st, kp);
      a.mk_out[g] = (uint8_t)((kp[0] ? 1 : 0) | (kp[1] ? 2 : 0) | (kp[2] ? 4 : 0) | (kp[3] ? 8 : 0));
    }
    return;
  }
  if ((int)blockIdx.x >= a.ncopy) {
    // voxel-major copy for the region-wise encoder's gather: tile = 64 voxels x 64 batch rows through LDS
    __shared__ float tile[64][65];
    const int ntc = (a.N + 63) / 64, ntr = (a.B + 63) / 64;
    for (int t = blockIdx.x - a.ncopy; t < ntc * ntr; t += a.nstage - a.ncopy) {
      const int c0 = (t % ntc) * 64, r0 = (t / ntc) * 64;
      __syncthreads();
      for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        tile[r][c] = (r0 + r < a.B && c0 + c < a.N) ? stage_ld(a.x + (long)(r0 + r) * a.N + c0 + c) : 0.f;
      }
      __syncthreads();
      for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int c = e >> 6, r = e & 63;
        if (c0 + c < a.N && r0 + r < a.ldt) a.xT[(long)(c0 + c) * a.ldt + r0 + r] = tile[r][c];
      }
    }
    return;
  }
  const long gid = (long)blockIdx.x * 256 + threadIdx.x, gsz = (long)a.ncopy * 256;
  if (a.N % 4 == 0 && a.ldx % 4 == 0) {
    const int n4 = a.N / 4;
    for (long e = gid; e < (long)a.B * n4; e += gsz) {
      const int r = (int)(e / n4), c = (int)(e % n4) * 4;
      *reinterpret_cast<float4*>(a.xd + (long)r * a.ldx + c) = stage_ld4(a.x + (long)r * a.N + c);
    }
  } else {
    for (long e = gid; e < (long)a.B * a.N; e += gsz) {
      const int r = (int)(e / a.N), c = (int)(e % a.N);
      a.xd[(long)r * a.ldx + c] = stage_ld(a.x + e);
    }
  }
  const int bt = a.B * a.T, bu = a.B * a.U;
  for (long e = gid; e < bt; e += gsz) {
    a.capd[e] = a.cap[e];
    if (a.tgt) { const int b = (int)(e / a.T), t = (int)(e % a.T); a.tgtd[t * a.B + b] = a.tgt[e]; }
  }
  for (long e = gid; e < bu; e += gsz) { a.h0[e] = a.a0[e]; a.c0d[e] = a.c0[e]; }
}

inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int32_t tnt_bn_nchunk(int32_t rows) { return chunk_count(rows); }

extern "C" int32_t tnt_dropout_f32(const float* x, float* y, int32_t rows, int32_t cols, int32_t ld, int32_t tmajor_B,
                                   int32_t lwidth, int32_t lcol0, int32_t rows_per_site, float rate, uint64_t seed,
                                   uint32_t site, uint32_t step, const uint32_t* step_dev, void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  const int rsite = rows_per_site > 0 ? rows_per_site : rows;
  if (tmajor_B > 0 && rsite % tmajor_B != 0) return TNT_BADARG(6);
  if (rows_per_site > 0 && tmajor_B > 0) return TNT_BADARG(9);     // per-site blocks are (rows_per_site, cols) row-major
  DropArgs a;
  a.x = x; a.y = y; a.rows = rows; a.cols = cols; a.ld = ld; a.tB = tmajor_B; a.lwidth = lwidth; a.lcol0 = lcol0;
  a.rows_per_site = rows_per_site; a.rate = rate; a.scale = 1.0f / (1.0f - rate); a.seed = seed; a.site = site;
  a.step = step; a.step_dev = step_dev;
  const bool vec = ((cols | ld | lwidth | lcol0) & 3) == 0 && tnt_aligned16(x) && tnt_aligned16(y);
  if (vec)
    hipLaunchKernelGGL(dropout4_kernel, dim3(ew_blocks((long)rows * (cols / 4))), dim3(256), 0, tnt_stream(stream), a);
  else
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_blocks((long)rows * cols)), dim3(256), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

// tnt_dropout_f32 (vector form) with the attention metric's partials as a rider: blocks [0, nd) are dropout4_kernel with a
// grid of nd, blocks [nd, nd + T * nc) are attention_metric_kernel's workgroup (t, c) -- partial[t * nc + c] = sum over its
// 64 regions r of (1 - sum_b alpha[t][b][r])^2, the same summation order (attention.hip).
__global__ __launch_bounds__(256) void dropout4_metric_kernel(DropArgs a, int nd, const float* alpha, float* partial, int B,
                                                              int R, int nc) {
  if ((int)blockIdx.x >= nd) {
    __shared__ float sb[4][64];
    const int k = blockIdx.x - nd, t = k / nc, c = k % nc, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = c * 64 + lane;
    float s = 0.f;
    if (r < R) {
#pragma unroll 8
      for (int b = w; b < B; b += 4) s += alpha[((long)t * B + b) * R + r];
    }
    sb[w][lane] = s;
    __syncthreads();
    if (w != 0) return;
    float acc = 0.f;
    if (r < R) {
      const float tot = ((sb[0][lane] + sb[1][lane]) + sb[2][lane]) + sb[3][lane];
      acc = (1.f - tot) * (1.f - tot);
    }
    acc = tnt_wave_sum(acc);
    if (lane == 0) partial[k] = acc;
    return;
  }
  const uint32_t step = a.step + (a.step_dev ? a.step_dev[0] : 0u);
  const int c4n = a.cols >> 2;
  const long total = (long)a.rows * c4n;
  const int T = a.tB > 0 ? a.rows / a.tB : 0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)nd * 256) {
    const int r = (int)(e / c4n), c = (int)(e % c4n) * 4;
    long lrow; uint32_t site;
    drop_row(a, r, T, lrow, site);
    const uint64_t le = (uint64_t)lrow * (uint64_t)a.lwidth + (uint64_t)(a.lcol0 + c);
    bool k[4];
    tnt_keep4(le, a.rate, a.seed, site, step, k);
    const long o = (long)r * a.ld + c;
    const float4 v = *reinterpret_cast<const float4*>(a.x + o);
    float4 w;
    w.x = k[0] ? v.x * a.scale : 0.f; w.y = k[1] ? v.y * a.scale : 0.f;
    w.z = k[2] ? v.z * a.scale : 0.f; w.w = k[3] ? v.w * a.scale : 0.f;
    *reinterpret_cast<float4*>(a.y + o) = w;
  }
}

extern "C" int32_t tnt_dropout_metric_f32(const float* x, float* y, int32_t rows, int32_t cols, int32_t ld, int32_t tmajor_B,
                                          int32_t lwidth, int32_t lcol0, int32_t rows_per_site, float rate, uint64_t seed,
                                          uint32_t site, uint32_t step, const uint32_t* step_dev, const float* alpha,
                                          float* partial, int32_t T, int32_t B, int32_t R, void* stream) {
  if (rows <= 0 || cols <= 0 || T <= 0 || B <= 0 || R <= 0 || alpha == nullptr || partial == nullptr) return TNT_BADARG(3);
  const int rsite = rows_per_site > 0 ? rows_per_site : rows;
  if (tmajor_B > 0 && rsite % tmajor_B != 0) return TNT_BADARG(6);
  if (rows_per_site > 0 && tmajor_B > 0) return TNT_BADARG(9);
  if (((cols | ld | lwidth | lcol0) & 3) != 0 || !tnt_aligned16(x) || !tnt_aligned16(y)) return TNT_BADARG(1);
  if (!(rate >= 0.f && rate < 1.f)) return TNT_BADARG(10);
  DropArgs a;
  a.x = x; a.y = y; a.rows = rows; a.cols = cols; a.ld = ld; a.tB = tmajor_B; a.lwidth = lwidth; a.lcol0 = lcol0;
  a.rows_per_site = rows_per_site; a.rate = rate; a.scale = 1.0f / (1.0f - rate); a.seed = seed; a.site = site;
  a.step = step; a.step_dev = step_dev;
  const int nd = ew_blocks((long)rows * (cols / 4)), nc = (R + 63) / 64;
  hipLaunchKernelGGL(dropout4_metric_kernel, dim3(nd + T * nc), dim3(256), 0, tnt_stream(stream), a, nd, alpha, partial, B, R,
                     nc);
  TNT_LAUNCH_CHECK();
  return 0;
}

// y = mask_b(mask_a(x)) in one launch: two Dropout masks over the same [rows][cols] matrix, each with its own logical layout
// (tmajor_B / lwidth / lcol0 / rows_per_site as in tnt_dropout_f32), rate and site; same seed and step.  Vectorised only:
// cols, ld, lwidth*, lcol0* % 4 == 0 and 16-byte aligned x / y (else TNT_BADARG: issue two tnt_dropout_f32 calls).
extern "C" int32_t tnt_dropout2_f32(const float* x, float* y, int32_t rows, int32_t cols, int32_t ld, int32_t tmajor_B_a,
                                    int32_t lwidth_a, int32_t lcol0_a, int32_t rows_per_site_a, float rate_a, uint32_t site_a,
                                    int32_t tmajor_B_b, int32_t lwidth_b, int32_t lcol0_b, int32_t rows_per_site_b,
                                    float rate_b, uint32_t site_b, uint64_t seed, uint32_t step, const uint32_t* step_dev,
                                    void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  if (!(rate_a > 0.f && rate_a < 1.f && rate_b > 0.f && rate_b < 1.f)) return TNT_BADARG(10);
  const int tb[2] = {tmajor_B_a, tmajor_B_b}, rps[2] = {rows_per_site_a, rows_per_site_b};
  for (int k = 0; k < 2; ++k) {
    const int rsite = rps[k] > 0 ? rps[k] : rows;
    if (tb[k] > 0 && rsite % tb[k] != 0) return TNT_BADARG(6);
    if (rps[k] > 0 && tb[k] > 0) return TNT_BADARG(9);
  }
  if (((cols | ld | lwidth_a | lcol0_a | lwidth_b | lcol0_b) & 3) != 0 || !tnt_aligned16(x) || !tnt_aligned16(y))
    return TNT_BADARG(3);
  DropArgs a, b;
  a.x = x; a.y = y; a.rows = rows; a.cols = cols; a.ld = ld; a.tB = tmajor_B_a; a.lwidth = lwidth_a; a.lcol0 = lcol0_a;
  a.rows_per_site = rows_per_site_a; a.rate = rate_a; a.scale = 1.0f / (1.0f - rate_a); a.seed = seed; a.site = site_a;
  a.step = step; a.step_dev = step_dev;
  b = a;
  b.tB = tmajor_B_b; b.lwidth = lwidth_b; b.lcol0 = lcol0_b; b.rows_per_site = rows_per_site_b; b.rate = rate_b;
  b.scale = 1.0f / (1.0f - rate_b); b.site = site_b;
  hipLaunchKernelGGL(dropout4x2_kernel, dim3(ew_blocks((long)rows * (cols / 4))), dim3(256), 0, tnt_stream(stream), a, b);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_dropout_mask4_u8(uint8_t* out, int64_t n, int32_t nsites, float rate, uint64_t seed,
                                       uint32_t site0, uint32_t step, const uint32_t* step_dev, void* stream) {
  if (n <= 0 || nsites <= 0) return 0;
  if (n % 4 != 0) return TNT_BADARG(2);
  const long n4 = n / 4, total = n4 * nsites;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dropout_mask4_kernel, dim3((unsigned)blocks), dim3(256), 0, tnt_stream(stream), out, n4, total, rate,
                     seed, site0, step, step_dev);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_act_bwd_f32(const float* pre, const float* dy, float* dx, int64_t n, int32_t act, float slope,
                                   void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, tnt_stream(stream), pre, dy, dx, (long)n, act,
                     slope);
  TNT_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------ fused dense-encoder tail
// NIC.py:126-128,138 at small batch: Dropout(features) -> BatchNorm (batch statistics) -> the LSTM layer's
// input dropout of the feature step, and its backward (+ LeakyReLU' + encoder bias gradient), as ONE launch each
// instead of 5 / 7 dependent small kernels (each costs ~3 us of launch latency in the captured step).
// A workgroup owns 32 columns; 8 row groups x 32 columns of threads; every thread keeps its <= MAXR rows in
// registers, column reductions go through LDS in a fixed order.
namespace {
constexpr int ET_CW = 32, ET_RG = 32, ET_MAXR = 8;      // rows <= ET_RG * ET_MAXR = 256; C % 4 == 0
// thread = (row group rg = tid / 8, column quad c4 = tid % 8): one float4 (and one Philox call) per 4 columns

struct EncTailArgs {
  // forward
  const float* y; const float* gamma; const float* beta; float* mov_mean; float* mov_var;
  float* out; float* xhat; float* inv_std;
  // backward
  const float* dout; const float* pre; float* dpre; float* dgamma; float* dbeta; float* dbias;
  int rows, C, ldo, training;
  float eps, momentum, r_feat, r_lstm, slope;
  uint64_t seed; uint32_t site_feat, site_lstm; const uint32_t* step_dev;
  // forward from split-K partials (SK): y = leaky(sum_s part[s] + bias), pre-activation kept for the backward
  const float* part; const float* bias; float* pre_out; int nsplit;
  // backward: an independent in-place dropout' job riding in the same launch (blocks >= nb_tail; rate 0: none) -- the text
  // call's LSTM-input dropout over the other rows of the same gradient buffer (NIC.py:131,140)
  DropArgs drop; int nb_tail;
  // forward (SK): the Embedding gather + the text call's LSTM-input dropout riding in the same launch (blocks >= nb_tail;
  // emb_table NULL: none): out rows t*B + b of emb_out (row stride ldo), mask element (b*T + t)*E + j of stream emb_site
  const float* emb_table; const int* emb_ids; float* emb_out; int emb_B, emb_T, emb_V; float emb_rate; uint32_t emb_site;
};

// column sums of a per-thread float4 over the 32 row groups, fixed order; result valid in every thread
__device__ __forceinline__ float4 et_colsum4(float4 v, float4 (*red)[ET_CW / 4], int rg, int c4) {
  __syncthreads();
  red[rg][c4] = v;
  __syncthreads();
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int k = 0; k < ET_RG; ++k) {
    const float4 r = red[k][c4];
    t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
  }
  return t;
}

__device__ __forceinline__ float4 et_drop4(float4 v, uint64_t e, float rate, float scale, uint64_t seed, uint32_t site,
                                           uint32_t step) {
  bool k[4];
  tnt_keep4(e, rate, seed, site, step, k);
  return make_float4(k[0] ? v.x * scale : 0.f, k[1] ? v.y * scale : 0.f, k[2] ? v.z * scale : 0.f,
                     k[3] ? v.w * scale : 0.f);
}

template <bool SK>
__global__ __launch_bounds__(256) void enc_tail_fwd_kernel(EncTailArgs a) {
  if (SK && a.emb_table != nullptr && (int)blockIdx.x >= a.nb_tail) {      // the riding Embedding job (as emb_fwd_drop_kernel)
    const int row = (blockIdx.x - a.nb_tail) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // row = t*B + b
    if (row >= a.emb_B * a.emb_T) return;
    const uint32_t estep = a.step_dev ? a.step_dev[0] : 0u;
    const int t = row / a.emb_B, b = row % a.emb_B, E = a.C;
    int id = a.emb_ids[b * a.emb_T + t];
    id = id < 0 ? 0 : (id >= a.emb_V ? a.emb_V - 1 : id);
    const float* src = a.emb_table + (long)id * E;
    const float scale = 1.0f / (1.0f - a.emb_rate);
    for (int j = lane * 4; j < E; j += 256) {
      float4 v = *reinterpret_cast<const float4*>(src + j);
      if (a.emb_rate > 0.f) {
        bool k[4];
        tnt_keep4((uint64_t)(b * a.emb_T + t) * E + j, a.emb_rate, a.seed, a.emb_site, estep, k);
        v = make_float4(k[0] ? v.x * scale : 0.f, k[1] ? v.y * scale : 0.f, k[2] ? v.z * scale : 0.f, k[3] ? v.w * scale : 0.f);
      }
      *reinterpret_cast<float4*>(a.emb_out + (long)row * a.ldo + j) = v;
    }
    return;
  }
  __shared__ float4 red[ET_RG][ET_CW / 4];
  const int c4 = threadIdx.x % (ET_CW / 4), rg = threadIdx.x / (ET_CW / 4);
  const int col = blockIdx.x * ET_CW + 4 * c4;
  const bool cok = col < a.C;
  const uint32_t step = a.step_dev ? a.step_dev[0] : 0u;
  const float sc_f = 1.0f / (1.0f - a.r_feat), sc_l = 1.0f / (1.0f - a.r_lstm);
  float4 v[ET_MAXR];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < ET_MAXR; ++k) {
    const int r = rg + ET_RG * k;
    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok && r < a.rows) {
      float4 x;
      if (SK) {       // the dense layer's K-split partials, summed in split order, + bias, LeakyReLU (NIC.py:125)
        x = *reinterpret_cast<const float4*>(a.bias + col);
        const float* pp = a.part + (long)r * a.C + col;
#pragma unroll 8
        for (int sp = 0; sp < a.nsplit; ++sp) {
          const float4 u = *reinterpret_cast<const float4*>(pp + (long)sp * a.rows * a.C);
          x.x += u.x; x.y += u.y; x.z += u.z; x.w += u.w;
        }
        *reinterpret_cast<float4*>(a.pre_out + (long)r * a.C + col) = x;
        x.x = x.x > 0.f ? x.x : x.x * a.slope; x.y = x.y > 0.f ? x.y : x.y * a.slope;
        x.z = x.z > 0.f ? x.z : x.z * a.slope; x.w = x.w > 0.f ? x.w : x.w * a.slope;
      } else {
        x = *reinterpret_cast<const float4*>(a.y + (long)r * a.C + col);
      }
      if (a.training && a.r_feat > 0.f) x = et_drop4(x, (uint64_t)r * a.C + col, a.r_feat, sc_f, a.seed, a.site_feat, step);
      v[k] = x;
      s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
    }
  }
  float4 mean, inv;
  if (a.training) {
    const float n = (float)a.rows;
    mean = et_colsum4(s, red, rg, c4);
    mean.x /= n; mean.y /= n; mean.z /= n; mean.w /= n;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < ET_MAXR; ++k) {
      const int r = rg + ET_RG * k;
      if (cok && r < a.rows) {
        const float dx = v[k].x - mean.x, dy = v[k].y - mean.y, dz = v[k].z - mean.z, dw = v[k].w - mean.w;
        q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
      }
    }
    float4 var = et_colsum4(q, red, rg, c4);                                 // biased, keras
    var.x /= n; var.y /= n; var.z /= n; var.w /= n;
    inv = make_float4(1.f / sqrtf(var.x + a.eps), 1.f / sqrtf(var.y + a.eps), 1.f / sqrtf(var.z + a.eps),
                      1.f / sqrtf(var.w + a.eps));
    if (rg == 0 && cok) {
      const float mo = a.momentum, om = 1.f - a.momentum;
      float4 mm = *reinterpret_cast<float4*>(a.mov_mean + col), mv = *reinterpret_cast<float4*>(a.mov_var + col);
      mm.x = mm.x * mo + mean.x * om; mm.y = mm.y * mo + mean.y * om; mm.z = mm.z * mo + mean.z * om; mm.w = mm.w * mo + mean.w * om;
      mv.x = mv.x * mo + var.x * om; mv.y = mv.y * mo + var.y * om; mv.z = mv.z * mo + var.z * om; mv.w = mv.w * mo + var.w * om;
      *reinterpret_cast<float4*>(a.mov_mean + col) = mm;
      *reinterpret_cast<float4*>(a.mov_var + col) = mv;
    }
  } else {
    mean = make_float4(0.f, 0.f, 0.f, 0.f); inv = mean;
    if (cok) {
      mean = *reinterpret_cast<const float4*>(a.mov_mean + col);
      const float4 mv = *reinterpret_cast<const float4*>(a.mov_var + col);
      inv = make_float4(1.f / sqrtf(mv.x + a.eps), 1.f / sqrtf(mv.y + a.eps), 1.f / sqrtf(mv.z + a.eps),
                        1.f / sqrtf(mv.w + a.eps));
    }
  }
  if (!cok) return;
  if (rg == 0) *reinterpret_cast<float4*>(a.inv_std + col) = inv;
  const float4 g = *reinterpret_cast<const float4*>(a.gamma + col), b = *reinterpret_cast<const float4*>(a.beta + col);
#pragma unroll
  for (int k = 0; k < ET_MAXR; ++k) {
    const int r = rg + ET_RG * k;
    if (r >= a.rows) continue;
    const float4 xh = make_float4((v[k].x - mean.x) * inv.x, (v[k].y - mean.y) * inv.y, (v[k].z - mean.z) * inv.z,
                                  (v[k].w - mean.w) * inv.w);
    *reinterpret_cast<float4*>(a.xhat + (long)r * a.C + col) = xh;
    float4 o = make_float4(xh.x * g.x + b.x, xh.y * g.y + b.y, xh.z * g.z + b.z, xh.w * g.w + b.w);
    if (a.training && a.r_lstm > 0.f) o = et_drop4(o, (uint64_t)r * a.C + col, a.r_lstm, sc_l, a.seed, a.site_lstm, step);
    *reinterpret_cast<float4*>(a.out + (long)r * a.ldo + col) = o;
  }
}

__global__ __launch_bounds__(256) void enc_tail_bwd_kernel(EncTailArgs a) {
  if (a.drop.rate > 0.f && (int)blockIdx.x >= a.nb_tail) {          // the riding dropout' job (as dropout4_kernel)
    const DropArgs& d = a.drop;
    const uint32_t dstep = d.step + (d.step_dev ? d.step_dev[0] : 0u);
    const int c4n = d.cols >> 2, nbk = gridDim.x - a.nb_tail;
    const long total = (long)d.rows * c4n;
    const int T = d.tB > 0 ? d.rows / d.tB : 0;
    for (long e = (long)(blockIdx.x - a.nb_tail) * 256 + threadIdx.x; e < total; e += (long)nbk * 256) {
      const int r = (int)(e / c4n), c = (int)(e % c4n) * 4;
      long lrow; uint32_t site;
      drop_row(d, r, T, lrow, site);
      bool k[4];
      tnt_keep4((uint64_t)lrow * (uint64_t)d.lwidth + (uint64_t)(d.lcol0 + c), d.rate, d.seed, site, dstep, k);
      const long o = (long)r * d.ld + c;
      const float4 v = *reinterpret_cast<const float4*>(d.x + o);
      *reinterpret_cast<float4*>(d.y + o) = make_float4(k[0] ? v.x * d.scale : 0.f, k[1] ? v.y * d.scale : 0.f,
                                                        k[2] ? v.z * d.scale : 0.f, k[3] ? v.w * d.scale : 0.f);
    }
    return;
  }
  __shared__ float4 red[ET_RG][ET_CW / 4];
  const int c4 = threadIdx.x % (ET_CW / 4), rg = threadIdx.x / (ET_CW / 4);
  const int col = blockIdx.x * ET_CW + 4 * c4;
  const bool cok = col < a.C;
  const uint32_t step = a.step_dev ? a.step_dev[0] : 0u;
  const float sc_f = 1.0f / (1.0f - a.r_feat), sc_l = 1.0f / (1.0f - a.r_lstm);
  float4 dy[ET_MAXR], xh[ET_MAXR];
  float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
#pragma unroll
  for (int k = 0; k < ET_MAXR; ++k) {
    const int r = rg + ET_RG * k;
    dy[k] = make_float4(0.f, 0.f, 0.f, 0.f); xh[k] = dy[k];
    if (cok && r < a.rows) {
      float4 g = *reinterpret_cast<const float4*>(a.dout + (long)r * a.ldo + col);
      if (a.r_lstm > 0.f) g = et_drop4(g, (uint64_t)r * a.C + col, a.r_lstm, sc_l, a.seed, a.site_lstm, step);
      dy[k] = g;
      xh[k] = *reinterpret_cast<const float4*>(a.xhat + (long)r * a.C + col);
      sb.x += g.x; sb.y += g.y; sb.z += g.z; sb.w += g.w;
      sg.x += g.x * xh[k].x; sg.y += g.y * xh[k].y; sg.z += g.z * xh[k].z; sg.w += g.w * xh[k].w;
    }
  }
  const float4 dbeta = et_colsum4(sb, red, rg, c4);
  const float4 dgamma = et_colsum4(sg, red, rg, c4);
  const float n = (float)a.rows;
  float4 kk = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cok) {
    const float4 g = *reinterpret_cast<const float4*>(a.gamma + col), iv = *reinterpret_cast<const float4*>(a.inv_std + col);
    kk = make_float4(g.x * iv.x / n, g.y * iv.y / n, g.z * iv.z / n, g.w * iv.w / n);
  }
  float4 sbias = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < ET_MAXR; ++k) {
    const int r = rg + ET_RG * k;
    if (cok && r < a.rows) {
      float4 dx = make_float4(kk.x * (n * dy[k].x - dbeta.x - xh[k].x * dgamma.x), kk.y * (n * dy[k].y - dbeta.y - xh[k].y * dgamma.y),
                              kk.z * (n * dy[k].z - dbeta.z - xh[k].z * dgamma.z), kk.w * (n * dy[k].w - dbeta.w - xh[k].w * dgamma.w));
      if (a.r_feat > 0.f) dx = et_drop4(dx, (uint64_t)r * a.C + col, a.r_feat, sc_f, a.seed, a.site_feat, step);
      const float4 p = *reinterpret_cast<const float4*>(a.pre + (long)r * a.C + col);
      dx.x = p.x > 0.f ? dx.x : dx.x * a.slope; dx.y = p.y > 0.f ? dx.y : dx.y * a.slope;     // LeakyReLU'
      dx.z = p.z > 0.f ? dx.z : dx.z * a.slope; dx.w = p.w > 0.f ? dx.w : dx.w * a.slope;
      *reinterpret_cast<float4*>(a.dpre + (long)r * a.C + col) = dx;
      sbias.x += dx.x; sbias.y += dx.y; sbias.z += dx.z; sbias.w += dx.w;
    }
  }
  const float4 dbias = et_colsum4(sbias, red, rg, c4);
  if (cok && rg == 0) {
    *reinterpret_cast<float4*>(a.dgamma + col) = dgamma;
    *reinterpret_cast<float4*>(a.dbeta + col) = dbeta;
    *reinterpret_cast<float4*>(a.dbias + col) = dbias;
  }
}

}  // namespace

extern "C" int32_t tnt_enc_tail_fwd_f32(const float* y, const float* gamma, const float* beta, float* mov_mean,
                                        float* mov_var, float* out, float* xhat, float* inv_std, int32_t rows,
                                        int32_t C, int32_t ldo, int32_t training, float eps, float momentum,
                                        float r_feat, float r_lstm, uint64_t seed, uint32_t site_feat,
                                        uint32_t site_lstm, const uint32_t* step_dev, void* stream) {
  if (rows <= 0 || rows > ET_RG * ET_MAXR) return TNT_BADARG(9);
  if (C <= 0 || C % 4 != 0 || ldo < C || ldo % 4 != 0) return TNT_BADARG(10);
  if (!tnt_aligned16(y) || !tnt_aligned16(out) || !tnt_aligned16(xhat) || !tnt_aligned16(gamma) || !tnt_aligned16(beta) ||
      !tnt_aligned16(mov_mean) || !tnt_aligned16(mov_var) || !tnt_aligned16(inv_std)) return TNT_BADARG(1);
  EncTailArgs a{};
  a.y = y; a.gamma = gamma; a.beta = beta; a.mov_mean = mov_mean; a.mov_var = mov_var; a.out = out; a.xhat = xhat;
  a.inv_std = inv_std; a.rows = rows; a.C = C; a.ldo = ldo; a.training = training; a.eps = eps; a.momentum = momentum;
  a.r_feat = r_feat; a.r_lstm = r_lstm; a.seed = seed; a.site_feat = site_feat; a.site_lstm = site_lstm;
  a.step_dev = step_dev;
  hipLaunchKernelGGL(enc_tail_fwd_kernel<false>, dim3((C + ET_CW - 1) / ET_CW), dim3(256), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_enc_tail_fwd_sk_emb_f32(const float* part, int32_t nsplit, const float* bias, float* pre, float slope,
                                               const float* gamma, const float* beta, float* mov_mean, float* mov_var,
                                               float* out, float* xhat, float* inv_std, int32_t rows, int32_t C, int32_t ldo,
                                               int32_t training, float eps, float momentum, float r_feat, float r_lstm,
                                               uint64_t seed, uint32_t site_feat, uint32_t site_lstm,
                                               const uint32_t* step_dev, const float* emb_table, const int32_t* emb_ids,
                                               float* emb_out, int32_t emb_B, int32_t emb_T, int32_t emb_V, float emb_rate,
                                               uint32_t emb_site, void* stream) {
  if (rows <= 0 || rows > ET_RG * ET_MAXR || nsplit <= 0) return TNT_BADARG(13);
  if (C <= 0 || C % 4 != 0 || ldo < C || ldo % 4 != 0) return TNT_BADARG(14);
  if (!tnt_aligned16(part) || !tnt_aligned16(bias) || !tnt_aligned16(pre) || !tnt_aligned16(out) || !tnt_aligned16(xhat) ||
      !tnt_aligned16(gamma) || !tnt_aligned16(beta) || !tnt_aligned16(mov_mean) || !tnt_aligned16(mov_var) ||
      !tnt_aligned16(inv_std)) return TNT_BADARG(1);
  EncTailArgs a{};
  a.part = part; a.nsplit = nsplit; a.bias = bias; a.pre_out = pre; a.slope = slope;
  a.gamma = gamma; a.beta = beta; a.mov_mean = mov_mean; a.mov_var = mov_var; a.out = out; a.xhat = xhat;
  a.inv_std = inv_std; a.rows = rows; a.C = C; a.ldo = ldo; a.training = training; a.eps = eps; a.momentum = momentum;
  a.r_feat = r_feat; a.r_lstm = r_lstm; a.seed = seed; a.site_feat = site_feat; a.site_lstm = site_lstm;
  a.step_dev = step_dev;
  a.nb_tail = (C + ET_CW - 1) / ET_CW;
  int nbe = 0;
  if (emb_table != nullptr) {       // Embedding width = C (both feed the same LSTM input), rows behind the feature rows
    if (emb_ids == nullptr || emb_out == nullptr || emb_B <= 0 || emb_T <= 0 || emb_V <= 0 || emb_rate < 0.f || emb_rate >= 1.f ||
        !tnt_aligned16(emb_table) || !tnt_aligned16(emb_out)) return TNT_BADARG(25);
    a.emb_table = emb_table; a.emb_ids = emb_ids; a.emb_out = emb_out; a.emb_B = emb_B; a.emb_T = emb_T; a.emb_V = emb_V;
    a.emb_rate = emb_rate; a.emb_site = emb_site;
    nbe = (emb_B * emb_T + 3) / 4;
  }
  hipLaunchKernelGGL(enc_tail_fwd_kernel<true>, dim3(a.nb_tail + nbe), dim3(256), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_enc_tail_fwd_sk_f32(const float* part, int32_t nsplit, const float* bias, float* pre, float slope,
                                           const float* gamma, const float* beta, float* mov_mean, float* mov_var,
                                           float* out, float* xhat, float* inv_std, int32_t rows, int32_t C, int32_t ldo,
                                           int32_t training, float eps, float momentum, float r_feat, float r_lstm,
                                           uint64_t seed, uint32_t site_feat, uint32_t site_lstm,
                                           const uint32_t* step_dev, void* stream) {
  return tnt_enc_tail_fwd_sk_emb_f32(part, nsplit, bias, pre, slope, gamma, beta, mov_mean, mov_var, out, xhat, inv_std, rows, C,
                                     ldo, training, eps, momentum, r_feat, r_lstm, seed, site_feat, site_lstm, step_dev,
                                     nullptr, nullptr, nullptr, 0, 0, 0, 0.f, 0u, stream);
}

extern "C" int32_t tnt_enc_tail_bwd_f32(const float* dout, const float* xhat, const float* gamma, const float* inv_std,
                                        const float* pre, float* dpre, float* dgamma, float* dbeta, float* dbias,
                                        int32_t rows, int32_t C, int32_t ldo, float r_feat, float r_lstm, float slope,
                                        uint64_t seed, uint32_t site_feat, uint32_t site_lstm, const uint32_t* step_dev,
                                        void* stream) {
  if (rows <= 0 || rows > ET_RG * ET_MAXR) return TNT_BADARG(10);
  if (C <= 0 || C % 4 != 0 || ldo < C || ldo % 4 != 0) return TNT_BADARG(11);
  if (!tnt_aligned16(dout) || !tnt_aligned16(xhat) || !tnt_aligned16(gamma) || !tnt_aligned16(inv_std) || !tnt_aligned16(pre) ||
      !tnt_aligned16(dpre) || !tnt_aligned16(dgamma) || !tnt_aligned16(dbeta) || !tnt_aligned16(dbias)) return TNT_BADARG(1);
  EncTailArgs a{};
  a.dout = dout; a.xhat = const_cast<float*>(xhat); a.gamma = gamma; a.inv_std = const_cast<float*>(inv_std); a.pre = pre; a.dpre = dpre; a.dgamma = dgamma;
  a.dbeta = dbeta; a.dbias = dbias; a.rows = rows; a.C = C; a.ldo = ldo; a.r_feat = r_feat; a.r_lstm = r_lstm;
  a.slope = slope; a.seed = seed; a.site_feat = site_feat; a.site_lstm = site_lstm; a.step_dev = step_dev;
  a.nb_tail = (C + ET_CW - 1) / ET_CW;
  hipLaunchKernelGGL(enc_tail_bwd_kernel, dim3(a.nb_tail), dim3(256), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_enc_tail_bwd_drop_f32(const float* dout, const float* xhat, const float* gamma, const float* inv_std,
                                             const float* pre, float* dpre, float* dgamma, float* dbeta, float* dbias,
                                             int32_t rows, int32_t C, int32_t ldo, float r_feat, float r_lstm, float slope,
                                             uint64_t seed, uint32_t site_feat, uint32_t site_lstm, const uint32_t* step_dev,
                                             float* drop_x, int32_t drop_rows, int32_t drop_cols, int32_t drop_ld,
                                             int32_t drop_tmajor_B, int32_t drop_lwidth, int32_t drop_lcol0, float drop_rate,
                                             uint32_t drop_site, void* stream) {
  if (rows <= 0 || rows > ET_RG * ET_MAXR) return TNT_BADARG(10);
  if (C <= 0 || C % 4 != 0 || ldo < C || ldo % 4 != 0) return TNT_BADARG(11);
  if (!tnt_aligned16(dout) || !tnt_aligned16(xhat) || !tnt_aligned16(gamma) || !tnt_aligned16(inv_std) || !tnt_aligned16(pre) ||
      !tnt_aligned16(dpre) || !tnt_aligned16(dgamma) || !tnt_aligned16(dbeta) || !tnt_aligned16(dbias)) return TNT_BADARG(1);
  if (drop_rate < 0.f || drop_rate >= 1.f || drop_rows <= 0 || ((drop_cols | drop_ld | drop_lwidth | drop_lcol0) & 3) != 0 ||
      !tnt_aligned16(drop_x) || (drop_tmajor_B > 0 && drop_rows % drop_tmajor_B != 0)) return TNT_BADARG(20);
  EncTailArgs a{};
  a.dout = dout; a.xhat = const_cast<float*>(xhat); a.gamma = gamma; a.inv_std = const_cast<float*>(inv_std); a.pre = pre; a.dpre = dpre; a.dgamma = dgamma;
  a.dbeta = dbeta; a.dbias = dbias; a.rows = rows; a.C = C; a.ldo = ldo; a.r_feat = r_feat; a.r_lstm = r_lstm;
  a.slope = slope; a.seed = seed; a.site_feat = site_feat; a.site_lstm = site_lstm; a.step_dev = step_dev;
  a.nb_tail = (C + ET_CW - 1) / ET_CW;
  DropArgs& d = a.drop;
  d.x = drop_x; d.y = drop_x; d.rows = drop_rows; d.cols = drop_cols; d.ld = drop_ld; d.tB = drop_tmajor_B; d.lwidth = drop_lwidth;
  d.lcol0 = drop_lcol0; d.rows_per_site = 0; d.rate = drop_rate; d.scale = 1.0f / (1.0f - drop_rate); d.seed = seed;
  d.site = drop_site; d.step = 0; d.step_dev = step_dev;
  const int nbd = drop_rate > 0.f ? ew_blocks((long)drop_rows * (drop_cols / 4)) : 0;
  hipLaunchKernelGGL(enc_tail_bwd_kernel, dim3(a.nb_tail + nbd), dim3(256), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

// work layout for BN: [mean C][partials 2*C*nchunk]
namespace {
void launch_bn_apply(const float* x, const float* mean, const float* inv_std, const float* gamma, const float* beta, float* y,
                     float* xhat, int rows, int C, int ldy, float rate, uint64_t seed, uint32_t site,
                     const uint32_t* step_dev, hipStream_t s) {
  const bool vec = ((C | ldy) & 3) == 0 && tnt_aligned16(x) && tnt_aligned16(y) && tnt_aligned16(xhat) && tnt_aligned16(mean) &&
                   tnt_aligned16(inv_std) && tnt_aligned16(gamma) && tnt_aligned16(beta);
  if (vec)
    hipLaunchKernelGGL(bn_apply4_kernel, dim3(ew_blocks((long)rows * (C / 4))), dim3(256), 0, s, x, mean, inv_std, gamma, beta,
                       y, xhat, rows, C, ldy, rate, seed, site, step_dev);
  else
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks((long)rows * C)), dim3(256), 0, s, x, mean, inv_std, gamma, beta, y,
                       xhat, rows, C, ldy, rate, seed, site, step_dev);
}
}  // namespace

extern "C" int32_t tnt_batchnorm_fwd_drop_f32(const float* x, const float* gamma, const float* beta, float* mov_mean,
                                              float* mov_var, float* y, float* xhat, float* inv_std, int32_t rows,
                                              int32_t C, int32_t ldy, int32_t training, float eps, float momentum,
                                              float* work, float rate, uint64_t seed, uint32_t site,
                                              const uint32_t* step_dev, void* stream) {
  if (rate < 0.f || rate >= 1.f) return TNT_BADARG(16);
  hipStream_t s = tnt_stream(stream);
  const int nchunk = chunk_count(rows);
  float* mean = work;
  float* part = work + C;
  if (training) {
    launch_col_partial<0>(x, C, nullptr, 0, rows, C, part, s);
    TNT_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, s, part, rows, C, nchunk, chunk_rows(rows),
                       eps, momentum, mov_mean, mov_var, mean, inv_std, 1);
    TNT_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(bn_infer_prep_kernel, dim3((C + 63) / 64), dim3(64), 0, s, mov_mean, mov_var, C, eps, mean,
                       inv_std);
    TNT_LAUNCH_CHECK();
  }
  launch_bn_apply(x, mean, inv_std, gamma, beta, y, xhat, rows, C, ldy, rate, seed, site, step_dev, s);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_batchnorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* mov_mean,
                                         float* mov_var, float* y, float* xhat, float* inv_std, int32_t rows, int32_t C,
                                         int32_t ldy, int32_t training, float eps, float momentum, float* work,
                                         void* stream) {
  return tnt_batchnorm_fwd_drop_f32(x, gamma, beta, mov_mean, mov_var, y, xhat, inv_std, rows, C, ldy, training, eps, momentum,
                                    work, 0.f, 0, 0, nullptr, stream);
}

extern "C" int32_t tnt_batchnorm_bwd_act_f32(const float* dy, const float* xhat, const float* gamma, const float* inv_std,
                                             float* dx, float* dgamma, float* dbeta, int32_t rows, int32_t C, int32_t lddy,
                                             int32_t training, float* work, const float* act_pre, float slope,
                                             void* stream) {
  hipStream_t s = tnt_stream(stream);
  if (dgamma != nullptr || dbeta != nullptr) {        // both null: input gradient only (per-step use inside a T-step chain)
    if (dgamma == nullptr || dbeta == nullptr || work == nullptr) return TNT_BADARG(6);
    const int nchunk = chunk_count(rows);
    float* part = work + C;
    launch_col_partial<1>(dy, lddy, xhat, C, rows, C, part, s);
    TNT_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_finalize_kernel<true>, dim3(C), dim3(64), 0, s, part, C, nchunk, dbeta, dgamma);
    TNT_LAUNCH_CHECK();
  }
  if (dx) {
    hipLaunchKernelGGL(bn_dx_kernel, dim3(ew_blocks((long)rows * C)), dim3(256), 0, s, dy, lddy, xhat, gamma, inv_std,
                       dgamma, dbeta, dx, rows, C, training, rows, act_pre, slope);
    TNT_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int32_t tnt_batchnorm_bwd_f32(const float* dy, const float* xhat, const float* gamma, const float* inv_std,
                                         float* dx, float* dgamma, float* dbeta, int32_t rows, int32_t C, int32_t lddy,
                                         int32_t training, float* work, void* stream) {
  return tnt_batchnorm_bwd_act_f32(dy, xhat, gamma, inv_std, dx, dgamma, dbeta, rows, C, lddy, training, work, nullptr, 0.f,
                                   stream);
}

// ---- synchronised BatchNorm (data parallel, opt-in): the three pieces of tnt_batchnorm_{fwd,bwd}_f32 around the caller's
// collectives.  stats: this replica's chunk partials (tnt_bn_nchunk(rows) * 2 * C floats); apply_stats: statistics of the
// nrep replicas' all-gathered partials, moving statistics, then y / xhat; dx: the input gradient from sums taken over all
// n_total rows (the caller all-reduces the local dgamma / dbeta sums of tnt_batchnorm_bwd_f32(dx = NULL)).
extern "C" int32_t tnt_batchnorm_stats_f32(const float* x, int32_t rows, int32_t C, float* part, void* stream) {
  if (rows <= 0 || C <= 0) return TNT_BADARG(1);
  launch_col_partial<0>(x, C, nullptr, 0, rows, C, part, tnt_stream(stream));
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_batchnorm_apply_stats_f32(const float* part_all, int32_t nrep, const float* x, const float* gamma,
                                                 const float* beta, float* mov_mean, float* mov_var, float* y, float* xhat,
                                                 float* inv_std, int32_t rows, int32_t C, int32_t ldy, float eps,
                                                 float momentum, float* mean_work, void* stream) {
  if (rows <= 0 || C <= 0 || nrep <= 0) return TNT_BADARG(10);
  hipStream_t s = tnt_stream(stream);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, s, part_all, rows, C, chunk_count(rows), chunk_rows(rows), eps,
                     momentum, mov_mean, mov_var, mean_work, inv_std, nrep);
  TNT_LAUNCH_CHECK();
  launch_bn_apply(x, mean_work, inv_std, gamma, beta, y, xhat, rows, C, ldy, 0.f, 0, 0, nullptr, s);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_batchnorm_dx_f32(const float* dy, int32_t lddy, const float* xhat, const float* gamma,
                                        const float* inv_std, const float* dgamma_sum, const float* dbeta_sum, float* dx,
                                        int32_t rows, int32_t C, int32_t n_total, void* stream) {
  if (rows <= 0 || C <= 0 || n_total < rows) return TNT_BADARG(8);
  hipLaunchKernelGGL(bn_dx_kernel, dim3(ew_blocks((long)rows * C)), dim3(256), 0, tnt_stream(stream), dy, lddy, xhat, gamma,
                     inv_std, dgamma_sum, dbeta_sum, dx, rows, C, 1, n_total, nullptr, 0.f);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* xhat,
                                         float* inv_std, int32_t rows, int32_t C, int32_t ldy, float eps, void* stream) {
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, tnt_stream(stream), x, gamma, beta, y, xhat,
                     inv_std, rows, C, ldy, eps);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_layernorm_bwd_f32(const float* dy, const float* xhat, const float* gamma, const float* inv_std,
                                         float* dx, float* dgamma, float* dbeta, int32_t rows, int32_t C, int32_t lddy,
                                         float* work, void* stream) {
  hipStream_t s = tnt_stream(stream);
  if (dgamma != nullptr || dbeta != nullptr) {        // both null: input gradient only (per-step use inside a T-step chain)
    if (dgamma == nullptr || dbeta == nullptr || work == nullptr) return TNT_BADARG(6);
    const int nchunk = chunk_count(rows);
    float* part = work + C;
    launch_col_partial<1>(dy, lddy, xhat, C, rows, C, part, s);
    TNT_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_finalize_kernel<true>, dim3(C), dim3(64), 0, s, part, C, nchunk, dbeta, dgamma);
    TNT_LAUNCH_CHECK();
  }
  if (dx) {
    hipLaunchKernelGGL(ln_dx_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, dy, lddy, xhat, gamma, inv_std, dx, rows, C);
    TNT_LAUNCH_CHECK();
  }
  return 0;
}

// Column sums of a short matrix (rows <= 2048: every bias gradient of a T*B-row activation) in ONE launch instead of
// partial + finalize: 32 column lanes x 32 row lanes per workgroup, each thread's loads independent (8 in flight),
// then a fixed-order sum over the row lanes through LDS.
__global__ __launch_bounds__(1024) void col_sum_direct_kernel(const float* x, int ld, int rows, int C, float* out) {
  __shared__ float sh[32][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float v = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int r = rl; r < rows; r += 32) v += x[(long)r * ld + c];
  }
  sh[rl][cl] = v;
  __syncthreads();
  if (rl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) t += sh[j][cl];
    out[c] = t;
  }
}

// up to four independent short matrices in one launch (bias gradients that become final at the same point of a step):
// job j owns the blocks [first[j], first[j+1]) -- one dependent launch instead of up to four in the captured step
struct ColJobs { const float* x[4]; float* out[4]; int ld[4], rows[4], C[4], first[5]; };

__global__ __launch_bounds__(1024) void col_sum_multi_kernel(ColJobs j) {
  __shared__ float sh[32][33];
  int q = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) q += (int)blockIdx.x >= j.first[k] ? 1 : 0;
  const float* x = j.x[0]; float* out = j.out[0]; int ld = j.ld[0], rows = j.rows[0], C = j.C[0], b0 = j.first[0];
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (q == k) { x = j.x[k]; out = j.out[k]; ld = j.ld[k]; rows = j.rows[k]; C = j.C[k]; b0 = j.first[k]; }
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (blockIdx.x - b0) * 32 + cl;
  float v = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int r = rl; r < rows; r += 32) v += x[(long)r * ld + c];
  }
  sh[rl][cl] = v;
  __syncthreads();
  if (rl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += sh[k][cl];
    out[c] = t;
  }
}

extern "C" int32_t tnt_colsum4_f32(const float* x0, float* out0, int32_t rows0, int32_t C0, int32_t ld0, const float* x1,
                                   float* out1, int32_t rows1, int32_t C1, int32_t ld1, const float* x2, float* out2,
                                   int32_t rows2, int32_t C2, int32_t ld2, const float* x3, float* out3, int32_t rows3,
                                   int32_t C3, int32_t ld3, void* stream) {
  ColJobs j{};
  const float* xs[4] = {x0, x1, x2, x3}; float* outs[4] = {out0, out1, out2, out3};
  const int rs[4] = {rows0, rows1, rows2, rows3}, cs[4] = {C0, C1, C2, C3}, lds[4] = {ld0, ld1, ld2, ld3};
  int nb = 0;
  for (int k = 0; k < 4; ++k) {          // unused jobs (x == NULL) get an empty block range
    j.first[k] = nb;
    j.x[k] = xs[k]; j.out[k] = outs[k]; j.ld[k] = lds[k]; j.rows[k] = rs[k]; j.C[k] = cs[k];
    if (xs[k] == nullptr) continue;
    if (rs[k] <= 0 || cs[k] <= 0 || rs[k] > 2048 || outs[k] == nullptr) return TNT_BADARG(2 + 5 * k);
    nb += (cs[k] + 31) / 32;
  }
  j.first[4] = nb;
  if (nb == 0) return 0;
  // an empty job must not capture blocks: give it the start of the next job (q counts first[k] <= block)
  for (int k = 3; k >= 0; --k) if (xs[k] == nullptr) j.first[k] = j.first[k + 1];
  hipLaunchKernelGGL(col_sum_multi_kernel, dim3(nb), dim3(1024), 0, tnt_stream(stream), j);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_colsum2_f32(const float* x0, float* out0, int32_t rows0, int32_t C0, int32_t ld0, const float* x1,
                                   float* out1, int32_t rows1, int32_t C1, int32_t ld1, void* stream) {
  if (x0 == nullptr || x1 == nullptr) return TNT_BADARG(0);
  return tnt_colsum4_f32(x0, out0, rows0, C0, ld0, x1, out1, rows1, C1, ld1, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, 0, 0,
                         0, stream);
}

// ---- backward of  y = Dropout(act(pre)),  pre = x W + b  up to the bias gradient, for a layer applied to T*B <= 2048 rows
// (TimeDistributed(Dense) + Dropout of the caption head, lc_NIC.py:271-275): dx = dropout'(dy) * act'(pre) in place of a
// dropout launch + an act_bwd launch, and db = column sums of dx in the same pass instead of a third launch.  A workgroup
// owns 32 columns (8 column quads x 128 row lanes: float4 loads, one Philox call per 4 elements).  A second, independent
// column-sum job (x1 -> out1: another layer's bias gradient that is final at the same point) rides in the same launch.
struct BadArgs {
  DropArgs d;              // x = dy, y = dx, rows, cols, ld, logical layout + stream of the forward's dropout (rate 0: none)
  const float* pre; float* dbias; int act; float slope; int nb0;
  const float* x1; float* out1; int rows1, C1, ld1;
};

// NARROW: a workgroup owns ONE column quad and all rows (one row, i.e. one Philox call, per thread) instead of 8 quads with
// 7-8 rows per thread: for the few-hundred-column activations of the vocabulary head the wide form put the whole pass on
// cols / 32 = 8 CUs, each thread walking its rows one Philox call after the other (12.7 us for 960 x 256 elements).
template <bool NARROW>
__global__ __launch_bounds__(1024) void bias_act_drop_bwd_kernel(BadArgs a) {
  __shared__ float4 red[128][9];
  __shared__ float sh[32][33];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.nb0) {          // plain column sums of the second job: 32 columns x 32 row lanes
    const int cl = tid & 31, rl = tid >> 5, c = (blockIdx.x - a.nb0) * 32 + cl;
    float v = 0.f;
    if (c < a.C1) {
      // (all of a lane's loads of a round in flight together: the rows come from HBM / the MALL, and four dependent rounds of
      // eight were four memory latencies, 8 of this launch's 12.7 us)
      for (int r = rl; r < a.rows1; r += 32 * 32) {
        float t[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) t[k] = r + 32 * k < a.rows1 ? a.x1[(long)(r + 32 * k) * a.ld1 + c] : 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) v += t[k];
      }
    }
    sh[rl][cl] = v;
    __syncthreads();
    if (rl == 0 && c < a.C1) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) t += sh[k][cl];
      a.out1[c] = t;
    }
    return;
  }
  const DropArgs& d = a.d;
  const uint32_t step = d.step + (d.step_dev ? d.step_dev[0] : 0u);
  const int cq = NARROW ? 0 : tid & 7, rl = NARROW ? tid : tid >> 3;         // wide: 8 column quads x 128 row lanes
  const int c = NARROW ? blockIdx.x * 4 : blockIdx.x * 32 + cq * 4;
  const int T = d.tB > 0 ? d.rows / d.tB : 0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < d.cols) {
    for (int r = rl; r < d.rows; r += NARROW ? 1024 : 128) {
      const long o = (long)r * d.ld + c;
      float4 v = *reinterpret_cast<const float4*>(d.x + o);
      if (d.rate > 0.f) {
        long lrow; uint32_t site;
        drop_row(d, r, T, lrow, site);
        bool k[4];
        tnt_keep4((uint64_t)lrow * (uint64_t)d.lwidth + (uint64_t)(d.lcol0 + c), d.rate, d.seed, site, step, k);
        v.x = k[0] ? v.x * d.scale : 0.f; v.y = k[1] ? v.y * d.scale : 0.f;
        v.z = k[2] ? v.z * d.scale : 0.f; v.w = k[3] ? v.w * d.scale : 0.f;
      }
      const float4 p = *reinterpret_cast<const float4*>(a.pre + o);
      v = make_float4(tnt_act_grad(p.x, v.x, a.act, a.slope), tnt_act_grad(p.y, v.y, a.act, a.slope),
                      tnt_act_grad(p.z, v.z, a.act, a.slope), tnt_act_grad(p.w, v.w, a.act, a.slope));
      *reinterpret_cast<float4*>(d.y + o) = v;
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (NARROW) {                                              // every lane holds the same quad: wave sums, then the 16 waves
    acc.x = tnt_wave_sum(acc.x); acc.y = tnt_wave_sum(acc.y); acc.z = tnt_wave_sum(acc.z); acc.w = tnt_wave_sum(acc.w);
    if ((tid & 63) == 0) red[tid >> 6][0] = acc;
    __syncthreads();
    if (tid == 0 && c < d.cols) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 16; ++k) { const float4 u = red[k][0]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
      *reinterpret_cast<float4*>(a.dbias + c) = t;
    }
    return;
  }
  red[rl][cq] = acc;
  __syncthreads();
  if (rl == 0 && c < d.cols) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int k = 0; k < 128; ++k) { const float4 u = red[k][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *reinterpret_cast<float4*>(a.dbias + c) = t;
  }
}

extern "C" int32_t tnt_bias_act_drop_bwd_f32(const float* dy, const float* pre, float* dx, float* dbias, int32_t rows,
                                             int32_t cols, int32_t ld, int32_t act, float slope, int32_t tmajor_B,
                                             int32_t lwidth, int32_t lcol0, float rate, uint64_t seed, uint32_t site,
                                             const uint32_t* step_dev, const float* x1, float* out1, int32_t rows1,
                                             int32_t C1, int32_t ld1, void* stream) {
  if (rows <= 0 || cols <= 0 || rows > 2048 || ((cols | ld | lwidth | lcol0) & 3) != 0) return TNT_BADARG(4);
  if (!tnt_aligned16(dy) || !tnt_aligned16(pre) || !tnt_aligned16(dx) || !tnt_aligned16(dbias)) return TNT_BADARG(1);
  if (tmajor_B > 0 && rows % tmajor_B != 0) return TNT_BADARG(9);
  if (rate < 0.f || rate >= 1.f) return TNT_BADARG(12);
  if (x1 != nullptr && (rows1 <= 0 || C1 <= 0 || rows1 > 2048 || out1 == nullptr)) return TNT_BADARG(18);
  BadArgs a{};
  a.d.x = dy; a.d.y = dx; a.d.rows = rows; a.d.cols = cols; a.d.ld = ld; a.d.tB = tmajor_B; a.d.lwidth = lwidth;
  a.d.lcol0 = lcol0; a.d.rows_per_site = 0; a.d.rate = rate; a.d.scale = 1.0f / (1.0f - rate); a.d.seed = seed;
  a.d.site = site; a.d.step = 0; a.d.step_dev = step_dev;
  const bool narrow = cols <= 1024;                          // up to 256 workgroups of one column quad
  a.pre = pre; a.dbias = dbias; a.act = act; a.slope = slope; a.nb0 = narrow ? cols / 4 : (cols + 31) / 32;
  a.x1 = x1; a.out1 = out1; a.rows1 = rows1; a.C1 = C1; a.ld1 = ld1;
  const int nb1 = x1 ? (C1 + 31) / 32 : 0;
  if (narrow) hipLaunchKernelGGL(bias_act_drop_bwd_kernel<true>, dim3(a.nb0 + nb1), dim3(1024), 0, tnt_stream(stream), a);
  else hipLaunchKernelGGL(bias_act_drop_bwd_kernel<false>, dim3(a.nb0 + nb1), dim3(1024), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_colsum_f32(const float* x, float* out, int32_t rows, int32_t C, int32_t ld, float* work,
                                  void* stream) {
  hipStream_t s = tnt_stream(stream);
  if (rows <= 0 || C <= 0) return 0;
  if (rows <= 2048) {
    hipLaunchKernelGGL(col_sum_direct_kernel, dim3((C + 31) / 32), dim3(1024), 0, s, x, ld, rows, C, out);
    TNT_LAUNCH_CHECK();
    return 0;
  }
  const int nchunk = chunk_count(rows);
  launch_col_partial<2>(x, ld, nullptr, 0, rows, C, work, s);
  TNT_LAUNCH_CHECK();
  hipLaunchKernelGGL(col_finalize_kernel<false>, dim3(C), dim3(64), 0, s, work, C, nchunk, out,
                     (float*)nullptr);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_sum_f32(const float* x, float* out, int32_t n, float scale, void* stream) {
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, tnt_stream(stream), x, out, n, scale);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_sum2_f32(const float* x0, float* out0, const float* x1, float* out1, int32_t n, float scale,
                                void* stream) {
  hipLaunchKernelGGL(sum2_kernel, dim3(2), dim3(1024), 0, tnt_stream(stream), x0, out0, x1, out1, n, scale);
  TNT_LAUNCH_CHECK();
  return 0;
}

namespace {
template <typename XT>
int32_t stage_batch_launch(const XT* x, float* x_dst, const int32_t* cap, int32_t* cap_dst, const int32_t* tgt,
                           int32_t* tgt_tmajor, const float* a0, float* h0, const float* c0, float* c0_dst, int32_t B,
                           int32_t T, int32_t N, int32_t ldx, int32_t U, float* xT_dst, int32_t ldt, void* stream,
                           uint8_t* keep_out = nullptr, int64_t keep_n = 0, int32_t keep_sites = 0, float keep_rate = 0.f,
                           uint64_t keep_seed = 0, uint32_t keep_site0 = 0, const uint32_t* keep_step_dev = nullptr) {
  if (B <= 0 || T <= 0 || N <= 0 || ldx < N || U <= 0) return TNT_BADARG(11);
  if (keep_out != nullptr && (keep_n <= 0 || keep_n % 4 != 0 || keep_sites <= 0)) return TNT_BADARG(19);
  if (xT_dst && ldt < B) return TNT_BADARG(16);
  const bool vec = (N % 4 == 0) && (ldx % 4 == 0);
  const bool src_ok = sizeof(XT) == 4 ? tnt_aligned16(x) : ((reinterpret_cast<uintptr_t>(x) & 7u) == 0);
  if (vec && (!src_ok || !tnt_aligned16(x_dst))) return TNT_BADARG(1);
  StageArgs<XT> a{};
  a.x = x; a.xd = x_dst; a.cap = cap; a.capd = cap_dst; a.tgt = tgt; a.tgtd = tgt_tmajor; a.a0 = a0; a.h0 = h0; a.c0 = c0;
  a.c0d = c0_dst; a.B = B; a.T = T; a.N = N; a.ldx = ldx; a.U = U; a.xT = xT_dst; a.ldt = ldt;
  a.ncopy = ew_blocks((long)B * N / (vec ? 4 : 1));
  int nt = 0;
  if (xT_dst) { nt = ((N + 63) / 64) * ((B + 63) / 64); if (nt > 1024) nt = 1024; }
  a.nstage = a.ncopy + nt;
  long nmask = 0;
  if (keep_out) {
    a.mk_out = keep_out; a.mk_n4 = keep_n / 4; a.mk_total = a.mk_n4 * keep_sites; a.mk_rate = keep_rate; a.mk_seed = keep_seed;
    a.mk_site0 = keep_site0; a.mk_step_dev = keep_step_dev;
    nmask = (a.mk_total + 255) / 256;
    if (nmask > 8192) nmask = 8192;
  }
  hipLaunchKernelGGL(stage_batch_kernel<XT>, dim3(a.nstage + (int)nmask), dim3(256), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int32_t tnt_stage_batch_f32(const float* x, float* x_dst, const int32_t* cap, int32_t* cap_dst,
                                       const int32_t* tgt, int32_t* tgt_tmajor, const float* a0, float* h0,
                                       const float* c0, float* c0_dst, int32_t B, int32_t T, int32_t N, int32_t ldx,
                                       int32_t U, float* xT_dst, int32_t ldt, void* stream) {
  return stage_batch_launch<float>(x, x_dst, cap, cap_dst, tgt, tgt_tmajor, a0, h0, c0, c0_dst, B, T, N, ldx, U, xT_dst,
                                   ldt, stream);
}

extern "C" int32_t tnt_stage_batch_masks_f32(const float* x, float* x_dst, const int32_t* cap, int32_t* cap_dst,
                                             const int32_t* tgt, int32_t* tgt_tmajor, const float* a0, float* h0,
                                             const float* c0, float* c0_dst, int32_t B, int32_t T, int32_t N, int32_t ldx,
                                             int32_t U, float* xT_dst, int32_t ldt, uint8_t* keep_out, int64_t keep_n,
                                             int32_t keep_sites, float keep_rate, uint64_t keep_seed, uint32_t keep_site0,
                                             const uint32_t* keep_step_dev, void* stream) {
  return stage_batch_launch<float>(x, x_dst, cap, cap_dst, tgt, tgt_tmajor, a0, h0, c0, c0_dst, B, T, N, ldx, U, xT_dst,
                                   ldt, stream, keep_out, keep_n, keep_sites, keep_rate, keep_seed, keep_site0, keep_step_dev);
}

extern "C" int32_t tnt_stage_batch_h16(const uint16_t* x_half, float* x_dst, const int32_t* cap, int32_t* cap_dst,
                                       const int32_t* tgt, int32_t* tgt_tmajor, const float* a0, float* h0,
                                       const float* c0, float* c0_dst, int32_t B, int32_t T, int32_t N, int32_t ldx,
                                       int32_t U, float* xT_dst, int32_t ldt, void* stream) {
  return stage_batch_launch<__half>(reinterpret_cast<const __half*>(x_half), x_dst, cap, cap_dst, tgt, tgt_tmajor, a0,
                                    h0, c0, c0_dst, B, T, N, ldx, U, xT_dst, ldt, stream);
}

extern "C" int32_t tnt_version(void) { return 104; }
