// Region-wise ("locally dense") voxel encoder, forward and weight-gradient.
//
// Reference: layers.LocallyDense.call (AttemptFour/Model/layers.py:43-48): R separate
// tf.gather + Dense(n_r -> D, LeakyReLU(0.2)) layers launched from a Python list
// comprehension, then stacked to (B, R, D).  Here: ONE launch, one workgroup per region.
// Groups are CSR (idx[goff[r]..goff[r+1])), ragged, and may overlap (layers.py:13).
// The workgroup gathers its region's voxel columns of the (B x N) beta matrix into an LDS
// tile (all 256 threads issuing independent loads, lanes along the index list so
// neighbouring voxels share cache lines), stages the region's (n_r x D) kernel next to it,
// and runs v_mfma_f32_16x16x4_f32 over it; the backward reuses the same gathered tile as
// the transposed operand (dW_r = X_r^T dpre_r), so x is never re-laid-out in HBM.
#include "tnt_common.h"

namespace {

constexpr int KC = 128;          // voxels of a region handled per LDS chunk
constexpr int XLD = KC + 2;      // Xs row stride: (lr*XLD + kq) hits 32 distinct banks
constexpr int MAXD = 64;
constexpr int WLD = MAXD + 16;   // Ws / Ds row stride (stride % 32 == 16)

struct EncArgs {
  const float* x; int ldx; const int* idx; const int* goff; const float* W; const float* bias;
  float* pre; float* y; const float* dpre; float* dW; float* db;
  int B, R, D; float slope;
};

// gather Xs[row][k] = x[row][idx[g0 + k0 + k]] for row < 64, k < kc
// The index list is staged in LDS first so that the voxel loads are independent of each other
// (8 in flight per thread) instead of each waiting on its own index load.
__device__ __forceinline__ void gather_tile(const EncArgs& g, float* Xs, int* Is, int g0, int k0, int kc) {
  for (int k = threadIdx.x; k < kc; k += 256) Is[k] = g.idx[g0 + k0 + k];
  __syncthreads();
  const int total = 64 * kc;
  for (int e0 = threadIdx.x; e0 < total; e0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256;
      const int row = e / kc, k = e % kc;
      v[u] = (e < total && row < g.B) ? g.x[(long)row * g.ldx + Is[k]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256;
      if (e < total) Xs[(e / kc) * XLD + e % kc] = v[u];
    }
  }
}

__global__ __launch_bounds__(256) void locally_dense_fwd_kernel(EncArgs g) {
  __shared__ float Xs[64 * XLD];
  __shared__ float Ws[KC * WLD];
  __shared__ int Is[KC];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, lr = lane & 15;
  const int g0 = g.goff[r], nr = g.goff[r + 1] - g0;
  const int DT = g.D / 16;
  floatx4 acc[MAXD / 16];
#pragma unroll
  for (int c = 0; c < MAXD / 16; ++c) acc[c] = (floatx4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < nr; k0 += KC) {
    const int kc = min(KC, nr - k0);
    const int kc4 = (kc + 3) & ~3;
    __syncthreads();
    gather_tile(g, Xs, Is, g0, k0, kc);
    // zero the k padding up to a multiple of 4 so the MFMA tail multiplies zeros
    for (int e = tid; e < 64 * (kc4 - kc); e += 256) Xs[(e / (kc4 - kc)) * XLD + kc + e % (kc4 - kc)] = 0.f;
    for (int e = tid; e < kc4 * g.D; e += 256) {
      const int k = e / g.D, d = e % g.D;
      Ws[k * WLD + d] = k < kc ? g.W[(long)(g0 + k0 + k) * g.D + d] : 0.f;
    }
    __syncthreads();
    // wave w owns batch rows [16w, 16w+16)
    for (int k = 0; k < kc4; k += 4) {
      const float av = Xs[(w * 16 + lr) * XLD + k + kq];
#pragma unroll
      for (int c = 0; c < MAXD / 16; ++c)
        if (c < DT) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Ws[(k + kq) * WLD + c * 16 + lr], acc[c], 0, 0, 0);
    }
  }
  // C/D map: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int c = 0; c < MAXD / 16; ++c) {
    if (c >= DT) continue;
    const int d = c * 16 + lr;
    const float bd = g.bias[(long)r * g.D + d];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int b = w * 16 + kq * 4 + j;
      if (b < g.B) {
        const float p = acc[c][j] + bd;
        const long o = ((long)b * g.R + r) * g.D + d;
        g.pre[o] = p;
        g.y[o] = p > 0.f ? p : p * g.slope;
      }
    }
  }
}

__global__ __launch_bounds__(256) void locally_dense_bwd_kernel(EncArgs g) {
  __shared__ float Xs[64 * XLD];
  __shared__ float Ds[64 * WLD];
  __shared__ int Is[KC];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, lr = lane & 15;
  const int g0 = g.goff[r], nr = g.goff[r + 1] - g0;
  const int DT = g.D / 16;
  // stage dpre[:, r, :] (rows beyond B are zero)
  for (int e = tid; e < 64 * g.D; e += 256) {
    const int b = e / g.D, d = e % g.D;
    Ds[b * WLD + d] = b < g.B ? g.dpre[((long)b * g.R + r) * g.D + d] : 0.f;
  }
  __syncthreads();
  if (tid < g.D) {
    float s = 0.f;
    for (int b = 0; b < 64; ++b) s += Ds[b * WLD + tid];
    g.db[(long)r * g.D + tid] = s;
  }
  for (int k0 = 0; k0 < nr; k0 += KC) {
    const int kc = min(KC, nr - k0);
    __syncthreads();
    gather_tile(g, Xs, Is, g0, k0, kc);
    __syncthreads();
    // dW[k][d] = sum_b Xs[b][k] * Ds[b][d]; wave w owns m-tiles w, w+4, ...
    for (int mt = w; mt * 16 < kc; mt += 4) {
      floatx4 acc[MAXD / 16];
#pragma unroll
      for (int c = 0; c < MAXD / 16; ++c) acc[c] = (floatx4){0.f, 0.f, 0.f, 0.f};
      const int m = mt * 16 + lr;
      for (int b0 = 0; b0 < 64; b0 += 4) {
        const float av = m < kc ? Xs[(b0 + kq) * XLD + m] : 0.f;
#pragma unroll
        for (int c = 0; c < MAXD / 16; ++c)
          if (c < DT) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Ds[(b0 + kq) * WLD + c * 16 + lr], acc[c], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < MAXD / 16; ++c) {
        if (c >= DT) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = mt * 16 + kq * 4 + j;
          if (k < kc) g.dW[(long)(g0 + k0 + k) * g.D + c * 16 + lr] = acc[c][j];
        }
      }
    }
  }
}

}  // namespace

extern "C" int32_t tnt_locally_dense_fwd_f32(const float* x, int32_t ldx, const int32_t* idx, const int32_t* goff,
                                             const float* W, const float* bias, float* pre, float* y, int32_t B,
                                             int32_t R, int32_t D, float slope, void* stream) {
  if (B <= 0 || B > 64) return TNT_BADARG(9);
  if (D <= 0 || D > MAXD || D % 16 != 0) return TNT_BADARG(11);
  if (R <= 0) return TNT_BADARG(10);
  EncArgs g{};
  g.x = x; g.ldx = ldx; g.idx = idx; g.goff = goff; g.W = W; g.bias = bias; g.pre = pre; g.y = y;
  g.B = B; g.R = R; g.D = D; g.slope = slope;
  hipLaunchKernelGGL(locally_dense_fwd_kernel, dim3(R), dim3(256), 0, tnt_stream(stream), g);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_locally_dense_bwd_f32(const float* x, int32_t ldx, const int32_t* idx, const int32_t* goff,
                                             const float* dpre, float* dW, float* db, int32_t B, int32_t R, int32_t D,
                                             void* stream) {
  if (B <= 0 || B > 64) return TNT_BADARG(8);
  if (D <= 0 || D > MAXD || D % 16 != 0) return TNT_BADARG(10);
  if (R <= 0) return TNT_BADARG(9);
  EncArgs g{};
  g.x = x; g.ldx = ldx; g.idx = idx; g.goff = goff; g.dpre = dpre; g.dW = dW; g.db = db;
  g.B = B; g.R = R; g.D = D;
  hipLaunchKernelGGL(locally_dense_bwd_kernel, dim3(R), dim3(256), 0, tnt_stream(stream), g);
  TNT_LAUNCH_CHECK();
  return 0;
}
