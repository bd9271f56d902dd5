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
#include "tnt_fin.h"

namespace {

constexpr int KC = 128;          // voxels of a region handled per LDS chunk
constexpr int XLD = KC + 2;      // Xs row stride: (lr*XLD + kq) hits 32 distinct banks
constexpr int MAXD = 64;
constexpr int WLD = MAXD + 16;   // Ws / Ds row stride (stride % 32 == 16)

struct EncArgs {
  const float* x; int ldx; const int* idx; const int* goff; const float* W; const float* bias;
  float* pre; float* y; const float* dpre; float* dW; float* db;
  int B, R, D; float slope; int accumulate;
  // split mode (nullable): the launch runs over VIRTUAL regions -- contiguous pieces of at most `piece` voxels of a
  // region's index list -- so that one large region does not set the kernel time (Glasser regions span 8..400+
  // voxels).  vgoff[v..v+1] = CSR range of piece v, vreg[v] = its region, vfirst[v] = 1 for a region's first piece.
  const int* vgoff; const int* vreg; const int* vfirst; float* partial;
  int x_voxel_major;       // 1: x is [n_voxels][ldx] (a voxel's batch values contiguous), 0: [B][ldx]
};

// gather Xs[row][k] = x[row][idx[g0 + k0 + k]] for row < 64, k < kc
// The index list is staged in LDS first so that the voxel loads are independent of each other
// (8 in flight per thread) instead of each waiting on its own index load.
__device__ __forceinline__ void gather_tile(const EncArgs& g, float* Xs, int* Is, int g0, int k0, int kc) {
  for (int k = threadIdx.x; k < kc; k += 256) Is[k] = g.idx[g0 + k0 + k];
  __syncthreads();
  if (g.x_voxel_major) {
    // one voxel = 64 consecutive floats: a wave reads whole 256-byte rows; every byte of x is fetched once per piece
    // (the batch-major gather below touches a different 128-byte line for each of its 4-byte elements)
    const int total4 = 16 * kc;                          // float4 per tile: kc voxels x 16
    for (int e0 = threadIdx.x; e0 < total4; e0 += 256 * 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 256;
        const int k = e >> 4, r4 = (e & 15) * 4;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < total4) {
          const float* src = g.x + (long)Is[k] * g.ldx + r4;
          if (r4 + 3 < g.B) v[u] = *reinterpret_cast<const float4*>(src);
          else { if (r4 < g.B) v[u].x = src[0]; if (r4 + 1 < g.B) v[u].y = src[1]; if (r4 + 2 < g.B) v[u].z = src[2]; }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 256;
        if (e < total4) {
          const int k = e >> 4, r4 = (e & 15) * 4;
          Xs[(r4 + 0) * XLD + k] = v[u].x; Xs[(r4 + 1) * XLD + k] = v[u].y;
          Xs[(r4 + 2) * XLD + k] = v[u].z; Xs[(r4 + 3) * XLD + k] = v[u].w;
        }
      }
    }
    return;
  }
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
  const int vr = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, lr = lane & 15;
  const int r = g.vreg ? g.vreg[vr] : vr;
  const int g0 = g.vreg ? g.vgoff[vr] : g.goff[r], nr = (g.vreg ? g.vgoff[vr + 1] : g.goff[r + 1]) - g0;
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
        if (g.vreg) {                        // split mode: partial sum of this piece, combined by the kernel below
          g.partial[((long)vr * 64 + b) * g.D + d] = acc[c][j];
        } else {
          const float p = acc[c][j] + bd;
          const long o = ((long)b * g.R + r) * g.D + d;
          g.pre[o] = p;
          g.y[o] = p > 0.f ? p : p * g.slope;
        }
      }
    }
  }
}

// split mode, second half of the forward: pre[b][r][:] = bias_r + sum over the pieces of region r (in piece order)
__global__ __launch_bounds__(256) void locally_dense_combine_kernel(EncArgs g, const int* rfirst) {
  const int r = blockIdx.x;
  const int v0 = rfirst[r], v1 = rfirst[r + 1];
  for (int e = threadIdx.x; e < g.B * g.D; e += 256) {
    const int b = e / g.D, d = e % g.D;
    float p = g.bias[(long)r * g.D + d];
    for (int v = v0; v < v1; ++v) p += g.partial[((long)v * 64 + b) * g.D + d];
    const long o = ((long)b * g.R + r) * g.D + d;
    g.pre[o] = p;
    g.y[o] = p > 0.f ? p : p * g.slope;
  }
}

__global__ __launch_bounds__(256) void locally_dense_bwd_kernel(EncArgs g) {
  __shared__ float Xs[64 * XLD];
  __shared__ float Ds[64 * WLD];
  __shared__ int Is[KC];
  const int vr = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, lr = lane & 15;
  const int r = g.vreg ? g.vreg[vr] : vr;
  const int g0 = g.vreg ? g.vgoff[vr] : g.goff[r], nr = (g.vreg ? g.vgoff[vr + 1] : g.goff[r + 1]) - g0;
  const bool first = g.vreg ? g.vfirst[vr] != 0 : true;      // the bias gradient belongs to the region, not the piece
  const int DT = g.D / 16;
  // stage dpre[:, r, :] (rows beyond B are zero)
  for (int e = tid; e < 64 * g.D; e += 256) {
    const int b = e / g.D, d = e % g.D;
    Ds[b * WLD + d] = b < g.B ? g.dpre[((long)b * g.R + r) * g.D + d] : 0.f;
  }
  __syncthreads();
  if (first && tid < g.D) {
    float s = 0.f;
    for (int b = 0; b < 64; ++b) s += Ds[b * WLD + tid];
    float* o = g.db + (long)r * g.D + tid;
    *o = g.accumulate ? *o + s : s;
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
          if (k < kc) {
            float* o = g.dW + (long)(g0 + k0 + k) * g.D + c * 16 + lr;
            *o = g.accumulate ? *o + acc[c][j] : acc[c][j];
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" int32_t tnt_locally_dense_fwd_f32(const float* x, int32_t ldx, const int32_t* idx, const int32_t* goff,
                                             const float* W, const float* bias, float* pre, float* y, int32_t B,
                                             int32_t R, int32_t D, float slope, void* stream) {
  if (B <= 0) return TNT_BADARG(9);
  if (D <= 0 || D > MAXD || D % 16 != 0) return TNT_BADARG(11);
  if (R <= 0) return TNT_BADARG(10);
  for (int b0 = 0; b0 < B; b0 += 64) {         // the kernels tile 64 batch rows; larger batches go in row blocks
    EncArgs g{};
    g.x = x + (long)b0 * ldx; g.ldx = ldx; g.idx = idx; g.goff = goff; g.W = W; g.bias = bias;
    g.pre = pre + (long)b0 * R * D; g.y = y + (long)b0 * R * D;
    g.B = B - b0 < 64 ? B - b0 : 64; g.R = R; g.D = D; g.slope = slope;
    hipLaunchKernelGGL(locally_dense_fwd_kernel, dim3(R), dim3(256), 0, tnt_stream(stream), g);
    TNT_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int32_t tnt_locally_dense_bwd_f32(const float* x, int32_t ldx, const int32_t* idx, const int32_t* goff,
                                             const float* dpre, float* dW, float* db, int32_t B, int32_t R, int32_t D,
                                             void* stream) {
  if (B <= 0) return TNT_BADARG(8);
  if (D <= 0 || D > MAXD || D % 16 != 0) return TNT_BADARG(10);
  if (R <= 0) return TNT_BADARG(9);
  for (int b0 = 0; b0 < B; b0 += 64) {         // row blocks of 64; blocks after the first accumulate (fixed order)
    EncArgs g{};
    g.x = x + (long)b0 * ldx; g.ldx = ldx; g.idx = idx; g.goff = goff; g.dpre = dpre + (long)b0 * R * D;
    g.dW = dW; g.db = db; g.B = B - b0 < 64 ? B - b0 : 64; g.R = R; g.D = D; g.accumulate = b0 > 0;
    hipLaunchKernelGGL(locally_dense_bwd_kernel, dim3(R), dim3(256), 0, tnt_stream(stream), g);
    TNT_LAUNCH_CHECK();
  }
  return 0;
}

// Split mode: the same two kernels launched over pieces of at most `piece` voxels (tables built once by the host from
// goff: vgoff[NV+1], vreg[NV], vfirst[NV], rfirst[R+1]); forward = piece partials + a combine launch.
// partial: NV * 64 * D floats (one 64-row block at a time).
extern "C" int32_t tnt_locally_dense_fwd_split_f32(const float* x, int32_t ldx, const int32_t* idx, const int32_t* vgoff,
                                                   const int32_t* vreg, const int32_t* rfirst, int32_t NV,
                                                   const float* W, const float* bias, float* pre, float* y,
                                                   float* partial, int32_t B, int32_t R, int32_t D, float slope,
                                                   int32_t x_voxel_major, void* stream) {
  if (B <= 0) return TNT_BADARG(13);
  if (x_voxel_major && (ldx % 4 != 0 || !tnt_aligned16(x))) return TNT_BADARG(2);
  if (D <= 0 || D > MAXD || D % 16 != 0) return TNT_BADARG(15);
  if (R <= 0 || NV < R) return TNT_BADARG(7);
  hipStream_t s = tnt_stream(stream);
  for (int b0 = 0; b0 < B; b0 += 64) {
    EncArgs g{};
    g.x = x_voxel_major ? x + b0 : x + (long)b0 * ldx; g.ldx = ldx; g.idx = idx; g.W = W; g.bias = bias;
    g.pre = pre + (long)b0 * R * D; g.y = y + (long)b0 * R * D;
    g.B = B - b0 < 64 ? B - b0 : 64; g.R = R; g.D = D; g.slope = slope;
    g.vgoff = vgoff; g.vreg = vreg; g.partial = partial; g.x_voxel_major = x_voxel_major;
    hipLaunchKernelGGL(locally_dense_fwd_kernel, dim3(NV), dim3(256), 0, s, g);
    TNT_LAUNCH_CHECK();
    hipLaunchKernelGGL(locally_dense_combine_kernel, dim3(R), dim3(256), 0, s, g, rfirst);
    TNT_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int32_t tnt_locally_dense_bwd_split_f32(const float* x, int32_t ldx, const int32_t* idx, const int32_t* vgoff,
                                                   const int32_t* vreg, const int32_t* vfirst, int32_t NV,
                                                   const float* dpre, float* dW, float* db, int32_t B, int32_t R,
                                                   int32_t D, int32_t x_voxel_major, void* stream) {
  if (B <= 0) return TNT_BADARG(11);
  if (x_voxel_major && (ldx % 4 != 0 || !tnt_aligned16(x))) return TNT_BADARG(2);
  if (D <= 0 || D > MAXD || D % 16 != 0) return TNT_BADARG(13);
  if (R <= 0 || NV < R) return TNT_BADARG(7);
  for (int b0 = 0; b0 < B; b0 += 64) {
    EncArgs g{};
    g.x = x_voxel_major ? x + b0 : x + (long)b0 * ldx; g.ldx = ldx; g.idx = idx; g.dpre = dpre + (long)b0 * R * D;
    g.dW = dW; g.db = db; g.B = B - b0 < 64 ? B - b0 : 64; g.R = R; g.D = D; g.accumulate = b0 > 0;
    g.vgoff = vgoff; g.vreg = vreg; g.vfirst = vfirst; g.x_voxel_major = x_voxel_major;
    hipLaunchKernelGGL(locally_dense_bwd_kernel, dim3(NV), dim3(256), 0, tnt_stream(stream), g);
    TNT_LAUNCH_CHECK();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Skinny-K weight gradient of the dense voxel encoder:  dW[N][E] = X^T[N][Bk] @ dpre[Bk][E]  with Bk <= 64
// (tape.gradient of NIC.dense_in / ThinkAndTell Encoder.fc, NIC.py:64-69,248-249; model.py:22-33).
// 41 MB of output from 5 MB + 128 KB of input: 8.3 us of FP32 matrix work and 8 us of HBM writes -- the generic
// tiled GEMM spends 28 us on it (2504 short-lived 64x64 tiles, K = 64 = two chunks: all prologue and epilogue).
// Here 256 persistent workgroups walk over 16-row strips of voxels.  The B operand (dpre) never changes, so
// each of the 8 waves keeps ITS fragments of it (<= 16 k-steps x 4 column tiles = 64 VGPRs) in registers for
// the whole kernel; the strip's X^T tile (Bk x 16) is staged through a double-buffered LDS tile one strip
// ahead, so a strip is 16 ds_read_b32 + 64 back-to-back v_mfma_f32_16x16x4_f32 + 16 row-piece stores.
namespace {

constexpr int DW_MS = 16;            // voxel rows per strip
constexpr int DW_LDA = 48;           // As row stride: 48 % 32 == 16 -> the two k-rows of a half-wave hit disjoint banks
constexpr int DW_KS = 16;            // k-steps of 4 (Bk <= 64)

struct DwArgs {
  const float* x; const float* dpre; float* dw; int N, E, Bk, ldx;
  // EPI 1 (squared norms) / EPI 2 (clip + Adam), see dense_dw_skinny_kernel
  float* theta; float* m; float* v; float* partial; int nslot; float lam2;
  const float* sq; const float* sq_override; const float* lr_t_dev; float b1, b2, eps, clipnorm; const uint32_t* guard;
  // EPI 2 without a finalize launch in front (tnt_dense_dw_adam_fin_f32): the variable's clip norm is summed here, by
  // every wave, from its span partials fin_partial[2 k], fin_k0 <= k < fin_k1 (tnt_seg_sums: the canonical order)
  const float* fin_partial; int fin_k0, fin_k1;
};

// PERM (every wave has all its TPW tiles, TPW = 2 or 4): MFMA column n of tile j is output column TPW n + j of the
// wave's 16 TPW columns, so a lane's TPW results of one row are adjacent and leave as ONE dwordx2/x4 store covering
// whole 128-byte lines per 16 lanes (4 stores per strip instead of 16 sixty-four-byte pieces).
//
// EPI (needs PERM, TPW = 4) -- the same product without ever writing dW, for the single-process optimizer step
// (optimizer.apply_gradients with clipnorm + Adam, NIC.py:250-251 / main.py:97):
//   1  per-workgroup partials of sum (g + 2 lambda theta)^2 and sum theta^2 over the workgroup's strips, written in the
//      span-partial layout of tnt_span_sqnorm_f32 (slot = workgroup; the other slots of the variable get 0);
//   2  clip + Adam on the strip as it leaves the MFMAs: theta, m, v stream through registers one strip ahead (their
//      loads are issued before the previous strip's stores, so waiting for them never drains the stores), g is never
//      stored -- 24 bytes per parameter instead of 4 (dW write) + 8 (norm pass) + 28 (Adam).
template <int TPW, bool PERM, int EPI = 0, bool NTM = false>   // 16-column tiles per wave; a workgroup covers 8 * TPW * 16 columns (blockIdx.y picks the group); NTM: non-temporal moments
__global__ __launch_bounds__(512) void dense_dw_skinny_kernel(DwArgs a) {
  static_assert(EPI == 0 || (PERM && TPW == 4), "fused epilogues use the vector row layout");
  constexpr int DW_MAXT = TPW;
  if (EPI == 2 && a.guard && a.guard[0] != 0u) return;       // the step's forward pass was invalid: leave the model untouched
  __shared__ float As[2][DW_KS * 4 * DW_LDA];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lk = lane >> 4, lc = lane & 15;
  const int NT = a.E / 16, t0 = (blockIdx.y * 8 + w) * TPW;        // 16-column tiles of this wave: t0 .. t0+TPW
  // B fragments of this wave, resident for the whole kernel: bq[ks][j] = dpre[4 ks + lk][(t0 + j) * 16 + lc].
  // Branch-free: out-of-range rows / tiles read a clamped address and are multiplied by 0.
  float bq[DW_KS][DW_MAXT];
  auto load_bq = [&]() {
#pragma unroll
    for (int ks = 0; ks < DW_KS; ++ks) {
      const int k = 4 * ks + lk;
      if (PERM && TPW == 4) {            // the lane's four adjacent columns: one 16-byte load per k-step
        const float keep = k < a.Bk ? 1.f : 0.f;
        const float4 b4 = *reinterpret_cast<const float4*>(a.dpre + (long)min(k, a.Bk - 1) * a.E + t0 * 16 + 4 * lc);
        bq[ks][0] = keep * b4.x; bq[ks][DW_MAXT > 1 ? 1 : 0] = keep * b4.y;
        bq[ks][DW_MAXT > 2 ? 2 : 0] = keep * b4.z; bq[ks][DW_MAXT > 3 ? 3 : 0] = keep * b4.w;
        continue;
      }
#pragma unroll
      for (int j = 0; j < DW_MAXT; ++j) {
        const int t = t0 + j;
        const float keep = (k < a.Bk && t < NT) ? 1.f : 0.f;
        bq[ks][j] = PERM ? keep * a.dpre[(long)min(k, a.Bk - 1) * a.E + t0 * 16 + TPW * lc + j]
                         : keep * a.dpre[(long)min(k, a.Bk - 1) * a.E + min(t, NT - 1) * 16 + lc];
      }
    }
  };
  const int nstrip = (a.N + DW_MS - 1) / DW_MS;
  // A staging: 64 x 16 floats per strip = 2 per thread; thread -> (k = e / 16, m = e % 16).  Addresses are hoisted:
  // per thread a fixed row pointer, advanced by a strip; columns past N are clamped (their products are never stored).
  float pa[2];
  const float* xrow[2];
  float xkeep[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * 512, k = e >> 4;
    xrow[i] = a.x + (long)min(k, a.Bk - 1) * a.ldx;
    xkeep[i] = k < a.Bk ? 1.f : 0.f;
  }
  const int mcol = tid & 15;
  auto gload = [&](int s) {
    const int m = min(min(s, nstrip - 1) * DW_MS + mcol, a.N - 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) pa[i] = xkeep[i] * xrow[i][m];
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * 512;
      As[buf][(e >> 4) * DW_LDA + (e & 15)] = pa[i];
    }
  };
  // C/D map of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg.  Row pointers hoisted; full strips (all but
  // possibly the last) and full tile sets store without per-lane predicates.
  float* wrow = a.dw + (long)(lk * 4) * a.E + t0 * 16 + (PERM ? TPW * lc : lc);
  const bool tiles_full = t0 + DW_MAXT <= NT;
  auto store_strip = [&](const floatx4* o, int st) {
    float* p0 = wrow + (long)st * DW_MS * a.E;
    if (PERM) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (st * DW_MS + lk * 4 + r >= a.N) continue;
        if (TPW == 4) *reinterpret_cast<float4*>(p0 + (long)r * a.E) = make_float4(o[0][r], o[1][r], o[2][r], o[3][r]);
        else *reinterpret_cast<float2*>(p0 + (long)r * a.E) = make_float2(o[0][r], o[TPW > 1 ? 1 : 0][r]);
      }
      return;
    }
    if (tiles_full && st * DW_MS + DW_MS <= a.N) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < DW_MAXT; ++j) p0[(long)r * a.E + j * 16] = o[j][r];
      return;
    }
#pragma unroll
    for (int j = 0; j < DW_MAXT; ++j) {
      if (t0 + j < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (st * DW_MS + lk * 4 + r < a.N) p0[(long)r * a.E + j * 16] = o[j][r];
      }
    }
  };
  int s = blockIdx.x, cur = 0, sprev = -1;
  floatx4 outp[DW_MAXT];
  // fused epilogues: this lane's 4 rows x 4 adjacent columns of theta / m / v for a strip
  const long eoff = (long)(lk * 4) * a.E + t0 * 16 + TPW * lc;
  float4 th[4], mm[4], vv[4], thn[4], mmn[4], vvn[4];
  float q = 0.f, wq = 0.f, cs = 1.f, lr_t = 0.f;
  auto tload = [&](int st, float4* t, float4* m_, float4* v_) {
    const int sc = min(st, nstrip - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long o = ((long)min(sc * DW_MS + lk * 4 + r, a.N - 1) - lk * 4) * a.E + eoff;
      t[r] = *reinterpret_cast<const float4*>(a.theta + o);
      if (EPI == 2) { m_[r] = tnt_ld4<NTM>(a.m + o); v_[r] = tnt_ld4<NTM>(a.v + o); }
    }
  };
  gload(s);                                            // the first strip's X is in flight while the B fragments load
  load_bq();
  sstore(0);
  gload(s + gridDim.x);
  if (EPI >= 1) tload(s, thn, mmn, vvn);
  // (behind the first strip's loads: the clip norm may come from a chain of dependent loads, tnt_dense_dw_adam_fin_f32)
  if (EPI == 2) {
    cs = 1.f;
    if (a.clipnorm > 0.f) {
      float sqv = a.fin_partial ? tnt_seg_sums(a.fin_partial, a.fin_k0, a.fin_k1, lane).x : a.sq[0];
      if (a.sq_override && a.sq_override[0] >= 0.f) sqv = a.sq_override[0];
      cs = a.clipnorm / fmaxf(sqrtf(sqv), a.clipnorm);
    }
    lr_t = a.lr_t_dev[0];
  }
  __syncthreads();
  for (; s < nstrip; s += gridDim.x, cur ^= 1) {
    float av[DW_KS];
#pragma unroll
    for (int ks = 0; ks < DW_KS; ++ks) av[ks] = As[cur][(4 * ks + lk) * DW_LDA + lc];
    // Stores count in vmcnt on gfx9, and the compiler drains it (vmcnt(0)) before the prefetched tile may be
    // written to LDS.  So the order inside an iteration is: consume the prefetch (everything still outstanding was
    // issued before the previous strip's 64 MFMAs and has long landed), THEN issue the previous strip's stores and
    // the next prefetch, THEN multiply -- the stores drain under the MFMAs instead of in front of a wait (PMC before
    // this reordering: waves 60 % in s_waitcnt, MFMA pipe 30 % busy).
    sstore(cur ^ 1);                                   // next strip (already in registers)
    if (EPI == 0 && sprev >= 0) store_strip(outp, sprev);
    gload(s + 2 * gridDim.x);                          // the one after, in flight during the MFMAs
    if (EPI >= 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        th[r] = thn[r];
        if (EPI == 2) { mm[r] = mmn[r]; vv[r] = vvn[r]; }
      }
      tload(s + gridDim.x, thn, mmn, vvn);             // next strip's state: a strip of MFMAs ahead of its use, and ahead of
      //                                                  this strip's stores (waiting for it never drains them)
    }
    floatx4 acc[DW_MAXT];
#pragma unroll
    for (int j = 0; j < DW_MAXT; ++j) acc[j] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < DW_KS; ++ks)
#pragma unroll
      for (int j = 0; j < DW_MAXT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bq[ks][j], acc[j], 0, 0, 0);
    if (EPI == 0) {
#pragma unroll
      for (int j = 0; j < DW_MAXT; ++j) outp[j] = acc[j];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (s * DW_MS + lk * 4 + r >= a.N) continue;
        const float g4[4] = {acc[0][r], acc[TPW > 1 ? 1 : 0][r], acc[TPW > 2 ? 2 : 0][r], acc[TPW > 3 ? 3 : 0][r]};
        float* wp = &th[r].x;
        if (EPI == 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a0 = g4[j] + a.lam2 * wp[j];
            q += a0 * a0; wq += wp[j] * wp[j];
          }
        } else {
          float* mp = &mm[r].x; float* vp = &vv[r].x;
          const float ob1 = 1.f - a.b1, ob2 = 1.f - a.b2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {                 // exactly adam_kernel's arithmetic (optim.hip)
            const float g = (g4[j] + a.lam2 * wp[j]) * cs;
            mp[j] = mp[j] + (g - mp[j]) * ob1;
            vp[j] = vp[j] + (g * g - vp[j]) * ob2;
            wp[j] = wp[j] - lr_t * mp[j] / (sqrtf(vp[j]) + a.eps);
          }
          const long o = (long)s * DW_MS * a.E + (long)r * a.E + eoff;
          *reinterpret_cast<float4*>(a.theta + o) = th[r];
          tnt_st4<NTM>(a.m + o, mm[r]);
          tnt_st4<NTM>(a.v + o, vv[r]);
        }
      }
    }
    sprev = s;
    // As[cur^1] complete; everyone done reading As[cur].  Only LDS traffic has to be ordered (not vmcnt).
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (EPI == 0 && sprev >= 0) store_strip(outp, sprev);
  if (EPI == 1) {
    __shared__ float rq[8], rw[8];
    q = tnt_wave_sum(q); wq = tnt_wave_sum(wq);
    if (lane == 0) { rq[w] = q; rw[w] = wq; }
    __syncthreads();
    if (tid == 0) {
      float sq_ = 0.f, sw_ = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { sq_ += rq[k]; sw_ += rw[k]; }
      const int slot = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
      a.partial[2 * slot] = sq_; a.partial[2 * slot + 1] = sw_;
      for (int k = slot + nwg; k < a.nslot; k += nwg) { a.partial[2 * k] = 0.f; a.partial[2 * k + 1] = 0.f; }
    }
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------
// Skinny-M forward of the dense voxel encoder:  part[s][B][E] = X[B][K-slice s] @ W[K-slice s][E]  with B <= 64 per
// row block (NIC.dense_in / ThinkAndTell Encoder.fc, NIC.py:64-69,125; model.py:22-33).  The product streams the
// 41 MB kernel once for 1.3 GFLOP: 6.5 us of HBM and 8.4 us of FP32 matrix work, where the generic tiled GEMM with
// split-K spends 20 us (512 short-lived 64x64 tiles of ten LDS-staged chunks each) plus a 7 us reduce launch.
// Here nothing goes through LDS on the way in.  A workgroup owns 32 output columns and one K split; its four waves
// split the K range once more, tile by tile of 16 k, and every wave accumulates the full 64 x 32 block:
//   A fragments  one dwordx4 per row tile: lane (kq, m) reads X[16 rt + m][k0 + 4 kq .. +3]  (k permuted inside the
//                tile: MFMA step j multiplies k0 + 4 kq + j, the same on both operands),
//   B fragments  one dwordx2 per k row:   lane (kq, n) reads W[k0 + 4 kq + j][c0 + 2 n, +1]  -- 16 lanes cover one
//                128-byte line; column tile c of the MFMA is column 2 n + c (a permutation undone at the store),
// i.e. 8 loads per 32 MFMAs, a ring of register stages per wave.  Tile i belongs to wave slot i % (4 nsplit), so at
// any moment the chip reads one contiguous window of W.  The four wave partials meet in LDS; the <= 16 K-split
// partials (2 MB) are summed by the encoder-tail kernel (tnt_enc_tail_fwd_sk_f32), which needs whole columns anyway.
namespace {
constexpr int DF_CW = 32;            // output columns per workgroup

struct DfArgs { const float* x; const float* w; float* part; int B, E, K, ldx, ldw, nsplit; float* gx_part; float* w2_part; };

// GRAM (training, when the optimizer step will consume the kernel's gradient X^T D without writing it): two by-products
// that let its clip-by-norm factor be computed from 64 x 64 matrices instead of a pass over the 10 M-element gradient --
//   gx_part[split][64][64]  K-split partials of the Gram matrix X X^T: the A fragments of row tile rt' in a lane are
//                           exactly the B fragments of column tile rt' of X^T, so tile (rt, rt') costs 4 extra MFMAs per
//                           k-tile from registers already loaded; the 16 column groups of a split take one tile each;
//   w2_part[workgroup]      sum of squares of the W elements the workgroup streamed (each element exactly once).
// ||X^T D||_F^2 = sum_{b,b'} (X X^T)[b,b'] (D D^T)[b,b'];  tnt_dense_gram_norm_f32 finishes the job.
template <int NW, int DEPTH, bool GRAM>
__global__ __launch_bounds__(64 * NW) void dense_fwd_stream_kernel(DfArgs a) {
  __shared__ float red[NW][64][DF_CW + 4];
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, ln = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroup -> (column group, K split).  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2:
  // when the splits divide by 8, an XCD gets ALL column groups of its nsplit / 8 splits, so its L2 sees whole rows of
  // W and only its own eighth of X.
  const int nct = a.E / DF_CW;
  int ct, split;
  if (a.nsplit % 8 == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    ct = slot % nct; split = xcd * (a.nsplit / 8) + slot / nct;
  } else {
    ct = blockIdx.x % nct; split = blockIdx.x / nct;
  }
  const int c0 = ct * DF_CW, rb = blockIdx.y * 64;
  const int nslot = a.nsplit * NW, q = split * NW + w;
  const int Tfull = a.K / 16;                       // full 16-k tiles; a K % 16 remainder is one masked tile at the end
  const int nt = q < Tfull ? (Tfull - q + nslot - 1) / nslot : 0;
  // Addresses = uniform tile base (scalar registers, scalar arithmetic) + a per-lane byte offset that never changes:
  // the loop issues no vector address arithmetic at all.  Rows past B read row B-1 (their outputs are never stored:
  // an output row of the MFMA depends on the same row of A only).
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) aoff[rt] = ((unsigned)min(rb + rt * 16 + ln, a.B - 1) * a.ldx + kq * 4) * 4u;
#pragma unroll
  for (int j = 0; j < 4; ++j) boff[j] = ((unsigned)(kq * 4 + j) * a.ldw + c0 + 2 * ln) * 4u;
  floatx4 acc[4][2];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[rt][c] = (floatx4){0.f, 0.f, 0.f, 0.f};
  float4 av[DEPTH][4];
  float2 bv[DEPTH][4];
  floatx4 accg = {0.f, 0.f, 0.f, 0.f};
  float w2 = 0.f;
  const int g_rt = (ct >> 2) & 3, g_rtp = ct & 3;   // this workgroup's Gram tile (column groups >= 16 repeat: not stored)
  float ga[4], gb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { ga[t] = g_rt == t ? 1.f : 0.f; gb[t] = g_rtp == t ? 1.f : 0.f; }
  auto load = [&](int s, int tile) {                // tiles past the end re-read the last full tile (never multiplied)
    const int tc = max(min(tile, Tfull - 1), 0);
    const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)tc * 64;
    const char* wb = reinterpret_cast<const char*>(a.w) + (size_t)tc * 64 * a.ldw;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) av[s][rt] = *reinterpret_cast<const float4*>(xb + aoff[rt]);
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[s][j] = *reinterpret_cast<const float2*>(wb + boff[j]);
  };
  auto mac = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const float4 v = av[s][rt];
        const float x = j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
        acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, bv[s][j].x, acc[rt][0], 0, 0, 0);
        acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, bv[s][j].y, acc[rt][1], 0, 0, 0);
      }
    }
  };
  auto gram = [&](int s) {        // its own scheduling region behind mac(s): 4 MFMAs + ~40 VALU on registers still live
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // row tiles g_rt / g_rtp of this stage, picked with 0/1 weights (exact; a ternary chain on the register array
      // is turned into an indexed scratch access by the compiler)
      auto el = [&](int t) { const float4 v = av[s][t]; return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; };
      const float xa = el(0) * ga[0] + el(1) * ga[1] + el(2) * ga[2] + el(3) * ga[3];
      const float xb = el(0) * gb[0] + el(1) * gb[1] + el(2) * gb[2] + el(3) * gb[3];
      accg = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, xb, accg, 0, 0, 0);
      w2 += bv[s][j].x * bv[s][j].x + bv[s][j].y * bv[s][j].y;
    }
  };
  // Ring of DEPTH register stages.  Region i multiplies tile i (stage i % DEPTH) while the loads of tile i + DEPTH - 1
  // are issued into the stage tile i - 1 just left, one load per four MFMAs: DEPTH - 2 whole regions (1024 MFMA cycles
  // each) lie between a load and its use.
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) load(s, q + s * nslot);
  int i = 0;
  for (; i + DEPTH <= nt; i += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      if (GRAM && i + s > 0) {       // the stage about to be refilled holds the tile of the previous region
        gram((s + DEPTH - 1) % DEPTH);
        __builtin_amdgcn_sched_barrier(0);
      }
      load((s + DEPTH - 1) % DEPTH, q + (i + s + DEPTH - 1) * nslot);
      mac(s);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // 4 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // 1 VMEM read
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (GRAM && i > 0) gram(DEPTH - 1);               // the last region's tile
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s)
    if (i + s < nt) {
      mac(s);
      if (GRAM) gram(s);
    }
  if ((a.K & 15) && q == Tfull % nslot) {           // the partial tile: lanes with k >= K clamp their address, A := 0
    const int k0 = Tfull * 16 + kq * 4, ka = min(k0, a.K - 4);
    const float keep = k0 < a.K ? 1.f : 0.f;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const float4 v = *reinterpret_cast<const float4*>(a.x + (long)min(rb + rt * 16 + ln, a.B - 1) * a.ldx + ka);
      av[0][rt] = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[0][j] = *reinterpret_cast<const float2*>(a.w + (long)(ka + j) * a.ldw + c0 + 2 * ln);
    mac(0);
  }
  // C/D map: col = lane & 15 -> output column 2 n + c, row = 4 kq + reg
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float2*>(&red[w][rt * 16 + kq * 4 + r][2 * ln]) = make_float2(acc[rt][0][r], acc[rt][1][r]);
  __syncthreads();
  for (int e = tid; e < 64 * (DF_CW / 4); e += 64 * NW) {
    const int row = e / (DF_CW / 4), c4 = (e % (DF_CW / 4)) * 4;
    if (rb + row >= a.B) continue;
    float4 t = *reinterpret_cast<const float4*>(&red[0][row][c4]);
#pragma unroll
    for (int k = 1; k < NW; ++k) {
      const float4 u = *reinterpret_cast<const float4*>(&red[k][row][c4]);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *reinterpret_cast<float4*>(a.part + ((long)split * a.B + rb + row) * a.E + c0 + c4) = t;
  }
  if (GRAM && blockIdx.y == 0) {
    __syncthreads();
    float* rg = &red[0][0][0];                      // [NW][16][17] Gram tile partials, then [NW] sums of squares
#pragma unroll
    for (int r = 0; r < 4; ++r) rg[(w * 16 + kq * 4 + r) * 17 + ln] = accg[r];
    w2 = tnt_wave_sum(w2);
    if (lane == 0) rg[NW * 16 * 17 + w] = w2;
    __syncthreads();
    if (tid < 256 && ct < 16) {
      const int row = tid >> 4, col = tid & 15;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) t += rg[(k * 16 + row) * 17 + col];
      a.gx_part[((long)split * 64 + g_rt * 16 + row) * 64 + g_rtp * 16 + col] = t;
    }
    if (tid == 0) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) t += rg[NW * 16 * 17 + k];
      a.w2_part[split * nct + ct] = t;
    }
  }
}
}  // namespace

static int32_t dense_fwd_stream_launch(const float* x, const float* w, float* part, float* gx_part, float* w2_part,
                                       int32_t B, int32_t E, int32_t K, int32_t ldx, int32_t ldw, int32_t nsplit,
                                       void* stream) {
  if (B <= 0 || K < 4 || K % 4 != 0 || ldx < K || ldx % 4 != 0) return TNT_BADARG(6);
  if (E <= 0 || E % DF_CW != 0 || ldw < E || ldw % 2 != 0 || nsplit <= 0 || nsplit > 64) return TNT_BADARG(5);
  if (!tnt_aligned16(x) || !tnt_aligned16(w) || !tnt_aligned16(part)) return TNT_BADARG(1);
  if ((long)B * ldx * 4 >= (1L << 32) || 16L * ldw * 4 >= (1L << 32)) return TNT_BADARG(4);      // 32-bit lane offsets
  DfArgs a{x, w, part, B, E, K, ldx, ldw, nsplit, gx_part, w2_part};
  // Ring depth / waves measured on MI355X at 64 x 20000 x 512 (tools/enc_fwd_probe.py): 4 waves x depth 4: 17.3 us per
  // back-to-back launch, depth 5 / 6: 18.0 / 18.5, 8 waves x depth 4 / 5: 18.3 / 19.6; loads served from L1 instead of
  // HBM: 17.3 -- the loop is bound by the 20480 MFMA cycles per wave plus ramp-up, not by memory.
  const dim3 g((E / DF_CW) * nsplit, (B + 63) / 64);
  if (gx_part) hipLaunchKernelGGL((dense_fwd_stream_kernel<4, 4, true>), g, dim3(256), 0, tnt_stream(stream), a);
  else hipLaunchKernelGGL((dense_fwd_stream_kernel<4, 4, false>), g, dim3(256), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_dense_fwd_stream_f32(const float* x, const float* w, float* part, int32_t B, int32_t E,
                                            int32_t K, int32_t ldx, int32_t ldw, int32_t nsplit, void* stream) {
  return dense_fwd_stream_launch(x, w, part, nullptr, nullptr, B, E, K, ldx, ldw, nsplit, stream);
}

extern "C" int32_t tnt_dense_fwd_stream_gram_f32(const float* x, const float* w, float* part, float* gx_part,
                                                 float* w2_part, int32_t B, int32_t E, int32_t K, int32_t ldx,
                                                 int32_t ldw, int32_t nsplit, void* stream) {
  // one Gram tile per column group (16 of them), whole 16-k tiles only, one row block
  if (gx_part == nullptr || w2_part == nullptr || E / DF_CW < 16 || K % 16 != 0 || B > 64) return TNT_BADARG(4);
  return dense_fwd_stream_launch(x, w, part, gx_part, w2_part, B, E, K, ldx, ldw, nsplit, stream);
}

// ---- clip-by-norm input of the dense encoder kernel without its gradient (see dense_fwd_stream_kernel<GRAM>):
//   sum (g + 2 l2 W)^2 = ||X^T D||^2 + 4 l2 <X^T D, W> + 4 l2^2 ||W||^2,   g = X^T D,
//   ||X^T D||^2 = sum_{b,b'} (X X^T)[b,b'] (D D^T)[b,b'],   <X^T D, W> = sum_{b,e} D[b,e] (X W)[b,e],  X W = pre - bias.
// Four workgroups per row b < Bk: 16 entries of row b of D D^T each (E-long dots in double), contracted with row b of
// X X^T (summed over the K splits), the first one adds the row's share of the middle term -> one span-partial slot per
// workgroup (layout of tnt_span_sqnorm_f32).  The last workgroup spreads the ||W||^2 parts over the following slots
// (q = 4 l2^2 w, wq = w) and zeroes the rest.
namespace {
constexpr int GN_Q = 4;        // workgroups per row b: 16 columns b' of the Gram contraction each
// The span norms of every OTHER variable of the step (tnt_span_sqnorm_f32's job) ride in the same launch as extra
// workgroups: both are inputs of the same finalize launch, and a launch less is ~4 us of the captured step.
struct GnSpans {
  const float* theta; const float* grad; const int32_t* span_seg; const int64_t* span_off; const int32_t* span_len;
  const float* seg_l2; float* partial; int nspan;
  // the "lr job" (optional): Adam's step size for the update that follows, see tnt_span_sqnorm_lr_f32 / tnt_adam_fin_f32
  const int64_t* adam_t; const float* lr; float* lr_t; float b1, b2;
  const float* ovr;            // optional: see tnt_span_norm (csrc/tnt_fin.h)
};

__global__ __launch_bounds__(256) void dense_gram_norm_kernel(const float* dpre, const float* pre, const float* bias,
                                                              const float* gx_part, int nsplit, const float* w2_part,
                                                              int nw2, float l2, float* partial, int nslot, int Bk, int E,
                                                              GnSpans sp) {
  __shared__ double sd[256];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x > GN_Q * Bk) {
    // ---- span role: partial[2 s] = sum (g + 2 lambda theta)^2, partial[2 s + 1] = sum theta^2 over span s (optim.hip)
    __shared__ float s0[4], s1[4];
    const int s = blockIdx.x - (GN_Q * Bk + 1);
    if (s >= sp.nspan) return;
    const SpanTab st{sp.span_seg, sp.span_off, sp.span_len, nullptr, sp.seg_l2};
    tnt_span_norm(sp.theta, sp.grad, st, s, sp.partial, sp.ovr, s0, s1);
    return;
  }
  if ((int)blockIdx.x == GN_Q * Bk) {
    if (sp.lr_t != nullptr && tid == 255) sp.lr_t[0] = tnt_adam_lr_t(sp.adam_t, sp.lr, sp.b1, sp.b2);
    for (int k = tid; k < nslot - GN_Q * Bk; k += 256) {
      const float w = k < nw2 ? w2_part[k] : 0.f;
      partial[2 * (GN_Q * Bk + k)] = 4.f * l2 * l2 * w;
      partial[2 * (GN_Q * Bk + k) + 1] = w;
    }
    return;
  }
  const int b = blockIdx.x / GN_Q, qd = blockIdx.x % GN_Q;
  // (D D^T)[b][b'] for this workgroup's 16 b': 16 threads per b', each a contiguous 1/16 of the row in 16-byte loads
  const int bp = qd * 16 + (tid >> 4), part = tid & 15;
  double acc = 0.0;
  if (bp < Bk) {
    const int n4 = E / 64;                                  // float4 per thread (E % 64 == 0)
    const float4* db = reinterpret_cast<const float4*>(dpre + (long)b * E) + part * n4;
    const float4* dp = reinterpret_cast<const float4*>(dpre + (long)bp * E) + part * n4;
#pragma unroll 8
    for (int e = 0; e < n4; ++e) {
      const float4 u = db[e], v = dp[e];
      acc += (double)u.x * (double)v.x + (double)u.y * (double)v.y + (double)u.z * (double)v.z + (double)u.w * (double)v.w;
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 16);      // the 16 parts of one b'
  double s = 0.0;
  if (part == 0 && bp < Bk) {
    double gx = 0.0;
#pragma unroll 8
    for (int sp = 0; sp < nsplit; ++sp) gx += (double)gx_part[((long)sp * 64 + b) * 64 + bp];
    s = gx * acc;
  }
  if (qd == 0) {                                            // middle term: 4 l2 sum_e D[b,e] (X W)[b,e]
    double s2 = 0.0;
#pragma unroll 4
    for (int e = tid; e < E; e += 256) s2 += (double)dpre[(long)b * E + e] * ((double)pre[(long)b * E + e] - (double)bias[e]);
    s += 4.0 * (double)l2 * s2;
  }
  sd[tid] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (tid < k) sd[tid] += sd[tid + k];
    __syncthreads();
  }
  if (tid == 0) { partial[2 * blockIdx.x] = (float)sd[0]; partial[2 * blockIdx.x + 1] = 0.f; }
}
}  // namespace

static int32_t gram_norm_launch(const float* dpre, const float* pre, const float* bias, const float* gx_part, int32_t nsplit,
                                const float* w2_part, int32_t nw2, float l2, float* partial, int32_t nslot, int32_t Bk,
                                int32_t E, const GnSpans& sp, void* stream) {
  if (Bk <= 0 || Bk > 64 || E <= 0 || E % 64 != 0 || nsplit <= 0 || nw2 <= 0 || nslot < GN_Q * Bk + nw2) return TNT_BADARG(9);
  if (!tnt_aligned16(dpre)) return TNT_BADARG(0);
  hipLaunchKernelGGL(dense_gram_norm_kernel, dim3(GN_Q * Bk + 1 + sp.nspan), dim3(256), 0, tnt_stream(stream), dpre, pre, bias,
                     gx_part, nsplit, w2_part, nw2, l2, partial, nslot, Bk, E, sp);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_dense_gram_norm_f32(const float* dpre, const float* pre, const float* bias, const float* gx_part,
                                           int32_t nsplit, const float* w2_part, int32_t nw2, float l2, float* partial,
                                           int32_t nslot, int32_t Bk, int32_t E, void* stream) {
  return gram_norm_launch(dpre, pre, bias, gx_part, nsplit, w2_part, nw2, l2, partial, nslot, Bk, E, GnSpans{}, stream);
}

extern "C" int32_t tnt_dense_gram_norm_spans_f32(const float* dpre, const float* pre, const float* bias,
                                                 const float* gx_part, int32_t nsplit, const float* w2_part, int32_t nw2,
                                                 float l2, float* partial, int32_t nslot, int32_t Bk, int32_t E,
                                                 const float* theta, const float* grad, const int32_t* span_seg,
                                                 const int64_t* span_off, const int32_t* span_len, const float* seg_l2,
                                                 float* span_partial, int32_t nspan, void* stream) {
  if (nspan < 0 || (nspan > 0 && (theta == nullptr || grad == nullptr || span_partial == nullptr))) return TNT_BADARG(19);
  return gram_norm_launch(dpre, pre, bias, gx_part, nsplit, w2_part, nw2, l2, partial, nslot, Bk, E,
                          GnSpans{theta, grad, span_seg, span_off, span_len, seg_l2, span_partial, nspan, nullptr, nullptr, nullptr,
                                  0.f, 0.f, nullptr}, stream);
}

extern "C" int32_t tnt_dense_gram_norm_spans_lr_f32(const float* dpre, const float* pre, const float* bias,
                                                    const float* gx_part, int32_t nsplit, const float* w2_part, int32_t nw2,
                                                    float l2, float* partial, int32_t nslot, int32_t Bk, int32_t E,
                                                    const float* theta, const float* grad, const int32_t* span_seg,
                                                    const int64_t* span_off, const int32_t* span_len, const float* seg_l2,
                                                    float* span_partial, int32_t nspan, const int64_t* adam_t, const float* lr,
                                                    float* lr_t, float beta1, float beta2, const float* sq_override,
                                                    void* stream) {
  if (nspan < 0 || (nspan > 0 && (theta == nullptr || grad == nullptr || span_partial == nullptr))) return TNT_BADARG(19);
  if (adam_t == nullptr || lr == nullptr || lr_t == nullptr) return TNT_BADARG(21);
  return gram_norm_launch(dpre, pre, bias, gx_part, nsplit, w2_part, nw2, l2, partial, nslot, Bk, E,
                          GnSpans{theta, grad, span_seg, span_off, span_len, seg_l2, span_partial, nspan, adam_t, lr, lr_t, beta1,
                                  beta2, sq_override}, stream);
}

namespace {
// ---- input gradient of a stack of per-region Dense layers (the deeper stages of deep_layers.LocallyDense,
// AttemptFour/Model/deep_layers.py:53-59: layer_r(x[:, r, :]) for every region r):
//   dx[b][r][k] = sum_n dpre[b][r][n] * W_r[k][n],   W = [R][Din][Dout] (keras (in, out) kernels, concatenated)
// One workgroup per region: W_r (<= 64 x 64) sits in LDS (odd row stride: the Din rows of a column group fall on
// different banks), every thread owns output elements (b, k) strided over the batch.
__global__ __launch_bounds__(256) void block_dense_dx_kernel(const float* dpre, const float* W, float* dx, int B, int R,
                                                             int Din, int Dout) {
  __shared__ float Ws[64 * 65];
  __shared__ float ds[4][64];
  const int r = blockIdx.x, tid = threadIdx.x;
  for (int e = tid; e < Din * Dout; e += 256) Ws[(e / Dout) * 65 + (e % Dout)] = W[(long)r * Din * Dout + e];
  const int k = tid & 63, bl = tid >> 6;
  for (int b0 = 0; b0 < B; b0 += 4) {
    const int b = b0 + bl;
    __syncthreads();
    if (b < B && k < Dout) ds[bl][k] = dpre[((long)b * R + r) * Dout + k];
    __syncthreads();
    if (b < B && k < Din) {
      float acc = 0.f;
      for (int n = 0; n < Dout; ++n) acc += ds[bl][n] * Ws[k * 65 + n];
      dx[((long)b * R + r) * Din + k] = acc;
    }
  }
}
}  // namespace

extern "C" int32_t tnt_dense_dw_skinny_f32(const float* x, const float* dpre, float* dw, int32_t N, int32_t E,
                                           int32_t Bk, int32_t ldx, void* stream) {
  if (N <= 0 || Bk <= 0 || Bk > 4 * DW_KS || ldx < N) return TNT_BADARG(6);
  if (E <= 0 || E % 16 != 0) return TNT_BADARG(5);
  const int nstrip = (N + DW_MS - 1) / DW_MS;
  DwArgs a{x, dpre, dw, N, E, Bk, ldx};
  // Tiles per wave / grid measured on MI355X at 20000 x 512 x 64 (tools/dw_dbg.py): every combination of 1/2/4 tiles
  // and 128..1250 workgroups lands at 21-24 us (the generic GEMM: 28.4); 4 tiles x 256 workgroups is the best by a hair.
  const int tpw = E > 256 ? 4 : (E > 128 ? 2 : 1), grid = 256;
  const int NT = E / 16;
  dim3 g(nstrip < grid ? nstrip : grid, (NT + 8 * tpw - 1) / (8 * tpw));
  const bool perm = tpw > 1 && E % (128 * tpw) == 0 && tnt_aligned16(dw);      // every wave full: vector row stores
  if (tpw == 1) hipLaunchKernelGGL((dense_dw_skinny_kernel<1, false>), g, dim3(512), 0, tnt_stream(stream), a);
  else if (tpw == 2 && perm) hipLaunchKernelGGL((dense_dw_skinny_kernel<2, true>), g, dim3(512), 0, tnt_stream(stream), a);
  else if (tpw == 2) hipLaunchKernelGGL((dense_dw_skinny_kernel<2, false>), g, dim3(512), 0, tnt_stream(stream), a);
  else if (perm) hipLaunchKernelGGL((dense_dw_skinny_kernel<4, true>), g, dim3(512), 0, tnt_stream(stream), a);
  else hipLaunchKernelGGL((dense_dw_skinny_kernel<4, false>), g, dim3(512), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

static int32_t dw_fused_check(const float* x, const float* dpre, int32_t N, int32_t E, int32_t Bk, int32_t ldx) {
  if (N <= 0 || Bk <= 0 || Bk > 4 * DW_KS || ldx < N) return TNT_BADARG(6);
  if (E <= 0 || E % 512 != 0) return TNT_BADARG(5);       // whole waves of 4 tiles; a strip of 16 rows = E / 512 spans
  if (!tnt_aligned16(x) || !tnt_aligned16(dpre)) return TNT_BADARG(1);
  return 0;
}

extern "C" int32_t tnt_dense_dw_sqnorm_f32(const float* x, const float* dpre, const float* theta, float l2,
                                           float* partial, int32_t nslot, int32_t N, int32_t E, int32_t Bk,
                                           int32_t ldx, void* stream) {
  if (int32_t rc = dw_fused_check(x, dpre, N, E, Bk, ldx)) return rc;
  const int nstrip = (N + DW_MS - 1) / DW_MS, grid = nstrip < 256 ? nstrip : 256;
  if (nslot < grid * (E / 512) || !tnt_aligned16(theta)) return TNT_BADARG(6);
  DwArgs a{};
  a.x = x; a.dpre = dpre; a.N = N; a.E = E; a.Bk = Bk; a.ldx = ldx;
  a.theta = const_cast<float*>(theta); a.partial = partial; a.nslot = nslot; a.lam2 = 2.f * l2;
  hipLaunchKernelGGL((dense_dw_skinny_kernel<4, true, 1>), dim3(grid, E / 512), dim3(512), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_dense_dw_adam_f32(const float* x, const float* dpre, float* theta, float* m, float* v, float l2,
                                         const float* sq, const float* sq_override, const float* lr_t_dev, float beta1,
                                         float beta2, float eps, float clipnorm, const uint32_t* guard, int32_t N,
                                         int32_t E, int32_t Bk, int32_t ldx, void* stream) {
  if (int32_t rc = dw_fused_check(x, dpre, N, E, Bk, ldx)) return rc;
  if (!tnt_aligned16(theta) || !tnt_aligned16(m) || !tnt_aligned16(v) || lr_t_dev == nullptr || sq == nullptr)
    return TNT_BADARG(3);
  const int nstrip = (N + DW_MS - 1) / DW_MS, grid = nstrip < 256 ? nstrip : 256;
  DwArgs a{};
  a.x = x; a.dpre = dpre; a.N = N; a.E = E; a.Bk = Bk; a.ldx = ldx;
  a.theta = theta; a.m = m; a.v = v; a.lam2 = 2.f * l2; a.sq = sq; a.sq_override = sq_override; a.lr_t_dev = lr_t_dev;
  a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.clipnorm = clipnorm; a.guard = guard;
  if (tnt_stream_policy_nt())
    hipLaunchKernelGGL((dense_dw_skinny_kernel<4, true, 2, true>), dim3(grid, E / 512), dim3(512), 0, tnt_stream(stream), a);
  else
    hipLaunchKernelGGL((dense_dw_skinny_kernel<4, true, 2>), dim3(grid, E / 512), dim3(512), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_dense_dw_adam_fin_f32(const float* x, const float* dpre, float* theta, float* m, float* v, float l2,
                                             const float* partial, int32_t k0, int32_t k1, const float* sq_override,
                                             const float* lr_t_dev, float beta1, float beta2, float eps, float clipnorm,
                                             const uint32_t* guard, int32_t N, int32_t E, int32_t Bk, int32_t ldx, void* stream) {
  if (int32_t rc = dw_fused_check(x, dpre, N, E, Bk, ldx)) return rc;
  if (!tnt_aligned16(theta) || !tnt_aligned16(m) || !tnt_aligned16(v) || lr_t_dev == nullptr || partial == nullptr || k0 < 0 || k1 <= k0)
    return TNT_BADARG(3);
  const int nstrip = (N + DW_MS - 1) / DW_MS, grid = nstrip < 256 ? nstrip : 256;
  DwArgs a{};
  a.x = x; a.dpre = dpre; a.N = N; a.E = E; a.Bk = Bk; a.ldx = ldx;
  a.theta = theta; a.m = m; a.v = v; a.lam2 = 2.f * l2; a.sq = nullptr; a.sq_override = sq_override; a.lr_t_dev = lr_t_dev;
  a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.clipnorm = clipnorm; a.guard = guard;
  a.fin_partial = partial; a.fin_k0 = k0; a.fin_k1 = k1;
  if (tnt_stream_policy_nt())
    hipLaunchKernelGGL((dense_dw_skinny_kernel<4, true, 2, true>), dim3(grid, E / 512), dim3(512), 0, tnt_stream(stream), a);
  else
    hipLaunchKernelGGL((dense_dw_skinny_kernel<4, true, 2>), dim3(grid, E / 512), dim3(512), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_block_dense_dx_f32(const float* dpre, const float* W, float* dx, int32_t B, int32_t R,
                                          int32_t Din, int32_t Dout, void* stream) {
  if (B <= 0 || R <= 0 || Din <= 0 || Dout <= 0 || Din > 64 || Dout > 64) return TNT_BADARG(5);
  hipLaunchKernelGGL(block_dense_dx_kernel, dim3(R), dim3(256), 0, tnt_stream(stream), dpre, W, dx, B, R, Din, Dout);
  TNT_LAUNCH_CHECK();
  return 0;
}
