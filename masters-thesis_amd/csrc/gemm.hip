// FP32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled.
//
// Replaces every keras Dense / TimeDistributed(Dense) matmul of the hot path and the
// matmuls tape.gradient derives from them (reference: AttemptFour/Model/layers.py:33,
// attention.py:21-23, lc_NIC.py:140-157,261,386-387, NIC.py:64-69,92-96,143,248-249).
// Exact f32 (fma chain per output element), which is what the 1e-4 logit parity needs.
//
// Tile: BM x BN x 32 per 256-thread workgroup (4 waves as 2x2, each wave owning
// (BM/2)x(BN/2) as 32x32 MFMA tiles).  Both operand tiles live k-major in LDS
// ([k][m], [k][n]) so an MFMA operand fetch is one conflict-free ds_read_b32 per lane;
// operands that are k-contiguous in HBM are transposed on the way in.
// Pipeline: global loads of chunk i+1 are issued before the MFMAs of chunk i (register
// prefetch), the LDS is double-buffered, one barrier per chunk.
// Placement: workgroups are dealt round-robin over the 8 XCDs, each with a private 4 MiB L2.
// The linear workgroup id is remapped so that one XCD owns a contiguous run of output tiles
// ordered m-fastest: an XCD then streams only 1/8 of the B panel (its own column tiles) and
// keeps it L2-resident across all row tiles (PMC: without this the 10 MB vocabulary kernel
// re-streamed from the Infinity Cache once per row tile and the MFMA pipe sat at 45 %).
#include "tnt_common.h"

namespace {

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias; float* pre; float* work;
  int M, N, K, lda, ldb, ldc;
  int act; float slope; int accumulate; int kchunk; int splitk;
};

constexpr int BK = 32;

template <bool VEC>
__device__ __forceinline__ float4 ldg4(const float* p, int valid) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid >= 4 && VEC) {
    v = *reinterpret_cast<const float4*>(p);
  } else {
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
    if (valid > 3) v.w = p[3];
  }
  return v;
}

// Operand tile loader.  KCONTIG: memory is [mn][ld] with k contiguous; else [k][ld] with mn contiguous.
template <int BMN, bool KCONTIG, bool VEC>
struct TileLoader {
  static constexpr int NV = BMN * BK / 4 / 256;   // float4 per thread
  // k-contiguous operands are transposed with 4 ds_write_b32: an odd row stride spreads the
  // (8 k-quads x 4 rows) of a 32-lane group over all 32 banks.  Direct operands need 16-B rows.
  static constexpr int PAD = KCONTIG ? 1 : 4;
  static constexpr int LDS_LD = BMN + PAD;
  float4 r[NV];

  __device__ __forceinline__ void load(const float* base, int ld, int mn0, int mn_lim, int k0, int k_lim, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * 256;
      if (KCONTIG) {
        const int mn = f >> 3, kq = (f & 7) * 4;
        const int gmn = mn0 + mn, gk = k0 + kq;
        int valid = (gmn < mn_lim) ? (k_lim - gk) : 0;
        r[i] = ldg4<VEC>(base + (long)gmn * ld + gk, valid);
      } else {
        const int kk = f / (BMN / 4), mn4 = (f % (BMN / 4)) * 4;
        const int gk = k0 + kk, gmn = mn0 + mn4;
        int valid = (gk < k_lim) ? (mn_lim - gmn) : 0;
        r[i] = ldg4<VEC>(base + (long)gk * ld + gmn, valid);
      }
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * 256;
      if (KCONTIG) {
        const int mn = f >> 3, kq = (f & 7) * 4;
        lds[(kq + 0) * LDS_LD + mn] = r[i].x;
        lds[(kq + 1) * LDS_LD + mn] = r[i].y;
        lds[(kq + 2) * LDS_LD + mn] = r[i].z;
        lds[(kq + 3) * LDS_LD + mn] = r[i].w;
      } else {
        const int kk = f / (BMN / 4), mn4 = (f % (BMN / 4)) * 4;
        *reinterpret_cast<float4*>(&lds[kk * LDS_LD + mn4]) = r[i];
      }
    }
  }
};

template <int BM, int BN, bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  using LA = TileLoader<BM, !TA, VEC>;
  using LB = TileLoader<BN, TB, VEC>;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int ASZ = BK * LA::LDS_LD, BSZ = BK * LB::LDS_LD;
  constexpr int AOFF = (2 * ASZ + 3) & ~3;       // keep the B region 16-byte aligned
  __shared__ __attribute__((aligned(16))) float lds[AOFF + 2 * BSZ];
  float* As = lds;
  float* Bs = lds + AOFF;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile remap (bijective for any tile count); speed only, never correctness
  const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
  const int nwg = MT * NT, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (t % MT) * BM, n0 = (t / MT) * BN;
  const int z = blockIdx.z;
  const int kbeg = z * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  LA la; LB lb;
  if (nk > 0) {
    la.load(g.A, g.lda, m0, g.M, kbeg, kend, tid);
    lb.load(g.B, g.ldb, n0, g.N, kbeg, kend, tid);
    la.store(As, tid);
    lb.store(Bs, tid);
  }
  __syncthreads();

  const int lrow = lane & 31, lk = lane >> 5;
  // per-lane LDS bases: every operand fetch below is base + compile-time offset
  const float* Abase = As + lk * LA::LDS_LD + wm * WM + lrow;
  const float* Bbase = Bs + lk * LB::LDS_LD + wn * WN + lrow;
  for (int i = 0; i < nk; ++i) {
    const int cur = i & 1;
    if (i + 1 < nk) {
      la.load(g.A, g.lda, m0, g.M, kbeg + (i + 1) * BK, kend, tid);
      lb.load(g.B, g.ldb, n0, g.N, kbeg + (i + 1) * BK, kend, tid);
    }
    const float* Ac = Abase + cur * ASZ;
    const float* Bc = Bbase + cur * BSZ;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) av[tm] = Ac[kk * LA::LDS_LD + tm * 32];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bv[tn] = Bc[kk * LB::LDS_LD + tn * 32];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
    }
    if (i + 1 < nk) {
      la.store(As + (cur ^ 1) * ASZ, tid);
      lb.store(Bs + (cur ^ 1) * BSZ, tid);
    }
    __syncthreads();
  }

  // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * WN + tn * 32 + lrow;
      if (col >= g.N) continue;
      const float bcol = (g.bias != nullptr && g.splitk == 1) ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= g.M) continue;
        float v = acc[tm][tn][r];
        if (g.splitk > 1) {
          g.work[((long)z * g.M + row) * g.N + col] = v;
        } else {
          v += bcol;
          const long o = (long)row * g.ldc + col;
          if (g.pre) g.pre[o] = v;
          v = tnt_act(v, g.act, g.slope);
          if (g.accumulate) v += g.C[o];
          g.C[o] = v;
        }
      }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g) {
  const long total = (long)g.M * g.N;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int row = (int)(e / g.N), col = (int)(e % g.N);
    float v = 0.f;
#pragma unroll 8
    for (int z = 0; z < g.splitk; ++z) v += g.work[(long)z * total + e];
    if (g.bias) v += g.bias[col];
    const long o = (long)row * g.ldc + col;
    if (g.pre) g.pre[o] = v;
    v = tnt_act(v, g.act, g.slope);
    if (g.accumulate) v += g.C[o];
    g.C[o] = v;
  }
}

template <int BM, int BN, bool TA, bool TB>
int32_t launch_cfg(const GemmArgs& g, bool vec, hipStream_t s) {
  dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM), 1, g.splitk);
  if (vec)
    hipLaunchKernelGGL((gemm_kernel<BM, BN, TA, TB, true>), grid, dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_kernel<BM, BN, TA, TB, false>), grid, dim3(256), 0, s, g);
  TNT_LAUNCH_CHECK();
  return 0;
}

template <bool TA, bool TB>
int32_t launch_layout(const GemmArgs& g, int bm, int bn, bool vec, hipStream_t s) {
  if (bm == 128 && bn == 128) return launch_cfg<128, 128, TA, TB>(g, vec, s);
  if (bm == 64 && bn == 128) return launch_cfg<64, 128, TA, TB>(g, vec, s);
  if (bm == 128 && bn == 64) return launch_cfg<128, 64, TA, TB>(g, vec, s);
  return launch_cfg<64, 64, TA, TB>(g, vec, s);
}

int32_t gemm_dispatch(const float* A, const float* B, float* C, const float* bias, float* pre, int32_t M, int32_t N,
                      int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t transA, int32_t transB, int32_t act,
                      float slope, int32_t accumulate, int32_t splitk, float* work, int force_bm, int force_bn,
                      void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return TNT_BADARG(6);
  if (transA && transB) return TNT_BADARG(12);
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && work == nullptr) return TNT_BADARG(18);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.pre = pre; g.work = work;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.act = act; g.slope = slope; g.accumulate = accumulate;
  int kchunk = (K + splitk - 1) / splitk;
  kchunk = (kchunk + BK - 1) / BK * BK;
  splitk = (K + kchunk - 1) / kchunk;
  g.kchunk = kchunk; g.splitk = splitk;

  // tile choice, calibrated with tools/gemm_bench.py / tools/gemm_scan.py on MI355X (profiles/r01_gemm_*):
  // at the hot-path sizes (<= ~2k tiles) the 64x64 tile wins on every shape because it fills the 256 CUs
  // (4 workgroups/CU by LDS) and quantises best; the larger tiles only pay off in steady state.
  const int cand[4][2] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}};
  const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64) * splitk;
  int best = 3;
  if (tiles64 >= 16384) best = 0;
  else if (tiles64 >= 8192) best = (M >= N) ? 2 : 1;
  for (int c = 0; c < 4; ++c)
    if (cand[c][0] == force_bm && cand[c][1] == force_bn) best = c;
  const bool vec = tnt_aligned16(A) && tnt_aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
  hipStream_t s = tnt_stream(stream);
  int32_t rc;
  if (!transA && !transB) rc = launch_layout<false, false>(g, cand[best][0], cand[best][1], vec, s);
  else if (!transA && transB) rc = launch_layout<false, true>(g, cand[best][0], cand[best][1], vec, s);
  else rc = launch_layout<true, false>(g, cand[best][0], cand[best][1], vec, s);
  if (rc) return rc;
  if (splitk > 1) {
    const long total = (long)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, g);
    TNT_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

extern "C" int32_t tnt_gemm_f32(const float* A, const float* B, float* C, const float* bias, float* pre,
                                int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc,
                                int32_t transA, int32_t transB, int32_t act, float slope,
                                int32_t accumulate, int32_t splitk, float* work, void* stream) {
  return gemm_dispatch(A, B, C, bias, pre, M, N, K, lda, ldb, ldc, transA, transB, act, slope, accumulate, splitk,
                       work, 0, 0, stream);
}

// tuning entry point: same as tnt_gemm_f32 with the workgroup tile forced to bm x bn
// (bm, bn in {64,128}); used by tools/gemm_bench.py to calibrate the tile heuristic.
extern "C" int32_t tnt_gemm_f32_tile(const float* A, const float* B, float* C, const float* bias, float* pre,
                                     int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc,
                                     int32_t transA, int32_t transB, int32_t act, float slope,
                                     int32_t accumulate, int32_t splitk, float* work, int32_t bm, int32_t bn,
                                     void* stream) {
  return gemm_dispatch(A, B, C, bias, pre, M, N, K, lda, ldb, ldc, transA, transB, act, slope, accumulate, splitk,
                       work, bm, bn, stream);
}
