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

  // The bias is fetched up front: a load still pending in the epilogue makes the compiler put
  // s_waitcnt vmcnt(0) in front of every guarded store there, which serialises the stores.
  const int lrow = lane & 31, lk = lane >> 5;
  float bcol[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * WN + tn * 32 + lrow;
    bcol[tn] = (g.bias != nullptr && g.splitk == 1 && col < g.N) ? g.bias[col] : 0.f;
  }

  LA la; LB lb;
  if (nk > 0) {
    la.load(g.A, g.lda, m0, g.M, kbeg, kend, tid);
    lb.load(g.B, g.ldb, n0, g.N, kbeg, kend, tid);
    la.store(As, tid);
    lb.store(Bs, tid);
  }
  __syncthreads();

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

  // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // The mode is workgroup-uniform and chosen outside the element loops, so each mode is a run of stores.
  const bool plain = g.pre == nullptr && !g.accumulate;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * WN + tn * 32 + lrow;
      if (col >= g.N) continue;
      const int row0 = m0 + wm * WM + tm * 32 + 4 * lk;
      if (g.splitk > 1) {
        float* W = g.work + ((long)z * g.M + row0) * g.N + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (row0 + dr < g.M) W[(long)dr * g.N] = acc[tm][tn][r];
        }
      } else if (plain) {
        float* Cp = g.C + (long)row0 * g.ldc + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (row0 + dr < g.M) Cp[(long)dr * g.ldc] = tnt_act(acc[tm][tn][r] + bcol[tn], g.act, g.slope);
        }
      } else {
        const long o0 = (long)row0 * g.ldc + col;
        float cold[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          cold[r] = (g.accumulate && row0 + dr < g.M) ? g.C[o0 + (long)dr * g.ldc] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (row0 + dr >= g.M) continue;
          const long o = o0 + (long)dr * g.ldc;
          float v = acc[tm][tn][r] + bcol[tn];
          if (g.pre) g.pre[o] = v;
          g.C[o] = tnt_act(v, g.act, g.slope) + cold[r];
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

// ---------------------------------------------------------------------------------------------------------
// One-round kernel.  The 64x64 tiles above quantise badly on the vocabulary head: 1185 tiles over 1024 resident
// workgroup slots = a second, nearly empty round.  Here the workgroup tile is sized so that the whole output is
// ONE round of at most 256 workgroups (one per CU, 16 waves): 160x128 covers logits[960x5001] in 240 tiles.
// 16x16x4 FP32 MFMAs so that tile edges only need to be multiples of 16; each wave owns 80x16 (5 MFMA tiles).
//   A (k-contiguous rows) sits in LDS as [m][k] rows of BK+2 floats: the 16 rows x 2 k-pairs of a 16-lane group
//   fall on 32 distinct banks and one 8-byte read feeds two MFMAs (k = 8s+2g and 8s+2g+1, g = lane>>4);
//   B (n-contiguous rows) sits as [k][n] with 2*LD = 16 mod 32, so lanes 0-15 (row k) and 16-31 (row k+2) of a
//   32-lane group use disjoint banks.  Both operands use the same k permutation, so the sum is the same set
//   of products in a different order from the 32x32x2 kernel (still one f32 fma chain per element).
// Measured on the head (tools/probe/gemm1r_probe.hip, profiles/r01_gemm_design_probe.txt): 54.9 us vs 61.4 us.
template <int BMN, int NT, bool KC>
struct Tile1r {
  static constexpr int QPR = BK / 4;
  static constexpr int NF4 = BMN * BK / 4;
  static constexpr int NV = (NF4 + NT - 1) / NT;
  static constexpr int LD = KC ? (BK + 2) : (BMN + ((40 - BMN % 32) % 32));
  static constexpr int SZ = ((KC ? BMN * LD : BK * LD) + 3) & ~3;
  float4 r[NV];
  __device__ __forceinline__ void load(const float* base, int ld, int mn0, int mn_lim, int k0, int k_lim, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (NF4 % NT != 0 && f >= NF4) continue;
      if (KC) {
        const int mn = f / QPR, kq = (f % QPR) * 4;
        const int valid = (mn0 + mn < mn_lim) ? (k_lim - (k0 + kq)) : 0;
        r[i] = ldg4<true>(base + (long)(mn0 + mn) * ld + k0 + kq, valid);
      } else {
        const int kk = f / (BMN / 4), mn4 = (f % (BMN / 4)) * 4;
        const int valid = (k0 + kk < k_lim) ? (mn_lim - (mn0 + mn4)) : 0;
        r[i] = ldg4<true>(base + (long)(k0 + kk) * ld + mn0 + mn4, valid);
      }
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      if (NF4 % NT != 0 && f >= NF4) continue;
      if (KC) {
        const int mn = f / QPR, kq = (f % QPR) * 4;
        *reinterpret_cast<float2*>(&lds[mn * LD + kq]) = make_float2(r[i].x, r[i].y);
        *reinterpret_cast<float2*>(&lds[mn * LD + kq + 2]) = make_float2(r[i].z, r[i].w);
      } else {
        const int kk = f / (BMN / 4), mn4 = (f % (BMN / 4)) * 4;
        *reinterpret_cast<float4*>(&lds[kk * LD + mn4]) = r[i];
      }
    }
  }
  static __device__ __forceinline__ float2 fetch(const float* lds, int mn, int s, int g) {
    if (KC) return *reinterpret_cast<const float2*>(&lds[mn * LD + 8 * s + 2 * g]);
    return make_float2(lds[(8 * s + 2 * g) * LD + mn], lds[(8 * s + 2 * g + 1) * LD + mn]);
  }
};

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm1r_kernel(GemmArgs g) {
  constexpr int NT = 64 * WGM * WGN;
  using OA = Tile1r<BM, NT, A_KC>;
  using OB = Tile1r<BN, NT, B_KC>;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  static_assert(WM % 16 == 0 && WN % 16 == 0, "wave tile must be whole 16x16 MFMA tiles");
  extern __shared__ __attribute__((aligned(16))) float lds1r[];
  float* As = lds1r;
  float* Bs = lds1r + 2 * OA::SZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int l16 = lane & 15, gq = lane >> 4;
  const int MT = (g.M + BM - 1) / BM, NTl = (g.N + BN - 1) / BN;
  const int nwg = MT * NTl, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;       // same XCD-contiguous remap as gemm_kernel
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (t % MT) * BM, n0 = (t / MT) * BN;
  const int nk = (g.K + BK - 1) / BK;

  floatx4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  float bcol[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * WN + tn * 16 + l16;
    bcol[tn] = (g.bias != nullptr && col < g.N) ? g.bias[col] : 0.f;
  }

  OA oa; OB ob;
  oa.load(g.A, g.lda, m0, g.M, 0, g.K, tid);
  ob.load(g.B, g.ldb, n0, g.N, 0, g.K, tid);
  oa.store(As, tid);
  ob.store(Bs, tid);
  __syncthreads();
  for (int i = 0; i < nk; ++i) {
    const int cur = i & 1;
    if (i + 1 < nk) {
      oa.load(g.A, g.lda, m0, g.M, (i + 1) * BK, g.K, tid);
      ob.load(g.B, g.ldb, n0, g.N, (i + 1) * BK, g.K, tid);
    }
    const float* Ac = As + cur * OA::SZ;
    const float* Bc = Bs + cur * OB::SZ;
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      float2 av[TM], bv[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) av[tm] = OA::fetch(Ac, wm * WM + tm * 16 + l16, s, gq);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bv[tn] = OB::fetch(Bc, wn * WN + tn * 16 + l16, s, gq);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm].x, bv[tn].x, acc[tm][tn], 0, 0, 0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm].y, bv[tn].y, acc[tm][tn], 0, 0, 0);
    }
    if (i + 1 < nk) {
      oa.store(As + (cur ^ 1) * OA::SZ, tid);
      ob.store(Bs + (cur ^ 1) * OB::SZ, tid);
    }
    __syncthreads();
  }

  // epilogue: C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
  const bool plain = g.pre == nullptr && !g.accumulate;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * WN + tn * 16 + l16;
    if (col >= g.N) continue;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row0 = m0 + wm * WM + tm * 16 + 4 * gq;
      const long o0 = (long)row0 * g.ldc + col;
      if (plain) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (row0 + r < g.M) g.C[o0 + (long)r * g.ldc] = tnt_act(acc[tm][tn][r] + bcol[tn], g.act, g.slope);
      } else {
        float cold[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) cold[r] = (g.accumulate && row0 + r < g.M) ? g.C[o0 + (long)r * g.ldc] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (row0 + r >= g.M) continue;
          const long o = o0 + (long)r * g.ldc;
          const float v = acc[tm][tn][r] + bcol[tn];
          if (g.pre) g.pre[o] = v;
          g.C[o] = tnt_act(v, g.act, g.slope) + cold[r];
        }
      }
    }
  }
}

constexpr int R1_BM = 160, R1_BN = 128, R1_WGM = 2, R1_WGN = 8;

// NN only (A rows k-contiguous, B rows n-contiguous): the gradient GEMMs of the same layers have no epilogue
// and go through the vendor library (csrc/blas.hip).
int32_t launch_one_round(const GemmArgs& g, hipStream_t s) {
  using OA = Tile1r<R1_BM, 64 * R1_WGM * R1_WGN, true>;
  using OB = Tile1r<R1_BN, 64 * R1_WGM * R1_WGN, false>;
  constexpr int shmem = (2 * OA::SZ + 2 * OB::SZ) * (int)sizeof(float);
  auto kern = gemm1r_kernel<R1_BM, R1_BN, R1_WGM, R1_WGN, true, false>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, shmem) != hipSuccess)
      return TNT_BADARG(90);
    attr_set = true;
  }
  const int tiles = ((g.M + R1_BM - 1) / R1_BM) * ((g.N + R1_BN - 1) / R1_BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * R1_WGM * R1_WGN), shmem, s, g);
  TNT_LAUNCH_CHECK();
  return 0;
}

// One round pays when the 64x64 grid would need a second, mostly empty round and the 160x128 grid fills most CUs.
bool one_round_fits(int M, int N, int K) {
  const long tiles = (long)((M + R1_BM - 1) / R1_BM) * ((N + R1_BN - 1) / R1_BN);
  const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64);
  if (tiles > 256 || tiles < 208 || K < 128) return false;
  const double useful = (double)M * N / ((double)tiles * R1_BM * R1_BN);
  return useful >= 0.9 && tiles64 > 1024;
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
  if (force_bm == R1_BM && force_bn == R1_BN) {      // tests / tuning: the one-round kernel on any NN shape
    if (transA || transB || splitk != 1 || !vec) return TNT_BADARG(91);
    return launch_one_round(g, s);
  }
  if (!transA && !transB && splitk == 1 && vec && force_bm == 0 && one_round_fits(M, N, K)) return launch_one_round(g, s);
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
// (bm, bn in {64,128}), or (160, 128) = the one-round kernel (NN, splitk 1, 16-byte aligned operands only);
// used by tools/gemm_bench.py to calibrate the heuristics and by the tests to reach both kernels on any shape.
extern "C" int32_t tnt_gemm_f32_tile(const float* A, const float* B, float* C, const float* bias, float* pre,
                                     int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc,
                                     int32_t transA, int32_t transB, int32_t act, float slope,
                                     int32_t accumulate, int32_t splitk, float* work, int32_t bm, int32_t bn,
                                     void* stream) {
  return gemm_dispatch(A, B, C, bias, pre, M, N, K, lda, ldb, ldc, transA, transB, act, slope, accumulate, splitk,
                       work, bm, bn, stream);
}
