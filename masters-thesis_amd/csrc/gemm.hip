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
  // one-round family only: a second product sharing B (blockIdx.y == 1: C2 = op(A2) op(B)), and the column sums of B
  // over k (TN weight gradients: the bias gradient rides along), written by the workgroups of the first row of tiles
  const float* A2; float* C2; float* colsum;
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
  int vld[NV];
  // Branch-free: every lane issues one 16-byte load (from `base` when its slot is out of range) and masks afterwards,
  // so the compiler can hoist and pipeline the loads (guarded loads compile to exec-mask branches with waits inside).
  // A float4 that starts inside the logical edge may run into the row's padding: rows are 16-byte aligned with ld % 4 == 0.
  __device__ __forceinline__ void load(const float* base, int ld, int mn0, int mn_lim, int k0, int k_lim, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      if (NF4 % NT != 0 && f >= NF4) { r[i] = make_float4(0.f, 0.f, 0.f, 0.f); vld[i] = 0; continue; }
      int valid;
      const float* p;
      if (KC) {
        const int mn = f / QPR, kq = (f % QPR) * 4;
        valid = (mn0 + mn < mn_lim) ? (k_lim - (k0 + kq)) : 0;
        p = base + (long)(mn0 + mn) * ld + k0 + kq;
      } else {
        const int kk = f / (BMN / 4), mn4 = (f % (BMN / 4)) * 4;
        valid = (k0 + kk < k_lim) ? (mn_lim - (mn0 + mn4)) : 0;
        p = base + (long)(k0 + kk) * ld + mn0 + mn4;
      }
      r[i] = *reinterpret_cast<const float4*>(valid > 0 ? p : base);      // masked when it is consumed (store / accumulate):
      vld[i] = valid;                                                      // a select here would wait for the load at once
    }
  }
  __device__ __forceinline__ float4 masked(int i) const {
    const int v = vld[i];
    return make_float4(v > 0 ? r[i].x : 0.f, v > 1 ? r[i].y : 0.f, v > 2 ? r[i].z : 0.f, v > 3 ? r[i].w : 0.f);
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      if (NF4 % NT != 0 && f >= NF4) continue;
      const float4 v = masked(i);
      if (KC) {
        const int mn = f / QPR, kq = (f % QPR) * 4;
        *reinterpret_cast<float2*>(&lds[mn * LD + kq]) = make_float2(v.x, v.y);
        *reinterpret_cast<float2*>(&lds[mn * LD + kq + 2]) = make_float2(v.z, v.w);
      } else {
        const int kk = f / (BMN / 4), mn4 = (f % (BMN / 4)) * 4;
        *reinterpret_cast<float4*>(&lds[kk * LD + mn4]) = v;
      }
    }
  }
  // running sum of everything this thread has loaded (non-KC operand: one float4 column group per register)
  __device__ __forceinline__ void accumulate(float4* acc) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) { const float4 v = masked(i); acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w; }
  }
  static __device__ __forceinline__ float2 fetch(const float* lds, int mn, int s, int g) {
    if (KC) return *reinterpret_cast<const float2*>(&lds[mn * LD + 8 * s + 2 * g]);
    return make_float2(lds[(8 * s + 2 * g) * LD + mn], lds[(8 * s + 2 * g + 1) * LD + mn]);
  }
};

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC, int PD>
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
  const bool second = blockIdx.y != 0;
  const float* Ag = second ? g.A2 : g.A;
  float* Cg = second ? g.C2 : g.C;
  const bool do_colsum = !B_KC && g.colsum != nullptr && m0 == 0 && !second;      // workgroup-uniform
  float4 csum[OB::NV];
#pragma unroll
  for (int i = 0; i < OB::NV; ++i) csum[i] = make_float4(0.f, 0.f, 0.f, 0.f);

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

  // Register prefetch ring of PD stages: LDS holds tile i while registers hold tiles i+1 .. i+PD, so a global load has
  // PD iterations of MFMAs to land (one workgroup per CU: nothing else hides the latency).  Tile t sits in stage (t-1) % PD.
  OA oa[PD]; OB ob[PD];
  oa[0].load(Ag, g.lda, m0, g.M, 0, g.K, tid);
  ob[0].load(g.B, g.ldb, n0, g.N, 0, g.K, tid);
  oa[0].store(As, tid);
  ob[0].store(Bs, tid);
  if (do_colsum) ob[0].accumulate(csum);
#pragma unroll
  for (int p = 0; p < PD; ++p)
    if (p + 1 < nk) {
      oa[p].load(Ag, g.lda, m0, g.M, (p + 1) * BK, g.K, tid);
      ob[p].load(g.B, g.ldb, n0, g.N, (p + 1) * BK, g.K, tid);
    }
  __syncthreads();
  for (int i0 = 0; i0 < nk; i0 += PD) {
#pragma unroll
    for (int j = 0; j < PD; ++j) {
      const int i = i0 + j;
      if (i >= nk) break;
      const int cur = i & 1;
      const float* Ac = As + cur * OA::SZ;
      const float* Bc = Bs + cur * OB::SZ;
      // operand fragments double-buffered in registers: the LDS reads of k-group s+1 are issued before the MFMAs of group s
      float2 av[2][TM], bv[2][TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) av[0][tm] = OA::fetch(Ac, wm * WM + tm * 16 + l16, 0, gq);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bv[0][tn] = OB::fetch(Bc, wn * WN + tn * 16 + l16, 0, gq);
#pragma unroll
      for (int s = 0; s < BK / 8; ++s) {
        const int c = s & 1, n = c ^ 1;
        if (s + 1 < BK / 8) {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) av[n][tm] = OA::fetch(Ac, wm * WM + tm * 16 + l16, s + 1, gq);
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) bv[n][tn] = OB::fetch(Bc, wn * WN + tn * 16 + l16, s + 1, gq);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][tm].x, bv[c][tn].x, acc[tm][tn], 0, 0, 0);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][tm].y, bv[c][tn].y, acc[tm][tn], 0, 0, 0);
      }
      if (i + 1 < nk) {
        oa[j].store(As + (cur ^ 1) * OA::SZ, tid);
        ob[j].store(Bs + (cur ^ 1) * OB::SZ, tid);
        if (do_colsum) ob[j].accumulate(csum);
      }
      if (i + 1 + PD < nk) {
        oa[j].load(Ag, g.lda, m0, g.M, (i + 1 + PD) * BK, g.K, tid);
        ob[j].load(g.B, g.ldb, n0, g.N, (i + 1 + PD) * BK, g.K, tid);
      }
      __syncthreads();
    }
  }
  if (do_colsum) {
    // each thread summed its float4 column group over the k rows it loaded (rows f / (BN/4) + multiples of the thread
    // stride); combine the BK partial rows through the operand area, which is free now (fixed order: deterministic)
    float* scr = lds1r;
#pragma unroll
    for (int i = 0; i < OB::NV; ++i) {
      const int f = tid + i * NT;
      if (OB::NF4 % NT != 0 && f >= OB::NF4) continue;
      *reinterpret_cast<float4*>(&scr[(f / (BN / 4)) * BN + (f % (BN / 4)) * 4]) = csum[i];
    }
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
      float v = 0.f;
#pragma unroll 8
      for (int r = 0; r < BK; ++r) v += scr[r * BN + c];
      if (n0 + c < g.N) g.colsum[n0 + c] = v;
    }
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
          if (row0 + r < g.M) Cg[o0 + (long)r * g.ldc] = tnt_act(acc[tm][tn][r] + bcol[tn], g.act, g.slope);
      } else {
        float cold[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) cold[r] = (g.accumulate && row0 + r < g.M) ? Cg[o0 + (long)r * g.ldc] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (row0 + r >= g.M) continue;
          const long o = o0 + (long)r * g.ldc;
          const float v = acc[tm][tn][r] + bcol[tn];
          if (g.pre) g.pre[o] = v;
          Cg[o] = tnt_act(v, g.act, g.slope) + cold[r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// gemm2: the one-round tile family with a straight-line, explicitly software-pipelined main loop.
// What the measurements on gemm1r showed (tools/gemm_cfg_scan.py, .s inspection): with ONE workgroup per CU nothing but
// the instruction stream itself hides latency, and hipcc (a) sinks every LDS operand read down to its first use
// ("ds_read; s_waitcnt lgkmcnt(0); 4 MFMAs" -- ~100 exposed cycles per 128 of matrix work with one wave per SIMD) and
// (b) turns guarded global loads into exec-mask branches whose join points force s_waitcnt vmcnt(0).
// Here: every load is unconditional (clamped address, masked when stored to LDS), the loop body is ONE basic block
// (two k-tiles per trip, two named register stages: the loads of tile i+2 are issued before the MFMAs of tile i and
// consumed after those of tile i+1), operand fragments are double-buffered in registers and
// __builtin_amdgcn_sched_group_barrier pins "LDS reads of k-group s+1 between the MFMAs of group s".
template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm2_kernel(GemmArgs g) {
  constexpr int NT = 64 * WGM * WGN;
  using OA = Tile1r<BM, NT, A_KC>;
  using OB = Tile1r<BN, NT, B_KC>;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  static_assert(WM % 16 == 0 && WN % 16 == 0, "wave tile must be whole 16x16 MFMA tiles");
  extern __shared__ __attribute__((aligned(16))) float lds2[];
  float* As = lds2;
  float* Bs = lds2 + 2 * OA::SZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int l16 = lane & 15, gq = lane >> 4;
  const int MT = (g.M + BM - 1) / BM, NTl = (g.N + BN - 1) / BN;
  const int nwg = MT * NTl, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;       // same XCD-contiguous remap as gemm_kernel
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (t % MT) * BM, n0 = (t / MT) * BN;
  const int nk = (g.K + BK - 1) / BK;
  const bool second = blockIdx.y != 0;
  const float* Ag = second ? g.A2 : g.A;
  float* Cg = second ? g.C2 : g.C;
  const bool do_colsum = !B_KC && g.colsum != nullptr && m0 == 0 && !second;      // workgroup-uniform
  float4 csum[OB::NV];
#pragma unroll
  for (int i = 0; i < OB::NV; ++i) csum[i] = make_float4(0.f, 0.f, 0.f, 0.f);
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
  const int arow = wm * WM + l16, bcolw = wn * WN + l16;
  float2 av[2][TM], bv[2][TN];
  auto fetch = [&](const float* Ac, const float* Bc, int s, int buf) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) av[buf][tm] = OA::fetch(Ac, arow + tm * 16, s, gq);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bv[buf][tn] = OB::fetch(Bc, bcolw + tn * 16, s, gq);
  };
  // the MFMAs of one k-tile held in LDS buffer `cur`; fragments of k-group 0 are already in av[0] / bv[0]
  auto mma_tile = [&](int cur) {
    const float* Ac = As + cur * OA::SZ;
    const float* Bc = Bs + cur * OB::SZ;
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      const int c = s & 1, n = c ^ 1;
      if (s + 1 < BK / 8) fetch(Ac, Bc, s + 1, n);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][tm].x, bv[c][tn].x, acc[tm][tn], 0, 0, 0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][tm].y, bv[c][tn].y, acc[tm][tn], 0, 0, 0);
      // pin the interleave: the (TM + TN) LDS reads of the next k-group go between the first MFMAs of this one
      if (s + 1 < BK / 8) {
#pragma unroll
        for (int q = 0; q < TM + TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN - (TM + TN), 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);
      }
    }
  };
  OA ra0, ra1; OB rb0, rb1;
  // prologue: tile 0 -> LDS[0], tile 1 -> stage 0
  ra0.load(Ag, g.lda, m0, g.M, 0, g.K, tid);
  rb0.load(g.B, g.ldb, n0, g.N, 0, g.K, tid);
  ra0.store(As, tid);
  rb0.store(Bs, tid);
  if (do_colsum) rb0.accumulate(csum);
  ra0.load(Ag, g.lda, m0, g.M, BK, g.K, tid);
  rb0.load(g.B, g.ldb, n0, g.N, BK, g.K, tid);
  __syncthreads();
  fetch(As, Bs, 0, 0);
  // one k-tile: issue the loads of tile i+2 into stage `ln`, multiply tile i, move tile i+1 (stage `st`) to the other
  // LDS buffer.  Loads past K read `base` and are masked to zero, so the body needs no branches.
#define TNT_GEMM2_STEP(i, st_a, st_b, ln_a, ln_b)                                               \
  {                                                                                              \
    const int cur_ = (i) & 1;                                                                    \
    ln_a.load(Ag, g.lda, m0, g.M, ((i) + 2) * BK, g.K, tid);                                     \
    ln_b.load(g.B, g.ldb, n0, g.N, ((i) + 2) * BK, g.K, tid);                                    \
    mma_tile(cur_);                                                                              \
    st_a.store(As + (cur_ ^ 1) * OA::SZ, tid);                                                   \
    st_b.store(Bs + (cur_ ^ 1) * OB::SZ, tid);                                                   \
    if (do_colsum) st_b.accumulate(csum);                                                        \
    __syncthreads();                                                                             \
    fetch(As + (cur_ ^ 1) * OA::SZ, Bs + (cur_ ^ 1) * OB::SZ, 0, 0);                             \
  }
  int i = 0;
  for (; i + 1 < nk; i += 2) {
    TNT_GEMM2_STEP(i, ra0, rb0, ra1, rb1)
    TNT_GEMM2_STEP(i + 1, ra1, rb1, ra0, rb0)
  }
  if (i < nk) TNT_GEMM2_STEP(i, ra0, rb0, ra1, rb1)
#undef TNT_GEMM2_STEP
  if (do_colsum) {
    // the stages past K held zeros; combine the BK partial rows through the operand area (fixed order: deterministic)
    __syncthreads();
    float* scr = lds2;
#pragma unroll
    for (int q = 0; q < OB::NV; ++q) {
      const int f = tid + q * NT;
      if (OB::NF4 % NT != 0 && f >= OB::NF4) continue;
      *reinterpret_cast<float4*>(&scr[(f / (BN / 4)) * BN + (f % (BN / 4)) * 4]) = csum[q];
    }
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
      float v = 0.f;
#pragma unroll 8
      for (int r = 0; r < BK; ++r) v += scr[r * BN + c];
      if (n0 + c < g.N) g.colsum[n0 + c] = v;
    }
  }
  // epilogue: C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * WN + tn * 16 + l16;
    if (col >= g.N) continue;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row0 = m0 + wm * WM + tm * 16 + 4 * gq;
      const long o0 = (long)row0 * g.ldc + col;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (row0 + r < g.M) Cg[o0 + (long)r * g.ldc] = acc[tm][tn][r] + bcol[tn];
    }
  }
}

constexpr int R1_BM = 160, R1_BN = 128;

// ---- one-round configurations (workgroup tile x wave grid); every layout of each is instantiated
template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC, int PD>
int32_t launch_1r_inst(const GemmArgs& g, hipStream_t s) {
  constexpr int NT = 64 * WGM * WGN;
  using OA = Tile1r<BM, NT, A_KC>;
  using OB = Tile1r<BN, NT, B_KC>;
  constexpr int shmem = (2 * OA::SZ + 2 * OB::SZ) * (int)sizeof(float);
  static_assert(BK * BN * (int)sizeof(float) <= shmem, "column-sum scratch must fit the operand area");
  auto kern = gemm1r_kernel<BM, BN, WGM, WGN, A_KC, B_KC, PD>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, shmem) != hipSuccess)
      return TNT_BADARG(90);
    attr_set = true;
  }
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles, g.A2 ? 2 : 1), dim3(NT), shmem, s, g);
  TNT_LAUNCH_CHECK();
  return 0;
}
template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC>
int32_t launch_2_inst(const GemmArgs& g, hipStream_t s) {
  constexpr int NT = 64 * WGM * WGN;
  using OA = Tile1r<BM, NT, A_KC>;
  using OB = Tile1r<BN, NT, B_KC>;
  constexpr int shmem = (2 * OA::SZ + 2 * OB::SZ) * (int)sizeof(float);
  static_assert(BK * BN * (int)sizeof(float) <= shmem, "column-sum scratch must fit the operand area");
  auto kern = gemm2_kernel<BM, BN, WGM, WGN, A_KC, B_KC>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, shmem) != hipSuccess)
      return TNT_BADARG(90);
    attr_set = true;
  }
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles, g.A2 ? 2 : 1), dim3(NT), shmem, s, g);
  TNT_LAUNCH_CHECK();
  return 0;
}
template <int BM, int BN, int WGM, int WGN>
int32_t launch_2_cfg(const GemmArgs& g, bool tA, bool tB, hipStream_t s) {
  if (!tA && !tB) return launch_2_inst<BM, BN, WGM, WGN, true, false>(g, s);
  if (tA && !tB) return launch_2_inst<BM, BN, WGM, WGN, false, false>(g, s);
  if (!tA && tB) return launch_2_inst<BM, BN, WGM, WGN, true, true>(g, s);
  return TNT_BADARG(12);
}

template <int BM, int BN, int WGM, int WGN, int PD>
int32_t launch_1r_cfg(const GemmArgs& g, bool tA, bool tB, hipStream_t s) {
  if (!tA && !tB) return launch_1r_inst<BM, BN, WGM, WGN, true, false, PD>(g, s);
  if (tA && !tB) return launch_1r_inst<BM, BN, WGM, WGN, false, false, PD>(g, s);
  if (!tA && tB) return launch_1r_inst<BM, BN, WGM, WGN, true, true, PD>(g, s);
  return TNT_BADARG(12);
}
// configuration ids of tnt_gemm_fused_f32 (cfg argument / tools/gemm_cfg_scan.py)
int32_t launch_1r(int cfg, const GemmArgs& g, bool tA, bool tB, hipStream_t s) {
  switch (cfg) {
    case 1: return launch_1r_cfg<160, 128, 2, 8, 1>(g, tA, tB, s);      // gemm1r, 16 waves, wave tile 80x16 (head forward)
    case 9: return launch_1r_cfg<64, 64, 4, 2, 3>(g, tA, tB, s);        // gemm1r, 8 waves, 3-stage register prefetch ring
    // gemm2 (straight-line pipelined loop)
    case 21: return launch_2_cfg<64, 64, 2, 2>(g, tA, tB, s);           //  4 waves, 32x32
    case 22: return launch_2_cfg<64, 64, 4, 2>(g, tA, tB, s);           //  8 waves, 16x32
    case 24: return launch_2_cfg<128, 64, 4, 2>(g, tA, tB, s);          //  8 waves, 32x32
    case 25: return launch_2_cfg<64, 32, 2, 2>(g, tA, tB, s);           //  4 waves, 32x16
    case 26: return launch_2_cfg<64, 160, 2, 5>(g, tA, tB, s);          // 10 waves, 32x32
    case 28: return launch_2_cfg<160, 128, 2, 8>(g, tA, tB, s);         // 16 waves, 80x16
    default: return TNT_BADARG(91);
  }
}

// NN only (A rows k-contiguous, B rows n-contiguous): the vocabulary head forward of tnt_gemm_f32
// (gemm2 at the same tile, cfg 28, is 8 % faster on random operands -- profiles/r02_gemm_cfg_scan.txt -- and level
// inside the training step, tools/ab_env.py: the head forward stays on gemm1r, which also carries the pre / act epilogue)
int32_t launch_one_round(const GemmArgs& g, hipStream_t s) { return launch_1r(1, g, false, false, s); }

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
  g.A2 = nullptr; g.C2 = nullptr; g.colsum = nullptr;
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

// ---- epilogue-free products of the backward pass and the LSTM input projection on the one-round family
// (include/tnt_hip.h: tnt_gemm_fused_f32)
namespace {
// automatic configuration: the largest-tile entry whose grid is ONE round of at most 256 workgroups with the
// best useful fraction; calibrated with tools/gemm_cfg_scan.py
int pick_cfg_1r(int M, int N, int K, bool tA, bool tB, int batch) {
  // Calibrated on MI355X with tools/gemm_cfg_scan.py (profiles/r02_gemm_cfg_scan.txt).  The one-round tiles pay on
  // NN / TN products whose grid fills most of the 256 CUs in one round; NT products with a long K and a narrow output
  // (input gradients) need a split contraction and are not served here (0: the caller routes them elsewhere).
  (void)K;
  if (tB) return 0;
  struct Cand { int cfg, bm, bn; };
  const Cand cands[] = {{24, 128, 64}, {26, 64, 160}, {21, 64, 64}, {25, 64, 32}};
  int best = 0; double best_score = 0.74;          // below 3/4 of the chip busy the tiled split-K kernel does better
  for (const Cand& c : cands) {
    const long tiles = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn) * batch;
    if (tiles > 256) continue;
    const double useful = (double)M * N * batch / ((double)tiles * c.bm * c.bn);
    const double score = useful * (double)tiles / 256.0;      // fraction of the chip's MFMA slots doing useful work
    if (score > best_score + 1e-9) { best_score = score; best = c.cfg; }
  }
  return best;
}
}  // namespace

extern "C" int32_t tnt_gemm_fused_f32(const float* A, const float* B, float* C, const float* bias, float* colsum,
                                      const float* A2, float* C2, int32_t M, int32_t N, int32_t K, int32_t lda,
                                      int32_t ldb, int32_t ldc, int32_t transA, int32_t transB, int32_t cfg,
                                      void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return TNT_BADARG(8);
  if (transA && transB) return TNT_BADARG(14);
  if ((A2 == nullptr) != (C2 == nullptr)) return TNT_BADARG(6);
  if (colsum != nullptr && !(transA && !transB)) return TNT_BADARG(5);
  if (!tnt_aligned16(A) || !tnt_aligned16(B) || (A2 && !tnt_aligned16(A2)) || lda % 4 || ldb % 4) return TNT_BADARG(1);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.pre = nullptr; g.work = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.act = TNT_ACT_NONE; g.slope = 0.f; g.accumulate = 0; g.kchunk = K; g.splitk = 1;
  g.A2 = A2; g.C2 = C2; g.colsum = colsum;
  if (cfg <= 0) cfg = pick_cfg_1r(M, N, K, transA != 0, transB != 0, A2 ? 2 : 1);
  if (cfg <= 0) return TNT_BADARG(92);           // no one-round configuration: the caller uses tnt_gemm_f32 (split-K)
  return launch_1r(cfg, g, transA != 0, transB != 0, tnt_stream(stream));
}

extern "C" int32_t tnt_gemm_fused_cfg(int32_t M, int32_t N, int32_t K, int32_t transA, int32_t transB, int32_t batch) {
  return pick_cfg_1r(M, N, K, transA != 0, transB != 0, batch > 1 ? 2 : 1);
}
