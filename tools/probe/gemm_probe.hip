// Standalone sweep of FP32-MFMA GEMM workgroup shapes on the hot-path sizes (NN layout).
// Same algorithm as csrc/gemm.hip, generalised over (BM, BN, BK, waves layout); picks what to port.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int BMN, int BK, int NT, bool KCONTIG>
struct Loader {
  static constexpr int NV = (BMN * BK / 4 + NT - 1) / NT;
  static constexpr int PAD = KCONTIG ? 1 : 4;
  static constexpr int LD = BMN + PAD;
  float4 r[NV];
  __device__ __forceinline__ void load(const float* base, int ld, int mn0, int mn_lim, int k0, int k_lim, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      r[i] = make_float4(0, 0, 0, 0);
      if (f >= BMN * BK / 4) continue;
      if (KCONTIG) {
        const int mn = f / (BK / 4), kq = (f % (BK / 4)) * 4;
        if (mn0 + mn < mn_lim && k0 + kq + 3 < k_lim) r[i] = *reinterpret_cast<const float4*>(base + (long)(mn0 + mn) * ld + k0 + kq);
      } else {
        const int kk = f / (BMN / 4), mn4 = (f % (BMN / 4)) * 4;
        if (k0 + kk < k_lim && mn0 + mn4 + 3 < mn_lim) r[i] = *reinterpret_cast<const float4*>(base + (long)(k0 + kk) * ld + mn0 + mn4);
      }
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      if (f >= BMN * BK / 4) continue;
      if (KCONTIG) {
        const int mn = f / (BK / 4), kq = (f % (BK / 4)) * 4;
        lds[(kq + 0) * LD + mn] = r[i].x; lds[(kq + 1) * LD + mn] = r[i].y;
        lds[(kq + 2) * LD + mn] = r[i].z; lds[(kq + 3) * LD + mn] = r[i].w;
      } else {
        const int kk = f / (BMN / 4), mn4 = (f % (BMN / 4)) * 4;
        *reinterpret_cast<float4*>(&lds[kk * LD + mn4]) = r[i];
      }
    }
  }
};

template <int BM, int BN, int BK, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_nn(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
  constexpr int NT = 64 * WGM * WGN;
  using LA = Loader<BM, BK, NT, true>;
  using LB = Loader<BN, BK, NT, false>;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
  constexpr int ASZ = BK * LA::LD, BSZ = BK * LB::LD, AOFF = (2 * ASZ + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float lds[AOFF + 2 * BSZ];
  float* As = lds; float* Bs = lds + AOFF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int MT = (M + BM - 1) / BM, NTl = (N + BN - 1) / BN;
  const int nwg = MT * NTl, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (t % MT) * BM, n0 = (t / MT) * BN;
  const int nk = (K + BK - 1) / BK;
  floatx16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  LA la; LB lb;
  la.load(A, lda, m0, M, 0, K, tid); lb.load(B, ldb, n0, N, 0, K, tid);
  la.store(As, tid); lb.store(Bs, tid);
  __syncthreads();
  const int lrow = lane & 31, lk = lane >> 5;
  const float* Abase = As + lk * LA::LD + wm * WM + lrow;
  const float* Bbase = Bs + lk * LB::LD + wn * WN + lrow;
  for (int i = 0; i < nk; ++i) {
    const int cur = i & 1;
    if (i + 1 < nk) { la.load(A, lda, m0, M, (i + 1) * BK, K, tid); lb.load(B, ldb, n0, N, (i + 1) * BK, K, tid); }
    const float* Ac = Abase + cur * ASZ; const float* Bc = Bbase + cur * BSZ;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) av[tm] = Ac[kk * LA::LD + tm * 32];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bv[tn] = Bc[kk * LB::LD + tn * 32];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
    }
    if (i + 1 < nk) { la.store(As + (cur ^ 1) * ASZ, tid); lb.store(Bs + (cur ^ 1) * BSZ, tid); }
    __syncthreads();
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * WN + tn * 32 + lrow;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < M) C[(long)row * ldc + col] = acc[tm][tn][r];
      }
    }
}


template <int BM, int BN, int BK, int WGM, int WGN, int ABL>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_nn_abl(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
  constexpr int NT = 64 * WGM * WGN;
  using LA = Loader<BM, BK, NT, true>;
  using LB = Loader<BN, BK, NT, false>;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
  constexpr int ASZ = BK * LA::LD, BSZ = BK * LB::LD, AOFF = (2 * ASZ + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float lds[AOFF + 2 * BSZ];
  float* As = lds; float* Bs = lds + AOFF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int MT = (M + BM - 1) / BM, NTl = (N + BN - 1) / BN;
  const int nwg = MT * NTl, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (t % MT) * BM, n0 = (t / MT) * BN;
  const int nk = (K + BK - 1) / BK;
  floatx16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  LA la; LB lb;
  la.load(A, lda, m0, M, 0, K, tid); lb.load(B, ldb, n0, N, 0, K, tid);
  la.store(As, tid); lb.store(Bs, tid);
  __syncthreads();
  const int lrow = lane & 31, lk = lane >> 5;
  const float* Abase = As + lk * LA::LD + wm * WM + lrow;
  const float* Bbase = Bs + lk * LB::LD + wn * WN + lrow;
  for (int i = 0; i < nk; ++i) {
    const int cur = i & 1;
    if (!(ABL & 1) && i + 1 < nk) { la.load(A, lda, m0, M, (i + 1) * BK, K, tid); lb.load(B, ldb, n0, N, (i + 1) * BK, K, tid); }
    const float* Ac = Abase + cur * ASZ; const float* Bc = Bbase + cur * BSZ;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) av[tm] = Ac[kk * LA::LD + tm * 32];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bv[tn] = Bc[kk * LB::LD + tn * 32];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
    }
    if (!(ABL & 2) && i + 1 < nk) { la.store(As + (cur ^ 1) * ASZ, tid); lb.store(Bs + (cur ^ 1) * BSZ, tid); }
    if (!(ABL & 4)) __syncthreads();
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * WN + tn * 32 + lrow;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < M && (!(ABL & 8) || acc[tm][tn][r] == 12345.678f)) C[(long)row * ldc + col] = acc[tm][tn][r];
      }
    }
}



// ---- decoupled waves: every wave owns a (32*TM) x (32*TN) output tile with a private LDS double buffer;
// no workgroup barrier anywhere (LDS is in-order per wave), so waves drift apart and fill each other's stalls.
template <int TM, int TN, int BK>
__global__ __launch_bounds__(256) void gemm_nn_wave(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
  constexpr int WM = 32 * TM, WN = 32 * TN;
  constexpr int ALD = WM + 1, BLD = WN + 4;
  constexpr int ASZ = BK * ALD, BSZ = BK * BLD, WSZ = ((2 * ASZ + 3) & ~3) + 2 * BSZ;
  __shared__ __attribute__((aligned(16))) float lds[4 * WSZ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* As = lds + wave * WSZ;
  float* Bs = As + ((2 * ASZ + 3) & ~3);
  const int MT = (M + WM - 1) / WM, NTl = (N + WN - 1) / WN;
  const int wid = blockIdx.x * 4 + wave;
  if (wid >= MT * NTl) return;
  // n-tile fastest inside a block (4 waves share the A rows), m-tiles next
  const int tn = wid % NTl, tm = wid / NTl;
  const int m0 = tm * WM, n0 = tn * WN;
  const int nk = (K + BK - 1) / BK;
  floatx16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  constexpr int NA = WM * BK / 4 / 64, NB = WN * BK / 4 / 64;
  float4 ra[NA], rb[NB];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = lane + i * 64, mn = f / (BK / 4), kq = (f % (BK / 4)) * 4;
      ra[i] = (m0 + mn < M && k0 + kq + 3 < K) ? *reinterpret_cast<const float4*>(A + (long)(m0 + mn) * lda + k0 + kq) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = lane + i * 64, kk = f / (WN / 4), n4 = (f % (WN / 4)) * 4;
      rb[i] = (k0 + kk < K && n0 + n4 + 3 < N) ? *reinterpret_cast<const float4*>(B + (long)(k0 + kk) * ldb + n0 + n4) : make_float4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int buf) {
    float* a = As + buf * ASZ; float* b = Bs + buf * BSZ;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = lane + i * 64, mn = f / (BK / 4), kq = (f % (BK / 4)) * 4;
      a[(kq + 0) * ALD + mn] = ra[i].x; a[(kq + 1) * ALD + mn] = ra[i].y; a[(kq + 2) * ALD + mn] = ra[i].z; a[(kq + 3) * ALD + mn] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = lane + i * 64, kk = f / (WN / 4), n4 = (f % (WN / 4)) * 4;
      *reinterpret_cast<float4*>(&b[kk * BLD + n4]) = rb[i];
    }
  };
  gload(0); sstore(0);
  const int lrow = lane & 31, lk = lane >> 5;
  for (int i = 0; i < nk; ++i) {
    const int cur = i & 1;
    if (i + 1 < nk) gload((i + 1) * BK);
    const float* Ac = As + cur * ASZ + lk * ALD + lrow;
    const float* Bc = Bs + cur * BSZ + lk * BLD + lrow;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) av[t] = Ac[kk * ALD + t * 32];
#pragma unroll
      for (int t = 0; t < TN; ++t) bv[t] = Bc[kk * BLD + t * 32];
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[u], acc[t][u], 0, 0, 0);
    }
    if (i + 1 < nk) sstore(cur ^ 1);
  }
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int col = n0 + u * 32 + lrow;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < M) C[(long)row * ldc + col] = acc[t][u][r];
      }
    }
}

template <int TM, int TN, int BK>
void runw(const char* name, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
  const int tiles = ((M + 32 * TM - 1) / (32 * TM)) * ((N + 32 * TN - 1) / (32 * TN));
  const int grid = (tiles + 3) / 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) gemm_nn_wave<TM, TN, BK><<<grid, 256>>>(A, B, C, M, N, K, lda, ldb, ldc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) gemm_nn_wave<TM, TN, BK><<<grid, 256>>>(A, B, C, M, N, K, lda, ldb, ldc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20;
  printf("  %-34s waves=%5d : %7.1f us  %6.1f TF\n", name, tiles, us, 2.0 * M * N * K / us / 1e6);
}

template <int BM, int BN, int BK, int WGM, int WGN>
void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
  const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) gemm_nn<BM, BN, BK, WGM, WGN><<<grid, 64 * WGM * WGN>>>(A, B, C, M, N, K, lda, ldb, ldc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) gemm_nn<BM, BN, BK, WGM, WGN><<<grid, 64 * WGM * WGN>>>(A, B, C, M, N, K, lda, ldb, ldc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20;
  printf("  %-34s grid=%5d : %7.1f us  %6.1f TF\n", name, grid, us, 2.0 * M * N * K / us / 1e6);
}



// ---- direct-to-LDS variant: global_load_lds_dwordx4 for both operands (no VGPR staging, no ds_write),
// A rows kept k-contiguous in LDS with an XOR chunk swizzle and read with ds_read_b128 (4 MFMAs per read).
template <int TMW>   // waves as 2x2, wave tile 32*TMW x 32  (TMW = 1: 64x64 block)
__global__ __launch_bounds__(256) void gemm_nn_glds(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
  constexpr int BM = 64 * TMW, BN = 64, BK = 32;
  __shared__ __attribute__((aligned(16))) float As[2][BM * BK];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int MT = (M + BM - 1) / BM, NTl = (N + BN - 1) / BN;
  const int nwg = MT * NTl, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (t % MT) * BM, n0 = (t / MT) * BN;
  const int nk = K / BK;
  floatx16 acc[TMW];
  for (int i = 0; i < TMW; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  auto issue = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < BM * BK / 4 / 256; ++i) {          // A: LDS float4 slot f <- global (row, chunk ^ swz)
      const int f = tid + i * 256, row = f >> 3, c = f & 7;
      const int grow = min(m0 + row, M - 1);
      const float* src = A + (long)grow * lda + k0 + 4 * (c ^ ((row >> 1) & 7));
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(&As[buf][(wave * 64 + i * 256) * 4]), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BK * BN / 4 / 256; ++i) {          // B: rows of 64 floats
      const int f = tid + i * 256, kk = f >> 4, c = f & 15;
      const int gcol = min(n0 + 4 * c, ldb - 4);
      const float* src = B + (long)(k0 + kk) * ldb + gcol;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(&Bs[buf][(wave * 64 + i * 256) * 4]), 16, 0, 0);
    }
  };
  issue(0, 0);
  __syncthreads();
  const int lrow = lane & 31, lk = lane >> 5;
  for (int i = 0; i < nk; ++i) {
    const int cur = i & 1;
    if (i + 1 < nk) issue(cur ^ 1, (i + 1) * BK);
    const float* Bc = &Bs[cur][wn * 32 + lrow];
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      float4 a4[TMW];
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm) {
        const int row = wm * 32 * TMW + tm * 32 + lrow;
        a4[tm] = *reinterpret_cast<const float4*>(&As[cur][row * BK + (((2 * j + lk) ^ ((row >> 1) & 7)) << 2)]);
      }
      const float b0 = Bc[(8 * j + 4 * lk + 0) * BN], b1 = Bc[(8 * j + 4 * lk + 1) * BN];
      const float b2 = Bc[(8 * j + 4 * lk + 2) * BN], b3 = Bc[(8 * j + 4 * lk + 3) * BN];
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm) {
        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].x, b0, acc[tm], 0, 0, 0);
        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].y, b1, acc[tm], 0, 0, 0);
        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].z, b2, acc[tm], 0, 0, 0);
        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].w, b3, acc[tm], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int tm = 0; tm < TMW; ++tm) {
    const int col = n0 + wn * 32 + lrow;
    if (col >= N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 * TMW + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (row < M) C[(long)row * ldc + col] = acc[tm][r];
    }
  }
}

template <int TMW>
void rung(const char* name, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
  const int grid = ((M + 64 * TMW - 1) / (64 * TMW)) * ((N + 63) / 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) gemm_nn_glds<TMW><<<grid, 256>>>(A, B, C, M, N, K, lda, ldb, ldc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) gemm_nn_glds<TMW><<<grid, 256>>>(A, B, C, M, N, K, lda, ldb, ldc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20;
  printf("  %-34s grid=%5d : %7.1f us  %6.1f TF\n", name, grid, us, 2.0 * M * N * K / us / 1e6);
}

// correctness check of the glds kernel against the plain kernel
void check(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
  std::vector<float> c1((size_t)M * ldc), c2((size_t)M * ldc);
  hipMemset(C, 0, (size_t)M * ldc * 4);
  gemm_nn<64, 64, 16, 2, 2><<<((M + 63) / 64) * ((N + 63) / 64), 256>>>(A, B, C, M, N, K, lda, ldb, ldc);
  hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
  hipMemset(C, 0, (size_t)M * ldc * 4);
  gemm_nn_glds<1><<<((M + 63) / 64) * ((N + 63) / 64), 256>>>(A, B, C, M, N, K, lda, ldb, ldc);
  hipMemcpy(c2.data(), C, c2.size() * 4, hipMemcpyDeviceToHost);
  double maxd = 0, maxv = 0;
  for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { maxd = fmax(maxd, fabs(c1[(size_t)m * ldc + n] - c2[(size_t)m * ldc + n])); maxv = fmax(maxv, fabs(c1[(size_t)m * ldc + n])); }
  printf("  glds check: max |diff| = %.3e (max |C| = %.3e)\n", maxd, maxv);
}

template <int ABL>
void runabl(const char* name, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
  const int grid = ((M + 63) / 64) * ((N + 63) / 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) gemm_nn_abl<64, 64, 16, 2, 2, ABL><<<grid, 256>>>(A, B, C, M, N, K, lda, ldb, ldc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) gemm_nn_abl<64, 64, 16, 2, 2, ABL><<<grid, 256>>>(A, B, C, M, N, K, lda, ldb, ldc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20;
  printf("  ablation %-28s : %7.1f us  %6.1f TF\n", name, us, 2.0 * M * N * K / us / 1e6);
}

int main() {
  struct Shape { const char* n; int M, N, K; } shapes[] = {{"head fwd", 960, 5001, 512}, {"xproj", 1024, 2048, 512}, {"out fwd c3", 960, 5001, 256}, {"big", 4096, 4096, 2048}};
  for (auto& s : shapes) {
    const int lda = s.K, ldb = (s.N + 3) / 4 * 4, ldc = ldb;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)s.M * lda * 4); hipMalloc(&B, (size_t)s.K * ldb * 4 + 4096); hipMalloc(&C, (size_t)s.M * ldc * 4);
    std::vector<float> h((size_t)s.K * ldb);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(A, h.data(), (size_t)s.M * lda * 4, hipMemcpyHostToDevice);
    printf("%s  M=%d N=%d K=%d\n", s.n, s.M, s.N, s.K);
    run<64, 64, 32, 2, 2>("64x64 bk32 2x2 (current)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<64, 64, 16, 2, 2>("64x64 bk16 2x2", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    check(A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    rung<1>("glds 64x64 bk32 (b128 A reads)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    rung<2>("glds 128x64 bk32 (wave 64x32)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runabl<0>("full", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runabl<1>("no global loads", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runabl<3>("no gloads, no lds stores", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runabl<7>("... and no barrier", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runabl<15>("... and no epilogue store", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runabl<8>("full minus epilogue store", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runabl<2>("full minus lds stores", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runw<1, 1, 16>("wave 32x32 bk16 (no barrier)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runw<1, 2, 16>("wave 32x64 bk16 (no barrier)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runw<2, 1, 16>("wave 64x32 bk16 (no barrier)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runw<2, 2, 16>("wave 64x64 bk16 (no barrier)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runw<1, 2, 32>("wave 32x64 bk32 (no barrier)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    runw<2, 2, 8>("wave 64x64 bk8 (no barrier)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<32, 128, 32, 1, 4>("32x128 bk32 1x4", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<64, 128, 16, 2, 2>("64x128 bk16 2x2 (wave 32x64)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<64, 128, 32, 2, 4>("64x128 bk32 2x4 (8 waves)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<128, 64, 16, 2, 2>("128x64 bk16 2x2 (wave 64x32)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<128, 128, 16, 2, 2>("128x128 bk16 2x2 (wave 64x64)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<128, 128, 16, 2, 4>("128x128 bk16 2x4 (8 waves)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    hipFree(A); hipFree(B); hipFree(C);
  }
  return 0;
}
