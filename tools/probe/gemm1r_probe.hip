// One-round FP32-MFMA GEMM probe: workgroup tiles sized so that the whole output is ONE wave of workgroups
// (tiles <= 256 CUs), 8 waves per workgroup, v_mfma_f32_16x16x4_f32 so that tile edges can be multiples of 16
// (160x128 covers the 960x5001 vocabulary head in 240 tiles; 64x160 covers its 512x5001 weight gradient in 256).
// Compares against the 64x64 kernel shape of csrc/gemm.hip on the same buffers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ float4 ldg4(const float* p, int valid) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid >= 4) {
    v = *reinterpret_cast<const float4*>(p);
  } else {
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
  }
  return v;
}

// Operand tile in LDS.  KC: memory is [mn][ld] with k contiguous, LDS keeps [mn][k] rows of BK+2 floats
// (16 rows x 2 k of a 32-lane group -> 32 banks; b64 stores conflict-free).  MC: memory is [k][ld] with mn
// contiguous, LDS keeps [k][mn] rows with 2 * LD = 16 mod 32 (lanes 0-15 read row k, lanes 16-31 row k + 2).
template <int BMN, int NT, bool KC, int BK>
struct Operand {
  static constexpr int QPR = BK / 4;   // float4 per k row
  static constexpr int NF4 = BMN * BK / 4;
  static constexpr int NV = (NF4 + NT - 1) / NT;
  static constexpr int LD = KC ? (BK + 2) : (BMN + ((40 - BMN % 32) % 32));   // MC: 2 * LD = 16 mod 32
  static constexpr int SZ = KC ? BMN * LD : BK * LD;
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
        r[i] = ldg4(base + (long)(mn0 + mn) * ld + k0 + kq, valid);
      } else {
        const int kk = f / (BMN / 4), mn4 = (f % (BMN / 4)) * 4;
        const int valid = (k0 + kk < k_lim) ? (mn_lim - (mn0 + mn4)) : 0;
        r[i] = ldg4(base + (long)(k0 + kk) * ld + mn0 + mn4, valid);
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
  // the k pair (8s + 2g, 8s + 2g + 1) of row/col `mn` (tile-local), g = lane >> 4
  static __device__ __forceinline__ float2 fetch(const float* lds, int mn, int s, int g) {
    if (KC) return *reinterpret_cast<const float2*>(&lds[mn * LD + 8 * s + 2 * g]);
    return make_float2(lds[(8 * s + 2 * g) * LD + mn], lds[(8 * s + 2 * g + 1) * LD + mn]);
  }
};

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC, int ABL = 0, int BK = 32>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm1r(const float* A, const float* B, float* C, const float* bias,
                                                         int M, int N, int K, int lda, int ldb, int ldc) {
  constexpr int NT = 64 * WGM * WGN;
  using OA = Operand<BM, NT, A_KC, BK>;
  using OB = Operand<BN, NT, B_KC, BK>;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  static_assert(WM % 16 == 0 && WN % 16 == 0, "wave tile");
  constexpr int ASZ = (OA::SZ + 3) & ~3, BSZ = (OB::SZ + 3) & ~3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds; float* Bs = lds + 2 * ASZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int MT = (M + BM - 1) / BM, NTl = (N + BN - 1) / BN;
  const int nwg = MT * NTl, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (t % MT) * BM, n0 = (t / MT) * BN;
  const int nk = (K + BK - 1) / BK;
  floatx4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int l16 = lane & 15, g = lane >> 4;
  float bcol[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * WN + tn * 16 + l16;
    bcol[tn] = (bias && col < N) ? bias[col] : 0.f;
  }
  OA oa; OB ob;
  oa.load(A, lda, m0, M, 0, K, tid); ob.load(B, ldb, n0, N, 0, K, tid);
  oa.store(As, tid); ob.store(Bs, tid);
  __syncthreads();
  if (ABL & 32) {
    // software-pipelined: fragments of step s+1 (and of the next chunk's step 0, after the barrier) are in flight
    // while the MFMAs of step s run
    float2 fa[2][TM], fb[2][TN];
    auto fetch = [&](int buf, const float* Ac, const float* Bc, int s) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) fa[buf][tm] = OA::fetch(Ac, wm * WM + tm * 16 + l16, s, g);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) fb[buf][tn] = OB::fetch(Bc, wn * WN + tn * 16 + l16, s, g);
    };
    fetch(0, As, Bs, 0);
    for (int i = 0; i < nk; ++i) {
      const int cur = i & 1;
      const bool more = i + 1 < nk;
      if (more) { oa.load(A, lda, m0, M, (i + 1) * BK, K, tid); ob.load(B, ldb, n0, N, (i + 1) * BK, K, tid); }
      const float* Ac = As + cur * ASZ; const float* Bc = Bs + cur * BSZ;
#pragma unroll
      for (int s = 0; s < BK / 8; ++s) {
        if (s + 1 < BK / 8) {
          fetch((s + 1) & 1, Ac, Bc, s + 1);
        } else {
          if (more) { oa.store(As + (cur ^ 1) * ASZ, tid); ob.store(Bs + (cur ^ 1) * BSZ, tid); }
          __syncthreads();
          if (more) fetch(0, As + (cur ^ 1) * ASZ, Bs + (cur ^ 1) * BSZ, 0);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s & 1][tm].x, fb[s & 1][tn].x, acc[tm][tn], 0, 0, 0);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s & 1][tm].y, fb[s & 1][tn].y, acc[tm][tn], 0, 0, 0);
      }
    }
  } else {
  for (int i = 0; i < nk; ++i) {
    const int cur = i & 1;
    if (i + 1 < nk && !(ABL & 1)) { oa.load(A, lda, m0, M, (i + 1) * BK, K, tid); ob.load(B, ldb, n0, N, (i + 1) * BK, K, tid); }
    const float* Ac = As + cur * ASZ; const float* Bc = Bs + cur * BSZ;
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      float2 av[TM], bv[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) av[tm] = (ABL & 16) ? make_float2(acc[tm][0][0], acc[tm][0][1]) : OA::fetch(Ac, wm * WM + tm * 16 + l16, s, g);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bv[tn] = (ABL & 16) ? make_float2(1.f + g, 2.f) : OB::fetch(Bc, wn * WN + tn * 16 + l16, s, g);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm].x, bv[tn].x, acc[tm][tn], 0, 0, 0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm].y, bv[tn].y, acc[tm][tn], 0, 0, 0);
    }
    if (i + 1 < nk && !(ABL & 2)) { oa.store(As + (cur ^ 1) * ASZ, tid); ob.store(Bs + (cur ^ 1) * BSZ, tid); }
    if (!(ABL & 4)) __syncthreads();
  }
  }
  // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
  const bool fullm = (m0 + BM <= M);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * WN + tn * 16 + l16;
    if (col >= N) continue;
    float* Cp = C + (long)(m0 + wm * WM + 4 * g) * ldc + col;
    if (fullm) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!(ABL & 8) || acc[tm][tn][r] == 12345.f) Cp[(long)(tm * 16 + r) * ldc] = acc[tm][tn][r] + bcol[tn];
    } else {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * WM + tm * 16 + 4 * g + r;
          if (row < M) Cp[(long)(tm * 16 + r) * ldc] = acc[tm][tn][r] + bcol[tn];
        }
    }
  }
}

// Direct-to-LDS variant (NN, K % 32 == 0, BK = 32, 1024 threads): global_load_lds_dwordx4 writes each wave's 1 KB
// straight into LDS, so there is no VGPR staging and no ds_write.  A: slot (row, c) holds k-chunk c ^ ((row>>1)&7)
// (conflict-free ds_read_b64 over 64 banks); B: slot (k, c) holds column chunk c ^ (((k>>1)&1)<<2).
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm1r_glds(const float* A, const float* B, float* C, const float* bias,
                                                              int M, int N, int K, int lda, int ldb, int ldc) {
  constexpr int BK = 32, NT = 64 * WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  constexpr int ASZ = BM * BK, BSZ = BK * BN;
  constexpr int NA = (BM * BK / 4 + NT - 1) / NT, NB = (BK * BN / 4 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds; float* Bs = lds + 2 * ASZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int MT = (M + BM - 1) / BM, NTl = (N + BN - 1) / BN;
  const int nwg = MT * NTl, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (t % MT) * BM, n0 = (t / MT) * BN;
  const int nk = K / BK;
  const int l16 = lane & 15, g = lane >> 4;
  floatx4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  float bcol[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * WN + tn * 16 + l16;
    bcol[tn] = (bias && col < N) ? bias[col] : 0.f;
  }
  auto issue = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * NT;
      if (BM * BK / 4 % NT != 0 && (wave * 64 + i * NT) >= BM * BK / 4) continue;      // wave-uniform
      const int row = f >> 3, c = f & 7;
      const int grow = min(m0 + row, M - 1);
      const float* src = A + (long)grow * lda + k0 + 4 * (c ^ ((row >> 1) & 7));
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(As + buf * ASZ + (wave * 64 + i * NT) * 4), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * NT;
      if (BK * BN / 4 % NT != 0 && (wave * 64 + i * NT) >= BK * BN / 4) continue;
      const int kk = f / (BN / 4), c = f % (BN / 4);
      const int gcol = min(n0 + 4 * (c ^ (((kk >> 1) & 1) << 2)), ldb - 4);
      const float* src = B + (long)(k0 + kk) * ldb + gcol;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(Bs + buf * BSZ + (wave * 64 + i * NT) * 4), 16, 0, 0);
    }
  };
  issue(0, 0);
  __syncthreads();
  for (int i = 0; i < nk; ++i) {
    const int cur = i & 1;
    if (i + 1 < nk) issue(cur ^ 1, (i + 1) * BK);
    const float* Ac = As + cur * ASZ; const float* Bc = Bs + cur * BSZ;
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      float2 av[TM], bv[TN];
      const int Q = 2 * s + (g >> 1);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row = wm * WM + tm * 16 + l16;
        av[tm] = *reinterpret_cast<const float2*>(&Ac[row * BK + ((Q ^ ((row >> 1) & 7)) << 2) + 2 * (g & 1)]);
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n = wn * WN + tn * 16 + l16, k = 8 * s + 2 * g;
        const int col = (((n >> 2) ^ (((k >> 1) & 1) << 2)) << 2) + (n & 3);          // k and k+1 share (k>>1)
        bv[tn] = make_float2(Bc[k * BN + col], Bc[(k + 1) * BN + col]);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm].x, bv[tn].x, acc[tm][tn], 0, 0, 0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm].y, bv[tn].y, acc[tm][tn], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * WN + tn * 16 + l16;
    if (col >= N) continue;
    float* Cp = C + (long)(m0 + wm * WM + 4 * g) * ldc + col;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * WM + tm * 16 + 4 * g + r;
        if (row < M) Cp[(long)(tm * 16 + r) * ldc] = acc[tm][tn][r] + bcol[tn];
      }
  }
}

__global__ void ref_gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                         int ldc, int a_kc, int b_kc) {
  const int col = blockIdx.x * 64 + threadIdx.x, row = blockIdx.y;
  if (col >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = a_kc ? A[(long)row * lda + k] : A[(long)k * lda + row];
    const float b = b_kc ? B[(long)col * ldb + k] : B[(long)k * ldb + col];
    s = fmaf(a, b, s);
  }
  C[(long)row * ldc + col] = s + (bias ? bias[col] : 0.f);
}

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC, int ABL = 0, int BK = 32>
void run(const char* name, const float* A, const float* B, float* C, const float* Cref, const float* bias, int M, int N, int K,
         int lda, int ldb, int ldc) {
  const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  constexpr int NT = 64 * WGM * WGN;
  using OA = Operand<BM, NT, A_KC, BK>; using OB = Operand<BN, NT, B_KC, BK>;
  const int shmem = (2 * ((OA::SZ + 3) & ~3) + 2 * ((OB::SZ + 3) & ~3)) * 4;
  auto kern = gemm1r<BM, BN, WGM, WGN, A_KC, B_KC, ABL, BK>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
  hipMemset(C, 0, (size_t)M * ldc * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kern<<<grid, NT, shmem>>>(A, B, C, bias, M, N, K, lda, ldb, ldc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) kern<<<grid, NT, shmem>>>(A, B, C, bias, M, N, K, lda, ldb, ldc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20;
  std::vector<float> h((size_t)M * ldc), hr((size_t)M * ldc);
  hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), Cref, h.size() * 4, hipMemcpyDeviceToHost);
  double md = 0, mx = 0;
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < N; ++c) {
      md = fmax(md, fabs((double)h[(size_t)r * ldc + c] - hr[(size_t)r * ldc + c]));
      mx = fmax(mx, fabs((double)hr[(size_t)r * ldc + c]));
    }
  hipError_t e = hipGetLastError();
  printf("  %-34s grid=%5d x%4d : %7.1f us  %6.1f TF   maxdiff %.2e (max|C| %.1f) %s\n", name, grid, NT, us,
         2.0 * M * N * K / us / 1e6, md, mx, e == hipSuccess ? "" : hipGetErrorString(e));
}

template <int BM, int BN, int WGM, int WGN>
void rung(const char* name, const float* A, const float* B, float* C, const float* Cref, const float* bias, int M, int N, int K,
          int lda, int ldb, int ldc) {
  if (K % 32) return;
  const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  constexpr int NT = 64 * WGM * WGN;
  const int shmem = (2 * BM * 32 + 2 * 32 * BN) * 4;
  auto kern = gemm1r_glds<BM, BN, WGM, WGN>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
  hipMemset(C, 0, (size_t)M * ldc * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kern<<<grid, NT, shmem>>>(A, B, C, bias, M, N, K, lda, ldb, ldc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) kern<<<grid, NT, shmem>>>(A, B, C, bias, M, N, K, lda, ldb, ldc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20;
  std::vector<float> h((size_t)M * ldc), hr((size_t)M * ldc);
  hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), Cref, h.size() * 4, hipMemcpyDeviceToHost);
  double md = 0;
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < N; ++c) md = fmax(md, fabs((double)h[(size_t)r * ldc + c] - hr[(size_t)r * ldc + c]));
  hipError_t e = hipGetLastError();
  printf("  %-34s grid=%5d x%4d : %7.1f us  %6.1f TF   maxdiff %.2e %s\n", name, grid, NT, us, 2.0 * M * N * K / us / 1e6, md,
         e == hipSuccess ? "" : hipGetErrorString(e));
}

int main() {
  struct Shape { const char* n; int M, N, K, a_kc, b_kc; } shapes[] = {
      {"head fwd NN", 960, 5001, 512, 1, 0}, {"c3 head fwd NN", 960, 5001, 256, 1, 0}, {"head dW TN", 512, 5001, 960, 0, 0},
      {"xproj NN", 1024, 2048, 512, 1, 0}, {"lstm dW TN", 512, 2048, 1024, 0, 0}};
  for (auto& s : shapes) {
    const int lda = s.a_kc ? s.K : (s.M + 3) / 4 * 4;
    const int ldb = s.b_kc ? s.K : (s.N + 3) / 4 * 4;
    const int ldc = (s.N + 3) / 4 * 4;
    const size_t na = (size_t)(s.a_kc ? s.M : s.K) * lda, nb = (size_t)(s.b_kc ? s.N : s.K) * ldb;
    float *A, *B, *C, *Cr, *bias;
    hipMalloc(&A, na * 4 + 4096); hipMalloc(&B, nb * 4 + 4096); hipMalloc(&C, (size_t)s.M * ldc * 4); hipMalloc(&Cr, (size_t)s.M * ldc * 4);
    hipMalloc(&bias, (size_t)ldc * 4);
    std::vector<float> h(na > nb ? na : nb);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 40503u + 17) % 997) / 498.f - 1.f;
    hipMemcpy(B, h.data(), nb * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, h.data(), (size_t)ldc * 4, hipMemcpyHostToDevice);
    hipMemset(Cr, 0, (size_t)s.M * ldc * 4);
    ref_gemm<<<dim3((s.N + 63) / 64, s.M), 64>>>(A, B, Cr, bias, s.M, s.N, s.K, lda, ldb, ldc, s.a_kc, s.b_kc);
    hipDeviceSynchronize();
    printf("%s  M=%d N=%d K=%d\n", s.n, s.M, s.N, s.K);
#define RUN(BM, BN, WGM, WGN, AK, BKC) run<BM, BN, WGM, WGN, AK, BKC>(#BM "x" #BN " waves " #WGM "x" #WGN, A, B, C, Cr, bias, s.M, s.N, s.K, lda, ldb, ldc)
    if (s.a_kc) {
      RUN(64, 64, 2, 2, true, false);
      RUN(128, 128, 2, 4, true, false);
      RUN(160, 128, 2, 4, true, false);
      RUN(160, 128, 2, 8, true, false);
      rung<160, 128, 2, 8>("160x128 2x8 direct-to-LDS", A, B, C, Cr, bias, s.M, s.N, s.K, lda, ldb, ldc);
      rung<160, 128, 2, 4>("160x128 2x4 direct-to-LDS", A, B, C, Cr, bias, s.M, s.N, s.K, lda, ldb, ldc);
#define RUNA(WGN, ABL) run<160, 128, 2, WGN, true, false, ABL>("160x128 2x" #WGN " ablation " #ABL, A, B, C, Cr, bias, s.M, s.N, s.K, lda, ldb, ldc)
      RUNA(4, 1); RUNA(4, 3); RUNA(4, 7); RUNA(4, 15); RUNA(4, 8); RUNA(4, 4);
      run<160, 128, 2, 8, true, false, 0, 64>("160x128 2x8 bk64", A, B, C, Cr, bias, s.M, s.N, s.K, lda, ldb, ldc);
      run<160, 128, 2, 8, true, false, 0, 16>("160x128 2x8 bk16", A, B, C, Cr, bias, s.M, s.N, s.K, lda, ldb, ldc);
      run<160, 128, 2, 4, true, false, 0, 64>("160x128 2x4 bk64", A, B, C, Cr, bias, s.M, s.N, s.K, lda, ldb, ldc);
      run<160, 128, 2, 8, true, false, 15, 64>("160x128 2x8 bk64 abl15", A, B, C, Cr, bias, s.M, s.N, s.K, lda, ldb, ldc);
      RUNA(4, 32); RUNA(8, 32); RUNA(4, 31); RUNA(8, 31); RUNA(8, 16);
      RUNA(8, 1); RUNA(8, 3); RUNA(8, 7); RUNA(8, 15); RUNA(8, 8); RUNA(8, 4);
      RUN(80, 128, 1, 8, true, false);
      RUN(80, 128, 1, 4, true, false);
      RUN(160, 64, 2, 4, true, false);
      RUN(160, 64, 2, 2, true, false);
      RUN(96, 128, 2, 4, true, false);
      RUN(96, 128, 2, 8, true, false);
    } else {
      RUN(64, 64, 2, 2, false, false);
      RUN(128, 128, 2, 4, false, false);
      RUN(64, 160, 2, 5, false, false);
      RUN(64, 160, 4, 2, false, false);
      RUN(128, 64, 2, 4, false, false);
      RUN(64, 128, 2, 4, false, false);
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(Cr); hipFree(bias);
  }
  return 0;
}
