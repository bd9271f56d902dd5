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

int main() {
  struct Shape { const char* n; int M, N, K; } shapes[] = {{"head fwd", 960, 5001, 512}, {"xproj", 1024, 2048, 512}, {"out fwd c3", 960, 5001, 256}, {"big", 4096, 4096, 2048}};
  for (auto& s : shapes) {
    const int lda = s.K, ldb = (s.N + 3) / 4 * 4, ldc = ldb;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)s.M * lda * 4); hipMalloc(&B, (size_t)s.K * ldb * 4); hipMalloc(&C, (size_t)s.M * ldc * 4);
    std::vector<float> h((size_t)s.K * ldb);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(A, h.data(), (size_t)s.M * lda * 4, hipMemcpyHostToDevice);
    printf("%s  M=%d N=%d K=%d\n", s.n, s.M, s.N, s.K);
    run<64, 64, 32, 2, 2>("64x64 bk32 2x2 (current)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<64, 64, 16, 2, 2>("64x64 bk16 2x2", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<64, 64, 64, 2, 2>("64x64 bk64 2x2", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<64, 64, 32, 1, 2>("64x64 bk32 1x2 (wave 64x32)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<64, 64, 32, 1, 1>("64x64 bk32 1x1 (wave 64x64)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<32, 64, 32, 1, 2>("32x64 bk32 1x2", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<32, 128, 32, 1, 4>("32x128 bk32 1x4", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<64, 128, 16, 2, 2>("64x128 bk16 2x2 (wave 32x64)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<64, 128, 32, 2, 4>("64x128 bk32 2x4 (8 waves)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<128, 64, 16, 2, 2>("128x64 bk16 2x2 (wave 64x32)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<128, 128, 16, 2, 2>("128x128 bk16 2x2 (wave 64x64)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<128, 128, 16, 2, 4>("128x128 bk16 2x4 (8 waves)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<128, 128, 32, 4, 4>("128x128 bk32 4x4 (16 waves)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    run<128, 256, 16, 2, 4>("128x256 bk16 2x4 (8 waves)", A, B, C, s.M, s.N, s.K, lda, ldb, ldc);
    hipFree(A); hipFree(B); hipFree(C);
  }
  return 0;
}
