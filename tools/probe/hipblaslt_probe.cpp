// hipBLASLt FP32 GEMM probe: for each of the config-2 / config-3 GEMM shapes, time every algorithm the heuristic returns
// (row-major C[M][N] = op(A) op(B) expressed as the column-major product the library sees), next to rocBLAS sgemm.
// build: hipcc -O2 --offload-arch=gfx950 tools/probe/hipblaslt_probe.cpp -o tools/probe/hipblaslt_probe -lhipblaslt -lrocblas
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#include <cstring>
#include <cmath>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

struct Shape { const char* name; int M, N, K; bool tA, tB; };

int main() {
  hipblasLtHandle_t lt; CK(hipblasLtCreate(&lt));
  rocblas_handle rb; CK(rocblas_create_handle(&rb));
  rocblas_set_atomics_mode(rb, rocblas_atomics_not_allowed);
  hipStream_t st; CK(hipStreamCreate(&st));
  rocblas_set_stream(rb, st);
  const size_t wsz = 256u << 20;
  void* ws; CK(hipMalloc(&ws, wsz));
  Shape shapes[] = {
      {"head fwd  NN 960x5001x512", 960, 5001, 512, false, false},
      {"head dX   NT 960x512x5001", 960, 512, 5001, false, true},
      {"head dW   TN 512x5001x960", 512, 5001, 960, true, false},
      {"lstm dW   TN 512x2048x1024", 512, 2048, 1024, true, false},
      {"lstm xprj NN 1024x2048x512", 1024, 2048, 512, false, false},
      {"lstm dX   NT 1024x512x2048", 1024, 512, 2048, false, true},
      {"c3 head fwd NN 960x5001x256", 960, 5001, 256, false, false},
      {"c3 head dX  NT 960x256x5001", 960, 256, 5001, false, true},
      {"c3 head dW  TN 256x5001x960", 256, 5001, 960, true, false},
  };
  for (auto& s : shapes) {
    // row-major A: [M][K] (or [K][M] if tA), B: [K][N] (or [N][K] if tB), C: [M][N]
    const int lda = s.tA ? s.M : s.K, ldb = s.tB ? s.K : s.N, ldc = s.N;
    float *A, *B, *C;
    CK(hipMalloc(&A, sizeof(float) * (size_t)s.M * s.K)); CK(hipMalloc(&B, sizeof(float) * (size_t)s.K * s.N));
    CK(hipMalloc(&C, sizeof(float) * (size_t)s.M * s.N));
    {   // random operands in [-1, 1): timing on zeros flatters the clocks, and the checks below need data
      std::vector<float> h((size_t)s.M * s.K);
      unsigned r = 12345u;
      for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (float)(int)(r >> 8) / 8388608.f - 1.f; }
      CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      h.resize((size_t)s.K * s.N);
      for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (float)(int)(r >> 8) / 8388608.f - 1.f; }
      CK(hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    // column-major view: C^T[N][M] = op(B)^T op(A)^T: first operand = B (col-major N x K with ld ldb if !tB), second = A
    hipblasOperation_t opB = s.tB ? HIPBLAS_OP_T : HIPBLAS_OP_N, opA = s.tA ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    hipblasLtMatmulDesc_t desc; CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opB, sizeof(opB)));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opA, sizeof(opA)));
    hipblasLtMatrixLayout_t la, lb, lc;
    // first operand: stored (col-major) as rows x cols = (tB ? K x N : N x K), ld = ldb
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_32F, s.tB ? s.K : s.N, s.tB ? s.N : s.K, ldb));
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_32F, s.tA ? s.M : s.K, s.tA ? s.K : s.M, lda));
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, s.N, s.M, ldc));
    hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
    std::vector<hipblasLtMatmulHeuristicResult_t> res(32);
    int nres = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(lt, desc, la, lb, lc, lc, pref, 32, res.data(), &nres));
    const float one = 1.f, zero = 0.f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double gf = 2.0 * s.M * s.N * s.K * 1e-9;      // GFLOP; GFLOP / us = 1000 TFLOP/s
    float best = 1e9; int besti = -1;
    for (int i = 0; i < nres; ++i) {
      auto run = [&]() { return hipblasLtMatmul(lt, desc, &one, B, la, A, lb, &zero, C, lc, C, lc, &res[i].algo, ws, wsz, st); };
      if (run() != HIPBLAS_STATUS_SUCCESS) continue;
      for (int k = 0; k < 5; ++k) run();
      hipEventRecord(e0, st);
      for (int k = 0; k < 50; ++k) run();
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const float us = ms * 1e3f / 50;
      if (us < best) { best = us; besti = i; }
      printf("  %s  lt algo %2d: %7.2f us %6.1f TF  ws %zu\n", s.name, i, us, gf / us * 1e3, res[i].workspaceSize);
    }
    // determinism + agreement with rocBLAS of the best algorithm: 20 runs, bitwise equal outputs
    std::vector<float> c0((size_t)s.M * s.N), c1((size_t)s.M * s.N);
    int ndiff = 0;
    if (besti >= 0) {
      for (int rep = 0; rep < 20; ++rep) {
        hipMemsetAsync(C, 0xff, c0.size() * 4, st);
        hipblasLtMatmul(lt, desc, &one, B, la, A, lb, &zero, C, lc, C, lc, &res[besti].algo, ws, wsz, st);
        hipStreamSynchronize(st);
        hipMemcpy(rep == 0 ? c0.data() : c1.data(), C, c0.size() * 4, hipMemcpyDeviceToHost);
        if (rep > 0 && memcmp(c0.data(), c1.data(), c0.size() * 4) != 0) ++ndiff;
      }
    }
    // rocBLAS reference (same call the library makes today)
    auto rrun = [&]() {
      return rocblas_sgemm(rb, s.tB ? rocblas_operation_transpose : rocblas_operation_none,
                           s.tA ? rocblas_operation_transpose : rocblas_operation_none, s.N, s.M, s.K, &one, B, ldb, A, lda,
                           &zero, C, ldc);
    };
    rrun(); for (int k = 0; k < 5; ++k) rrun();
    hipEventRecord(e0, st);
    for (int k = 0; k < 50; ++k) rrun();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipStreamSynchronize(st);
    hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
    double maxd = 0, maxv = 0;
    for (size_t i = 0; i < c0.size(); ++i) { double d = fabs((double)c0[i] - c1[i]); if (d > maxd) maxd = d; if (fabs(c1[i]) > maxv) maxv = fabs(c1[i]); }
    printf("%s: best algo runs differing from the first: %d of 19; max |lt - rocblas| = %.3e (max |c| %.3e)\n", s.name, ndiff, maxd, maxv);
    printf("%s: rocBLAS %7.2f us %6.1f TF | hipBLASLt best algo %d of %d: %7.2f us %6.1f TF\n", s.name, ms * 20, gf / (ms * 20) * 1e3,
           besti, nres, best, gf / best * 1e3);
    hipFree(A); hipFree(B); hipFree(C);
  }
  return 0;
}
