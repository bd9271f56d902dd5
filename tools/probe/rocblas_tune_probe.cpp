// Enumerates every rocBLAS (Tensile) solution for the epilogue-free GEMM shapes of the step, times each one and
// checks run-to-run bit-reproducibility (atomics are disabled on the handle).  Prints the library's default pick
// against the best solution index: the evidence behind csrc/blas.hip's per-shape solution cache.
#define ROCBLAS_BETA_FEATURES_API
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { auto e = (x); if ((int)e != 0) { printf("error %d at line %d\n", (int)e, __LINE__); return 1; } } while (0)
int main() {
  rocblas_handle h; CK(rocblas_create_handle(&h));
  CK(rocblas_set_atomics_mode(h, rocblas_atomics_not_allowed));
  hipStream_t s; CK(hipStreamCreate(&s)); CK(rocblas_set_stream(h, s));
  struct Sh { const char* n; int M, N, K; bool tA, tB; int ldn; };
  Sh shapes[] = {{"head dW TN", 512, 5001, 960, true, false, 5004}, {"head dX NT", 960, 512, 5001, false, true, 0},
                 {"xproj NN", 1024, 2048, 512, false, false, 0}, {"lstm dW TN", 512, 2048, 1024, true, false, 0},
                 {"lstm dX NT", 1024, 512, 2048, false, true, 0}, {"c3 head dW TN", 256, 5001, 960, true, false, 5004},
                 {"c3 head dX NT", 960, 256, 5001, false, true, 0}};
  for (auto& q : shapes) {
    // row-major C[M][N] = opA(A) opB(B); column-major view C^T[N][M] = opB(B)^T opA(A)^T
    const int lda = q.tA ? q.M : q.K;
    const int ldb = q.tB ? (q.ldn ? q.ldn : (q.K + 3) / 4 * 4) : (q.ldn ? q.ldn : q.N);
    const int ldc = q.ldn && !q.tB ? q.ldn : q.N;
    const size_t na = (size_t)(q.tA ? q.K : q.M) * lda, nb = (size_t)(q.tB ? q.N : q.K) * ldb, nc = (size_t)q.M * ldc;
    float *A, *B, *C;
    CK(hipMalloc(&A, na * 4)); CK(hipMalloc(&B, nb * 4)); CK(hipMalloc(&C, nc * 4));
    std::vector<float> ha(na), hb(nb);
    for (size_t i = 0; i < na; ++i) ha[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    for (size_t i = 0; i < nb; ++i) hb[i] = (float)((i * 40503u + 17) % 997) / 498.f - 1.f;
    CK(hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice));
    const float one = 1.f, zero = 0.f;
    const rocblas_operation opB = q.tB ? rocblas_operation_transpose : rocblas_operation_none;
    const rocblas_operation opA = q.tA ? rocblas_operation_transpose : rocblas_operation_none;
    auto run = [&](int sol) {
      return rocblas_gemm_ex(h, opB, opA, q.N, q.M, q.K, &one, B, rocblas_datatype_f32_r, ldb, A, rocblas_datatype_f32_r, lda,
                             &zero, C, rocblas_datatype_f32_r, ldc, C, rocblas_datatype_f32_r, ldc, rocblas_datatype_f32_r,
                             sol < 0 ? rocblas_gemm_algo_standard : rocblas_gemm_algo_solution_index, sol < 0 ? 0 : sol, 0);
    };
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time_it = [&](int sol, double* us) -> int {
      for (int i = 0; i < 2; ++i) if (run(sol) != rocblas_status_success) return 1;
      hipStreamSynchronize(s);
      hipEventRecord(a, s);
      for (int i = 0; i < 10; ++i) run(sol);
      hipEventRecord(b, s); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      *us = ms / 10 * 1e3;
      return 0;
    };
    double us_def = 0; CK(time_it(-1, &us_def));
    std::vector<float> ref(nc), got(nc), got2(nc);
    CK(hipMemcpy(ref.data(), C, nc * 4, hipMemcpyDeviceToHost));
    rocblas_int nsol = 0;
    CK(rocblas_gemm_ex_get_solutions(h, opB, opA, q.N, q.M, q.K, &one, B, rocblas_datatype_f32_r, ldb, A, rocblas_datatype_f32_r,
                                     lda, &zero, C, rocblas_datatype_f32_r, ldc, C, rocblas_datatype_f32_r, ldc,
                                     rocblas_datatype_f32_r, rocblas_gemm_algo_solution_index, 0, nullptr, &nsol));
    std::vector<rocblas_int> sols(nsol);
    CK(rocblas_gemm_ex_get_solutions(h, opB, opA, q.N, q.M, q.K, &one, B, rocblas_datatype_f32_r, ldb, A, rocblas_datatype_f32_r,
                                     lda, &zero, C, rocblas_datatype_f32_r, ldc, C, rocblas_datatype_f32_r, ldc,
                                     rocblas_datatype_f32_r, rocblas_gemm_algo_solution_index, 0, sols.data(), &nsol));
    struct R { double us; int sol; bool repro; double err; };
    std::vector<R> res;
    for (int i = 0; i < nsol; ++i) {
      double us;
      if (time_it(sols[i], &us)) continue;
      hipMemcpy(got.data(), C, nc * 4, hipMemcpyDeviceToHost);
      run(sols[i]); hipStreamSynchronize(s);
      hipMemcpy(got2.data(), C, nc * 4, hipMemcpyDeviceToHost);
      double err = 0;
      for (int r = 0; r < q.M; ++r) for (int c = 0; c < q.N; ++c) { const size_t o = (size_t)r * ldc + c; err = std::max(err, (double)fabsf(got[o] - ref[o])); }
      bool same = true;
      for (int r = 0; r < q.M && same; ++r) same = memcmp(&got[(size_t)r * ldc], &got2[(size_t)r * ldc], (size_t)q.N * 4) == 0;
      res.push_back({us, sols[i], same, err});
    }
    std::sort(res.begin(), res.end(), [](const R& x, const R& y) { return x.us < y.us; });
    const double fl = 2.0 * q.M * q.N * q.K;
    printf("%-14s M=%5d N=%5d K=%5d: default %7.2f us %6.1f TF | %d solutions; best:", q.n, q.M, q.N, q.K, us_def, fl / us_def / 1e6, nsol);
    for (size_t i = 0; i < res.size() && i < 4; ++i)
      printf("  #%d %.2f us %.1f TF%s err %.1e", res[i].sol, res[i].us, fl / res[i].us / 1e6, res[i].repro ? "" : " NONREPRO", res[i].err);
    printf("\n");
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
  }
  return 0;
}
