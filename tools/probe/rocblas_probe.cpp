// rocBLAS sgemm on the hot-path shapes (row-major C = op(A) op(B) expressed as column-major C^T = op(B)^T op(A)^T).
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#define CK(x) do { auto e = (x); if ((int)e != 0) { printf("error %d at line %d\n", (int)e, __LINE__); return 1; } } while (0)
int main() {
  rocblas_handle h; CK(rocblas_create_handle(&h));
  hipStream_t s; CK(hipStreamCreate(&s)); CK(rocblas_set_stream(h, s));
  struct Sh { const char* n; int M, N, K; bool tA, tB; };
  Sh shapes[] = {{"head fwd NN", 960, 5004, 512, false, false}, {"head dW TN", 512, 5004, 960, true, false},
                 {"head dX NT", 960, 512, 5004, false, true}, {"xproj NN", 1024, 2048, 512, false, false},
                 {"lstm dW TN", 512, 2048, 1024, true, false}, {"lstm dX NT", 1024, 512, 2048, false, true},
                 {"enc dW TN", 20000, 512, 64, true, false}};
  for (auto& q : shapes) {
    float *A, *B, *C;
    size_t na = (size_t)q.M * q.K, nb = (size_t)q.K * q.N, nc = (size_t)q.M * q.N;
    CK(hipMalloc(&A, na * 4)); CK(hipMalloc(&B, nb * 4)); CK(hipMalloc(&C, nc * 4));
    std::vector<float> ha(na, 0.5f), hb(nb, 0.25f);
    CK(hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice));
    const float one = 1.f, zero = 0.f;
    // row-major: C[M][N] = opA(A) opB(B).  column-major view: C^T[N][M] = opB(B)^T opA(A)^T
    const int lda = q.tA ? q.M : q.K, ldb = q.tB ? q.K : q.N;
    auto run = [&]() {
      return rocblas_sgemm(h, q.tB ? rocblas_operation_transpose : rocblas_operation_none,
                           q.tA ? rocblas_operation_transpose : rocblas_operation_none, q.N, q.M, q.K, &one, B, ldb, A, lda,
                           &zero, C, q.N);
    };
    for (int i = 0; i < 3; ++i) CK(run());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < 20; ++i) CK(run());
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms / 20 * 1e3;
    printf("%-12s M=%5d N=%5d K=%5d: %8.2f us %7.1f TF\n", q.n, q.M, q.N, q.K, us, 2.0 * q.M * q.N * q.K / us / 1e6);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
  }
  return 0;
}
