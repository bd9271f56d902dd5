// Plain library GEMM (rocBLAS sgemm, exact f32) for the matmuls of the step that carry no fused epilogue: the weight
// and input gradients tape.gradient derives from the Dense / LSTM layers (lc_NIC.py:386-387, NIC.py:248-249).
// On MI355X the vendor's stream-K Tensile kernels reach 105-118 TF on these skinny-output / long-K shapes
// (512x2048x1024, 1024x512x2048, 512x5004x960) where the hand-written tiled kernel of gemm.hip needs a split-K pass
// plus a reduce launch (66-80 TF effective; tools/probe/rocblas_probe.cpp, profiles/r01_gemm_design_probe.txt).
// The fused GEMMs (bias / activation / pre-activation epilogues, the vocabulary-head forward, the encoder) stay on
// gemm.hip.  Row-major C = op(A) op(B) is issued as the column-major product C^T = op(B)^T op(A)^T.
#include "tnt_common.h"
#include <rocblas/rocblas.h>

namespace {
rocblas_handle g_handle = nullptr;       // one per process (one process per GPU, one launching thread)

int32_t blas_init() {
  if (g_handle) return 0;
  // No rocblas_initialize(): preloading every Tensile code object costs ~30 s per process.  The kernels a step needs
  // are loaded lazily by the first (eager, un-captured) pass that every captured launch sequence is preceded by.
  if (rocblas_create_handle(&g_handle) != rocblas_status_success) { g_handle = nullptr; return -1001; }
  // bitwise run-to-run reproducibility is part of this library's contract: no atomics-based split reductions
  if (rocblas_set_atomics_mode(g_handle, rocblas_atomics_not_allowed) != rocblas_status_success) return -1002;
  return 0;
}
}  // namespace

extern "C" int32_t tnt_gemm_blas_f32(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K,
                                     int32_t lda, int32_t ldb, int32_t ldc, int32_t transA, int32_t transB,
                                     int32_t accumulate, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return TNT_BADARG(4);
  if (transA && transB) return TNT_BADARG(10);
  if (int32_t rc = blas_init()) return rc;
  if (rocblas_set_stream(g_handle, tnt_stream(stream)) != rocblas_status_success) return -1003;
  const float one = 1.f, beta = accumulate ? 1.f : 0.f;
  const rocblas_status st =
      rocblas_sgemm(g_handle, transB ? rocblas_operation_transpose : rocblas_operation_none,
                    transA ? rocblas_operation_transpose : rocblas_operation_none, N, M, K, &one, B, ldb, A, lda, &beta, C, ldc);
  return st == rocblas_status_success ? 0 : -(1100 + (int32_t)st);
}
