// Plain library GEMM (rocBLAS sgemm, exact f32) for the matmuls of the step that carry no fused epilogue: the weight
// and input gradients tape.gradient derives from the Dense / LSTM layers (lc_NIC.py:386-387, NIC.py:248-249).
// On MI355X the vendor's stream-K Tensile kernels reach 105-118 TF on these skinny-output / long-K shapes
// (512x2048x1024, 1024x512x2048, 512x5004x960) where the hand-written tiled kernel of gemm.hip needs a split-K pass
// plus a reduce launch (66-80 TF effective; tools/probe/rocblas_probe.cpp, profiles/r01_gemm_design_probe.txt).
// The fused GEMMs (bias / activation / pre-activation epilogues, the vocabulary-head forward, the encoder) stay on
// gemm.hip.  Row-major C = op(A) op(B) is issued as the column-major product C^T = op(B)^T op(A)^T.
// PROCESS-GLOBAL STATE (the one exception to the "no global state" rule of include/tnt_hip.h, documented there): the
// rocBLAS handle, the hipBLASLt handle and the per-shape hipBLASLt plans live for the life of the process; g_blas_mu
// serialises their creation and every call that touches them, so the two entry points may be called from any thread.
// Since round 3 neither is on a default training path (the hand-written kernels of gemm3.hip are); both remain as A/B tools.
#include "tnt_common.h"
#include <rocblas/rocblas.h>
#include <mutex>

namespace {
std::mutex g_blas_mu;
rocblas_handle g_handle = nullptr;       // one per process (one process per GPU)

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
  std::lock_guard<std::mutex> lock(g_blas_mu);
  if (int32_t rc = blas_init()) return rc;
  if (rocblas_set_stream(g_handle, tnt_stream(stream)) != rocblas_status_success) return -1003;
  const float one = 1.f, beta = accumulate ? 1.f : 0.f;
  const rocblas_status st =
      rocblas_sgemm(g_handle, transB ? rocblas_operation_transpose : rocblas_operation_none,
                    transA ? rocblas_operation_transpose : rocblas_operation_none, N, M, K, &one, B, ldb, A, lda, &beta, C, ldc);
  return st == rocblas_status_success ? 0 : -(1100 + (int32_t)st);
}

// ---------------------------------------------------------------------------------------------------
// hipBLASLt FP32 matmul (HIPBLAS_COMPUTE_32F: exact f32 products, no reduced-precision passes) with the optional bias
// epilogue, for the three vocabulary-head GEMMs of a step (NIC.py:143 and its gradients): on MI355X its kernels run
// 960x5001x512 / 960x512x5001 / 512x5001x960 at 112-117 TF, against 92 TF for gemm.hip's one-round kernel and
// 79-92 TF for rocBLAS (tools/probe/hipblaslt_probe.cpp, which also checks 20 runs bitwise equal).  The algorithm is
// fixed per shape, so every process multiplies with the same kernel and results stay reproducible run to run: a solution
// index from the table below when the shape is listed and this hipBLASLt build still supports it for the problem, else
// the heuristic's first workspace-free candidate.  The table was produced on MI355X (ROCm 7.2.0) by running both bench
// workloads with TNT_LT_TUNE=1, which times the heuristic's workspace-free candidates on the caller's stream at the first
// call outside a stream capture and prints the winner in the table's format (tools/lt_tune.py).
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <cstdio>
#include <map>
#include <tuple>
#include <vector>

namespace {
struct LtPreset { int M, N, K, tA, tB, bias, index; };
const LtPreset g_lt_presets[] = {
    // {M, N, K, transA, transB, bias, solution index}
    {960, 5001, 512, 0, 0, 1, 626981},      // config 2 head: logits = Out Wo + b        (NIC.py:143)
    {512, 5001, 960, 1, 0, 0, 626697},      //                dWo = Out^T dlogits
    {960, 512, 5001, 0, 1, 0, 627486},      //                dOut = dlogits Wo^T
    {960, 5001, 256, 0, 0, 1, 626947},      // config 3 head (lc_NIC.py:264-266)
    {256, 5001, 960, 1, 0, 0, 626698},
    {960, 256, 5001, 0, 1, 0, 627499},
};

struct LtPlan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  bool ok = false;
};
// the bias pointer is part of the key: a plan's descriptor is written once, when the plan is made, and never again
using LtKey = std::tuple<int, int, int, int, int, int, int, int, const float*>;
hipblasLtHandle_t g_lt = nullptr;
std::map<LtKey, LtPlan> g_lt_plans;

int32_t lt_run(const LtPlan& p, const hipblasLtMatmulAlgo_t& algo, const float* A, const float* B, float* C, hipStream_t s) {
  const float one = 1.f, zero = 0.f;
  // column-major view: C^T = op(B)^T op(A)^T, first operand B
  return hipblasLtMatmul(g_lt, p.desc, &one, B, p.la, A, p.lb, &zero, C, p.lc, C, p.lc, &algo, nullptr, 0, s) ==
                 HIPBLAS_STATUS_SUCCESS ? 0 : -1;
}
}  // namespace

extern "C" int32_t tnt_gemm_lt_f32(const float* A, const float* B, float* C, const float* bias, int32_t M, int32_t N,
                                   int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t transA, int32_t transB,
                                   void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return TNT_BADARG(5);
  if (transA && transB) return TNT_BADARG(11);
  std::lock_guard<std::mutex> lock(g_blas_mu);
  if (!g_lt && hipblasLtCreate(&g_lt) != HIPBLAS_STATUS_SUCCESS) { g_lt = nullptr; return -1201; }
  hipStream_t s = tnt_stream(stream);
  const LtKey key{M, N, K, lda, ldb, ldc, transA, transB, bias};
  LtPlan& p = g_lt_plans[key];
  if (!p.ok) {
    const hipblasOperation_t opB = transB ? HIPBLAS_OP_T : HIPBLAS_OP_N, opA = transA ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    if (hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return -1202;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opB, sizeof(opB));
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opA, sizeof(opA));
    if (bias) {       // one value per output column of the row-major C = per row of the column-major C^T
      const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
      hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi));
      hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
    }
    if (hipblasLtMatrixLayoutCreate(&p.la, HIP_R_32F, transB ? K : N, transB ? N : K, ldb) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_32F, transA ? M : K, transA ? K : M, lda) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_32F, N, M, ldc) != HIPBLAS_STATUS_SUCCESS)
      return -1203;
    hipblasLtMatmulPreference_t pref;
    if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return -1204;
    const size_t no_ws = 0;
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &no_ws, sizeof(no_ws));
    std::vector<hipblasLtMatmulHeuristicResult_t> res(40);
    int nres = 0;
    const auto hs = hipblasLtMatmulAlgoGetHeuristic(g_lt, p.desc, p.la, p.lb, p.lc, p.lc, pref, (int)res.size(), res.data(), &nres);
    hipblasLtMatmulPreferenceDestroy(pref);
    if (hs != HIPBLAS_STATUS_SUCCESS || nres <= 0) return -1205;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    int best = -1;
    static const bool tune0 = getenv("TNT_LT_TUNE") && atoi(getenv("TNT_LT_TUNE")) != 0;
    if (!tune0) {
      for (const LtPreset& q : g_lt_presets) {
        if (q.index < 0 || q.M != M || q.N != N || q.K != K || q.tA != (transA != 0) || q.tB != (transB != 0) ||
            q.bias != (bias != nullptr))
          continue;
        std::vector<int> idx{q.index};
        std::vector<hipblasLtMatmulHeuristicResult_t> got;
        size_t need = 0;
        const float one = 1.f, zero = 0.f;
        if (hipblaslt_ext::getAlgosFromIndex(g_lt, idx, got) == HIPBLAS_STATUS_SUCCESS && !got.empty() &&
            hipblaslt_ext::matmulIsAlgoSupported(g_lt, p.desc, &one, p.la, p.lb, &zero, p.lc, p.lc, got[0].algo, need) ==
                HIPBLAS_STATUS_SUCCESS && need == 0) {
          p.algo = got[0].algo;
          p.ok = true;
        }
        break;
      }
    }
    const bool tune = tune0;
    if (p.ok) {
    } else if (tune && cap == hipStreamCaptureStatusNone) {
      hipEvent_t e0, e1;
      if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1206;
      float best_ms = 1e30f;
      for (int i = 0; i < nres; ++i) {
        if (res[i].workspaceSize != 0 || lt_run(p, res[i].algo, A, B, C, s) != 0) continue;
        lt_run(p, res[i].algo, A, B, C, s);
        (void)hipEventRecord(e0, s);
        for (int k = 0; k < 8; ++k) lt_run(p, res[i].algo, A, B, C, s);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess) continue;
        float ms = 1e30f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best_ms) { best_ms = ms; best = i; }
      }
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    } else {
      for (int i = 0; i < nres && best < 0; ++i) if (res[i].workspaceSize == 0) best = i;
    }
    if (!p.ok) {
      if (best < 0) return -1207;
      p.algo = res[best].algo;
      p.ok = true;
      if (tune)
        fprintf(stderr, "tnt_gemm_lt preset: {%d, %d, %d, %d, %d, %d, %d},\n", M, N, K, transA != 0, transB != 0, bias != nullptr,
                hipblaslt_ext::getIndexFromAlgo(p.algo));
    }
  }
  return lt_run(p, p.algo, A, B, C, s) == 0 ? 0 : -1208;
}
