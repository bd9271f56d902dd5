// Shared helpers for the gfx950 kernel library (see include/tnt_hip.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tnt_hip.h"

#define TNT_WAVE 64

#define TNT_LAUNCH_CHECK()                      \
  do {                                          \
    hipError_t e_ = hipGetLastError();          \
    if (e_ != hipSuccess) return -(int32_t)e_;  \
  } while (0)

#define TNT_BADARG(k) (-1000 - (k))

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

static inline hipStream_t tnt_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline bool tnt_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float tnt_act(float x, int act, float slope) {
  switch (act) {
    case TNT_ACT_LEAKY: return x > 0.f ? x : x * slope;
    case TNT_ACT_RELU: return x > 0.f ? x : 0.f;
    case TNT_ACT_TANH: return tanhf(x);
    default: return x;
  }
}

__device__ __forceinline__ float tnt_act_grad(float pre, float dy, int act, float slope) {
  switch (act) {
    case TNT_ACT_LEAKY: return pre > 0.f ? dy : dy * slope;
    case TNT_ACT_RELU: return pre > 0.f ? dy : 0.f;
    case TNT_ACT_TANH: { float t = tanhf(pre); return dy * (1.f - t * t); }
    default: return dy;
  }
}

__device__ __forceinline__ float tnt_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// tanh via one hardware exp and one reciprocal: 1 - 2/(e^{2x}+1).  Absolute error ~1e-7 (a few ulp of 1),
// saturates correctly at +-inf; ~4x fewer VALU instructions than ocml tanhf, which matters in the
// attention step kernels (R x A tanh per sample per step, forward and recomputed in backward).
__device__ __forceinline__ float tnt_tanh(float x) {
  const float t = __expf(2.f * x);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);      // v_rcp_f32 (1 ulp), not the 10-instruction IEEE divide
}

// logistic via one hardware exp and one reciprocal (absolute error ~1e-7); used where the activation sits on
// the serial critical path of the T-step chain (LSTM gate math: ocml expf/tanhf cost ~300 VALU instructions per
// step there, ~0.4 us of a 6.6 us step).
__device__ __forceinline__ float tnt_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// wave-wide (64 lanes) reductions
__device__ __forceinline__ float tnt_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float tnt_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
