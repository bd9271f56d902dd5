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

// ---- cross-lane exchange without the LDS crossbar.  `__shfl_xor` compiles to ds_bpermute_b32: an LDS-pipe instruction with an
// address register and ~100 cycles of latency, three to six of them chained in every reduction of the step kernels and the
// chains.  The DPP forms ride in the VALU instruction itself (v_add_f32_dpp), the gfx950 permlane swaps are one VALU op.
//   tnt_x1 / tnt_x2     partner lane ^ 1, lane ^ 2            (quad_perm: exact)
//   tnt_ror<N>          partner (lane + N) % 16 inside the row of 16; = lane ^ 8 for N = 8, and equal in VALUE to lane ^ N for
//                       N = 4, 2, 1 once the values are periodic in the row with period 2 N (the state of a butterfly reduction
//                       after its larger steps)
//   tnt_hmirror / tnt_mirror   partner 7 - i inside the half row / 15 - i inside the row: the other half's value once each half
//                       (8 lanes / 16 lanes ... ) is uniform, i.e. as the LAST steps of a reduction over adjacent lanes
//   tnt_x16_add / tnt_x32_add  v + v[lane ^ 16], v + v[lane ^ 32]   (v_permlane16_swap / v_permlane32_swap: exact)
template <int CTRL>
__device__ __forceinline__ float tnt_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float tnt_x1(float v) { return tnt_dpp<0xB1>(v); }          // quad_perm:[1,0,3,2]
__device__ __forceinline__ float tnt_x2(float v) { return tnt_dpp<0x4E>(v); }          // quad_perm:[2,3,0,1]
template <int N> __device__ __forceinline__ float tnt_ror(float v) { return tnt_dpp<0x120 + N>(v); }   // row_ror:N
__device__ __forceinline__ float tnt_hmirror(float v) { return tnt_dpp<0x141>(v); }    // row_half_mirror
__device__ __forceinline__ float tnt_mirror(float v) { return tnt_dpp<0x140>(v); }     // row_mirror
__device__ __forceinline__ float tnt_x16_add(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float tnt_x32_add(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float tnt_x16_max(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float tnt_x32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// wave-wide (64 lanes) reductions: the butterfly 32, 16, 8, 4, 2, 1 -- the same partners' VALUES, in the same order, as the
// __shfl_xor form it replaces (bit-identical results)
__device__ __forceinline__ float tnt_wave_sum(float v) {
  v = tnt_x32_add(v); v = tnt_x16_add(v);
  v += tnt_ror<8>(v); v += tnt_ror<4>(v); v += tnt_x2(v); v += tnt_x1(v);
  return v;
}
__device__ __forceinline__ float tnt_wave_max(float v) {
  v = tnt_x32_max(v); v = tnt_x16_max(v);
  v = fmaxf(v, tnt_ror<8>(v)); v = fmaxf(v, tnt_ror<4>(v)); v = fmaxf(v, tnt_x2(v)); v = fmaxf(v, tnt_x1(v));
  return v;
}

// 16-byte accesses with an optional non-temporal hint: streams that are touched once per step (the optimizer moments, a
// gradient being consumed) should not displace what the next kernels re-read from the L2s / the Infinity Cache.
template <bool NT>
__device__ __forceinline__ float4 tnt_ld4(const float* p) {
  if (NT) {
    const floatx4 v = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
  }
  return *reinterpret_cast<const float4*>(p);
}
template <bool NT>
__device__ __forceinline__ void tnt_st4(float* p, const float4& v) {
  if (NT) {
    const floatx4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<floatx4*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}
// Process-wide policy of the streaming optimizer kernels: non-temporal moments / consumed gradient unless TNT_STREAM_NT=0
// (the A/B switch).  Measured on config 2 (tools/step_breakdown.py, same box): the kernels timed alone, the same launch
// repeated, are SLOWER non-temporal (fused encoder update 44.4 -> 51.0 us, Adam of the rest 33.7 -> 35.4: a repeated launch
// finds its own moments in the Infinity Cache), the training step is FASTER, 0.4849 -> 0.4722 ms: 140 MB of moments per step
// no longer push the weights, gradients and activations (~160 MB) out of the 256 MB cache between their producer and consumer.
static inline bool tnt_stream_policy_nt() {
  static const int on = [] { const char* e = getenv("TNT_STREAM_NT"); return e != nullptr && e[0] == '0' ? 0 : 1; }();
  return on != 0;
}
