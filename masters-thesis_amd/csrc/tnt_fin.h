// Shared pieces of the optimizer step's scalar tail (csrc/optim.hip, csrc/encoder.hip): the span table, the canonical
// span -> variable reduction and Adam's bias-corrected step size.  "Canonical" matters: since round 3 the clip norm of a
// variable is summed from the span partials by EVERY workgroup that needs it (tnt_adam_fin_f32, tnt_dense_dw_adam_fin_f32)
// instead of by one finalize launch in front of them -- all of them, and the workgroup that files the norms for the host,
// must add the same numbers in the same order.
#pragma once
#include "tnt_common.h"

struct SpanTab {
  const int32_t* span_seg;     // segment of span s
  const int64_t* span_off;     // first element (arena offset, multiple of 4)
  const int32_t* span_len;     // elements in span (<= SPAN)
  const int32_t* seg_first;    // [nseg+1] first span of each segment
  const float* seg_l2;         // L2 lambda per segment
};

// (sum of partial[2 k], sum of partial[2 k + 1]) over the spans k0 <= k < k1 of one variable, computed by a whole wave,
// every lane returning the same value: up to 8 spans serially in span order, more lane-strided + the fixed shuffle tree
// (the orders tnt_step_finalize_f32 has always used).
// The loads of one round are issued together (8 per lane) and only then added in order: a load-add loop is a chain of
// dependent L2 round trips, ~0.5 us each, in front of every workgroup that calls this (20 of them for the dense encoder's
// 1250 spans -- measured +5 us on a 43 us launch).
__device__ __forceinline__ float2 tnt_seg_sums(const float* partial, int k0, int k1, int lane) {
  const float2* p2 = reinterpret_cast<const float2*>(partial);         // (sum g^2, sum theta^2) of span k
  float q = 0.f, w = 0.f;
  float2 p[8];
  if (k1 - k0 <= 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = k0 + j < k1 ? p2[k0 + j] : make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { q += p[j].x; w += p[j].y; }          // + 0 behind the last span changes nothing
  } else {
    for (int kb = k0 + lane; kb < k1; kb += 512) {
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = kb + 64 * j < k1 ? p2[kb + 64 * j] : make_float2(0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { q += p[j].x; w += p[j].y; }
    }
    q = tnt_wave_sum(q); w = tnt_wave_sum(w);
  }
  return make_float2(q, w);
}

// One span's pair (sum (g + 2 lambda theta)^2, sum theta^2), one 256-thread workgroup, result in thread 0's slot of
// `partial`.  `ovr` (optional, the norm launches of the fused step): what nobody will read is not loaded --
//   * a variable whose clip norm is supplied from elsewhere (ovr[seg] >= 0: the Embedding's IndexedSlices norm, filed there
//     by the previous update and re-filed by this one) needs no pass over its gradient;
//   * sum theta^2 only feeds the L2 metric lambda * sum theta^2: without a regulariser theta is not read and the slot is 0.
// For BASELINE config 2 that is 33 MB instead of 58 MB per step behind the norm launch (HBM-bound).
__device__ __forceinline__ void tnt_span_norm(const float* theta, const float* grad, const SpanTab& t, int sp, float* partial,
                                              const float* ovr, float* s0, float* s1) {
  const int tid = threadIdx.x;
  const long off = t.span_off[sp];
  const int len = t.span_len[sp];
  const int seg = t.span_seg[sp];
  const float lam2 = 2.f * t.seg_l2[seg];
  const bool need_w = ovr == nullptr || lam2 != 0.f;
  const bool need_g = ovr == nullptr || !(ovr[seg] >= 0.f);
  float q = 0.f, wq = 0.f;
  const int len4 = len & ~3;
  if (need_g && need_w) {
#pragma unroll 4
    for (int i = tid * 4; i < len4; i += 1024) {
      const float4 g = *reinterpret_cast<const float4*>(grad + off + i);
      const float4 w = *reinterpret_cast<const float4*>(theta + off + i);
      const float a0 = g.x + lam2 * w.x, a1 = g.y + lam2 * w.y, a2 = g.z + lam2 * w.z, a3 = g.w + lam2 * w.w;
      q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
      wq += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
    }
    for (int i = len4 + tid; i < len; i += 256) {
      const float w = theta[off + i], a0 = grad[off + i] + lam2 * w;
      q += a0 * a0; wq += w * w;
    }
  } else if (need_g) {                       // lambda == 0: the gradient alone
#pragma unroll 8
    for (int i = tid * 4; i < len4; i += 1024) {
      const float4 g = *reinterpret_cast<const float4*>(grad + off + i);
      q += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
    }
    for (int i = len4 + tid; i < len; i += 256) q += grad[off + i] * grad[off + i];
  } else if (need_w) {                       // norm supplied, regulariser present: sum theta^2 for the L2 metric
#pragma unroll 8
    for (int i = tid * 4; i < len4; i += 1024) {
      const float4 w = *reinterpret_cast<const float4*>(theta + off + i);
      wq += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
    }
    for (int i = len4 + tid; i < len; i += 256) wq += theta[off + i] * theta[off + i];
  }
  q = tnt_wave_sum(q); wq = tnt_wave_sum(wq);
  if ((tid & 63) == 0) { s0[tid >> 6] = q; s1[tid >> 6] = wq; }
  __syncthreads();
  if (tid == 0) {
    partial[2 * sp + 0] = s0[0] + s0[1] + s0[2] + s0[3];
    partial[2 * sp + 1] = s1[0] + s1[1] + s1[2] + s1[3];
  }
}

// lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) for the step being applied, t = *adam_t + 1 (the counter is advanced at the END of the
// update launch, by its last workgroup, when nobody reads it any more)
__device__ __forceinline__ float tnt_adam_lr_t(const int64_t* adam_t, const float* lr, float b1, float b2) {
  const int64_t t = adam_t[0] + 1;
  const double p1 = pow((double)b1, (double)t), p2 = pow((double)b2, (double)t);
  return (float)((double)lr[0] * sqrt(1.0 - p2) / (1.0 - p1));
}
