// Philox4x32-10 dropout stream (device side).  Bit-exact twin of oracle/philox.py:
//   element e of the logical tensor: counter = (lo32(e>>2), hi32(e>>2), site, step),
//   key = (lo32(seed), hi32(seed)); r = philox(counter,key)[e&3];
//   u = (r>>8) * 2^-24; keep = u >= rate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct TntPhilox4 { uint32_t v[4]; };

// 32 x 32 -> 64-bit product as ONE v_mad_u64_u32: hipcc lowers __umulhi(a, b) and a * b to v_mul_hi_u32 + v_mul_lo_u32, two
// quarter-rate instructions for the two halves of the same product.  64 M Philox calls on MI355X: 174 -> 119 us
// (tools/probe/philox_mad_probe.hip; identical words).
__device__ __forceinline__ uint64_t tnt_mul_wide(uint32_t a, uint32_t b) {
  uint64_t p, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(carry) : "v"(a), "v"(b));
  return p;
}

__device__ __forceinline__ TntPhilox4 tnt_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                         uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = tnt_mul_wide(0xD2511F53u, c0), p1 = tnt_mul_wide(0xCD9E8D57u, c2);
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  TntPhilox4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// u = (r >> 8) * 2^-24 >= rate  <=>  (r >> 8) >= ceil(rate * 2^24): both sides of the first form are exact in float32 (a
// 24-bit integer times a power of two; rate times a power of two), so the integer form is the same decision without the
// conversion and the multiply per word.
__device__ __forceinline__ uint32_t tnt_keep_threshold(float rate) { return (uint32_t)ceilf(rate * 16777216.0f); }

// keep decision for logical element e
__device__ __forceinline__ bool tnt_keep(uint64_t e, float rate, uint64_t seed, uint32_t site, uint32_t step) {
  const uint64_t g = e >> 2;
  TntPhilox4 r = tnt_philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), site, step, (uint32_t)seed, (uint32_t)(seed >> 32));
  // select chain, NOT r.v[e & 3]: a runtime index into the 4-word result sends the array to scratch memory
  // (measured: +10 us on a kernel in which 32 threads make this call once)
  const uint32_t sel = (uint32_t)e & 3u;
  const uint32_t w = sel == 0u ? r.v[0] : (sel == 1u ? r.v[1] : (sel == 2u ? r.v[2] : r.v[3]));
  return (w >> 8) >= tnt_keep_threshold(rate);
}

// 4 consecutive elements starting at e (e % 4 == 0): one Philox call
__device__ __forceinline__ void tnt_keep4(uint64_t e, float rate, uint64_t seed, uint32_t site, uint32_t step, bool k[4]) {
  const uint64_t g = e >> 2;
  TntPhilox4 r = tnt_philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), site, step, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t thr = tnt_keep_threshold(rate);
#pragma unroll
  for (int j = 0; j < 4; ++j) k[j] = (r.v[j] >> 8) >= thr;
}
