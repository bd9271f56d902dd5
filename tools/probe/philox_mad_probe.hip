// Philox4x32-10 round throughput: v_mul_hi_u32 + v_mul_lo_u32 (what hipcc emits for __umulhi / *) against one
// v_mad_u64_u32 per product.  usage: ./philox_mad_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ void mulhilo_a(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) { hi = __umulhi(a, b); lo = a * b; }
__device__ __forceinline__ void mulhilo_b(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
  uint64_t p, carry;
  asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(carry) : "v"(a), "v"(b));
  hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
}

template <int V>
__global__ void k(uint32_t* out, int iters, uint32_t k0i, uint32_t k1i) {
  uint32_t c0 = blockIdx.x * 256 + threadIdx.x, c1 = 0, c2 = 7, c3 = 9, acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint32_t k0 = k0i, k1 = k1i;
    c0 += it;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint32_t hi0, lo0, hi1, lo1;
      if (V == 0) { mulhilo_a(0xD2511F53u, c0, hi0, lo0); mulhilo_a(0xCD9E8D57u, c2, hi1, lo1); }
      else { mulhilo_b(0xD2511F53u, c0, hi0, lo0); mulhilo_b(0xCD9E8D57u, c2, hi1, lo1); }
      const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
      c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    acc ^= c0 ^ c1 ^ c2 ^ c3;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  uint32_t* out; hipMalloc(&out, 4096 * 256 * 4);
  uint32_t h[2][4];
  for (int v = 0; v < 2; ++v) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, 0, out, 64, 1u, 2u);
      else hipLaunchKernelGGL(k<1>, dim3(4096), dim3(256), 0, 0, out, 64, 1u, 2u);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("variant %d: %.1f us for %d M philox calls\n", v, ms * 1e3, 4096 * 256 * 64 >> 20);
    }
    hipMemcpy(h[v], out, 16, hipMemcpyDeviceToHost);
  }
  printf("results %s\n", (h[0][0] == h[1][0] && h[0][1] == h[1][1] && h[0][3] == h[1][3]) ? "identical" : "DIFFER");
  return 0;
}
