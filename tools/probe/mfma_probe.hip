// Micro-probe: what does a wave sustain on v_mfma_f32_32x32x2_f32 with operands (a) in registers,
// (b) fetched from LDS with one ds_read_b32 per operand per MFMA (the GEMM inner loop)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  __shared__ float lds[2 * 32 * 136];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2 * 32 * 136; i += 256) lds[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  floatx16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = lane * 0.001f, b = lane * 0.002f;
  const float* base = lds + (lane >> 5) * 136 + (lane & 31) + (threadIdx.x >> 6) * 32;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      if (MODE == 1) { a = base[kk * 136]; b = base[32 * 136 + kk * 136]; }
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    if (MODE == 1) asm volatile("" ::: "memory");
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char* name, int blocks, float* d) {
  const int iters = 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, NACC><<<blocks, 256>>>(d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) probe<MODE, NACC><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 10.0 * blocks * 4 * iters * 16 * NACC * 2.0 * 32 * 32 * 2;
  printf("%-28s blocks=%5d acc=%d : %7.1f TF\n", name, blocks, NACC, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float* d; hipMalloc(&d, 8192 * 256 * 4);
  for (int blocks : {256, 512, 1024, 2048}) {
    run<0, 1>("regs, 1 acc chain", blocks, d);
    run<0, 4>("regs, 4 acc", blocks, d);
    run<1, 1>("lds b32 x2 per mfma, 1 acc", blocks, d);
    run<1, 4>("lds b32 x2 per 4 mfma, 4 acc", blocks, d);
  }
  return 0;
}
