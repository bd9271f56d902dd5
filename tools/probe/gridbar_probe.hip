// Cost of a software grid barrier (atomic counter + fences) on MI355X for a persistent LSTM-style kernel:
// G workgroups x 512 threads, each iteration writes a 64x512 float "h" slice, barrier, reads all of h.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(ctr, 1u);
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { *err = 1u; ok = false; break; }
    }
    __threadfence();
  }
  __syncthreads();
  return ok;
}

template <int MODE>   // 0: barrier only; 1: + write/read h through global memory
__global__ __launch_bounds__(512) void k(unsigned* ctr, unsigned* err, float* h, float* sink, int iters, int G) {
  const int tid = threadIdx.x, wg = blockIdx.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    float* hw = h + (size_t)(it & 1) * 64 * 512;
    if (MODE == 1) {
      // this WG owns 256 floats of h (G = 128): rows (wg/32)*16.., units (wg%32)*16..
      if (tid < 256) hw[(size_t)((wg >> 5) * 16 + (tid >> 4)) * 512 + (wg & 31) * 16 + (tid & 15)] = (float)(it + tid) * 1e-3f + acc * 1e-9f;
    }
    grid_barrier(ctr, (unsigned)(it + 1) * G, err);
    if (MODE == 1) {
      // every WG reads its 16 rows x 512 of h (32 KB), like the recurrent matmul operand
      const float4* src = reinterpret_cast<const float4*>(hw + (size_t)(wg >> 5) * 16 * 512);
      for (int i = tid; i < 16 * 512 / 4; i += 512) { const float4 t = src[i]; acc += t.x + t.y + t.z + t.w; }
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int MODE>
int run(int G, int iters) {
  unsigned *ctr, *err; float *h, *sink;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&h, 2 * 64 * 512 * 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(err, 0, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(ctr, 0, 4));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(512), 0, 0, ctr, err, h, sink, iters, G);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  unsigned e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
  printf("G=%3d mode=%d iters=%d : %.2f us per iteration (err=%u)\n", G, MODE, iters, best * 1e3f / iters, e);
  CK(hipFree(ctr)); CK(hipFree(err)); CK(hipFree(h)); CK(hipFree(sink));
  return 0;
}

int main() {
  for (int G : {32, 128, 256}) { if (run<0>(G, 64)) return 1; }
  if (run<1>(128, 64)) return 1;
  if (run<1>(128, 16)) return 1;
  return 0;
}
