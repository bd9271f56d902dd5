// Where do the 25 us of the row-softmax go?  Variants of a one-WG-per-row kernel over [960][5004] floats.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* x, float* y, int V, int ld) {
  __shared__ float sh[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (long)row * ld;
  float4 v[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int j = 4 * (tid + 256 * i);
    v[i] = make_float4(-1e30f, -1e30f, -1e30f, -1e30f);
    if (j + 3 < V) v[i] = *reinterpret_cast<const float4*>(xr + j);
  }
  float m = 0.f, s = 0.f;
  if (MODE >= 1) {          // block max
    m = -1e30f;
#pragma unroll
    for (int i = 0; i < 5; ++i) m = fmaxf(fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)), m);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0) sh[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
  }
  if (MODE >= 2) {          // exp + block sum
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (MODE == 3) { v[i].x = __expf(v[i].x - m); v[i].y = __expf(v[i].y - m); v[i].z = __expf(v[i].z - m); v[i].w = __expf(v[i].w - m); }
      else { v[i].x = expf(v[i].x - m); v[i].y = expf(v[i].y - m); v[i].z = expf(v[i].z - m); v[i].w = expf(v[i].w - m); }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) sh[tid >> 6] = s;
    __syncthreads();
    s = 1.f / (sh[0] + sh[1] + sh[2] + sh[3]);
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int j = 4 * (tid + 256 * i);
    if (j + 3 < V) {
      float4 o = v[i];
      if (MODE >= 2) { o.x *= s; o.y *= s; o.z *= s; o.w *= s; }
      *reinterpret_cast<float4*>(y + (long)row * ld + j) = o;
    }
  }
}

template <int MODE>
float run(const float* x, float* y, int rows, int V, int ld) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(rows), dim3(256), 0, 0, x, y, V, ld);
  hipEventRecord(a);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<MODE>, dim3(rows), dim3(256), 0, 0, x, y, V, ld);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 50 * 1e3f;
}

int main() {
  const int rows = 960, V = 5001;
  for (int ld : {5004, 5120}) {
    float *x, *y;
    CK(hipMalloc(&x, (size_t)rows * ld * 4)); CK(hipMalloc(&y, (size_t)rows * ld * 4));
    std::vector<float> h((size_t)rows * ld);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 250.f - 2.f;
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("ld=%d  copy %.2f us | +max %.2f | +expf,sum %.2f | +__expf,sum %.2f\n", ld, run<0>(x, y, rows, V, ld),
           run<1>(x, y, rows, V, ld), run<2>(x, y, rows, V, ld), run<3>(x, y, rows, V, ld));
    hipFree(x); hipFree(y);
  }
  return 0;
}
