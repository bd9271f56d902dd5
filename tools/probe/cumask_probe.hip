// Which XCDs does a kernel land on when launched on a stream created with a CU mask -- eagerly, and when the launch was
// captured into a hipGraph and replayed?  build: hipcc -O2 --offload-arch=gfx950 tools/probe/cumask_probe.hip -o tools/probe/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void where(int* xcc, int* cu) {
  if (threadIdx.x == 0) {
    int x, c;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 8, 4)" : "=s"(c));
    xcc[blockIdx.x] = x; cu[blockIdx.x] = c;
  }
  // keep the CU busy a little so blocks spread
  float v = threadIdx.x;
  for (int i = 0; i < 20000; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.f) xcc[0] = -1;
}

static void report(const char* what, int* dx, int n) {
  std::vector<int> h(n);
  hipMemcpy(h.data(), dx, n * 4, hipMemcpyDeviceToHost);
  int cnt[16] = {0};
  for (int v : h) if (v >= 0 && v < 16) cnt[v]++;
  printf("%-44s blocks per XCC:", what);
  for (int i = 0; i < 8; ++i) printf(" %4d", cnt[i]);
  printf("\n");
}

int main() {
  const int n = 2048;
  int *dx, *dc;
  CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dc, n * 4));
  hipStream_t s0; CK(hipStreamCreate(&s0));
  hipLaunchKernelGGL(where, dim3(n), dim3(256), 0, s0, dx, dc); CK(hipStreamSynchronize(s0));
  report("plain stream", dx, n);
  for (int variant = 0; variant < 3; ++variant) {
    uint32_t mask[8] = {0};
    const char* name;
    if (variant == 0) { for (int i = 0; i < 4; ++i) mask[i] = 0xffffffffu; name = "mask = low 128 bits"; }
    else if (variant == 1) { for (int i = 4; i < 8; ++i) mask[i] = 0xffffffffu; name = "mask = high 128 bits"; }
    else { for (int i = 0; i < 8; ++i) mask[i] = 0xf0f0f0f0u; name = "mask = bits with (i % 8) >= 4"; }
    hipStream_t sm;
    hipError_t e = hipExtStreamCreateWithCUMask(&sm, 8, mask);
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); continue; }
    CK(hipMemset(dx, 0xff, n * 4));
    hipLaunchKernelGGL(where, dim3(n), dim3(256), 0, sm, dx, dc); CK(hipStreamSynchronize(sm));
    char buf[128]; snprintf(buf, sizeof buf, "%s, eager", name); report(buf, dx, n);
    // captured into a graph and replayed on a plain stream
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(sm, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(where, dim3(n), dim3(256), 0, sm, dx, dc);
    CK(hipStreamEndCapture(sm, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipMemset(dx, 0xff, n * 4));
    CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
    snprintf(buf, sizeof buf, "%s, graph on plain stream", name); report(buf, dx, n);
    CK(hipMemset(dx, 0xff, n * 4));
    CK(hipGraphLaunch(ge, sm)); CK(hipStreamSynchronize(sm));
    snprintf(buf, sizeof buf, "%s, graph on masked stream", name); report(buf, dx, n);
  }
  return 0;
}
