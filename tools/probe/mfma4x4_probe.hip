// Layout probe for v_mfma_f32_4x4x1_16b_f32 with the A-broadcast (cbsz / abid) and B lane-group (blgp) modifiers:
// prints, for chosen modifier settings, which (lane -> a, lane -> b) pairs each output element multiplied.
// a[lane] = 1000 + lane, b[lane] = 1 + lane / 1000.0 would be ambiguous; use powers: a = lane + 1, b = 2^-k trick is
// overkill -- run twice with one-hot inputs instead.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID, int BLGP>
__global__ void probe(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, CBSZ, ABID, BLGP);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

template <int CBSZ, int ABID, int BLGP>
void run(const char* name) {
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  float ha[64], hb[64], hd[256];
  // a[lane] = lane + 1 (1..64), b[lane] = 100 * (lane + 1): product identifies the pair (a_lane, b_lane) uniquely
  for (int i = 0; i < 64; ++i) { ha[i] = (float)(i + 1); hb[i] = 100.f * (i + 1); }
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((probe<CBSZ, ABID, BLGP>), dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  printf("== %s (cbsz %d abid %d blgp %d): out[lane][vgpr] = a_lane x b_lane\n", name, CBSZ, ABID, BLGP);
  for (int l = 0; l < 64; ++l) {
    if (!(l < 8 || (l >= 28 && l < 40) || l >= 60)) continue;
    printf("  lane %2d:", l);
    for (int r = 0; r < 4; ++r) {
      const long v = (long)(hd[l * 4 + r] + 0.5f);
      const long bl = v / 100;          // v = a * 100 * bl  -> find (a, bl) with a in 1..64, bl in 1..64
      int fa = -1, fb = -1;
      for (int x = 1; x <= 64 && fa < 0; ++x) if (bl % x == 0 && bl / x >= 1 && bl / x <= 64 && v == (long)x * 100 * (bl / x)) {
        // ambiguous factorisations exist; prefer pairs in the same 4-lane block or documented broadcast targets: print all
      }
      printf(" %8ld", v);
    }
    printf("\n");
  }
  hipFree(a); hipFree(b); hipFree(d);
}

int main() {
  run<0, 0, 0>("plain");
  run<3, 2, 0>("A of block 2 broadcast to its group of 8 blocks");
  run<3, 5, 0>("A of block 5 broadcast");
  run<0, 0, 1>("blgp 1");
  run<0, 0, 2>("blgp 2");
  run<3, 2, 1>("cbsz 3 abid 2 blgp 1");
  return 0;
}
