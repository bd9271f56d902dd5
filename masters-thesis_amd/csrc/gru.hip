// Fused GRU cell step (forward and backward), keras GRU v2 semantics (reset_after=True, sigmoid / tanh):
// the decoder of ThinkAndTell/att_model.py (self.gru, att_model.py:84-93, called at :118).
//
//   rec = h_prev @ Uk + b_r                                  (recurrent matmul, 64 x U x 3U per step)
//   z = sigmoid(xz_z + rec_z);  r = sigmoid(xz_r + rec_r);  hh = tanh(xz_h + r * rec_h)
//   h = z * h_prev + (1 - z) * hh                            (xz = x @ W + b_i comes from one GEMM over all steps)
//
// Same machinery as the LSTM step (lstm.hip): workgroup = 16 batch rows x 16 units, 8 waves split the contraction
// axis and combine through LDS, weights streamed straight into VGPRs for v_mfma_f32_16x16x4_f32.  The three gates of
// a unit share the LSTM's gate-interleaved layout [k][U][4] (slots z, r, h, 0): one 16-byte load per unit, the gate
// math in one lane, the fourth slot is a zero pad (weights, gradients and saved values).
// Saved for backward per unit: (z, r, hh, rec_h).
#include "tnt_common.h"

namespace {

constexpr int NW = 8;

__device__ __forceinline__ float4 ld4g(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

struct GruFwdArgs {
  const float* xz; const float* h_prev; const float* Uk; const float* br; float* h; float* gates; int B, U;
};

__global__ __launch_bounds__(512) void gru_fwd_kernel(GruFwdArgs a) {
  __shared__ float red[NW][3][16][17];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, lr = lane & 15;
  const int ub = blockIdx.x, rb = blockIdx.y;
  const int U = a.U, B = a.B;
  const int nchunk = (U + 63) / 64;
  const int arow = rb * 16 + lr, ucol = ub * 16 + lr;
  floatx4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = (floatx4){0.f, 0.f, 0.f, 0.f};
  // epilogue operands, fetched under the weight stream
  const int erow = tid >> 4, ecol = tid & 15;
  const int eb = rb * 16 + erow, eu = ub * 16 + ecol;
  const bool eok = tid < 256 && eb < B;
  const long ee = (long)eb * U + eu;
  float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = x4;
  float hp = 0.f;
  if (eok) {
    x4 = *reinterpret_cast<const float4*>(a.xz + ee * 4);
    b4 = *reinterpret_cast<const float4*>(a.br + (long)eu * 4);
    hp = a.h_prev[ee];
  }
  for (int ci = w; ci < nchunk; ci += NW) {
    const int kbase = ci * 64 + kq * 16;          // U % 16 == 0: a 16-run never straddles U
    const bool kok = kbase < U;                   // lanes past U feed zeros (MFMA runs wave-wide, no divergence)
    float av[16];
    float4 bv[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = ld4g(a.h_prev + (long)arow * U + kbase + 4 * j, kok && arow < B);
      av[4 * j + 0] = t.x; av[4 * j + 1] = t.y; av[4 * j + 2] = t.z; av[4 * j + 3] = t.w;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) bv[s] = ld4g(a.Uk + ((long)(kbase + s) * U + ucol) * 4, kok);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].z, acc[2], 0, 0, 0);
    }
  }
  // C/D map of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][g][kq * 4 + r][lr] = acc[g][r];
  __syncthreads();
  if (eok) {
    float rec[3] = {b4.x, b4.y, b4.z};
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) s += red[k][g][erow][ecol];
      rec[g] += s;
    }
    const float z = tnt_sigmoid_fast(x4.x + rec[0]), r = tnt_sigmoid_fast(x4.y + rec[1]);
    const float hh = tnt_tanh(x4.z + r * rec[2]);
    a.h[ee] = z * hp + (1.f - z) * hh;
    *reinterpret_cast<float4*>(a.gates + ee * 4) = make_float4(z, r, hh, rec[2]);
  }
}

struct GruBwdArgs {
  const float* drec_next; const float* Uk; const float* dh_pass_in; const float* dh_ext; const float* gates;
  const float* h_prev; float* dxz; float* drec; float* dh_pass_out; int B, U;
};

// dh = dh_ext + dh_pass_in + drec_next @ Uk^T; cell backward -> dxz (input side), drec (recurrent side),
// dh_pass_out = dh * z (the direct path to h_prev; the matmul path is applied by the next call through drec).
__global__ __launch_bounds__(512) void gru_bwd_kernel(GruBwdArgs a) {
  __shared__ float red[NW][16][17];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, lr = lane & 15;
  const int ub = blockIdx.x, rb = blockIdx.y;
  const int B = a.B, K = 4 * a.U;
  floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
  const int erow = tid >> 4, ecol = tid & 15;
  const int eb = rb * 16 + erow, eu = ub * 16 + ecol;
  const bool eok = tid < 256 && eb < B;
  const long e = (long)eb * a.U + eu;
  float dh0 = 0.f, hp = 0.f;
  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (eok) {
    if (a.dh_pass_in) dh0 += a.dh_pass_in[e];
    if (a.dh_ext) dh0 += a.dh_ext[e];
    g4 = *reinterpret_cast<const float4*>(a.gates + e * 4);
    hp = a.h_prev[e];
  }
  if (a.drec_next) {
    const int arow = rb * 16 + lr, ucol = ub * 16 + lr;
    const int nchunk = K / 64;
    for (int c0 = w; c0 < nchunk; c0 += NW * 4) {
      float4 ta[4][4], tb[4][4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int ci = c0 + cc * NW;
        const int kbase = ci * 64 + kq * 4;
        const bool cok = ci < nchunk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ta[cc][j] = ld4g(a.drec_next + (long)arow * K + kbase + 16 * j, cok && arow < B);
          tb[cc][j] = ld4g(a.Uk + (long)ucol * K + kbase + 16 * j, cok);
        }
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[cc][j].x, tb[cc][j].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[cc][j].y, tb[cc][j].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[cc][j].z, tb[cc][j].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[cc][j].w, tb[cc][j].w, acc, 0, 0, 0);
        }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][kq * 4 + r][lr] = acc[r];
  __syncthreads();
  if (eok) {
    float dh = dh0;
#pragma unroll
    for (int k = 0; k < NW; ++k) dh += red[k][erow][ecol];
    const float z = g4.x, r = g4.y, hh = g4.z, rech = g4.w;
    const float dz = dh * (hp - hh);
    const float dah = dh * (1.f - z) * (1.f - hh * hh);        // pre-activation of hh
    const float daz = dz * z * (1.f - z);
    const float dar = dah * rech * r * (1.f - r);
    *reinterpret_cast<float4*>(a.dxz + e * 4) = make_float4(daz, dar, dah, 0.f);
    *reinterpret_cast<float4*>(a.drec + e * 4) = make_float4(daz, dar, dah * r, 0.f);
    if (a.dh_pass_out) a.dh_pass_out[e] = dh * z;
  }
}

}  // namespace

extern "C" int32_t tnt_gru_step_fwd_f32(const float* xz, const float* h_prev, const float* Uk, const float* br, float* h,
                                        float* gates, int32_t B, int32_t U, void* stream) {
  if (U <= 0 || U % 16 != 0) return TNT_BADARG(8);
  if (B <= 0) return TNT_BADARG(7);
  if (h == h_prev) return TNT_BADARG(5);
  GruFwdArgs a{xz, h_prev, Uk, br, h, gates, B, U};
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(U / 16, (B + 15) / 16), dim3(512), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_gru_step_bwd_f32(const float* drec_next, const float* Uk, const float* dh_pass_in,
                                        const float* dh_ext, const float* gates, const float* h_prev, float* dxz,
                                        float* drec, float* dh_pass_out, int32_t B, int32_t U, void* stream) {
  if (U <= 0 || U % 16 != 0) return TNT_BADARG(11);
  if (B <= 0) return TNT_BADARG(10);
  if (drec == drec_next) return TNT_BADARG(8);
  GruBwdArgs a{drec_next, Uk, dh_pass_in, dh_ext, gates, h_prev, dxz, drec, dh_pass_out, B, U};
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(U / 16, (B + 15) / 16), dim3(512), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}
