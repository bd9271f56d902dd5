// Additive (Bahdanau / Show-Attend-and-Tell) attention step, forward and backward.
//
// Reference: attention.Attention.call (AttemptFour/Model/attention.py:25-44) called once
// per timestep from lc_NIC.call_attention (lc_NIC.py:244-256) and
// greedy_predict_attention (lc_NIC.py:607-611).  P = LeakyReLU(W1 F + b1) is
// loop-invariant (the reference recomputes it every step) and is produced once by
// tnt_gemm_f32.  One workgroup per batch row: the R x A score tile and the softmax over
// the R regions live in LDS / registers, reductions are wave shuffles; nothing but
// qpre[A] and alpha[R] is saved for the backward (tanh is recomputed, the dropout mask is
// regenerated from the Philox stream).
#include "tnt_common.h"
#include "tnt_rng.h"

namespace {

constexpr int MAXR = 2048;   // regions per sample held in LDS
constexpr int MAXU = 4096;

__device__ __forceinline__ float block_sum256(float v, float* sh) {
  v = tnt_wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_max256(float v, float* sh) {
  v = tnt_wave_max(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  return r;
}
// sum over the AP-lane group (AP = 32 or 64) a lane belongs to
template <int AP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = AP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct AttArgs {
  const float* h; const float* F; const float* P; const float* W2; const float* b2; const float* v; const float* bv;
  float* qpre; float* alpha; float* ctx; float* ctx_d; float* s_out;
  // backward
  const float* dctx_d; const float* dz; const float* Wc; const float* qpre_in; const float* alpha_in;
  const float* dctx_part; int nparts;     // [nparts][B][D] partial context gradients from tnt_lstm_step_bwd_f32
  float* dP; float* dF; float* dvb;
  float* dqpre; float* dh;
  int B, R, D, A, U, in_lwidth;
  float slope, rate_attn, rate_in;
  uint64_t seed; uint32_t site_attn, site_in, step; const uint32_t* step_dev;
  const uint8_t* keep4;       // nullable: attention-dropout keep bits from tnt_dropout_mask4_u8 (wide kernels only)
  // backward only: dalpha[r] += alpha_mse * (alpha[r] - 1) -- the gradient of c * sum (1 - alpha)^2 with alpha_mse = 2c
  // (lc_NIC.train_step_sam adds MSE(ones, attention_scores) to the loss of its first pass, lc_NIC.py:751-752,766)
  float alpha_mse;
  // backward only: 1 = the accumulators dP / dF / dvb are written, not added to (the first executed step of a chain: saves
  // the caller a zero fill of the three buffers)
  int fresh;
};

// q[a] = LeakyReLU(sum_k h[k] W2[k][a] + b2[a]); threads = (a = tid % AP, part = tid / AP)
template <int AP>
__device__ __forceinline__ void compute_q(const AttArgs& g, const float* hs, float* part, float* qs, float* qpre_s) {
  const int a = threadIdx.x % AP, kp = threadIdx.x / AP;
  constexpr int KP = 256 / AP;
  float s = 0.f;
  if (a < g.A) {
#pragma unroll 8
    for (int k = kp; k < g.U; k += KP) s += hs[k] * g.W2[(long)k * g.A + a];
  }
  part[kp * AP + a] = s;
  __syncthreads();
  if (threadIdx.x < AP && a < g.A) {
    float t = g.b2[a];
    for (int j = 0; j < KP; ++j) t += part[j * AP + a];
    qpre_s[a] = t;
    qs[a] = t > 0.f ? t : t * g.slope;
  }
  __syncthreads();
}

template <int AP>
__global__ __launch_bounds__(256) void attention_fwd_kernel(AttArgs g) {
  __shared__ float hs[MAXU];
  __shared__ float es[MAXR];
  __shared__ float part[256];
  __shared__ float qs[AP], qpre_s[AP], red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint32_t step = g.step + (g.step_dev ? g.step_dev[0] : 0u);
  for (int k = tid; k < g.U; k += 256) hs[k] = g.h[(long)b * g.U + k];
  __syncthreads();
  compute_q<AP>(g, hs, part, qs, qpre_s);
  if (tid < g.A) g.qpre[(long)b * g.A + tid] = qpre_s[tid];

  // scores
  const int a = tid % AP, rp = tid / AP;
  constexpr int RP = 256 / AP;
  const float scale_a = 1.f / (1.f - g.rate_attn);
  const float va = a < g.A ? g.v[a] : 0.f;
  const float bv = g.bv[0];
  for (int r0 = 0; r0 < g.R; r0 += RP * 4) {
    float pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                       // 4 independent loads in flight
      const int r = r0 + u * RP + rp;
      pv[u] = (r < g.R && a < g.A) ? g.P[((long)b * g.R + r) * g.A + a] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * RP + rp;
      float t = 0.f;
      if (r < g.R && a < g.A) {
        const long e = ((long)b * g.R + r) * g.A + a;
        float s = tnt_tanh(pv[u] + qs[a]);
        if (g.rate_attn > 0.f) s = tnt_keep((uint64_t)e, g.rate_attn, g.seed, g.site_attn, step) ? s * scale_a : 0.f;
        if (g.s_out) g.s_out[e] = s;
        t = s * va;
      }
      t = group_sum<AP>(t);
      if (a == 0 && r < g.R) es[r] = t + bv;
    }
  }
  __syncthreads();
  // softmax over regions (keras Softmax(axis=1), attention.py:16,40)
  float m = -INFINITY;
  for (int r = tid; r < g.R; r += 256) m = fmaxf(m, es[r]);
  m = block_max256(m, red);
  float z = 0.f;
  for (int r = tid; r < g.R; r += 256) { const float ex = expf(es[r] - m); es[r] = ex; z += ex; }
  z = block_sum256(z, red);
  const float invz = 1.f / z;
  for (int r = tid; r < g.R; r += 256) { const float al = es[r] * invz; es[r] = al; g.alpha[(long)b * g.R + r] = al; }
  __syncthreads();
  // context = sum_r alpha[r] F[b][r][:]   threads = (d = tid % DP, part = tid / DP), DP = AP
  const int d = tid % AP, pp = tid / AP;
  float c = 0.f;
  if (d < g.D) {
#pragma unroll 8
    for (int r = pp; r < g.R; r += RP) c += es[r] * g.F[((long)b * g.R + r) * g.D + d];
  }
  part[pp * AP + d] = c;
  __syncthreads();
  if (tid < AP && tid < g.D) {
    float t = 0.f;
    for (int j = 0; j < RP; ++j) t += part[j * AP + tid];
    g.ctx[(long)b * g.D + tid] = t;
    if (g.ctx_d) {
      float td = t;
      if (g.rate_in > 0.f)
        td = tnt_keep((uint64_t)b * (uint64_t)g.in_lwidth + tid, g.rate_in, g.seed, g.site_in, step)
                 ? t * (1.f / (1.f - g.rate_in)) : 0.f;
      g.ctx_d[(long)b * g.D + tid] = td;
    }
  }
}

template <int AP>
__global__ __launch_bounds__(256) void attention_bwd_kernel(AttArgs g) {
  __shared__ float als[MAXR];      // alpha, then de
  __shared__ float das[MAXR];      // dalpha
  __shared__ float part[256], part2[256];
  __shared__ float qs[AP], dcs[AP], dq_s[AP], red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint32_t step = g.step + (g.step_dev ? g.step_dev[0] : 0u);
  constexpr int RP = 256 / AP;
  if (g.dctx_part) {
    if (tid < g.D) {
      float t = 0.f;
      for (int p = 0; p < g.nparts; ++p) t += g.dctx_part[((long)p * g.B + b) * g.D + tid];
      part[tid] = t;
    }
    __syncthreads();
  } else if (g.dz) {
    // fused: dctx_d[b][d] = sum_n dz[b][n] * Wc[d][n]   (the LSTM input-kernel rows of the context)
    const int K4 = 4 * g.U, w = tid >> 6, lane = tid & 63;
    const float4* dzr = reinterpret_cast<const float4*>(g.dz + (long)b * K4);
    for (int dd = w; dd < g.D; dd += 4) {
      const float4* wr = reinterpret_cast<const float4*>(g.Wc + (long)dd * K4);
      float sacc = 0.f;
#pragma unroll 4
      for (int i = lane; i < K4 / 4; i += 64) {
        const float4 x = dzr[i], y = wr[i];
        sacc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      }
      sacc = tnt_wave_sum(sacc);
      if (lane == 0) part[dd] = sacc;
    }
    __syncthreads();
  }
  if (tid < AP) {
    float q = 0.f, dc = 0.f;
    if (tid < g.A) { q = g.qpre_in[(long)b * g.A + tid]; q = q > 0.f ? q : q * g.slope; }
    if (tid < g.D) {
      dc = (g.dz || g.dctx_part) ? part[tid] : g.dctx_d[(long)b * g.D + tid];
      if (g.rate_in > 0.f)
        dc = tnt_keep((uint64_t)b * (uint64_t)g.in_lwidth + tid, g.rate_in, g.seed, g.site_in, step)
                 ? dc * (1.f / (1.f - g.rate_in)) : 0.f;
    }
    qs[tid] = q; dcs[tid] = dc;
  }
  for (int r = tid; r < g.R; r += 256) als[r] = g.alpha_in[(long)b * g.R + r];
  __syncthreads();
  // dalpha[r] = sum_d dctx[d] F[r][d];  dF[r][d] += alpha[r] dctx[d]
  const int d = tid % AP, rp = tid / AP;
  __syncthreads();          // part[] (fused dctx) fully consumed before it is reused below
  for (int r0 = 0; r0 < g.R; r0 += RP * 4) {
    float fv[4], dfv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * RP + rp;
      const bool ok = r < g.R && d < g.D;
      const long e = ((long)b * g.R + r) * g.D + d;
      fv[u] = ok ? g.F[e] : 0.f;
      dfv[u] = (ok && !g.fresh) ? g.dF[e] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * RP + rp;
      float t = 0.f;
      if (r < g.R && d < g.D) {
        const long e = ((long)b * g.R + r) * g.D + d;
        t = dcs[d] * fv[u];
        g.dF[e] = dfv[u] + als[r] * dcs[d];
      }
      t = group_sum<AP>(t);
      if (d == 0 && r < g.R) das[r] = t + g.alpha_mse * (als[r] - 1.f);
    }
  }
  __syncthreads();
  float dot = 0.f;
  for (int r = tid; r < g.R; r += 256) dot += als[r] * das[r];
  dot = block_sum256(dot, red);
  float dbv = 0.f;
  for (int r = tid; r < g.R; r += 256) { const float de = als[r] * (das[r] - dot); als[r] = de; dbv += de; }
  dbv = block_sum256(dbv, red);     // also orders the als[] writes before the reads below
  // through e = s_d . v, dropout, tanh
  const int a = d;
  const float scale_a = 1.f / (1.f - g.rate_attn);
  const float va = a < g.A ? g.v[a] : 0.f;
  float dv = 0.f, dq = 0.f;
  if (a < g.A) {
    for (int r0 = rp; r0 < g.R; r0 += RP * 4) {
      float pv[4], dpv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * RP;
        const long e = ((long)b * g.R + r) * g.A + a;
        pv[u] = r < g.R ? g.P[e] : 0.f;
        dpv[u] = (r < g.R && !g.fresh) ? g.dP[e] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * RP;
        if (r >= g.R) continue;
        const long e = ((long)b * g.R + r) * g.A + a;
        const float s = tnt_tanh(pv[u] + qs[a]);
        bool keep = true;
        if (g.rate_attn > 0.f) keep = tnt_keep((uint64_t)e, g.rate_attn, g.seed, g.site_attn, step);
        const float ks = keep ? (g.rate_attn > 0.f ? scale_a : 1.f) : 0.f;
        const float de = als[r];
        dv += s * ks * de;
        const float dsum = de * va * ks * (1.f - s * s);
        g.dP[e] = dpv[u] + dsum;
        dq += dsum;
      }
    }
  }
  part[rp * AP + a] = dv; part2[rp * AP + a] = dq;
  __syncthreads();
  if (tid < AP && tid < g.A) {
    float tv = 0.f, tq = 0.f;
    for (int j = 0; j < RP; ++j) { tv += part[j * AP + tid]; tq += part2[j * AP + tid]; }
    g.dvb[(long)b * (g.A + 1) + tid] = (g.fresh ? 0.f : g.dvb[(long)b * (g.A + 1) + tid]) + tv;
    const float qp = g.qpre_in[(long)b * g.A + tid];
    const float dqp = qp > 0.f ? tq : tq * g.slope;
    dq_s[tid] = dqp;
    g.dqpre[(long)b * g.A + tid] = dqp;
  }
  if (tid == 0) g.dvb[(long)b * (g.A + 1) + g.A] = (g.fresh ? 0.f : g.dvb[(long)b * (g.A + 1) + g.A]) + dbv;
  __syncthreads();
  // dh[k] = sum_a dqpre[a] W2[k][a]
  for (int k = tid; k < g.U; k += 256) {
    float t = 0.f;
#pragma unroll 8
    for (int j = 0; j < g.A; ++j) t += dq_s[j] * g.W2[(long)k * g.A + j];
    g.dh[(long)b * g.U + k] = t;
  }
}

// workgroup (t, c): partial[t * nc + c] = sum over its 64 regions r of (1 - sum_b alpha[t][b][r])^2; the 4 waves split the
// batch (independent loads), their partial sums meet in LDS in wave order
__global__ __launch_bounds__(256) void attention_metric_kernel(const float* alpha, float* partial, int T, int B, int R,
                                                               long tstride) {
  __shared__ float sb[4][64];
  const int t = blockIdx.x, c = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = c * 64 + lane;
  float s = 0.f;
  if (r < R) {
#pragma unroll 8
    for (int b = w; b < B; b += 4) s += alpha[(long)t * tstride + (long)b * R + r];
  }
  sb[w][lane] = s;
  __syncthreads();
  if (w != 0) return;
  float acc = 0.f;
  if (r < R) {
    const float tot = ((sb[0][lane] + sb[1][lane]) + sb[2][lane]) + sb[3][lane];
    acc = (1.f - tot) * (1.f - tot);
  }
  acc = tnt_wave_sum(acc);
  if (lane == 0) partial[t * gridDim.y + c] = acc;
}
__global__ void attention_metric_final_kernel(const float* partial, float* out, int n, float scale) {
  float s = 0.f;
  for (int t = threadIdx.x; t < n; t += 64) s += partial[t];
  s = tnt_wave_sum(s);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// ---------------------------------------------------------------------------------------------
// Wide variants (A % 4 == 0, D % 4 == 0): 1024 threads per sample, 16-byte loads, one Philox call per
// 4 mask decisions, every thread's loads of a phase issued before their first use.  The 256-thread
// kernels above remain the general fallback.
// ---------------------------------------------------------------------------------------------
constexpr int WT = 1024;            // threads per workgroup
constexpr int WW = WT / 64;         // waves
constexpr int MAXP = 6;             // row passes held in registers: R <= MAXP * (WT / (A/4))

__device__ __forceinline__ float block_sum_w(float v, float* sh) {
  v = tnt_wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int k = 0; k < WW; ++k) r += sh[k];
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_max_w(float v, float* sh) {
  v = tnt_wave_max(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
#pragma unroll
  for (int k = 1; k < WW; ++k) r = fmaxf(r, sh[k]);
  __syncthreads();
  return r;
}
// sum of v over the lanes of a wave that share (lane % G): xor offsets G, 2G, ..., 32 (DPP / permlane forms, tnt_common.h: exact)
template <int G>
__device__ __forceinline__ float stride_sum(float v) {
  static_assert(G == 8 || G == 16, "lanes per row group");
  if (G == 8) v += tnt_ror<8>(v);
  v = tnt_x16_add(v);
  v = tnt_x32_add(v);
  return v;
}
// sum over a group of G adjacent lanes, in every lane of the group: 1, 2 exact, then the mirrors (each half already uniform)
template <int G>
__device__ __forceinline__ float adj_sum(float v) {
  static_assert(G == 4 || G == 8 || G == 16, "adjacent lanes per group");
  v += tnt_x1(v);
  v += tnt_x2(v);
  if (G >= 8) v += tnt_hmirror(v);
  if (G >= 16) v += tnt_mirror(v);
  return v;
}

// G4 = A/4 (= D/4 after padding to the larger of the two): lanes per row
// LDS of one attention forward step (shared by the per-step kernel and the persistent chain kernel)
struct AttFwdLds {
  float* hs;                 // [MAXU] the sample's h, 16-byte aligned
  float* es;                 // [MAXR]
  float (*wred)[64];         // [WW][64], 16-byte aligned
  float* qs;                 // [64], 16-byte aligned
  float* red;                // [WW]
};

// One attention forward step of sample b, h already in l.hs (and a workgroup barrier behind it).  PERSIST: the caller is the
// persistent chain kernel, whose consumers poll ctx_d for a sentinel -- the thread that publishes ctx_d[b][tid] drains its
// earlier stores (its reset of the NEXT step's element) first.
template <int G4, bool PERSIST, int NP = MAXP>        // NP: row passes held in registers (R <= NP * WT / G4)
__device__ __forceinline__ void att_fwd_body(const AttArgs& g, int b, const AttFwdLds& l, int si = 0, long keep_stride = 0) {
  constexpr int RPP = WT / G4;               // rows per pass
  // step si of a chain (PERSIST): the per-step outputs / masks / dropout sites are those of step 0 advanced by si
  float* const o_qpre = g.qpre + (long)si * g.B * g.A;
  float* const o_alpha = g.alpha + (long)si * g.B * g.R;
  float* const o_ctx = g.ctx + (long)si * g.B * g.D;
  float* const o_ctxd = g.ctx_d ? g.ctx_d + (long)si * g.B * g.D : nullptr;
  const uint8_t* const keep4 = g.keep4 ? g.keep4 + (long)si * keep_stride : nullptr;
  const uint32_t site_attn = g.site_attn + (uint32_t)si, site_in = g.site_in + (uint32_t)si;
  float* hs = l.hs; float* es = l.es; float (*wred)[64] = l.wred; float* qs = l.qs; float* red = l.red;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c4 = tid % G4, rl = tid / G4;    // 4-column group, row lane
  const int A = g.A, D = g.D, R = g.R, U = g.U;
  const uint32_t step = g.step + (g.step_dev ? g.step_dev[0] : 0u);
  // ---- q = LeakyReLU(h W2 + b2): thread (c4, rl) sums rows k = rl, rl+RPP, ...
  {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 * 4 < A) {
#pragma unroll 4
      for (int k = rl; k < U; k += RPP) {
        const float4 wv = *reinterpret_cast<const float4*>(g.W2 + (long)k * A + c4 * 4);
        const float hk = hs[k];
        acc.x += hk * wv.x; acc.y += hk * wv.y; acc.z += hk * wv.z; acc.w += hk * wv.w;
      }
    }
    acc.x = stride_sum<G4>(acc.x); acc.y = stride_sum<G4>(acc.y);
    acc.z = stride_sum<G4>(acc.z); acc.w = stride_sum<G4>(acc.w);
    if (lane < G4) *reinterpret_cast<float4*>(&wred[w][lane * 4]) = acc;
    __syncthreads();
    if (tid < A) {
      float t = g.b2[tid];
#pragma unroll
      for (int k = 0; k < WW; ++k) t += wred[k][tid];
      o_qpre[(long)b * A + tid] = t;
      qs[tid] = t > 0.f ? t : t * g.slope;
    }
    __syncthreads();
  }
  // ---- scores e[r] = sum_a dropout(tanh(P + q)) v + bv
  {
    float4 pv[NP];
    uint32_t mk[NP];
    const bool cok = c4 * 4 < A;
    const bool stored = keep4 != nullptr && g.rate_attn > 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int r = p * RPP + rl;
      pv[p] = (cok && r < R) ? *reinterpret_cast<const float4*>(g.P + ((long)b * R + r) * A + c4 * 4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
      mk[p] = (stored && cok && r < R) ? keep4[(((long)b * R + r) * A + c4 * 4) >> 2] : 0u;
    }
    const float4 q4 = cok ? *reinterpret_cast<const float4*>(&qs[c4 * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v4 = cok ? *reinterpret_cast<const float4*>(g.v + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float scale_a = 1.f / (1.f - g.rate_attn), bv = g.bv[0];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int r = p * RPP + rl;
      float t = 0.f;
      if (cok && r < R) {
        const long e = ((long)b * R + r) * A + c4 * 4;
        float s0 = tnt_tanh(pv[p].x + q4.x), s1 = tnt_tanh(pv[p].y + q4.y), s2 = tnt_tanh(pv[p].z + q4.z), s3 = tnt_tanh(pv[p].w + q4.w);
        if (g.rate_attn > 0.f) {
          bool k[4];
          if (stored) { k[0] = mk[p] & 1u; k[1] = mk[p] & 2u; k[2] = mk[p] & 4u; k[3] = mk[p] & 8u; }
          else tnt_keep4((uint64_t)e, g.rate_attn, g.seed, site_attn, step, k);
          s0 = k[0] ? s0 * scale_a : 0.f; s1 = k[1] ? s1 * scale_a : 0.f;
          s2 = k[2] ? s2 * scale_a : 0.f; s3 = k[3] ? s3 * scale_a : 0.f;
        }
        if (g.s_out) *reinterpret_cast<float4*>(g.s_out + e) = make_float4(s0, s1, s2, s3);
        t = s0 * v4.x + s1 * v4.y + s2 * v4.z + s3 * v4.w;
      }
      t = adj_sum<G4>(t);
      if (c4 == 0 && r < R) es[r] = t + bv;
    }
    __syncthreads();
  }
  // ---- softmax over regions
  float m = -INFINITY;
  for (int r = tid; r < R; r += WT) m = fmaxf(m, es[r]);
  m = block_max_w(m, red);
  float z = 0.f;
  for (int r = tid; r < R; r += WT) { const float ex = expf(es[r] - m); es[r] = ex; z += ex; }
  z = block_sum_w(z, red);
  const float invz = 1.f / z;
  for (int r = tid; r < R; r += WT) { const float al = es[r] * invz; es[r] = al; o_alpha[(long)b * R + r] = al; }
  __syncthreads();
  // ---- context = sum_r alpha[r] F[r][:]
  {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool cok = c4 * 4 < D;
    float4 fv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int r = p * RPP + rl;
      fv[p] = (cok && r < R) ? *reinterpret_cast<const float4*>(g.F + ((long)b * R + r) * D + c4 * 4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int r = p * RPP + rl;
      const float al = r < R ? es[r] : 0.f;
      acc.x += al * fv[p].x; acc.y += al * fv[p].y; acc.z += al * fv[p].z; acc.w += al * fv[p].w;
    }
    acc.x = stride_sum<G4>(acc.x); acc.y = stride_sum<G4>(acc.y);
    acc.z = stride_sum<G4>(acc.z); acc.w = stride_sum<G4>(acc.w);
    if (lane < G4) *reinterpret_cast<float4*>(&wred[w][lane * 4]) = acc;
    __syncthreads();
    if (tid < D) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < WW; ++k) t += wred[k][tid];
      o_ctx[(long)b * D + tid] = t;
      if (o_ctxd) {
        float td = t;
        if (g.rate_in > 0.f)
          td = tnt_keep((uint64_t)b * (uint64_t)g.in_lwidth + tid, g.rate_in, g.seed, site_in, step)
                   ? t * (1.f / (1.f - g.rate_in)) : 0.f;
        if (PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        o_ctxd[(long)b * D + tid] = td;
      }
    }
  }
}

template <int G4>
__global__ __launch_bounds__(WT) void attention_fwd_wide_kernel(AttArgs g) {
  __shared__ __attribute__((aligned(16))) float hs[MAXU];
  __shared__ float es[MAXR];
  __shared__ __attribute__((aligned(16))) float wred[WW][64];
  __shared__ __attribute__((aligned(16))) float qs[64];
  __shared__ float red[WW];
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < g.U; k += WT) hs[k] = g.h[(long)b * g.U + k];
  __syncthreads();
  att_fwd_body<G4, false>(g, b, AttFwdLds{hs, es, wred, qs, red});
}

template <int G4>
__global__ __launch_bounds__(WT) void attention_bwd_wide_kernel(AttArgs g) {
  constexpr int RPP = WT / G4;
  __shared__ float als[MAXR];      // alpha, then de
  __shared__ float das[MAXR];      // dalpha
  __shared__ __attribute__((aligned(16))) float wred[WW][64], wred2[WW][64];
  __shared__ __attribute__((aligned(16))) float qs[64], dcs[64], dq_s[64];
  __shared__ float red[WW];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c4 = tid % G4, rl = tid / G4;
  const int A = g.A, D = g.D, R = g.R, U = g.U;
  const uint32_t step = g.step + (g.step_dev ? g.step_dev[0] : 0u);
  // ---- dctx (sum of the LSTM backward's per-unit-block partials, or fused dz Wc^T, or given), un-dropped
  if (g.dctx_part) {
    // thread (p, d) holds one partial (np*D <= 1024 by the launch check).  Fixed-order tree: when a wave holds whole
    // rows of D = 32 values, lanes l and l^32 are two parts of the same d and are added by a shuffle first, so each wave
    // leaves D sums; a last pass of D threads adds the per-wave rows.  (Other D: per-part rows straight to LDS.)
    float* scratch = &wred[0][0];                       // [WW][64] floats, free here
    const int np = g.nparts;
    float v = tid < np * D ? g.dctx_part[((long)(tid / D) * g.B + b) * D + tid % D] : 0.f;
    if (D == 32) {
      v += __shfl_xor(v, 32, 64);
      if (lane < 32) scratch[w * 32 + lane] = v;        // wave w covers parts 2w, 2w+1 (zero past np)
      __syncthreads();
      if (tid < 32) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < WW; ++k) t += scratch[k * 32 + tid];
        dcs[tid] = t;
      }
    } else {
      scratch[tid] = v;
      __syncthreads();
      if (tid < D) {
        float t = 0.f;
        for (int p = 0; p < np; ++p) t += scratch[p * D + tid];
        dcs[tid] = t;
      }
    }
  } else if (g.dz) {
    const int K4 = 4 * U;
    const float4* dzr = reinterpret_cast<const float4*>(g.dz + (long)b * K4);
    for (int dd = w; dd < D; dd += WW) {
      const float4* wr = reinterpret_cast<const float4*>(g.Wc + (long)dd * K4);
      float sacc = 0.f;
#pragma unroll 8
      for (int i = lane; i < K4 / 4; i += 64) {
        const float4 x = dzr[i], y = wr[i];
        sacc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      }
      sacc = tnt_wave_sum(sacc);
      if (lane == 0) dcs[dd] = sacc;
    }
  } else if (tid < D) {
    dcs[tid] = g.dctx_d[(long)b * D + tid];
  }
  if (tid < 64) {
    float q = 0.f;
    if (tid < A) { q = g.qpre_in[(long)b * A + tid]; q = q > 0.f ? q : q * g.slope; }
    qs[tid] = q;
  }
  for (int r = tid; r < R; r += WT) als[r] = g.alpha_in[(long)b * R + r];
  __syncthreads();
  if (tid < 64) {
    float dc = tid < D ? dcs[tid] : 0.f;
    if (tid < D && g.rate_in > 0.f)
      dc = tnt_keep((uint64_t)b * (uint64_t)g.in_lwidth + tid, g.rate_in, g.seed, g.site_in, step)
               ? dc * (1.f / (1.f - g.rate_in)) : 0.f;
    dcs[tid] = dc;
  }
  __syncthreads();
  // ---- dalpha[r] = dctx . F[r];  dF[r] += alpha[r] dctx
  {
    const bool cok = c4 * 4 < D;
    const float4 dc4 = cok ? *reinterpret_cast<const float4*>(&dcs[c4 * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 fv[MAXP], dfv[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int r = p * RPP + rl;
      const bool ok = cok && r < R;
      const long e = ((long)b * R + r) * D + c4 * 4;
      fv[p] = ok ? *reinterpret_cast<const float4*>(g.F + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      dfv[p] = (ok && !g.fresh) ? *reinterpret_cast<const float4*>(g.dF + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int r = p * RPP + rl;
      float t = 0.f;
      if (cok && r < R) {
        const long e = ((long)b * R + r) * D + c4 * 4;
        t = dc4.x * fv[p].x + dc4.y * fv[p].y + dc4.z * fv[p].z + dc4.w * fv[p].w;
        const float al = als[r];
        *reinterpret_cast<float4*>(g.dF + e) = make_float4(dfv[p].x + al * dc4.x, dfv[p].y + al * dc4.y,
                                                           dfv[p].z + al * dc4.z, dfv[p].w + al * dc4.w);
      }
      t = adj_sum<G4>(t);
      if (c4 == 0 && r < R) das[r] = t + g.alpha_mse * (als[r] - 1.f);
    }
    __syncthreads();
  }
  float dot = 0.f;
  for (int r = tid; r < R; r += WT) dot += als[r] * das[r];
  dot = block_sum_w(dot, red);
  float dbv = 0.f;
  for (int r = tid; r < R; r += WT) { const float de = als[r] * (das[r] - dot); als[r] = de; dbv += de; }
  dbv = block_sum_w(dbv, red);
  // ---- through e = s_d . v, dropout, tanh
  {
    const bool cok = c4 * 4 < A;
    const float4 q4 = cok ? *reinterpret_cast<const float4*>(&qs[c4 * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v4 = cok ? *reinterpret_cast<const float4*>(g.v + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float scale_a = g.rate_attn > 0.f ? 1.f / (1.f - g.rate_attn) : 1.f;
    float4 pv[MAXP], dpv[MAXP];
    uint32_t mk[MAXP];
    const bool stored = g.keep4 != nullptr && g.rate_attn > 0.f;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int r = p * RPP + rl;
      const bool ok = cok && r < R;
      const long e = ((long)b * R + r) * A + c4 * 4;
      pv[p] = ok ? *reinterpret_cast<const float4*>(g.P + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      dpv[p] = (ok && !g.fresh) ? *reinterpret_cast<const float4*>(g.dP + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      mk[p] = (stored && ok) ? g.keep4[e >> 2] : 0u;
    }
    float4 dv = make_float4(0.f, 0.f, 0.f, 0.f), dq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int r = p * RPP + rl;
      if (!(cok && r < R)) continue;
      const long e = ((long)b * R + r) * A + c4 * 4;
      const float s0 = tnt_tanh(pv[p].x + q4.x), s1 = tnt_tanh(pv[p].y + q4.y), s2 = tnt_tanh(pv[p].z + q4.z), s3 = tnt_tanh(pv[p].w + q4.w);
      bool k[4] = {true, true, true, true};
      if (stored) { k[0] = mk[p] & 1u; k[1] = mk[p] & 2u; k[2] = mk[p] & 4u; k[3] = mk[p] & 8u; }
      else if (g.rate_attn > 0.f) tnt_keep4((uint64_t)e, g.rate_attn, g.seed, g.site_attn, step, k);
      const float k0 = k[0] ? scale_a : 0.f, k1 = k[1] ? scale_a : 0.f, k2 = k[2] ? scale_a : 0.f, k3 = k[3] ? scale_a : 0.f;
      const float de = als[r];
      dv.x += s0 * k0 * de; dv.y += s1 * k1 * de; dv.z += s2 * k2 * de; dv.w += s3 * k3 * de;
      const float d0 = de * v4.x * k0 * (1.f - s0 * s0), d1 = de * v4.y * k1 * (1.f - s1 * s1);
      const float d2 = de * v4.z * k2 * (1.f - s2 * s2), d3 = de * v4.w * k3 * (1.f - s3 * s3);
      *reinterpret_cast<float4*>(g.dP + e) = make_float4(dpv[p].x + d0, dpv[p].y + d1, dpv[p].z + d2, dpv[p].w + d3);
      dq.x += d0; dq.y += d1; dq.z += d2; dq.w += d3;
    }
    dv.x = stride_sum<G4>(dv.x); dv.y = stride_sum<G4>(dv.y); dv.z = stride_sum<G4>(dv.z); dv.w = stride_sum<G4>(dv.w);
    dq.x = stride_sum<G4>(dq.x); dq.y = stride_sum<G4>(dq.y); dq.z = stride_sum<G4>(dq.z); dq.w = stride_sum<G4>(dq.w);
    if (lane < G4) {
      *reinterpret_cast<float4*>(&wred[w][lane * 4]) = dv;
      *reinterpret_cast<float4*>(&wred2[w][lane * 4]) = dq;
    }
    __syncthreads();
    if (tid < A) {
      float tv = 0.f, tq = 0.f;
#pragma unroll
      for (int k = 0; k < WW; ++k) { tv += wred[k][tid]; tq += wred2[k][tid]; }
      g.dvb[(long)b * (A + 1) + tid] = (g.fresh ? 0.f : g.dvb[(long)b * (A + 1) + tid]) + tv;
      const float qp = g.qpre_in[(long)b * A + tid];
      const float dqp = qp > 0.f ? tq : tq * g.slope;
      dq_s[tid] = dqp;
      g.dqpre[(long)b * A + tid] = dqp;
    }
    if (tid == 0) g.dvb[(long)b * (A + 1) + A] = (g.fresh ? 0.f : g.dvb[(long)b * (A + 1) + A]) + dbv;
    __syncthreads();
  }
  // ---- dh[k] = sum_a dqpre[a] W2[k][a]
  for (int k = tid; k < U; k += WT) {
    float t = 0.f;
    const float4* wr = reinterpret_cast<const float4*>(g.W2 + (long)k * A);
#pragma unroll 4
    for (int j = 0; j < A / 4; ++j) {
      const float4 wv = wr[j];
      t += dq_s[4 * j] * wv.x + dq_s[4 * j + 1] * wv.y + dq_s[4 * j + 2] * wv.z + dq_s[4 * j + 3] * wv.w;
    }
    g.dh[(long)b * U + k] = t;
  }
}

inline bool wide_ok(int R, int D, int A) {
  if ((A & 3) || (D & 3)) return false;
  const int g4 = (A > D ? A : D) / 4 <= 8 ? 8 : 16;
  return R <= MAXP * (WT / g4);
}

int32_t check_dims(int B, int R, int D, int A, int U) {
  if (B <= 0) return TNT_BADARG(1);
  if (R <= 0 || R > MAXR) return TNT_BADARG(2);
  if (D <= 0 || D > 64) return TNT_BADARG(3);
  if (A <= 0 || A > 64) return TNT_BADARG(4);
  if (U <= 0 || U > MAXU) return TNT_BADARG(5);
  return 0;
}

}  // namespace

extern "C" int32_t tnt_attention_step_fwd_f32(const float* h, const float* F, const float* P, const float* W2,
                                              const float* b2, const float* v, const float* bv, float* qpre,
                                              float* alpha, float* ctx, float* ctx_d, float* s_out, int32_t B, int32_t R,
                                              int32_t D, int32_t A, int32_t U, float slope, float rate_attn,
                                              float rate_in, int32_t in_lwidth, uint64_t seed, uint32_t site_attn,
                                              uint32_t site_in, uint32_t step, const uint32_t* step_dev,
                                              const uint8_t* keep4, void* stream) {
  if (int32_t rc = check_dims(B, R, D, A, U)) return rc;
  AttArgs g{};
  g.keep4 = keep4;
  g.h = h; g.F = F; g.P = P; g.W2 = W2; g.b2 = b2; g.v = v; g.bv = bv; g.qpre = qpre; g.alpha = alpha; g.ctx = ctx;
  g.ctx_d = ctx_d; g.s_out = s_out; g.B = B; g.R = R; g.D = D; g.A = A; g.U = U; g.in_lwidth = in_lwidth;
  g.slope = slope; g.rate_attn = rate_attn; g.rate_in = rate_in; g.seed = seed; g.site_attn = site_attn;
  g.site_in = site_in; g.step = step; g.step_dev = step_dev;
  if (wide_ok(R, D, A) && tnt_aligned16(P) && tnt_aligned16(F) && tnt_aligned16(W2) && tnt_aligned16(v) &&
      (s_out == nullptr || tnt_aligned16(s_out))) {
    if (A <= 32 && D <= 32)
      hipLaunchKernelGGL(attention_fwd_wide_kernel<8>, dim3(B), dim3(WT), 0, tnt_stream(stream), g);
    else
      hipLaunchKernelGGL(attention_fwd_wide_kernel<16>, dim3(B), dim3(WT), 0, tnt_stream(stream), g);
  } else if (A <= 32 && D <= 32)
    hipLaunchKernelGGL(attention_fwd_kernel<32>, dim3(B), dim3(256), 0, tnt_stream(stream), g);
  else
    hipLaunchKernelGGL(attention_fwd_kernel<64>, dim3(B), dim3(256), 0, tnt_stream(stream), g);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_attention_step_bwd_f32(const float* dctx_d, const float* F, const float* P, const float* W2,
                                              const float* v, const float* qpre, const float* alpha, float* dP, float* dF,
                                              float* dvb, float* dqpre, float* dh, int32_t B, int32_t R, int32_t D,
                                              int32_t A, int32_t U, float slope, float rate_attn, float rate_in,
                                              int32_t in_lwidth, uint64_t seed, uint32_t site_attn, uint32_t site_in,
                                              uint32_t step, const uint32_t* step_dev, const float* dz, const float* Wc,
                                              const float* dctx_part, int32_t nparts, const uint8_t* keep4,
                                              float alpha_mse_coef, int32_t fresh, void* stream) {
  if (int32_t rc = check_dims(B, R, D, A, U)) return rc;
  if (dz != nullptr && (U % 16 != 0 || Wc == nullptr)) return TNT_BADARG(27);
  if (dz == nullptr && dctx_d == nullptr && dctx_part == nullptr) return TNT_BADARG(1);
  if (dctx_part != nullptr && (nparts <= 0 || nparts * D > 1024)) return TNT_BADARG(30);
  AttArgs g{};
  g.dz = dz; g.Wc = Wc; g.dctx_part = dctx_part; g.nparts = nparts; g.keep4 = keep4; g.alpha_mse = alpha_mse_coef; g.fresh = fresh != 0;
  g.dctx_d = dctx_d; g.F = F; g.P = P; g.W2 = W2; g.v = v; g.qpre_in = qpre; g.alpha_in = alpha; g.dP = dP; g.dF = dF;
  g.dvb = dvb; g.dqpre = dqpre; g.dh = dh; g.B = B; g.R = R; g.D = D; g.A = A; g.U = U; g.in_lwidth = in_lwidth;
  g.slope = slope; g.rate_attn = rate_attn; g.rate_in = rate_in; g.seed = seed; g.site_attn = site_attn;
  g.site_in = site_in; g.step = step; g.step_dev = step_dev;
  if (wide_ok(R, D, A) && tnt_aligned16(P) && tnt_aligned16(F) && tnt_aligned16(W2) && tnt_aligned16(v) &&
      tnt_aligned16(dP) && tnt_aligned16(dF) && (dz == nullptr || (tnt_aligned16(dz) && tnt_aligned16(Wc)))) {
    if (A <= 32 && D <= 32)
      hipLaunchKernelGGL(attention_bwd_wide_kernel<8>, dim3(B), dim3(WT), 0, tnt_stream(stream), g);
    else
      hipLaunchKernelGGL(attention_bwd_wide_kernel<16>, dim3(B), dim3(WT), 0, tnt_stream(stream), g);
  } else if (A <= 32 && D <= 32)
    hipLaunchKernelGGL(attention_bwd_kernel<32>, dim3(B), dim3(256), 0, tnt_stream(stream), g);
  else
    hipLaunchKernelGGL(attention_bwd_kernel<64>, dim3(B), dim3(256), 0, tnt_stream(stream), g);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_attention_metric_parts(int32_t T, int32_t R) { return T * ((R + 63) / 64); }

extern "C" int32_t tnt_attention_metric_f32(const float* alpha, float* out, float* work, int32_t T, int32_t B,
                                            int32_t R, int64_t tstride, void* stream) {
  if (T <= 0 || B <= 0 || R <= 0) return TNT_BADARG(4);
  if (tstride <= 0) tstride = (int64_t)B * R;
  const int nc = (R + 63) / 64;
  hipLaunchKernelGGL(attention_metric_kernel, dim3(T, nc), dim3(256), 0, tnt_stream(stream), alpha, work, T, B, R,
                     (long)tstride);
  TNT_LAUNCH_CHECK();
  if (out == nullptr) return 0;           // the caller totals work[0 .. T * nc) * 1 / (T R) (tnt_step_finalize_f32's x2 job)
  hipLaunchKernelGGL(attention_metric_final_kernel, dim3(1), dim3(64), 0, tnt_stream(stream), work, out, T * nc,
                     1.f / ((float)T * (float)R));
  TNT_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Everything behind the T-step chain that consumes the accumulated score gradient dP [B*R][A] (the hoisted
// P = LeakyReLU(F W1 + b1) of attention.py:32; lc_NIC.py:386-387 for its gradients):
//   g = dP * LeakyReLU'(Ppre)   (never stored: nothing downstream reads it)
//   db1 = column sums of g,  dW1 = F^T g,  dF += g W1^T
// in TWO launches instead of act_bwd + a 32 x 32 x 23040 GEMM (21 us on the vendor library for 47 MFLOP) + a two-launch
// column sum + a 23040 x 32 x 32 GEMM (12 us).  Launch 1 has two kinds of workgroups: [0, nA) each reduce a chunk of
// AF_CHUNK rows to a partial of dW1 / db1 (tiles of g / F in LDS, outputs strided over the threads); [nA, nA + nB) update
// one row of dF per thread (g in registers, W1 broadcast from LDS).  Launch 2 sums the nA partials in chunk order
// (deterministic).
namespace {
constexpr int AF_TILE = 120, AF_CHUNK = 120, AF_BROWS = 64, AF_W = 32;      // D = A = AF_W (the reference's sizes)

__global__ __launch_bounds__(256) void attention_front_bwd_kernel(const float* Ppre, const float* dP, const float* F,
                                                                  const float* W1, float* dF, float* part, int rows,
                                                                  float slope, int nA, float drop_rate, uint64_t drop_seed,
                                                                  uint32_t drop_site, const uint32_t* drop_step_dev) {
  constexpr int W = AF_W, LD = AF_W + 4;                 // LDS rows of 36 floats: 16-byte aligned float4 reads
  __shared__ __attribute__((aligned(16))) float gs[AF_TILE * LD];
  __shared__ __attribute__((aligned(16))) float Fs[AF_TILE * LD];
  const int tid = threadIdx.x;
  // coalesced tile loader: float4 per thread, 8 float4 per row; g = dP * LeakyReLU'(Ppre) on the fly
  auto load_g = [&](int row0, int nr, int tile_rows) {
    for (int e = tid; e < tile_rows * (W / 4); e += 256) {
      const int r = e >> 3, c = (e & 7) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < nr) {
        const long o = (long)(row0 + r) * W + c;
        const float4 p = *reinterpret_cast<const float4*>(Ppre + o), d = *reinterpret_cast<const float4*>(dP + o);
        v = make_float4(tnt_act_grad(p.x, d.x, 1, slope), tnt_act_grad(p.y, d.y, 1, slope), tnt_act_grad(p.z, d.z, 1, slope),
                        tnt_act_grad(p.w, d.w, 1, slope));
      }
      *reinterpret_cast<float4*>(gs + r * LD + c) = v;
    }
  };
  if ((int)blockIdx.x >= nA) {
    // ---- dF[rows of this workgroup] += g W1^T: thread = (row, 8 consecutive d)
    float* Ws = Fs;                                        // [W][LD]: W1[d][a]
    const int row0 = (blockIdx.x - nA) * AF_BROWS, nr = min(AF_BROWS, rows - row0);
    load_g(row0, nr, AF_BROWS);
    for (int e = tid; e < W * (W / 4); e += 256) {
      const int d = e >> 3, c = (e & 7) * 4;
      *reinterpret_cast<float4*>(Ws + d * LD + c) = *reinterpret_cast<const float4*>(W1 + d * W + c);
    }
    __syncthreads();
    const int r = tid >> 2, d0 = (tid & 3) * 8;
    if (r >= nr) return;
    float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a4 = 0; a4 < W; a4 += 4) {
      const float4 g = *reinterpret_cast<const float4*>(gs + r * LD + a4);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(Ws + (d0 + k) * LD + a4);
        t[k] += g.x * w.x + g.y * w.y + g.z * w.z + g.w * w.w;
      }
    }
    float4* o = reinterpret_cast<float4*>(dF + (long)(row0 + r) * W + d0);
    float4 u0 = o[0], u1 = o[1];
    u0.x += t[0]; u0.y += t[1]; u0.z += t[2]; u0.w += t[3]; u1.x += t[4]; u1.y += t[5]; u1.z += t[6]; u1.w += t[7];
    if (drop_rate > 0.f) {
      // dF is complete here: the backward of the feature Dropout in front of the attention (layers.py:51) in the same pass
      const uint32_t step = drop_step_dev ? drop_step_dev[0] : 0u;
      const float sc = 1.f / (1.f - drop_rate);
      const uint64_t le = (uint64_t)(row0 + r) * W + d0;
      bool k0[4], k1[4];
      tnt_keep4(le, drop_rate, drop_seed, drop_site, step, k0);
      tnt_keep4(le + 4, drop_rate, drop_seed, drop_site, step, k1);
      u0.x = k0[0] ? u0.x * sc : 0.f; u0.y = k0[1] ? u0.y * sc : 0.f; u0.z = k0[2] ? u0.z * sc : 0.f; u0.w = k0[3] ? u0.w * sc : 0.f;
      u1.x = k1[0] ? u1.x * sc : 0.f; u1.y = k1[1] ? u1.y * sc : 0.f; u1.z = k1[2] ? u1.z * sc : 0.f; u1.w = k1[3] ? u1.w * sc : 0.f;
    }
    o[0] = u0; o[1] = u1;
    return;
  }
  // ---- partial of dW1 (thread = (d, 4 consecutive a)) and db1 (threads 0..31) over this chunk's rows
  const int c0 = blockIdx.x * AF_CHUNK, c1 = min(rows, c0 + AF_CHUNK);
  const int d = tid >> 3, a0 = (tid & 7) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float bsum = 0.f;
  for (int t0 = c0; t0 < c1; t0 += AF_TILE) {
    const int nr = min(AF_TILE, c1 - t0);
    __syncthreads();
    load_g(t0, nr, AF_TILE);
    for (int e = tid; e < AF_TILE * (W / 4); e += 256) {
      const int r = e >> 3, c = (e & 7) * 4;
      *reinterpret_cast<float4*>(Fs + r * LD + c) =
          r < nr ? *reinterpret_cast<const float4*>(F + (long)(t0 + r) * W + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < AF_TILE; ++r) {
      const float f = Fs[r * LD + d];
      const float4 g = *reinterpret_cast<const float4*>(gs + r * LD + a0);
      acc.x += f * g.x; acc.y += f * g.y; acc.z += f * g.z; acc.w += f * g.w;
    }
    if (tid < W) {
#pragma unroll 8
      for (int r = 0; r < AF_TILE; ++r) bsum += gs[r * LD + tid];
    }
  }
  float* mine = part + (long)blockIdx.x * (W * W + W);
  *reinterpret_cast<float4*>(mine + d * W + a0) = acc;
  if (tid < W) mine[W * W + tid] = bsum;
}

// 64 outputs per workgroup x AFF_S slices of the chunk list: every thread sums its slice's partials (independent loads, all in
// flight), the slices are combined through LDS in slice order -- the same summation tree on every run.
constexpr int AFF_S = 16;
__global__ __launch_bounds__(64 * AFF_S) void attention_front_finalize_kernel(const float* part, int nA, int nout, int nw,
                                                                              float* dW1, float* db1) {
  __shared__ float red[AFF_S][64];
  const int ol = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + ol;
  const int per = (nA + AFF_S - 1) / AFF_S, g0 = sl * per, g1 = min(nA, g0 + per);
  float t = 0.f;
  if (o < nout) {
#pragma unroll 4
    for (int g = g0; g < g1; ++g) t += part[(long)g * nout + o];
  }
  red[sl][ol] = t;
  __syncthreads();
  if (sl != 0 || o >= nout) return;
#pragma unroll
  for (int k = 1; k < AFF_S; ++k) t += red[k][ol];
  if (o < nw) dW1[o] = t; else db1[o - nw] = t;
}
}  // namespace

extern "C" int32_t tnt_attention_front_bwd_parts(int32_t rows, int32_t D, int32_t A) {
  return ((rows + AF_CHUNK - 1) / AF_CHUNK) * (D * A + A);
}

extern "C" int32_t tnt_attention_front_bwd_drop_f32(const float* Ppre, const float* dP, const float* F, const float* W1,
                                                    float* dF, float* dW1, float* db1, float* part, int32_t rows, int32_t D,
                                                    int32_t A, float slope, float drop_rate, uint64_t drop_seed,
                                                    uint32_t drop_site, const uint32_t* drop_step_dev, void* stream) {
  if (drop_rate < 0.f || drop_rate >= 1.f) return TNT_BADARG(13);
  if (rows <= 0 || D != AF_W || A != AF_W) return TNT_BADARG(9);          // callers keep the unfused sequence otherwise
  if (!tnt_aligned16(Ppre) || !tnt_aligned16(dP) || !tnt_aligned16(F) || !tnt_aligned16(W1) || !tnt_aligned16(dF) ||
      !tnt_aligned16(part)) return TNT_BADARG(1);
  const int nA = (rows + AF_CHUNK - 1) / AF_CHUNK, nB = (rows + AF_BROWS - 1) / AF_BROWS;
  hipStream_t s = tnt_stream(stream);
  hipLaunchKernelGGL(attention_front_bwd_kernel, dim3(nA + nB), dim3(256), 0, s, Ppre, dP, F, W1, dF, part, rows, slope, nA,
                     drop_rate, drop_seed, drop_site, drop_step_dev);
  TNT_LAUNCH_CHECK();
  hipLaunchKernelGGL(attention_front_finalize_kernel, dim3((D * A + A + 63) / 64), dim3(64 * AFF_S), 0, s, part, nA, D * A + A, D * A,
                     dW1, db1);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_attention_front_bwd_f32(const float* Ppre, const float* dP, const float* F, const float* W1,
                                               float* dF, float* dW1, float* db1, float* part, int32_t rows, int32_t D,
                                               int32_t A, float slope, void* stream) {
  return tnt_attention_front_bwd_drop_f32(Ppre, dP, F, W1, dF, dW1, db1, part, rows, D, A, slope, 0.f, 0, 0, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------------
// The forward chain of the attention captioner (lc_NIC.py:244-256: for every step, attention over the regions with the
// previous h as the query, then ONE LSTM step on [context, word]) as ONE persistent launch instead of 2 T dependent
// launches.  Placement, tickets, epochs and the data-polling hand-off are those of lstm_seq_fwd_kernel<POLL> (lstm.hip,
// tnt_seq_sync.h): a 16-row block of the batch lives on one XCD, its 32 workgroups each own 16 LSTM units (recurrent
// weights resident in VGPRs), and the first 16 of them also own one SAMPLE each for the attention phase.  Per step:
//   attention (16 workgroups): poll h[i][b] (sentinel), att_fwd_body, publish ctx_d[i][b];
//   LSTM (32 workgroups):      poll h[i] (this wave's K chunk) and ctx_d[i] (waves < D/4: one k-step of the context part
//                              each), MFMAs, combine, gates, publish h[i+1].
// Sentinel resets, by the owning thread: h[i+2] and ctx_d[i+1] at the top of step i (drained before the same thread
// publishes h[i+1] / ctx_d[i]); h[1] and ctx_d[0] at kernel start behind the launch's only flag barrier.
#include "tnt_seq_sync.h"

namespace {
struct LcSeqArgs {
  AttArgs att;             // per-step pointers (qpre, alpha, ctx, ctx_d, keep4) point at step 0; sites at step 0
  long keep_stride;        // bytes between the stored keep masks of consecutive steps
  const float* xz;         // [T][B][U][4] text half of x W (no bias)
  const float* Wc;         // [D][U][4] context rows of lstm/kernel
  const float* Ur;         // [U][U][4]
  const float* zbias;      // [U][4]
  float* hs; float* cs;    // [T+1][B][U]
  float* gates;            // [T][B][U][4]
  float* hd;               // nullable [T][B][U]: Dropout(hs[1:]) with one site per step (tnt_dropout_f32, rows_per_site = B)
  float rate_out; uint32_t site_out0;
  float* qp;               // [3][B][64][16]: the 16 LSTM workgroups' partials of q = h W2 per sample (ring of three)
  int T;
  unsigned* sync; float* guard_out;
};

constexpr unsigned LC_SENTINEL = 0x7FC5EED5u;          // = TNT_SEQ_SENTINEL of lstm.hip
constexpr int LC_SEQ_LDS_BYTES = 16 * 4 * 16 * 17 * 4 + 16;

// Stores / loads through a buffer resource: the base lives in 4 SGPRs, a lane's address is ONE 32-bit VGPR (its byte offset inside
// a slab) plus a wave-uniform SGPR offset (the slab of the step).  The chain kernels' LSTM roles keep 64 resident weight
// registers; with 64-bit pointers per output array the compiler spilled address registers to scratch and reloaded them on the
// step's critical path (a scratch round trip in front of every hand-off store: measured 2 us).
typedef unsigned lc_u4 __attribute__((ext_vector_type(4)));
// a wave-uniform float computed by the vector ALU (a reciprocal of a kernel argument): parked in an SGPR instead of a VGPR
__device__ __forceinline__ float lc_uniform(float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); }
__device__ __forceinline__ void lc_st1(__amdgpu_buffer_rsrc_t rsrc, float v, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, (int)voff, (int)soff, 0);
}
// HARDWARE HAZARD (gfx950, found the hard way: tools/probe/lc_bwd_dbg.py): a 16-byte buffer store whose soffset is an SGPR must
// not be followed directly by a VALU write of its first data register -- the store then carries the NEW value in that dword.
// LLVM's hazard recognizer inserts the wait state only for the immediate-soffset form, so the helper keeps the data registers
// alive across one s_nop: nothing can be scheduled into the slot behind the store.
__device__ __forceinline__ void lc_st4(__amdgpu_buffer_rsrc_t rsrc, float4 v, unsigned voff, unsigned soff) {
  const lc_u4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, (int)voff, (int)soff, 0);
  asm volatile("s_nop 1" :: "v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w));
}
__device__ __forceinline__ float lc_ld1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float4 lc_ld4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const tnt_f4 v = __builtin_bit_cast(tnt_f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 lc_ld4_l2(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {      // sc1: served by L2
  const tnt_f4 v = __builtin_bit_cast(tnt_f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, /*sc1*/ 16));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float lc_ld1_l2(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)byte_off, 0, /*sc1*/ 16));
}

#ifdef TNT_LC_TRACE
__device__ unsigned long long lc_trace[64];
extern "C" int32_t tnt_debug_lc_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(lc_trace), sizeof(lc_trace)) == hipSuccess ? 0 : -1;
}
// timestamps are parked in LDS and written out when the workgroup leaves: a global store at a trace point would sit in
// front of the very s_waitcnt vmcnt(0) it is meant to time
#define LCT(k) do { if (tid == 0 && rb == 0 && i == LCT_STEP) lct_l[(k)] = wall_clock64(); } while (0)
#define LCT_T(k, t) do { if (tid == (t) && rb == 0 && i == LCT_STEP) lct_l[(k)] = wall_clock64(); } while (0)
#define LCT_DECL __shared__ unsigned long long lct_l[64];
#define LCT_STEP 5
#define LCT_DUMP(lo, hi) do { if (tid == 0 && rb == 0) for (int q_ = (lo); q_ < (hi); ++q_) lc_trace[q_] = lct_l[q_]; } while (0)
// per-step timeline of one workgroup per role: [0] entry, [1] behind the launch's flag barrier, [2 + k] the k-th step's mark,
// [38] loop done, [39] outputs stored (roles: 0 forward attention, 1 forward LSTM, 2 backward attention, 3 backward LSTM)
__device__ unsigned long long lc_steps[4][40];
__device__ unsigned long long lc_pub[10][32];      // backward, step LCT_STEP, row block 0: [0] dh_att in, [1] parts published, per LSTM workgroup; [2] dq out per attention workgroup; [3] step top, [4] pushed, [5] gathered per LSTM workgroup
extern "C" int32_t tnt_debug_lc_pub(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(lc_pub), sizeof(lc_pub)) == hipSuccess ? 0 : -1;
}
extern "C" int32_t tnt_debug_lc_steps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(lc_steps), sizeof(lc_steps)) == hipSuccess ? 0 : -1;
}
#define LCS_DECL __shared__ unsigned long long lcs_l[40];
#define LCS(k) do { if (tid == 0 && rb == 0) lcs_l[(k)] = wall_clock64(); } while (0)
#define LCS_DUMP(role) do { if (tid == 0 && rb == 0) for (int q_ = 0; q_ < 40; ++q_) lc_steps[role][q_] = lcs_l[q_]; } while (0)
#else
#define LCS_DECL
#define LCS(k) do {} while (0)
#define LCS_DUMP(role) do {} while (0)
#define LCT(k) do {} while (0)
#define LCT_T(k, t) do {} while (0)
#define LCT_DECL
#define LCT_DUMP(lo, hi) do {} while (0)
#endif
// Role-specialised workgroups.  A group is the 32 workgroups of one XCD and owns one block of 16 samples: slots 0..15 run
// the ATTENTION of one sample each, slots 16..31 the LSTM step of 32 units each (two 16-unit blocks) for the 16 samples.
// Everything step-invariant stays on chip for all T steps: an attention workgroup keeps its sample's P and F rows in
// registers and W2 in LDS (A <= 32); an LSTM workgroup keeps its recurrent-kernel fragments in registers and its context
// rows of the LSTM kernel in LDS.  Per step the two roles overlap: while the attention workgroups work on step i, the LSTM
// workgroups already multiply h[i] with the recurrent kernel (the bulk of their step, 512 of the 544 input columns); only
// the 32-column context term and the gate math wait for the attention's ctx_d[i].  Hand-offs are data-polling on the
// sentinel (h and ctx_d are their own flags, reset-before-publish by the owning thread, as in lstm_seq_fwd_kernel).
// RB: samples per row block (= per XCD): 16, or 8 to spread B <= 64 over all 8 XCDs (then 8 attention workgroups per XCD;
// the LSTM workgroups' A fragments are half empty, but a group hands over, polls and reduces half as many rows).
template <int G4, int NP, int RB>
__global__ __launch_bounds__(1024) void lc_seq_fwd_kernel(LcSeqArgs a) {
  constexpr int NWF = 16, SS = 8, CK = 32, RPP = WT / G4;
  extern __shared__ __attribute__((aligned(16))) float lc_lds[];
  float (*red)[4][16][17] = reinterpret_cast<float (*)[4][16][17]>(lc_lds);          // LSTM role: [NWF][4][16][17]
  unsigned* s_slot = reinterpret_cast<unsigned*>(lc_lds + NWF * 4 * 16 * 17);
  __shared__ __attribute__((aligned(16))) float wred_l[WW][64];
  __shared__ __attribute__((aligned(16))) float qs_l[64], qe_l[64], zred_l[WW];      // attention role: q, e^{2q}, per-wave sums
  __shared__ float red_l[WW];
  __shared__ int qbig_l;
  __shared__ float w2s_l[32 * 64];              // LSTM role: W2 rows of this workgroup's 32 units, [32][64]
  __shared__ __attribute__((aligned(16))) float ctx_l[16 * 64];                      // LSTM role: the step's context, [row][D]
  __shared__ float hq_l[16 * 36];               // LSTM role: the h just computed, [row][unit], row stride 36
  __shared__ __attribute__((aligned(16))) float zc_l[2 * 16 * 132];                  // LSTM role: the context term's two k-halves
  __shared__ float wcb_l[16 * 8 * 64];          // LSTM role: B operands of the context-term MFMAs, [wave][quad slot][lane]
  // LSTM role, per epilogue thread: the gate bias, and (text projection + bias) of the current step -- parked in LDS: resident
  // in registers they are 8 of the VGPRs this role spills
  __shared__ __attribute__((aligned(16))) float4 zb_l[512], zx_l[512];
  LCT_DECL
  LCS_DECL
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, lr = lane & 15;
  const int U = a.att.U, B = a.att.B, D = a.att.D, R = a.att.R, A = a.att.A, T = a.T;
  const unsigned xcc = tnt_xcc_id();
  const int nrb = (B + RB - 1) / RB;
  if ((int)xcc >= nrb) return;
  unsigned* bar = a.sync + xcc * 64;
  unsigned* err = a.sync + TNT_SEQ_ERR;
  const TntSeqSlot slot = tnt_seq_enter(a.sync, xcc, s_slot);
  if (slot.ub < 0) {
    if (tid == 0 && a.guard_out) a.guard_out[0] = 2.f;
    return;
  }
  const int ub = slot.ub, rb = __builtin_amdgcn_readfirstlane((int)xcc);
  const long BU = (long)B * U;
  const __amdgpu_buffer_rsrc_t hs_rsrc = tnt_rsrc(a.hs, (unsigned)((long)(T + 1) * BU * 4));
  const __amdgpu_buffer_rsrc_t cx_rsrc = tnt_rsrc(a.att.ctx_d, (unsigned)((long)T * B * D * 4));
  const float sentinel = __uint_as_float(LC_SENTINEL);
  const AttArgs& g = a.att;
  LCS(0);

  if (ub < 16) {
    // =========================================================== attention role: sample ab
    // Per step: the 16 LSTM workgroups of the group each leave the 32-unit partial of q = h W2 for this sample (qp, its own
    // flag); this workgroup sums them, and then ONE pass over its rows does scores, exp and the weighted sum of F together:
    //   e_r = dropout(tanh(P_r + q)) . v + bv;   x_r = exp(e_r - m);   c = sum_r x_r F_r;   z = sum_r x_r;   ctx = c / z
    // with m an upper bound of every e_r that is known before the step (bv + sum_a |v_a| / (1 - rate): |tanh| <= 1), so the
    // softmax needs no maximum, no second pass and no reduction of its own -- z rides in the reduction of c.  Two workgroup
    // barriers per step (behind q, behind the per-wave sums) instead of nine.  tanh(P + q) = 1 - 2 / (e^{2P} e^{2q} + 1): e^{2P}
    // is step-invariant and lives in the registers P used to, e^{2q} costs A exps per step, so an element costs ONE
    // transcendental (the reciprocal) instead of two.  Where that could lose accuracy (|P| or |q| > 40: the product of two
    // clamped exponentials) the step uses tanh(P + q) itself, and where the bound m is too far above the scores for exp
    // (2 sum |v| / (1 - rate) > 60) the maximum is reduced as before: both uniform branches, both exercised by the tests.
    const int ab = rb * RB + ub;
    const bool live = ub < RB && ab < B;
    const int c4 = tid % G4, rl = tid / G4;
    const bool cokA = c4 * 4 < A, cokD = c4 * 4 < D;
    float4 pv[NP], fv[NP];                       // e^{2 P} (P itself when `direct`), F
    float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float b2v = 0.f, bvv = 0.f, mb = 0.f;
    bool direct = false, bounded = true;
    const float scale_a = lc_uniform(1.f / (1.f - g.rate_attn));
    const float scale_in = lc_uniform(g.rate_in > 0.f ? 1.f / (1.f - g.rate_in) : 1.f);
    if (live) {
      bool big = false;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int r = p * RPP + rl;
        pv[p] = (cokA && r < R) ? *reinterpret_cast<const float4*>(g.P + ((long)ab * R + r) * A + c4 * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        fv[p] = (cokD && r < R) ? *reinterpret_cast<const float4*>(g.F + ((long)ab * R + r) * D + c4 * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        big = big || !(fabsf(pv[p].x) <= 40.f && fabsf(pv[p].y) <= 40.f && fabsf(pv[p].z) <= 40.f && fabsf(pv[p].w) <= 40.f);
      }
      if (cokA) v4 = *reinterpret_cast<const float4*>(g.v + c4 * 4);
      if (tid < 4 * A) b2v = g.b2[tid >> 2];
      bvv = g.bv[0];
      if (tid < D) g.ctx_d[(long)ab * D + tid] = sentinel;
      if (tid == 0) qbig_l = 0;
      direct = __syncthreads_or(big ? 1 : 0) != 0;
      if (!direct) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          pv[p].x = __expf(2.f * pv[p].x); pv[p].y = __expf(2.f * pv[p].y);
          pv[p].z = __expf(2.f * pv[p].z); pv[p].w = __expf(2.f * pv[p].w);
        }
      }
      const float vsum = adj_sum<G4>(fabsf(v4.x) + fabsf(v4.y) + fabsf(v4.z) + fabsf(v4.w)) * (g.rate_attn > 0.f ? scale_a : 1.f);
      bounded = 2.f * vsum <= 60.f;
      mb = bvv + vsum;
      // the dropout scale rides in v
      if (g.rate_attn > 0.f) { v4.x *= scale_a; v4.y *= scale_a; v4.z *= scale_a; v4.w *= scale_a; }
    }
    tnt_seq_group_barrier(bar, ub, tnt_seq_target(slot.epoch, 1), err);
    LCS(1);
    const uint32_t step = g.step + (g.step_dev ? g.step_dev[0] : 0u);
    const bool stored = g.keep4 != nullptr && g.rate_attn > 0.f;
    const __amdgpu_buffer_rsrc_t qp_rsrc = tnt_rsrc(a.qp, (unsigned)(3u * (unsigned)B * 1024u * 4u));
    if (live) for (int i = 0; i < T; ++i) {
      float* const o_qpre = g.qpre + (long)i * B * A;
      float* const o_alpha = g.alpha + (long)i * B * R;
      float* const o_ctx = g.ctx + (long)i * B * D;
      float* const o_ctxd = g.ctx_d + (long)i * B * D;
      const uint32_t site_attn = g.site_attn + (uint32_t)i, site_in = g.site_in + (uint32_t)i;
      if (tid < D && i + 1 < T) g.ctx_d[((long)(i + 1) * B + ab) * D + tid] = sentinel;
      LCT(32);
      // the context's input-dropout decision of this step (a Philox call) does not depend on the chain: taken here
      bool kin = true;
      if (tid < D && g.rate_in > 0.f) kin = tnt_keep((uint64_t)ab * (uint64_t)g.in_lwidth + tid, g.rate_in, g.seed, site_in, step);
      // this step's keep bits: in flight while the q partials are polled
      uint32_t mk[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int r = p * RPP + rl;
        mk[p] = (stored && cokA && r < R) ? g.keep4[(long)i * a.keep_stride + ((((long)ab * R + r) * A + c4 * 4) >> 2)] : 0xFu;
      }
      // ---- q[a] = b2[a] + the 16 partials of this sample: thread 4 a + c polls partials 4 c .. 4 c + 3 of column a
      {
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool mine = tid < 4 * A;
        const unsigned off = (unsigned)((((i % 3) * B + ab) * 1024 + tid * 4) * 4);
        unsigned spins = 0;
        if (mine) for (;;) {
          qv = tnt_ld4_l2(qp_rsrc, off);
          const bool ok = __float_as_uint(qv.x) != LC_SENTINEL && __float_as_uint(qv.y) != LC_SENTINEL &&
                          __float_as_uint(qv.z) != LC_SENTINEL && __float_as_uint(qv.w) != LC_SENTINEL;
          if (ok) break;                                     // per lane: a lane leaves the loop when ITS chunk is in
          if (++spins > TNT_SEQ_SPIN_LIMIT) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
        LCT(33);
        if (i < 36) LCS(2 + i);
        if (tid < 4 * A) {                                   // whole waves up to the last one (A % 4 == 0: groups of 4 lanes are whole)
          const float t = adj_sum<4>((qv.x + qv.y) + (qv.z + qv.w)) + b2v;
          if ((tid & 3) == 0) {
            const float q = t > 0.f ? t : t * g.slope;
            o_qpre[(long)ab * A + (tid >> 2)] = t;
            qs_l[tid >> 2] = q;
            qe_l[tid >> 2] = __expf(2.f * q);
            if (!(fabsf(q) <= 40.f)) qbig_l = i + 1;
          }
        }
      }
      __syncthreads();
      LCT(34);
      // ---- scores, exp and the weighted sums in one pass
      float ex[NP];
      float zl = 0.f;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      {
        const bool fast = !direct && qbig_l != i + 1;
        const float4 q4 = cokA ? *reinterpret_cast<const float4*>(fast ? &qe_l[c4 * 4] : &qs_l[c4 * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int r = p * RPP + rl;
          float t = 0.f;
          if (cokA && r < R) {
            const long e = ((long)ab * R + r) * A + c4 * 4;
            float s0, s1, s2, s3;
            if (fast) {
              s0 = 1.f - 2.f * __builtin_amdgcn_rcpf(pv[p].x * q4.x + 1.f); s1 = 1.f - 2.f * __builtin_amdgcn_rcpf(pv[p].y * q4.y + 1.f);
              s2 = 1.f - 2.f * __builtin_amdgcn_rcpf(pv[p].z * q4.z + 1.f); s3 = 1.f - 2.f * __builtin_amdgcn_rcpf(pv[p].w * q4.w + 1.f);
            } else {
              const float4 p4 = direct ? pv[p] : *reinterpret_cast<const float4*>(g.P + e);
              s0 = tnt_tanh(p4.x + q4.x); s1 = tnt_tanh(p4.y + q4.y); s2 = tnt_tanh(p4.z + q4.z); s3 = tnt_tanh(p4.w + q4.w);
            }
            uint32_t kb = mk[p];
            if (g.rate_attn > 0.f && !stored) {
              bool k[4];
              tnt_keep4((uint64_t)e, g.rate_attn, g.seed, site_attn, step, k);
              kb = (k[0] ? 1u : 0u) | (k[1] ? 2u : 0u) | (k[2] ? 4u : 0u) | (k[3] ? 8u : 0u);
            }
            t = ((kb & 1u) ? s0 * v4.x : 0.f) + ((kb & 2u) ? s1 * v4.y : 0.f) + ((kb & 4u) ? s2 * v4.z : 0.f) + ((kb & 8u) ? s3 * v4.w : 0.f);
          }
          ex[p] = adj_sum<G4>(t) + bvv;                      // the score of row r, in every lane of its group
        }
        float m = mb;
        if (!bounded) {                                      // uniform: the bound is too far off, reduce the maximum
          m = -INFINITY;
#pragma unroll
          for (int p = 0; p < NP; ++p) if (p * RPP + rl < R) m = fmaxf(m, ex[p]);
          m = block_max_w(m, red_l);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const float x = p * RPP + rl < R ? __expf(ex[p] - m) : 0.f;
          ex[p] = x; zl += x;
          acc.x += x * fv[p].x; acc.y += x * fv[p].y; acc.z += x * fv[p].z; acc.w += x * fv[p].w;
        }
        acc.x = stride_sum<G4>(acc.x); acc.y = stride_sum<G4>(acc.y);
        acc.z = stride_sum<G4>(acc.z); acc.w = stride_sum<G4>(acc.w);
        zl = stride_sum<G4>(zl);                             // every row of the wave once: the lanes of a group hold the same x
        if (lane < G4) *reinterpret_cast<float4*>(&wred_l[w][lane * 4]) = acc;
        if (lane == 0) zred_l[w] = zl;
      }
      __syncthreads();
      LCT(36);
      float z = 0.f;
      {
        const float4* zr = reinterpret_cast<const float4*>(zred_l);
#pragma unroll
        for (int k = 0; k < WW / 4; ++k) { const float4 t = zr[k]; z += (t.x + t.y) + (t.z + t.w); }
      }
      const float invz = 1.f / z;
      if (tid < D) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < WW; ++k) t += wred_l[k][tid];
        t *= invz;
        const float td = g.rate_in > 0.f ? (kin ? t * scale_in : 0.f) : t;
        // the reset of ctx_d[i+1] (issued at the top of the step, long drained) is in L2 before ctx_d[i] is published;
        // this thread's other stores of the step (qpre) are as old
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        o_ctxd[(long)ab * D + tid] = td;
        o_ctx[(long)ab * D + tid] = t;
      }
      LCT(37);
      // off the critical path: alpha of this step
      if (c4 == 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int r = p * RPP + rl;
          if (r < R) o_alpha[(long)ab * R + r] = ex[p] * invz;
        }
      }
      // qs_l / qe_l / wred_l / zred_l are rewritten next step: every reader of this step is in front of the next step's first
      // barrier, every writer behind it (q) or behind the second one (the sums)
    }
    if (ub == 0) LCT_DUMP(32, 40);
    LCS(38); LCS(39);
    if (ub == 0) LCS_DUMP(0);
    tnt_seq_leave(a.sync, xcc, a.guard_out);
    return;
  }

  // ============================================================= LSTM role: units [32 j, 32 j + 32) of the 16 samples
  const int j = ub - 16;
  const int arow = lr < RB ? rb * RB + lr : B;               // rows past RB: zero fragments
  // recurrent-kernel fragments of both 16-unit blocks: resident for all T steps
  float4 bv[2][SS];
  // RB == 8: the batched 4x4x1 MFMA of lstm_seq_fwd_kernel (lstm.hip: 8 rows x 32 gate columns x one k per instruction, lane
  // l = 32 rg + 4 cg + j, A broadcast from block `abid` of each half, B from one half of the wave) -- no padded rows
  const int x_rg = lane >> 5, x_cg = (lane >> 2) & 7, x_j = lane & 3;
  float bx[2][2][8][2];                    // [unit block][column half][abid = k quad][m pair]
  if (RB == 8) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int qd = 0; qd < 8; ++qd)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
            bx[q][ch][qd][pr] = a.Ur[((long)(w * CK + 4 * qd + 2 * pr + x_rg) * U + (2 * j + q) * 16) * 4 + ch * 32 + x_cg * 4 + x_j];
  } else {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ucol = (2 * j + q) * 16 + lr;
#pragma unroll
      for (int s = 0; s < SS; ++s)
        bv[q][s] = *reinterpret_cast<const float4*>(a.Ur + ((long)(w * CK + (s >> 2) * 16 + kq * 4 + (s & 3)) * U + ucol) * 4);
    }
  }
  // the context term ctx[16][D] Wc[D][128 gate columns of this workgroup] runs on the MFMAs (a scalar loop over D is a chain of
  // 2 D dependent LDS round trips: 1.3 us on the critical path at D = 32): wave w owns column tile w & 7 (4 units x 4 gates)
  // and half kh = w >> 3 of the contraction.  Every wave polls ITS A operands straight into registers -- lane (kq, lr) takes
  // the float4 ctx_d[row lr][DH kh + 16 u + 4 kq ..] (u < NU), element s & 3 of float4 s >> 2 feeds MFMA s, so the contraction
  // index of MFMA s is d = DH kh + 16 (s >> 2) + 4 kq + (s & 3) and the B operands are laid out to match -- no LDS round trip
  // and no workgroup barrier between the hand-off and the MFMAs.  The B operands are parked in LDS, not in registers (this
  // role spills).
  const int t8 = w & 7, kh = w >> 3;
  constexpr int NU = G4 == 8 ? 1 : 2;          // float4 per lane: D <= 32 -> 1, D <= 64 -> 2 (operands past D are zero)
  constexpr int NQ = 4 * NU, DH = 16 * NU;
#pragma unroll
  for (int sq = 0; sq < NQ; ++sq) {
    const int d = DH * kh + 16 * (sq >> 2) + 4 * kq + (sq & 3);
    wcb_l[(w * 8 + sq) * 64 + lane] = d < D ? a.Wc[((long)d * U + j * 32) * 4 + t8 * 16 + lr] : 0.f;
  }
  // q = h W2 leaves this role in 32-unit partials, [16 rows][32 units] x [32][A] on the MFMAs: wave 8 + t owns column tile t
  // (16 columns), lane (kq, lr) feeds A[row lr][unit 4 s + kq] from hq_l (row stride 36: conflict-free) and B[unit 4 s + kq][column
  // 16 t + lr] from w2s_l for the 8 k-steps and receives rows 4 kq .. + 3 of column 16 t + lr.  (A scalar loop -- 32 FMAs per
  // thread with its operands in LDS -- took 1.5 us here: this role has no registers left to batch the LDS reads in.)
  // (row stride 64 whatever A is, columns past A zero: every operand address is the lane's base plus a compile-time offset)
  for (int e = tid; e < 32 * 64; e += WT) w2s_l[e] = (e & 63) < A ? g.W2[(long)(j * 32 + (e >> 6)) * A + (e & 63)] : 0.f;
  const int qtile = w - 8, qcol = qtile * 16 + lr;
  const bool qwave = qtile >= 0 && qtile * 16 < A;            // wave-uniform
  // this lane's element (row 4 kq + r of the block, column qcol, partial j) of ring buffer buf: byte offset qoff + uniform part
  const __amdgpu_buffer_rsrc_t qw_rsrc = tnt_rsrc(a.qp, (unsigned)(3u * (unsigned)B * 1024u * 4u));
  const unsigned qoff = (unsigned)((((rb * RB + kq * 4) * 64 + qcol) * 16 + j) * 4);
  auto qput = [&](int buf, int r, float v) { lc_st1(qw_rsrc, v, qoff, (unsigned)((buf * B + r) * 4096)); };
  auto qpart = [&]() {
    floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float qa_[4], qb_[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qa_[s] = hq_l[lr * 36 + 16 * h + 4 * s + kq];
        qb_[s] = w2s_l[(16 * h + 4 * s + kq) * 64 + qcol];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa_[s], qb_[s], acc, 0, 0, 0);
    }
    return acc;
  };
  // this lane's rows of the partial: 4 kq + r, live below RB and inside the batch
  auto qrow_ok = [&](int r) { return qwave && qcol < A && kq * 4 + r < RB && rb * RB + kq * 4 + r < B; };
  if (tid < 16 * 32) {                         // q of step 0 comes from the initial state
    const int r = tid >> 5, u = tid & 31;
    hq_l[r * 36 + u] = (r < RB && rb * RB + r < B) ? a.hs[(long)(rb * RB + r) * U + j * 32 + u] : 0.f;
  }
  __syncthreads();
  if (qwave) {
    const floatx4 t = qpart();
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (qrow_ok(r)) {
        qput(0, r, t[r]);                      // in L2 before the launch's flag barrier: buffer 0 needs no sentinel
        if (T > 1) qput(1, r, sentinel);
      }
  }
  constexpr int ZCLD = 132;
  float* zc = zc_l;                        // [2 k-halves][16 rows][ZCLD] (not in `red`: the waves that do not reduce the bulk
                                           // partials are already writing here while the others still read them)
  // epilogue threads: (unit block q, row, unit)
  const int eq = tid >> 8, erow = (tid & 255) >> 4, ecol = tid & 15;
  const int eb = rb * RB + erow, eu = (2 * j + eq) * 16 + ecol;
  const bool eok = tid < 512 && erow < RB && eb < B;
  const long ee = (long)eb * U + eu;
  const unsigned ee4 = (unsigned)(eb * U + eu) * 4u, BU4 = (unsigned)BU * 4u;      // byte offsets inside / between [B][U] slabs
  const __amdgpu_buffer_rsrc_t cs_rsrc = tnt_rsrc(a.cs, (unsigned)((long)(T + 1) * BU * 4));
  const __amdgpu_buffer_rsrc_t gt_rsrc = tnt_rsrc(a.gates, (unsigned)((long)T * BU * 16));
  const __amdgpu_buffer_rsrc_t xz_rsrc = tnt_rsrc(a.xz, (unsigned)((long)T * BU * 16));
  const __amdgpu_buffer_rsrc_t hd_rsrc = tnt_rsrc(a.hd, a.hd ? (unsigned)((long)T * BU * 4) : 0u);
  float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float cp = 0.f;
  const uint32_t step_l = a.att.step + (a.att.step_dev ? a.att.step_dev[0] : 0u);
  const float oscale = lc_uniform(a.rate_out > 0.f ? 1.f / (1.f - a.rate_out) : 1.f);
  if (eok) {
    zb_l[tid] = a.zbias ? *reinterpret_cast<const float4*>(a.zbias + (long)eu * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    cp = a.cs[ee];
    x4 = lc_ld4(xz_rsrc, ee4 * 4u, 0u);
    lc_st1(hs_rsrc, sentinel, ee4, BU4);
  }
  tnt_seq_group_barrier(bar, ub, tnt_seq_target(slot.epoch, 1), err);
  LCS(1);
  // Register discipline of this loop.  64 VGPRs hold the resident weights, so everything else a step needs has 64 to live in,
  // and the compiler's habit of computing every lane-constant index and address ONCE, in front of the loop, and keeping it
  // for all T steps pushed ~30 of them into scratch -- reloaded on the critical path, a scratch round trip in front of each
  // hand-off (measured: 2 us in front of the q partial alone).  Each phase therefore derives its indices from an opaque copy of
  // the thread index (LC_TID: the compiler cannot hoist what depends on it), a handful of integer instructions per phase;
  // what is wave-uniform (w, the step, slab offsets) is kept scalar.
#define LC_TID(v) int v = threadIdx.x; asm volatile("" : "+v"(v))
  for (int i0 = 0; i0 < T; ++i0) {
    const int i = __builtin_amdgcn_readfirstlane(i0);
    LCT(48);
    bool kout = true;
    {
      LC_TID(t0);
      const int erow0 = (t0 & 255) >> 4, eb0 = rb * RB + erow0;
      const bool eok0 = t0 < 512 && erow0 < RB && eb0 < B;
      const unsigned ee0 = (unsigned)(eb0 * U + (2 * j + (t0 >> 8)) * 16 + (t0 & 15));
      if (eok0) {       // x4 = xz[i] (loaded behind the previous step's publish) meets the bias here and leaves the registers
        const float4 b4 = zb_l[t0];
        zx_l[t0] = make_float4(x4.x + b4.x, x4.y + b4.y, x4.z + b4.z, x4.w + b4.w);
      }
      if (eok0 && i + 2 <= T) lc_st1(hs_rsrc, sentinel, ee0 * 4u, (unsigned)(i + 2) * BU4);
      if (qwave && i + 2 < T) {
        const int kq0 = (t0 & 63) >> 4, qcol0 = qtile * 16 + (t0 & 15);
        const unsigned qoff0 = (unsigned)((((rb * RB + kq0 * 4) * 64 + qcol0) * 16 + j) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (qcol0 < A && kq0 * 4 + r < RB && rb * RB + kq0 * 4 + r < B)
            lc_st1(qw_rsrc, sentinel, qoff0, (unsigned)((((i + 2) % 3) * B + r) * 4096));
      }
      if (eok0 && a.hd && a.rate_out > 0.f)
        kout = tnt_keep((uint64_t)ee0, a.rate_out, a.att.seed, a.site_out0 + (uint32_t)i, step_l);
    }
    unsigned spins = 0;
    float zs[4] = {0.f, 0.f, 0.f, 0.f};
    if (RB == 8) {
      float* rx = &red[0][0][0][0];
      {
        // ---- this lane's float4 of h[i]: row rg*4 + j of the block, k = 32 w + 4 cg .. + 3
        LC_TID(t1);
        const int l1 = t1 & 63, rg = l1 >> 5, cg = (l1 >> 2) & 7, xj = l1 & 3;
        const int xrow = rb * RB + rg * 4 + xj;
        const unsigned hoff = (unsigned)((xrow * U + w * CK + cg * 4) * 4);
        float4 am = make_float4(0.f, 0.f, 0.f, 0.f);
        for (;;) {
          bool ok = true;
          if (xrow < B) {
            am = lc_ld4_l2(hs_rsrc, hoff, (unsigned)i * BU4);
            ok = __float_as_uint(am.x) != LC_SENTINEL && __float_as_uint(am.y) != LC_SENTINEL &&
                 __float_as_uint(am.z) != LC_SENTINEL && __float_as_uint(am.w) != LC_SENTINEL;
          }
          if (__all(ok)) break;
          if (++spins > TNT_SEQ_SPIN_LIMIT) {
            if (l1 == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
        LCT(49);
        // ---- h[i] U for both unit blocks (off the critical path: the attention of step i runs meanwhile); the partials of both
        // blocks go to LDS together -- [block][wave][8 rows][64 + 4 columns] fills `red` exactly -- and meet ONE barrier
        float* rw = rx + (w * 8 + rg * 4) * 68 + cg * 4 + xj;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          floatx4 xa[2];
          xa[0] = (floatx4){0.f, 0.f, 0.f, 0.f}; xa[1] = xa[0];
#define TNT_X4(qd)                                                                                               \
          xa[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.x, bx[q][0][qd][0], xa[0], 3, qd, 1);                         \
          xa[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.x, bx[q][1][qd][0], xa[1], 3, qd, 1);                         \
          xa[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.y, bx[q][0][qd][0], xa[0], 3, qd, 2);                         \
          xa[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.y, bx[q][1][qd][0], xa[1], 3, qd, 2);                         \
          xa[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.z, bx[q][0][qd][1], xa[0], 3, qd, 1);                         \
          xa[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.z, bx[q][1][qd][1], xa[1], 3, qd, 1);                         \
          xa[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.w, bx[q][0][qd][1], xa[0], 3, qd, 2);                         \
          xa[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.w, bx[q][1][qd][1], xa[1], 3, qd, 2);
          TNT_X4(0) TNT_X4(1) TNT_X4(2) TNT_X4(3) TNT_X4(4) TNT_X4(5) TNT_X4(6) TNT_X4(7)
#undef TNT_X4
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int r = 0; r < 4; ++r) rw[(q * NWF * 8 + r) * 68 + ch * 32] = xa[ch][r];
        }
      }
      __syncthreads();
      {
        LC_TID(t2);
        const int erow2 = (t2 & 255) >> 4;
        if (t2 < 512 && erow2 < 8) {
          const float* rr = rx + ((t2 >> 8) * NWF * 8 + erow2) * 68 + (t2 & 15) * 4;
          float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int k = 0; k < NWF; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(rr + k * (8 * 68));
            sacc.x += t.x; sacc.y += t.y; sacc.z += t.z; sacc.w += t.w;
          }
          zs[0] = sacc.x; zs[1] = sacc.y; zs[2] = sacc.z; zs[3] = sacc.w;
        }
      }
    } else {
    // ---- A fragments = h[i] (this wave's K chunk)
      float av[SS];
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < SS / 4; ++q) {
          const float4 t = arow < B ? tnt_ld4_l2(hs_rsrc, (unsigned)(((long)i * BU + (long)arow * U + w * CK + q * 16 + kq * 4) * 4))
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
          av[4 * q + 0] = t.x; av[4 * q + 1] = t.y; av[4 * q + 2] = t.z; av[4 * q + 3] = t.w;
          ok = ok && __float_as_uint(t.x) != LC_SENTINEL && __float_as_uint(t.y) != LC_SENTINEL &&
               __float_as_uint(t.z) != LC_SENTINEL && __float_as_uint(t.w) != LC_SENTINEL;
        }
        if (__all(ok)) break;
        if (++spins > TNT_SEQ_SPIN_LIMIT) {
          if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
      LCT(49);
      // ---- h[i] U for both unit blocks (off the critical path: the attention of step i runs meanwhile)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        floatx4 acc[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) acc[gg] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < SS; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[q][s].x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[q][s].y, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[q][s].z, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[q][s].w, acc[3], 0, 0, 0);
        }
        if (q == 1) __syncthreads();           // block 0's sums have been read
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[w][gg][kq * 4 + r][lr] = acc[gg][r];
        __syncthreads();
        if (tid < 512 && eq == q) {
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) {
            float sacc = 0.f;
#pragma unroll
            for (int k = 0; k < NWF; ++k) sacc += red[k][gg][erow][ecol];
            zs[gg] = sacc;
          }
        }
      }
    }
    LCT(50);
    // ---- the attention's context of step i.  Polled by as few lanes as cover it (16-byte loads: RB rows x D / 4 chunks --
    // one wave at the benchmark's size): the rows of a block are a few adjacent cache lines behind ONE L2 channel, and with
    // every wave of the group's 16 LSTM workgroups polling them the hand-off took 1.4 us instead of 0.5.
    {
      LC_TID(t3);
      const int l3 = t3 & 63, kq3 = l3 >> 4, lr3 = l3 & 15;
      float cb[NQ];
#pragma unroll
      for (int sq = 0; sq < NQ; ++sq) cb[sq] = wcb_l[(w * 8 + sq) * 64 + l3];
      const int D4 = D >> 2;
      if (t3 < RB * D4) {                                    // (wave-uniform where RB D / 4 is a multiple of 64; else the tail wave diverges)
        const int prow_ = t3 / D4, pc_ = t3 - prow_ * D4;
        const bool pm = rb * RB + prow_ < B;
        const unsigned coff = (unsigned)((((rb * RB + prow_) * D) + pc_ * 4) * 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        spins = 0;
        if (pm) for (;;) {
          v = lc_ld4_l2(cx_rsrc, coff, (unsigned)(i * B * D * 4));
          const bool ok = __float_as_uint(v.x) != LC_SENTINEL && __float_as_uint(v.y) != LC_SENTINEL &&
                          __float_as_uint(v.z) != LC_SENTINEL && __float_as_uint(v.w) != LC_SENTINEL;
          if (ok) break;                                     // per lane: a lane leaves the loop when ITS chunk is in
          if (++spins > TNT_SEQ_SPIN_LIMIT) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
        *reinterpret_cast<float4*>(&ctx_l[prow_ * 64 + pc_ * 4]) = v;
      }
      __syncthreads();
      float4 ca[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int d0 = DH * kh + 16 * u + 4 * kq3;
        ca[u] = (lr3 < RB && d0 < D) ? *reinterpret_cast<const float4*>(&ctx_l[lr3 * 64 + d0]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      LCT(51);
      floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[u].x, cb[4 * u + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[u].y, cb[4 * u + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[u].z, cb[4 * u + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[u].w, cb[4 * u + 3], acc, 0, 0, 0);
      }
      float* zw = zc + (kh * 16 + kq3 * 4) * ZCLD + t8 * 16 + lr3;
#pragma unroll
      for (int r = 0; r < 4; ++r) zw[r * ZCLD] = acc[r];
    }
    LCT(53);
    __syncthreads();
    LCT(54);
    {
      LC_TID(t4);
      const int erow4 = (t4 & 255) >> 4, ecol4 = t4 & 15, eq4 = t4 >> 8, eb4 = rb * RB + erow4;
      if (t4 < 512 && erow4 < RB && eb4 < B) {
        const unsigned ee4_ = (unsigned)(eb4 * U + (2 * j + eq4) * 16 + ecol4) * 4u;
        const float4 zx = zx_l[t4];
        float z[4] = {zx.x + zs[0], zx.y + zs[1], zx.z + zs[2], zx.w + zs[3]};
        const float4 c0 = *reinterpret_cast<const float4*>(zc + erow4 * ZCLD + (eq4 * 16 + ecol4) * 4);
        const float4 c1 = *reinterpret_cast<const float4*>(zc + (16 + erow4) * ZCLD + (eq4 * 16 + ecol4) * 4);
        z[0] += c0.x + c1.x; z[1] += c0.y + c1.y; z[2] += c0.z + c1.z; z[3] += c0.w + c1.w;
        const float gi = tnt_sigmoid_fast(z[0]), gf = tnt_sigmoid_fast(z[1]), gg = tnt_tanh(z[2]), go = tnt_sigmoid_fast(z[3]);
        const float c2 = gf * cp + gi * gg;
        const float h2 = go * tnt_tanh(c2);
        LCT(55);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's reset of hs[i+2] is in L2 first
        lc_st1(hs_rsrc, h2, ee4_, (unsigned)(i + 1) * BU4);
        hq_l[erow4 * 36 + eq4 * 16 + ecol4] = h2;
        lc_st1(cs_rsrc, c2, ee4_, (unsigned)(i + 1) * BU4);
        lc_st4(gt_rsrc, make_float4(gi, gf, gg, go), ee4_ * 4u, (unsigned)i * BU4 * 4u);
        if (a.hd) lc_st1(hd_rsrc, kout ? h2 * oscale : 0.f, ee4_, (unsigned)i * BU4);
        cp = c2;
        if (i + 1 < T) x4 = lc_ld4(xz_rsrc, ee4_ * 4u, (unsigned)(i + 1) * BU4 * 4u);
      }
    }
    LCT(52);
    if (i < 36) LCS(2 + i);
    // `red` is rewritten next step only behind barriers that every thread passes after this point
    __syncthreads();
    // ---- this workgroup's partial of q for the attention of step i + 1 (the reset of the slot behind it is long drained)
    LCT_T(56, 512);
    if (qwave && i + 1 < T) {
      LC_TID(t5);
      const int l5 = t5 & 63, kq5 = l5 >> 4, lr5 = l5 & 15, qcol5 = qtile * 16 + lr5;
      const float* qa_p = hq_l + lr5 * 36 + kq5;
      const float* qb_p = w2s_l + kq5 * 64 + qcol5;
      floatx4 t = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float qa_[4], qb_[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) { qa_[s2] = qa_p[16 * h + 4 * s2]; qb_[s2] = qb_p[(16 * h + 4 * s2) * 64]; }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) t = __builtin_amdgcn_mfma_f32_16x16x4f32(qa_[s2], qb_[s2], t, 0, 0, 0);
      }
      LCT_T(57, 512);
      const unsigned qoff5 = (unsigned)((((rb * RB + kq5 * 4) * 64 + qcol5) * 16 + j) * 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (qcol5 < A && kq5 * 4 + r < RB && rb * RB + kq5 * 4 + r < B)
          lc_st1(qw_rsrc, t[r], qoff5, (unsigned)((((i + 1) % 3) * B + r) * 4096));
      LCT_T(58, 512);
    }
  }
#undef LC_TID
  if (ub == 16) LCT_DUMP(48, 60);
  LCS(38); LCS(39);
  if (ub == 16) LCS_DUMP(1);
  tnt_seq_leave(a.sync, xcc, a.guard_out);
}
}  // namespace

extern "C" int32_t tnt_lc_seq_fwd_drop_f32(const float* F, const float* P, const float* W2, const float* b2, const float* v,
                                      const float* bv, float* qpre, float* alpha, float* ctx, float* ctx_d,
                                      const uint8_t* keep4, int64_t keep_stride, const float* xz, const float* Wc,
                                      const float* Ur, const float* xz_bias, float* hs, float* cs, float* gates, int32_t T,
                                      int32_t B, int32_t R, int32_t D, int32_t A, int32_t U, float slope, float rate_attn,
                                      float rate_in, int32_t in_lwidth, uint64_t seed, uint32_t site_attn0,
                                      uint32_t site_in0, const uint32_t* step_dev, float* hd, float rate_out,
                                      uint32_t site_out0, float* work, uint32_t* sync, float* guard_out, void* stream) {
  if (T <= 0 || sync == nullptr || work == nullptr || U != 512 || B <= 0 || B > 128) return TNT_BADARG(24);
  if (hd && !(rate_out >= 0.f && rate_out < 1.f)) return TNT_BADARG(36);
  if (!wide_ok(R, D, A) || R > 512 || D > 64 || A > 64) return TNT_BADARG(21);
  if ((long)(T + 1) * B * U * 16 >= (1L << 32)) return TNT_BADARG(19);
  if (!tnt_aligned16(P) || !tnt_aligned16(F) || !tnt_aligned16(W2) || !tnt_aligned16(v) || !tnt_aligned16(xz) ||
      !tnt_aligned16(Wc) || !tnt_aligned16(Ur) || !tnt_aligned16(gates) || !tnt_aligned16(ctx_d)) return TNT_BADARG(1);
  LcSeqArgs a{};
  AttArgs& g = a.att;
  g.F = F; g.P = P; g.W2 = W2; g.b2 = b2; g.v = v; g.bv = bv; g.qpre = qpre; g.alpha = alpha; g.ctx = ctx; g.ctx_d = ctx_d;
  g.s_out = nullptr; g.B = B; g.R = R; g.D = D; g.A = A; g.U = U; g.in_lwidth = in_lwidth; g.slope = slope;
  g.rate_attn = rate_attn; g.rate_in = rate_in; g.seed = seed; g.site_attn = site_attn0; g.site_in = site_in0; g.step = 0;
  g.step_dev = step_dev; g.keep4 = keep4;
  a.keep_stride = keep_stride; a.xz = xz; a.Wc = Wc; a.Ur = Ur; a.zbias = xz_bias; a.hs = hs; a.cs = cs; a.gates = gates;
  a.T = T; a.sync = sync; a.guard_out = guard_out; a.hd = hd; a.rate_out = hd ? rate_out : 0.f; a.site_out0 = site_out0;
  a.qp = work;
  // row passes of the attention phase held in registers: as few as R needs (the LSTM weights are resident next to them)
  const int g4 = (A <= 32 && D <= 32) ? 8 : 16, np = (R + WT / g4 - 1) / (WT / g4);
  void (*kern)(LcSeqArgs) = nullptr;
  static const bool rb16 = getenv("TNT_SEQ_RB16") && atoi(getenv("TNT_SEQ_RB16")) != 0;               // A/B switch
  if (B <= 64 && !rb16) {
    if (g4 == 8) kern = np <= 3 ? lc_seq_fwd_kernel<8, 3, 8> : lc_seq_fwd_kernel<8, 6, 8>;
    else kern = np <= 3 ? lc_seq_fwd_kernel<16, 3, 8> : lc_seq_fwd_kernel<16, 6, 8>;
  } else {
    if (g4 == 8) kern = np <= 3 ? lc_seq_fwd_kernel<8, 3, 16> : lc_seq_fwd_kernel<8, 6, 16>;
    else kern = np <= 3 ? lc_seq_fwd_kernel<16, 3, 16> : lc_seq_fwd_kernel<16, 6, 16>;
  }
  if (np > 6) return TNT_BADARG(21);
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LC_SEQ_LDS_BYTES) != hipSuccess)
    return TNT_BADARG(90);
  hipLaunchKernelGGL(kern, dim3(256), dim3(1024), LC_SEQ_LDS_BYTES, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_lc_seq_fwd_f32(const float* F, const float* P, const float* W2, const float* b2, const float* v,
                                      const float* bv, float* qpre, float* alpha, float* ctx, float* ctx_d,
                                      const uint8_t* keep4, int64_t keep_stride, const float* xz, const float* Wc,
                                      const float* Ur, const float* xz_bias, float* hs, float* cs, float* gates, int32_t T,
                                      int32_t B, int32_t R, int32_t D, int32_t A, int32_t U, float slope, float rate_attn,
                                      float rate_in, int32_t in_lwidth, uint64_t seed, uint32_t site_attn0,
                                      uint32_t site_in0, const uint32_t* step_dev, float* work, uint32_t* sync,
                                      float* guard_out, void* stream) {
  return tnt_lc_seq_fwd_drop_f32(F, P, W2, b2, v, bv, qpre, alpha, ctx, ctx_d, keep4, keep_stride, xz, Wc, Ur, xz_bias, hs, cs,
                                 gates, T, B, R, D, A, U, slope, rate_attn, rate_in, in_lwidth, seed, site_attn0, site_in0,
                                 step_dev, nullptr, 0.f, 0, work, sync, guard_out, stream);
}

extern "C" int32_t tnt_lc_seq_fwd_work_floats(int32_t B) { return B > 0 && B <= 128 ? 3 * B * 1024 : 0; }

// ---------------------------------------------------------------------------------------------------
// The backward chain of the attention captioner (tape.gradient through lc_NIC.py:244-256: for i = T-1 .. 0 the LSTM step
// backward, then the attention backward of the same step, whose query gradient feeds the LSTM step before) as ONE
// persistent launch, with the forward chain's placement: per XCD / block of 16 samples, slots 0..15 run the ATTENTION
// backward of one sample each, slots 16..31 the LSTM backward of 32 units each.
//   LSTM workgroup: the "push" BPTT of lstm_seq_bwd_kernel (lstm.hip), two 16-unit blocks per workgroup: its Ur^T slices
//     stay in registers, its dz tiles in LDS; per step it multiplies the tiles into partial da tiles for all 32 unit blocks
//     (ring of three exchange buffers, the tiles are their own flags), gathers the 32 partials of its own blocks, adds the
//     attention's query gradient dh_att of the step behind, runs the cell backward, and leaves the partial context
//     gradient of its 32 units (dz . Wc^T, Wc slice in LDS) for the attention workgroups.
//   attention workgroup: its sample's F and P rows AND the dF / dP accumulators stay in registers for all T steps (the
//     per-step kernel reads and rewrites both accumulators every step: 11.8 of its 17.7 MB of traffic), W2 in LDS
//     (A <= 32), dv / dbv accumulators in LDS; it waits for the 16 context-gradient parts of its sample, runs the
//     attention backward, and publishes dh_att = dq W2^T for the LSTM workgroups.  dP, dF and dvb are WRITTEN once at the
//     end (no zero fill, no accumulation by the caller).
//   Overlap: while the attention backward of step i runs, the LSTM workgroups already push / gather dz_i Ur^T for step
//   i - 1; only the cell backward waits for dh_att.
// Hand-offs are data-polling on the sentinel with the reset-before-publish rule of lstm_seq_bwd_kernel<POLL>: every
// hand-off buffer is a ring of three indexed by the exchange number, the thread that publishes a chunk resets the same
// chunk of the next buffer first (and drains), a consumer reads every chunk of a producer each step.
namespace {
struct LcSeqBwdArgs {
  AttArgs att;             // F, P, W2, v; qpre_in / alpha_in / dqpre / keep4 point at step 0; dP, dF, dvb: outputs
  long keep_stride;
  const float* Ur;         // [U][U][4]
  const float* Wc;         // [D][U][4]
  const float* dout;       // [T][B][U]   gradient w.r.t. the LSTM outputs
  const float* gates;      // [T][B][U][4]
  const float* cs;         // [T+1][B][U]
  float* dz;               // [T][B][U][4]
  float* xch;              // 3 * nrb * 32 * 32 * 256   partial da tiles
  float* dhx;              // 3 * B * U                 dh_att
  float* parts;            // 3 * nrb * 16 * 16 * 64    context-gradient parts
  float rate_out; uint32_t site_out0;   // Dropout' of the LSTM outputs applied to dout as it is read (one site per step)
  int T;
  unsigned* sync; float* guard_out;
};
constexpr int LB_DZLD = 68;
constexpr int LB_RED_FLOATS = 16 * 4 * 16 * 17;                  // gather sums [2][16][256] / context-part tiles [16 waves][4][16][17]
constexpr int LB_LDS_FLOATS = 2 * 16 * LB_DZLD + LB_RED_FLOATS;  // LSTM role: dz tiles, reduction buffer
constexpr int LB_W2LD = 68;                                      // row stride of the LSTM role's W2 slice [32][64] (16-byte rows, conflict-free b128 reads)
constexpr int LB_LSTM_FLOATS = LB_LDS_FLOATS + 32 * LB_W2LD + 16 * 64 + 16 * 2 * 4 * 64;   // + W2 slice + the 16 samples' dq + Wc operands
constexpr int LB_PF_FLOATS = 35 * 1024;                          // attention role: P, F rows and the dF accumulator of the sample, R (A + 2 D) <= this
constexpr int LB_LDS_BYTES = (LB_LSTM_FLOATS > LB_PF_FLOATS ? LB_LSTM_FLOATS : LB_PF_FLOATS) * 4 + 16;

template <int G4, int NP, int RB>
__global__ __launch_bounds__(1024) void lc_seq_bwd_kernel(LcSeqBwdArgs a) {
  constexpr int RPP = WT / G4, NTW = 2, NWB = 16;
  extern __shared__ __attribute__((aligned(16))) float lb_lds[];
  unsigned* s_slot = reinterpret_cast<unsigned*>(lb_lds + (LB_LDS_BYTES - 16) / 4);
  __shared__ __attribute__((aligned(16))) float wred[WW][64], wred2[WW][64];
  __shared__ __attribute__((aligned(16))) float qs[64], qe[64], dcs[4 * 64];
  __shared__ __attribute__((aligned(16))) float red_l[WW];
  __shared__ int qbig_l;
  LCT_DECL
  LCS_DECL
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, lr = lane & 15;
  const AttArgs& g = a.att;
  const int U = g.U, B = g.B, D = g.D, R = g.R, A = g.A, T = a.T;
  const unsigned xcc = tnt_xcc_id();
  const int nrb = (B + RB - 1) / RB;
  if ((int)xcc >= nrb) return;
  unsigned* bar = a.sync + xcc * 64;
  unsigned* err = a.sync + TNT_SEQ_ERR;
  const TntSeqSlot slot = tnt_seq_enter(a.sync, xcc, s_slot);
  if (slot.ub < 0) {
    if (tid == 0 && a.guard_out) a.guard_out[0] = 2.f;
    return;
  }
  const int ub = slot.ub, rb = __builtin_amdgcn_readfirstlane((int)xcc);
  LCS(0);
  const long BU = (long)B * U;
  const float sentinel = __uint_as_float(LC_SENTINEL);
  const float4 sent4 = make_float4(sentinel, sentinel, sentinel, sentinel);
  const __amdgpu_buffer_rsrc_t x_rsrc = tnt_rsrc(a.xch, (unsigned)(3u * nrb * 32u * 32u * 1024u));
  const __amdgpu_buffer_rsrc_t dq_rsrc = tnt_rsrc(a.dhx, (unsigned)(3 * B * 64 * 4));          // [3][B][64]: dq of the attention
  const __amdgpu_buffer_rsrc_t pt_rsrc = tnt_rsrc(a.parts, (unsigned)(3u * nrb * 16u * 1024u * 4u));
  auto poll_fail = [&](unsigned& spins) {          // true: give up (error word set here or elsewhere)
    if (++spins > TNT_SEQ_SPIN_LIMIT) {
      if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
    return (spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  };

  if (ub < 16) {
    // =========================================================== attention role: sample ab
    // Everything of the sample that outlives a step stays in registers for all T steps: e^{2P} (tanh(P + q) is recomputed as
    // 1 - 2 / (e^{2P} e^{2q} + 1), one transcendental per element; P itself where that could lose accuracy, as in the forward
    // chain), F, and the accumulators dP, dF, dv, dbv -- 48 + 5 registers; the reductions of dv and dbv happen ONCE, behind the
    // loop.  A step has three workgroup barriers: behind the context gradient (the 16 parts of the LSTM workgroups, summed by 4 D
    // lanes with 16-byte polls), behind sum_r alpha_r dalpha_r (one value per wave), behind the per-wave sums of dq.  The dF
    // update and the stores of the step sit behind the publish of dq.
    const int ab = rb * RB + ub;
    const bool live = ub < RB && ab < B;
    const int c4 = tid % G4, rl = tid / G4;
    const bool cokA = c4 * 4 < A, cokD = c4 * 4 < D;
    float4 pv[NP], fv[NP], dpa[NP], dfa[NP];
    float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f), dva = make_float4(0.f, 0.f, 0.f, 0.f);
    float dbv_acc = 0.f;
    bool direct = false;
    const float scale_a = lc_uniform(g.rate_attn > 0.f ? 1.f / (1.f - g.rate_attn) : 1.f);
    const float scale_in = lc_uniform(g.rate_in > 0.f ? 1.f / (1.f - g.rate_in) : 1.f);
    if (live) {
      bool big = false;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int r = p * RPP + rl;
        pv[p] = (cokA && r < R) ? *reinterpret_cast<const float4*>(g.P + ((long)ab * R + r) * A + c4 * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        fv[p] = (cokD && r < R) ? *reinterpret_cast<const float4*>(g.F + ((long)ab * R + r) * D + c4 * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        dpa[p] = make_float4(0.f, 0.f, 0.f, 0.f); dfa[p] = dpa[p];
        big = big || !(fabsf(pv[p].x) <= 40.f && fabsf(pv[p].y) <= 40.f && fabsf(pv[p].z) <= 40.f && fabsf(pv[p].w) <= 40.f);
      }
      if (cokA) v4 = *reinterpret_cast<const float4*>(g.v + c4 * 4);
      if (tid < A) a.dhx[(long)ab * 64 + tid] = sentinel;         // this thread's element of the dq hand-off: buffer 0 armed
      if (tid == 0) qbig_l = -1;
      direct = __syncthreads_or(big ? 1 : 0) != 0;
      if (!direct) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          pv[p].x = __expf(2.f * pv[p].x); pv[p].y = __expf(2.f * pv[p].y);
          pv[p].z = __expf(2.f * pv[p].z); pv[p].w = __expf(2.f * pv[p].w);
        }
      }
    }
    tnt_seq_group_barrier(bar, ub, tnt_seq_target(slot.epoch, 1), err);
    LCS(1);
    const uint32_t step = g.step + (g.step_dev ? g.step_dev[0] : 0u);
    const bool stored = g.keep4 != nullptr && g.rate_attn > 0.f;
    // per-step operands through buffer resources: one 32-bit lane offset each, the step's slab as a scalar offset
    const __amdgpu_buffer_rsrc_t al_rsrc = tnt_rsrc(g.alpha_in, (unsigned)((long)T * B * R * 4));
    const __amdgpu_buffer_rsrc_t kp_rsrc = tnt_rsrc(g.keep4, stored ? (unsigned)((long)T * a.keep_stride) : 0u);
    const __amdgpu_buffer_rsrc_t qi_rsrc = tnt_rsrc(g.qpre_in, (unsigned)((long)T * B * A * 4));
    const __amdgpu_buffer_rsrc_t qo_rsrc = tnt_rsrc(g.dqpre, (unsigned)((long)T * B * A * 4));
    const unsigned al_off = (unsigned)((ab * R + rl) * 4), kp_off = (unsigned)((ab * R + rl) * (A >> 2) + c4);
    if (live) for (int i0 = T - 1; i0 >= 0; --i0) {
      const int i = __builtin_amdgcn_readfirstlane(i0);
      const int pi = (T - 1 - i) % 3, pn = (pi + 1) % 3;
      const uint32_t site_attn = g.site_attn + (uint32_t)i, site_in = g.site_in + (uint32_t)i;
      if (tid < A && i > 0) lc_st1(dq_rsrc, sentinel, (unsigned)((ab * 64 + tid) * 4), (unsigned)(pn * B * 256));
      LCT(0);
      // ---- operands of this step that do not depend on the chain
      // the context gradient arrives as 16 parts [part][row][d]: thread G4 part + c polls the float4 (part, d = 4 c .. 4 c + 3);
      // the lanes of a wave that share c are summed by shuffles, lane c < G4 of each polling wave applies the input-dropout
      // decisions of its 4 elements (Philox calls, taken here) and files the wave's sum
      constexpr int NWP = 16 * G4 / 64;                      // polling waves
      const int ppart = tid / G4;                            // (c = c4)
      const bool pmine = tid < 16 * G4 && cokD;
      float al[NP];
      uint32_t mk[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int r = p * RPP + rl;
        al[p] = r < R ? lc_ld1(al_rsrc, al_off + (unsigned)(p * RPP * 4), (unsigned)(i * B * R * 4)) : 0.f;
        mk[p] = (stored && cokA && r < R) ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(kp_rsrc, (int)(kp_off + (unsigned)(p * RPP * (A >> 2))),
                                                                                            (int)((long)i * a.keep_stride), 0) : 0xFu;
      }
      float qp = 0.f;
      if (tid < A) qp = lc_ld1(qi_rsrc, (unsigned)((ab * A + tid) * 4), (unsigned)(i * B * A * 4));
      // (the loads above are in flight -- HBM latency -- while the Philox calls run; the memory counter is in-order, so a poll
      // issued behind loads that are still out would wait for them however early the hand-off lands)
      bool kin[4] = {true, true, true, true};
      if (pmine && lane < G4 && g.rate_in > 0.f) {
        const uint64_t e0 = (uint64_t)ab * (uint64_t)g.in_lwidth + c4 * 4;
        if ((e0 & 3u) == 0u) tnt_keep4(e0, g.rate_in, g.seed, site_in, step, kin);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) kin[e] = tnt_keep(e0 + e, g.rate_in, g.seed, site_in, step);
        }
      }
      if (tid < A) {
        const float q = qp > 0.f ? qp : qp * g.slope;
        qs[tid] = q; qe[tid] = __expf(2.f * q);
        if (!(fabsf(q) <= 40.f)) qbig_l = i;
      }
      // ---- the 16 context-gradient parts of this sample
      if (tid < 16 * G4) {                                   // whole waves
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned spins = 0;
        const unsigned off = (unsigned)((((((pi * nrb + rb) * 16 + ppart) * 16 + ub) * 64) + c4 * 4) * 4);
        LCT(7);
        if (pmine) for (;;) {
          v = tnt_ld4_l2(pt_rsrc, off);
          const bool ok = __float_as_uint(v.x) != LC_SENTINEL && __float_as_uint(v.y) != LC_SENTINEL &&
                          __float_as_uint(v.z) != LC_SENTINEL && __float_as_uint(v.w) != LC_SENTINEL;
#ifdef TNT_LC_TRACE
          if (tid == 0 && rb == 0 && i == LCT_STEP && spins < 6) lct_l[26 + spins] = wall_clock64();       // the first polls' returns
#endif
          if (ok) break;                                     // per lane: a lane leaves the loop when ITS chunk is in
          if (++spins > TNT_SEQ_SPIN_LIMIT) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
        LCT(1);
#ifdef TNT_LC_TRACE
        if (tid == 0 && rb == 0 && i == LCT_STEP) lct_l[8] = spins;
#endif
        v.x = stride_sum<G4>(v.x); v.y = stride_sum<G4>(v.y); v.z = stride_sum<G4>(v.z); v.w = stride_sum<G4>(v.w);
        if (lane < G4 && cokD) {
          if (g.rate_in > 0.f) {
            v.x = kin[0] ? v.x * scale_in : 0.f; v.y = kin[1] ? v.y * scale_in : 0.f;
            v.z = kin[2] ? v.z * scale_in : 0.f; v.w = kin[3] ? v.w * scale_in : 0.f;
          }
          *reinterpret_cast<float4*>(&dcs[w * 64 + c4 * 4]) = v;
        }
      }
      __syncthreads();
      LCT(2);
      // ---- dalpha[r] = dctx . F[r] (+ the alpha term of the SAM loss), and the softmax' dot product
      float4 dc4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cokD) {
#pragma unroll
        for (int k = 0; k < NWP; ++k) {
          const float4 t = *reinterpret_cast<const float4*>(&dcs[k * 64 + c4 * 4]);
          dc4.x += t.x; dc4.y += t.y; dc4.z += t.z; dc4.w += t.w;
        }
      }
      float da[NP];
      float dot = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float t = dc4.x * fv[p].x + dc4.y * fv[p].y + dc4.z * fv[p].z + dc4.w * fv[p].w;
        da[p] = adj_sum<G4>(t) + g.alpha_mse * (al[p] - 1.f);           // rows past R: al = 0, nothing of them is used
        dot += al[p] * da[p];
      }
      dot = tnt_wave_sum(dot) * (1.f / G4);                  // every row sits in G4 lanes: exact scaling
      if (lane == 0) red_l[w] = dot;
      __syncthreads();
      LCT(3);
      {
        const float4* rr = reinterpret_cast<const float4*>(red_l);
        dot = 0.f;
#pragma unroll
        for (int k = 0; k < WW / 4; ++k) { const float4 t = rr[k]; dot += (t.x + t.y) + (t.z + t.w); }
      }
      LCT(4);
      // ---- through e = s_d . v, dropout, tanh
      float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
      {
        const bool fast = !direct && qbig_l != i;
        const float4 q4 = cokA ? *reinterpret_cast<const float4*>(fast ? &qe[c4 * 4] : &qs[c4 * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int r = p * RPP + rl;
          if (!(cokA && r < R)) continue;
          const long e = ((long)ab * R + r) * A + c4 * 4;
          float s0, s1, s2, s3;
          if (fast) {
            s0 = 1.f - 2.f * __builtin_amdgcn_rcpf(pv[p].x * q4.x + 1.f); s1 = 1.f - 2.f * __builtin_amdgcn_rcpf(pv[p].y * q4.y + 1.f);
            s2 = 1.f - 2.f * __builtin_amdgcn_rcpf(pv[p].z * q4.z + 1.f); s3 = 1.f - 2.f * __builtin_amdgcn_rcpf(pv[p].w * q4.w + 1.f);
          } else {
            const float4 p4 = direct ? pv[p] : *reinterpret_cast<const float4*>(g.P + e);
            s0 = tnt_tanh(p4.x + q4.x); s1 = tnt_tanh(p4.y + q4.y); s2 = tnt_tanh(p4.z + q4.z); s3 = tnt_tanh(p4.w + q4.w);
          }
          uint32_t kb = mk[p];
          if (g.rate_attn > 0.f && !stored) {
            bool k[4];
            tnt_keep4((uint64_t)e, g.rate_attn, g.seed, site_attn, step, k);
            kb = (k[0] ? 1u : 0u) | (k[1] ? 2u : 0u) | (k[2] ? 4u : 0u) | (k[3] ? 8u : 0u);
          }
          const float k0 = (kb & 1u) ? scale_a : 0.f, k1 = (kb & 2u) ? scale_a : 0.f, k2 = (kb & 4u) ? scale_a : 0.f, k3 = (kb & 8u) ? scale_a : 0.f;
          const float de = al[p] * (da[p] - dot);
          dva.x += s0 * k0 * de; dva.y += s1 * k1 * de; dva.z += s2 * k2 * de; dva.w += s3 * k3 * de;
          const float d0 = de * v4.x * k0 * (1.f - s0 * s0), d1 = de * v4.y * k1 * (1.f - s1 * s1);
          const float d2 = de * v4.z * k2 * (1.f - s2 * s2), d3 = de * v4.w * k3 * (1.f - s3 * s3);
          dpa[p].x += d0; dpa[p].y += d1; dpa[p].z += d2; dpa[p].w += d3;
          dq.x += d0; dq.y += d1; dq.z += d2; dq.w += d3;
          if (c4 == 0) dbv_acc += de;
        }
        dq.x = stride_sum<G4>(dq.x); dq.y = stride_sum<G4>(dq.y); dq.z = stride_sum<G4>(dq.z); dq.w = stride_sum<G4>(dq.w);
        if (lane < G4) *reinterpret_cast<float4*>(&wred2[w][lane * 4]) = dq;
      }
      __syncthreads();
      LCT(5);
      // ---- hand dq = d(query pre-activation) to the LSTM workgroups: each takes dh_att = dq W2^T for its own 32 units
      if (tid < A) {
        float tq = 0.f;
#pragma unroll
        for (int k = 0; k < WW; ++k) tq += wred2[k][tid];
        const float dqp = qp > 0.f ? tq : tq * g.slope;
        if (i > 0) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's reset of the next buffer's element is in L2 first
          lc_st1(dq_rsrc, dqp, (unsigned)((ab * 64 + tid) * 4), (unsigned)(pi * B * 256));
        }
        lc_st1(qo_rsrc, dqp, (unsigned)((ab * A + tid) * 4), (unsigned)(i * B * A * 4));      // behind the publish: its drain is off the critical path
      }
      LCT(6);
#ifdef TNT_LC_TRACE
      if (tid == 0 && rb == 0 && i == LCT_STEP) lc_pub[2][ub] = wall_clock64();
#endif
      if (T - 1 - i < 36) LCS(2 + T - 1 - i);
      // ---- behind the publish: dF[r] += alpha[r] dctx
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        dfa[p].x += al[p] * dc4.x; dfa[p].y += al[p] * dc4.y; dfa[p].z += al[p] * dc4.z; dfa[p].w += al[p] * dc4.w;
      }
      // dcs / qs / qe are rewritten next step in front of its first barrier, by threads that have passed this step's last one
      // (every reader is in front of that); red_l and wred2 behind the next step's first / second barrier
    }
    LCS(38);
    if (live) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int r = p * RPP + rl;
        if (r < R && cokA) *reinterpret_cast<float4*>(g.dP + ((long)ab * R + r) * A + c4 * 4) = dpa[p];
        if (r < R && cokD) *reinterpret_cast<float4*>(g.dF + ((long)ab * R + r) * D + c4 * 4) = dfa[p];
      }
      // dv and dbv: reduced once
      dva.x = stride_sum<G4>(dva.x); dva.y = stride_sum<G4>(dva.y); dva.z = stride_sum<G4>(dva.z); dva.w = stride_sum<G4>(dva.w);
      __syncthreads();
      if (lane < G4) *reinterpret_cast<float4*>(&wred[w][lane * 4]) = dva;
      const float dbv = block_sum_w(dbv_acc, red_l);
      if (tid < A) {
        float tv = 0.f;
#pragma unroll
        for (int k = 0; k < WW; ++k) tv += wred[k][tid];
        g.dvb[(long)ab * (A + 1) + tid] = tv;
      }
      if (tid == 0) g.dvb[(long)ab * (A + 1) + A] = dbv;
    }
    if (ub == 0) { LCT_DUMP(0, 16); LCT_DUMP(26, 32); }
    LCS(39);
    if (ub == 0) LCS_DUMP(2);
    tnt_seq_leave(a.sync, xcc, a.guard_out);
    return;
  }

  // ============================================================= LSTM role: unit blocks 2 j and 2 j + 1 of the 16 samples
  const int j = ub - 16;
  float* dzs = lb_lds;                                         // [2][16][LB_DZLD]
  float* red = lb_lds + 2 * 16 * LB_DZLD;                      // [2][NWB][256], later [NWB][4][16][17]
  float* w2s = lb_lds + LB_LDS_FLOATS;                         // [32][LB_W2LD]: W2[32 j .. 32 j + 32][A], columns past A zero
  float* dha_l = w2s + 32 * LB_W2LD;                           // [16][36]: dh_att of the 16 samples for this workgroup's 32 units
  constexpr int NUA = G4 == 8 ? 2 : 4;                         // float4 per lane of a dq row: A <= 32 -> 2, A <= 64 -> 4
  // resident B operands: Ur^T[k][n] = Ur[n][vub * 64 + k], lane (kq, lr) of column tile t holds n = w*32 + t*16 + lr and the
  // contraction indices k = kq*16 + ks (lstm_seq_bwd_kernel's layout)
  float bw[2][NTW][16];
  auto load_bw = [&](int q, const float* Ur) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const float* src = Ur + ((long)(w * 32 + t * 16 + lr) * U + (2 * j + q) * 16) * 4 + kq * 16;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 x = *reinterpret_cast<const float4*>(src + 4 * c);
        bw[q][t][4 * c + 0] = x.x; bw[q][t][4 * c + 1] = x.y; bw[q][t][4 * c + 2] = x.z; bw[q][t][4 * c + 3] = x.w;
      }
    }
  };
  // RB == 8: the batched 4x4x1 MFMA of lstm_seq_bwd_kernel (lstm.hip): lane l = 32 rg + 4 cg + j owns column n = 32 w + 4 cg + j
  // of the partial da and rows rg*4 .. +3; per unit block 64 instructions of 8 cycles instead of 32 of 32
  const int x_rg = lane >> 5, x_cg = (lane >> 2) & 7, x_j = lane & 3;
  float bx[2][2][8][2];                    // [unit block][k half][abid = k quad][m pair]
  if (RB == 8) {
    // A lane's 64 operands of a block are 64 consecutive floats of ONE row of Ur, a different row per lane: read straight from
    // memory that is 64 load instructions of 32 cache lines each per wave (measured: 20 us of this kernel's 32 us prologue, the
    // CU's one address path serialising 16 waves).  Staged instead: the block's [512 rows][64] slab comes in with coalesced
    // 16-byte loads (a row's 256 bytes by 16 adjacent lanes), is parked in the dynamic LDS block (free until the loop
    // starts; row stride 65: the 32 rows a wave reads side by side fall into 32 banks) and each lane picks its operands there.
    constexpr int ULD = 65;
    static_assert(512 * ULD <= LB_PF_FLOATS, "the staging slab must fit the dynamic LDS block");
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float4 st[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = tid + 1024 * k;                        // chunk c: row c / 16, floats 4 (c % 16) .. + 3
        st[k] = *reinterpret_cast<const float4*>(a.Ur + ((long)(c >> 4) * U + (2 * j + q) * 16) * 4 + (c & 15) * 4);
      }
      if (q == 1) __syncthreads();                           // block 0's operands have been picked
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = tid + 1024 * k;
        float* d = lb_lds + (c >> 4) * ULD + (c & 15) * 4;
        d[0] = st[k].x; d[1] = st[k].y; d[2] = st[k].z; d[3] = st[k].w;
      }
      __syncthreads();
      const float* src = lb_lds + (w * 32 + x_cg * 4 + x_j) * ULD;
#pragma unroll
      for (int kh2 = 0; kh2 < 2; ++kh2)
#pragma unroll
        for (int qd = 0; qd < 8; ++qd)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) bx[q][kh2][qd][pr] = src[kh2 * 32 + 4 * qd + 2 * pr + x_rg];
    }
    __syncthreads();                                         // ... before the block is carved up below
  } else {
    load_bw(0, a.Ur);
    load_bw(1, a.Ur);
  }
  // context-gradient part of this workgroup: part[16 rows][D] = dz[16][128 k] Wc^T[128 k][D] on the MFMAs; wave w owns the
  // contraction quads 2 w and 2 w + 1 (k = 4 quad + kq), its B operands (Wc[d = 16 t + lr][32 j units][k]) stay in registers
  const int ntile = (D + 15) / 16;
  // (parked in LDS, [wave][quad][tile][lane]: 8 resident VGPRs per lane are 8 more spilled ones in this role)
  float* wcb_l = dha_l + 16 * 64;                              // [16][2][4][64]
#pragma unroll
  for (int sq = 0; sq < 2; ++sq)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      wcb_l[((w * 2 + sq) * 4 + t) * 64 + lane] =
          (t * 16 + lr < D) ? a.Wc[((long)(t * 16 + lr) * U + j * 32) * 4 + (2 * w + sq) * 4 + kq] : 0.f;
  for (int e = tid; e < 32 * 64; e += WT) w2s[(e >> 6) * LB_W2LD + (e & 63)] = (e & 63) < A ? g.W2[(long)(j * 32 + (e >> 6)) * A + (e & 63)] : 0.f;
  for (int e = tid; e < 2 * 16 * LB_DZLD; e += WT) dzs[e] = 0.f;             // rows past B stay zero
  const int eq = tid >> 8, erow = (tid & 255) >> 4, ecol = tid & 15;
  const int eb = rb * RB + erow, eu = (2 * j + eq) * 16 + ecol;
  const bool eok = tid < 512 && erow < RB && eb < B;
  const long ee = (long)eb * U + eu;
  const uint32_t step_l = a.att.step + (a.att.step_dev ? a.att.step_dev[0] : 0u);
  const float oscale = lc_uniform(a.rate_out > 0.f ? 1.f / (1.f - a.rate_out) : 1.f);
  const int ridx = ((erow >> 2) * 16 + ecol) * 4 + (erow & 3);
  float dc_c = 0.f;
  // with row blocks of 8 samples only the half of a partial tile that holds rows < 8 (lanes 0..31 of the MFMA's C layout) is
  // pushed, reset and gathered
  const bool xl = kq * 4 < RB;
  // this lane's chunk of the tile (dest, src) in ring buffer buf
  auto xslot = [&](int buf, int dest, int src) { return a.xch + ((((long)(buf * nrb + rb) * 32 + dest) * 32 + src) * 256) + lane * 4; };
  // RB == 8 (4x4x1 form): this lane's float4 -- rows rg*4 .. +3 of column 4 (cg & 3) + j -- sits where the 16x16 C layout keeps
  // those rows: chunk rg*16 + 4 (cg & 3) + j of the tile for workgroup 2 w + (cg >> 2); one chunk per lane and source block
  auto xslot8 = [&](int buf, int src) {
    return a.xch + ((((long)(buf * nrb + rb) * 32 + (w * 2 + (x_cg >> 2))) * 32 + src) * 256) + (x_rg * 16 + (x_cg & 3) * 4 + x_j) * 4;
  };
  // this thread's element (row, d) of this workgroup's context-gradient part in ring buffer buf
  const int prow = tid / D, pd = tid - prow * D;
  const bool pmine = tid < 16 * D;
  // (layout [buffer][row block][part j][row][d < 64]: a part's row is one coalesced store; with the 16 parts of an element side by
  // side instead -- every workgroup scattering 4-byte stores into lines it shares with the 15 others -- the hand-off took 1.4 us
  // instead of 0.4)
  auto pslot = [&](int buf) { return a.parts + ((((long)(buf * nrb + rb) * 16 + j) * 16 + prow) * 64 + pd); };
  const __amdgpu_buffer_rsrc_t pw_rsrc = tnt_rsrc(a.parts, (unsigned)(3u * nrb * 16u * 1024u * 4u));
  const __amdgpu_buffer_rsrc_t xw_rsrc = tnt_rsrc(a.xch, (unsigned)(3u * nrb * 32u * 32u * 1024u));
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (RB == 8) *reinterpret_cast<float4*>(xslot8(0, 2 * j + q)) = sent4;
    else
#pragma unroll
      for (int t = 0; t < NTW; ++t) if (xl) *reinterpret_cast<float4*>(xslot(0, w * NTW + t, 2 * j + q)) = sent4;
  }
  if (pmine) *pslot(0) = sentinel;
  tnt_seq_group_barrier(bar, ub, tnt_seq_target(slot.epoch, 1), err);
  LCS(1);
  // (register discipline: as in the forward chain -- LC_TID phases, scalar step / slab offsets, buffer addressing)
#define LC_TID(v) int v = threadIdx.x; asm volatile("" : "+v"(v))
  const unsigned BU4 = (unsigned)BU * 4u;
  const __amdgpu_buffer_rsrc_t gt_rsrc = tnt_rsrc(a.gates, (unsigned)((long)T * BU * 16));
  const __amdgpu_buffer_rsrc_t cs_rsrc = tnt_rsrc(a.cs, (unsigned)((long)(T + 1) * BU * 4));
  const __amdgpu_buffer_rsrc_t do_rsrc = tnt_rsrc(a.dout, (unsigned)((long)T * BU * 4));
  const __amdgpu_buffer_rsrc_t dz_rsrc = tnt_rsrc(a.dz, (unsigned)((long)T * BU * 16));
  const int D4 = D >> 2;
  for (int i0 = T - 1; i0 >= 0; --i0) {
    const int i = __builtin_amdgcn_readfirstlane(i0);
    const int pi = (T - 1 - i) % 3, pn = (pi + 1) % 3;
    // ---- the step's resets first (the next parts buffer, the next exchange buffer), DRAINED (stores are acknowledged by L2 in
    // ~0.15 us), and only then the epilogue operands of the step (gates, cell states, dout: they come from HBM).  The memory
    // counter is in-order: with the loads in front, the drain that the tile stores need ("my resets are in L2 before I
    // publish") also waited for the loads, and the tiles of the waves that own epilogue threads -- those for the first eight
    // workgroups -- were published an HBM latency late; those workgroups ran 1-2 us behind the others for the whole chain
    // (profiles/r03_lc_trace.txt) and set its period.
    float4 g4;
    float cval, cprev, dout_t, okeep = 1.f;
    {
      LC_TID(t0);
      if (t0 < 16 * D && i > 0) {
        const int pr0 = t0 / D, pd0 = t0 - pr0 * D;
        lc_st1(pw_rsrc, sentinel, (unsigned)((((j * 16 + pr0) * 64) + pd0) * 4), (unsigned)((pn * nrb + rb) * 65536));
      }
      if (RB == 8 && i > 0 && i < T - 1) {
        const int l0 = t0 & 63, rg = l0 >> 5, cg = (l0 >> 2) & 7, xj = l0 & 3;
        const unsigned xoff = (unsigned)(((((w * 2 + (cg >> 2)) * 32) * 256) + (rg * 16 + (cg & 3) * 4 + xj) * 4) * 4);
        lc_st4(xw_rsrc, sent4, xoff, (unsigned)((((((T - 2 - i + 1) % 3) * nrb + rb) * 32 * 32) + 2 * j) * 1024));
      }
      LCT(23);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      LCT(24);
      // The four loads are UNCONDITIONAL (lanes without an epilogue element aim past the end of the buffers and receive zeros): as
      // the body of an `if` their results met the defaults in copies at the join, and the compiler waited for the loads -- an HBM
      // latency, 0.8 us -- right here for the copies' sake.  For the same reason the values are laundered where the cell backward
      // takes them up (the first instructions of tanh(c) had been hoisted up here, behind the load).
      const int erow0 = (t0 & 255) >> 4, eb0 = rb * RB + erow0;
      const bool ok0 = t0 < 512 && erow0 < RB && eb0 < B;
      const unsigned ee0 = ok0 ? (unsigned)(eb0 * U + (2 * j + (t0 >> 8)) * 16 + (t0 & 15)) : 0x07FFFFF0u;     // x 16 bytes: past every buffer
      g4 = lc_ld4(gt_rsrc, ee0 * 16u, (unsigned)i * BU4 * 4u);
      cval = lc_ld1(cs_rsrc, ee0 * 4u, (unsigned)(i + 1) * BU4); cprev = lc_ld1(cs_rsrc, ee0 * 4u, (unsigned)i * BU4);
      dout_t = lc_ld1(do_rsrc, ee0 * 4u, (unsigned)i * BU4);
    }
    LCT(16);
#ifdef TNT_LC_TRACE
    if (tid == 0 && rb == 0 && i == LCT_STEP) lc_pub[3][j] = wall_clock64();
#endif
    float da = 0.f;
    if (i < T - 1) {
      const int par = (T - 2 - i) % 3;                       // exchange of dz_{i+1} Ur^T; also the buffer of dh_att_{i+1}
      if (RB == 8) {
        {
          LC_TID(t1);
          const int l1 = t1 & 63, rg = l1 >> 5, cg = (l1 >> 2) & 7, xj = l1 & 3;
          // this lane's float4 of a tile for workgroup 2 w + (cg >> 2): chunk rg*16 + 4 (cg & 3) + j (where the 16x16 C layout
          // keeps rows rg*4 .. +3 of column 4 (cg & 3) + j); one chunk per lane and source block
          const unsigned xoff = (unsigned)(((((w * 2 + (cg >> 2)) * 32) * 256) + (rg * 16 + (cg & 3) * 4 + xj) * 4) * 4);
          floatx4 xa[2];
          const float* azp = dzs + (rg * 4 + xj) * LB_DZLD + cg * 4;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float4 am[2];
#pragma unroll
            for (int kh2 = 0; kh2 < 2; ++kh2) am[kh2] = *reinterpret_cast<const float4*>(azp + q * 16 * LB_DZLD + kh2 * 32);
            xa[q] = (floatx4){0.f, 0.f, 0.f, 0.f};
#define TNT_X4(k2, qd)                                                                                   \
            xa[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(am[k2].x, bx[q][k2][qd][0], xa[q], 3, qd, 1);          \
            xa[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(am[k2].y, bx[q][k2][qd][0], xa[q], 3, qd, 2);          \
            xa[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(am[k2].z, bx[q][k2][qd][1], xa[q], 3, qd, 1);          \
            xa[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(am[k2].w, bx[q][k2][qd][1], xa[q], 3, qd, 2);
            TNT_X4(0, 0) TNT_X4(0, 1) TNT_X4(0, 2) TNT_X4(0, 3) TNT_X4(0, 4) TNT_X4(0, 5) TNT_X4(0, 6) TNT_X4(0, 7)
            TNT_X4(1, 0) TNT_X4(1, 1) TNT_X4(1, 2) TNT_X4(1, 3) TNT_X4(1, 4) TNT_X4(1, 5) TNT_X4(1, 6) TNT_X4(1, 7)
#undef TNT_X4
          }
          // (this thread's reset of the buffer behind this one was drained at the top of the step)
          // The workgroup's two 16-unit blocks are summed HERE and leave as one tile per destination (source slot 2 j): half
          // the exchange traffic -- a round of the all-to-all was taking 1.3 us at ~0.8 TB/s per XCD, most of the gather.
          lc_st4(xw_rsrc, make_float4(xa[0][0] + xa[1][0], xa[0][1] + xa[1][1], xa[0][2] + xa[1][2], xa[0][3] + xa[1][3]), xoff,
                 (unsigned)((((par * nrb + rb) * 32 * 32) + 2 * j) * 1024));
        }
        LCT(17);
#ifdef TNT_LC_TRACE
        if (tid == 0 && rb == 0 && i == LCT_STEP) lc_pub[4][j] = wall_clock64();
#endif
        // (Dropout' of the LSTM outputs: the Philox call -- ~0.8 us of a SIMD for the waves that own epilogue threads: quarter-
        // rate integer multiplies -- sits HERE, behind the push, while the tiles of the other workgroups are on their way; at the
        // step's top it delayed exactly those waves' tiles, i.e. the gathers of the workgroups they are for.  Its factor meets dout
        // in the cell backward.)
        if (a.rate_out > 0.f) {
          LC_TID(tk);
          const int erowk = (tk & 255) >> 4, ebk = rb * RB + erowk;
          if (tk < 512 && erowk < RB && ebk < B)
            okeep = tnt_keep((uint64_t)(unsigned)(ebk * U + (2 * j + (tk >> 8)) * 16 + (tk & 15)), a.rate_out, a.att.seed,
                             a.site_out0 + (uint32_t)i, step_l) ? oscale : 0.f;
        }
        // ---- gather the 16 partial tiles (one per source workgroup) of each of this workgroup's two blocks: wave w takes
        // source w (both blocks' loads in flight together; `red` holds [2][NWB][256]); only lanes 0..31 of a tile hold rows < 8
        {
          LC_TID(t2);
          const int l2 = t2 & 63;
          const bool xl2 = l2 < 32;
          float4 p[2];
          unsigned spins = 0;
          for (;;) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              p[q] = xl2 ? lc_ld4_l2(x_rsrc, (unsigned)l2 * 16u, (unsigned)(((((par * nrb + rb) * 32 + 2 * j + q) * 32) + 2 * w) * 1024))
                         : make_float4(0.f, 0.f, 0.f, 0.f);
              ok = ok && __float_as_uint(p[q].x) != LC_SENTINEL && __float_as_uint(p[q].y) != LC_SENTINEL &&
                   __float_as_uint(p[q].z) != LC_SENTINEL && __float_as_uint(p[q].w) != LC_SENTINEL;
            }
            if (__all(ok)) break;
            if (poll_fail(spins)) break;
          }
#ifdef TNT_LC_TRACE
          if (rb == 0 && i == LCT_STEP && (tid == 0 || tid == 256 || tid == 512)) {      // waves 0, 4, 8: loop exit, rounds
            const int k_ = tid == 0 ? 6 : (tid == 256 ? 7 : 8);
            lc_pub[k_][j] = wall_clock64(); if (tid == 256) lc_pub[9][j] = spins + 1;
          }
#endif
#pragma unroll
          for (int q = 0; q < 2; ++q) *reinterpret_cast<float4*>(red + (q * NWB + w) * 256 + l2 * 4) = p[q];
        }
        __syncthreads();
        {
          LC_TID(t3);
          const int erow3 = (t3 & 255) >> 4;
          if (t3 < 512 && erow3 < RB) {
            const float* rp = red + (t3 >> 8) * NWB * 256 + ((erow3 >> 2) * 16 + (t3 & 15)) * 4 + (erow3 & 3);
#pragma unroll
            for (int k = 0; k < NWB; ++k) da += rp[k * 256];
          }
        }
      } else {
        if (i > 0) {
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < NTW; ++t) if (xl) *reinterpret_cast<float4*>(xslot((par + 1) % 3, w * NTW + t, 2 * j + q)) = sent4;
        }
        {
        floatx4 acc[2][NTW];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float av[16];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float4 x = *reinterpret_cast<const float4*>(dzs + (q * 16 + lr) * LB_DZLD + kq * 16 + 4 * c);
            av[4 * c + 0] = x.x; av[4 * c + 1] = x.y; av[4 * c + 2] = x.z; av[4 * c + 3] = x.w;
          }
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[q][t] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bw[q][t][ks], acc[q][t], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's resets are in L2 first
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int t = 0; t < NTW; ++t)
            if (xl) *reinterpret_cast<float4*>(xslot(par, w * NTW + t, 2 * j + q)) = make_float4(acc[q][t][0], acc[q][t][1], acc[q][t][2], acc[q][t][3]);
      }
        LCT(17);
        if (eok && a.rate_out > 0.f)
          okeep = tnt_keep((uint64_t)ee, a.rate_out, a.att.seed, a.site_out0 + (uint32_t)i, step_l) ? oscale : 0.f;
        {
          const unsigned base = (unsigned)((((par * nrb + rb) * 32 + 2 * j) * 32) * 1024) + (unsigned)lane * 16u;
          float4 p[2][2];
          unsigned spins = 0;
          for (;;) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                p[q][h] = xl ? tnt_ld4_l2(x_rsrc, base + (unsigned)q * 32u * 1024u + (unsigned)(w + 16 * h) * 1024u)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
                ok = ok && __float_as_uint(p[q][h].x) != LC_SENTINEL && __float_as_uint(p[q][h].y) != LC_SENTINEL &&
                     __float_as_uint(p[q][h].z) != LC_SENTINEL && __float_as_uint(p[q][h].w) != LC_SENTINEL;
              }
            if (__all(ok)) break;
            if (poll_fail(spins)) break;
          }
#pragma unroll
          for (int q = 0; q < 2; ++q)
            *reinterpret_cast<float4*>(red + (q * NWB + w) * 256 + lane * 4) =
                make_float4(p[q][0].x + p[q][1].x, p[q][0].y + p[q][1].y, p[q][0].z + p[q][1].z, p[q][0].w + p[q][1].w);
          __syncthreads();
          if (eok) {
#pragma unroll
            for (int k = 0; k < NWB; ++k) da += red[(eq * NWB + k) * 256 + ridx];
          }
        }
      }
      LCT(18);
#ifdef TNT_LC_TRACE
      if (tid == 0 && rb == 0 && i == LCT_STEP) lc_pub[5][j] = wall_clock64();
#endif
      // ---- the attention's query gradient of the step behind: dh_att_{i+1} = dq_{i+1} W2^T for this workgroup's 32 units, on
      // the MFMAs: wave 8 + t owns unit tile t; lane (kq, lr) polls the float4 dq[row lr][16 u + 4 kq ..] (element e of float4 u
      // feeds MFMA 4 u + e, whose contraction index is therefore a = 16 u + 4 kq + e) and reads the matching float4 of row
      // 16 t + lr of the W2 slice; rows 4 kq .. + 3 of unit 16 t + lr go to dha_l for the cell backward below
      if (w >= 8 && w < 10) {
        LC_TID(t4);
        const int l4 = t4 & 63, kq4 = l4 >> 4, lr4 = l4 & 15, tile = w - 8;
        const int qrow4 = rb * RB + lr4;
        const bool qm = lr4 < RB && qrow4 < B;
        float4 qa[NUA], qb[NUA];
#pragma unroll
        for (int u = 0; u < NUA; ++u) qb[u] = *reinterpret_cast<const float4*>(w2s + (tile * 16 + lr4) * LB_W2LD + 16 * u + 4 * kq4);
        unsigned spins = 0;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int u = 0; u < NUA; ++u) {
            qa[u] = (qm && 16 * u + 4 * kq4 < A) ? lc_ld4_l2(dq_rsrc, (unsigned)((qrow4 * 64 + 16 * u + 4 * kq4) * 4), (unsigned)(par * B * 256))
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            ok = ok && __float_as_uint(qa[u].x) != LC_SENTINEL && __float_as_uint(qa[u].y) != LC_SENTINEL &&
                 __float_as_uint(qa[u].z) != LC_SENTINEL && __float_as_uint(qa[u].w) != LC_SENTINEL;
          }
          if (__all(ok)) break;
          if (poll_fail(spins)) break;
        }
        floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NUA; ++u) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[u].x, qb[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[u].y, qb[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[u].z, qb[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[u].w, qb[u].w, acc, 0, 0, 0);
        }
        float* dw = dha_l + (kq4 * 4) * 36 + tile * 16 + lr4;
#pragma unroll
        for (int r = 0; r < 4; ++r) dw[r * 36] = acc[r];
      }
      LCT(19);
    }
    if (i == T - 1 && a.rate_out > 0.f) {                   // the first step of the chain has no push to sit behind
      LC_TID(tk);
      const int erowk = (tk & 255) >> 4, ebk = rb * RB + erowk;
      if (tk < 512 && erowk < RB && ebk < B)
        okeep = tnt_keep((uint64_t)(unsigned)(ebk * U + (2 * j + (tk >> 8)) * 16 + (tk & 15)), a.rate_out, a.att.seed,
                         a.site_out0 + (uint32_t)i, step_l) ? oscale : 0.f;
    }
    // ---- cell backward (the arithmetic of bwd_epilogue, lstm.hip)
#ifdef TNT_LC_TRACE
    if (tid == 512 && rb == 0 && i == LCT_STEP) lc_pub[0][j] = wall_clock64();
#endif
    __syncthreads();          // dh_att is in; every wave has read the dz tiles of the step behind (MFMA operands) before they are rewritten
    float4 dz_keep = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      LC_TID(t5);
      asm volatile("" : "+v"(g4.x), "+v"(g4.y), "+v"(g4.z), "+v"(g4.w), "+v"(cval), "+v"(cprev), "+v"(dout_t));
      const int erow5 = (t5 & 255) >> 4, ecol5 = t5 & 15, eq5 = t5 >> 8;
      if (t5 < 512 && erow5 < RB && rb * RB + erow5 < B) {
        if (i < T - 1) da += dha_l[erow5 * 36 + eq5 * 16 + ecol5];
        const float gi = g4.x, gf = g4.y, gg = g4.z, go = g4.w;
        const float tc = tnt_tanh(cval);
        const float dh = da + dout_t * okeep;
        const float dgo = dh * tc;
        const float dc = dc_c + dh * go * (1.f - tc * tc);
        const float4 dz4 = make_float4(dc * gg * gi * (1.f - gi), dc * cprev * gf * (1.f - gf), dc * gi * (1.f - gg * gg),
                                       dgo * go * (1.f - go));
        dc_c = dc * gf;
        dz_keep = dz4;
        *reinterpret_cast<float4*>(dzs + (eq5 * 16 + erow5) * LB_DZLD + ecol5 * 4) = dz4;
      }
    }
    __syncthreads();
    LCT(20);
    // ---- partial context gradient of this workgroup's 32 units: part[row][d] = sum_c dz[row][c] Wc[d][c]
    {
      LC_TID(t6);
      const int l6 = t6 & 63, kq6 = l6 >> 4, lr6 = l6 & 15;
      float pa[2];
#pragma unroll
      for (int sq = 0; sq < 2; ++sq) {
        const int kk = (2 * w + sq) * 4 + kq6;                 // 0..127: block kk / 64, column kk % 64 of its dz tile
        pa[sq] = dzs[((kk >> 6) * 16 + lr6) * LB_DZLD + (kk & 63)];
      }
      float (*pr)[4][16][17] = reinterpret_cast<float (*)[4][16][17]>(red);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < ntile) {
          floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[0], wcb_l[((w * 2 + 0) * 4 + t) * 64 + l6], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[1], wcb_l[((w * 2 + 1) * 4 + t) * 64 + l6], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) pr[w][t][kq6 * 4 + r][lr6] = acc[r];
        }
      }
      __syncthreads();
      LC_TID(t7);
      if (t7 < 16 * D) {
        const int pr7 = t7 / D, pd7 = t7 - pr7 * D;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NWB; ++k) t += pr[k][pd7 >> 4][pr7][pd7 & 15];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's reset of the next buffer's element is in L2 first
        lc_st1(pw_rsrc, t, (unsigned)((((j * 16 + pr7) * 64) + pd7) * 4), (unsigned)((pi * nrb + rb) * 65536));
      }
    }
    LCT(21);
#ifdef TNT_LC_TRACE
    if (tid == 0 && rb == 0 && i == LCT_STEP) lc_pub[1][j] = wall_clock64();
#endif
    if (T - 1 - i < 36) LCS(2 + T - 1 - i);
    // behind the publish (its drain is off the critical path): this step's dz for the weight-gradient GEMMs after the chain
    {
      LC_TID(t8);
      const int erow8 = (t8 & 255) >> 4, eb8 = rb * RB + erow8;
      if (t8 < 512 && erow8 < RB && eb8 < B)
        lc_st4(dz_rsrc, dz_keep, (unsigned)(eb8 * U + (2 * j + (t8 >> 8)) * 16 + (t8 & 15)) * 16u, (unsigned)i * BU4 * 4u);
    }
    LCT(25);
  }
#undef LC_TID
  if (ub == 16) LCT_DUMP(16, 32);
  LCS(38); LCS(39);
  if (ub == 16) LCS_DUMP(3);
  tnt_seq_leave(a.sync, xcc, a.guard_out);
}
}  // namespace

extern "C" int32_t tnt_lc_seq_bwd_work_floats(int32_t B, int32_t U) {
  if (B <= 0 || B > 128 || U != 512) return 0;
  const int nrb = B <= 64 ? (B + 7) / 8 : (B + 15) / 16;     // row blocks of 8 samples when the batch fits the 8 XCDs that way
  return 3 * nrb * 32 * 32 * 256 + 3 * B * U + 3 * nrb * 16 * 16 * 64;
}

extern "C" int32_t tnt_lc_seq_bwd_drop_f32(const float* F, const float* P, const float* W2, const float* v, const float* qpre,
                                      const float* alpha, const uint8_t* keep4, int64_t keep_stride, float* dP, float* dF,
                                      float* dvb, float* dqpre, const float* Ur, const float* Wc, const float* dout,
                                      const float* gates, const float* cs, float* dz, float* work, int32_t T, int32_t B,
                                      int32_t R, int32_t D, int32_t A, int32_t U, float slope, float rate_attn,
                                      float rate_in, int32_t in_lwidth, uint64_t seed, uint32_t site_attn0,
                                      uint32_t site_in0, const uint32_t* step_dev, float alpha_mse_coef, float rate_out,
                                      uint32_t site_out0, uint32_t* sync, float* guard_out, void* stream) {
  if (T <= 0 || sync == nullptr || work == nullptr || U != 512 || B <= 0 || B > 128) return TNT_BADARG(24);
  if (!(rate_out >= 0.f && rate_out < 1.f)) return TNT_BADARG(36);
  if (!wide_ok(R, D, A) || R > 512 || D > 64 || A > 64) return TNT_BADARG(21);
  if ((long)(T + 1) * B * U * 4 >= (1L << 32)) return TNT_BADARG(19);
  if (!tnt_aligned16(P) || !tnt_aligned16(F) || !tnt_aligned16(W2) || !tnt_aligned16(v) || !tnt_aligned16(dP) ||
      !tnt_aligned16(dF) || !tnt_aligned16(Wc) || !tnt_aligned16(Ur) || !tnt_aligned16(gates) || !tnt_aligned16(dz) ||
      !tnt_aligned16(work)) return TNT_BADARG(1);
  LcSeqBwdArgs a{};
  AttArgs& g = a.att;
  g.F = F; g.P = P; g.W2 = W2; g.v = v; g.qpre_in = qpre; g.alpha_in = alpha; g.keep4 = keep4; g.dP = dP; g.dF = dF; g.dvb = dvb;
  g.dqpre = dqpre; g.B = B; g.R = R; g.D = D; g.A = A; g.U = U; g.in_lwidth = in_lwidth; g.slope = slope;
  g.rate_attn = rate_attn; g.rate_in = rate_in; g.seed = seed; g.site_attn = site_attn0; g.site_in = site_in0; g.step = 0;
  g.step_dev = step_dev; g.alpha_mse = alpha_mse_coef;
  static const bool rb16 = getenv("TNT_SEQ_RB16") && atoi(getenv("TNT_SEQ_RB16")) != 0;               // A/B switch
  const bool rb8 = B <= 64 && !rb16;
  const int64_t nrb = rb8 ? (B + 7) / 8 : (B + 15) / 16;
  a.keep_stride = keep_stride; a.Ur = Ur; a.Wc = Wc; a.dout = dout; a.gates = gates; a.cs = cs; a.dz = dz;
  a.rate_out = rate_out; a.site_out0 = site_out0;
  a.xch = work; a.dhx = work + 3 * nrb * 32 * 32 * 256; a.parts = a.dhx + 3 * (int64_t)B * U;
  a.T = T; a.sync = sync; a.guard_out = guard_out;
  const int g4 = (A <= 32 && D <= 32) ? 8 : 16, np = (R + WT / g4 - 1) / (WT / g4);
  if (np > 6) return TNT_BADARG(21);
  void (*kern)(LcSeqBwdArgs) = nullptr;
  if (rb8) {
    if (g4 == 8) kern = np <= 3 ? lc_seq_bwd_kernel<8, 3, 8> : lc_seq_bwd_kernel<8, 6, 8>;
    else kern = np <= 3 ? lc_seq_bwd_kernel<16, 3, 8> : lc_seq_bwd_kernel<16, 6, 8>;
  } else {
    if (g4 == 8) kern = np <= 3 ? lc_seq_bwd_kernel<8, 3, 16> : lc_seq_bwd_kernel<8, 6, 16>;
    else kern = np <= 3 ? lc_seq_bwd_kernel<16, 3, 16> : lc_seq_bwd_kernel<16, 6, 16>;
  }
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LB_LDS_BYTES) != hipSuccess)
    return TNT_BADARG(90);
  hipLaunchKernelGGL(kern, dim3(256), dim3(1024), LB_LDS_BYTES, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_lc_seq_bwd_f32(const float* F, const float* P, const float* W2, const float* v, const float* qpre,
                                      const float* alpha, const uint8_t* keep4, int64_t keep_stride, float* dP, float* dF,
                                      float* dvb, float* dqpre, const float* Ur, const float* Wc, const float* dout,
                                      const float* gates, const float* cs, float* dz, float* work, int32_t T, int32_t B,
                                      int32_t R, int32_t D, int32_t A, int32_t U, float slope, float rate_attn,
                                      float rate_in, int32_t in_lwidth, uint64_t seed, uint32_t site_attn0,
                                      uint32_t site_in0, const uint32_t* step_dev, float alpha_mse_coef, uint32_t* sync,
                                      float* guard_out, void* stream) {
  return tnt_lc_seq_bwd_drop_f32(F, P, W2, v, qpre, alpha, keep4, keep_stride, dP, dF, dvb, dqpre, Ur, Wc, dout, gates, cs, dz,
                                 work, T, B, R, D, A, U, slope, rate_attn, rate_in, in_lwidth, seed, site_attn0, site_in0,
                                 step_dev, alpha_mse_coef, 0.f, 0, sync, guard_out, stream);
}
