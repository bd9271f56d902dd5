// Fused LSTM cell step (forward and backward) for the serial T-step decoder chain.
//
// Reference: keras LSTM layer 'lstm' called one timestep at a time
// (AttemptFour/Model/lc_NIC.py:118-124,255) or over a masked sequence
// (AttemptFour/Model/NIC.py:82-88,138-140); semantics in SURVEY.md 9.6.
//
// One launch per timestep does the recurrent matmul AND the gate math:
//   fwd:  z = xz + [h_prev, ctx] @ [Ur; Wc]  ->  i,f,g,o -> c,h     (64 x (U+D) x 4U)
//   bwd:  da = dz_next @ Ur^T (+ pass-through terms) -> cell backward -> dz
// The batch is skinny (B = 64), so there is no operand reuse to stage through LDS:
// every wave streams its own slice of the weights straight into VGPRs with 16-byte
// loads and feeds v_mfma_f32_16x16x4_f32.  Workgroup = 16 batch rows x 16 hidden units;
// its 8 waves split the contraction axis and combine through LDS.  The gate-interleaved
// weight layout [k][U][4] puts i,f,g,o of one unit in one 16-byte load and in one lane's
// four accumulators, so the gate math needs no cross-lane traffic.
#include "tnt_common.h"
#include "tnt_seq_sync.h"

namespace {

constexpr int NW = 8;  // waves per workgroup

__device__ __forceinline__ float4 ld4g(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

struct LstmFwdArgs {
  const float* xz; const float* h_prev; const float* c_prev; const float* Ur; const float* ctx; const float* Wc;
  const int* mask_ids; const float* out_prev; float* h; float* c; float* out; float* gates;
  int D, mask_T, mask_t, B, U;
  const float* zbias;      // nullable [U][4]: added to xz here, so the input projection can be an epilogue-free GEMM
};

// NWF waves split the contraction axis; each takes chunks of CK = 4*SS columns (SS MFMA k-steps of 4 per chunk)
template <int NWF, int SS>
__global__ __launch_bounds__(64 * NWF) void lstm_fwd_kernel(LstmFwdArgs a) {
  constexpr int CK = 4 * SS;
  __shared__ float red[NWF][4][16][17];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, lr = lane & 15;
  const int ub = blockIdx.x, rb = blockIdx.y;
  const int U = a.U, B = a.B;
  const int nchunk = (U + CK - 1) / CK;       // 64-wide chunks of the recurrent contraction, one per wave at U = 512
  const int arow = rb * 16 + lr;          // batch row this lane feeds as the A operand
  const int ucol = ub * 16 + lr;          // hidden unit this lane feeds as the B operand
  floatx4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (floatx4){0.f, 0.f, 0.f, 0.f};

  // epilogue operands do not depend on the matmul: fetch them now, under the weight stream
  const int erow = tid >> 4, ecol = tid & 15;
  const int eb = rb * 16 + erow, eu = ub * 16 + ecol;
  const bool eok = tid < 256 && eb < B;
  const long ee = (long)eb * U + eu;
  float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float cp = 0.f, hp = 0.f, op = 0.f;
  int mid = 1;
  if (eok) {
    x4 = *reinterpret_cast<const float4*>(a.xz + ee * 4);
    if (a.zbias) {
      const float4 b4 = *reinterpret_cast<const float4*>(a.zbias + (long)eu * 4);
      x4.x += b4.x; x4.y += b4.y; x4.z += b4.z; x4.w += b4.w;
    }
    cp = a.c_prev[ee]; hp = a.h_prev[ee];
    if (a.mask_ids) mid = a.mask_ids[eb * a.mask_T + a.mask_t];
    if (a.out_prev) op = a.out_prev[ee];
  }

  // context rows (the attention model's second operand, D extra contraction rows): spread over the waves in
  // MFMA k-steps of 4 -- wave w takes k = 4w .. 4w+3, 4w+32 .. -- instead of a ninth chunk that one wave would
  // have to run after its own (D = 32: one extra k-step per wave, +4 MFMAs, instead of +64 on wave 0)
  // (operands of the first such step are fetched here, ahead of the weight stream; the MFMAs come after it)
  float cav = 0.f;
  float4 cbv = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool has_ctx = a.ctx != nullptr && w * 4 < a.D;
  if (has_ctx) {
    const int kd = w * 4 + kq;
    const bool ok = kd < a.D;
    cav = (ok && arow < B) ? a.ctx[(long)arow * a.D + kd] : 0.f;
    cbv = ld4g(a.Wc + ((long)min(kd, a.D - 1) * U + ucol) * 4, ok);
  }
  for (int ci = w; ci < nchunk; ci += NWF) {
    const int kbase = ci * CK + kq * SS;
    float av[SS];
    float4 bv[SS];
    if (ci * CK + CK <= U) {
      // whole chunk inside the recurrent part: k = ci*64 + 16*(s>>2) + 4*kq + (s&3), so that one
      // load instruction reads 64 contiguous bytes per row (4 lanes x 16 B) instead of 16-B pieces
#pragma unroll
      for (int j = 0; j < SS / 4; ++j) {
        const float4 t = ld4g(a.h_prev + (long)arow * U + ci * CK + j * 16 + kq * 4, arow < B);
        av[4 * j + 0] = t.x; av[4 * j + 1] = t.y; av[4 * j + 2] = t.z; av[4 * j + 3] = t.w;
      }
#pragma unroll
      for (int s = 0; s < SS; ++s)
        bv[s] = ld4g(a.Ur + ((long)(ci * CK + (s >> 2) * 16 + kq * 4 + (s & 3)) * U + ucol) * 4, true);
    } else {
      // last, partial chunk (U % 64 != 0; U % 16 == 0, so a 16-run never straddles U): lanes past U feed zeros
      const bool kok = kbase < U;
#pragma unroll
      for (int j = 0; j < SS / 4; ++j) {
        const float4 t = ld4g(a.h_prev + (long)arow * U + kbase + 4 * j, kok && arow < B);
        av[4 * j + 0] = t.x; av[4 * j + 1] = t.y; av[4 * j + 2] = t.z; av[4 * j + 3] = t.w;
      }
#pragma unroll
      for (int s = 0; s < SS; ++s) bv[s] = ld4g(a.Ur + ((long)(kbase + s) * U + ucol) * 4, kok);
    }
#pragma unroll
    for (int s = 0; s < SS; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].w, acc[3], 0, 0, 0);
    }
  }
  if (has_ctx) {
    for (int st = w;;) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cav, cbv.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cav, cbv.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(cav, cbv.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(cav, cbv.w, acc[3], 0, 0, 0);
      st += NWF;
      if (st * 4 >= a.D) break;
      const int kd = st * 4 + kq;                      // D > 32: further k-steps of this wave
      const bool ok = kd < a.D;
      cav = (ok && arow < B) ? a.ctx[(long)arow * a.D + kd] : 0.f;
      cbv = ld4g(a.Wc + ((long)min(kd, a.D - 1) * U + ucol) * 4, ok);
    }
  }
  // C/D map of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][g][kq * 4 + r][lr] = acc[g][r];
  __syncthreads();
  if (eok) {
    const int row = erow, col = ecol;
    const long e = ee;
    float z[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NWF; ++k) s += red[k][g][row][col];
      z[g] += s;
    }
    const float gi = tnt_sigmoid_fast(z[0]), gf = tnt_sigmoid_fast(z[1]), gg = tnt_tanh(z[2]), go = tnt_sigmoid_fast(z[3]);
    const float c2 = gf * cp + gi * gg;
    const float h2 = go * tnt_tanh(c2);
    const bool m = mid != 0;
    a.h[e] = m ? h2 : hp;
    a.c[e] = m ? c2 : cp;
    if (a.out) a.out[e] = m ? h2 : op;
    *reinterpret_cast<float4*>(a.gates + e * 4) = make_float4(gi, gf, gg, go);
  }
}

struct LstmBwdArgs {
  const float* dz_next; const float* Ur; const float* da_pass_in; const float* dh_ext; const float* dc_in;
  const float* dout_in; const float* dout_t; const int* mask_ids; const float* gates; const float* c;
  const float* c_prev; float* dz; float* da_pass_out; float* dc_out; float* dout_out;
  int mask_T, mask_t, B, U;
  // optional (attention model): partial context gradient of this workgroup's 16 units,
  // dctx_part[ub][b][d] = sum_{u in block ub, g} dz[b][u][g] * Wc[d][u][g];  the attention backward sums the U/16 parts
  // instead of running the whole dz @ Wc^T product per sample on its own critical path.
  const float* Wc; float* dctx_part; int D;
};

// ---- shared pieces of the two backward variants
struct BwdEpi {
  float da0, dout, dcin, cval, cprev; float4 g4; int mid; bool eok; long e; int erow, ecol;
};
__device__ __forceinline__ BwdEpi bwd_prefetch(const LstmBwdArgs& a, int tid, int rb, int ub) {
  BwdEpi p;
  p.erow = tid >> 4; p.ecol = tid & 15;
  const int eb = rb * 16 + p.erow, eu = ub * 16 + p.ecol;
  p.eok = tid < 256 && eb < a.B;
  p.e = (long)eb * a.U + eu;
  p.da0 = 0.f; p.dout = 0.f; p.dcin = 0.f; p.cval = 0.f; p.cprev = 0.f; p.mid = 1;
  p.g4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.eok) {
    if (a.da_pass_in) p.da0 += a.da_pass_in[p.e];
    if (a.dh_ext) p.da0 += a.dh_ext[p.e];
    if (a.dout_in) p.dout += a.dout_in[p.e];
    if (a.dout_t) p.dout += a.dout_t[p.e];
    if (a.dc_in) p.dcin = a.dc_in[p.e];
    if (a.mask_ids) p.mid = a.mask_ids[eb * a.mask_T + a.mask_t];
    p.g4 = *reinterpret_cast<const float4*>(a.gates + p.e * 4);
    p.cval = a.c[p.e]; p.cprev = a.c_prev[p.e];
  }
  return p;
}
constexpr int CXLD = 68;      // row stride of the dz / Wc tiles of the context-gradient epilogue (68 % 32 == 4)

// context-gradient partial of one workgroup (see LstmBwdArgs): dzs[16][CXLD] holds this step's dz tile (16 rows x 64
// gate columns), wcs[D][CXLD] the matching 64 columns of Wc.  All 512 threads; D <= 64.
__device__ __forceinline__ void bwd_ctx_partial(const LstmBwdArgs& a, const float* dzs, const float* wcs, int ub, int rb) {
  for (int o = threadIdx.x; o < 16 * a.D; o += blockDim.x) {
    const int row = o / a.D, d = o % a.D;
    const float4* zr = reinterpret_cast<const float4*>(dzs + row * CXLD);
    const float4* wr = reinterpret_cast<const float4*>(wcs + d * CXLD);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float4 z = zr[j], w = wr[j];
      acc += z.x * w.x + z.y * w.y + z.z * w.z + z.w * w.w;
    }
    const int b = rb * 16 + row;
    if (b < a.B) a.dctx_part[((long)ub * a.B + b) * a.D + d] = acc;
  }
}
// the 64 Wc columns of unit block ub, D rows: D*16 float4, <= 2 per thread, fetched at kernel start
__device__ __forceinline__ void bwd_ctx_prefetch(const LstmBwdArgs& a, int ub, float4 wq[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = threadIdx.x + i * blockDim.x;
    wq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.dctx_part && f < a.D * 16) wq[i] = *reinterpret_cast<const float4*>(a.Wc + ((long)(f >> 4) * a.U + ub * 16) * 4 + (f & 15) * 4);
  }
}
__device__ __forceinline__ void bwd_ctx_stage(const LstmBwdArgs& a, float* wcs, const float4 wq[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = threadIdx.x + i * blockDim.x;
    if (f < a.D * 16) *reinterpret_cast<float4*>(wcs + (f >> 4) * CXLD + (f & 15) * 4) = wq[i];
  }
}

__device__ __forceinline__ void bwd_epilogue(const LstmBwdArgs& a, const BwdEpi& p, float da, float* dzs = nullptr) {
  if (!p.eok) return;
  const bool m = p.mid != 0;
  float4 dz4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float dc_o = p.dcin, da_o = da, dout_o = p.dout;
  if (m) {
    const float gi = p.g4.x, gf = p.g4.y, gg = p.g4.z, go = p.g4.w;
    const float tc = tnt_tanh(p.cval);
    const float dh = da + p.dout;
    const float dgo = dh * tc;
    const float dc = p.dcin + dh * go * (1.f - tc * tc);
    dz4.x = dc * gg * gi * (1.f - gi);
    dz4.y = dc * p.cprev * gf * (1.f - gf);
    dz4.z = dc * gi * (1.f - gg * gg);
    dz4.w = dgo * go * (1.f - go);
    dc_o = dc * gf; da_o = 0.f; dout_o = 0.f;
  }
  *reinterpret_cast<float4*>(a.dz + p.e * 4) = dz4;
  if (dzs) *reinterpret_cast<float4*>(dzs + p.erow * CXLD + p.ecol * 4) = dz4;
  if (a.dc_out) a.dc_out[p.e] = dc_o;
  if (a.da_pass_out) a.da_pass_out[p.e] = da_o;
  if (a.dout_out) a.dout_out[p.e] = dout_o;
}

// general variant: operands straight to registers (any U % 16 == 0; also the no-matmul first step)
__global__ __launch_bounds__(512) void lstm_bwd_kernel(LstmBwdArgs a) {
  __shared__ float red[NW][16][17];
  __shared__ __attribute__((aligned(16))) float cx[(16 + 64) * CXLD];      // dz tile + Wc tile of the context epilogue
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, lr = lane & 15;
  const int ub = blockIdx.x, rb = blockIdx.y;
  const int B = a.B, K = 4 * a.U;
  floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
  const BwdEpi ep = bwd_prefetch(a, tid, rb, ub);
  float4 wq[2];
  bwd_ctx_prefetch(a, ub, wq);
  if (a.dz_next) {
    const int arow = rb * 16 + lr, ucol = ub * 16 + lr;
    const int nchunk = K / 64;   // U % 16 == 0  =>  4U % 64 == 0
    // 4 chunks per pass: all 32 16-byte loads of a pass are in flight before the first MFMA
    for (int c0 = w; c0 < nchunk; c0 += NW * 4) {
      float4 ta[4][4], tb[4][4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int ci = c0 + cc * NW;
        const int kbase = ci * 64 + kq * 4;      // + 16*j: 64 contiguous bytes per row per instruction
        const bool cok = ci < nchunk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ta[cc][j] = ld4g(a.dz_next + (long)arow * K + kbase + 16 * j, cok && arow < B);
          tb[cc][j] = ld4g(a.Ur + (long)ucol * K + kbase + 16 * j, cok);
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
  if (a.dctx_part) {
    bwd_ctx_stage(a, cx + 16 * CXLD, wq);
    for (int e = tid; e < 16 * CXLD; e += 512) cx[e] = 0.f;          // rows past B stay zero
    __syncthreads();
  }
  if (ep.eok) {
    float da = ep.da0;
#pragma unroll
    for (int k = 0; k < NW; ++k) da += red[k][ep.erow][ep.ecol];
    bwd_epilogue(a, ep, da, a.dctx_part ? cx : nullptr);
  }
  if (a.dctx_part) {
    __syncthreads();
    bwd_ctx_partial(a, cx, cx + 16 * CXLD, ub, rb);
  }
}

// LDS-staged variant (4U % 1024 == 0): both operands are k-contiguous rows (dz_next[b][:], Ur[u][:]),
// so fragment-shaped register loads touch 16 rows x 64 B per instruction -- slow in the texture
// addresser.  Here all 512 threads stream whole rows (1 KiB contiguous per wave-instruction) into LDS,
// 1024 k at a time, and the waves read their MFMA operands back with conflict-free ds_read_b128
// (row stride KC+8 floats).  The loads of chunk i+1 are in flight during the MFMAs of chunk i.
constexpr int KC = 1024, KLD = KC + 8;
template <int NWB>
__global__ __launch_bounds__(64 * NWB) void lstm_bwd_lds_kernel(LstmBwdArgs a) {
  constexpr int NT = 64 * NWB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zs = smem;                       // [16][KLD]  dz_next rows of this row block
  float* Us = smem + 16 * KLD;            // [16][KLD]  Ur rows of this unit block
  float (*red)[16][17] = reinterpret_cast<float (*)[16][17]>(smem + 32 * KLD);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, lr = lane & 15;
  const int ub = blockIdx.x, rb = blockIdx.y;
  const int B = a.B, K = 4 * a.U;
  const int nchunk = K / KC;
  floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
  const BwdEpi ep = bwd_prefetch(a, tid, rb, ub);
  float4 wq[2];
  bwd_ctx_prefetch(a, ub, wq);
  constexpr int NLD = 16 * KC / 4 / NT;     // float4 per thread per operand per chunk
  float4 rz[NLD], ru[NLD];
  auto gload = [&](int c) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + i * NT;                  // 16 rows x KC/4 float4
      const int row = f / (KC / 4), c4 = (f % (KC / 4)) * 4;
      rz[i] = ld4g(a.dz_next + (long)(rb * 16 + row) * K + c * KC + c4, rb * 16 + row < B);
      ru[i] = ld4g(a.Ur + (long)(ub * 16 + row) * K + c * KC + c4, true);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + i * NT;
      const int row = f / (KC / 4), c4 = (f % (KC / 4)) * 4;
      *reinterpret_cast<float4*>(&Zs[row * KLD + c4]) = rz[i];
      *reinterpret_cast<float4*>(&Us[row * KLD + c4]) = ru[i];
    }
  };
  gload(0);
  sstore();
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    if (c + 1 < nchunk) gload(c + 1);
    const float* zr = Zs + lr * KLD + w * (KC / NWB) + 4 * kq;
    const float* ur = Us + lr * KLD + w * (KC / NWB) + 4 * kq;
#pragma unroll
    for (int j = 0; j < KC / NWB / 16; ++j) {
      const float4 x = *reinterpret_cast<const float4*>(zr + 16 * j);
      const float4 y = *reinterpret_cast<const float4*>(ur + 16 * j);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, y.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, y.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, y.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, y.w, acc, 0, 0, 0);
    }
    if (c + 1 < nchunk) {
      __syncthreads();
      sstore();
      __syncthreads();
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][kq * 4 + r][lr] = acc[r];
  __syncthreads();
  // the operand staging area is free now: reuse its head for the context-gradient tiles
  float* cx = smem;
  if (a.dctx_part) {
    for (int e = tid; e < 16 * CXLD; e += NT) cx[e] = 0.f;
    bwd_ctx_stage(a, cx + 16 * CXLD, wq);
    __syncthreads();
  }
  if (ep.eok) {
    float da = ep.da0;
#pragma unroll
    for (int k = 0; k < NWB; ++k) da += red[k][ep.erow][ep.ecol];
    bwd_epilogue(a, ep, da, a.dctx_part ? cx : nullptr);
  }
  if (a.dctx_part) {
    __syncthreads();
    bwd_ctx_partial(a, cx, cx + 16 * CXLD, ub, rb);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Persistent sequence forward (the dense-encoder NIC's masked LSTM over S steps, NIC.py:138-140): ONE launch runs
// all S dependent steps.  A dependent kernel launch costs ~3.1 us inside a captured graph; here a step pays one
// XCD-local barrier instead (~1.8 us with the h exchange, tools/probe/xcdbar_probe.hip), and the recurrent weights
// never leave the VGPRs.
//   * Work split as in lstm_fwd_kernel<16, 8>: workgroup = 16 batch rows x 16 hidden units, 16 waves, wave w owns
//     the 32 contraction rows k = 32w .. 32w+31; its weight fragments (8 x float4 per lane = the whole 128 KB slice
//     over the workgroup) are loaded ONCE.
//   * The 32 workgroups of one row block exchange h through global memory and must share an L2: a workgroup reads the
//     XCD it runs on (XCC_ID), takes a ticket in that XCD (= its unit block) and works on row block XCC_ID; 256
//     workgroups are launched (one per CU: 1024 threads + 64 KB of LDS), the workgroups of XCDs without a row block exit.
//     Hand-off per step (tnt_seq_sync.h): plain h stores, drained by every storing wave, one flag per workgroup; the
//     consumers read the h slab with L1-bypassing (sc1) loads only, so no L1 invalidate is needed.
//   * Epilogue operands that the same thread produced a step earlier (c, h, previous output) are carried in registers.
//   * Barriers spin a bounded number of times; on timeout, or on a ticket outside 0..31 (a launch that did not place
//     exactly 32 workgroups on the XCD), the error word is set and every wave leaves -- wrong results, never a hung
//     grid.  The last workgroup of a group to leave resets the group's counters and advances its epoch; barrier
//     targets are epoch * 64 + step.  tnt_lstm_seq_supported() checks the census once per process before a model opts in.
struct LstmSeqArgs {
  const float* xz; float* hs; float* cs; const float* Ur; const float* zbias;
  const int* mask_ids; float* out; float* gates;
  int S, B, U, mask_T, mask_s0;
  unsigned* sync;        // TNT_SEQ_SYNC_WORDS words, layout and protocol in tnt_seq_sync.h
  float* guard_out;      // nullable: set to the error code when the error word is seen set
};

// POLL: the hand-off of h between the steps needs no flag round at all.  hs[st+1] is written exactly once per launch, so
// the DATA is its own flag: every element of hs[st+1] is reset to a sentinel bit pattern (a NaN no arithmetic produces)
// before anyone can look for it, and a consuming wave simply re-issues its L1-bypassing loads of the 8 values it needs
// until none of them is the sentinel.  Per step that removes: the drain of the h stores, two workgroup barriers, the flag
// store, and the flag poll's L2 round trip -- and a wave waits only for the TWO workgroups that produce its 32 units
// instead of for all 32.  Reset protocol (per element, by the thread that owns it): hs[1] at kernel start, made visible by
// the launch's single flag barrier; hs[st+2] at the top of step st, completed (vmcnt(0)) before the same thread
// publishes hs[st+1].  A wave that polls hs[st+2] has read every element of hs[st+1], so every owner's reset of hs[st+2]
// is already in L2: stale values of the previous launch can never be taken for this launch's.  hs[st+2]'s old content has
// no reader in this launch.  A poll that exceeds the spin limit sets the error word (the step is then rejected by the
// guard, model_base.DeviceGuardError) and carries on with what it has, so no wave ever leaves the common control flow.
constexpr unsigned TNT_SEQ_SENTINEL = 0x7FC5EED5u;

#ifdef TNT_LC_TRACE
__device__ unsigned long long ls_trace[64];
extern "C" int32_t tnt_debug_ls_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ls_trace), sizeof(ls_trace)) == hipSuccess ? 0 : -1;
}
#define LST(k) do { if (tid == 0 && rb == 0 && ub == 0 && st == 5) lst_l[(k)] = wall_clock64(); } while (0)
#define LST_DECL __shared__ unsigned long long lst_l[16];
#define LST_DUMP do { if (tid == 0 && rb == 0 && ub == 0) for (int q_ = 0; q_ < 16; ++q_) ls_trace[q_] = lst_l[q_]; } while (0)
#else
#define LST(k) do {} while (0)
#define LST_DECL
#define LST_DUMP do {} while (0)
#endif
// RB: batch rows per row block (= per XCD).  16 fills the MFMA's M dimension; 8 spreads B <= 64 over all 8 XCDs (half of every
// A fragment is zero -- the MFMA count per wave is the same -- but a group exchanges, polls and reduces half as many rows).
template <bool POLL, int RB>
__global__ __launch_bounds__(1024) void lstm_seq_fwd_kernel(LstmSeqArgs a) {
  constexpr int NWF = 16, SS = 8, CK = 32;
  extern __shared__ __attribute__((aligned(16))) float seq_lds[];       // > 64 KB requested: one workgroup per CU
  float (*red)[4][16][17] = reinterpret_cast<float (*)[4][16][17]>(seq_lds);       // [NWF][4][16][17] = 69.6 KB
  unsigned* s_slot = reinterpret_cast<unsigned*>(seq_lds + NWF * 4 * 16 * 17);      // 2 words behind the reduction buffer
  LST_DECL
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, lr = lane & 15;
  const int U = a.U, B = a.B;
  const unsigned xcc = tnt_xcc_id();
  const int nrb = (B + RB - 1) / RB;
  if ((int)xcc >= nrb) return;                             // this XCD has no row block
  unsigned* bar = a.sync + xcc * 64;
  unsigned* err = a.sync + TNT_SEQ_ERR;
  const TntSeqSlot slot = tnt_seq_enter(a.sync, xcc, s_slot);
  if (slot.ub < 0) {                                       // error word set (tnt_seq_sync.h)
    if (tid == 0 && a.guard_out) a.guard_out[0] = 2.f;
    return;
  }
  const int ub = slot.ub, rb = __builtin_amdgcn_readfirstlane((int)xcc);
  const __amdgpu_buffer_rsrc_t hs_rsrc = tnt_rsrc(a.hs, (unsigned)((long)(a.S + 1) * B * U * 4));
  const int arow = (lr < RB) ? rb * RB + lr : B, ucol = ub * 16 + lr;      // rows past RB: zero fragments
  // ---- this wave's weight fragments, resident for the whole sequence
  float4 bv[SS];
  // RB == 8: the batched 4x4x1 MFMA (16 independent 4 x 4 blocks = 8 rows x 32 gate columns per instruction: no padded rows).
  // Lane l = 32 rg + 4 cg + j: row group rg, column group cg, column j.  One instruction handles ONE k: its A column (the 8
  // rows' h[k]) is taken from block `abid` of each half and broadcast to the half's 8 blocks (cbsz = 3), so a lane's float4 of
  // h -- rows rg*4 + j, k = 32 w + 4 cg + m -- serves the instructions (m, abid = cg); its B row comes from one half of the
  // wave (blgp 1 / 2), so every B register holds U[k][col] for an even k in lanes 0..31 and for the next k in lanes 32..63.
  const int x_rg = lane >> 5, x_cg = (lane >> 2) & 7, x_j = lane & 3;
  float bx[2][8][2];                       // [column half][abid = k quad][m pair]
  if (RB == 8) {
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int k = w * CK + 4 * q + 2 * pr + x_rg;
          bx[ch][q][pr] = a.Ur[((long)k * U + ub * 16) * 4 + ch * 32 + x_cg * 4 + x_j];
        }
  } else {
#pragma unroll
    for (int s = 0; s < SS; ++s) bv[s] = ld4g(a.Ur + ((long)(w * CK + (s >> 2) * 16 + kq * 4 + (s & 3)) * U + ucol) * 4, true);
  }
  // ---- epilogue thread state
  const int erow = tid >> 4, ecol = tid & 15;
  const int eb = rb * RB + erow, eu = ub * 16 + ecol;
  const bool eok = tid < 16 * RB && eb < B;
  const long ee = (long)eb * U + eu;
  const long BU = (long)B * U;
  float4 zb = make_float4(0.f, 0.f, 0.f, 0.f);
  float cp = 0.f, hp = 0.f, op = 0.f;
  float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (eok) {
    if (a.zbias) zb = *reinterpret_cast<const float4*>(a.zbias + (long)eu * 4);
    cp = a.cs[ee]; hp = a.hs[ee];
    x4 = *reinterpret_cast<const float4*>(a.xz + ee * 4);
  }
  const float sentinel = __uint_as_float(TNT_SEQ_SENTINEL);
  if (POLL) {
    if (eok) a.hs[BU + ee] = sentinel;
    tnt_seq_group_barrier(bar, ub, tnt_seq_target(slot.epoch, 1), err);
  }
  // The resident weights were loaded above and the barrier drained the memory counter, but the compiler does not know: it
  // keeps its own vmcnt(31) ... vmcnt(0) ladder in front of their first uses -- INSIDE the loop, where the final vmcnt(0) then
  // waits for whatever the step has in flight (the prefetch of the next step's operands).  Launder them once here.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (RB == 8) {
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) asm volatile("" : "+v"(bx[ch][q][pr]));
  }
  // The epilogue's operands of step st + 1 (the input projection xz, the mask id) are fetched during step st, BEHIND the poll
  // of h[st]: the memory counter is in-order, so a load from HBM issued in front of the poll -- as the round-2 kernel did,
  // right behind the publish -- holds the poll's own return back by its latency in exactly the waves that own epilogue
  // threads, and the workgroup's barrier waits for them (the compiler also put the compare of the mask id, hence a full
  // drain, at the step's top).
  int mid = 1;
  const long ees = eok ? ee : 0;
  const int ebs = eok ? eb : 0;
  const bool have_ids = a.mask_ids != nullptr;
  if (have_ids && a.mask_s0 <= 0) mid = a.mask_ids[ebs * a.mask_T - a.mask_s0];
  for (int st = 0; st < a.S; ++st) {
    if (POLL && eok && st + 2 <= a.S) a.hs[(long)(st + 2) * BU + ee] = sentinel;       // published by step st + 1
    LST(0);
    float4 x4n = make_float4(0.f, 0.f, 0.f, 0.f);
    int midn = 1;
    const bool pf_x = st + 1 < a.S, pf_m = have_ids && st + 1 < a.S && st + 1 >= a.mask_s0;      // uniform
    if (RB == 8) {
      // ---- this lane's float4 of h[st]: row rg*4 + j of the block, k = 32 w + 4 cg .. + 3 (sc1 loads, polled)
      const int xrow = rb * RB + x_rg * 4 + x_j;
      float4 am = make_float4(0.f, 0.f, 0.f, 0.f);
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
        if (xrow < B) {
          am = tnt_ld4_l2(hs_rsrc, (unsigned)(((long)st * BU + (long)xrow * U + w * CK + x_cg * 4) * 4));
          if (POLL)
            ok = __float_as_uint(am.x) != TNT_SEQ_SENTINEL && __float_as_uint(am.y) != TNT_SEQ_SENTINEL &&
                 __float_as_uint(am.z) != TNT_SEQ_SENTINEL && __float_as_uint(am.w) != TNT_SEQ_SENTINEL;
        }
        if (!POLL || __all(ok)) break;
        if (++spins > TNT_SEQ_SPIN_LIMIT) {
          if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
      LST(1);
      if (pf_x) x4n = *reinterpret_cast<const float4*>(a.xz + ((long)(st + 1) * BU + ees) * 4);
      if (pf_m) midn = a.mask_ids[ebs * a.mask_T + (st + 1 - a.mask_s0)];
      floatx4 xa[2];
      xa[0] = (floatx4){0.f, 0.f, 0.f, 0.f}; xa[1] = xa[0];
#define TNT_X4(q)                                                                                              \
      xa[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.x, bx[0][q][0], xa[0], 3, q, 1);                             \
      xa[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.x, bx[1][q][0], xa[1], 3, q, 1);                             \
      xa[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.y, bx[0][q][0], xa[0], 3, q, 2);                             \
      xa[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.y, bx[1][q][0], xa[1], 3, q, 2);                             \
      xa[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.z, bx[0][q][1], xa[0], 3, q, 1);                             \
      xa[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.z, bx[1][q][1], xa[1], 3, q, 1);                             \
      xa[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.w, bx[0][q][1], xa[0], 3, q, 2);                             \
      xa[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(am.w, bx[1][q][1], xa[1], 3, q, 2);
      TNT_X4(0) TNT_X4(1) TNT_X4(2) TNT_X4(3) TNT_X4(4) TNT_X4(5) TNT_X4(6) TNT_X4(7)
#undef TNT_X4
      LST(2);
      // partial z[row rg*4 + r][col ch*32 + cg*4 + j] of this wave's k chunk: red viewed as [NWF][8 rows][64 + 4]
      float* rx = &red[0][0][0][0] + w * (8 * 68);
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int r = 0; r < 4; ++r) rx[(x_rg * 4 + r) * 68 + ch * 32 + x_cg * 4 + x_j] = xa[ch][r];
      __syncthreads();
    } else {
    // ---- A fragments: this row block's h of the previous step (own XCD's L2)
      // (sc1 loads: the slab was stored by the other workgroups of this group)
      float av[SS];
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < SS / 4; ++j) {
          const float4 t = arow < B ? tnt_ld4_l2(hs_rsrc, (unsigned)(((long)st * BU + (long)arow * U + w * CK + j * 16 + kq * 4) * 4))
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
          av[4 * j + 0] = t.x; av[4 * j + 1] = t.y; av[4 * j + 2] = t.z; av[4 * j + 3] = t.w;
          if (POLL)
            ok = ok && __float_as_uint(t.x) != TNT_SEQ_SENTINEL && __float_as_uint(t.y) != TNT_SEQ_SENTINEL &&
                 __float_as_uint(t.z) != TNT_SEQ_SENTINEL && __float_as_uint(t.w) != TNT_SEQ_SENTINEL;
        }
        if (!POLL || __all(ok)) break;
        if (++spins > TNT_SEQ_SPIN_LIMIT) {        // never hang the grid: flag the error, go on with what there is
          if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
      if (pf_x) x4n = *reinterpret_cast<const float4*>(a.xz + ((long)(st + 1) * BU + ees) * 4);
      if (pf_m) midn = a.mask_ids[ebs * a.mask_T + (st + 1 - a.mask_s0)];
      floatx4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < SS; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].w, acc[3], 0, 0, 0);
      }
      LST(2);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][g][kq * 4 + r][lr] = acc[g][r];
      __syncthreads();
    }
    LST(3);
    // (the prefetched operands change hands HERE, in front of the step's stores: the copy waits for the prefetch -- issued
    // an MFMA phase ago -- and behind the stores it would also wait for their acknowledgements)
    const float4 xc = x4;
    const int midc = mid;
    x4 = x4n; mid = midn;
    if (eok) {
      float z[4] = {xc.x + zb.x, xc.y + zb.y, xc.z + zb.z, xc.w + zb.w};
      if (RB == 8) {
        const float* rx = &red[0][0][0][0] + erow * 68 + ecol * 4;
        float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NWF; ++k) {
          const float4 t = *reinterpret_cast<const float4*>(rx + k * (8 * 68));
          sacc.x += t.x; sacc.y += t.y; sacc.z += t.z; sacc.w += t.w;
        }
        z[0] += sacc.x; z[1] += sacc.y; z[2] += sacc.z; z[3] += sacc.w;
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float sacc = 0.f;
#pragma unroll
          for (int k = 0; k < NWF; ++k) sacc += red[k][g][erow][ecol];
          z[g] += sacc;
        }
      }
      const float gi = tnt_sigmoid_fast(z[0]), gf = tnt_sigmoid_fast(z[1]), gg = tnt_tanh(z[2]), go = tnt_sigmoid_fast(z[3]);
      const float c2 = gf * cp + gi * gg;
      const float h2 = go * tnt_tanh(c2);
      const bool m = midc != 0;
      const bool seq = st >= a.mask_s0;
      const float hn = m ? h2 : hp, cn = m ? c2 : cp;
      LST(4);
      if (POLL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's reset of hs[st+2] is in L2 first
      a.hs[(long)(st + 1) * BU + ee] = hn;
      a.cs[(long)(st + 1) * BU + ee] = cn;
      if (a.out && seq) { op = m ? h2 : op; a.out[(long)(st - a.mask_s0) * BU + ee] = op; }
      *reinterpret_cast<float4*>(a.gates + ((long)st * BU + ee) * 4) = make_float4(gi, gf, gg, go);
      hp = hn; cp = cn;
    }
    LST(5);
    if (st + 1 == a.S) break;
    if (POLL) {
      __syncthreads();        // `red` is rewritten by the next step's MFMA phase
    } else {
      // ---- XCD-local barrier (tnt_seq_sync.h): slices are in L2 once vmcnt drains, one flag word per workgroup
      tnt_seq_group_barrier(bar, ub, tnt_seq_target(slot.epoch, st + 1), err);
    }
  }
  LST_DUMP;
  tnt_seq_leave(a.sync, xcc, a.guard_out);
}


// ---------------------------------------------------------------------------------------------------------
// Persistent sequence backward (BPTT over the S dependent steps of the same sequence, tape.gradient through the keras
// LSTM of NIC.py:138-140 / lc_NIC.py:317-318): ONE launch, weights stationary, "push" formulation.
//   da[b][u] = sum_k dz_next[b][k] * Ur[u][k]   (k = (unit', gate), 4U of them)  is what chains the steps.
// The per-step kernel PULLS: workgroup (16 rows x 16 units u) reads its 16 rows of dz_next (128 KB) and its 16 rows of
// Ur (128 KB) every step.  Here a workgroup keeps the 64 gate columns k it PRODUCES (its own dz tile, 16 x 64, in LDS)
// and the matching Ur^T slice [64 k][512 u] (128 KB = 32 floats per lane over 1024 lanes, loaded ONCE), multiplies
// them into a partial da[16 x 512] (its k-slice of the sum) and pushes one 16 x 16 tile (1 KB) to each of the 32
// workgroups of its row block; after one XCD-local barrier every workgroup gathers the 32 partial tiles of its own
// 16 x 16 block (32 KB, fixed summation order: deterministic) and runs the cell backward, which yields its next dz
// tile.  Per step and workgroup: 32 KB out + 32 KB in through the XCD's L2 instead of 256 KB in.
// Same placement / synchronisation scheme as lstm_seq_fwd_kernel (row block = XCC_ID, tickets, epoch targets,
// tnt_seq_sync.h); the exchange buffer is double-buffered by step parity.
struct LstmSeqBwdArgs {
  const float* Ur; const float* dout_seq; const int* mask_ids; const float* gates; const float* cs;
  float* dz; float* xch; unsigned* sync; float* guard_out;
  int S, B, U, mask_T, mask_s0;
};

constexpr int SB_DZLD = 68;                                   // row stride of the dz tile in LDS (16-byte rows, 68 % 64 == 4)
constexpr int SB_LDS_BYTES = 131 * 1024;     // the loop uses 16*68 + 16*256 + 4 floats, the prologue stages the recurrent kernel's slab
                                             // (512 x 65 floats) here; > half of the CU's 160 KB requested: one workgroup per CU

// POLL: as in lstm_seq_fwd_kernel<POLL> the pushed tiles are their own flags.  The exchange area is a ring of THREE
// buffers; exchange t (step s = S - 2 - t) uses buffer t % 3.  The thread that writes a 16-byte chunk (dest, src = its
// workgroup, lane) owns that chunk in every buffer: it resets the chunk of buffer (t + 1) % 3 to the sentinel at the top
// of exchange t, drains (vmcnt(0)), and only then publishes its chunk of buffer t % 3 -- so a workgroup that polls
// buffer (t + 1) % 3 has already seen data that was stored after the reset of every chunk it will poll.  The old content
// of buffer (t + 1) % 3 (exchange t - 2) was consumed before any workgroup could publish exchange t - 1, which the
// resetting workgroup has fully gathered.  Buffer 0 is reset at kernel start behind the launch's only flag barrier.
// RB: batch rows per row block, as in lstm_seq_fwd_kernel.  With RB = 8 only the half of every partial tile that holds rows < 8
// (lanes 0..31 in the MFMA's C layout) is pushed and gathered.
template <bool POLL, int RB>
__global__ __launch_bounds__(1024) void lstm_seq_bwd_kernel(LstmSeqBwdArgs a) {
  constexpr int NWB = 16, NTW = 2;                            // U = 512 = 16 waves x 2 column tiles x 16 units
  extern __shared__ __attribute__((aligned(16))) float sb_lds[];
  float* dzs = sb_lds;                                        // [16][SB_DZLD]: this workgroup's dz tile of the previous step
  float* red = sb_lds + 16 * SB_DZLD;                         // [NWB][64][4]
  unsigned* s_slot = reinterpret_cast<unsigned*>(red + NWB * 256);
  LST_DECL
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, lr = lane & 15;
  const int U = a.U, B = a.B, S = a.S;
  const unsigned xcc = tnt_xcc_id();
  const int nrb = (B + RB - 1) / RB;
  const bool xl = kq * 4 < RB;                               // this lane's rows of a partial tile exist
  if ((int)xcc >= nrb) return;
  unsigned* bar = a.sync + xcc * 64;
  unsigned* err = a.sync + TNT_SEQ_ERR;
  const TntSeqSlot slot = tnt_seq_enter(a.sync, xcc, s_slot);
  if (slot.ub < 0) {
    if (tid == 0 && a.guard_out) a.guard_out[0] = 2.f;
    return;
  }
  const int ub = slot.ub, rb = __builtin_amdgcn_readfirstlane((int)xcc);
  const __amdgpu_buffer_rsrc_t x_rsrc = tnt_rsrc(a.xch, (unsigned)(3u * nrb * 32u * 32u * 1024u));
  // ---- resident weights: B operand of the MFMAs, Ur^T[k][n] = Ur[n][ub*64 + k]; lane (kq, lr) of column tile j holds
  // n = w*32 + j*16 + lr and the 16 contraction indices k = kq*16 + ks (a permutation of the MFMA's natural k order,
  // applied to both operands: each lane's slice is 64 contiguous bytes in HBM and in LDS)
  float bw[NTW][16];
  // RB == 8: the batched 4x4x1 MFMA as in lstm_seq_fwd_kernel (8 rows x 32 columns x ONE k per instruction, no padded rows).
  // Lane l = 32 rg + 4 cg + j owns column n = 32 w + 4 cg + j of the partial da and rows rg*4 .. +3; B registers hold
  // Ur^T[k][n] = Ur[n][64 ub + k] for an even k in lanes 0..31 and the next k in lanes 32..63 (blgp 1 / 2), A comes from
  // the block `abid` of each half (cbsz 3): a lane's two float4 of the dz tile -- row rg*4 + j, k = 32 kh + 4 cg + m.
  const int x_rg = lane >> 5, x_cg = (lane >> 2) & 7, x_j = lane & 3;
  float bx[2][8][2];                       // [k half][abid = k quad][m pair]
  if (RB == 8) {
    // A lane's 32 operands are 64 consecutive floats of ONE row of Ur (every other one), a different row per lane: read
    // straight from memory that is 32 load instructions of 32 cache lines each per wave, the CU's one address path serialising
    // 16 waves (the same pattern cost the attention model's backward chain 20 us of prologue).  Staged instead: the block's
    // [512 rows][64] slab comes in with coalesced 16-byte loads, is parked in the dynamic LDS block (free until the loop starts;
    // row stride 65: the 32 rows a wave reads side by side fall into 32 banks) and each lane picks its operands there.
    constexpr int ULD = 65;
    static_assert(512 * ULD * 4 + 16 <= SB_LDS_BYTES, "the staging slab must fit the dynamic LDS block");
    float4 st[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = tid + 1024 * k;                          // chunk c: row c / 16, floats 4 (c % 16) .. + 3
      st[k] = *reinterpret_cast<const float4*>(a.Ur + ((long)(c >> 4) * U + ub * 16) * 4 + (c & 15) * 4);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = tid + 1024 * k;
      float* d = sb_lds + (c >> 4) * ULD + (c & 15) * 4;
      d[0] = st[k].x; d[1] = st[k].y; d[2] = st[k].z; d[3] = st[k].w;
    }
    __syncthreads();
    const float* src = sb_lds + (w * 32 + x_cg * 4 + x_j) * ULD;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) bx[kh][q][pr] = src[kh * 32 + 4 * q + 2 * pr + x_rg];
    __syncthreads();                                         // ... before the block is carved up below
  } else {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const float* src = a.Ur + ((long)(w * 32 + j * 16 + lr) * U + ub * 16) * 4 + kq * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(src + 4 * q);
        bw[j][4 * q + 0] = t.x; bw[j][4 * q + 1] = t.y; bw[j][4 * q + 2] = t.z; bw[j][4 * q + 3] = t.w;
      }
    }
  }
  for (int e = tid; e < 16 * SB_DZLD; e += 1024) dzs[e] = 0.f;           // rows past B stay zero
  // ---- epilogue thread state (element (erow, ecol) of the 16 x 16 block), carried across the steps in registers
  const int erow = tid >> 4, ecol = tid & 15;
  const int eb = rb * RB + erow, eu = ub * 16 + ecol;
  const bool eok = tid < 16 * RB && eb < B;
  const long ee = (long)eb * U + eu;
  const long BU = (long)B * U;
  const int ridx = ((erow >> 2) * 16 + ecol) * 4 + (erow & 3);           // where the MFMA C layout keeps (erow, ecol)
  float da_c = 0.f, dc_c = 0.f, dout_c = 0.f;
  const float sentinel = __uint_as_float(TNT_SEQ_SENTINEL);
  const float4 sent4 = make_float4(sentinel, sentinel, sentinel, sentinel);
  // this lane's chunk of the tile for workgroup `dest` in ring buffer `buf`: slot [dest][src = ub], lane-major 1 KB tiles
  auto xslot = [&](int buf, int dest) { return a.xch + ((((long)(buf * nrb + rb) * 32 + dest) * 32 + ub) * 256) + lane * 4; };
  // RB == 8: this lane's float4 (rows rg*4 .. +3 of column 4 (cg & 3) + j) sits where the 16x16 C layout keeps those rows:
  // chunk rg*16 + 4 (cg & 3) + j of the tile for workgroup 2 w + (cg >> 2); one chunk per lane, owned in every ring buffer
  auto xslot8 = [&](int buf) {
    return a.xch + ((((long)(buf * nrb + rb) * 32 + (w * 2 + (x_cg >> 2))) * 32 + ub) * 256) + (x_rg * 16 + (x_cg & 3) * 4 + x_j) * 4;
  };
  __syncthreads();
  if (POLL) {
    if (RB == 8) *reinterpret_cast<float4*>(xslot8(0)) = sent4;
    else
#pragma unroll
      for (int j = 0; j < NTW; ++j) if (xl) *reinterpret_cast<float4*>(xslot(0, w * NTW + j)) = sent4;
    tnt_seq_group_barrier(bar, ub, tnt_seq_target(slot.epoch, 1), err);
  }
  for (int s = S - 1; s >= 0; --s) {
    const int st = S - 1 - s;          // trace step index (LST)
    LST(8);
    // epilogue operands of this step do not depend on the chain: fetch them first
    const bool seq = s >= a.mask_s0;
    // Epilogue operands of the step (gates, cell states, dout, the mask id: they come from HBM).  The loads are unconditional
    // (lanes without an epilogue element read element 0) and their values are laundered where the cell backward takes them up:
    // as the body of an `if`, or with a use up here (the compare of the mask id), the compiler waits for them at the step's top.
    // They are issued BEHIND the push (`epi_loads` below): the memory counter is in-order, so in front of it the drain that the
    // tile stores need would also wait for them, and the tiles of the waves that own epilogue threads would leave an HBM
    // latency late.  Behind the push the first poll of the gather waits for them instead, a wait that is there anyway.
    const long ees = eok ? ee : 0;
    float4 g4;
    float cval, cprev, dout_t = 0.f;
    int mid = 1;
    auto epi_loads = [&]() {
      g4 = *reinterpret_cast<const float4*>(a.gates + ((long)s * BU + ees) * 4);
      cval = a.cs[(long)(s + 1) * BU + ees]; cprev = a.cs[(long)s * BU + ees];
      if (seq && a.dout_seq) dout_t = a.dout_seq[(long)(s - a.mask_s0) * BU + ees];
      if (seq && a.mask_ids) mid = a.mask_ids[(eok ? eb : 0) * a.mask_T + (s - a.mask_s0)];
    };
    if (s == S - 1) epi_loads();                             // the first step of the chain has no push
    float da = da_c;
    if (s < S - 1) {
      const int xt = S - 2 - s, par = xt % 3;
      if (POLL && s > 0) {         // the next exchange's buffer: reset this thread's chunks before publishing this one's
        if (RB == 8) *reinterpret_cast<float4*>(xslot8((xt + 1) % 3)) = sent4;
        else
#pragma unroll
          for (int j = 0; j < NTW; ++j) if (xl) *reinterpret_cast<float4*>(xslot((xt + 1) % 3, w * NTW + j)) = sent4;
      }
      // ---- partial da = dz_tile[16 x 64] @ Ur^T slice[64 x 512]: this wave's 2 column tiles
      if (RB == 8) {
        float4 am[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
          am[kh] = *reinterpret_cast<const float4*>(dzs + (x_rg * 4 + x_j) * SB_DZLD + kh * 32 + x_cg * 4);
        floatx4 xa = (floatx4){0.f, 0.f, 0.f, 0.f};
#define TNT_X4(kh, q)                                                                         \
        xa = __builtin_amdgcn_mfma_f32_4x4x1f32(am[kh].x, bx[kh][q][0], xa, 3, q, 1);          \
        xa = __builtin_amdgcn_mfma_f32_4x4x1f32(am[kh].y, bx[kh][q][0], xa, 3, q, 2);          \
        xa = __builtin_amdgcn_mfma_f32_4x4x1f32(am[kh].z, bx[kh][q][1], xa, 3, q, 1);          \
        xa = __builtin_amdgcn_mfma_f32_4x4x1f32(am[kh].w, bx[kh][q][1], xa, 3, q, 2);
        TNT_X4(0, 0) TNT_X4(0, 1) TNT_X4(0, 2) TNT_X4(0, 3) TNT_X4(0, 4) TNT_X4(0, 5) TNT_X4(0, 6) TNT_X4(0, 7)
        TNT_X4(1, 0) TNT_X4(1, 1) TNT_X4(1, 2) TNT_X4(1, 3) TNT_X4(1, 4) TNT_X4(1, 5) TNT_X4(1, 6) TNT_X4(1, 7)
#undef TNT_X4
        if (POLL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's reset is in L2 first
        *reinterpret_cast<float4*>(xslot8(par)) = make_float4(xa[0], xa[1], xa[2], xa[3]);
      } else {
        float av[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(dzs + lr * SB_DZLD + kq * 16 + 4 * q);
          av[4 * q + 0] = t.x; av[4 * q + 1] = t.y; av[4 * q + 2] = t.z; av[4 * q + 3] = t.w;
        }
        floatx4 acc[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[j] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
          for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bw[j][ks], acc[j], 0, 0, 0);
        // ---- push: tile j belongs to workgroup (w*2 + j) of this row block
        if (POLL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's resets are in L2 first
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          if (xl) *reinterpret_cast<float4*>(xslot(par, w * NTW + j)) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
      }
      epi_loads();
      LST(9);
      if (!POLL) tnt_seq_group_barrier(bar, ub, tnt_seq_target(slot.epoch, S - 1 - s), err);
      // ---- gather the 32 partial tiles of this workgroup's block (sc1 loads: stored by other workgroups): wave w sums
      // sources w and w + 16, the 16 wave sums are combined through LDS in fixed order
      {
        const unsigned base = (unsigned)((((par * nrb + rb) * 32 + ub) * 32) * 1024) + (unsigned)lane * 16u;
        float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0;
        unsigned spins = 0;
        for (;;) {
          if (xl) {
            p0 = tnt_ld4_l2(x_rsrc, base + (unsigned)w * 1024u);
            p1 = tnt_ld4_l2(x_rsrc, base + (unsigned)(w + 16) * 1024u);
          }
          if (!POLL) break;
          const bool ok = __float_as_uint(p0.x) != TNT_SEQ_SENTINEL && __float_as_uint(p0.y) != TNT_SEQ_SENTINEL &&
                          __float_as_uint(p0.z) != TNT_SEQ_SENTINEL && __float_as_uint(p0.w) != TNT_SEQ_SENTINEL &&
                          __float_as_uint(p1.x) != TNT_SEQ_SENTINEL && __float_as_uint(p1.y) != TNT_SEQ_SENTINEL &&
                          __float_as_uint(p1.z) != TNT_SEQ_SENTINEL && __float_as_uint(p1.w) != TNT_SEQ_SENTINEL;
          if (__all(ok)) break;
          if (++spins > TNT_SEQ_SPIN_LIMIT) {      // never hang the grid: flag the error, go on with what there is
            if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
        LST(10);
        *reinterpret_cast<float4*>(red + w * 256 + lane * 4) = make_float4(p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w);
      }
      __syncthreads();
      LST(11);
      if (eok) {
#pragma unroll
        for (int k = 0; k < NWB; ++k) da += red[k * 256 + ridx];
      }
    }
    // ---- cell backward (same arithmetic as bwd_epilogue of the per-step kernel)
    asm volatile("" : "+v"(g4.x), "+v"(g4.y), "+v"(g4.z), "+v"(g4.w), "+v"(cval), "+v"(cprev), "+v"(dout_t), "+v"(mid));
    if (eok) {
      const bool m = mid != 0;
      const float dout = (seq ? dout_c : 0.f) + dout_t;
      float4 dz4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m) {
        const float gi = g4.x, gf = g4.y, gg = g4.z, go = g4.w;
        const float tc = tnt_tanh(cval);
        const float dh = da + dout;
        const float dgo = dh * tc;
        const float dc = dc_c + dh * go * (1.f - tc * tc);
        dz4.x = dc * gg * gi * (1.f - gi);
        dz4.y = dc * cprev * gf * (1.f - gf);
        dz4.z = dc * gi * (1.f - gg * gg);
        dz4.w = dgo * go * (1.f - go);
        dc_c = dc * gf; da_c = 0.f; dout_c = 0.f;
      } else {
        da_c = da; dout_c = dout;
      }
      *reinterpret_cast<float4*>(a.dz + ((long)s * BU + ee) * 4) = dz4;
      *reinterpret_cast<float4*>(dzs + erow * SB_DZLD + ecol * 4) = dz4;
    }
    LST(12);
    __syncthreads();          // the dz tile is complete before the next step's A fragments are read; `red` is free again
  }
  LST_DUMP;
  tnt_seq_leave(a.sync, xcc, a.guard_out);
}


// ---------------------------------------------------------------------------------------------------------
// LayerNormLSTMCell (tensorflow_addons.rnn.LayerNormLSTMCell, the use_layer_norm branch of lc_NIC.py:126-136):
//   z = LN_kernel(x W) + LN_recurrent(h U) + b;  c' = LN_state(sig(f) c + sig(i) tanh(g));  h' = sig(o) tanh(c')
// The two 4U-wide LayerNorms are tnt_layernorm_*_f32 launches around the matmuls; this pair of kernels is the cell
// itself: gate math + the state LayerNorm (a row reduction over U) forward, and its exact reverse.  One workgroup per
// batch row, the row lives in registers (U <= 4096).  Tensors gate-interleaved [B][U][4] like the plain cell.
struct LnCellArgs {
  const float* zk; const float* zr; const float* bias; const float* c_prev; const float* gs; const float* bs;
  float* gates; float* chat; float* istd; float* c; float* h;
  // backward
  const float* dh_a; const float* dh_b; const float* dh_c; const float* dcn_in; const float* c_in;
  float* dz; float* dc_prev; float* dcnt;
  int B, U; float eps;
};

constexpr int LN_MAXU_PER_THREAD = 16;

__device__ __forceinline__ float ln_block_sum(float v, float* sh) {
  v = tnt_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void ln_lstm_cell_fwd_kernel(LnCellArgs a) {
  __shared__ float sh[4];
  const int b = blockIdx.x, U = a.U;
  float craw[LN_MAXU_PER_THREAD], og[LN_MAXU_PER_THREAD];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < LN_MAXU_PER_THREAD; ++q) {
    const int u = threadIdx.x + q * 256;
    craw[q] = 0.f; og[q] = 0.f;
    if (u < U) {
      const long e = (long)b * U + u;
      const float4 k4 = *reinterpret_cast<const float4*>(a.zk + e * 4), r4 = *reinterpret_cast<const float4*>(a.zr + e * 4);
      const float4 b4 = *reinterpret_cast<const float4*>(a.bias + (long)u * 4);
      const float gi = tnt_sigmoid(k4.x + r4.x + b4.x), gf = tnt_sigmoid(k4.y + r4.y + b4.y);
      const float gg = tanhf(k4.z + r4.z + b4.z), go = tnt_sigmoid(k4.w + r4.w + b4.w);
      *reinterpret_cast<float4*>(a.gates + e * 4) = make_float4(gi, gf, gg, go);
      craw[q] = gf * a.c_prev[e] + gi * gg;
      og[q] = go;
      s += craw[q];
    }
  }
  const float mean = ln_block_sum(s, sh) / (float)U;
  float v = 0.f;
#pragma unroll
  for (int q = 0; q < LN_MAXU_PER_THREAD; ++q) {
    const int u = threadIdx.x + q * 256;
    if (u < U) { const float d = craw[q] - mean; v += d * d; }
  }
  const float inv = rsqrtf(ln_block_sum(v, sh) / (float)U + a.eps);
  if (threadIdx.x == 0) a.istd[b] = inv;
#pragma unroll
  for (int q = 0; q < LN_MAXU_PER_THREAD; ++q) {
    const int u = threadIdx.x + q * 256;
    if (u < U) {
      const long e = (long)b * U + u;
      const float xh = (craw[q] - mean) * inv;
      const float cn = xh * a.gs[u] + a.bs[u];
      a.chat[e] = xh;
      a.c[e] = cn;
      a.h[e] = og[q] * tanhf(cn);
    }
  }
}

__global__ __launch_bounds__(256) void ln_lstm_cell_bwd_kernel(LnCellArgs a) {
  __shared__ float sh[4];
  const int b = blockIdx.x, U = a.U;
  float dxh[LN_MAXU_PER_THREAD], xh[LN_MAXU_PER_THREAD], dgo[LN_MAXU_PER_THREAD];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < LN_MAXU_PER_THREAD; ++q) {
    const int u = threadIdx.x + q * 256;
    dxh[q] = 0.f; xh[q] = 0.f; dgo[q] = 0.f;
    if (u < U) {
      const long e = (long)b * U + u;
      float dh = 0.f;
      if (a.dh_a) dh += a.dh_a[e];
      if (a.dh_b) dh += a.dh_b[e];
      if (a.dh_c) dh += a.dh_c[e];
      const float go = a.gates[e * 4 + 3];
      const float tc = tanhf(a.c_in[e]);
      dgo[q] = dh * tc * go * (1.f - go);
      const float dcn = (a.dcn_in ? a.dcn_in[e] : 0.f) + dh * go * (1.f - tc * tc);
      a.dcnt[e] = dcn;
      xh[q] = a.chat[e];
      dxh[q] = dcn * a.gs[u];
      s1 += dxh[q];
      s2 += dxh[q] * xh[q];
    }
  }
  const float t1 = ln_block_sum(s1, sh), t2 = ln_block_sum(s2, sh);
  const float inv = a.istd[b], n = (float)U;
#pragma unroll
  for (int q = 0; q < LN_MAXU_PER_THREAD; ++q) {
    const int u = threadIdx.x + q * 256;
    if (u < U) {
      const long e = (long)b * U + u;
      const float dcr = inv / n * (n * dxh[q] - t1 - xh[q] * t2);
      const float4 g4 = *reinterpret_cast<const float4*>(a.gates + e * 4);
      const float gi = g4.x, gf = g4.y, gg = g4.z;
      *reinterpret_cast<float4*>(a.dz + e * 4) = make_float4(dcr * gg * gi * (1.f - gi), dcr * a.c_prev[e] * gf * (1.f - gf),
                                                             dcr * gi * (1.f - gg * gg), dgo[q]);
      a.dc_prev[e] = dcr * gf;
    }
  }
}

// census of a 256 x 1024-thread launch: how many workgroups land on each XCC_ID
__global__ __launch_bounds__(1024) void xcc_census_kernel(unsigned* hist) {
  extern __shared__ float seq_lds[];
  if (threadIdx.x == 0) { seq_lds[0] = 0.f; atomicAdd(&hist[tnt_xcc_id() & 15u], 1u); }
}

constexpr int SEQ_LDS_BYTES = 16 * 4 * 16 * 17 * 4 + 16;      // reduction buffer + slot words; > 64 KB, so one workgroup per CU

}  // namespace

extern "C" int32_t tnt_lstm_step_fwd_f32(const float* xz, const float* h_prev, const float* c_prev, const float* Ur,
                                         const float* ctx, const float* Wc, int32_t D, const int32_t* mask_ids,
                                         int32_t mask_T, int32_t mask_t, const float* out_prev, float* h, float* c,
                                         float* out, float* gates, int32_t B, int32_t U, const float* xz_bias,
                                         void* stream) {
  if (U <= 0 || U % 16 != 0) return TNT_BADARG(17);
  if (B <= 0) return TNT_BADARG(16);
  if (h == h_prev) return TNT_BADARG(12);
  LstmFwdArgs a;
  a.xz = xz; a.h_prev = h_prev; a.c_prev = c_prev; a.Ur = Ur; a.ctx = ctx; a.Wc = Wc; a.mask_ids = mask_ids;
  a.out_prev = out_prev; a.h = h; a.c = c; a.out = out; a.gates = gates;
  a.D = D; a.mask_T = mask_T; a.mask_t = mask_t; a.B = B; a.U = U; a.zbias = xz_bias;
  // 16 waves x 32-column chunks when the contraction has at least 16 such chunks (U >= 512): the per-wave MFMA chain and
  // load burst halve; measured 7.1 -> 6.3 us per step at U = 512 (0.738 -> 0.727 ms per training step)
  if (U % 32 == 0 && U / 32 >= 16)
    hipLaunchKernelGGL((lstm_fwd_kernel<16, 8>), dim3(U / 16, (B + 15) / 16), dim3(1024), 0, tnt_stream(stream), a);
  else
    hipLaunchKernelGGL((lstm_fwd_kernel<8, 16>), dim3(U / 16, (B + 15) / 16), dim3(512), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_lstm_step_bwd_f32(const float* dz_next, const float* Ur, const float* da_pass_in,
                                         const float* dh_ext, const float* dc_in, const float* dout_in,
                                         const float* dout_t, const int32_t* mask_ids, int32_t mask_T, int32_t mask_t,
                                         const float* gates, const float* c, const float* c_prev, float* dz,
                                         float* da_pass_out, float* dc_out, float* dout_out, int32_t B, int32_t U,
                                         const float* Wc, int32_t D, float* dctx_part, void* stream) {
  if (U <= 0 || U % 16 != 0) return TNT_BADARG(19);
  if (B <= 0) return TNT_BADARG(18);
  if (dz == dz_next) return TNT_BADARG(14);
  if (dctx_part && (Wc == nullptr || D <= 0 || D > 64 || !tnt_aligned16(Wc))) return TNT_BADARG(20);
  LstmBwdArgs a;
  a.dz_next = dz_next; a.Ur = Ur; a.da_pass_in = da_pass_in; a.dh_ext = dh_ext; a.dc_in = dc_in; a.dout_in = dout_in;
  a.dout_t = dout_t; a.mask_ids = mask_ids; a.gates = gates; a.c = c; a.c_prev = c_prev; a.dz = dz;
  a.da_pass_out = da_pass_out; a.dc_out = dc_out; a.dout_out = dout_out;
  a.mask_T = mask_T; a.mask_t = mask_t; a.B = B; a.U = U;
  a.Wc = Wc; a.dctx_part = dctx_part; a.D = dctx_part ? D : 0;
  if (dz_next != nullptr && (4 * U) % KC == 0) {
    // 8 waves: a 16-wave variant of this kernel measured no faster (config 2) or slower (config 3, with the context
    // epilogue) -- unlike the forward kernel
    const size_t smem = (size_t)(32 * KLD + 8 * 16 * 17) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_lds_kernel<8>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return -(int32_t)e;
      attr_set = true;
    }
    hipLaunchKernelGGL(lstm_bwd_lds_kernel<8>, dim3(U / 16, (B + 15) / 16), dim3(512), smem, tnt_stream(stream), a);
  } else {
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(U / 16, (B + 15) / 16), dim3(512), 0, tnt_stream(stream), a);
  }
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_lstm_seq_bwd_work_floats(int32_t B, int32_t U);

/* see include/tnt_hip.h */
extern "C" int32_t tnt_lstm_seq_supported(int32_t B, int32_t U) {
  static int cached = -1;
  if (U != 512 || B <= 0 || B > 128) return 0;
  if (cached < 0) {
    cached = 0;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (prop.multiProcessorCount != 256) return 0;
    if (hipFuncSetAttribute((const void*)lstm_seq_fwd_kernel<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, SEQ_LDS_BYTES) != hipSuccess) return 0;
    if (hipFuncSetAttribute((const void*)lstm_seq_fwd_kernel<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, SEQ_LDS_BYTES) != hipSuccess) return 0;
    if (hipFuncSetAttribute((const void*)lstm_seq_fwd_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SEQ_LDS_BYTES) != hipSuccess) return 0;
    if (hipFuncSetAttribute((const void*)xcc_census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SEQ_LDS_BYTES) != hipSuccess) return 0;
    unsigned* hist = nullptr;
    if (hipMalloc(&hist, 64) != hipSuccess) return 0;
    bool ok = true;
    for (int rep = 0; rep < 3 && ok; ++rep) {
      unsigned h[16];
      ok = hipMemset(hist, 0, 64) == hipSuccess;
      hipLaunchKernelGGL(xcc_census_kernel, dim3(256), dim3(1024), SEQ_LDS_BYTES, 0, hist);
      ok = ok && hipDeviceSynchronize() == hipSuccess && hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost) == hipSuccess;
      for (int i = 0; i < 8 && ok; ++i) ok = h[i] == 32u;
    }
    (void)hipFree(hist);
    cached = ok ? 1 : 0;
  }
  return cached;
}

extern "C" int32_t tnt_lstm_seq_fwd_f32(const float* xz, float* hs, float* cs, const float* Ur, const float* xz_bias,
                                        const int32_t* mask_ids, int32_t mask_T, int32_t mask_s0, float* out,
                                        float* gates, int32_t S, int32_t B, int32_t U, uint32_t* sync, float* guard_out,
                                        void* stream) {
  if (S <= 0 || S - 1 > TNT_SEQ_MAX_BARRIERS || sync == nullptr) return TNT_BADARG(11);
  if (!tnt_lstm_seq_supported(B, U)) return TNT_BADARG(13);
  if ((long)(S + 1) * B * U * 4 >= (1L << 32)) return TNT_BADARG(2);
  if (mask_s0 < 0 || mask_s0 > S || (mask_ids != nullptr && S - mask_s0 > mask_T)) return TNT_BADARG(7);
  hipStream_t s = tnt_stream(stream);
  LstmSeqArgs a;
  a.xz = xz; a.hs = hs; a.cs = cs; a.Ur = Ur; a.zbias = xz_bias; a.mask_ids = mask_ids; a.out = out; a.gates = gates;
  a.S = S; a.B = B; a.U = U; a.mask_T = mask_T; a.mask_s0 = mask_s0; a.sync = sync; a.guard_out = guard_out;
  static const bool flags_only = getenv("TNT_SEQ_FLAGS") && atoi(getenv("TNT_SEQ_FLAGS")) != 0;       // A/B switch
  static const bool rb8 = !(getenv("TNT_SEQ_RB16") && atoi(getenv("TNT_SEQ_RB16")) != 0);             // A/B switch: 16-row blocks
  if (flags_only) hipLaunchKernelGGL((lstm_seq_fwd_kernel<false, 16>), dim3(256), dim3(1024), SEQ_LDS_BYTES, s, a);
  else if (rb8 && B <= 64) hipLaunchKernelGGL((lstm_seq_fwd_kernel<true, 8>), dim3(256), dim3(1024), SEQ_LDS_BYTES, s, a);
  else hipLaunchKernelGGL((lstm_seq_fwd_kernel<true, 16>), dim3(256), dim3(1024), SEQ_LDS_BYTES, s, a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_lstm_seq_bwd_f32(const float* Ur, const float* dout_seq, const int32_t* mask_ids, int32_t mask_T,
                                        int32_t mask_s0, const float* gates, const float* cs, float* dz, float* work,
                                        int64_t work_floats, int32_t S, int32_t B, int32_t U, uint32_t* sync,
                                        float* guard_out, void* stream) {
  if (S <= 0 || S - 1 > TNT_SEQ_MAX_BARRIERS || sync == nullptr || work == nullptr) return TNT_BADARG(11);
  if (!tnt_lstm_seq_supported(B, U)) return TNT_BADARG(13);
  if (mask_s0 < 0 || mask_s0 > S || (mask_ids != nullptr && S - mask_s0 > mask_T)) return TNT_BADARG(5);
  if (work_floats < (int64_t)tnt_lstm_seq_bwd_work_floats(B, U)) return TNT_BADARG(10);
  if (!tnt_aligned16(work) || !tnt_aligned16(Ur) || !tnt_aligned16(gates) || !tnt_aligned16(dz)) return TNT_BADARG(1);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)lstm_seq_bwd_kernel<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)lstm_seq_bwd_kernel<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)lstm_seq_bwd_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS_BYTES) != hipSuccess)
      return TNT_BADARG(90);
    attr_set = true;
  }
  LstmSeqBwdArgs a;
  a.Ur = Ur; a.dout_seq = dout_seq; a.mask_ids = mask_ids; a.gates = gates; a.cs = cs; a.dz = dz; a.xch = work;
  a.sync = sync; a.guard_out = guard_out; a.S = S; a.B = B; a.U = U; a.mask_T = mask_T; a.mask_s0 = mask_s0;
  static const bool flags_only = getenv("TNT_SEQ_FLAGS") && atoi(getenv("TNT_SEQ_FLAGS")) != 0;       // A/B switch
  static const bool rb8 = !(getenv("TNT_SEQ_RB16") && atoi(getenv("TNT_SEQ_RB16")) != 0);             // A/B switch: 16-row blocks
  if (flags_only) hipLaunchKernelGGL((lstm_seq_bwd_kernel<false, 16>), dim3(256), dim3(1024), SB_LDS_BYTES, tnt_stream(stream), a);
  else if (rb8 && B <= 64) hipLaunchKernelGGL((lstm_seq_bwd_kernel<true, 8>), dim3(256), dim3(1024), SB_LDS_BYTES, tnt_stream(stream), a);
  else hipLaunchKernelGGL((lstm_seq_bwd_kernel<true, 16>), dim3(256), dim3(1024), SB_LDS_BYTES, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_lstm_seq_bwd_work_floats(int32_t B, int32_t U) {
  (void)U;
  // ring of three exchange buffers, one [32 dest][32 src] set of 1 KB tiles per row block; row blocks of 8 rows when the
  // batch fits the 8 XCDs that way
  const int nrb = B <= 64 ? (B + 7) / 8 : (B + 15) / 16;
  return 3 * nrb * 32 * 32 * 256;
}

extern "C" int32_t tnt_ln_lstm_cell_fwd_f32(const float* zk, const float* zr, const float* bias, const float* c_prev,
                                            const float* gamma_s, const float* beta_s, float* gates, float* chat,
                                            float* istd, float* c, float* h, int32_t B, int32_t U, float eps, void* stream) {
  if (B <= 0 || U <= 0 || U > 256 * LN_MAXU_PER_THREAD) return TNT_BADARG(12);
  if (!tnt_aligned16(zk) || !tnt_aligned16(zr) || !tnt_aligned16(bias) || !tnt_aligned16(gates)) return TNT_BADARG(1);
  LnCellArgs a{};
  a.zk = zk; a.zr = zr; a.bias = bias; a.c_prev = c_prev; a.gs = gamma_s; a.bs = beta_s; a.gates = gates; a.chat = chat;
  a.istd = istd; a.c = c; a.h = h; a.B = B; a.U = U; a.eps = eps;
  hipLaunchKernelGGL(ln_lstm_cell_fwd_kernel, dim3(B), dim3(256), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_ln_lstm_cell_bwd_f32(const float* dh_a, const float* dh_b, const float* dh_c, const float* dcn_in,
                                            const float* gates, const float* c_prev, const float* c, const float* chat,
                                            const float* istd, const float* gamma_s, float* dz, float* dc_prev,
                                            float* dcnt, int32_t B, int32_t U, void* stream) {
  if (B <= 0 || U <= 0 || U > 256 * LN_MAXU_PER_THREAD) return TNT_BADARG(14);
  if (!tnt_aligned16(gates) || !tnt_aligned16(dz)) return TNT_BADARG(1);
  if (dc_prev == dcn_in && dcn_in != nullptr) { /* in place is fine: every element is read before it is written by its own thread */ }
  LnCellArgs a{};
  a.dh_a = dh_a; a.dh_b = dh_b; a.dh_c = dh_c; a.dcn_in = dcn_in; a.gates = const_cast<float*>(gates); a.c_prev = c_prev;
  a.c_in = c; a.chat = const_cast<float*>(chat); a.istd = const_cast<float*>(istd); a.gs = gamma_s; a.dz = dz;
  a.dc_prev = dc_prev; a.dcnt = dcnt; a.B = B; a.U = U;
  hipLaunchKernelGGL(ln_lstm_cell_bwd_kernel, dim3(B), dim3(256), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}
