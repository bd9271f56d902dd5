// Per-variable clipnorm + Adam / SGD over one flat parameter arena (HBM-bound: 7 words
// moved per parameter per step).
// Reference: tf.keras.optimizers.Adam(1e-4, 0.9, 0.98, 1e-8, clipnorm=0.1) at
// AttemptFour/main.py:97 applied by optimizer.apply_gradients at lc_NIC.py:389 /
// NIC.py:250; SGD(momentum=0.9) at main.py:100-102; L2 regularisers lc_NIC.py:47-50.
// Semantics in SURVEY.md 9.9: each variable's gradient (data gradient + 2*lambda*theta)
// is scaled by c/max(||g||, c) on its own, then the dense Adam formulas are applied.
//
// The arena is cut into spans of <= SPAN elements, each inside one variable ("segment");
// the host builds the span table once.  Norms are reduced span -> segment in fixed order.
#include "tnt_common.h"
#include "tnt_fin.h"

namespace {

struct LrJob { const int64_t* adam_t; const float* lr; float* lr_t; float b1, b2; const float* ovr; };

__global__ __launch_bounds__(256) void span_sqnorm_kernel(const float* theta, const float* grad, SpanTab t,
                                                          float* partial, int nspan, LrJob lj) {
  __shared__ float s0[4], s1[4];
  const int sp = blockIdx.x;
  // the "lr job": Adam's step size for the update that follows this launch (the step counter advances at the end of that
  // update: tnt_adam_fin_f32), one thread of the launch, beside the norms
  if (lj.lr_t != nullptr && sp == 0 && threadIdx.x == 255) lj.lr_t[0] = tnt_adam_lr_t(lj.adam_t, lj.lr, lj.b1, lj.b2);
  if (sp >= nspan) return;
  tnt_span_norm(theta, grad, t, sp, partial, lj.ovr, s0, s1);
}

// one wave per segment: lanes stride over the segment's spans, then a fixed shuffle tree
__global__ __launch_bounds__(64) void seg_finalize_kernel(const float* partial, SpanTab t, float* sq, float* wsq,
                                                          int nseg) {
  const int s = blockIdx.x;
  if (s >= nseg) return;
  float a = 0.f, b = 0.f;
  for (int k = t.seg_first[s] + threadIdx.x; k < t.seg_first[s + 1]; k += 64) { a += partial[2 * k]; b += partial[2 * k + 1]; }
  a = tnt_wave_sum(a); b = tnt_wave_sum(b);
  if (threadIdx.x == 0) { sq[s] = a; wsq[s] = b; }
}

// L2 metric: sum_s lambda_s * ||theta_s||^2  (tf.add_n(self.losses), lc_NIC.py:379), fixed order
__global__ __launch_bounds__(256) void l2_total_kernel(const float* wsq, const float* seg_l2, int nseg, float* out) {
  __shared__ float sw[4];
  float a = 0.f;
  for (int s = threadIdx.x; s < nseg; s += 256) a += seg_l2[s] * wsq[s];
  a = tnt_wave_sum(a);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sw[0] + sw[1] + sw[2] + sw[3];
}

__device__ __forceinline__ float clip_scale(const float* sq, const float* sq_override, int seg, float clipnorm) {
  if (clipnorm <= 0.f) return 1.f;
  float q = sq[seg];
  if (sq_override && sq_override[seg] >= 0.f) q = sq_override[seg];
  return clipnorm / fmaxf(sqrtf(q), clipnorm);
}

struct MetRing { const float* met; float* ring; uint32_t* ring_t; int nmet, rows; };

__global__ __launch_bounds__(256) void adam_kernel(float* theta, float* m, float* v, const float* grad, SpanTab t,
                                                   const float* sq, const float* sq_override, int nspan, float lr_t,
                                                   const float* lr_t_dev, float b1, float b2, float eps,
                                                   float clipnorm, const uint32_t* guard, MetRing r) {
  const int sp = blockIdx.x;
  if (sp >= nspan) return;
  // Metrics ring (one wave of workgroup 0, off every critical path: the metrics are final before this launch starts): the
  // step's metrics vector met[0..nmet) goes to row (*ring_t % rows) of `ring` with the launch number in column nmet, and
  // *ring_t advances -- the host reads the row when it wants the numbers instead of cloning `met` behind every step.
  if (r.ring && sp == 0 && (int)threadIdx.x <= r.nmet) {
    const uint32_t rt = r.ring_t[0];                       // nmet + 1 <= 64 lanes: one wave, the load precedes the store
    float* row = r.ring + (long)(rt % (uint32_t)r.rows) * (r.nmet + 1);
    if ((int)threadIdx.x < r.nmet) row[threadIdx.x] = r.met[threadIdx.x];
    else { row[r.nmet] = (float)(rt & 0xFFFFFFu); r.ring_t[0] = rt + 1u; }
  }
  if (guard && guard[0] != 0u) return;       // the step's forward pass was invalid: leave the model untouched
  if (lr_t_dev) lr_t = lr_t_dev[0];
  const long off = t.span_off[sp];
  const int len = t.span_len[sp];
  const int seg = t.span_seg[sp];
  const float lam2 = 2.f * t.seg_l2[seg];
  const float cs = clip_scale(sq, sq_override, seg, clipnorm);
  const float ob1 = 1.f - b1, ob2 = 1.f - b2;
  const int len4 = len & ~3;
  for (int i = threadIdx.x * 4; i < len4; i += 1024) {
    float4 w = *reinterpret_cast<float4*>(theta + off + i);
    const float4 g4 = *reinterpret_cast<const float4*>(grad + off + i);
    float4 mm = *reinterpret_cast<float4*>(m + off + i);
    float4 vv = *reinterpret_cast<float4*>(v + off + i);
    float* wp = &w.x; float* mp = &mm.x; float* vp = &vv.x; const float* gp = &g4.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = (gp[j] + lam2 * wp[j]) * cs;
      mp[j] = mp[j] + (g - mp[j]) * ob1;
      vp[j] = vp[j] + (g * g - vp[j]) * ob2;
      wp[j] = wp[j] - lr_t * mp[j] / (sqrtf(vp[j]) + eps);
    }
    *reinterpret_cast<float4*>(theta + off + i) = w;
    *reinterpret_cast<float4*>(m + off + i) = mm;
    *reinterpret_cast<float4*>(v + off + i) = vv;
  }
  for (int i = len4 + threadIdx.x; i < len; i += 256) {
    const float w = theta[off + i];
    const float g = (grad[off + i] + lam2 * w) * cs;
    const float mm = m[off + i] + (g - m[off + i]) * ob1;
    const float vv = v[off + i] + (g * g - v[off + i]) * ob2;
    m[off + i] = mm; v[off + i] = vv;
    theta[off + i] = w - lr_t * mm / (sqrtf(vv) + eps);
  }
}

// ---- clip + Adam with the step's scalar tail INSIDE the launch (tnt_adam_fin_f32; single-process step).
// The finalize launch that used to sit in front of the update (12 us of a 500 us step for a few hundred scalars: a dependent
// launch whose every load is a cold round trip) is gone:
//  * every span workgroup sums ITS variable's clip norm from the span partials itself (tnt_seg_sums: a few coalesced loads)
//    (lr_t for the step was left by the norm launch in front of this one: tnt_span_sqnorm_lr_f32 / the Gram-norm launch);
//  * one extra workgroup (blockIdx 0) files what the host reads -- per-variable norms, L2 metric, loss / accuracy /
//    extra totals, the Embedding's sparse-norm total, ids -> prev_ids, the metrics ring -- beside the update, off every
//    critical path;
//  * the step counters advance when the LAST workgroup of the launch arrives (an atomic ticket), i.e. when nobody reads
//    them any more; the same workgroup re-arms the ticket.
struct FinArgs {
  const float* partial; const int32_t* seg_first; const float* seg_l2; float* sq; float* wsq; float* l2_out; int nseg;
  const float* x0; float* out0; const float* x1; float* out1; int n; float scale;
  const float* extra_part; float* extra; int n_extra; int extra_seg;
  const int32_t* ids_src; int32_t* ids_dst; int n_ids;
  const float* x2; float* out2; int n2; float scale2;
  int64_t* adam_t; uint32_t* drop_step; const float* lr; float* lr_t; float b1, b2; const uint32_t* guard;
  uint32_t* arrive;
};

__device__ __forceinline__ void fin_arrive_and_tick(const FinArgs& f, unsigned total) {
  // called by ONE thread of every workgroup of the launch, as its last action
  const unsigned old = __hip_atomic_fetch_add(f.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old != total - 1u) return;
  __hip_atomic_store(f.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (f.guard && f.guard[0] != 0u) return;
  if (f.drop_step) f.drop_step[0] += 1u;
  if (f.adam_t) f.adam_t[0] += 1;
}

constexpr int AF_U = 4;          // rounds of 1024 elements in flight per span workgroup (tnt_adam_fin_f32)

template <bool NT>
__global__ __launch_bounds__(256) void adam_fin_kernel(float* theta, float* m, float* v, const float* grad, SpanTab t,
                                                       const float* sq_override, int nspan, float eps, float clipnorm,
                                                       FinArgs f, MetRing r) {
  // workgroup 0 is the side workgroup: its work is a chain of dependent loads (~8 us end to end), so it is dispatched FIRST
  // and runs beside the whole update instead of hanging off its last round
  const int sp = (int)blockIdx.x - 1, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bool bad = f.guard && f.guard[0] != 0u;            // the step's forward pass was invalid: leave the model untouched
  if (sp >= 0) {
    if (!bad) {
      const long off = t.span_off[sp];
      const int len = t.span_len[sp];
      const int seg = t.span_seg[sp];
      const int len4 = len & ~3;
      // The span's operands are read AF_U rounds at a time (AF_U x 4 x 16 bytes per lane in flight): with one round per
      // trip the launch is bound by concurrency x latency, not by HBM (3.4 workgroups per CU x 4 KB per wave in flight
      // ~ 5 TB/s).  The first group is in flight BEFORE the clip norm's chain of dependent loads (span -> variable -> its
      // partials -> sum): every span workgroup pays that chain, and behind it the chain would add ~1 us to each.
      float4 wv[AF_U], g4[AF_U], mm[AF_U], vv[AF_U];
      auto load = [&](int base) {
#pragma unroll
        for (int u = 0; u < AF_U; ++u) {
          const int i = base + u * 1024;
          if (i < len4) {
            wv[u] = *reinterpret_cast<const float4*>(theta + off + i);
            g4[u] = tnt_ld4<NT>(grad + off + i);
            mm[u] = tnt_ld4<NT>(m + off + i);
            vv[u] = tnt_ld4<NT>(v + off + i);
          }
        }
      };
      load(tid * 4);
      const float lr_t = f.lr_t[0];                        // written by the norm launch in front of this one (its "lr job")
      const float lam2 = 2.f * f.seg_l2[seg];
      float cs = 1.f;
      if (clipnorm > 0.f) {
        float q;
        if (seg == f.extra_seg && f.n_extra > 0) {         // the Embedding: norm of the un-merged IndexedSlices rows
          q = 0.f;
          for (int i = lane; i < f.n_extra; i += 64) q += f.extra_part[i];
          q = tnt_wave_sum(q);
        } else if (sq_override && sq_override[seg] >= 0.f) {
          q = sq_override[seg];
        } else {
          q = tnt_seg_sums(f.partial, f.seg_first[seg], f.seg_first[seg + 1], lane).x;
        }
        cs = clipnorm / fmaxf(sqrtf(q), clipnorm);
      }
      const float ob1 = 1.f - f.b1, ob2 = 1.f - f.b2;
      for (int base = tid * 4; base < len4; base += 1024 * AF_U) {
        if (base != tid * 4) load(base);
#pragma unroll
        for (int u = 0; u < AF_U; ++u) {
          const int i = base + u * 1024;
          if (i < len4) {
            float* wp = &wv[u].x; float* mp = &mm[u].x; float* vp = &vv[u].x; const float* gp = &g4[u].x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float g = (gp[j] + lam2 * wp[j]) * cs;
              mp[j] = mp[j] + (g - mp[j]) * ob1;
              vp[j] = vp[j] + (g * g - vp[j]) * ob2;
              wp[j] = wp[j] - lr_t * mp[j] / (sqrtf(vp[j]) + eps);
            }
            *reinterpret_cast<float4*>(theta + off + i) = wv[u];
            tnt_st4<NT>(m + off + i, mm[u]);
            tnt_st4<NT>(v + off + i, vv[u]);
          }
        }
      }
      for (int i = len4 + tid; i < len; i += 256) {
        const float wv = theta[off + i];
        const float g = (grad[off + i] + lam2 * wv) * cs;
        const float mm = m[off + i] + (g - m[off + i]) * ob1;
        const float vv = v[off + i] + (g * g - v[off + i]) * ob2;
        m[off + i] = mm; v[off + i] = vv;
        theta[off + i] = wv - lr_t * mm / (sqrtf(vv) + eps);
      }
    }
  } else {
    // ---- the side workgroup: everything the host (and the next step) reads, beside the update
    __shared__ float sw[5][4];
    float l = 0.f, s0 = 0.f, s1 = 0.f, e = 0.f, s2 = 0.f;
    for (int s = tid; s < f.nseg; s += 256) {              // small variables: one per thread (serial span order)
      const int k0 = f.seg_first[s], k1 = f.seg_first[s + 1];
      if (k1 - k0 > 8) continue;
      float q = 0.f, ws = 0.f;
      for (int k = k0; k < k1; ++k) { q += f.partial[2 * k]; ws += f.partial[2 * k + 1]; }
      f.sq[s] = q; f.wsq[s] = ws;
      l += f.seg_l2[s] * ws;
    }
    for (int i = tid; i < f.n; i += 256) { s0 += f.x0[i]; if (f.x1) s1 += f.x1[i]; }
    for (int i = tid; i < f.n2; i += 256) s2 += f.x2[i];
    for (int i = tid; i < f.n_ids; i += 256) f.ids_dst[i] = f.ids_src[i];
    if (w == 0 && f.n_extra > 0) {                         // same order as the span workgroups use
      for (int i = lane; i < f.n_extra; i += 64) e += f.extra_part[i];
      e = tnt_wave_sum(e);
      if (lane == 0 && f.extra) f.extra[0] = e;
    }
    // large variables: one per wave (tnt_seg_sums order).  A wave looks at 64 variables at a time (one per lane) and then
    // walks the few that have more than 8 spans -- a per-variable loop costs a dependent load pair per variable, 700+ of them
    // in the region-wise model.
    for (int s0w = w * 64; s0w < f.nseg; s0w += 256) {
      const int sl = s0w + lane;
      int k0 = 0, k1 = 0;
      if (sl < f.nseg) { k0 = f.seg_first[sl]; k1 = f.seg_first[sl + 1]; }
      unsigned long long big = __ballot(k1 - k0 > 8);
      while (big) {
        const int j = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int s = s0w + j;
        const float2 qs = tnt_seg_sums(f.partial, __shfl(k0, j, 64), __shfl(k1, j, 64), lane);
        if (lane == 0) { f.sq[s] = qs.x; f.wsq[s] = qs.y; l += f.seg_l2[s] * qs.y; }
      }
    }
    l = tnt_wave_sum(l); s0 = tnt_wave_sum(s0); s1 = tnt_wave_sum(s1); s2 = tnt_wave_sum(s2);
    if (lane == 0) { sw[0][w] = l; sw[1][w] = s0; sw[2][w] = s1; sw[4][w] = s2; }
    __syncthreads();
    if (lane == 0 && w < 4) {
      const int j = w == 3 ? 4 : w;
      const float tt = (sw[j][0] + sw[j][1]) + (sw[j][2] + sw[j][3]);
      if (j == 0 && f.l2_out != nullptr) f.l2_out[0] = tt;
      if (j == 1 && f.n > 0) f.out0[0] = tt * f.scale;
      if (j == 2 && f.n > 0 && f.x1) f.out1[0] = tt * f.scale;
      if (j == 4 && f.n2 > 0) f.out2[0] = tt * f.scale2;
    }
    __syncthreads();                                       // the totals are written: file the metrics vector in the ring
    if (r.ring && tid <= r.nmet) {
      __threadfence_block();
      const uint32_t rt = r.ring_t[0];
      float* row = r.ring + (long)(rt % (uint32_t)r.rows) * (r.nmet + 1);
      if (tid < r.nmet) row[tid] = __hip_atomic_load(r.met + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else { row[r.nmet] = (float)(rt & 0xFFFFFFu); r.ring_t[0] = rt + 1u; }
    }
  }
  __syncthreads();
  if (tid == 0) fin_arrive_and_tick(f, (unsigned)nspan + 1u);
}

__global__ __launch_bounds__(256) void sgd_kernel(float* theta, float* mom, const float* grad, SpanTab t,
                                                  const float* sq, const float* sq_override, int nspan, float lr,
                                                  const float* lr_dev, float momentum, float clipnorm,
                                                  const uint32_t* guard) {
  const int sp = blockIdx.x;
  if (sp >= nspan) return;
  if (guard && guard[0] != 0u) return;
  if (lr_dev) lr = lr_dev[0];
  const long off = t.span_off[sp];
  const int len = t.span_len[sp];
  const int seg = t.span_seg[sp];
  const float lam2 = 2.f * t.seg_l2[seg];
  const float cs = clip_scale(sq, sq_override, seg, clipnorm);
  for (int i = threadIdx.x; i < len; i += 256) {
    const float w = theta[off + i];
    const float g = (grad[off + i] + lam2 * w) * cs;
    const float mv = momentum * mom[off + i] - lr * g;
    mom[off + i] = mv;
    theta[off + i] = w + mv;
  }
}

// Sharpness-aware minimisation helper (CaptionGenerator.train_step_SAM, ThinkAndTell/model.py:166-233):
// mode 0: e_w = (g + 2 lambda theta) * rho / (||g_total|| + 1e-12); theta += e_w; e_w is kept
// mode 1: theta -= e_w (restore).  ||g_total||^2 = sum of the per-segment squared norms sq[].
__global__ __launch_bounds__(256) void sam_kernel(float* theta, float* grad_rw, float* ew, SpanTab t, const float* sq,
                                                  const float* sq_override, int nseg, int nspan, float rho, int mode) {
  __shared__ float sw[4];
  const int sp = blockIdx.x;
  if (sp >= nspan) return;
  const long off = t.span_off[sp];
  const int len = t.span_len[sp];
  if (mode == 1) {
    // restore; the second gradient was taken at theta + e_w, L2 term included (2 lambda (theta + e_w)): the optimizer
    // adds 2 lambda theta at the RESTORED weights, so the difference 2 lambda e_w goes into the stored data gradient
    const float lam2r = 2.f * t.seg_l2[t.span_seg[sp]];
    for (int i = threadIdx.x; i < len; i += 256) {
      const float e = ew[off + i];
      theta[off + i] -= e;
      if (lam2r != 0.f) grad_rw[off + i] += lam2r * e;
    }
    return;
  }
  float a = 0.f;
  // tf.linalg.global_norm takes an IndexedSlices gradient by its un-deduplicated values: sq_override where given
  for (int s = threadIdx.x; s < nseg; s += 256) a += (sq_override && sq_override[s] >= 0.f) ? sq_override[s] : sq[s];
  a = tnt_wave_sum(a);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = a;
  __syncthreads();
  const float scale = rho / (sqrtf(sw[0] + sw[1] + sw[2] + sw[3]) + 1e-12f);
  const float lam2 = 2.f * t.seg_l2[t.span_seg[sp]];
  for (int i = threadIdx.x; i < len; i += 256) {
    const float w = theta[off + i];
    const float e = (grad_rw[off + i] + lam2 * w) * scale;
    ew[off + i] = e;
    theta[off + i] = w + e;
  }
}


// ---- adaptive gradient clipping (AttemptFour/Model/agc.py:20-38; call site lc_NIC.py:388), unit-wise:
// for a kernel of keras shape (in, out) every output column is its own unit; vectors are one unit.
//   p_norm = ||theta_col||, max_norm = max(p_norm, eps) * clip_factor, g_norm = ||g_col||
//   g_col <- g_col * max_norm / max(g_norm, 1e-6)   where g_norm >= max_norm
// g is the full gradient the tape sees (data gradient + 2 lambda theta); the arena keeps the data gradient, so the
// kernel writes back  g_clipped - 2 lambda theta  and the optimizer's own "+ 2 lambda theta" restores g_clipped.
// Work is cut into items = (variable, block of 64 columns, chunk of rows): pass 1 leaves per-item column partials,
// pass 2 sums the partials of its column block in fixed order (deterministic) and rescales its rows.
struct AgcTab {
  const int64_t* var_off; const int32_t* var_ld; const float* var_lam;     // per variable
  const int32_t* item;         // [nitem][6]: var, c0, ncols, r0, r1, colblock
  const int32_t* cb_first;     // [ncb + 1] first item of each column block
};

__global__ __launch_bounds__(256) void agc_partial_kernel(const float* theta, const float* grad, AgcTab t, float* partial) {
  __shared__ float sp[4][64], sg[4][64];
  const int it = blockIdx.x;
  const int32_t* d = t.item + it * 6;
  const int var = d[0], c0 = d[1], ncols = d[2], r0 = d[3], r1 = d[4];
  const long off = t.var_off[var];
  const int ld = t.var_ld[var];
  const float lam2 = 2.f * t.var_lam[var];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  float ps = 0.f, gs = 0.f;
  if (c < ncols)
    for (int r = r0 + rl; r < r1; r += 4) {
      const long e = off + (long)r * ld + c0 + c;
      const float w = theta[e], g = grad[e] + lam2 * w;
      ps += w * w; gs += g * g;
    }
  sp[rl][c] = ps; sg[rl][c] = gs;
  __syncthreads();
  if (threadIdx.x < 64) {
    partial[(long)it * 128 + c] = sp[0][c] + sp[1][c] + sp[2][c] + sp[3][c];
    partial[(long)it * 128 + 64 + c] = sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c];
  }
}

// gsq_cols (nullable) + gsq_var: column sums of squares that replace the dense gradient's for ONE variable -- the
// Embedding, whose gradient is an IndexedSlices in the reference: agc.py:25-30 takes the norm of its un-deduplicated
// rows.  sq_part_out (nullable): per column block of that variable, sum_c scale_c^2 * gsq_cols[c] -- the squared
// norm of the clipped un-deduplicated rows that clip-by-norm then uses.
__global__ __launch_bounds__(256) void agc_apply_kernel(const float* theta, float* grad, AgcTab t, const float* partial,
                                                        const float* gsq_cols, int gsq_var, float* sq_part_out,
                                                        int gsq_cb0, float clip_factor, float eps) {
  __shared__ float sscale[64];
  __shared__ float ssq[64];
  const int it = blockIdx.x;
  const int32_t* d = t.item + it * 6;
  const int var = d[0], c0 = d[1], ncols = d[2], r0 = d[3], r1 = d[4], cb = d[5];
  const long off = t.var_off[var];
  const int ld = t.var_ld[var];
  const float lam2 = 2.f * t.var_lam[var];
  if (threadIdx.x < 64) {
    const int c = threadIdx.x;
    float ps = 0.f, gs = 0.f;
    for (int k = t.cb_first[cb]; k < t.cb_first[cb + 1]; ++k) { ps += partial[(long)k * 128 + c]; gs += partial[(long)k * 128 + 64 + c]; }
    const bool ovr = gsq_cols != nullptr && var == gsq_var && c < ncols;
    if (ovr) gs = gsq_cols[c0 + c];
    const float max_norm = fmaxf(sqrtf(ps), eps) * clip_factor;
    const float gn = sqrtf(gs);
    const float sc = (c < ncols && !(gn < max_norm)) ? max_norm / fmaxf(gn, 1e-6f) : 1.f;
    sscale[c] = sc;
    ssq[c] = ovr ? sc * sc * gs : 0.f;
  }
  __syncthreads();
  if (sq_part_out != nullptr && var == gsq_var && r0 == 0 && threadIdx.x == 0) {
    float a = 0.f;
    for (int c = 0; c < 64; ++c) a += ssq[c];
    sq_part_out[cb - gsq_cb0] = a;
  }
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const float sc = sscale[c];
  if (c < ncols && sc != 1.f)
    for (int r = r0 + rl; r < r1; r += 4) {
      const long e = off + (long)r * ld + c0 + c;
      const float w = theta[e];
      grad[e] = (grad[e] + lam2 * w) * sc - lam2 * w;
    }
}

// column sums of squares of a row-major matrix (the Embedding's row gradients), one workgroup per 64 columns
__global__ __launch_bounds__(256) void colsq_kernel(const float* x, int rows, int cols, int ld, float* out) {
  __shared__ float s[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float a = 0.f;
  if (c < cols)
    for (int r = rl; r < rows; r += 4) { const float v = x[(long)r * ld + c]; a += v * v; }
  s[rl][threadIdx.x & 63] = a;
  __syncthreads();
  if (threadIdx.x < 64 && c < cols) out[c] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}

__global__ void agc_sq_total_kernel(const float* part, int n, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float a = 0.f;
  for (int i = 0; i < n; ++i) a += part[i];
  out[0] = a;
}

// advances the device-resident step state (so a captured hipGraph replays with fresh values):
//   adam_t += 1; lr_t = lr * sqrt(1-b2^t)/(1-b1^t); drop_step += 1
__global__ void step_tick_kernel(int64_t* adam_t, uint32_t* drop_step, const float* lr, float* lr_t, float b1,
                                 float b2, const uint32_t* guard) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (guard && guard[0] != 0u) return;
  if (drop_step) drop_step[0] += 1u;
  if (adam_t) {
    const int64_t t = adam_t[0] + 1;
    adam_t[0] = t;
    if (lr_t) {
      const double p1 = pow((double)b1, (double)t), p2 = pow((double)b2, (double)t);
      lr_t[0] = (float)((double)lr[0] * sqrt(1.0 - p2) / (1.0 - p1));
    }
  }
}


// ---- one launch for the scalar tail of a training step (each of these used to be its own dependent ~4.6 us launch):
//   (1) span partials -> per-variable squared norms sq[s] (clip-by-norm) and wsq[s] (L2 metric), fixed order;
//   (2) l2_out = sum_s lambda_s * wsq[s]                                   (tf.add_n(self.losses), lc_NIC.py:379);
//   (3) loss / accuracy totals of the step: out0 = scale * sum x0[0..n), out1 = scale * sum x1[0..n) (lc_NIC.py:370-376);
//   (4) extra[0] = sum of extra_part[0..n_extra): the Embedding's IndexedSlices squared norm from per-block partials;
//   (5) the device-resident step state advances (tnt_step_tick), unless the guard word is set.
// One workgroup of 1024 threads; every piece is optional (null pointers / zero counts).
struct FinalizeArgs {
  const float* partial; SpanTab t; float* sq; float* wsq; float* l2_out; int nseg;
  const float* x0; float* out0; const float* x1; float* out1; int n; float scale;
  const float* extra_part; float* extra; int n_extra;
  const int32_t* ids_src; int32_t* ids_dst; int n_ids;       // (4b) ids of this step -> prev_ids (tnt_embedding_bwd_sparse_f32)
  const float* x2; float* out2; int n2; float scale2;        // (4c) one more scaled total (the attention metric's T partials)
  int64_t* adam_t; uint32_t* drop_step; const float* lr; float* lr_t; float b1, b2; const uint32_t* guard;
};

__global__ __launch_bounds__(1024) void step_finalize_kernel(FinalizeArgs a) {
  // One workgroup, ONE barrier: every job's global loads are issued up front and its per-thread partial kept in a register
  // (the jobs used to run one after the other, each a cold load round trip + a barrier pair: 8 us back to back, 13 us inside
  // the captured step, where every input was just written by other CUs), then all wave partials meet in LDS once and four
  // different threads finish the four scalars while a fifth ticks the step counters.
  __shared__ float sw[5][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float l = 0.f, s0 = 0.f, s1 = 0.f, e = 0.f, s2 = 0.f;
  // (1) per-variable norms.  Small variables (<= 8 spans: every variable of the region-wise model, hundreds of them) one per
  // THREAD, serially over their spans; large ones one per wave, lanes striding over the spans + a fixed shuffle tree.  (A wave
  // per small variable would walk 700+ variables 16 at a time, each step a dependent load pair: ~30 us on the attention
  // model.)  (2) The L2 metric sum_s lambda_s ||theta_s||^2 is accumulated from the same sums.
  for (int s = tid; s < a.nseg; s += 1024) {
    const int k0 = a.t.seg_first[s], k1 = a.t.seg_first[s + 1];
    if (k1 - k0 > 8) continue;
    float q = 0.f, ws = 0.f;
    for (int k = k0; k < k1; ++k) { q += a.partial[2 * k]; ws += a.partial[2 * k + 1]; }
    a.sq[s] = q; a.wsq[s] = ws;
    l += a.t.seg_l2[s] * ws;
  }
  // (3) loss / accuracy totals, (4) extra partials (the Embedding's sparse norm), ids hand-over: loads in flight together
  for (int i = tid; i < a.n; i += 1024) { s0 += a.x0[i]; if (a.x1) s1 += a.x1[i]; }
  for (int i = tid; i < a.n_extra; i += 1024) e += a.extra_part[i];
  for (int i = tid; i < a.n2; i += 1024) s2 += a.x2[i];
  for (int i = tid; i < a.n_ids; i += 1024) a.ids_dst[i] = a.ids_src[i];
  for (int s = w; s < a.nseg; s += 16) {
    const int k0 = a.t.seg_first[s], k1 = a.t.seg_first[s + 1];
    if (k1 - k0 <= 8) continue;                 // wave-uniform
    float q = 0.f, ws = 0.f;
    for (int k = k0 + lane; k < k1; k += 64) { q += a.partial[2 * k]; ws += a.partial[2 * k + 1]; }
    q = tnt_wave_sum(q); ws = tnt_wave_sum(ws);
    if (lane == 0) { a.sq[s] = q; a.wsq[s] = ws; l += a.t.seg_l2[s] * ws; }
  }
  l = tnt_wave_sum(l); s0 = tnt_wave_sum(s0); s1 = tnt_wave_sum(s1); e = tnt_wave_sum(e); s2 = tnt_wave_sum(s2);
  if (lane == 0) { sw[0][w] = l; sw[1][w] = s0; sw[2][w] = s1; sw[3][w] = e; sw[4][w] = s2; }
  __syncthreads();
  if (lane == 0 && w < 5) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sw[w][k];
    if (w == 0 && a.l2_out != nullptr) a.l2_out[0] = t;
    if (w == 1 && a.n > 0) a.out0[0] = t * a.scale;
    if (w == 2 && a.n > 0 && a.x1) a.out1[0] = t * a.scale;
    if (w == 3 && a.n_extra > 0) a.extra[0] = t;
    if (w == 4 && a.n2 > 0) a.out2[0] = t * a.scale2;
  }
  // (5)
  if (tid == 256 && !(a.guard && a.guard[0] != 0u)) {
    if (a.drop_step) a.drop_step[0] += 1u;
    if (a.adam_t) {
      const int64_t t = a.adam_t[0] + 1;
      a.adam_t[0] = t;
      if (a.lr_t) {
        const double p1 = pow((double)a.b1, (double)t), p2 = pow((double)a.b2, (double)t);
        a.lr_t[0] = (float)((double)a.lr[0] * sqrt(1.0 - p2) / (1.0 - p1));
      }
    }
  }
}
}  // namespace

extern "C" int32_t tnt_step_tick(int64_t* adam_t, uint32_t* drop_step, const float* lr, float* lr_t, float beta1,
                                 float beta2, const uint32_t* guard, void* stream) {
  hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(64), 0, tnt_stream(stream), adam_t, drop_step, lr, lr_t, beta1,
                     beta2, guard);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_seg_sqnorm_f32(const float* theta, const float* grad, const int32_t* span_seg,
                                      const int64_t* span_off, const int32_t* span_len, const int32_t* seg_first,
                                      const float* seg_l2, float* partial, float* sq, float* wsq, float* l2_out,
                                      int32_t nspan, int32_t nseg, void* stream) {
  if (nspan <= 0 || nseg <= 0) return 0;
  SpanTab t{span_seg, span_off, span_len, seg_first, seg_l2};
  hipStream_t s = tnt_stream(stream);
  hipLaunchKernelGGL(span_sqnorm_kernel, dim3(nspan), dim3(256), 0, s, theta, grad, t, partial, nspan, LrJob{});
  TNT_LAUNCH_CHECK();
  hipLaunchKernelGGL(seg_finalize_kernel, dim3(nseg), dim3(64), 0, s, partial, t, sq, wsq, nseg);
  TNT_LAUNCH_CHECK();
  if (l2_out) {
    hipLaunchKernelGGL(l2_total_kernel, dim3(1), dim3(256), 0, s, wsq, seg_l2, nseg, l2_out);
    TNT_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int32_t tnt_l2_total_f32(const float* wsq, const float* seg_l2, int32_t nseg, float* out, void* stream) {
  if (nseg <= 0) return 0;
  hipLaunchKernelGGL(l2_total_kernel, dim3(1), dim3(256), 0, tnt_stream(stream), wsq, seg_l2, nseg, out);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_adam_ring_f32(float* theta, float* m, float* v, const float* grad, const int32_t* span_seg,
                                     const int64_t* span_off, const int32_t* span_len, const float* seg_l2, const float* sq,
                                     const float* sq_override, int32_t nspan, float lr_t, const float* lr_t_dev, float beta1,
                                     float beta2, float eps, float clipnorm, const uint32_t* guard, const float* met,
                                     int32_t nmet, float* ring, int32_t ring_rows, uint32_t* ring_t, void* stream) {
  if (nspan <= 0) return ring ? TNT_BADARG(11) : 0;
  if (ring != nullptr && (met == nullptr || ring_t == nullptr || nmet <= 0 || nmet > 62 || ring_rows <= 0)) return TNT_BADARG(20);
  SpanTab t{span_seg, span_off, span_len, nullptr, seg_l2};
  hipLaunchKernelGGL(adam_kernel, dim3(nspan), dim3(256), 0, tnt_stream(stream), theta, m, v, grad, t, sq, sq_override,
                     nspan, lr_t, lr_t_dev, beta1, beta2, eps, clipnorm, guard, MetRing{met, ring, ring_t, nmet, ring_rows});
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_adam_f32(float* theta, float* m, float* v, const float* grad, const int32_t* span_seg,
                                const int64_t* span_off, const int32_t* span_len, const float* seg_l2, const float* sq,
                                const float* sq_override, int32_t nspan, float lr_t, const float* lr_t_dev, float beta1,
                                float beta2, float eps, float clipnorm, const uint32_t* guard, void* stream) {
  return tnt_adam_ring_f32(theta, m, v, grad, span_seg, span_off, span_len, seg_l2, sq, sq_override, nspan, lr_t, lr_t_dev, beta1,
                           beta2, eps, clipnorm, guard, nullptr, 0, nullptr, 0, nullptr, stream);
}

extern "C" int32_t tnt_sgd_f32(float* theta, float* mom, const float* grad, const int32_t* span_seg,
                               const int64_t* span_off, const int32_t* span_len, const float* seg_l2, const float* sq,
                               const float* sq_override, int32_t nspan, float lr, const float* lr_dev, float momentum,
                               float clipnorm, const uint32_t* guard, void* stream) {
  if (nspan <= 0) return 0;
  SpanTab t{span_seg, span_off, span_len, nullptr, seg_l2};
  hipLaunchKernelGGL(sgd_kernel, dim3(nspan), dim3(256), 0, tnt_stream(stream), theta, mom, grad, t, sq, sq_override,
                     nspan, lr, lr_dev, momentum, clipnorm, guard);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_sam_f32(float* theta, float* grad, float* ew, const int32_t* span_seg,
                               const int64_t* span_off, const int32_t* span_len, const float* seg_l2, const float* sq,
                               const float* sq_override, int32_t nseg, int32_t nspan, float rho, int32_t mode,
                               void* stream) {
  if (nspan <= 0) return 0;
  SpanTab t{span_seg, span_off, span_len, nullptr, seg_l2};
  hipLaunchKernelGGL(sam_kernel, dim3(nspan), dim3(256), 0, tnt_stream(stream), theta, grad, ew, t, sq, sq_override, nseg, nspan,
                     rho, mode);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_agc_f32(const float* theta, float* grad, const int64_t* var_off, const int32_t* var_ld,
                               const float* var_lam, const int32_t* item, const int32_t* cb_first, int32_t nitem,
                               float* partial, const float* gsq_cols, int32_t gsq_var, int32_t gsq_cb0, int32_t gsq_ncb,
                               float* sq_part, float* sq_out, float clip_factor, float eps, void* stream) {
  if (nitem <= 0) return 0;
  if (gsq_cols != nullptr && (sq_part == nullptr || sq_out == nullptr || gsq_ncb <= 0)) return TNT_BADARG(12);
  AgcTab t{var_off, var_ld, var_lam, item, cb_first};
  hipStream_t s = tnt_stream(stream);
  hipLaunchKernelGGL(agc_partial_kernel, dim3(nitem), dim3(256), 0, s, theta, grad, t, partial);
  TNT_LAUNCH_CHECK();
  hipLaunchKernelGGL(agc_apply_kernel, dim3(nitem), dim3(256), 0, s, theta, grad, t, partial, gsq_cols, gsq_var,
                     gsq_cols ? sq_part : nullptr, gsq_cb0, clip_factor, eps);
  TNT_LAUNCH_CHECK();
  if (gsq_cols != nullptr) {
    hipLaunchKernelGGL(agc_sq_total_kernel, dim3(1), dim3(64), 0, s, sq_part, gsq_ncb, sq_out);
    TNT_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int32_t tnt_colsq_f32(const float* x, float* out, int32_t rows, int32_t cols, int32_t ld, void* stream) {
  if (rows <= 0 || cols <= 0) return TNT_BADARG(2);
  hipLaunchKernelGGL(colsq_kernel, dim3((cols + 63) / 64), dim3(256), 0, tnt_stream(stream), x, rows, cols, ld, out);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_span_sqnorm_f32(const float* theta, const float* grad, const int32_t* span_seg,
                                       const int64_t* span_off, const int32_t* span_len, const float* seg_l2,
                                       float* partial, int32_t nspan, void* stream) {
  if (nspan <= 0) return 0;
  SpanTab t{span_seg, span_off, span_len, nullptr, seg_l2};
  hipLaunchKernelGGL(span_sqnorm_kernel, dim3(nspan), dim3(256), 0, tnt_stream(stream), theta, grad, t, partial, nspan, LrJob{});
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_span_sqnorm_lr_f32(const float* theta, const float* grad, const int32_t* span_seg,
                                          const int64_t* span_off, const int32_t* span_len, const float* seg_l2,
                                          float* partial, int32_t nspan, const int64_t* adam_t, const float* lr, float* lr_t,
                                          float beta1, float beta2, const float* sq_override, void* stream) {
  if (nspan <= 0) return TNT_BADARG(8);
  if (adam_t == nullptr || lr == nullptr || lr_t == nullptr) return TNT_BADARG(9);
  SpanTab t{span_seg, span_off, span_len, nullptr, seg_l2};
  hipLaunchKernelGGL(span_sqnorm_kernel, dim3(nspan), dim3(256), 0, tnt_stream(stream), theta, grad, t, partial, nspan,
                     LrJob{adam_t, lr, lr_t, beta1, beta2, sq_override});
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_step_finalize_f32(const float* partial, const int32_t* seg_first, const float* seg_l2, float* sq,
                                         float* wsq, float* l2_out, int32_t nseg, const float* x0, float* out0,
                                         const float* x1, float* out1, int32_t n, float scale, const float* extra_part,
                                         float* extra, int32_t n_extra, const int32_t* ids_src, int32_t* ids_dst,
                                         int32_t n_ids, const float* x2, float* out2, int32_t n2, float scale2,
                                         int64_t* adam_t, uint32_t* drop_step, const float* lr,
                                         float* lr_t, float beta1, float beta2, const uint32_t* guard, void* stream) {
  if (nseg < 0 || n < 0 || n_extra < 0 || n2 < 0) return TNT_BADARG(7);
  if (n2 > 0 && (x2 == nullptr || out2 == nullptr)) return TNT_BADARG(20);
  if (n > 0 && (x0 == nullptr || out0 == nullptr)) return TNT_BADARG(8);
  FinalizeArgs a;
  a.partial = partial; a.t = SpanTab{nullptr, nullptr, nullptr, seg_first, seg_l2}; a.sq = sq; a.wsq = wsq; a.l2_out = l2_out;
  a.nseg = nseg; a.x0 = x0; a.out0 = out0; a.x1 = x1; a.out1 = out1; a.n = n; a.scale = scale;
  a.extra_part = extra_part; a.extra = extra; a.n_extra = n_extra;
  a.ids_src = ids_src; a.ids_dst = ids_dst; a.n_ids = (ids_src && ids_dst) ? n_ids : 0;
  a.x2 = x2; a.out2 = out2; a.n2 = n2; a.scale2 = scale2;
  a.adam_t = adam_t; a.drop_step = drop_step; a.lr = lr; a.lr_t = lr_t; a.b1 = beta1; a.b2 = beta2; a.guard = guard;
  hipLaunchKernelGGL(step_finalize_kernel, dim3(1), dim3(1024), 0, tnt_stream(stream), a);
  TNT_LAUNCH_CHECK();
  return 0;
}

namespace {
int32_t fin_fill(FinArgs& f, const tnt_finalize_desc* d) {
  if (d == nullptr || d->partial == nullptr || d->seg_first == nullptr || d->seg_l2 == nullptr || d->sq == nullptr ||
      d->wsq == nullptr || d->adam_t == nullptr || d->lr == nullptr || d->arrive == nullptr)
    return TNT_BADARG(1);
  if (d->nseg < 0 || d->n < 0 || d->n_extra < 0 || d->n2 < 0 || d->n_ids < 0) return TNT_BADARG(2);
  if (d->n > 0 && (d->x0 == nullptr || d->out0 == nullptr)) return TNT_BADARG(3);
  if (d->n2 > 0 && (d->x2 == nullptr || d->out2 == nullptr)) return TNT_BADARG(4);
  f.partial = d->partial; f.seg_first = d->seg_first; f.seg_l2 = d->seg_l2; f.sq = d->sq; f.wsq = d->wsq; f.l2_out = d->l2_out;
  f.nseg = d->nseg; f.x0 = d->x0; f.out0 = d->out0; f.x1 = d->x1; f.out1 = d->out1; f.n = d->n; f.scale = d->scale;
  f.extra_part = d->extra_part; f.extra = d->extra; f.n_extra = d->extra_part ? d->n_extra : 0; f.extra_seg = d->extra_seg;
  f.ids_src = d->ids_src; f.ids_dst = d->ids_dst; f.n_ids = (d->ids_src && d->ids_dst) ? d->n_ids : 0;
  f.x2 = d->x2; f.out2 = d->out2; f.n2 = d->n2; f.scale2 = d->scale2;
  f.adam_t = d->adam_t; f.drop_step = d->drop_step; f.lr = d->lr; f.lr_t = d->lr_t; f.b1 = d->beta1; f.b2 = d->beta2;
  f.guard = d->guard; f.arrive = d->arrive;
  return 0;
}
}  // namespace

extern "C" int32_t tnt_adam_fin_f32(float* theta, float* m, float* v, const float* grad, const int32_t* span_seg,
                                    const int64_t* span_off, const int32_t* span_len, const float* sq_override,
                                    int32_t nspan, float eps, float clipnorm, const tnt_finalize_desc* fin, const float* met,
                                    int32_t nmet, float* ring, int32_t ring_rows, uint32_t* ring_t, void* stream) {
  if (nspan < 0) return TNT_BADARG(9);
  if (ring != nullptr && (met == nullptr || ring_t == nullptr || nmet <= 0 || nmet > 62 || ring_rows <= 0)) return TNT_BADARG(13);
  FinArgs f;
  if (int32_t rc = fin_fill(f, fin)) return rc;
  SpanTab t{span_seg, span_off, span_len, nullptr, nullptr};
  if (tnt_stream_policy_nt())
    hipLaunchKernelGGL(adam_fin_kernel<true>, dim3(nspan + 1), dim3(256), 0, tnt_stream(stream), theta, m, v, grad, t, sq_override,
                       nspan, eps, clipnorm, f, MetRing{met, ring, ring_t, nmet, ring_rows});
  else
    hipLaunchKernelGGL(adam_fin_kernel<false>, dim3(nspan + 1), dim3(256), 0, tnt_stream(stream), theta, m, v, grad, t, sq_override,
                       nspan, eps, clipnorm, f, MetRing{met, ring, ring_t, nmet, ring_rows});
  TNT_LAUNCH_CHECK();
  return 0;
}
