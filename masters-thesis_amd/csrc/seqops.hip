// Embedding gather / deterministic scatter, softmax + categorical cross-entropy +
// accuracy (+ its gradient), argmax helpers.
// Reference ops: keras Embedding (lc_NIC.py:105-112,233; NIC.py:75-79,131), Softmax
// activation of the output Dense (lc_NIC.py:153; NIC.py:93), CategoricalCrossentropy
// (main.py:107-110 via lc_NIC.py:461-466), accuracy_calculation (lc_NIC.py:468-486),
// np.argmax in the greedy decoders (lc_NIC.py:627; NIC.py:192).
#include "tnt_common.h"
#include "tnt_rng.h"

namespace {

__global__ __launch_bounds__(256) void emb_fwd_kernel(const float* table, const int* ids, float* out, int B, int T,
                                                      int E, int ldo, int V) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // row = t*B + b
  if (row >= B * T) return;
  const int t = row / B, b = row % B;
  int id = ids[b * T + t];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  const float* src = table + (long)id * E;
  float* dst = out + (long)row * ldo;
  for (int j = lane; j < E; j += 64) dst[j] = src[j];
}

// gather + keras Dropout over the logical (B,T,E) tensor in one pass (NIC.py:131,140: Embedding -> the LSTM
// layer's input dropout): element index (b*T + t)*E + j, 4 columns per Philox call (E % 4 == 0).
__global__ __launch_bounds__(256) void emb_fwd_drop_kernel(const float* table, const int* ids, float* out, float* out_drop,
                                                           int B, int T, int E, int ldo, int V, float rate, uint64_t seed,
                                                           uint32_t site, uint32_t step, const uint32_t* step_dev,
                                                           float rate2, uint32_t site2, int lwidth2, int lcol0_2) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // row = t*B + b
  if (row >= B * T) return;
  if (step_dev) step += step_dev[0];
  const int t = row / B, b = row % B;
  int id = ids[b * T + t];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  const float* src = table + (long)id * E;
  const float scale = 1.0f / (1.0f - rate);
  for (int j = lane * 4; j < E; j += 256) {
    const float4 v = *reinterpret_cast<const float4*>(src + j);
    if (out) *reinterpret_cast<float4*>(out + (long)row * ldo + j) = v;
    bool k[4];
    tnt_keep4((uint64_t)(b * T + t) * E + j, rate, seed, site, step, k);
    float4 w = make_float4(k[0] ? v.x * scale : 0.f, k[1] ? v.y * scale : 0.f, k[2] ? v.z * scale : 0.f, k[3] ? v.w * scale : 0.f);
    if (rate2 > 0.f) {
      // second mask: the LSTM layer's per-call input dropout over its (B, lwidth2) input, of which these are columns
      // lcol0_2.. of the call of timestep t (site2 + t) -- lc_NIC.py:255 with dropout on the cell input
      const float sc2 = 1.0f / (1.0f - rate2);
      tnt_keep4((uint64_t)b * lwidth2 + lcol0_2 + j, rate2, seed, site2 + (uint32_t)t, step, k);
      w = make_float4(k[0] ? w.x * sc2 : 0.f, k[1] ? w.y * sc2 : 0.f, k[2] ? w.z * sc2 : 0.f, k[3] ? w.w * sc2 : 0.f);
    }
    *reinterpret_cast<float4*>(out_drop + (long)row * ldo + j) = w;
  }
}

__global__ __launch_bounds__(256) void rowsq_kernel(const float* x, float* rowsq, int rows, int cols, int ld) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int j = lane; j < cols; j += 64) { const float v = x[(long)row * ld + j]; s += v * v; }
  s = tnt_wave_sum(s);
  if (lane == 0) rowsq[row] = s;
}

// Deterministic scatter-add without atomics.  Block (k, jc): contribution row k (k = b*T + t),
// 64-column chunk jc.  The block whose row is the FIRST occurrence of its id owns that vocabulary
// row: its 4 waves sum a fixed strided subset of the later duplicates each (4 independent row loads
// (16 at a time) in flight per wave -- the pad/start tokens have hundreds of duplicates), and the 4 partials are
// combined in wave order through LDS.  Fixed partition + fixed order = bitwise reproducible.
// Rows of dtable that no token references are zeroed by zero_fill_kernel launched before this
// kernel (a captured hipMemsetAsync node did not replay reliably inside torch's hipGraph).
__global__ __launch_bounds__(256) void zero_fill_kernel(float* p, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0.f;
}

__global__ __launch_bounds__(256) void emb_bwd_kernel(const float* drows, const int* ids, float* dtable, int B, int T,
                                                      int E, int ldd, int V) {
  __shared__ float part[4][64];
  __shared__ int s_dup;
  const int k = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = B * T;
  int id = ids[k];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  if (threadIdx.x == 0) s_dup = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < k; i += 256) {           // an earlier occurrence owns the row
    int other = ids[i];
    other = other < 0 ? 0 : (other >= V ? V - 1 : other);
    if (other == id) s_dup = 1;
  }
  __syncthreads();
  if (s_dup) return;
  const int j = blockIdx.y * 64 + lane;
  const bool jok = j < E;
  float acc = 0.f;
  if (w == 0 && jok) acc = drows[(long)((k % T) * B + k / T) * ldd + j];
  // wave w scans positions k+1+64w, +256, ... ; hits processed 16 at a time (16 row loads in flight: the
  // pad / start tokens have hundreds of duplicates and their owner block is the kernel's critical path)
  constexpr int NF = 16;
  for (int base = k + 1 + 64 * w; base < n; base += 256) {
    const int i = base + lane;
    int other = i < n ? ids[i] : -1;
    other = other >= V ? V - 1 : other;
    unsigned long long hit = __ballot(i < n && other == id);
    while (hit) {
      int kk[NF];
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        kk[q] = -1;
        if (hit) { kk[q] = base + __ffsll((long long)hit) - 1; hit &= hit - 1; }
      }
      float v[NF];
#pragma unroll
      for (int q = 0; q < NF; ++q)
        v[q] = (kk[q] >= 0 && jok) ? drows[(long)((kk[q] % T) * B + kk[q] / T) * ldd + j] : 0.f;
#pragma unroll
      for (int q = 0; q < NF; ++q) acc += v[q];
    }
  }
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0 && jok) dtable[(long)id * E + j] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// float4 variant (E % 4 == 0, 16-byte aligned rows): a block covers 256 columns instead of 64, so the grid is 4x
// smaller (1920 blocks at 960 x 512: one round of the chip instead of four) and every row load moves 16 bytes per lane.
__global__ __launch_bounds__(256) void emb_bwd4_kernel(const float* drows, const int* ids, float* dtable, int B, int T,
                                                       int E, int ldd, int V) {
  __shared__ float4 part[4][64];
  __shared__ int s_dup;
  const int k = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = B * T;
  int id = ids[k];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  if (threadIdx.x == 0) s_dup = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < k; i += 256) {           // an earlier occurrence owns the row
    int other = ids[i];
    other = other < 0 ? 0 : (other >= V ? V - 1 : other);
    if (other == id) s_dup = 1;
  }
  __syncthreads();
  if (s_dup) return;
  const int j = blockIdx.y * 256 + lane * 4;
  const bool jok = j < E;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (w == 0 && jok) acc = *reinterpret_cast<const float4*>(drows + (long)((k % T) * B + k / T) * ldd + j);
  constexpr int NF = 8;
  for (int base = k + 1 + 64 * w; base < n; base += 256) {
    const int i = base + lane;
    int other = i < n ? ids[i] : -1;
    other = other >= V ? V - 1 : other;
    unsigned long long hit = __ballot(i < n && other == id);
    while (hit) {
      int kk[NF];
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        kk[q] = -1;
        if (hit) { kk[q] = base + __ffsll((long long)hit) - 1; hit &= hit - 1; }
      }
      float4 v[NF];
#pragma unroll
      for (int q = 0; q < NF; ++q)
        v[q] = (kk[q] >= 0 && jok) ? *reinterpret_cast<const float4*>(drows + (long)((kk[q] % T) * B + kk[q] / T) * ldd + j)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < NF; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
    }
  }
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0 && jok) {
    const float4 a = part[0][lane], b = part[1][lane], c = part[2][lane], d = part[3][lane];
    *reinterpret_cast<float4*>(dtable + (long)id * E + j) =
        make_float4(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y, ((a.z + b.z) + c.z) + d.z, ((a.w + b.w) + c.w) + d.w);
  }
}


// Embedding backward without the table-wide zero fill and without the separate row-norm launches (single-process step):
//   blocks [0, n)      scatter: as emb_bwd4_kernel (first occurrence of an id owns the row), plus the block's share of
//                      the IndexedSlices squared norm sum_{rows it reads} |row chunk|^2 -> sq_part[k * ny + y]
//                      (every (b,t) row is read by exactly one owner per column chunk: the parts add up to the norm
//                      of the un-deduplicated rows; non-owners write 0);
//   blocks [n, 2n)     cleanup: entry k of prev_ids (the ids of the PREVIOUS step, -1 = none) whose id does not occur
//                      in this step's ids gets its gradient row zeroed (first occurrence in prev_ids only).
// Rows touched by neither step are zero already (the table gradient starts zeroed and nothing else writes it), rows of
// this step are fully overwritten by their owner: no race between the two halves.  The caller copies ids -> prev_ids
// after this launch (tnt_step_finalize_f32 does).
// `rate` > 0: the rows are the gradient w.r.t. the DROPPED-OUT embedding output (the LSTM layer's input dropout of the
// text call, NIC.py:131,140): the keep mask of element k * E + j in stream (seed, site, step) -- the one
// tnt_embedding_fwd_drop_f32 applied -- is applied to every row chunk as it is read, instead of a dropout launch over
// the row buffer in front of this one.
struct EmbDrop { float rate, scale; uint64_t seed; uint32_t site; const uint32_t* step_dev; };

__device__ __forceinline__ float4 emb_row4(const float* drows, int k, int B, int T, int E, int ldd, int j, const EmbDrop& d,
                                           uint32_t step) {
  float4 v = *reinterpret_cast<const float4*>(drows + (long)((k % T) * B + k / T) * ldd + j);
  if (d.rate > 0.f) {
    bool kp[4];
    tnt_keep4((uint64_t)k * (uint64_t)E + (uint64_t)j, d.rate, d.seed, d.site, step, kp);
    v.x = kp[0] ? v.x * d.scale : 0.f; v.y = kp[1] ? v.y * d.scale : 0.f;
    v.z = kp[2] ? v.z * d.scale : 0.f; v.w = kp[3] ? v.w * d.scale : 0.f;
  }
  return v;
}

// EBS_W = waves per workgroup = ways the duplicate list of an id is split: 16 when the padding id's ~400 rows have to be
// summed, 4 when the caller declares them zero (zero_id) -- then the launch is bound by the dispatch of its 4 B T workgroups,
// and a quarter of the waves is a quarter of that
template <int EBS_W>
__global__ __launch_bounds__(64 * EBS_W) void emb_bwd_sparse_kernel(const float* drows, const int* ids, const int* prev_ids,
                                                                    float* dtable, float* sq_part, int B, int T, int E,
                                                                    int ldd, int V, EmbDrop drop, int zero_id) {
  // The critical path of this launch is the owner of the most frequent id (the padding id: ~400 of 960 rows in a caption
  // batch): its row sum is split 16 ways (4 ways cost 9 us for that one workgroup, 2 ways 17).
  constexpr int NT = 64 * EBS_W;
  __shared__ float4 part[EBS_W][64];
  __shared__ float sqw[EBS_W];
  __shared__ int s_flag;
  const int n = B * T, ny = gridDim.y;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.y * 256 + lane * 4;
  const bool jok = j < E;
  if (threadIdx.x == 0) s_flag = 0;
  __syncthreads();
  if ((int)blockIdx.x >= n) {
    // ---- cleanup of a row that only the previous step touched
    const int k = blockIdx.x - n;
    const int id = prev_ids[k];
    if (id < 0 || id >= V) return;
    for (int i = threadIdx.x; i < k; i += NT) if (prev_ids[i] == id) s_flag = 1;           // an earlier entry handles it
    for (int i = threadIdx.x; i < n; i += NT) {
      int cur = ids[i];
      cur = cur < 0 ? 0 : (cur >= V ? V - 1 : cur);
      if (cur == id) s_flag = 1;                                                            // rewritten by its owner
    }
    __syncthreads();
    if (s_flag) return;
    if (w == 0 && jok) *reinterpret_cast<float4*>(dtable + (long)id * E + j) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const int k = blockIdx.x;
  int id = ids[k];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  for (int i = threadIdx.x; i < k; i += NT) {            // an earlier occurrence owns the row
    int other = ids[i];
    other = other < 0 ? 0 : (other >= V ? V - 1 : other);
    if (other == id) s_flag = 1;
  }
  __syncthreads();
  if (s_flag) {
    if (threadIdx.x == 0) sq_part[(long)k * ny + blockIdx.y] = 0.f;
    return;
  }
  if (id == zero_id) {
    // the caller declares the rows of this id zero (mask_zero Embedding in front of a masked LSTM, NIC.py:131,140: a masked
    // step passes no gradient to its input): the owner files a zero row without reading the ~40 % of a caption batch that is
    // padding -- that one row sum was the critical path of the launch
    if (w == 0 && jok) *reinterpret_cast<float4*>(dtable + (long)id * E + j) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x == 0) sq_part[(long)k * ny + blockIdx.y] = 0.f;
    return;
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float sq = 0.f;
  const uint32_t dstep = drop.step_dev ? drop.step_dev[0] : 0u;
  if (w == 0 && jok) {
    acc = emb_row4(drows, k, B, T, E, ldd, j, drop, dstep);
    sq = acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
  }
  constexpr int NF = 8;
  for (int base = k + 1 + 64 * w; base < n; base += NT) {
    const int i = base + lane;
    int other = i < n ? ids[i] : -1;
    other = other >= V ? V - 1 : other;
    unsigned long long hit = __ballot(i < n && other == id);
    while (hit) {
      int kk[NF];
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        kk[q] = -1;
        if (hit) { kk[q] = base + __ffsll((long long)hit) - 1; hit &= hit - 1; }
      }
      float4 v[NF];
#pragma unroll
      for (int q = 0; q < NF; ++q)
        v[q] = (kk[q] >= 0 && jok) ? emb_row4(drows, kk[q], B, T, E, ldd, j, drop, dstep) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w;
        sq += v[q].x * v[q].x + v[q].y * v[q].y + v[q].z * v[q].z + v[q].w * v[q].w;
      }
    }
  }
  part[w][lane] = acc;
  sq = tnt_wave_sum(sq);
  if (lane == 0) sqw[w] = sq;
  __syncthreads();
  if (w == 0 && jok) {
    float4 t = part[0][lane];
#pragma unroll
    for (int q = 1; q < EBS_W; ++q) { const float4 u = part[q][lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *reinterpret_cast<float4*>(dtable + (long)id * E + j) = t;
  }
  if (threadIdx.x == 0) {
    float t = sqw[0];
#pragma unroll
    for (int q = 1; q < EBS_W; ++q) t += sqw[q];
    sq_part[(long)k * ny + blockIdx.y] = t;
  }
}

__global__ __launch_bounds__(1024) void sum_accum_kernel(const float* x, float* out, int n) {
  __shared__ float sw[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += x[i];
  s = tnt_wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sw[w];
    out[0] = t;
  }
}

struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax argmax_combine(ArgMax a, ArgMax b) {
  // larger value wins; ties -> smaller index (np.argmax / tf.argmax first-max rule)
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}
__device__ __forceinline__ ArgMax block_argmax(ArgMax a, ArgMax* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax b; b.v = __shfl_xor(a.v, o, 64); b.i = __shfl_xor(a.i, o, 64);
    a = argmax_combine(a, b);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = a;
  __syncthreads();
  ArgMax r = sh[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) r = argmax_combine(r, sh[k]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = tnt_wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) r += sh[k];
  __syncthreads();
  return r;
}

// one 256-thread workgroup per row
__global__ __launch_bounds__(256) void softmax_cce_kernel(const float* logits, const int* target, float* probs,
                                                          float* loss_row, float* correct_row, float* dlogits, int rows,
                                                          int V, int ld, float gscale, int from_logits, int mask_zero) {
  __shared__ ArgMax sha[4];
  __shared__ float shf[4];
  const int row = blockIdx.x;
  const float* x = logits + (long)row * ld;
  ArgMax am; am.v = -INFINITY; am.i = 0x7fffffff;
  for (int j = threadIdx.x; j < V; j += 256) {
    const float v = x[j];
    if (v > am.v) { am.v = v; am.i = j; }
  }
  am = block_argmax(am, sha);
  const float m = am.v;
  const int y = target ? target[row] : -1;
  // read the target logit before any thread may overwrite the row (outputs may alias logits)
  const float xy = (y >= 0 && y < V) ? x[y] : -INFINITY;
  float s = 0.f;
  for (int j = threadIdx.x; j < V; j += 256) s += expf(x[j] - m);
  const float Z = block_sum(s, shf);
  const float invZ = 1.f / Z;
  const float py = (y >= 0 && y < V) ? expf(xy - m) * invZ : 0.f;
  // keras from_logits=False: clip(p, 1e-7, 1-1e-7), zero gradient where the clip is active.
  // from_logits=True (SparseCategoricalCrossentropy, ThinkAndTell/train.py:262-263): lse - x_y, no clip.
  // mask_zero: rows whose target id is 0 contribute neither loss nor gradient (model.py:319-334).
  const bool live = !(mask_zero && y == 0);
  const bool active = live && (from_logits || ((py >= 1e-7f) && (py <= 1.f - 1e-7f)));
  if (threadIdx.x == 0 && target) {
    float l;
    if (from_logits) l = logf(Z) + m - xy;
    else l = -logf(fminf(fmaxf(py, 1e-7f), 1.f - 1e-7f));
    if (loss_row) loss_row[row] = live ? l : 0.f;
    if (correct_row) correct_row[row] = (am.i == y) ? 1.f : 0.f;
  }
  // logits may alias probs/dlogits: every thread reads its own elements before overwriting them
  for (int j = threadIdx.x; j < V; j += 256) {
    const float p = expf(x[j] - m) * invZ;
    if (dlogits) dlogits[(long)row * ld + j] = active ? (p - (j == y ? 1.f : 0.f)) * gscale : 0.f;
    if (probs && probs != dlogits) probs[(long)row * ld + j] = p;
  }
}

// Register-resident variant: the row (V <= 1024*NV4, ld % 4 == 0, 16-byte aligned rows) is read once with
// float4 loads, kept in VGPRs through max / exp / sum, and written once.  The generic kernel above is
// instruction-bound at V = 5001 (1750 instructions per wave: index-tracking argmax, two expf per element,
// per-element target select); here the inner work per element is mask, max, expf, add, mul -- the argmax
// index is recovered from an equality pass and the target element is patched by its owner afterwards.
// Pad columns [V, ld) are read (they are zero by the layout contract) and rewritten as zero.
template <int NV4>
__global__ __launch_bounds__(256) void softmax_cce_reg_kernel(const float* logits, const int* target, float* probs,
                                                              float* loss_row, float* correct_row, float* dlogits,
                                                              int rows, int V, int ld, float gscale, int from_logits,
                                                              int mask_zero) {
  __shared__ float shf[4];
  __shared__ int shi[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (long)row * ld;
  float4 v[NV4];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int j = 4 * (tid + 256 * i);
    v[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (j < ld) {
      v[i] = *reinterpret_cast<const float4*>(x + j);
      if (j + 1 >= V) v[i].y = -INFINITY;
      if (j + 2 >= V) v[i].z = -INFINITY;
      if (j + 3 >= V) v[i].w = -INFINITY;
      if (j >= V) v[i].x = -INFINITY;
    }
    m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
  }
  m = tnt_wave_max(m);
  if ((tid & 63) == 0) shf[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(shf[0], shf[1]), fmaxf(shf[2], shf[3]));
  // first index holding the maximum (np.argmax / tf.argmax rule)
  int am = 0x7fffffff;
#pragma unroll
  for (int i = NV4 - 1; i >= 0; --i) {
    const int j = 4 * (tid + 256 * i);
    if (v[i].w == m) am = j + 3;
    if (v[i].z == m) am = j + 2;
    if (v[i].y == m) am = j + 1;
    if (v[i].x == m) am = j;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) am = min(am, __shfl_xor(am, o, 64));
  if ((tid & 63) == 0) shi[tid >> 6] = am;
  const int y = target ? target[row] : -1;
  const float xy = (y >= 0 && y < V) ? x[y] : -INFINITY;     // before any thread overwrites the row (aliasing)
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    v[i].x = expf(v[i].x - m); v[i].y = expf(v[i].y - m); v[i].z = expf(v[i].z - m); v[i].w = expf(v[i].w - m);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  s = tnt_wave_sum(s);
  __syncthreads();                       // everyone has read shf (max) before it is reused for the sum
  if ((tid & 63) == 0) shf[tid >> 6] = s;
  __syncthreads();
  const float Z = (shf[0] + shf[1]) + (shf[2] + shf[3]);
  am = min(min(shi[0], shi[1]), min(shi[2], shi[3]));
  const float invZ = 1.f / Z;
  const float py = (y >= 0 && y < V) ? expf(xy - m) * invZ : 0.f;
  const bool live = !(mask_zero && y == 0);
  const bool active = live && (from_logits || ((py >= 1e-7f) && (py <= 1.f - 1e-7f)));
  if (tid == 0 && target) {
    float l;
    if (from_logits) l = logf(Z) + m - xy;
    else l = -logf(fminf(fmaxf(py, 1e-7f), 1.f - 1e-7f));
    if (loss_row) loss_row[row] = live ? l : 0.f;
    if (correct_row) correct_row[row] = (am == y) ? 1.f : 0.f;
  }
  const float gs = active ? gscale : 0.f;
  float* drow = dlogits ? dlogits + (long)row * ld : nullptr;
  float* prow = (probs && probs != dlogits) ? probs + (long)row * ld : nullptr;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int j = 4 * (tid + 256 * i);
    if (j >= ld) continue;
    const float4 p = make_float4(v[i].x * invZ, v[i].y * invZ, v[i].z * invZ, v[i].w * invZ);
    if (drow) *reinterpret_cast<float4*>(drow + j) = make_float4(p.x * gs, p.y * gs, p.z * gs, p.w * gs);
    if (prow) *reinterpret_cast<float4*>(prow + j) = p;
  }
  // the thread that owns column y patches d[y] = (p_y - 1) * gs after its own vector store (same thread, same
  // address: program order)
  if (drow && y >= 0 && y < V && ((y >> 2) & 255) == tid) drow[y] = (py - 1.f) * gs;
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* x, int* out, int rows, int V, int ld) {
  __shared__ ArgMax sha[4];
  const int row = blockIdx.x;
  ArgMax am; am.v = -INFINITY; am.i = 0x7fffffff;
  for (int j = threadIdx.x; j < V; j += 256) {
    const float v = x[(long)row * ld + j];
    if (v > am.v) { am.v = v; am.i = j; }
  }
  am = block_argmax(am, sha);
  if (threadIdx.x == 0) out[row] = am.i == 0x7fffffff ? 0 : am.i;
}

__global__ __launch_bounds__(256) void onehot_argmax_kernel(const float* onehot, int* ids, int B, int T, int V) {
  __shared__ ArgMax sha[4];
  const int k = blockIdx.x;          // k = b*T + t
  const int b = k / T, t = k % T;
  const float* x = onehot + (long)k * V;
  ArgMax am; am.v = -INFINITY; am.i = 0x7fffffff;
  for (int j = threadIdx.x; j < V; j += 256) {
    const float v = x[j];
    if (v > am.v) { am.v = v; am.i = j; }
  }
  am = block_argmax(am, sha);
  if (threadIdx.x == 0) ids[t * B + b] = am.i == 0x7fffffff ? 0 : am.i;
}


// Beam-search expansion, one workgroup per sample (the reference only sketches beam search: lc_NIC.py:640-692,
// ThinkAndTell/evaluate.py:203-228; the definition below is this library's, restated by oracle LcNIC.beam_search):
// the k beams of a sample are rows b*k .. b*k+k-1 of probs; candidate (j, v) scores  score[j] + log(max(p[j][v], 1e-30));
// a finished beam (it has emitted end_id) has ONE candidate: token 0 (padding) at its own score.  The k best
// candidates become the new beams, ties -> the lower flat index j*V + v (the first-maximum rule of np.argmax).
// out: score_out[b*k + r], parent[b*k + r] = global row of the beam it extends, token[b*k + r], fin_out.
__global__ __launch_bounds__(256) void beam_topk_kernel(const float* probs, const float* score_in, const int* fin_in,
                                                        int V, int ld, int k, int end_id, float* score_out,
                                                        int* parent, int* token, int* fin_out) {
  __shared__ ArgMax sha[4];
  __shared__ int taken[16];
  __shared__ float sc[16];
  __shared__ int fn[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < k) { sc[tid] = score_in[b * k + tid]; fn[tid] = fin_in[b * k + tid]; }
  __syncthreads();
  const int ncand = k * V;
  for (int r = 0; r < k; ++r) {
    ArgMax am; am.v = -INFINITY; am.i = 0x7fffffff;
    for (int cnd = tid; cnd < ncand; cnd += 256) {
      const int j = cnd / V, v = cnd - j * V;
      bool skip = false;
      for (int q = 0; q < r; ++q) skip |= taken[q] == cnd;
      if (skip) continue;
      float val;
      if (fn[j]) val = v == 0 ? sc[j] : -INFINITY;
      else val = sc[j] + logf(fmaxf(probs[(long)(b * k + j) * ld + v], 1e-30f));
      if (val > am.v) { am.v = val; am.i = cnd; }
    }
    am = block_argmax(am, sha);
    if (tid == 0) {
      const int cnd = am.i == 0x7fffffff ? 0 : am.i;
      const int j = cnd / V, v = cnd - j * V;
      taken[r] = cnd;
      score_out[b * k + r] = am.v;
      parent[b * k + r] = b * k + j;
      token[b * k + r] = v;
      fin_out[b * k + r] = (fn[j] || v == end_id) ? 1 : 0;
    }
    __syncthreads();
  }
}

// out[0] = mean_i (c - x[i])^2, fixed reduction order (one workgroup): tf.keras.losses.MeanSquaredError()(c, x) for a
// constant target -- the 'attention' entry lc_NIC.train_step_sam returns (lc_NIC.py:813-814,838)
__global__ __launch_bounds__(1024) void sqdiff_mean_kernel(const float* x, long n, float c, float* out) {
  __shared__ float s[16];
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) { const float d = c - x[i]; a += d * d; }
  a = tnt_wave_sum(a);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += s[w];
    out[0] = t / (float)n;
  }
}

// Categorical sampling, one workgroup per row (tf.random.categorical(logits / temperature, 1),
// ThinkAndTell/evaluate.py:223,278; lc_NIC.sample_choice, lc_NIC.py:571-575).  Inverse CDF in a fixed
// order with one Philox uniform per row, so the CPU oracle can restate it:
//   w_j = exp((l_j - max_j l) / temperature), l = x (logits) or log(x) (probabilities)
//   pick the first j with  w_0 + ... + w_j > u * sum(w),  u = uniform24(element = row)
__global__ __launch_bounds__(256) void sample_rows_kernel(const float* x, int* out, int rows, int V, int ld,
                                                          float inv_temp, int from_logits, uint64_t seed, uint32_t site,
                                                          uint32_t step, const uint32_t* step_dev) {
  __shared__ float shm[4];
  __shared__ float part[257];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (long)row * ld;
  const int C = (V + 255) / 256;
  const int j0 = tid * C, j1 = min(V, j0 + C);
  float mx = -INFINITY;
  for (int j = j0; j < j1; ++j) {
    const float l = from_logits ? xr[j] : logf(xr[j]);
    mx = fmaxf(mx, l);
  }
  mx = tnt_wave_max(mx);
  if ((tid & 63) == 0) shm[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
  float loc = 0.f;
  for (int j = j0; j < j1; ++j) {
    const float l = from_logits ? xr[j] : logf(xr[j]);
    loc += expf((l - mx) * inv_temp);
  }
  part[tid + 1] = loc;
  __syncthreads();
  if (tid == 0) {                      // fixed-order prefix over the 256 chunk sums
    part[0] = 0.f;
    float run = 0.f;
    for (int t = 1; t <= 256; ++t) { run += part[t]; part[t] = run; }
  }
  __syncthreads();
  if (step_dev) step += step_dev[0];
  const uint64_t e = (uint64_t)row, g = e >> 2;
  const TntPhilox4 r = tnt_philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), site, step, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t sel = (uint32_t)e & 3u;      // select chain: a runtime index would put r.v in scratch memory
  const uint32_t rw = sel == 0u ? r.v[0] : (sel == 1u ? r.v[1] : (sel == 2u ? r.v[2] : r.v[3]));
  const float u = (float)(rw >> 8) * 5.9604644775390625e-08f;
  const float target = u * part[256];
  const int tlast = (V - 1) / C;       // last thread that owns elements (takes the u*sum == sum rounding case)
  if (part[tid] <= target && (target < part[tid + 1] || tid == tlast)) {
    float run = part[tid];
    int pick = j1 - 1;
    for (int j = j0; j < j1; ++j) {
      const float l = from_logits ? xr[j] : logf(xr[j]);
      run += expf((l - mx) * inv_temp);
      if (run > target) { pick = j; break; }
    }
    out[row] = pick;
  }
}

}  // namespace

extern "C" int32_t tnt_sample_rows_f32(const float* x, int32_t* out, int32_t rows, int32_t V, int32_t ld,
                                       float temperature, int32_t from_logits, uint64_t seed, uint32_t site,
                                       uint32_t step, const uint32_t* step_dev, void* stream) {
  if (rows <= 0 || V <= 0 || !(temperature > 0.f)) return TNT_BADARG(3);
  hipLaunchKernelGGL(sample_rows_kernel, dim3(rows), dim3(256), 0, tnt_stream(stream), x, out, rows, V, ld,
                     1.f / temperature, from_logits, seed, site, step, step_dev);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_embedding_fwd_f32(const float* table, const int32_t* ids, float* out, int32_t B, int32_t T,
                                         int32_t E, int32_t ldo, int32_t V, void* stream) {
  hipLaunchKernelGGL(emb_fwd_kernel, dim3((B * T + 3) / 4), dim3(256), 0, tnt_stream(stream), table, ids, out, B, T, E,
                     ldo, V);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_embedding_fwd_drop2_f32(const float* table, const int32_t* ids, float* out, float* out_drop,
                                               int32_t B, int32_t T, int32_t E, int32_t ldo, int32_t V, float rate,
                                               uint64_t seed, uint32_t site, uint32_t step, const uint32_t* step_dev,
                                               float rate2, uint32_t site2, int32_t lwidth2, int32_t lcol0_2, void* stream) {
  if (E % 4 != 0 || ldo % 4 != 0 || !tnt_aligned16(table) || !tnt_aligned16(out_drop) || (out && !tnt_aligned16(out)))
    return TNT_BADARG(7);
  if (rate < 0.f || rate >= 1.f || rate2 < 0.f || rate2 >= 1.f) return TNT_BADARG(10);
  if (rate2 > 0.f && ((lwidth2 | lcol0_2) % 4 != 0 || lcol0_2 < 0 || lcol0_2 + E > lwidth2)) return TNT_BADARG(17);
  hipLaunchKernelGGL(emb_fwd_drop_kernel, dim3((B * T + 3) / 4), dim3(256), 0, tnt_stream(stream), table, ids, out,
                     out_drop, B, T, E, ldo, V, rate, seed, site, step, step_dev, rate2, site2, lwidth2, lcol0_2);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_embedding_fwd_drop_f32(const float* table, const int32_t* ids, float* out, float* out_drop,
                                              int32_t B, int32_t T, int32_t E, int32_t ldo, int32_t V, float rate,
                                              uint64_t seed, uint32_t site, uint32_t step, const uint32_t* step_dev,
                                              void* stream) {
  return tnt_embedding_fwd_drop2_f32(table, ids, out, out_drop, B, T, E, ldo, V, rate, seed, site, step, step_dev, 0.f, 0, 0, 0,
                                     stream);
}

extern "C" int32_t tnt_embedding_bwd_f32(const float* drows, const int32_t* ids, float* dtable, float* sq_norm,
                                         float* rowsq_work, int32_t B, int32_t T, int32_t E, int32_t ldd, int32_t V,
                                         void* stream) {
  hipStream_t s = tnt_stream(stream);
  if (sq_norm) {
    if (!rowsq_work) return TNT_BADARG(5);
    hipLaunchKernelGGL(rowsq_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, drows, rowsq_work, B * T, E, ldd);
    TNT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_accum_kernel, dim3(1), dim3(1024), 0, s, rowsq_work, sq_norm, B * T);
    TNT_LAUNCH_CHECK();
  }
  {
    const long nz = (long)V * E;
    long zb = (nz + 255) / 256;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)(zb > 2048 ? 2048 : zb)), dim3(256), 0, s, dtable, nz);
    TNT_LAUNCH_CHECK();
  }
  if (E % 4 == 0 && ldd % 4 == 0 && tnt_aligned16(drows) && tnt_aligned16(dtable))
    hipLaunchKernelGGL(emb_bwd4_kernel, dim3(B * T, (E + 255) / 256), dim3(256), 0, s, drows, ids, dtable, B, T, E, ldd, V);
  else
    hipLaunchKernelGGL(emb_bwd_kernel, dim3(B * T, (E + 63) / 64), dim3(256), 0, s, drows, ids, dtable, B, T, E, ldd, V);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_softmax_cce_f32(const float* logits, const int32_t* target, float* probs, float* loss_row,
                                       float* correct_row, float* dlogits, int32_t rows, int32_t V, int32_t ld,
                                       float gscale, int32_t from_logits, int32_t mask_zero, void* stream) {
  if (rows <= 0) return 0;
  hipStream_t s = tnt_stream(stream);
  const bool al = (ld % 4 == 0) && tnt_aligned16(logits) && (!probs || tnt_aligned16(probs)) &&
                  (!dlogits || tnt_aligned16(dlogits));
  const int nv4 = (V + 1023) / 1024;
#define TNT_SMX(N)                                                                                              \
  hipLaunchKernelGGL((softmax_cce_reg_kernel<N>), dim3(rows), dim3(256), 0, s, logits, target, probs, loss_row, \
                     correct_row, dlogits, rows, V, ld, gscale, from_logits, mask_zero)
  if (al && nv4 == 1) TNT_SMX(1);
  else if (al && nv4 == 2) TNT_SMX(2);
  else if (al && nv4 <= 4) TNT_SMX(4);
  else if (al && nv4 <= 5) TNT_SMX(5);
  else if (al && nv4 <= 8) TNT_SMX(8);
  else
    hipLaunchKernelGGL(softmax_cce_kernel, dim3(rows), dim3(256), 0, s, logits, target, probs, loss_row, correct_row,
                       dlogits, rows, V, ld, gscale, from_logits, mask_zero);
#undef TNT_SMX
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_onehot_argmax_f32(const float* onehot, int32_t* ids_tmajor, int32_t B, int32_t T, int32_t V,
                                         void* stream) {
  hipLaunchKernelGGL(onehot_argmax_kernel, dim3(B * T), dim3(256), 0, tnt_stream(stream), onehot, ids_tmajor, B, T, V);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_argmax_rows_f32(const float* x, int32_t* out, int32_t rows, int32_t V, int32_t ld, void* stream) {
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, tnt_stream(stream), x, out, rows, V, ld);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_beam_topk_f32(const float* probs, const float* score_in, const int32_t* fin_in, int32_t B,
                                     int32_t V, int32_t ld, int32_t k, int32_t end_id, float* score_out,
                                     int32_t* parent, int32_t* token, int32_t* fin_out, void* stream) {
  if (B <= 0 || V <= 0 || k <= 0 || k > 16) return TNT_BADARG(7);
  if (score_out == score_in || fin_out == fin_in) return TNT_BADARG(9);
  hipLaunchKernelGGL(beam_topk_kernel, dim3(B), dim3(256), 0, tnt_stream(stream), probs, score_in, fin_in, V, ld, k,
                     end_id, score_out, parent, token, fin_out);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_sqdiff_mean_f32(const float* x, float* out, int64_t n, float c, void* stream) {
  if (n <= 0) return TNT_BADARG(3);
  hipLaunchKernelGGL(sqdiff_mean_kernel, dim3(1), dim3(1024), 0, tnt_stream(stream), x, (long)n, c, out);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_embedding_bwd_parts(int32_t B, int32_t T, int32_t E) { return B * T * ((E + 255) / 256); }

extern "C" int32_t tnt_embedding_bwd_sparse_f32(const float* drows, const int32_t* ids, const int32_t* prev_ids,
                                                float* dtable, float* sq_part, int32_t B, int32_t T, int32_t E,
                                                int32_t ldd, int32_t V, float drop_rate, uint64_t drop_seed,
                                                uint32_t drop_site, const uint32_t* drop_step_dev, int32_t zero_id,
                                                void* stream) {
  if (B <= 0 || T <= 0 || E <= 0 || V <= 0) return TNT_BADARG(6);
  if (E % 4 != 0 || ldd % 4 != 0 || !tnt_aligned16(drows) || !tnt_aligned16(dtable)) return TNT_BADARG(1);
  if (prev_ids == nullptr || sq_part == nullptr || prev_ids == ids) return TNT_BADARG(3);
  if (zero_id >= V) return TNT_BADARG(14);
  if (drop_rate < 0.f || drop_rate >= 1.f) return TNT_BADARG(11);
  const EmbDrop drop{drop_rate, 1.0f / (1.0f - drop_rate), drop_seed, drop_site, drop_step_dev};
  if (zero_id >= 0)
    hipLaunchKernelGGL(emb_bwd_sparse_kernel<4>, dim3(2 * B * T, (E + 255) / 256), dim3(256), 0, tnt_stream(stream), drows,
                       ids, prev_ids, dtable, sq_part, B, T, E, ldd, V, drop, zero_id);
  else
    hipLaunchKernelGGL(emb_bwd_sparse_kernel<16>, dim3(2 * B * T, (E + 255) / 256), dim3(1024), 0, tnt_stream(stream), drows,
                       ids, prev_ids, dtable, sq_part, B, T, E, ldd, V, drop, zero_id);
  TNT_LAUNCH_CHECK();
  return 0;
}
