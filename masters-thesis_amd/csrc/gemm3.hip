// gemm3: the FP32-MFMA GEMM family of round 3 (include/tnt_hip.h: tnt_gemm3_f32).
//
// Replaces the vendor-library products of a training step -- the vocabulary head forward and its two gradients
// (NIC.py:143,248-249; lc_NIC.py:261,386-387), the LSTM input projection and the LSTM kernel / recurrent-kernel / input
// gradients (keras LSTM under tape.gradient, NIC.py:138-140) -- with hand-written gfx950 kernels.  C = op(A) op(B)
// (+ bias[N]), exact f32 (v_mfma_f32_16x16x4_f32 == an fmaf chain in k order), fixed summation order.
//
// Structure (one template, three operand layouts):
//  * An operand is either K-CONTIGUOUS ("KC": rows of the stored matrix run along k -- A of NN / NT, B of NT) or
//    M/N-CONTIGUOUS ("MC": rows run along the output dimension -- B of NN / TN, A of TN).
//  * Global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no VGPR staging, no ds_write, hardware bounds check
//    (rows past the matrix read as zero).  A ring of NS K-stages per workgroup, ONE s_barrier per stage.
//    Every wave issues its share of a stage's pieces behind that barrier, spaced between MFMAs.  (A dedicated loader
//    wave -- one wave issuing all 26-36 pieces of a stage for four consumer waves -- was built and measured slower on
//    every shape: one wave cannot issue pieces fast enough; see DESIGN.md.)
//  * LDS images are written lane-linear (that is what LDS-DMA does); bank conflicts are avoided by permuting the SOURCE
//    address instead: a KC image [rows][BK] stores 16-byte chunk c of row r at chunk position c ^ sw(r), so that the
//    ds_read_b128 of one MFMA operand fragment (16 rows x 4 k per 16-lane group) touches every bank once; an MC image
//    [BK][cols] is read along cols, which is conflict-free as it stands.
//  * Every operand fragment comes from ONE ds_read_b128 per four MFMAs' worth of that operand:
//      KC: lane (r = lane & 15, q = lane >> 4) reads k = 16 kb + 4 q + {0..3} of row r      -> element e feeds k-step e;
//      MC: lane reads columns 4 r + {0..3} of k-row 16 kb + 4 q + e                           -> element c feeds tile 4 g + c,
//    i.e. an MC operand's four MFMA tiles of a 64-wide group interleave their columns (tile c owns columns = c mod 4).
//    The k order inside a 16-deep block is a permutation (lane group q takes k = 16 kb + 4 q + e at step e); both
//    operands use the same one, so the sum is the plain dot product, in a fixed order.
//  * Epilogue: with an MC B operand a lane holds four adjacent columns of a row in four accumulators -> one 16-byte
//    store per row and 64-column group.
//  * Work decomposition: one workgroup per (output tile, K split), dealt XCD-contiguously so that the S splits of a tile
//    share an XCD.  With S > 1 the S workgroups of a tile reduce IN the launch: every accumulator tile row (wave w, tile
//    row tm) has an owner split (w * TM + tm) mod S; a workgroup stores the rows it does not own to the work buffer
//    (write-through), arrives on the tile's counter, waits for the S arrivals and adds the peers' partials to the rows it
//    owns, in split order (bitwise reproducible), before the ordinary epilogue.  Each workgroup moves (S-1)/S of a tile
//    out and in; no reduce launch, no second pass over C.
#include "tnt_common.h"
#include <type_traits>

namespace {

struct G3Args {
  const float* A; const float* B; float* C; const float* bias;
  const float* A2; float* C2;   // optional second product C2 = op(A2) op(B) in the same launch (blockIdx.y = 1)
  float* colsum;                // optional (TN, no split): colsum[n] = sum_k B[k][n]
  float* work; unsigned* sync;  // split-K exchange: partial tiles (sentinel-armed); sync[0] = error word
  int M, N, K, lda, ldb, ldc;
  int nst;                    // K stages in all
  int splitk, nst_split;      // workgroups per output tile, stages per workgroup
  int err_word;               // index of the error word in sync
  int nx;                     // workgroups per product (tiles x splitk)
};

// One LDS-DMA piece: 64 lanes x 16 bytes from the buffer `rsrc` + voff (per lane; out-of-range lanes deliver zeros) to
// the 1 KB at LDS byte address `dst` (wave-uniform), lane-linear.  Inline asm on purpose: hipcc's waitcnt pass treats an
// LDS-DMA it can see as a store that any later ds_read may alias and puts `s_waitcnt vmcnt(0)` in front of the next
// fragment read -- which drains the prefetch ring every stage.  Hidden in asm the pieces are counted by hand
// (s_waitcnt vmcnt(N) before the barrier that publishes a stage); the compiler's own waits can only over-wait.
// M0 carries the LDS address and is written in the same statement that uses it.
typedef int g3_v4i __attribute__((ext_vector_type(4)));
constexpr int G3_SENTINEL = 0x7FC5EED5;       // "not written yet" in the split-K exchange buffer (a quiet NaN payload)
__device__ __forceinline__ g3_v4i g3_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  return g3_v4i{(int)(unsigned)p, (int)(unsigned)((p >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void g3_dma16(g3_v4i rsrc, unsigned dst, int voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
               :: "v"(voff), "s"(dst), "s"(rsrc) : "memory");
#endif
}

// chunk swizzle of a KC image row (CH = BK / 4 sixteen-byte chunks per row); derivations in DESIGN.md section 4
template <int CH> __device__ __forceinline__ int g3_sw(int row) {
  if constexpr (CH == 16) return row & 15;
  else if constexpr (CH == 8) return (row >> 1) & 7;
  else return (0 - (row >> 2)) & 3;
}

// One operand of the product: its LDS image, its LDS-DMA fill and its MFMA fragments.
//   KC   : K-contiguous (image [EXT][BK], chunk-swizzled) or M/N-contiguous (image [BK][EXT])
//   EXT  : extent of the workgroup tile along this operand's output dimension
//   T    : 16-wide MFMA tiles per wave along that dimension;  NW: waves of the workgroup
template <bool KC, int EXT, int T, int BK, int NW>
struct G3Op {
  static constexpr int CH = BK / 4;
  static constexpr int BYTES = EXT * BK * 4;
  static constexpr int NI = BYTES / 1024;                 // LDS-DMA pieces (wave-instructions) per stage
  static constexpr int NIW = (NI + NW - 1) / NW;          // ... per wave (pieces are dealt round-robin)
  static constexpr int G4 = KC ? 0 : T / 4, G2 = KC ? 0 : (T % 4) / 2, G1 = KC ? 0 : T % 2;
  static_assert(BYTES % 1024 == 0, "stage image must be whole 1 KB LDS-DMA pieces");
  static_assert(!KC || (EXT * CH) % 64 == 0, "KC image: whole pieces of 64 chunks");

  // byte offset inside the matrix of what lane `lane` of piece j loads (k0 = 0); kc4 = first k of its chunk (KC)
  static __device__ __forceinline__ int src_off(int j, int lane, int row0, int ld, int& kc4) {
    const int s = j * 64 + lane;
    if constexpr (KC) {
      const int row = s / CH, pos = s % CH, c = pos ^ g3_sw<CH>(row);
      kc4 = 4 * c;
      return ((row0 + row) * ld + 4 * c) * 4;
    } else {
      const int krow = s / (EXT / 4), c4 = s % (EXT / 4);
      kc4 = 0;
      return (krow * ld + row0 + 4 * c4) * 4;
    }
  }
};

// L1-bypassing 16-byte load / write-through 16-byte store for the split-K exchange (bytes another workgroup wrote /
// will read in this launch): buffer_load_dwordx4 ... sc1 / buffer_store_dwordx4 ... sc0 sc1
__device__ __forceinline__ floatx4 g3_ld4_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, /*sc1*/ 16));
}
__device__ __forceinline__ void g3_st4_wt(__amdgpu_buffer_rsrc_t rs, unsigned off, floatx4 v) {
  typedef unsigned g3_v4u __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(g3_v4u, v), rs, (int)off, 0, /*sc0 sc1*/ 17);
}

// One configuration of the family: operand layouts, wave tile (TM x TN MFMA tiles), wave grid, stage depth, ring depth.
// body(g, bx, by) is the whole workgroup program for unit bx of product by (0, or 1 = the second product of a dual launch).
template <bool A_KC, bool B_KC, int TM, int TN, int WGM, int WGN, int BK, int NS>
struct G3Cfg {
  static constexpr int NTHREADS = 64 * WGM * WGN;
  static constexpr int BM_ = 16 * TM * WGM, BN_ = 16 * TN * WGN, BK_ = BK;
  static constexpr int SHMEM = NS * (BM_ + BN_) * BK * 4 + 1024;
  static constexpr bool TA = !A_KC, TB = B_KC;
  static_assert(SHMEM <= 160 * 1024, "LDS ring does not fit");
  static __device__ __forceinline__ void body(const G3Args& g, int bx, int by);
};

template <bool A_KC, bool B_KC, int TM, int TN, int WGM, int WGN, int BK, int NS>
__device__ __forceinline__ void G3Cfg<A_KC, B_KC, TM, TN, WGM, WGN, BK, NS>::body(const G3Args& g, const int bx, const int by) {
  constexpr int NW = WGM * WGN;
  constexpr int WM = 16 * TM, WN = 16 * TN, BM = WM * WGM, BN = WN * WGN;
  using OA = G3Op<A_KC, BM, TM, BK, NW>;
  using OB = G3Op<B_KC, BN, TN, BK, NW>;
  constexpr int STAGE = OA::BYTES + OB::BYTES;
  constexpr int KB = BK / 16;                         // 16-deep k blocks per stage
  constexpr bool KMASK = A_KC && B_KC;                // NT: the k tail of A must be zeroed by hand (B's then meets zeros)
  constexpr int NPW = OA::NIW + OB::NIW;              // pieces per wave and stage -- the SAME for every wave (vmcnt counts
  constexpr int SCRATCH = NS * STAGE;                 // them): a wave with no real piece left issues an out-of-range one
  static_assert(KB % 2 == 0, "register buffer parity must repeat per stage");      // into 1 KB of scratch
  static_assert((NS - 2) * NPW <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(1024))) unsigned char g3_lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 15, q = lane >> 4;
  const int MT = (g.M + BM - 1) / BM, NTl = (g.N + BN - 1) / BN;
  const int S = g.splitk;
  const int nwg = MT * NTl * S, bid = bx;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;       // XCD-contiguous unit order (bijective for any nwg)
  const int u = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int t = u / S, z = u - t * S;                          // output tile, K split
  const int m0 = (t % MT) * BM, n0 = (t / MT) * BN;
  const bool second = by != 0;                                 // the second product of a dual launch
  const float* Ag = second ? g.A2 : g.A;
  float* Cg = second ? g.C2 : g.C;
  const int s_begin = z * g.nst_split;
  const int s_end = min(g.nst, s_begin + g.nst_split);        // this workgroup's stages [s_begin, s_end)
  const int nloc = max(s_end - s_begin, 0);

  // ---- LDS-DMA sources (the extern array is the kernel's only LDS object: byte offsets inside it are LDS addresses)
  const int rowsA = A_KC ? g.M : g.K, rowsB = B_KC ? g.N : g.K;
  const g3_v4i rsA = g3_rsrc(Ag, (unsigned)(rowsA * g.lda * 4));
  const g3_v4i rsB = g3_rsrc(g.B, (unsigned)(rowsB * g.ldb * 4));
  int offA[OA::NIW], offB[OB::NIW], kcA[KMASK ? OA::NIW : 1];
  {
    const int ka = s_begin * BK;
#pragma unroll
    for (int i = 0; i < OA::NIW; ++i) {
      int kc4;
      const int j = wave + i * NW;
      offA[i] = OA::src_off(j, lane, m0, g.lda, kc4) + (A_KC ? ka * 4 : ka * g.lda * 4);
      if (OA::NI % NW != 0 && j >= OA::NI) offA[i] = 0x7fffffff;        // a dummy piece: out of range for good
      if constexpr (KMASK) kcA[i] = kc4 + ka;
    }
#pragma unroll
    for (int i = 0; i < OB::NIW; ++i) {
      int kc4;
      const int j = wave + i * NW;
      offB[i] = OB::src_off(j, lane, n0, g.ldb, kc4) + (B_KC ? ka * 4 : ka * g.ldb * 4);
      if (OB::NI % NW != 0 && j >= OB::NI) offB[i] = 0x7fffffff;
    }
  }
  const int k_end = min(((g.K + 3) & ~3), s_end * BK);       // first k this workgroup must not multiply (NT mask)
  const int stepA = A_KC ? BK * 4 : BK * g.lda * 4, stepB = B_KC ? BK * 4 : BK * g.ldb * 4;
  // Piece p (0 .. NPW-1) of this wave for the NEXT stage not yet issued -> the LDS buffer at byte offset boff; the
  // per-lane offsets then advance by one stage.  Stages past the workgroup's last one are issued too (every stage has the
  // same piece count): an MC operand's rows past K are out of range (zeros, no traffic); a KC operand's columns past K
  // read on into the following rows -- harmless, never multiplied (NT: masked; NN: B's rows there are zero).
  auto issue_piece = [&](int boff, int p) __attribute__((always_inline)) {
    if (p < OA::NIW) {
      const int i = p, j = wave + i * NW;
      const bool real = (OA::NI % NW == 0) || j < OA::NI;
      int vo = offA[i];
      if constexpr (KMASK) { vo = (kcA[i] < k_end) ? vo : 0x7fffffff; kcA[i] += BK; }
      g3_dma16(rsA, (unsigned)(real ? boff + j * 1024 : SCRATCH), vo);
      if ((OA::NI % NW == 0) || real) offA[i] += stepA;
    } else {
      const int i = p - OA::NIW, j = wave + i * NW;
      const bool real = (OB::NI % NW == 0) || j < OB::NI;
      g3_dma16(rsB, (unsigned)(real ? boff + OA::BYTES + j * 1024 : SCRATCH), offB[i]);
      if ((OB::NI % NW == 0) || real) offB[i] += stepB;
    }
  };

  // ---- fragment read addresses (bytes inside a stage buffer)
  int fa[A_KC ? KB : 3], fb[B_KC ? KB : 3];
  if constexpr (A_KC) {
    const int row = wm * WM + r;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) fa[kb] = (row * OA::CH + ((4 * kb + q) ^ g3_sw<OA::CH>(row))) * 16;
  } else {
    fa[0] = (4 * q * BM + wm * WM + 4 * r) * 4;
    fa[1] = (4 * q * BM + wm * WM + 64 * OA::G4 + 2 * r) * 4;
    fa[2] = (4 * q * BM + wm * WM + 64 * OA::G4 + 32 * OA::G2 + r) * 4;
  }
  if constexpr (B_KC) {
    const int row = wn * WN + r;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) fb[kb] = OA::BYTES + (row * OB::CH + ((4 * kb + q) ^ g3_sw<OB::CH>(row))) * 16;
  } else {
    fb[0] = OA::BYTES + (4 * q * BN + wn * WN + 4 * r) * 4;
    fb[1] = OA::BYTES + (4 * q * BN + wn * WN + 64 * OB::G4 + 2 * r) * 4;
    fb[2] = OA::BYTES + (4 * q * BN + wn * WN + 64 * OB::G4 + 32 * OB::G2 + r) * 4;
  }

  float av[2][TM][4], bv[2][TN][4];                     // [register buffer][tile][k-step of the block]
  // fragments of k block kb of the LDS buffer at byte offset boff -> register buffer `rb`
  auto fetch = [&](int boff, int kb, int rb) __attribute__((always_inline)) {
    const unsigned char* base = g3_lds + boff;
    if constexpr (A_KC) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const floatx4 v = *reinterpret_cast<const floatx4*>(base + fa[kb] + tm * 16 * OA::CH * 16);
        av[rb][tm][0] = v.x; av[rb][tm][1] = v.y; av[rb][tm][2] = v.z; av[rb][tm][3] = v.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ko = (16 * kb + e) * BM * 4;
#pragma unroll
        for (int gq = 0; gq < OA::G4; ++gq) {
          const floatx4 v = *reinterpret_cast<const floatx4*>(base + fa[0] + ko + gq * 256);
          av[rb][4 * gq + 0][e] = v.x; av[rb][4 * gq + 1][e] = v.y; av[rb][4 * gq + 2][e] = v.z; av[rb][4 * gq + 3][e] = v.w;
        }
        if constexpr (OA::G2) {
          const float2 v = *reinterpret_cast<const float2*>(base + fa[1] + ko);
          av[rb][4 * OA::G4][e] = v.x; av[rb][4 * OA::G4 + 1][e] = v.y;
        }
        if constexpr (OA::G1) av[rb][TM - 1][e] = *reinterpret_cast<const float*>(base + fa[2] + ko);
      }
    }
    if constexpr (B_KC) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const floatx4 v = *reinterpret_cast<const floatx4*>(base + fb[kb] + tn * 16 * OB::CH * 16);
        bv[rb][tn][0] = v.x; bv[rb][tn][1] = v.y; bv[rb][tn][2] = v.z; bv[rb][tn][3] = v.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ko = (16 * kb + e) * BN * 4;
#pragma unroll
        for (int gq = 0; gq < OB::G4; ++gq) {
          const floatx4 v = *reinterpret_cast<const floatx4*>(base + fb[0] + ko + gq * 256);
          bv[rb][4 * gq + 0][e] = v.x; bv[rb][4 * gq + 1][e] = v.y; bv[rb][4 * gq + 2][e] = v.z; bv[rb][4 * gq + 3][e] = v.w;
        }
        if constexpr (OB::G2) {
          const float2 v = *reinterpret_cast<const float2*>(base + fb[1] + ko);
          bv[rb][4 * OB::G4][e] = v.x; bv[rb][4 * OB::G4 + 1][e] = v.y;
        }
        if constexpr (OB::G1) bv[rb][TN - 1][e] = *reinterpret_cast<const float*>(base + fb[2] + ko);
      }
    }
  };

  floatx4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  // column sums of B (the bias gradient of the layer whose kernel gradient this product is; TN only): the waves of the
  // first tile row keep a running sum of the B fragments they hold anyway.  Branch-free (every wave runs the FMAs with a
  // 0 / 1 factor) so that the pinned block below stays one basic block.
  constexpr bool CS = !A_KC && !B_KC;
  float cs[CS ? TN : 1];
  const float csf = (CS && g.colsum != nullptr && m0 == 0 && wm == 0 && !second) ? 1.f : 0.f;
#pragma unroll
  for (int tn = 0; tn < (CS ? TN : 1); ++tn) cs[tn] = 0.f;
  auto colacc = [&](int rb) __attribute__((always_inline)) {
    if constexpr (CS) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        cs[tn] = fmaf((bv[rb][tn][0] + bv[rb][tn][1]) + (bv[rb][tn][2] + bv[rb][tn][3]), csf, cs[tn]);
    }
  };
  // MFMAs i0 .. i1-1 of a block in the order (k-step e, tile row, tile column): 4 TM TN per block
  auto mma = [&](int rb, int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4 * TM * TN; ++i) {
      if (i < i0 || i >= i1) continue;
      const int e = i / (TM * TN), tm = (i / TN) % TM, tn = i % TN;
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb][tm][e], bv[rb][tn][e], acc[tm][tn], 0, 0, 0);
    }
  };

  // epilogue operands before the loop (a load still pending in the epilogue serialises the stores behind vmcnt(0))
  float bcol[TN];
  auto col_of = [&](int tn) {                 // column of accumulator tile tn for this lane (tile column r)
    if constexpr (B_KC) return n0 + wn * WN + 16 * tn + r;
    else {
      if (tn < 4 * OB::G4) return n0 + wn * WN + 64 * (tn / 4) + 4 * r + (tn % 4);
      if (tn < 4 * OB::G4 + 2 * OB::G2) return n0 + wn * WN + 64 * OB::G4 + 2 * r + (tn - 4 * OB::G4);
      return n0 + wn * WN + 64 * OB::G4 + 32 * OB::G2 + r;
    }
  };
  auto row_of = [&](int tm, int i) {          // row of accumulator tile tm, tile row i = 4 q + reg
    if constexpr (A_KC) return m0 + wm * WM + 16 * tm + i;
    else {
      if (tm < 4 * OA::G4) return m0 + wm * WM + 64 * (tm / 4) + 4 * i + (tm % 4);
      if (tm < 4 * OA::G4 + 2 * OA::G2) return m0 + wm * WM + 64 * OA::G4 + 2 * i + (tm - 4 * OA::G4);
      return m0 + wm * WM + 64 * OA::G4 + 32 * OA::G2 + i;
    }
  };
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = col_of(tn);
    bcol[tn] = (g.bias != nullptr && col < g.N) ? g.bias[col] : 0.f;
  }

  // ---- main loop.  A ring of NS LDS buffers, at most NS - 1 stages in flight.  Two k blocks live in registers
  // (register buffers 0 / 1; a stage has an even number of blocks, so every stage starts on register buffer 0).  Block b of
  // a stage multiplies register buffer b & 1 while the fragments of the next block are read.  The LAST block of a stage
  // first retires the stage's LDS traffic (this wave has read all its fragments: lgkmcnt(0); its pieces of the next stage
  // have landed: vmcnt leaves only the younger stages outstanding), meets the other waves, refills the buffer just freed
  // with the next stage not yet issued and reads the first fragments of stage s + 1.
  // Instruction order is pinned.  hipcc otherwise sinks every ds_read down to its first use ("reads; lgkmcnt(0); MFMAs",
  // ~150 exposed cycles per block with one wave per SIMD): the reads of the next block go one per MFMA behind the first
  // MFMAs of this block (sched_group_barrier); the LDS-DMA pieces, asm statements the scheduler cannot classify, sit
  // DMA_GAP MFMAs apart between scheduling fences.
  constexpr int NRD = (A_KC ? TM : 4 * (OA::G4 + OA::G2 + OA::G1)) + (B_KC ? TN : 4 * (OB::G4 + OB::G2 + OB::G1));
  constexpr int NMM = 4 * TM * TN;
  constexpr int DMA_GAP = (NMM - NRD) / NPW >= 4 ? 4 : ((NMM - NRD) / NPW >= 2 ? 2 : 1);
  static_assert(NRD + NPW * DMA_GAP <= NMM, "not enough MFMAs in a block to carry its loads");
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int p = 0; p < NPW; ++p) issue_piece(s * STAGE, p);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * NPW) : "memory");       // the first stage has landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  fetch(0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  int bcur = 0;
  for (int s = 0; s < nloc; ++s) {
    const int bnext = (bcur + STAGE == NS * STAGE) ? 0 : bcur + STAGE;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int rb = kb & 1;
      if (kb + 1 < KB) {
        fetch(bcur, kb + 1, rb ^ 1);
        mma(rb, 0, NMM);
        colacc(rb);
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMM - NRD, 0);
      } else {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * NPW) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        fetch(bnext, 0, rb ^ 1);
        mma(rb, 0, NRD);
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int p = 0; p < NPW; ++p) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(bcur, p);
          mma(rb, NRD + p * DMA_GAP, NRD + (p + 1) * DMA_GAP);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(rb, NRD + NPW * DMA_GAP, NMM);
        colacc(rb);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    bcur = bnext;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the pieces issued past the last stage

  if constexpr (CS) {
    if (csf != 0.f) {            // wave-uniform.  Lane group q summed the k rows 16 kb + 4 q + e: fold the four groups
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        float v = cs[tn];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const int col = col_of(tn);
        if (q == 0 && col < g.N) g.colsum[col] = v;
      }
    }
  }

  // ---- split-K fix-up (see the header): rows owned by another split go out, rows owned by this one take the peers in.
  // THE DATA IS ITS OWN FLAG: every 16-byte granule of the work buffer holds the sentinel (a NaN bit pattern no arithmetic
  // here produces) unless a producer has written it in THIS launch.  A producer stores its rows write-through and is done
  // -- no drain, no counter; the owner re-issues its L1-bypassing loads of a row until no word is the sentinel, adds in
  // split order, and writes the sentinel back, which re-arms the granule for the next launch (launches that share a work
  // buffer run one after the other on a stream).  One memory round trip each way instead of store-drain + barrier +
  // counter add + counter poll + barrier + loads (measured: 6-8 us per launch whatever the tile size).
  if (S > 1) {
    constexpr int ROWB = TN * 64 * 16;                 // bytes of one (wave, tile row): TN tiles x 64 lanes x 16 bytes
    constexpr int WGB = NW * TM * ROWB;                // bytes of one workgroup's partial tile
    float* wk = g.work + (second ? (long)nwg * (WGB / 4) : 0);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(wk, 0, (unsigned)(nwg * WGB), 0x00020000);
    const unsigned rowoff = (unsigned)(wave * TM * ROWB + lane * 16);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if ((wave * TM + tm) % S == z) continue;         // wave-uniform
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) g3_st4_wt(rw, (unsigned)(u * WGB) + rowoff + tm * ROWB + tn * 1024, acc[tm][tn]);
    }
    const floatx4 sent4 = __builtin_bit_cast(floatx4, g3_v4i{G3_SENTINEL, G3_SENTINEL, G3_SENTINEL, G3_SENTINEL});
    bool timed_out = false;
    // one owned row: pre = the splits before this one, in order; row = pre + row; then the splits behind it, in order
    // (a + b == b + a exactly, so this is the split-order sum with the own row in its place)
    auto fix_row = [&](floatx4 (&row)[TN], int tm) __attribute__((always_inline)) {
      floatx4 pre[TN];
      for (int zz = 0; zz < S; ++zz) {
        if (zz == z) {
          if (zz > 0) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) row[tn] = pre[tn] + row[tn];
          }
          continue;
        }
        floatx4 pv[TN];
        const unsigned peer = (unsigned)((t * S + zz) * WGB) + rowoff + tm * ROWB;
        for (int spin = 0;; ++spin) {
          int pending = 0;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            pv[tn] = g3_ld4_sc1(rw, peer + tn * 1024);
            const g3_v4i b = __builtin_bit_cast(g3_v4i, pv[tn]);
            pending |= (b.x == G3_SENTINEL) | (b.y == G3_SENTINEL) | (b.z == G3_SENTINEL) | (b.w == G3_SENTINEL);
          }
          if (!__any(pending)) break;                   // the wave moves on together
          if (spin > (1 << 20)) { timed_out = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) g3_st4_wt(rw, peer + tn * 1024, sent4);          // re-arm
        if (zz < z) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) pre[tn] = (zz == 0) ? pv[tn] : pre[tn] + pv[tn];
        } else {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) row[tn] = row[tn] + pv[tn];
        }
      }
    };
    // compile-time tile-row index (a runtime-indexed accumulator array would live in scratch memory)
    auto each_row = [&](auto self, auto I) __attribute__((always_inline)) -> void {
      if constexpr (decltype(I)::value < TM) {
        constexpr int tm = decltype(I)::value;
        if ((wave * TM + tm) % S == z) fix_row(acc[tm], tm);
        self(self, std::integral_constant<int, tm + 1>{});
      }
    };
    each_row(each_row, std::integral_constant<int, 0>{});
    if (timed_out && lane == 0)                         // a peer never delivered: flag it (C is invalid), never hang
      __hip_atomic_store(g.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- epilogue (with S > 1: only the rows this split owns)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if (S > 1 && (wave * TM + tm) % S != z) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = row_of(tm, 4 * q + reg);
      if (row >= g.M) continue;
      float* crow = Cg + (long)row * g.ldc;
      if constexpr (B_KC) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int col = col_of(tn);
          if (col < g.N) crow[col] = acc[tm][tn][reg] + bcol[tn];
        }
      } else {
#pragma unroll
        for (int gq = 0; gq < OB::G4; ++gq) {
          const int col = col_of(4 * gq);
          const floatx4 v = {acc[tm][4 * gq][reg] + bcol[4 * gq], acc[tm][4 * gq + 1][reg] + bcol[4 * gq + 1],
                             acc[tm][4 * gq + 2][reg] + bcol[4 * gq + 2], acc[tm][4 * gq + 3][reg] + bcol[4 * gq + 3]};
          if (col + 3 < g.N) {
            *reinterpret_cast<floatx4*>(crow + col) = v;
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (col + c < g.N) crow[col + c] = v[c];
          }
        }
#pragma unroll
        for (int tn = 4 * OB::G4; tn < TN; ++tn) {
          const int col = col_of(tn);
          if (col < g.N) crow[col] = acc[tm][tn][reg] + bcol[tn];
        }
      }
    }
  }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::NTHREADS) void gemm3_kernel(G3Args g) { Cfg::body(g, blockIdx.x, blockIdx.y); }

// Two independent products in ONE launch: workgroups [0, n1) run problem 1, the rest problem 2.  A launch boundary costs a
// GEMM its ramp-up and its drain (~4-6 us of a 25-50 us kernel); back to back in one grid the second problem's workgroups
// start on the CUs the first one's tail leaves idle.  (Workgroup programs of both configurations in one code object: the
// register and LDS footprint is the larger of the two.)
template <class C1, class C2>
__global__ __launch_bounds__(256) void gemm3_pair_kernel(G3Args g1, G3Args g2, int n1) {
  static_assert(C1::NTHREADS == 256 && C2::NTHREADS == 256, "pair launches use 4-wave configurations");
  const int b = blockIdx.x;
  if (b < n1) C1::body(g1, b % g1.nx, b / g1.nx);
  else C2::body(g2, (b - n1) % g2.nx, (b - n1) / g2.nx);
}

// derived launch parameters + argument checks of one problem for configuration Cfg; 0 or an error code
template <class Cfg>
int32_t g3_prepare(G3Args& g) {
  constexpr int BM = Cfg::BM_, BN = Cfg::BN_, BK = Cfg::BK_;
  g.nst = (g.K + BK - 1) / BK;
  if (g.splitk < 1) g.splitk = 1;
  if (g.splitk > g.nst) g.splitk = g.nst;
  g.nst_split = (g.nst + g.splitk - 1) / g.splitk;
  g.splitk = (g.nst + g.nst_split - 1) / g.nst_split;          // no empty split
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  g.err_word = 0;
  g.nx = tiles * g.splitk;
  if (g.splitk > 1) {
    if (g.work == nullptr || g.sync == nullptr) return TNT_BADARG(17);
    if ((long)tiles * g.splitk * BM * BN * 4 > (1L << 31)) return TNT_BADARG(17);
    if (tiles * g.splitk * (g.A2 ? 2 : 1) > 1024) return TNT_BADARG(16);   // the splits of a tile wait for each other: one round only
    if (g.colsum != nullptr) return TNT_BADARG(7);              // the column-sum rider needs the whole K in one workgroup
  }
  return 0;
}

template <class Cfg>
int32_t g3_launch_cfg(const G3Args& g0, hipStream_t s) {
  G3Args g = g0;
  if (int32_t rc = g3_prepare<Cfg>(g)) return rc;
  auto kern = gemm3_kernel<Cfg>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SHMEM) != hipSuccess)
      return TNT_BADARG(90);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.nx, g.A2 ? 2 : 1), dim3(Cfg::NTHREADS), Cfg::SHMEM, s, g);
  TNT_LAUNCH_CHECK();
  return 0;
}

template <class C1, class C2>
int32_t g3_launch_pair(const G3Args& a0, const G3Args& b0, hipStream_t s) {
  G3Args a = a0, b = b0;
  if (int32_t rc = g3_prepare<C1>(a)) return rc;
  if (int32_t rc = g3_prepare<C2>(b)) return rc;
  const int n1 = a.nx * (a.A2 ? 2 : 1), n2 = b.nx * (b.A2 ? 2 : 1);
  if ((a.splitk > 1 || b.splitk > 1) && n1 + n2 > 1024) return TNT_BADARG(16);
  constexpr int shmem = C1::SHMEM > C2::SHMEM ? C1::SHMEM : C2::SHMEM;
  auto kern = gemm3_pair_kernel<C1, C2>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, shmem) != hipSuccess)
      return TNT_BADARG(90);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(n1 + n2), dim3(256), shmem, s, a, b, n1);
  TNT_LAUNCH_CHECK();
  return 0;
}

template <int TM, int TN, int WGM, int WGN, int BK, int NS>
int32_t g3_layout(const G3Args& g, bool tA, bool tB, hipStream_t s) {
  if (!tA && !tB) return g3_launch_cfg<G3Cfg<true, false, TM, TN, WGM, WGN, BK, NS>>(g, s);
  if (tA && !tB) return g3_launch_cfg<G3Cfg<false, false, TM, TN, WGM, WGN, BK, NS>>(g, s);
  if (!tA && tB) return g3_launch_cfg<G3Cfg<true, true, TM, TN, WGM, WGN, BK, NS>>(g, s);
  return TNT_BADARG(12);
}

struct G3Tile { int bm, bn, lds; float eff; };
// Workgroup tile, LDS bytes and steady-state MFMA efficiency of configuration `tile` (the table of tnt_gemm3_f32).
// eff = measured fraction of the FP32-MFMA rate the K loop sustains (slope of time over K, tools/gemm3_scan.py with
// G3_KSCALE=1; profiles/r03_gemm3_scan.txt): big wave tiles amortise the LDS-DMA pieces and fragment reads best.
bool g3_tile(int tile, G3Tile& o) {
  switch (tile) {
    case 1: o = {160, 128, 3 * 36864 + 1024, 0.93f}; return true;
    case 2: o = {128, 160, 3 * 36864 + 1024, 0.89f}; return true;
    case 3: o = {128, 128, 3 * 32768 + 1024, 0.90f}; return true;
    case 4: o = {128, 80, 3 * 26624 + 1024, 0.88f}; return true;
    case 5: o = {64, 128, 3 * 24576 + 1024, 0.87f}; return true;
    case 6: o = {128, 64, 3 * 24576 + 1024, 0.85f}; return true;
    case 7: o = {64, 64, 3 * 16384 + 1024, 0.82f}; return true;
    case 8: o = {256, 80, 3 * 43008 + 1024, 0.90f}; return true;
    case 9: o = {64, 64, 3 * 32768 + 1024, 0.83f}; return true;
    case 10: o = {64, 128, 3 * 49152 + 1024, 0.88f}; return true;
    case 11: o = {128, 64, 3 * 49152 + 1024, 0.84f}; return true;
    default: return false;
  }
}
constexpr int G3_NTILES = 11;

// Estimated launch time in us of (tile, splitk) on a 256-CU gfx950: fixed cost (launch, first stage, epilogue stores) +
// the busiest CU's MFMA time / eff + the in-launch split-K exchange (a cross-workgroup hand-off costs ~6 us on this chip
// whatever its size, even with the data as its own flag; MI355X_MICROARCH.md price list, "splitk-seam").  Calibrated against
// tools/gemm3_scan.py.
double g3_cost(int M, int N, int K, int batch, int tile, int splitk, bool* ok) {
  G3Tile tl;
  *ok = false;
  if (!g3_tile(tile, tl)) return 1e30;
  const long tiles = (long)((M + tl.bm - 1) / tl.bm) * ((N + tl.bn - 1) / tl.bn);
  const long units = tiles * splitk * batch;
  if (splitk > 1 && units > 256) return 1e30;                 // the splits of a tile wait for each other: one round only
  if (units > 4096) return 1e30;
  const int bk = tile >= 9 ? 64 : 32;
  const int nst = (K + bk - 1) / bk, per = (nst + splitk - 1) / splitk;
  if (splitk > 1 && (per < 4 || (nst + per - 1) / per != splitk)) return 1e30;
  const int resident = 163840 / tl.lds < 1 ? 1 : 163840 / tl.lds;
  const long per_cu = (units + 255) / 256;                    // workgroups the busiest CU runs
  const double t_wg = (double)tl.bm * tl.bn * per * bk * 2.0 / 256.0 / 2400.0;       // us at the full MFMA rate
  double eff = tl.eff;
  if (per_cu > 1 && resident > 1) eff = eff * 1.04 > 0.95 ? 0.95 : eff * 1.04;      // co-resident workgroups hide each other's stalls
  // fixed: launch + the first stages' latency (a 64-deep stage takes longer to land) + the epilogue's stores
  const double fixed = 4.0 + (bk == 64 ? 0.8 : 0.0) + (double)M * N * batch * 4.0 / 4.0e6;
  *ok = true;
  // exchange: a fixed hand-off latency + the rows a workgroup sends and fetches ((S-1)/S of its tile each way, ~20 KB/us)
  const double xch = splitk > 1 ? 4.0 + 0.4 * splitk + tl.bm * tl.bn * 4.0e-3 * (splitk - 1) / splitk / 20.0 : 0.0;
  return fixed + per_cu * t_wg / eff + xch;
}

}  // namespace

extern "C" int32_t tnt_gemm3_work_floats(int32_t M, int32_t N, int32_t tile, int32_t splitk, int32_t batch) {
  G3Tile tl;
  if (!g3_tile(tile, tl) || splitk <= 1) return 0;
  const long tiles = (long)((M + tl.bm - 1) / tl.bm) * ((N + tl.bn - 1) / tl.bn);
  return (int32_t)(tiles * splitk * tl.bm * tl.bn * (batch > 1 ? 2 : 1));
}
extern "C" int32_t tnt_gemm3_sync_words(int32_t M, int32_t N, int32_t tile, int32_t batch) {
  (void)M; (void)N; (void)tile; (void)batch;
  return 1;
}

namespace {
__global__ __launch_bounds__(256) void g3_arm_kernel(g3_v4i* w, long n4) {
  const g3_v4i v{G3_SENTINEL, G3_SENTINEL, G3_SENTINEL, G3_SENTINEL};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) w[i] = v;
}
}  // namespace

extern "C" int32_t tnt_gemm3_work_arm(float* work, int64_t floats, void* stream) {
  if (work == nullptr || floats <= 0 || floats % 4 || !tnt_aligned16(work)) return TNT_BADARG(1);
  const long n4 = floats / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(g3_arm_kernel, dim3((unsigned)blocks), dim3(256), 0, tnt_stream(stream), reinterpret_cast<g3_v4i*>(work), n4);
  TNT_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t tnt_gemm3_plan(int32_t M, int32_t N, int32_t K, int32_t transA, int32_t transB, int32_t batch,
                                  int32_t allow_split, int32_t* tile, int32_t* splitk) {
  if (M <= 0 || N <= 0 || K <= 0 || !tile || !splitk) return TNT_BADARG(1);
  if (transA && transB) return TNT_BADARG(5);
  double best = 1e30;
  *tile = 0; *splitk = 1;
  const int ss[] = {1, 2, 3, 4, 6, 8};
  for (int t = 1; t <= G3_NTILES; ++t)
    for (int s : ss) {
      if (s > 1 && !allow_split) continue;
      bool ok;
      const double c = g3_cost(M, N, K, batch > 1 ? 2 : 1, t, s, &ok);
      if (ok && c < best) { best = c; *tile = t; *splitk = s; }
    }
  return *tile ? 0 : TNT_BADARG(2);
}

namespace {
int32_t g3_fill(G3Args& g, const float* A, const float* B, float* C, const float* bias, float* colsum, const float* A2, float* C2,
                int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t transA, int32_t transB,
                int32_t splitk, float* work, uint32_t* sync) {
  if (M <= 0 || N <= 0 || K <= 0) return TNT_BADARG(8);
  if (transA && transB) return TNT_BADARG(15);
  if (!tnt_aligned16(A) || !tnt_aligned16(B) || !tnt_aligned16(C) || lda % 4 || ldb % 4 || ldc % 4) return TNT_BADARG(1);
  if ((A2 == nullptr) != (C2 == nullptr)) return TNT_BADARG(6);
  if (A2 && (!tnt_aligned16(A2) || !tnt_aligned16(C2))) return TNT_BADARG(6);
  if (colsum != nullptr && !(transA && !transB)) return TNT_BADARG(5);
  if (work && !tnt_aligned16(work)) return TNT_BADARG(15);
  const long rowsA = transA ? K : M, rowsB = transB ? N : K;
  if (rowsA * lda >= (1L << 29) || rowsB * ldb >= (1L << 29)) return TNT_BADARG(2);      // 32-bit buffer offsets
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.work = work; g.sync = sync;
  g.A2 = A2; g.C2 = C2; g.colsum = colsum;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.nst = 0; g.splitk = splitk;
  return 0;
}
}  // namespace

extern "C" int32_t tnt_gemm3_f32(const float* A, const float* B, float* C, const float* bias, float* colsum,
                                 const float* A2, float* C2, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
                                 int32_t ldc, int32_t transA, int32_t transB, int32_t tile, int32_t splitk, float* work,
                                 uint32_t* sync, void* stream) {
  G3Args g;
  if (int32_t rc = g3_fill(g, A, B, C, bias, colsum, A2, C2, M, N, K, lda, ldb, ldc, transA, transB, splitk, work, sync)) return rc;
  hipStream_t s = tnt_stream(stream);
  const bool tA = transA != 0, tB = transB != 0;
  switch (tile) {
    //                  TM TN WGM WGN BK NS
    case 1: return g3_layout<5, 4, 2, 2, 32, 3>(g, tA, tB, s);       // 160 x 128, 4 waves of 80 x 64
    case 2: return g3_layout<4, 5, 2, 2, 32, 3>(g, tA, tB, s);       // 128 x 160, 4 waves of 64 x 80
    case 3: return g3_layout<4, 4, 2, 2, 32, 3>(g, tA, tB, s);       // 128 x 128, 4 waves of 64 x 64
    case 4: return g3_layout<2, 5, 4, 1, 32, 3>(g, tA, tB, s);       // 128 x  80, 4 waves of 32 x 80
    case 5: return g3_layout<2, 4, 2, 2, 32, 3>(g, tA, tB, s);       //  64 x 128, 4 waves of 32 x 64
    case 6: return g3_layout<4, 2, 2, 2, 32, 3>(g, tA, tB, s);       // 128 x  64, 4 waves of 64 x 32
    case 7: return g3_layout<2, 2, 2, 2, 32, 3>(g, tA, tB, s);       //  64 x  64, 4 waves of 32 x 32
    case 8: return g3_layout<4, 5, 4, 1, 32, 3>(g, tA, tB, s);       // 256 x  80, 4 waves of 64 x 80
    case 9: return g3_layout<2, 2, 2, 2, 64, 3>(g, tA, tB, s);       //  64 x  64, 64-deep stages
    case 10: return g3_layout<2, 4, 2, 2, 64, 3>(g, tA, tB, s);      //  64 x 128
    case 11: return g3_layout<4, 2, 2, 2, 64, 3>(g, tA, tB, s);      // 128 x  64
    default: return TNT_BADARG(13);
  }
}

// Pair launches exist for the combinations a training step has back to back: a TN product (a kernel gradient, with its
// riders) on tile 4, 5 or 7 followed by an independent NT product (an input gradient) on tile 7 or 5, and two independent TN
// products (two small kernel gradients of the same step) on tiles 5 / 7.
extern "C" int32_t tnt_gemm3_pair_supported(int32_t tile1, int32_t transA1, int32_t transB1, int32_t tile2, int32_t transA2,
                                            int32_t transB2) {
  const bool tn1 = transA1 && !transB1, tn2 = transA2 && !transB2, nt2 = !transA2 && transB2;
  if (tn1 && nt2) return ((tile1 == 4 || tile1 == 5 || tile1 == 7) && (tile2 == 7 || tile2 == 5)) ? 1 : 0;
  if (tn1 && tn2) return ((tile1 == 5 || tile1 == 7) && (tile2 == 5 || tile2 == 7)) ? 1 : 0;
  return 0;
}

extern "C" int32_t tnt_gemm3_pair_f32(const tnt_gemm3_desc* p, const tnt_gemm3_desc* q, void* stream) {
  if (p == nullptr || q == nullptr) return TNT_BADARG(1);
  if (!tnt_gemm3_pair_supported(p->tile, p->transA, p->transB, q->tile, q->transA, q->transB)) return TNT_BADARG(13);
  G3Args a, b;
  if (int32_t rc = g3_fill(a, p->A, p->B, p->C, p->bias, p->colsum, p->A2, p->C2, p->M, p->N, p->K, p->lda, p->ldb, p->ldc, p->transA,
                           p->transB, p->splitk, p->work, p->sync)) return rc;
  if (int32_t rc = g3_fill(b, q->A, q->B, q->C, q->bias, q->colsum, q->A2, q->C2, q->M, q->N, q->K, q->lda, q->ldb, q->ldc, q->transA,
                           q->transB, q->splitk, q->work, q->sync)) return rc;
  if (a.splitk > 1 && b.splitk > 1 && a.work == b.work) return TNT_BADARG(2);       // concurrent exchanges need their own space
  hipStream_t s = tnt_stream(stream);
  using TN4 = G3Cfg<false, false, 2, 5, 4, 1, 32, 3>;
  using TN5 = G3Cfg<false, false, 2, 4, 2, 2, 32, 3>;
  using NT7 = G3Cfg<true, true, 2, 2, 2, 2, 32, 3>;
  using NT5 = G3Cfg<true, true, 2, 4, 2, 2, 32, 3>;
  using TN7 = G3Cfg<false, false, 2, 2, 2, 2, 32, 3>;
  if (q->transA) {                                           // TN + TN
    if (p->tile == 5 && q->tile == 5) return g3_launch_pair<TN5, TN5>(a, b, s);
    if (p->tile == 5 && q->tile == 7) return g3_launch_pair<TN5, TN7>(a, b, s);
    if (p->tile == 7 && q->tile == 5) return g3_launch_pair<TN7, TN5>(a, b, s);
    return g3_launch_pair<TN7, TN7>(a, b, s);
  }
  if (p->tile == 4 && q->tile == 7) return g3_launch_pair<TN4, NT7>(a, b, s);
  if (p->tile == 4 && q->tile == 5) return g3_launch_pair<TN4, NT5>(a, b, s);
  if (p->tile == 5 && q->tile == 7) return g3_launch_pair<TN5, NT7>(a, b, s);
  if (p->tile == 5 && q->tile == 5) return g3_launch_pair<TN5, NT5>(a, b, s);
  if (p->tile == 7 && q->tile == 7) return g3_launch_pair<TN7, NT7>(a, b, s);
  return g3_launch_pair<TN7, NT5>(a, b, s);
}
