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
//  * ROLE-SPECIALISED WAVES: the workgroup is WGM x WGN consumer waves plus ONE loader wave.  The loader issues every
//    LDS-DMA piece of a stage (a piece costs its issuing wave ~60-100 cycles in which it can issue nothing else; inside a
//    consumer's stream that showed up as 10-20 % idle MFMA time) and keeps NS - 1 stages in flight; the consumers'
//    streams are ds_read_b128 + MFMA only.  Consumers and loader meet at the one barrier per stage: the consumers
//    arrive when they have read the last fragment of stage s, the loader when its pieces of stage s + 1 have landed.
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
//  * Work decomposition: one workgroup per output tile, tiles dealt XCD-contiguously; `splitk` > 1 splits K over
//    workgroups that publish their partial tiles (write-through) and then EACH reduce 1/splitk of the tile in split
//    order (no extra launch, bitwise reproducible) -- see the fix-up section.
#include "tnt_common.h"

namespace {

struct G3Args {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K, lda, ldb, ldc;
  int nst;                    // K stages per workgroup (a multiple of 2)
};

// One LDS-DMA piece: 64 lanes x 16 bytes from the buffer `rsrc` + voff (per lane; out-of-range lanes deliver zeros) to
// the 1 KB at LDS byte address `dst` (wave-uniform), lane-linear.  Inline asm on purpose: hipcc's waitcnt pass treats an
// LDS-DMA it can see as a store that any later ds_read may alias and puts `s_waitcnt vmcnt(0)` in front of the next
// fragment read -- which drains the prefetch ring every stage.  Hidden in asm the pieces are counted by hand
// (s_waitcnt vmcnt(N) before the barrier that publishes a stage); the compiler's own waits can only over-wait.
// M0 carries the LDS address and is written in the same statement that uses it.
typedef int g3_v4i __attribute__((ext_vector_type(4)));
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
//   T    : 16-wide MFMA tiles per wave along that dimension
constexpr int g3_gcd(int a, int b) { return b == 0 ? a : g3_gcd(b, a % b); }
template <bool KC, int EXT, int T, int BK>
struct G3Op {
  static constexpr int CH = BK / 4;
  static constexpr int BYTES = EXT * BK * 4;
  static constexpr int NI = BYTES / 1024;                 // LDS-DMA pieces (wave-instructions) per stage
  static constexpr int G4 = KC ? 0 : T / 4, G2 = KC ? 0 : (T % 4) / 2, G1 = KC ? 0 : T % 2;
  // Piece j loads image slots 64 j .. 64 j + 63 (a slot = 16 bytes).  The lane -> (row, chunk) pattern of a piece repeats
  // every P pieces = RP image rows, so a piece's per-lane source offset is  base[j % P] + (j / P) * RP * ld * 4.
  static constexpr int SPR = KC ? CH : EXT / 4;           // slots per image row
  static constexpr int P = KC ? (CH >= 4 ? CH / 4 : 1) : SPR / g3_gcd(64, SPR);
  static constexpr int RP = 64 * P / SPR;
  static_assert(BYTES % 1024 == 0, "stage image must be whole 1 KB LDS-DMA pieces");
  static_assert(!KC || (EXT * CH) % 64 == 0, "KC image: whole pieces of 64 chunks");
  static_assert((64 * P) % SPR == 0, "a period is whole image rows");

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

template <bool A_KC, bool B_KC, int TM, int TN, int WGM, int WGN, int BK, int NS>
__global__ __launch_bounds__(64 * (WGM * WGN + 1)) void gemm3_kernel(G3Args g) {
  constexpr int NW = WGM * WGN;                       // consumer waves; wave NW is the loader
  constexpr int WM = 16 * TM, WN = 16 * TN, BM = WM * WGM, BN = WN * WGN;
  using OA = G3Op<A_KC, BM, TM, BK>;
  using OB = G3Op<B_KC, BN, TN, BK>;
  constexpr int STAGE = OA::BYTES + OB::BYTES;
  constexpr int KB = BK / 16;                         // 16-deep k blocks per stage
  constexpr bool KMASK = A_KC && B_KC;                // NT: the k tail of A must be zeroed by hand (B's then meets zeros)
  constexpr int NP = OA::NI + OB::NI;                 // pieces per stage
  static_assert(KB % 2 == 0, "register buffer parity must repeat per stage");
  extern __shared__ __attribute__((aligned(1024))) unsigned char g3_lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int MT = (g.M + BM - 1) / BM, NTl = (g.N + BN - 1) / BN;
  const int nwg = MT * NTl, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;       // XCD-contiguous tile order (bijective for any nwg)
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (t % MT) * BM, n0 = (t / MT) * BN;

  if (wave == NW) {
    // =============================================================================================== loader wave
    // (the extern array is the kernel's only LDS object: byte offsets inside it are LDS addresses)
    const int rowsA = A_KC ? g.M : g.K, rowsB = B_KC ? g.N : g.K;
    const g3_v4i rsA = g3_rsrc(g.A, (unsigned)(rowsA * g.lda * 4));
    const g3_v4i rsB = g3_rsrc(g.B, (unsigned)(rowsB * g.ldb * 4));
    int baseA[OA::P], baseB[OB::P], kcA[KMASK ? OA::P : 1];
#pragma unroll
    for (int v = 0; v < OA::P; ++v) {
      int kc4;
      baseA[v] = OA::src_off(v, lane, m0, g.lda, kc4);
      if constexpr (KMASK) kcA[v] = kc4;
    }
#pragma unroll
    for (int v = 0; v < OB::P; ++v) {
      int kc4;
      baseB[v] = OB::src_off(v, lane, n0, g.ldb, kc4);
    }
    const int k4 = (g.K + 3) & ~3;
    // every piece of stage s -> the LDS buffer at byte offset boff.  Stages past the last one are issued too (vmcnt counts
    // pieces, so every stage must have the same number) with out-of-range offsets: they fetch nothing.
    auto issue = [&](int s, int boff) __attribute__((always_inline)) {
      const int k0 = s * BK;
      const bool live = s < g.nst;
      const int sa = A_KC ? k0 * 4 : k0 * g.lda * 4, sb = B_KC ? k0 * 4 : k0 * g.ldb * 4;
#pragma unroll
      for (int j = 0; j < OA::NI; ++j) {
        int vo = baseA[j % OA::P] + (j / OA::P) * OA::RP * g.lda * 4 + sa;
        if constexpr (KMASK) vo = (k0 + kcA[j % OA::P] < k4) ? vo : 0x7fffffff;
        vo = live ? vo : 0x7fffffff;
        g3_dma16(rsA, (unsigned)(boff + j * 1024), vo);
      }
#pragma unroll
      for (int j = 0; j < OB::NI; ++j) {
        int vo = baseB[j % OB::P] + (j / OB::P) * OB::RP * g.ldb * 4 + sb;
        vo = live ? vo : 0x7fffffff;
        g3_dma16(rsB, (unsigned)(boff + OA::BYTES + j * 1024), vo);
      }
    };
    constexpr int W0 = (NS - 1) * NP > 63 ? 63 : (NS - 1) * NP;       // vmcnt is a 6-bit counter: a larger count
    constexpr int W1 = (NS - 2) * NP > 63 ? 63 : (NS - 2) * NP;       // over-waits a little, never under-waits
#pragma unroll
    for (int s = 0; s < NS; ++s) issue(s, s * STAGE);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W0) : "memory");         // stage 0 landed
    __builtin_amdgcn_s_barrier();                                      // barrier 0
    int bcur = 0;
    for (int s = 0; s < g.nst; ++s) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1) : "memory");       // stage s + 1 landed
      __builtin_amdgcn_s_barrier();                                    // barrier s + 1: the consumers are done with stage s
      issue(s + NS, bcur);
      bcur = (bcur + STAGE == NS * STAGE) ? 0 : bcur + STAGE;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ================================================================================================= consumer waves
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 15, q = lane >> 4;
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
  // the MFMAs of a block in the order (k-step e, tile row, tile column)
  auto mma = [&](int rb) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb][tm][e], bv[rb][tn][e], acc[tm][tn], 0, 0, 0);
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

  // ---- main loop.  Two k blocks live in registers (register buffers 0 / 1; a stage has an even number of blocks, so
  // every stage starts on register buffer 0).  Block b of a stage multiplies register buffer b & 1 while the fragments of
  // the next block are read; the LAST block of a stage first retires this wave's reads of the stage (lgkmcnt(0)), meets the
  // other waves and the loader (whose arrival says: stage s + 1 is in LDS) and reads the first fragments of stage s + 1.
  // Instruction order is pinned: hipcc otherwise sinks every ds_read down to its first use ("reads; lgkmcnt(0); MFMAs",
  // ~150 exposed cycles per block with one wave per SIMD); here the reads of the next block go one per MFMA behind the
  // first MFMAs of this block.
  constexpr int NRD = (A_KC ? TM : 4 * (OA::G4 + OA::G2 + OA::G1)) + (B_KC ? TN : 4 * (OB::G4 + OB::G2 + OB::G1));
  constexpr int NMM = 4 * TM * TN;
  static_assert(NRD <= NMM, "not enough MFMAs in a block to carry its reads");
  __builtin_amdgcn_s_barrier();                                        // barrier 0: stage 0 is in LDS
  asm volatile("" ::: "memory");
  fetch(0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  int bcur = 0;
  for (int s = 0; s < g.nst; ++s) {
    const int bnext = (bcur + STAGE == NS * STAGE) ? 0 : bcur + STAGE;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int rb = kb & 1;
      if (kb + 1 < KB) {
        fetch(bcur, kb + 1, rb ^ 1);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // barrier s + 1
        __builtin_amdgcn_sched_barrier(0);
        fetch(bnext, 0, rb ^ 1);
      }
      mma(rb);
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NMM - NRD, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    bcur = bnext;
  }

  // ---- epilogue
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = row_of(tm, 4 * q + reg);
      if (row >= g.M) continue;
      float* crow = g.C + (long)row * g.ldc;
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

template <bool A_KC, bool B_KC, int TM, int TN, int WGM, int WGN, int BK, int NS>
int32_t g3_launch(const G3Args& g0, hipStream_t s) {
  constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
  constexpr int shmem = NS * (BM + BN) * BK * 4;
  static_assert(shmem <= 160 * 1024, "LDS ring does not fit");
  G3Args g = g0;
  g.nst = (g.K + BK - 1) / BK;
  auto kern = gemm3_kernel<A_KC, B_KC, TM, TN, WGM, WGN, BK, NS>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, shmem) != hipSuccess)
      return TNT_BADARG(90);
    attr_set = true;
  }
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * (WGM * WGN + 1)), shmem, s, g);
  TNT_LAUNCH_CHECK();
  return 0;
}

template <int TM, int TN, int WGM, int WGN, int BK, int NS>
int32_t g3_layout(const G3Args& g, bool tA, bool tB, hipStream_t s) {
  if (!tA && !tB) return g3_launch<true, false, TM, TN, WGM, WGN, BK, NS>(g, s);
  if (tA && !tB) return g3_launch<false, false, TM, TN, WGM, WGN, BK, NS>(g, s);
  if (!tA && tB) return g3_launch<true, true, TM, TN, WGM, WGN, BK, NS>(g, s);
  return TNT_BADARG(12);
}

}  // namespace

extern "C" int32_t tnt_gemm3_f32(const float* A, const float* B, float* C, const float* bias, int32_t M, int32_t N,
                                 int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t transA, int32_t transB,
                                 int32_t cfg, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return TNT_BADARG(5);
  if (transA && transB) return TNT_BADARG(12);
  if (!tnt_aligned16(A) || !tnt_aligned16(B) || !tnt_aligned16(C) || lda % 4 || ldb % 4 || ldc % 4) return TNT_BADARG(1);
  const long rowsA = transA ? K : M, rowsB = transB ? N : K;
  if (rowsA * lda >= (1L << 29) || rowsB * ldb >= (1L << 29)) return TNT_BADARG(2);      // 32-bit buffer offsets
  G3Args g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.nst = 0;
  hipStream_t s = tnt_stream(stream);
  const bool tA = transA != 0, tB = transB != 0;
  switch (cfg) {
    //                  TM TN WGM WGN BK NS
    case 1: return g3_layout<5, 4, 2, 2, 32, 2>(g, tA, tB, s);       // 160 x 128, 4 waves of 80 x 64
    case 2: return g3_layout<5, 4, 2, 2, 32, 3>(g, tA, tB, s);
    case 3: return g3_layout<5, 4, 2, 2, 32, 4>(g, tA, tB, s);
    case 4: return g3_layout<2, 5, 4, 1, 32, 2>(g, tA, tB, s);       // 128 x  80, 4 waves of 32 x 80
    case 5: return g3_layout<2, 5, 4, 1, 32, 3>(g, tA, tB, s);
    case 6: return g3_layout<2, 5, 4, 1, 32, 4>(g, tA, tB, s);
    case 7: return g3_layout<2, 4, 2, 2, 32, 2>(g, tA, tB, s);       //  64 x 128, 4 waves of 32 x 64
    case 8: return g3_layout<2, 4, 2, 2, 32, 4>(g, tA, tB, s);
    case 9: return g3_layout<2, 2, 2, 2, 32, 2>(g, tA, tB, s);       //  64 x  64, 4 waves of 32 x 32
    case 10: return g3_layout<2, 2, 2, 2, 32, 4>(g, tA, tB, s);
    case 11: return g3_layout<4, 4, 2, 2, 32, 3>(g, tA, tB, s);      // 128 x 128, 4 waves of 64 x 64
    case 12: return g3_layout<4, 2, 2, 2, 32, 4>(g, tA, tB, s);      // 128 x  64, 4 waves of 64 x 32
    case 13: return g3_layout<4, 5, 2, 2, 32, 3>(g, tA, tB, s);      // 128 x 160, 4 waves of 64 x 80
    case 14: return g3_layout<4, 4, 4, 2, 32, 2>(g, tA, tB, s);      // 256 x 128, 8 waves of 64 x 64
    default: return TNT_BADARG(13);
  }
}
