// XCD-local group synchronisation of the persistent chain kernels (lstm.hip: lstm_seq_fwd_kernel, lstm_seq_bwd_kernel).
//
// A "group" is the 32 workgroups that a 256-workgroup, one-per-CU launch places on one XCD (tnt_lstm_seq_supported
// checks that census once per process).  A workgroup reads the XCD it actually runs on from HW_REG_XCC_ID and works on
// THAT XCD's row block, so the 32 members of a group share one L2 by construction, whatever the dispatcher did.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, specialised to one XCD):
//   producer : plain stores (they stay in the XCD's L2) -> EVERY storing wave drains them (s_waitcnt vmcnt(0)) ->
//              workgroup barrier -> ONE lane raises the workgroup's flag word (agent-scope relaxed store);
//   consumer : ONE wave polls the group's 32 flag words with agent-scope relaxed loads (sc1: served by L2, never by
//              the CU's L1) -> workgroup barrier -> EVERY load of handed-off bytes is an sc1 load (tnt_ld4_l2 below):
//              a CU's vector L1 is never refreshed by another CU's stores, so a plain load could return a line the L1
//              kept from an earlier step or launch.  No L1 invalidate (buffer_inv sc1, ~1.5 us per barrier) is needed
//              because no handed-off byte is ever read through L1.
//
// State (uint32 words, zeroed by the owner before first use and after an error -- never by a graph node):
//   [8][64] flags      32 used per XCD; the 32 flags of a group share one 128-byte line
//   [8][64] control    word 0 = ticket counter, 1 = exit counter, 2 = launch epoch of that XCD
//   [1]     error word 0 = fine, 1 = a barrier timed out, 2 = ticket out of range (a launch that did not place exactly
//                      32 workgroups on the XCD, or state that was not reset after an error)
// A launch takes tickets 0..31 (= the workgroup's unit block); the LAST workgroup of a group to leave the kernel
// resets the ticket and exit counters and advances the epoch, so the next launch -- a hipGraph replay included,
// whose kernel arguments are frozen -- starts from a known state without any host or memset involvement.  Barrier
// targets are absolute: epoch * 64 + k for the k-th barrier of the launch (k <= 63), so a stale flag from an earlier
// launch can never satisfy a later one.  A ticket >= 32 sets the error word and the workgroup leaves at once: the
// failure is loud and sticky (every later launch fails the same way) until the host zeroes the state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr unsigned TNT_SEQ_SPIN_LIMIT = 1u << 21;
constexpr int TNT_SEQ_CTRL = 8 * 64;             // first control word
constexpr int TNT_SEQ_ERR = 2 * 8 * 64;          // index of the error word
constexpr int TNT_SEQ_SYNC_WORDS = TNT_SEQ_ERR + 1;
constexpr int TNT_SEQ_MAX_BARRIERS = 63;         // per launch (targets are epoch * 64 + k)

typedef float tnt_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned tnt_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}

// 16-byte load that bypasses the CU's vector L1 (buffer_load_dwordx4 ... sc1): for bytes another workgroup of the
// same XCD stored earlier in this launch.  `rsrc` covers the whole buffer; `byte_off` < 4 GiB.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tnt_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 tnt_ld4_l2(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  const tnt_f4 v = __builtin_bit_cast(tnt_f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, /*sc1*/ 16));
  return make_float4(v.x, v.y, v.z, v.w);
}

struct TntSeqSlot {
  int ub;            // this workgroup's slot (unit block) in its group, 0..31; -1 = invalid (error word set)
  unsigned epoch;    // launch epoch of the group
};

// Called by ALL threads at kernel start.  `lds_tmp` = two uint32 words of LDS.
__device__ __forceinline__ TntSeqSlot tnt_seq_enter(unsigned* sync, unsigned xcc, unsigned* lds_tmp) {
  unsigned* ctrl = sync + TNT_SEQ_CTRL + xcc * 64;
  if (threadIdx.x == 0) {
    lds_tmp[0] = atomicAdd(ctrl + 0, 1u);
    lds_tmp[1] = __hip_atomic_load(ctrl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  TntSeqSlot s;
  const unsigned n = lds_tmp[0];
  s.epoch = __builtin_amdgcn_readfirstlane(lds_tmp[1]);
  s.ub = n < 32u ? __builtin_amdgcn_readfirstlane((int)n) : -1;
  if (n >= 32u && threadIdx.x == 0) __hip_atomic_store(sync + TNT_SEQ_ERR, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return s;
}

// Called by ALL threads of a workgroup with a valid slot when it is done (also after a barrier timeout).
// `guard_out` (nullable): one float that carries the error code to the host along with the step's metrics.
__device__ __forceinline__ void tnt_seq_leave(unsigned* sync, unsigned xcc, float* guard_out) {
  unsigned* ctrl = sync + TNT_SEQ_CTRL + xcc * 64;
  if (threadIdx.x == 0 && atomicAdd(ctrl + 1, 1u) == 31u) {
    const unsigned e = __hip_atomic_load(sync + TNT_SEQ_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e != 0u && guard_out) guard_out[0] = (float)e;
    __hip_atomic_store(ctrl + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ctrl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicAdd(ctrl + 2, 1u);
  }
}

__device__ __forceinline__ unsigned tnt_seq_target(unsigned epoch, int k) { return epoch * 64u + (unsigned)k; }

// Called by ALL threads of the workgroup after their stores of the phase.  `flags` = this XCD's flag line, `ub` = this
// workgroup's slot, `target` = tnt_seq_target(epoch, k) for the k-th barrier of the launch (k = 1, 2, ...).
__device__ __forceinline__ void tnt_seq_group_barrier(unsigned* flags, int ub, unsigned target, unsigned* err) {
  // every storing wave drains its stores: they are in the XCD's L2 once vmcnt reaches 0 (inline asm: the compiler
  // must not drop or move the wait, MI355X_MICROARCH.md "Compiler hazard")
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
#ifdef TNT_SEQ_FLAG_PLAIN     // experiment: plain store (stays in the XCD's L2) instead of the write-through sc1 store
    if (lane == 0) *reinterpret_cast<volatile unsigned*>(flags + ub) = target;
#else
    if (lane == 0) __hip_atomic_store(flags + ub, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    unsigned spins = 0;
    for (;;) {
      const unsigned v = lane < 32 ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
      if (__all((int)(v - target) >= 0)) break;
      if (++spins > TNT_SEQ_SPIN_LIMIT) {        // never hang the grid: flag the error and let every wave leave
        if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
  }
  __syncthreads();       // the other waves' sc1 loads of the handed-off bytes are issued after this barrier
}
