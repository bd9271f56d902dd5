// XCD-local group barrier of the persistent chain kernels (lstm.hip: lstm_seq_fwd_kernel; chain.hip: attention + LSTM).
// A "group" is the 32 workgroups a 256-workgroup launch places on one XCD (tnt_lstm_seq_supported checks that census);
// they share that XCD's L2, so publishing data needs no cache maintenance beyond draining the stores.
//   sync buffer (uint32): [8][64] flags (32 used per XCD; the 32 flags of a group share one 128-byte line),
//                         [8][64] tickets, then the error word.  Zero-initialised ONCE by the owner and never reset:
//   tickets count modulo 32, flags only grow -- a launch counts its barriers from the value its own flag had at start.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr unsigned TNT_SEQ_SPIN_LIMIT = 1u << 21;
constexpr int TNT_SEQ_ERR = 2 * 8 * 64;          // index of the error word
constexpr int TNT_SEQ_SYNC_WORDS = TNT_SEQ_ERR + 1;

__device__ __forceinline__ unsigned tnt_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}

// Called by ALL threads of the workgroup after their stores of the phase.  `flags` = this XCD's flag line, `ub` = this
// workgroup's slot, `target` = base + (number of barriers passed so far in this launch, including this one).
__device__ __forceinline__ void tnt_seq_group_barrier(unsigned* flags, int ub, unsigned target, unsigned* err) {
  // a workgroup-scope release fence alone does not drain vmcnt (stores are already ordered within a CU): do it explicitly
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (lane == 0) __hip_atomic_store(flags + ub, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
