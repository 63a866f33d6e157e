// frcnn_sync.h -- the agent-scope publish / acquire primitives used where workgroups of one launch hand data
// to each other (split tiles of the stream-K convolution).  gfx950: per-CU L1s are never refreshed by other
// CUs' stores and the 8 XCD L2s are not coherent, so visibility comes ONLY from an agent-scope release on the
// producer and an agent-scope acquire on the consumer (cdna_hip_programming.md section 6, Guideline 16):
//   producer: plain stores -> every wave drains (s_waitcnt vmcnt(0)) -> __syncthreads() -> ONE lane:
//             release fence -> drain again (the compiler may drop the wait behind buffer_wbl2) -> relaxed atomic
//   consumer: the lane that drew the last ticket: acquire fence -> __syncthreads() -> plain loads
// (the test emulator shadows this header with no-op versions: it runs workgroups one after the other)
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void frcnn_drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void frcnn_release_agent() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void frcnn_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ int frcnn_ticket(int *counter) {
    return __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void frcnn_counter_reset(int *counter) {
    __hip_atomic_exchange(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // same path as the tickets
}
