// frcnn_buffer.h -- bounds-checked global loads through a gfx950 buffer descriptor (SRD).
// A raw buffer load whose byte offset is >= the descriptor's size returns 0 instead of faulting, so halo
// padding, ragged channel counts and "this lane has nothing to fetch" (offset = kBufOob) need no branch and no
// per-element predicate: the staging code of the implicit-GEMM kernels is straight-line.  The descriptor is
// built from kernel arguments only (wave-uniform by construction -> it lives in SGPRs, no waterfall loop).
// Offsets are 32-bit byte offsets: a tensor behind a descriptor must be smaller than 2 GiB.
// (the test emulator shadows this header with a host version)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __amdgpu_buffer_rsrc_t frcnn_buf_t;
typedef unsigned frcnn_u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kBufOob = 0x80000000u;      // any offset with this bit set is out of range

__device__ __forceinline__ frcnn_buf_t frcnn_make_buf(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float frcnn_buf_load_f32(frcnn_buf_t b, uint32_t byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b, (int)byte_off, 0, 0));
}
// 4-byte load at (per-lane offset, range-checked) + (wave-uniform scalar offset, added after the check): the scalar rides in the instruction
__device__ __forceinline__ float frcnn_buf_load_f32_soff(frcnn_buf_t b, uint32_t byte_off, uint32_t soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b, (int)byte_off, (int)soff, 0));
}
__device__ __forceinline__ float4 frcnn_buf_load_f32x4(frcnn_buf_t b, uint32_t byte_off) {
    const frcnn_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// 16-byte load at (per-lane offset, range-checked) + (wave-uniform scalar offset, added after the check)
__device__ __forceinline__ float4 frcnn_buf_load_f32x4_soff(frcnn_buf_t b, uint32_t byte_off, uint32_t soff) {
    const frcnn_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b, (int)byte_off, (int)soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Write-through (sc1) 16-byte store: the bytes leave this XCD's L2 for memory at once, so another workgroup -- on any
// XCD -- that acquires AFTER the store has drained (s_waitcnt vmcnt(0)) reads them without the producer running an
// agent-scope release fence (buffer_wbl2 writes back EVERY dirty line of the XCD's L2: ~3x the cost for tens of KB per
// workgroup; cdna_hip_programming.md Guideline 16 form R1, MI355X_MICROARCH.md row "publish-large").
__device__ __forceinline__ void frcnn_buf_store_f32x4_wt(frcnn_buf_t b, uint32_t byte_off, float4 v) {
    frcnn_u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, b, (int)byte_off, 0, /*aux: sc1*/ 16);
}

// plain 16-byte store through a descriptor: a 32-bit offset register instead of a 64-bit address pair per store, and lanes whose
// offset is kBufOob store nothing (no exec-mask branch around ragged edges)
__device__ __forceinline__ void frcnn_buf_store_b128(frcnn_buf_t b, uint32_t byte_off, uint4 v) {
    frcnn_u32x4 u;
    u.x = v.x; u.y = v.y; u.z = v.z; u.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(u, b, (int)byte_off, 0, 0);
}

// 16-byte store at (per-lane offset, range-checked) + (wave-uniform scalar offset, added after the check), cache policy AUX as a
// template argument: 0 = write-back, 16 = sc1 (write-through: the bytes leave the L2 while the kernel is still running instead of
// after its last wave -- 30 MB of output cost 3.6 us that way and 4.5 us as plain stores, scripts/micro/store_micro.hip)
template <int AUX>
__device__ __forceinline__ void frcnn_buf_store_f32x4_soff(frcnn_buf_t b, uint32_t byte_off, uint32_t soff, float4 v) {
    frcnn_u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, b, (int)byte_off, (int)soff, AUX);
}

// 8-byte load through a descriptor (out-of-range lanes read 0)
__device__ __forceinline__ uint2 frcnn_buf_load_b64(frcnn_buf_t b, uint32_t byte_off) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(b, (int)byte_off, 0, 0);
    return make_uint2(v.x, v.y);
}
// 8-byte load at (per-lane offset, range-checked) + (wave-uniform scalar offset, added after the check)
__device__ __forceinline__ uint2 frcnn_buf_load_b64_soff(frcnn_buf_t b, uint32_t byte_off, uint32_t soff) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(b, (int)byte_off, (int)soff, 0);
    return make_uint2(v.x, v.y);
}
// streaming reads with a cache policy: AUX as in the stores (1 = sc0, 2 = nt, 16 = sc1) -- for bytes that are read exactly once
template <int AUX>
__device__ __forceinline__ float frcnn_buf_load_f32_soff_aux(frcnn_buf_t b, uint32_t byte_off, uint32_t soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b, (int)byte_off, (int)soff, AUX));
}
template <int AUX>
__device__ __forceinline__ float4 frcnn_buf_load_f32x4_soff_aux(frcnn_buf_t b, uint32_t byte_off, uint32_t soff) {
    const frcnn_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b, (int)byte_off, (int)soff, AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <int AUX>
__device__ __forceinline__ uint2 frcnn_buf_load_b64_soff_aux(frcnn_buf_t b, uint32_t byte_off, uint32_t soff) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(b, (int)byte_off, (int)soff, AUX);
    return make_uint2(v.x, v.y);
}
// 8-byte store at (per-lane offset, range-checked) + (wave-uniform scalar offset), cache policy AUX (16 = sc1 write-through) as a template argument
template <int AUX>
__device__ __forceinline__ void frcnn_buf_store_b64_soff(frcnn_buf_t b, uint32_t byte_off, uint32_t soff, uint2 v) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t u;
    u.x = v.x; u.y = v.y;
    __builtin_amdgcn_raw_buffer_store_b64(u, b, (int)byte_off, (int)soff, AUX);
}
// 8-byte store through a descriptor (lanes whose offset is kBufOob store nothing)
__device__ __forceinline__ void frcnn_buf_store_b64(frcnn_buf_t b, uint32_t byte_off, uint2 v) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t u;
    u.x = v.x; u.y = v.y;
    __builtin_amdgcn_raw_buffer_store_b64(u, b, (int)byte_off, 0, 0);
}

__device__ __forceinline__ void frcnn_buf_store_f32(frcnn_buf_t b, uint32_t byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), b, (int)byte_off, 0, 0);
}

// LDS-DMA: one wave-instruction moves 64 x 16 bytes from the buffer straight into LDS at lds_wave_base + 16 * lane -- no staging
// VGPRs, no ds_write.  The destination is lane-linear by construction (any swizzle goes on the per-lane SOURCE offset and on the
// reader, cdna_hip_programming.md rule 21); out-of-range lanes deposit zeros; `soff` is a wave-uniform byte offset added after the
// range check.  Completion is counted by vmcnt: the caller waits with frcnn_wait_vmcnt<N>() -- the compiler does not see these loads,
// which is the point (it would drain vmcnt(0) at the next barrier and serialise the prefetch with the MFMAs).
__device__ __forceinline__ void frcnn_buf_load_lds_b128(frcnn_buf_t b, void *lds_wave_base, uint32_t byte_off, uint32_t soff) {
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_wave_base;
#ifdef FRCNN_DMA_KEEP_M0
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(la), "v"(byte_off), "s"(b), "s"(soff)
                 : "memory");
#else
    // M0 is clobbered, not saved and restored: nothing else in these kernels reads it between two pieces (the compiler re-materialises
    // it where IT needs it: ds_append, readlane), and the two extra scalar moves per 1 KB piece sat in the issue stream of every wave
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(la), "v"(byte_off), "s"(b), "s"(soff)
                 : "memory", "m0");
#endif
}
// same with the non-temporal policy (`nt`) on the read: for a stream that ONE workgroup reads ONCE (the FC weight tiles of linear_bf16.hip) -- the lines
// are not kept in the L2 for a second reader that never comes (MI355X_MICROARCH.md row "nt-weights": issued -> landed -18 %); never for operands other
// workgroups re-read from the L2
__device__ __forceinline__ void frcnn_buf_load_lds_b128_nt(frcnn_buf_t b, void *lds_wave_base, uint32_t byte_off, uint32_t soff) {
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
                 :
                 : "s"(la), "v"(byte_off), "s"(b), "s"(soff)
                 : "memory", "m0");
}
// 4-byte store with a cache policy (16 = sc1: write-through -- a split-K slab drains while the other workgroups are still multiplying, not after the last wave)
template <int AUX>
__device__ __forceinline__ void frcnn_buf_store_f32_aux(frcnn_buf_t b, uint32_t byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), b, (int)byte_off, 0, AUX);
}
// same, 4 bytes per lane: 256 B per wave-instruction at lds_wave_base + 4 * lane
__device__ __forceinline__ void frcnn_buf_load_lds_b32(frcnn_buf_t b, void *lds_wave_base, uint32_t byte_off, uint32_t soff) {
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_wave_base;
#ifdef FRCNN_DMA_KEEP_M0
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(la), "v"(byte_off), "s"(b), "s"(soff)
                 : "memory");
#else
    // (la / soff are wave-uniform by contract; inside a divergent region the compiler may still hold them in VGPRs, and the "s"
    // constraint does not move them back: readfirstlane does, and folds away where the value already is scalar)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"       // M0 clobbered, as in the 16-byte form
                 :
                 : "s"(__builtin_amdgcn_readfirstlane((int)la)), "v"(byte_off), "s"(b), "s"(__builtin_amdgcn_readfirstlane((int)soff))
                 : "memory", "m0");
#endif
}
template <int N>
__device__ __forceinline__ void frcnn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier WITHOUT the fences of __syncthreads() (which would drain every outstanding LDS-DMA): the caller has already
// waited for exactly the loads it needs; the "memory" clobbers keep the compiler from moving LDS accesses across it.
__device__ __forceinline__ void frcnn_barrier_nofence() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// idle for about n x 64 clocks (s_sleep takes a 7-bit immediate)
__device__ __forceinline__ void frcnn_sleep_64clk(int n) {
    for (; n >= 127; n -= 127) __builtin_amdgcn_s_sleep(127);
    for (; n >= 16; n -= 16) __builtin_amdgcn_s_sleep(16);
    for (; n > 0; --n) __builtin_amdgcn_s_sleep(1);
}
