// Emulator stand-in for csrc/frcnn_buffer.h: a descriptor is (base, size); out-of-range loads return 0,
// per dword, as the hardware's raw-buffer range check does.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
struct frcnn_buf_t { const char *base; uint32_t bytes; };
constexpr uint32_t kBufOob = 0x80000000u;
static inline frcnn_buf_t frcnn_make_buf(const void *base, uint32_t bytes) { return frcnn_buf_t{(const char *)base, bytes}; }
static inline float frcnn_buf_load_f32(frcnn_buf_t b, uint32_t off) {
    float v = 0.0f;
    if ((uint64_t)off + 4 <= b.bytes) memcpy(&v, b.base + off, 4);
    return v;
}
static inline float frcnn_buf_load_f32_soff(frcnn_buf_t b, uint32_t off, uint32_t soff) {
    float v = 0.0f;
    if ((uint64_t)off + 4 <= b.bytes) memcpy(&v, b.base + off + soff, 4);
    return v;
}
static inline float4 frcnn_buf_load_f32x4(frcnn_buf_t b, uint32_t off) {
    return make_float4(frcnn_buf_load_f32(b, off), frcnn_buf_load_f32(b, off + 4), frcnn_buf_load_f32(b, off + 8), frcnn_buf_load_f32(b, off + 12));
}
static inline float4 frcnn_buf_load_f32x4_soff(frcnn_buf_t b, uint32_t off, uint32_t soff) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((uint64_t)off + 16 <= b.bytes) memcpy(&v, b.base + off + soff, 16);
    return v;
}
static inline void frcnn_buf_store_f32x4_wt(frcnn_buf_t b, uint32_t off, float4 v) {
    if ((uint64_t)off + 16 <= b.bytes) memcpy(const_cast<char *>(b.base) + off, &v, 16);
}
static inline void frcnn_buf_store_b128(frcnn_buf_t b, uint32_t off, uint4 v) {
    if ((uint64_t)off + 16 <= b.bytes) memcpy(const_cast<char *>(b.base) + off, &v, 16);
}
template <int AUX> static inline void frcnn_buf_store_f32x4_soff(frcnn_buf_t b, uint32_t off, uint32_t soff, float4 v) {
    if ((uint64_t)off + 16 <= b.bytes) memcpy(const_cast<char *>(b.base) + off + soff, &v, 16);
}
static inline uint2 frcnn_buf_load_b64(frcnn_buf_t b, uint32_t off) {
    uint2 v = make_uint2(0u, 0u);
    if ((uint64_t)off + 8 <= b.bytes) memcpy(&v, b.base + off, 8);
    return v;
}
static inline uint2 frcnn_buf_load_b64_soff(frcnn_buf_t b, uint32_t off, uint32_t soff) {
    uint2 v = make_uint2(0u, 0u);
    if ((uint64_t)off + 8 <= b.bytes) memcpy(&v, b.base + off + soff, 8);
    return v;
}
template <int AUX> static inline float frcnn_buf_load_f32_soff_aux(frcnn_buf_t b, uint32_t off, uint32_t soff) { return frcnn_buf_load_f32_soff(b, off, soff); }
template <int AUX> static inline float4 frcnn_buf_load_f32x4_soff_aux(frcnn_buf_t b, uint32_t off, uint32_t soff) { return frcnn_buf_load_f32x4_soff(b, off, soff); }
template <int AUX> static inline uint2 frcnn_buf_load_b64_soff_aux(frcnn_buf_t b, uint32_t off, uint32_t soff) { return frcnn_buf_load_b64_soff(b, off, soff); }
template <int AUX> static inline void frcnn_buf_store_b64_soff(frcnn_buf_t b, uint32_t off, uint32_t soff, uint2 v) {
    if ((uint64_t)off + 8 <= b.bytes) memcpy(const_cast<char *>(b.base) + off + soff, &v, 8);
}
static inline void frcnn_buf_store_b64(frcnn_buf_t b, uint32_t off, uint2 v) {
    if ((uint64_t)off + 8 <= b.bytes) memcpy(const_cast<char *>(b.base) + off, &v, 8);
}
static inline void frcnn_buf_store_f32(frcnn_buf_t b, uint32_t off, float v) {
    if ((uint64_t)off + 4 <= b.bytes) memcpy(const_cast<char *>(b.base) + off, &v, 4);
}
static inline void frcnn_buf_load_lds_b128(frcnn_buf_t b, void *lds_wave_base, uint32_t off, uint32_t soff) {
    // lane-linear destination; the range check sees the per-lane offset only (soff is added after it), as on the hardware
    // (per DWORD: a 16-byte access that straddles the end of the buffer keeps its in-range dwords -- scripts/micro/dma_align_micro.hip)
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 4; ++k)
        if ((uint64_t)off + 4 * k + 4 <= b.bytes) memcpy(&v[k], b.base + off + soff + 4 * k, 4);
    hipemu::dma_deposit((char *)lds_wave_base + 16 * (threadIdx.x & 63), v, 16);       // lands now, or at the covering wait (HIPEMU_DMA_DEFER=1)
}
static inline void frcnn_buf_load_lds_b128_nt(frcnn_buf_t b, void *lds_wave_base, uint32_t off, uint32_t soff) { frcnn_buf_load_lds_b128(b, lds_wave_base, off, soff); }
template <int AUX> static inline void frcnn_buf_store_f32_aux(frcnn_buf_t b, uint32_t off, float v) { frcnn_buf_store_f32(b, off, v); }
static inline void frcnn_buf_load_lds_b32(frcnn_buf_t b, void *lds_wave_base, uint32_t off, uint32_t soff) {
    float v = 0.0f;
    if ((uint64_t)off + 4 <= b.bytes) memcpy(&v, b.base + off + soff, 4);
    hipemu::dma_deposit((char *)lds_wave_base + 4 * (threadIdx.x & 63), &v, 4);
}
template <int N> static inline void frcnn_wait_vmcnt() { hipemu::dma_wait(N); }
static inline void frcnn_barrier_nofence() { __builtin_amdgcn_s_barrier(); }       // the barrier alone: no wait on loads in flight
static inline void frcnn_sleep_64clk(int) {}
