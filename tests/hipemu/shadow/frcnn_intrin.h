// Emulator stand-in for csrc/frcnn_intrin.h.
#pragma once
#include <hip/hip_runtime.h>
static inline float frcnn_max_f32(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
static inline uint32_t hipemu_f32_to_bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
#ifdef FRCNN_HALF_F16        // the fp16 instantiation of the 16-bit chain (csrc/frcnn_intrin.h): pack / widen / MFMA operate on IEEE half
static inline uint32_t hipemu_f32_to_f16_rne(float f) { const _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static inline uint32_t frcnn_pack_bf16x2(float lo, float hi) { return (hipemu_f32_to_f16_rne(lo) & 0xffffu) | (hipemu_f32_to_f16_rne(hi) << 16); }
static inline float frcnn_h16_to_f32(uint16_t h) { _Float16 v; memcpy(&v, &h, 2); return (float)v; }
#else
static inline uint32_t frcnn_pack_bf16x2(float lo, float hi) { return (hipemu_f32_to_bf16_rne(lo) & 0xffffu) | (hipemu_f32_to_bf16_rne(hi) << 16); }
static inline float frcnn_h16_to_f32(uint16_t h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
#endif
static inline uint32_t frcnn_f32_to_h16_exact(float v) { return frcnn_pack_bf16x2(v, 0.0f) & 0xffffu; }
static inline uint32_t frcnn_wave_or_u32(uint32_t v) {
    int x = (int)v;
    for (int d = 32; d > 0; d >>= 1) x |= __shfl_xor(x, d);
    return (uint32_t)x;
}
static inline float frcnn_min_f32(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
static inline float frcnn_max3_f32(float a, float b, float c) { return frcnn_max_f32(frcnn_max_f32(a, b), c); }

typedef float frcnn_f32x16 __attribute__((ext_vector_type(16)));
static inline float hipemu_bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
__attribute__((noinline)) static frcnn_f32x16 frcnn_mfma_32x32x16_bf16(uint4 a, uint4 b, frcnn_f32x16 c) {
    unsigned short ab[16];
    memcpy(ab, &a, 16);
    memcpy(ab + 8, &b, 16);
    auto e = hipemu::wave_exchange(ab, sizeof(ab), HIPEMU_SITE());
    if (e.present != ~0ull) { fprintf(stderr, "hipemu: MFMA with a partial wave\n"); abort(); }
    const int l = hipemu::G().cur->lane, j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            unsigned short av, bv;
            memcpy(&av, e.vals[i + 32 * (k >> 3)] + 2 * (k & 7), 2);
            memcpy(&bv, e.vals[j + 32 * (k >> 3)] + 16 + 2 * (k & 7), 2);
            acc += frcnn_h16_to_f32(av) * frcnn_h16_to_f32(bv);
        }
        c[r] = acc;
    }
    return c;
}
static inline void frcnn_pin(float4 &) {}       // (a scheduling constraint on the device; nothing to do on the host)
static inline void frcnn_pin(float &) {}
static inline uint32_t frcnn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
static inline void frcnn_split3_pair(float v0, float v1, uint32_t &h, uint32_t &m, uint32_t &l) {
    h = frcnn_pack_bf16x2(v0, v1);
    const float d0 = v0 - __uint_as_float(h << 16), d1 = v1 - __uint_as_float(h & 0xffff0000u);
    m = frcnn_pack_bf16x2(d0, d1);
    l = frcnn_pack_bf16x2(d0 - __uint_as_float(m << 16), d1 - __uint_as_float(m & 0xffff0000u));
}

__attribute__((noinline)) static int frcnn_lds_append(int *ctr) {
    int z = 0;
    auto e = hipemu::wave_exchange(&z, sizeof(z), HIPEMU_SITE());
    const int first = __builtin_ctzll(e.present);
    int old = 0;
    if (hipemu::G().cur->lane == first) { old = *ctr; *ctr += __builtin_popcountll(e.present); }
    return __builtin_amdgcn_readfirstlane(old);
}
__attribute__((noinline)) static float frcnn_wave_shl1_f32(float v) {
    auto e = hipemu::wave_exchange(&v, sizeof(v), HIPEMU_SITE());
    const int s = hipemu::G().cur->lane + 1;
    float r = 0.0f;
    if (s < 64 && ((e.present >> s) & 1)) memcpy(&r, e.vals[s], sizeof(r));
    return r;
}
static inline float frcnn_lane_xor1_f32(float v) { return __shfl_xor(v, 1); }
static inline float frcnn_max_lane_xor1_f32(float v) { return frcnn_max_f32(v, __shfl_xor(v, 1)); }
