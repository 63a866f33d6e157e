// frcnn_intrin.h -- single-instruction helpers the compiler will not emit on its own.
// (the test emulator shadows this header with plain C++ of the same semantics)
#pragma once
#include <hip/hip_runtime.h>

// v_max_f32: IEEE maxNum -- a NaN operand is ignored.  fmaxf() compiles to the same instruction plus two
// canonicalising v_max x,x,x (sNaN quieting) that a hot loop of compares cannot afford.
__device__ __forceinline__ float frcnn_max_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float frcnn_min_f32(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// OR of a 32-bit value over the 64 lanes of the wave, returned wave-uniform (SGPR): four DPP row shifts, two row broadcasts and a
// v_readlane -- 7 VALU instructions, no LDS crossbar round trips (six ds_bpermute steps cost ~6 x 100+ cycles of latency each).
__device__ __forceinline__ uint32_t frcnn_wave_or_u32(uint32_t v) {
    int x = (int)v;
    x |= __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);    // row_shr:1
    x |= __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);    // row_shr:2
    x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);    // row_shr:4
    x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);    // row_shr:8   -> lane 15 of each row holds the row's OR
    x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);    // row_bcast:15 into rows 1 and 3
    x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's OR
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

// The 16-bit operand format of the translation unit.  Default: bf16.  A TU compiled with FRCNN_HALF_F16 (conv_f16.hip, conv_f16_pair.hip, linear_f16.hip: the SAME
// kernel sources included under frcnn_f16_names.h, which renames their entry points *_bf16* -> *_f16*) is the fp16 instantiation of the 16-bit chain that
// north_star allows ("fp16/bf16 accumulate fp32"): same layouts, same LDS-DMA, same MFMA rate (v_mfma_f32_32x32x16_f16), 10 mantissa bits instead of 7.  In such a
// TU the three helpers below -- pack, widen, MFMA -- are the fp16 ones; "bf16" in their names (and in the kernels' names) then reads "the TU's 16-bit format".
// This is a dtype instantiation of one gfx950 code path, not a platform switch.  fp16 has 5 exponent bits: |v| > 65504 rounds to Inf (bf16 keeps fp32's range).
#ifdef FRCNN_HALF_F16
__device__ __forceinline__ uint32_t frcnn_pack_bf16x2(float lo, float hi) {            // two fp32 -> two fp16, round to nearest even (v_cvt_pk_f16_f32); `lo` in bits 0-15
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ float frcnn_h16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }     // exact (v_cvt_f32_f16)
#else
// v_cvt_pk_bf16_f32: two fp32 -> two bf16 (round to nearest even) in ONE instruction; `lo` lands in bits 0-15.  (The software
// form -- add 0x7fff + lsb, shift -- is ~8 VALU instructions per value: 2 M of the 6.7 M VALU instructions of the bf16-output RoI
// kernel, r02 counters.)
__device__ __forceinline__ uint32_t frcnn_pack_bf16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float frcnn_h16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }          // bf16 is the top half of an fp32
#endif
// the 16 bits of a value that IS representable in the TU's 16-bit format (a maximum of such values): no rounding happens
__device__ __forceinline__ uint32_t frcnn_f32_to_h16_exact(float v) { return frcnn_pack_bf16x2(v, 0.0f) & 0xffffu; }

// v_max3_f32: max(max(a, b), c) with the same NaN rule, one instruction for two updates of a running maximum
__device__ __forceinline__ float frcnn_max3_f32(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// v_mfma_f32_32x32x16_bf16: D(32x32 f32) += A(32x16 bf16) * B(16x32 bf16).  Lane l supplies A[i = l&31][k = 8*(l>>5) .. +7] and
// B[k = 8*(l>>5) .. +7][j = l&31], eight bf16 each = one uint4 (element t in the low/high half of word t/2); D uses the standard
// 32x32 map (register r of lane l = row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31).
typedef float frcnn_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ frcnn_f32x16 frcnn_mfma_32x32x16_bf16(uint4 a, uint4 b, frcnn_f32x16 c) {
#ifdef FRCNN_HALF_F16
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));                       // v_mfma_f32_32x32x16_f16: same shape, same rate, fp16 operands
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
#else
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
}

// v_alignbit_b32: bits [sh+31 : sh] of the 64-bit value hi:lo
// Pins a loaded vector in registers at this point of the program: everything fetched into pinned values before this line has to
// have arrived before anything after it runs, so a batch of loads followed by pins is issued as a batch and waited for once.  (Without
// it the compiler is free to pair each load with its first use -- load, s_waitcnt vmcnt(0), add -- one dependent round trip per vector.)
__device__ __forceinline__ void frcnn_pin(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void frcnn_pin(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ uint32_t frcnn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// (the split tensors of conv_f32s.hip are bf16 by construction: in an fp16 translation unit this helper still parses -- roi_pool.hip names it in a template
// branch the fp16 twins never instantiate -- but must not be called)
// (v0, v1) -> the three packed bf16 pairs (h, m, l) with h + m + l == v exactly (round to nearest even at every step; the
// differences are exact in fp32): the "split tensors" of conv_f32s.hip.  PRECONDITION: |v| finite and below the largest bf16
// (3.39e38): for +-Inf, or a value whose bf16 rounding overflows, h is Inf and the lower terms are NaN (Inf - Inf), i.e. an overflowed
// activation turns the whole six-product sum into NaN where the fp32 MFMA path would carry Inf.  Both mean "this step has
// diverged"; the split path does not pay two extra VALU instructions per pair in its epilogues to tell them apart.
__device__ __forceinline__ void frcnn_split3_pair(float v0, float v1, uint32_t &h, uint32_t &m, uint32_t &l) {
    h = frcnn_pack_bf16x2(v0, v1);
    const float d0 = v0 - __uint_as_float(h << 16), d1 = v1 - __uint_as_float(h & 0xffff0000u);
    m = frcnn_pack_bf16x2(d0, d1);
    l = frcnn_pack_bf16x2(d0 - __uint_as_float(m << 16), d1 - __uint_as_float(m & 0xffff0000u));
}


// DPP quad_perm [1,0,3,2]: every lane receives the value of lane ^ 1 -- one VALU instruction (a __shfl_xor may go through the LDS crossbar)
__device__ __forceinline__ float frcnn_lane_xor1_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
}

// max(v, value of lane ^ 1) as ONE instruction: v_max_f32 with the DPP modifier on its first source (quad_perm [1,0,3,2]).  The ISA wants two
// wait states between a VALU write of a VGPR and a DPP read of it, and the compiler's hazard recognizer sees neither a DPP read inside an asm
// string nor (as the producer) a VALU write inside one -- so the wait states are part of the string (the bare instruction sat ONE
// s_nop behind its producer in conv1_pair_pc_bf16_kernel).  tests/test_isa_waits.py checks every DPP read of the library's listings.
__device__ __forceinline__ float frcnn_max_lane_xor1_f32(float v) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}

// ds_append: ONE wave-level LDS operation that adds the number of active lanes to a counter and hands every lane the old value.  The
// hardware addresses the counter through M0[15:0]: it must sit in the first 64 KB of the workgroup's LDS.
__device__ __forceinline__ int frcnn_lds_append(int *ctr) {
    return __builtin_amdgcn_ds_append((__attribute__((address_space(3))) int *)ctr);
}
// DPP wave_shl:1 -- lane l receives lane l + 1's value, lane 63 receives 0: one VALU instruction, no LDS crossbar trip
__device__ __forceinline__ float frcnn_wave_shl1_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
