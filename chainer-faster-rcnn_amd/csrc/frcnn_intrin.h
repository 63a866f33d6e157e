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
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
