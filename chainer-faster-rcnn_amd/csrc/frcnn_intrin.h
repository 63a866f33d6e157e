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
