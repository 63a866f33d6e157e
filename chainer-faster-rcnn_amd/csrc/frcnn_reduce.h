// frcnn_reduce.h -- the split-K combine of the fully connected layers (gemm.hip, conv_bf16.hip, conv_f32s.hip).
#pragma once
#include <frcnn_intrin.h>   // angle brackets: the test emulator shadows it

// Sum of output i over the `splits` partial slabs ([splits][total] fp32), IN SLAB ORDER -- bit-identical to
//     for (s = 0; s < splits; ++s) v += part[s * total + i];
// but with the loads issued in batches of eight: written as that loop, every load was followed by s_waitcnt vmcnt(0) and the add (8-16
// dependent memory round trips per output).
__device__ __forceinline__ float frcnn_sum_splits(const float *__restrict__ part, size_t total, size_t i, int splits) {
    float v = 0.0f;
    int s = 0;
    for (; s + 8 <= splits; s += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(s + u) * total + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) frcnn_pin(t[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    if (s < splits) {                                     // the remainder as one more batch; slabs past the end are neither loaded nor added
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = s + u < splits ? part[(size_t)(s + u) * total + i] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) frcnn_pin(t[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s + u < splits) v += t[u];
    }
    return v;
}
