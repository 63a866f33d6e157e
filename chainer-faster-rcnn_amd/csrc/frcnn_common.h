// frcnn_common.h -- shared helpers of libfrcnn_hip (MI355X / gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "frcnn_hip.h"

#define FRCNN_WAVE 64

// Turn the status of the launches issued so far into the ABI's error code.
static inline int frcnn_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FRCNN_OK : -(1000 + (int)e);
}
#define FRCNN_HIP_TRY(expr)                                   \
    do {                                                      \
        hipError_t e__ = (expr);                              \
        if (e__ != hipSuccess) return -(1000 + (int)e__);     \
    } while (0)

static inline size_t frcnn_align256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int frcnn_cdiv(int a, int b) { return (a + b - 1) / b; }
