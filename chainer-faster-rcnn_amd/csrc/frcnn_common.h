// frcnn_common.h -- shared helpers of libfrcnn_hip (MI355X / gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "frcnn_hip.h"
#include "frcnn_tune.h"

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

// compute units of the current device (256 on MI355X; the test emulator reports its own count)
static inline int frcnn_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    return cus;
}

