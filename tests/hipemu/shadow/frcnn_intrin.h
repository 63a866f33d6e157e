// Emulator stand-in for csrc/frcnn_intrin.h.
#pragma once
#include <hip/hip_runtime.h>
static inline float frcnn_max_f32(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
