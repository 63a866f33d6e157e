// conv_f16.hip -- the fp16 instantiation of conv_bf16.hip: the 3x3 / 1x1 convolution kernels (conv_dma / strip forms), packing and layout kernels, 2x2 max-pool, the fused RPN heads, the round-5 FC kernel.
// north_star: "MFMA-tiled ... conv stack (fp16/bf16 accumulate fp32)".  Same kernels, layouts, LDS-DMA schedule and MFMA rate as the bf16 line (BASELINE configs[2]);
// the operands carry 10 mantissa bits instead of 7, which is what VERDICT r05 (weak #3) asked for: bf16 operands leave conv5_3 at 1.1e-2 of the fp32 map and
// 23 / 300 proposal indices in place from the image.  Entry points: the *_f16* twins of the *_bf16* ones (frcnn_f16_names.h; declared in include/frcnn_hip.h).
#define FRCNN_HALF_F16 1
#include "frcnn_f16_names.h"
#include "conv_bf16.hip"
