// roi_f16.hip -- the fp16 instantiation of roi_pool.hip's two 16-bit entry points: RoI pooling with the pooled maxima written as fp16 (the fp16 FC head's input)
// and RoI pooling straight from the fp16 chain's channel-blocked map (a cell's channels widen with v_cvt_f32_f16; a maximum of fp16 values is an fp16 value, so the
// fp16 output form rounds nothing).  roi_pool.hip leaves its fp32 entry points out of a translation unit compiled with FRCNN_HALF_F16.
// Replaces F.roi_pooling_2d (models/faster_rcnn.py:125-126) on the fp16 line.
#define FRCNN_HALF_F16 1
#include "frcnn_f16_names.h"
#include "roi_pool.hip"
