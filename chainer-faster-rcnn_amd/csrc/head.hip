// head.hip -- detection-head epilogue: per-class box decoding + clipping and the class softmax.
// Replaces bbox_transform_inv + clip_boxes + F.softmax at /root/reference/models/faster_rcnn.py:175-178
// (bbox_transform.py:41-99 on (R,4) boxes x (R,4*ncls) deltas).  fp32, the reference's operation order,
// no FMA contraction; exp evaluated in double and rounded to fp32.
#include "frcnn_common.h"

namespace {

__device__ __forceinline__ float clip_like_numpy(float v, float hi) { return (v != v) ? v : fmaxf(fminf(v, hi), 0.0f); }

__global__ void __launch_bounds__(256)
head_decode_kernel(const float *__restrict__ boxes, const float *__restrict__ deltas, int R, int ncls, int im_h, int im_w,
                   float *__restrict__ pred) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * ncls) return;
    const int r = i / ncls;
    const float4 b = reinterpret_cast<const float4 *>(boxes)[r];
    const float4 d = reinterpret_cast<const float4 *>(deltas)[i];
    const float widths = b.z - b.x + 1.0f, heights = b.w - b.y + 1.0f;
    const float ctr_x = b.x + 0.5f * widths, ctr_y = b.y + 0.5f * heights;
    const float pcx = d.x * widths + ctr_x, pcy = d.y * heights + ctr_y;
    const float pw = (float)exp((double)d.z) * widths, ph = (float)exp((double)d.w) * heights;
    const float mx = (float)(im_w - 1), my = (float)(im_h - 1);
    float4 o;
    o.x = clip_like_numpy(pcx - 0.5f * pw, mx);
    o.y = clip_like_numpy(pcy - 0.5f * ph, my);
    o.z = clip_like_numpy(pcx + 0.5f * pw, mx);
    o.w = clip_like_numpy(pcy + 0.5f * ph, my);
    reinterpret_cast<float4 *>(pred)[i] = o;
}

__global__ void __launch_bounds__(256)
row_softmax_kernel(const float *__restrict__ s, int R, int n, float *__restrict__ p) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float *row = s + (size_t)r * n;
    float m = row[0];
    for (int c = 1; c < n; ++c) m = fmaxf(m, row[c]);
    float sum = 0.0f;
    for (int c = 0; c < n; ++c) sum += expf(row[c] - m);
    for (int c = 0; c < n; ++c) p[(size_t)r * n + c] = expf(row[c] - m) / sum;
}

}  // namespace

extern "C" int frcnn_head_decode(const float *boxes, const float *deltas, const float *cls_score, int R, int ncls, int im_h, int im_w,
                                 float *pred_boxes, float *cls_prob, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!boxes || !deltas || !cls_score || !pred_boxes || !cls_prob || R < 0 || ncls < 1) return FRCNN_ERR_INVALID;
    if (R == 0) return FRCNN_OK;
    hipLaunchKernelGGL(head_decode_kernel, dim3(frcnn_cdiv(R * ncls, 256)), dim3(256), 0, stream, boxes, deltas, R, ncls, im_h, im_w,
                       pred_boxes);
    hipLaunchKernelGGL(row_softmax_kernel, dim3(frcnn_cdiv(R, 256)), dim3(256), 0, stream, cls_score, R, ncls, cls_prob);
    return frcnn_launch_status();
}
