// head.hip -- detection-head epilogue: per-class box decoding + clipping and the class softmax.
// Replaces bbox_transform_inv + clip_boxes + F.softmax at /root/reference/models/faster_rcnn.py:175-178
// (bbox_transform.py:41-99 on (R,4) boxes x (R,4*ncls) deltas).  fp32, the reference's operation order,
// no FMA contraction; exp evaluated in double and rounded to fp32.
#include "frcnn_common.h"

namespace {

__device__ __forceinline__ float clip_like_numpy(float v, float hi) { return (v != v) ? v : fmaxf(fminf(v, hi), 0.0f); }

__global__ void __launch_bounds__(256)
head_decode_kernel(const float *__restrict__ boxes, const float *__restrict__ deltas, int R, int ncls, int clip, int im_h, int im_w,
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
    float4 o = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
    if (clip) {
        o.x = clip_like_numpy(o.x, mx); o.y = clip_like_numpy(o.y, my);
        o.z = clip_like_numpy(o.z, mx); o.w = clip_like_numpy(o.w, my);
    }
    reinterpret_cast<float4 *>(pred)[i] = o;
}

// clip_boxes (bbox_transform.py:79-99) on n boxes of 4 floats, in place
__global__ void __launch_bounds__(256)
clip_boxes_kernel(float *__restrict__ boxes, int n, int im_h, int im_w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float mx = (float)(im_w - 1), my = (float)(im_h - 1);
    float4 b = reinterpret_cast<float4 *>(boxes)[i];
    b.x = clip_like_numpy(b.x, mx); b.y = clip_like_numpy(b.y, my);
    b.z = clip_like_numpy(b.z, mx); b.w = clip_like_numpy(b.w, my);
    reinterpret_cast<float4 *>(boxes)[i] = b;
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

// forward.py:50-53: for cls_id in 1..ncls-1: dets = hstack(bbox[:, 4*cls_id:4*cls_id+4], clss[:, cls_id]) -> (ncls-1, R, 5)
__global__ void __launch_bounds__(256)
class_dets_kernel(const float *__restrict__ cls_prob, const float *__restrict__ pred_boxes, int R, int ncls, float *__restrict__ dets) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (ncls - 1) * R) return;
    const int c = i / R + 1, r = i - (c - 1) * R;
    const float4 b = reinterpret_cast<const float4 *>(pred_boxes)[(size_t)r * ncls + c];
    float *d = dets + (size_t)i * 5;
    d[0] = b.x; d[1] = b.y; d[2] = b.z; d[3] = b.w;
    d[4] = cls_prob[(size_t)r * ncls + c];
}

// forward.py:33-45 img_preprocessing on the device: uint8 HWC (BGR, as cv.imread returns) -> float32 mean-subtracted,
// bilinearly resized (cv.resize(..., fx=fy=im_scale, INTER_LINEAR) on the float image: half-pixel centres, edge clamp,
// horizontal then vertical blend in float), transposed to CHW.  One thread per output pixel.
struct PixelMeans { double m[4]; };
__global__ void __launch_bounds__(256)
preprocess_kernel(const uint8_t *__restrict__ img, int H, int W, int C, PixelMeans means, int OH, int OW, double inv_scale_x, double inv_scale_y,
                  float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= OH * OW) return;
    const int oy = i / OW, ox = i - oy * OW;
    float fx = (float)(((double)ox + 0.5) * inv_scale_x - 0.5), fy = (float)(((double)oy + 0.5) * inv_scale_y - 0.5);
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= (float)sx; fy -= (float)sy;
    if (sx < 0) { fx = 0.0f; sx = 0; }
    if (sx >= W - 1) { fx = 0.0f; sx = W - 1; }
    if (sy < 0) { fy = 0.0f; sy = 0; }
    if (sy >= H - 1) { fy = 0.0f; sy = H - 1; }
    const int sx1 = min(sx + 1, W - 1), sy1 = min(sy + 1, H - 1);
    for (int c = 0; c < C; ++c) {
        const double mean = means.m[c];
        const float p00 = (float)((double)img[((size_t)sy * W + sx) * C + c] - mean), p01 = (float)((double)img[((size_t)sy * W + sx1) * C + c] - mean);
        const float p10 = (float)((double)img[((size_t)sy1 * W + sx) * C + c] - mean), p11 = (float)((double)img[((size_t)sy1 * W + sx1) * C + c] - mean);
        const float r0 = p00 * (1.0f - fx) + p01 * fx, r1 = p10 * (1.0f - fx) + p11 * fx;
        out[((size_t)c * OH + oy) * OW + ox] = r0 * (1.0f - fy) + r1 * fy;
    }
}

// cls_score and bbox_pred of the stacked head GEMM (one launch instead of two: `out` is (R, ld) with the class scores in columns
// [0, ncls) and the 4*ncls deltas from column `dcol`, 16-byte aligned), decoded in one kernel: thread (r, c) decodes + clips class c's
// box of RoI r and writes softmax(score[r])[c].  The softmax is evaluated per thread exactly as row_softmax_kernel does per row
// (max, sum of expf(s - m) in class order, expf(s - m) / sum), so the probabilities are the same bits.
__global__ void __launch_bounds__(256)
head_decode_softmax_kernel(const float *__restrict__ boxes, const float *__restrict__ out, int ld, int dcol, int R, int ncls, int im_h,
                           int im_w, float *__restrict__ pred, float *__restrict__ prob) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * ncls) return;
    const int r = i / ncls, c = i - r * ncls;
    const float *row = out + (size_t)r * ld;
    const float4 b = reinterpret_cast<const float4 *>(boxes)[r];
    const float4 d = *reinterpret_cast<const float4 *>(row + dcol + 4 * c);
    const float widths = b.z - b.x + 1.0f, heights = b.w - b.y + 1.0f;
    const float ctr_x = b.x + 0.5f * widths, ctr_y = b.y + 0.5f * heights;
    const float pcx = d.x * widths + ctr_x, pcy = d.y * heights + ctr_y;
    const float pw = (float)exp((double)d.z) * widths, ph = (float)exp((double)d.w) * heights;
    const float mx = (float)(im_w - 1), my = (float)(im_h - 1);
    float4 o = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
    o.x = clip_like_numpy(o.x, mx); o.y = clip_like_numpy(o.y, my);
    o.z = clip_like_numpy(o.z, mx); o.w = clip_like_numpy(o.w, my);
    reinterpret_cast<float4 *>(pred)[i] = o;
    float m = row[0];
    for (int k = 1; k < ncls; ++k) m = fmaxf(m, row[k]);
    float sum = 0.0f;
    for (int k = 0; k < ncls; ++k) sum += expf(row[k] - m);
    prob[i] = expf(row[c] - m) / sum;
}

}  // namespace

extern "C" {

int frcnn_preprocess_u8(const uint8_t *img, int H, int W, int C, const double *means_host, double im_scale, int OH, int OW, float *out,
                        void *stream) {
    if (!img || !means_host || !out || H < 1 || W < 1 || C < 1 || C > 4 || OH < 1 || OW < 1 || !(im_scale > 0.0)) return FRCNN_ERR_INVALID;
    PixelMeans pm;
    for (int c = 0; c < 4; ++c) pm.m[c] = c < C ? means_host[c] : 0.0;
    hipLaunchKernelGGL(preprocess_kernel, dim3(frcnn_cdiv(OH * OW, 256)), dim3(256), 0, (hipStream_t)stream, img, H, W, C, pm, OH, OW,
                       1.0 / im_scale, 1.0 / im_scale, out);
    return frcnn_launch_status();
}


int frcnn_class_dets(const float *cls_prob, const float *pred_boxes, int R, int ncls, float *dets, void *stream) {
    if (!cls_prob || !pred_boxes || !dets || R < 0 || ncls < 2) return FRCNN_ERR_INVALID;
    if (R == 0) return FRCNN_OK;
    hipLaunchKernelGGL(class_dets_kernel, dim3(frcnn_cdiv((ncls - 1) * R, 256)), dim3(256), 0, (hipStream_t)stream, cls_prob, pred_boxes, R, ncls,
                       dets);
    return frcnn_launch_status();
}


int frcnn_bbox_transform_inv(const float *boxes, const float *deltas, int R, int ncls, float *pred_boxes, void *stream) {
    if (!boxes || !deltas || !pred_boxes || R < 0 || ncls < 1) return FRCNN_ERR_INVALID;
    if (R == 0) return FRCNN_OK;
    hipLaunchKernelGGL(head_decode_kernel, dim3(frcnn_cdiv(R * ncls, 256)), dim3(256), 0, (hipStream_t)stream, boxes, deltas, R, ncls, 0,
                       1, 1, pred_boxes);
    return frcnn_launch_status();
}

int frcnn_clip_boxes(float *boxes, int n_boxes, int im_h, int im_w, void *stream) {
    if (!boxes || n_boxes < 0) return FRCNN_ERR_INVALID;
    if (n_boxes == 0) return FRCNN_OK;
    hipLaunchKernelGGL(clip_boxes_kernel, dim3(frcnn_cdiv(n_boxes, 256)), dim3(256), 0, (hipStream_t)stream, boxes, n_boxes, im_h, im_w);
    return frcnn_launch_status();
}

int frcnn_softmax_rows(const float *scores, int R, int n, float *probs, void *stream) {
    if (!scores || !probs || R < 0 || n < 1) return FRCNN_ERR_INVALID;
    if (R == 0) return FRCNN_OK;
    hipLaunchKernelGGL(row_softmax_kernel, dim3(frcnn_cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, scores, R, n, probs);
    return frcnn_launch_status();
}

int frcnn_head_decode(const float *boxes, const float *deltas, const float *cls_score, int R, int ncls, int im_h, int im_w,
                      float *pred_boxes, float *cls_prob, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!boxes || !deltas || !cls_score || !pred_boxes || !cls_prob || R < 0 || ncls < 1) return FRCNN_ERR_INVALID;
    if (R == 0) return FRCNN_OK;
    hipLaunchKernelGGL(head_decode_kernel, dim3(frcnn_cdiv(R * ncls, 256)), dim3(256), 0, stream, boxes, deltas, R, ncls, 1, im_h, im_w,
                       pred_boxes);
    hipLaunchKernelGGL(row_softmax_kernel, dim3(frcnn_cdiv(R, 256)), dim3(256), 0, stream, cls_score, R, ncls, cls_prob);
    return frcnn_launch_status();
}

/* stacked form: `out` (R, ld) holds cls_score in columns [0, ncls) and bbox_pred from column dcol (dcol % 4 == 0, ld % 4 == 0) */
int frcnn_head_decode_stacked(const float *boxes, const float *out, int ld, int dcol, int R, int ncls, int im_h, int im_w,
                              float *pred_boxes, float *cls_prob, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!boxes || !out || !pred_boxes || !cls_prob || R < 0 || ncls < 1 || (ld % 4) != 0 || (dcol % 4) != 0 || dcol < ncls ||
        dcol + 4 * ncls > ld)
        return FRCNN_ERR_INVALID;
    if (R == 0) return FRCNN_OK;
    hipLaunchKernelGGL(head_decode_softmax_kernel, dim3(frcnn_cdiv(R * ncls, 256)), dim3(256), 0, stream, boxes, out, ld, dcol, R, ncls, im_h,
                       im_w, pred_boxes, cls_prob);
    return frcnn_launch_status();
}

}  // extern "C"
