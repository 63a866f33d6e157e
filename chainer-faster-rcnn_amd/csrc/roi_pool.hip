// roi_pool.hip -- RoIPooling2D forward / backward for gfx950 (HBM-bound: 35 MB algorithmic per image).
//
// Replaces chainer.functions.roi_pooling_2d as called at /root/reference/models/faster_rcnn.py:125-126
// (Chainer v1 ROIPooling2D; semantics restated in oracle/c/frcnn_oracle.c:oracle_roi_pool_fwd).
//
// Layout decision.  All C channels of one RoI share the same 7x7 bin geometry, so the natural wave
// mapping is lane = channel: every bin cell is then ONE fully coalesced read (64 lanes x VEC floats,
// uniform address + lane), there is no per-lane index arithmetic and no divergence.  That needs the
// feature map channel-last, so the forward runs in two launches:
//   chw_to_hwc_kernel   (C,H*W) -> (H*W,C) through a 64x65 LDS tile          4.9 MB in, 4.9 MB out
//   roi_pool_hwc_kernel block = (roi, 64*VEC channels), wave = output row ph; the 49 maxima per channel go
//                       through LDS ([channel][49], odd stride -> conflict-free) so the (R,C,7,7) result
//                       leaves as one contiguous, float4-coalesced 64*VEC*49-float run per block.
// The pipeline can skip the first launch by keeping conv5_3's output channel-last (frcnn_roi_pool_fwd_hwc).
//
// Bin edges replicate Chainer's CPU arithmetic exactly: round-half-even of the float32 product
// x*spatial_scale, then floor(p*stride) / ceil((p+1)*stride) with stride = (double)extent/out in IEEE
// double -- NOT integer division (7*(29/7.) = 29.000000000000004 -> ceil 30; see
// tests/test_oracle_pinned.py::test_roi_bin_edges_need_double_arithmetic).  Empty bins give 0 / argmax -1.
#include "frcnn_common.h"

namespace {

__global__ void __launch_bounds__(256)
chw_to_hwc_kernel(const float *__restrict__ x, float *__restrict__ xt, int C, int HW) {
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (c < C && p < HW) ? x[(size_t)c * HW + p] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int p = p0 + i, c = c0 + tx;
        if (p < HW && c < C) xt[(size_t)p * C + c] = tile[tx][i];
    }
}

constexpr int kBatch = 8;   // bin cells fetched per round

struct RoiGeom { int xs, ys, rw, rh; };

__device__ __forceinline__ RoiGeom roi_geometry(const float *__restrict__ roi, float scale) {
    // Python round() on the float32 product = round-half-to-even (rintf under the default rounding mode)
    const int xs = (int)rintf(roi[1] * scale), ys = (int)rintf(roi[2] * scale);
    const int xe = (int)rintf(roi[3] * scale), ye = (int)rintf(roi[4] * scale);
    RoiGeom g;
    g.xs = xs; g.ys = ys;
    g.rw = max(xe - xs + 1, 1);
    g.rh = max(ye - ys + 1, 1);
    return g;
}
__device__ __forceinline__ void bin_range(int p, int extent, int out, int offset, int limit, int &lo, int &hi) {
    const double stride = (double)extent / (double)out;
    lo = (int)floor((double)p * stride) + offset;
    hi = (int)ceil((double)(p + 1) * stride) + offset;
    lo = min(max(lo, 0), limit);
    hi = min(max(hi, 0), limit);
}

// One block = one RoI x (64*VEC) channels; OUTH waves, wave ph computes output row ph.
template <int VEC, bool ARGMAX>
__global__ void __launch_bounds__(448)
roi_pool_hwc_kernel(const float *__restrict__ xt, int C, int H, int W, const float *__restrict__ rois, int roi_cols, int outh,
                    int outw, float scale, float *__restrict__ y, int32_t *__restrict__ argmax) {
    constexpr int CB = 64 * VEC;
    __shared__ __attribute__((aligned(16))) float s_val[CB * 49];
    __shared__ int32_t s_idx[ARGMAX ? CB * 49 : 1];
    const int r = blockIdx.x, c0 = blockIdx.y * CB;
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int bins = outh * outw;
    // rows are [batch,x1,y1,x2,y2] (roi_cols 5) or bare [x1,y1,x2,y2] (roi_cols 4, ProposalLayer's output)
    const RoiGeom g = roi_geometry(rois + (size_t)roi_cols * r + (roi_cols - 5), scale);
    if (ph < outh) {
        int hs, he;
        bin_range(ph, g.rh, outh, g.ys, H, hs, he);
        const int cl = c0 + lane * VEC;
        for (int pw = 0; pw < outw; ++pw) {
            int ws, we;
            bin_range(pw, g.rw, outw, g.xs, W, ws, we);
            float m[VEC];
            int32_t mi[VEC];
            const bool empty = (he <= hs) || (we <= ws);
#pragma unroll
            for (int v = 0; v < VEC; ++v) { m[v] = 0.0f; mi[v] = -1; }
            if (!empty && cl < C) {
                // Cells of the bin in row-major order, kBatch independent loads in flight at a time (the wave
                // is latency-bound otherwise).  The tail re-reads the bin's last cell: with the strict `>`
                // below a duplicate can never displace the first maximum, so no predication is needed.
                const int bw = we - ws, ncell = (he - hs) * bw;
                const float *base = xt + cl;
                int hh = hs, ww = ws;                       // wave-uniform cursor
#pragma unroll
                for (int v = 0; v < VEC; ++v) { m[v] = 0.0f; mi[v] = hs * W + ws; }
                for (int i0 = 0; i0 < ncell; i0 += kBatch) {
                    float val[kBatch][VEC];
                    int pos[kBatch];
#pragma unroll
                    for (int q = 0; q < kBatch; ++q) {
                        pos[q] = hh * W + ww;
                        const float *src = base + (size_t)pos[q] * C;
                        if constexpr (VEC == 2) { const float2 t2 = *reinterpret_cast<const float2 *>(src); val[q][0] = t2.x; val[q][1] = t2.y; }
                        else val[q][0] = src[0];
                        if (i0 + q + 1 < ncell) { if (++ww == we) { ww = ws; ++hh; } }   // stay on the last cell past the end
                    }
#pragma unroll
                    for (int q = 0; q < kBatch; ++q)
#pragma unroll
                        for (int v = 0; v < VEC; ++v)
                            if ((i0 == 0 && q == 0) || val[q][v] > m[v]) { m[v] = val[q][v]; mi[v] = pos[q]; }   // first maximum wins
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                s_val[(lane * VEC + v) * bins + ph * outw + pw] = m[v];
                if (ARGMAX) s_idx[(lane * VEC + v) * bins + ph * outw + pw] = mi[v];
            }
        }
    }
    __syncthreads();
    // (r, c0 .. c0+CB, :, :) is one contiguous run of CB*bins floats in y
    const int nch = min(CB, C - c0);
    const int total = nch * bins;
    const size_t base = ((size_t)r * C + c0) * bins;
    if (((base | (size_t)total) & 3) == 0) {
        float4 *dst = reinterpret_cast<float4 *>(y + base);
        const float4 *srcv = reinterpret_cast<const float4 *>(s_val);
        for (int i = threadIdx.x; i < total / 4; i += blockDim.x) dst[i] = srcv[i];
    } else {
        for (int i = threadIdx.x; i < total; i += blockDim.x) y[base + i] = s_val[i];
    }
    if (ARGMAX)
        for (int i = threadIdx.x; i < total; i += blockDim.x) argmax[base + i] = s_idx[i];
}

// dx[c, argmax] += dy for every (roi, c, bin) with argmax >= 0 (Chainer backward_cpu).  fp32 atomics:
// the accumulation order differs from the reference's roi-major loop only in rounding.
__global__ void __launch_bounds__(256)
roi_pool_bwd_kernel(const float *__restrict__ dy, const int32_t *__restrict__ argmax, int C, int HW, int bins, size_t total,
                    float *__restrict__ dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int32_t a = argmax[i];
        if (a >= 0) {
            const int c = (int)((i / bins) % C);
            atomicAdd(&dx[(size_t)c * HW + a], dy[i]);
        }
    }
}

}  // namespace

extern "C" {

size_t frcnn_roi_pool_workspace_bytes(int C, int H, int W) {
    if (C < 1 || H < 1 || W < 1) return 0;
    return frcnn_align256((size_t)C * H * W * sizeof(float));
}

int frcnn_chw_to_hwc(const float *x, int C, int H, int W, float *xt, void *stream_) {
    if (!x || !xt || C < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    hipLaunchKernelGGL(chw_to_hwc_kernel, dim3(frcnn_cdiv(H * W, 64), frcnn_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream_, x,
                       xt, C, H * W);
    return frcnn_launch_status();
}

int frcnn_roi_pool_fwd_hwc(const float *xt, int C, int H, int W, const float *rois, int R, int roi_cols, int outh, int outw,
                           float spatial_scale, float *y, int32_t *argmax, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!xt || !rois || !y || C < 1 || H < 1 || W < 1 || R < 0) return FRCNN_ERR_INVALID;
    if (outh < 1 || outw < 1 || outh > 7 || outw > 7 || (roi_cols != 4 && roi_cols != 5)) return FRCNN_ERR_INVALID;   // LDS tile is sized for <= 7x7 bins
    if (R == 0) return FRCNN_OK;
    const dim3 blk(64 * outh);
    if (C % 128 == 0) {
        const dim3 grid(R, C / 128);
        if (argmax) hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_hwc_kernel<2, true>), grid, blk, 0, stream, xt, C, H, W, rois, roi_cols, outh, outw, spatial_scale, y, argmax);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_hwc_kernel<2, false>), grid, blk, 0, stream, xt, C, H, W, rois, roi_cols, outh, outw, spatial_scale, y, argmax);
    } else {
        const dim3 grid(R, frcnn_cdiv(C, 64));
        if (argmax) hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_hwc_kernel<1, true>), grid, blk, 0, stream, xt, C, H, W, rois, roi_cols, outh, outw, spatial_scale, y, argmax);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_hwc_kernel<1, false>), grid, blk, 0, stream, xt, C, H, W, rois, roi_cols, outh, outw, spatial_scale, y, argmax);
    }
    return frcnn_launch_status();
}

int frcnn_roi_pool_fwd(const float *x, int C, int H, int W, const float *rois, int R, int outh, int outw, float spatial_scale,
                       float *y, int32_t *argmax, void *workspace, size_t workspace_bytes, void *stream) {
    if (!workspace || workspace_bytes < frcnn_roi_pool_workspace_bytes(C, H, W)) return FRCNN_ERR_INVALID;
    int st = frcnn_chw_to_hwc(x, C, H, W, (float *)workspace, stream);
    if (st != FRCNN_OK) return st;
    return frcnn_roi_pool_fwd_hwc((const float *)workspace, C, H, W, rois, R, 5, outh, outw, spatial_scale, y, argmax, stream);
}

int frcnn_roi_pool_bwd(const float *dy, const int32_t *argmax, int R, int C, int H, int W, int outh, int outw, float *dx,
                       void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dy || !argmax || !dx || R < 0 || C < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    FRCNN_HIP_TRY(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)C * H * W, stream));
    const size_t total = (size_t)R * C * outh * outw;
    if (total == 0) return FRCNN_OK;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dy, argmax, C, H * W, outh * outw, total, dx);
    return frcnn_launch_status();
}

}  // extern "C"
