// train.hip -- the RPN training step's kernels for gfx950.
//
// Replaces, on the device:
//   AnchorTargetLayer.__call__ / _create_bbox_labels / _calc_overlaps   /root/reference/models/anchor_target_layer.py:66-198
//   keep_inside, bbox_transform                                         models/bbox_transform.py:18-38,112-130
//   bbox_overlaps (float64 IoU matrix)                                  models/bbox.pyx:16-56
//   _calc_rpn_loss_cls / _calc_rpn_loss_bbox + their gradients          models/region_proposal_network.py:160-204
//   backward of L.Convolution2D (weight + bias gradients), F.max_pooling_2d  (Chainer v1 semantics)
//   MomentumSGD + WeightDecay                                           train_rpn.py:165-167
// The random fg/bg subsample of the labels (anchor_target_layer.py:147-167) stays on the host: it draws from NumPy's
// global RNG and the draws must be reproducible seed for seed (models/anchor_target_layer.py in the host package).
// The input gradient of a convolution is the forward MFMA kernel (conv.hip) run on re-packed weights
// (frcnn_pack_conv3x3_dgrad_w) with the producing ReLU's mask fused into its epilogue.
//
// Arithmetic parity: float64 where the reference is float64 (anchors, IoU, targets), float32 where it is float32
// (the ground-truth side of bbox_transform), no FMA contraction (built with -ffp-contract=off).
#include "frcnn_common.h"
#include <string.h>
#include <stdlib.h>
#include <frcnn_buffer.h>   // angle brackets: shadowed by the test emulator
#include <frcnn_intrin.h>
#include "frcnn_reduce.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct AnchorsD { double a[32][4]; };

// all_bbox[i] = anchors[a] + shift(k), i = k*A + a, k = h*W + w  (proposal_layer.py:207-221, kept in float64)
__device__ __forceinline__ void anchor_box(const AnchorsD &anc, int i, int A, int W, int stride, double &x1, double &y1, double &x2, double &y2) {
    const int k = i / A, a = i - k * A;
    const int h = k / W, w = k - h * W;
    const double sx = (double)(w * stride), sy = (double)(h * stride);
    x1 = anc.a[a][0] + sx; y1 = anc.a[a][1] + sy; x2 = anc.a[a][2] + sx; y2 = anc.a[a][3] + sy;
}

// keep_inside (bbox_transform.py:124-129): ascending indices of the anchors that lie completely inside the image.
// One workgroup; ordered compaction = wave ballot + per-wave offsets.
__global__ void __launch_bounds__(1024)
atl_inside_kernel(AnchorsD anc, int A, int H, int W, int stride, int im_h, int im_w, int32_t *__restrict__ inds, int32_t *__restrict__ n_inside) {
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int n_all = A * H * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_all; i0 += 1024) {
        const int i = i0 + tid;
        int in = 0;
        if (i < n_all) {
            double x1, y1, x2, y2;
            anchor_box(anc, i, A, W, stride, x1, y1, x2, y2);
            in = (x1 >= 0.0) && (y1 >= 0.0) && (x2 < (double)im_w) && (y2 < (double)im_h);
        }
        const unsigned long long bal = __ballot(in);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (in) inds[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_cnt[w]; base_s += t; }
        __syncthreads();
    }
    if (tid == 0) n_inside[0] = base_s;
}

// bbox.pyx:24-55 for one (box, query) pair
__device__ __forceinline__ double iou_f64(double bx1, double by1, double bx2, double by2, double qx1, double qy1, double qx2, double qy2) {
    const double box_area = (qx2 - qx1 + 1) * (qy2 - qy1 + 1);
    const double iw = (bx2 < qx2 ? bx2 : qx2) - (bx1 > qx1 ? bx1 : qx1) + 1;
    if (iw > 0) {
        const double ih = (by2 < qy2 ? by2 : qy2) - (by1 > qy1 ? by1 : qy1) + 1;
        if (ih > 0) {
            const double ua = (bx2 - bx1 + 1) * (by2 - by1 + 1) + box_area - iw * ih;
            return iw * ih / ua;
        }
    }
    return 0.0;
}

__global__ void __launch_bounds__(256)
bbox_overlaps_kernel(const double *__restrict__ boxes, int N, const double *__restrict__ query, int K, double *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)N * K) return;
    const int n = (int)(t / K), k = (int)(t - (size_t)n * K);
    out[t] = iou_f64(boxes[4 * n], boxes[4 * n + 1], boxes[4 * n + 2], boxes[4 * n + 3], query[4 * k], query[4 * k + 1], query[4 * k + 2],
                     query[4 * k + 3]);
}

// overlaps of every inside anchor with every gt box + row max / first argmax (anchor_target_layer.py:183-193)
__global__ void __launch_bounds__(256)
atl_overlap_kernel(AnchorsD anc, int A, int W, int stride, const int32_t *__restrict__ inds, const int32_t *__restrict__ n_inside,
                   const float *__restrict__ gt, int G, double *__restrict__ overlaps, double *__restrict__ max_ov, int32_t *__restrict__ argmax) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_inside[0]) return;
    double x1, y1, x2, y2;
    anchor_box(anc, inds[j], A, W, stride, x1, y1, x2, y2);
    double best = 0.0;
    int bi = 0;
    for (int g = 0; g < G; ++g) {
        const double o = iou_f64(x1, y1, x2, y2, (double)gt[5 * g], (double)gt[5 * g + 1], (double)gt[5 * g + 2], (double)gt[5 * g + 3]);
        overlaps[(size_t)j * G + g] = o;
        if (g == 0 || o > best) { best = o; bi = g; }      // numpy argmax: first maximum
    }
    max_ov[j] = best;
    argmax[j] = bi;
}

// column maxima: gt_max_overlaps (anchor_target_layer.py:190-195); one workgroup per gt box
__global__ void __launch_bounds__(256)
atl_gtmax_kernel(const double *__restrict__ overlaps, const int32_t *__restrict__ n_inside, int G, double *__restrict__ gt_max) {
    __shared__ double red[256];
    const int g = blockIdx.x, n = n_inside[0];
    double m = -1.0;
    for (int j = threadIdx.x; j < n; j += 256) { const double o = overlaps[(size_t)j * G + g]; m = o > m ? o : m; }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x + s] > red[threadIdx.x] ? red[threadIdx.x + s] : red[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) gt_max[g] = red[0];
}

// labels before the random subsample (anchor_target_layer.py:129-145) and regression targets (:113-118)
__global__ void __launch_bounds__(256)
atl_label_kernel(AnchorsD anc, int A, int W, int stride, const int32_t *__restrict__ inds, const int32_t *__restrict__ n_inside,
                 const float *__restrict__ gt, int G, const double *__restrict__ overlaps, const double *__restrict__ max_ov,
                 const int32_t *__restrict__ argmax, const double *__restrict__ gt_max, double neg_thr, double pos_thr,
                 int32_t *__restrict__ labels, float *__restrict__ targets) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_inside[0]) return;
    bool is_gt_argmax = false;                                    // xp.where(overlaps == gt_max_overlaps)[0]: ties included
    for (int g = 0; g < G; ++g) is_gt_argmax |= overlaps[(size_t)j * G + g] == gt_max[g];
    const double mo = max_ov[j];
    int lab = -1;
    if (mo < neg_thr) lab = 0;
    if (is_gt_argmax) lab = 1;
    if (mo >= pos_thr) lab = 1;
    if (mo < neg_thr) lab = 0;                                    // negatives clobber positives (:144-145)
    labels[j] = lab;
    // bbox_transform(ex = float64 anchors, gt = float32 rows): the gt side is float32 arithmetic, then promoted
    double x1, y1, x2, y2;
    anchor_box(anc, inds[j], A, W, stride, x1, y1, x2, y2);
    const float *q = gt + 5 * argmax[j];
    const double ew = x2 - x1 + 1.0, eh = y2 - y1 + 1.0;
    const double ecx = x1 + 0.5 * ew, ecy = y1 + 0.5 * eh;
    const float gw = q[2] - q[0] + 1.0f, gh = q[3] - q[1] + 1.0f;
    const float gcx = q[0] + 0.5f * gw, gcy = q[1] + 0.5f * gh;
    float4 t;
    t.x = (float)(((double)gcx - ecx) / ew);
    t.y = (float)(((double)gcy - ecy) / eh);
    t.z = (float)log((double)gw / ew);
    t.w = (float)log((double)gh / eh);
    reinterpret_cast<float4 *>(targets)[j] = t;
}

// ------------------------------------------------------------------------------------------------
// RPN losses and their gradients, one workgroup (the sums are over <= A*H*W anchors; a fixed reduction tree
// keeps the result bit-reproducible).  out[0] = rpn_loss_cls, out[1] = rpn_loss_bbox, out[2] = rpn_cls_accuracy.
__global__ void __launch_bounds__(1024)
rpn_loss_kernel(const float *__restrict__ score, const float *__restrict__ bbox_pred, const int32_t *__restrict__ labels,
                const float *__restrict__ targets, const int32_t *__restrict__ inds, int n_in, int A, int HW, float delta, float lambda,
                float *__restrict__ out, float *__restrict__ dscore, float *__restrict__ dbbox) {
    __shared__ double red[4][1024];
    const int tid = threadIdx.x;
    const int n_all = A * HW;
    double s_cls = 0.0, s_box = 0.0, s_acc = 0.0, s_cnt = 0.0;
    for (int j = tid; j < n_in; j += 1024) {
        const int idx = inds[j], lab = labels[j];
        const int k = idx / A, a = idx - k * A;
        // F.softmax_cross_entropy on score.reshape(1,2,A,H,W): classes = channels a (bg) and A+a (fg) (:175-178)
        const float s0 = score[(size_t)a * HW + k], s1 = score[(size_t)(A + a) * HW + k];
        if (lab != -1) {
            const float m = fmaxf(s0, s1);
            const float logz = m + logf(expf(s0 - m) + expf(s1 - m));
            s_cls += (double)(logz - (lab == 1 ? s1 : s0));
            s_cnt += 1.0;
            s_acc += ((s1 > s0 ? 1 : 0) == lab) ? 1.0 : 0.0;
        }
        // Huber over ALL inside anchors; channel = coord*A + a (:187-201 -- the reference's own re-interpretation)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d = bbox_pred[(size_t)(c * A + a) * HW + k] - targets[4 * j + c];
            const float ad = fabsf(d);
            s_box += (double)(ad < delta ? 0.5f * d * d : delta * (ad - 0.5f * delta));
        }
    }
    red[0][tid] = s_cls; red[1][tid] = s_box; red[2][tid] = s_acc; red[3][tid] = s_cnt;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][tid] += red[q][tid + s];
        }
        __syncthreads();
    }
    const double cnt = red[3][0] > 1.0 ? red[3][0] : 1.0;
    if (tid == 0) {
        out[0] = (float)(red[0][0] / cnt);
        out[1] = (float)(red[1][0] / (double)n_all);
        out[2] = (float)(red[2][0] / cnt);
    }
    if (!dscore || !dbbox) return;
    const float inv_cnt = (float)(1.0 / cnt), inv_all = lambda / (float)n_all;
    for (int j = tid; j < n_in; j += 1024) {
        const int idx = inds[j], lab = labels[j];
        const int k = idx / A, a = idx - k * A;
        if (lab != -1) {
            const float s0 = score[(size_t)a * HW + k], s1 = score[(size_t)(A + a) * HW + k];
            const float m = fmaxf(s0, s1);
            const float e0 = expf(s0 - m), e1 = expf(s1 - m);
            const float z = e0 + e1;
            dscore[(size_t)a * HW + k] = (e0 / z - (lab == 0 ? 1.0f : 0.0f)) * inv_cnt;
            dscore[(size_t)(A + a) * HW + k] = (e1 / z - (lab == 1 ? 1.0f : 0.0f)) * inv_cnt;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d = bbox_pred[(size_t)(c * A + a) * HW + k] - targets[4 * j + c];
            const float g = fabsf(d) < delta ? d : (d > 0.0f ? delta : -delta);
            dbbox[(size_t)(c * A + a) * HW + k] = g * inv_all;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// F.max_pooling_2d(2, 2) backward, gather form: an input cell receives the window's gradient iff it is the FIRST
// maximum of its window in (ky, kx) scan order (Chainer's im2col argmax); windows do not overlap, so no atomics.
// One thread per WINDOW: its four cells are read once (the cell-per-thread form read the window four times and spent three 64-bit
// divisions per cell), its gradient once, its four outputs written as two pairs.
__global__ void __launch_bounds__(256)
maxpool2x2_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx, int C, int H, int W, int OH, int OW) {
    const uint32_t total = (uint32_t)C * OH * OW;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t ow = i % (uint32_t)OW, t = i / (uint32_t)OW, oh = t % (uint32_t)OH, c = t / (uint32_t)OH;
        const size_t base = ((size_t)c * H + 2 * oh) * W + 2 * ow;
        const float *p = x + base;
        const bool hasx = 2 * ow + 1 < (uint32_t)W, hasy = 2 * oh + 1 < (uint32_t)H;
        const float v0 = p[0], v1 = hasx ? p[1] : 0.0f, v2 = hasy ? p[W] : 0.0f, v3 = (hasx && hasy) ? p[W + 1] : 0.0f;
        float m = v0;
        int arg = 0;
        if (hasx && v1 > m) { m = v1; arg = 1; }
        if (hasy && v2 > m) { m = v2; arg = 2; }
        if (hasx && hasy && v3 > m) { m = v3; arg = 3; }
        const float g = dy[i];
        float *d = dx + base;
        d[0] = arg == 0 ? g : 0.0f;
        if (hasx) d[1] = arg == 1 ? g : 0.0f;
        if (hasy) d[W] = arg == 2 ? g : 0.0f;
        if (hasx && hasy) d[W + 1] = arg == 3 ? g : 0.0f;
    }
}

// the same routing from the arg-max bytes the fused conv + ReLU + pool epilogue wrote (conv.hip, act 5): no pre-pool map to read
__global__ void __launch_bounds__(256)
maxpool2x2_bwd_idx_kernel(const unsigned char *__restrict__ idx, const float *__restrict__ dy, float *__restrict__ dx, int C, int H, int W, int OH, int OW) {
    const uint32_t total = (uint32_t)C * OH * OW;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t ow = i % (uint32_t)OW, t = i / (uint32_t)OW, oh = t % (uint32_t)OH, c = t / (uint32_t)OH;
        const size_t base = ((size_t)c * H + 2 * oh) * W + 2 * ow;
        const bool hasx = 2 * ow + 1 < (uint32_t)W, hasy = 2 * oh + 1 < (uint32_t)H;
        const int arg = idx[i] & 3;                                  // (bit 2 of the byte is the ReLU mask of the layer above, read by conv.hip's act 6)
        const float g = dy[i];
        float *d = dx + base;
        d[0] = arg == 0 ? g : 0.0f;
        if (hasx) d[1] = arg == 1 ? g : 0.0f;
        if (hasy) d[W] = arg == 2 ? g : 0.0f;
        if (hasx && hasy) d[W + 1] = arg == 3 ? g : 0.0f;
    }
}

// db[c] = sum over pixels of dy[c][:].  Two fixed-shape stages (deterministic): `parts` workgroups per channel each reduce a
// contiguous slice (float4 loads), then one wave per channel adds the partials in order.
__global__ void __launch_bounds__(256)
bias_grad_partial_kernel(const float *__restrict__ dy, int HW, int parts, float *__restrict__ partial) {
    __shared__ float red[256];
    const int c = blockIdx.x, part = blockIdx.y;
    const int per = (HW + parts - 1) / parts;
    const int begin = part * per, end = min(HW, begin + per);
    const float *p = dy + (size_t)c * HW;
    float s = 0.0f;
    // float4 loads over the 16-byte aligned middle of the slice, scalars at its ragged ends (a thread's partial sum is then added
    // in a fixed order: still deterministic, though not the scalar loop's order)
    const uintptr_t addr = reinterpret_cast<uintptr_t>(p + begin);
    int head = (int)(((16 - (addr & 15)) & 15) >> 2);
    if (head > end - begin) head = end - begin;
    const int n4 = (end - begin - head) >> 2;
    if ((int)threadIdx.x < head) s += p[begin + threadIdx.x];
    const float4 *p4 = reinterpret_cast<const float4 *>(p + begin + head);
    // four independent 16-byte loads in flight per thread and trip (round 5: one load per trip left the kernel waiting for a memory round trip per 4 KB
    // and workgroup -- 32 us per layer on average next to the input-gradient convolutions); fixed order: four running sums per thread, added at the end
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int i = threadIdx.x;
    for (; i + 768 < n4; i += 1024) {
        const float4 v0 = p4[i], v1 = p4[i + 256], v2 = p4[i + 512], v3 = p4[i + 768];
        s0 += (v0.x + v0.y) + (v0.z + v0.w);
        s1 += (v1.x + v1.y) + (v1.z + v1.w);
        s2 += (v2.x + v2.y) + (v2.z + v2.w);
        s3 += (v3.x + v3.y) + (v3.z + v3.w);
    }
    for (; i < n4; i += 256) {
        const float4 v = p4[i];
        s0 += (v.x + v.y) + (v.z + v.w);
    }
    s += (s0 + s1) + (s2 + s3);
    for (int i = begin + head + 4 * n4 + threadIdx.x; i < end; i += 256) s += p[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int q = 128; q > 0; q >>= 1) {
        if (threadIdx.x < q) red[threadIdx.x] += red[threadIdx.x + q];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)c * parts + part] = red[0];
}

__global__ void __launch_bounds__(256)
bias_grad_final_kernel(const float *__restrict__ partial, int C, int parts, float *__restrict__ db) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    db[c] = frcnn_sum_splits(partial + (size_t)c * parts, 1, 0, parts);          // in part order, loads in batches of eight (frcnn_reduce.h)
}

// forward-packed (Cin*9, Cout) -> the packed weights of the input-gradient convolution: (Cout*9, Cin) with the taps
// rotated by 180 degrees: wd[(co*9 + t)][ci] = wp[(ci*9 + 8 - t)][co]
__global__ void __launch_bounds__(256)
pack_dgrad_w_kernel(const float *__restrict__ wp, int Cin, int Cout, int taps, float *__restrict__ wd) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)Cin * Cout * taps;
    if (i >= total) return;
    const int ci = (int)(i % Cin);
    const int t = (int)((i / Cin) % taps), co = (int)(i / ((size_t)Cin * taps));
    wd[i] = wp[((size_t)ci * taps + (taps - 1 - t)) * Cout + co];
}

// the same for up to 16 layers in ONE launch (a training step re-packs every layer's weights after the optimizer has moved them:
// 14 launches of 3-17 us each; worth 0.03 ms of the step); block b belongs to the first layer whose block_end exceeds it
struct DgradPackArgs {
    const float *wp[16];
    float *wd[16];
    int cin[16], cout[16], taps[16];
    unsigned block_end[16];
    int n;
};
__global__ void __launch_bounds__(256)
pack_dgrad_w_many_kernel(DgradPackArgs a) {
    // per (layer, tap) this is a (Cin, Cout) -> (Cout, Cin) transpose: 64 x 64 tiles through LDS, both sides coalesced (as one element per
    // thread with the source index computed from the destination's, every read touched its own cache line: 87 us for the 14 layers)
    __shared__ float tile[64][65];
    int li = 0;
    while (li + 1 < a.n && blockIdx.x >= a.block_end[li]) ++li;
    const unsigned first = li == 0 ? 0u : a.block_end[li - 1];
    const int Cin = a.cin[li], Cout = a.cout[li], taps = a.taps[li];
    const int cit = (Cin + 63) / 64, cot = (Cout + 63) / 64;
    const unsigned u = blockIdx.x - first;                       // (tap, ci tile, co tile)
    const int t = (int)(u / (unsigned)(cit * cot)), rem = (int)(u % (unsigned)(cit * cot));
    const int ci0 = (rem / cot) * 64, co0 = (rem % cot) * 64;
    const float *wp = a.wp[li];
    float *wd = a.wd[li];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {                              // source rows: (ci, rotated tap), columns co
        const int ci = ci0 + i, co = co0 + tx;
        tile[i][tx] = (ci < Cin && co < Cout) ? wp[((size_t)ci * taps + (taps - 1 - t)) * Cout + co] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {                              // destination rows: (co, tap), columns ci
        const int co = co0 + i, ci = ci0 + tx;
        if (co < Cout && ci < Cin) wd[((size_t)co * taps + t) * Cin + ci] = tile[tx][i];
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a KS x KS / stride 1 / pad KS/2 convolution on v_mfma_f32_32x32x2_f32:
//   dWp[(ci*T + tap)][co] = sum over pixels p of x[ci][p + offset(tap)] * dy[co][p]         (T = KS*KS)
// GEMM with the PIXELS as the reduction axis.  MFMA A = activations: lane l supplies x[ci0 + (l&31)][pixel (l>>5)]
// for one tap; B = output gradients: lane l supplies dy[co0 + (l&31)][pixel (l>>5)]; D[ci][co] per tap, so a wave
// holds T accumulators for a 32 ci x 32 co block and every LDS fragment read has the 32 lanes on 32 different
// channels at one pixel -- conflict-free because the per-channel pitches are odd.  A workgroup (4 waves = 2 ci x 2 co
// blocks) walks its share of the image in tiles of WG_ROWS rows x 32 px, staging the tile's halo of 64 input channels
// and 64 output-gradient channels in LDS (global reads through buffer descriptors: zero padding for free).
// The pixel range of the image is split over `splits` workgroups per (ci, co) tile; every split writes its partial
// dWp tile to its own slab and wgrad_reduce_kernel adds the slabs in a fixed order (deterministic).
constexpr int WG_ROWS = 2;

template <int KS>
__global__ void __launch_bounds__(256)
conv_wgrad_mfma_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ slabs, int Cin, int Cout, int H, int W,
                       int xtiles, int nblocks, int splits) {
    constexpr int T = KS * KS, PAD = KS / 2;
    constexpr int HP = 32 + KS - 1;                        // halo row pitch
    constexpr int HR = WG_ROWS + KS - 1;
    constexpr int CHP = HR * HP + ((HR * HP) % 2 == 0 ? 1 : 0);     // per-channel pitch, odd
    constexpr int DP = WG_ROWS * 32 + 1;                   // dy per-channel pitch, odd
    __shared__ float x_lds[64 * CHP];
    __shared__ float dy_lds[64 * DP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wci = wave & 1, wco = wave >> 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int HWs = H * W;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, split = blockIdx.z;
    const int b_begin = (int)((long long)split * nblocks / splits), b_end = (int)((long long)(split + 1) * nblocks / splits);
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)Cin * HWs * sizeof(float)));
    const frcnn_buf_t dbuf = frcnn_make_buf(dy, (uint32_t)((size_t)Cout * HWs * sizeof(float)));

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // Staging elements of this thread: fixed (channel, halo row, halo column) / (channel, row, pixel) slots, so only the
    // image-border test and one add depend on the tile.  Register-staged double buffer: tile b+1 is fetched while tile b
    // feeds the MFMAs and written to LDS after the barrier.
    constexpr int XN = 64 * HR * HP, DN = 64 * WG_ROWS * 32;
    constexpr int XIT = (XN + 255) / 256, DIT = (DN + 255) / 256;
    float xreg[XIT], dreg[DIT];
    auto fetch = [&](int blk) {
        const int tx = blk % xtiles, ty = blk / xtiles;
        const int x0 = tx * 32, y0 = ty * WG_ROWS;
#pragma unroll
        for (int q = 0; q < XIT; ++q) {
            const int e = tid + q * 256;
            const int c = e / (HR * HP), rem = e - c * (HR * HP);
            const int hr = rem / HP, hx = rem - hr * HP;
            const int gy = y0 - PAD + hr, gx = x0 - PAD + hx, gc = ci0 + c;
            const bool inside = e < XN && gc < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W;
            xreg[q] = frcnn_buf_load_f32(xbuf, inside ? (uint32_t)(gc * HWs + gy * W + gx) * 4u : kBufOob);
        }
#pragma unroll
        for (int q = 0; q < DIT; ++q) {
            const int e = tid + q * 256;
            const int c = e / (WG_ROWS * 32), rem = e - c * (WG_ROWS * 32);
            const int gy = y0 + (rem >> 5), gx = x0 + (rem & 31), gc = co0 + c;
            const bool inside = gc < Cout && gy < H && gx < W;
            dreg[q] = frcnn_buf_load_f32(dbuf, inside ? (uint32_t)(gc * HWs + gy * W + gx) * 4u : kBufOob);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < XIT; ++q) {
            const int e = tid + q * 256;
            if (e < XN) { const int c = e / (HR * HP); x_lds[c * CHP + (e - c * (HR * HP))] = xreg[q]; }
        }
#pragma unroll
        for (int q = 0; q < DIT; ++q) {
            const int e = tid + q * 256;
            const int c = e / (WG_ROWS * 32);
            dy_lds[c * DP + (e - c * (WG_ROWS * 32))] = dreg[q];
        }
    };
    if (b_begin < b_end) fetch(b_begin);
    for (int b = b_begin; b < b_end; ++b) {
        __syncthreads();                       // every wave is done reading the previous tile
        stage();
        __syncthreads();
        if (b + 1 < b_end) fetch(b + 1);       // in flight during the MFMAs below
        const float *xa = x_lds + (wci * 32 + l31) * CHP + khalf;
        const float *db = dy_lds + (wco * 32 + l31) * DP + khalf;
#pragma unroll 1
        for (int r = 0; r < WG_ROWS; ++r) {
#pragma unroll 4
            for (int pp = 0; pp < 16; ++pp) {
                const float bv = db[r * 32 + 2 * pp];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int ky = t / KS, kx = t % KS;
                    const float av = xa[(r + ky) * HP + 2 * pp + kx];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial tile -> slab `split`, in dWp's own layout: row (ci*T + tap), column co
    float *slab = slabs + (size_t)split * ((size_t)Cin * T * Cout);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            const int co = co0 + wco * 32 + l31;
            if (ci < Cin && co < Cout) slab[((size_t)ci * T + t) * Cout + co] = acc[t][r];
        }
}

// LDS-DMA form of the same kernel.  The register-staged kernel above holds 144 accumulator registers AND a 50-register staging set
// AND its address arithmetic: 384 VGPRs, ONE wave per SIMD, nothing to hide a load or an LDS round trip behind.  Here a tile's
// rows go from L2 straight into the same padded LDS images (buffer_load_dword ... lds), one DMA per (channel, halo row): the row
// base is a scalar offset, the per-lane offset (column, or out-of-range for padding / beyond the image) is fixed for the tile, so a
// DMA costs a handful of SALU operations and no VALU -- and no staging registers: two workgroups per CU, one's MFMAs over the
// other's loads, plus fragment reads software-pipelined one step ahead of the MFMAs.
// DB = double-buffered LDS images (103 KB: one workgroup per CU, one wave per SIMD): tile b+1's DMAs are issued right after the
// barrier that hands over tile b and land under tile b's 288 MFMAs per wave.  Picked per layer (see wgrad_double_buffered()).
// One fence-less barrier per tile, no counted waits (a wave issues 80 DMAs per tile,
// more than vmcnt can count: the wait for tile b+1 is the vmcnt(0) at the top of the next trip, a whole compute phase later).
// ABL (timing ablations, WRONG results, only in -DFRCNN_TIMING_ABLATIONS builds): 1 no DMA, 4 no MFMA phase, 8 no slab stores.
// (Measured in round 3 and removed again, DESIGN 3.6 / 3.11: three single-buffer workgroups per CU at 168 VGPRs -- +-1 %; wave priorities
// and a staggered start by hardware wave slot -- no effect: a wave with an fp32 MFMA ready blocks the SIMD's issue stage whatever the priorities.)
template <int KS, bool DB = false, int ABL = 0>
__global__ void __launch_bounds__(256, DB ? 1 : 2)
conv_wgrad_dma_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ slabs, int Cin, int Cout, int H, int W,
                      int xtiles, int nblocks, int splits) {
    constexpr int T = KS * KS, PAD = KS / 2;
    constexpr int HP = 32 + KS - 1;
    constexpr int HR = WG_ROWS + KS - 1;
    constexpr int CHP = HR * HP + ((HR * HP) % 2 == 0 ? 1 : 0);
    constexpr int DP = WG_ROWS * 32 + 1;
    constexpr int NB = DB ? 2 : 1;
    __shared__ float x_lds[NB][64 * CHP];
    __shared__ float dy_lds[NB][64 * DP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wci = wave & 1, wco = wave >> 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int HWs = H * W;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, split = blockIdx.z;
    const int b_begin = (int)((long long)split * nblocks / splits), b_end = (int)((long long)(split + 1) * nblocks / splits);
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)Cin * HWs * sizeof(float)));
    const frcnn_buf_t dbuf = frcnn_make_buf(dy, (uint32_t)((size_t)Cout * HWs * sizeof(float)));

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    static_assert(WG_ROWS == 2, "a dy piece is the two 32-pixel rows of one channel");
    // DMA of tile b into LDS image `buf`: wave w moves channels w, w+4, ... (16 channels: HR x rows + 1 dy piece each)
    // DMA of a tile into LDS image `buf`: wave w moves channels w, w+4, ... (16 channels: HR x rows + 1 dy piece each).  issue_setup(b)
    // fixes the per-lane parts for tile b, issue_channel(q, buf) moves the wave's q-th channel, issue(b, buf) = all of it at once.
    int t_x0 = 0, t_y0 = 0, t_xs = 0;
    uint32_t t_vx = kBufOob, t_vd = kBufOob;
    auto issue_setup = [&](int b) {
        const int tx = b % xtiles, ty = b / xtiles;
        t_x0 = tx * 32; t_y0 = ty * WG_ROWS;
        // per-lane parts, fixed for the tile.  x piece = one halo row of one channel (HP floats: lanes >= HP sit it out), columns
        // x0-PAD .. x0+32+PAD-1; the scalar row base points at column xs = max(x0-PAD, 0) so no offset is ever negative.
        t_xs = t_x0 - PAD > 0 ? t_x0 - PAD : 0;
        const int gx = t_x0 - PAD + lane;
        t_vx = (lane < HP && gx >= 0 && gx < W) ? (uint32_t)(gx - t_xs) * 4u : kBufOob;
        // dy piece = both rows of one channel (64 floats): lane -> (row lane>>5, column lane&31)
        const int dgy = t_y0 + (lane >> 5), dgx = t_x0 + (lane & 31);
        t_vd = (dgy < H && dgx < W) ? (uint32_t)((lane >> 5) * W + (lane & 31)) * 4u : kBufOob;
    };
    // a channel's HR x pieces under ONE exec mask (lanes < HP), then its dy piece with every lane: as `if (lane < HP) dma` per piece
    // every piece paid an exec save / branch / restore of its own.  (The scalar row offsets are made uniform OUTSIDE the masked
    // region: a wave collective inside it would not see every lane.)
    auto issue_channel = [&](int q, int buf) __attribute__((always_inline)) {
        if constexpr ((ABL & 1) != 0) return;
        const int c = wave + 4 * q;
        const int gc = ci0 + c, gco = co0 + c;
        uint32_t so[HR];
        bool row_ok[HR];
#pragma unroll
        for (int hr = 0; hr < HR; ++hr) {
            const int gy = t_y0 - PAD + hr;
            row_ok[hr] = gc < Cin && gy >= 0 && gy < H;                       // wave-uniform
            so[hr] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(row_ok[hr] ? (uint32_t)(((size_t)gc * H + gy) * W + t_xs) * 4u : 0u));
        }
        if (lane < HP) {
#pragma unroll
            for (int hr = 0; hr < HR; ++hr)
                frcnn_buf_load_lds_b32(xbuf, &x_lds[buf][c * CHP + hr * HP], row_ok[hr] ? t_vx : kBufOob, so[hr]);
        }
        const bool ch_ok = gco < Cout && t_y0 < H;
        const uint32_t sd = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ch_ok ? (uint32_t)(((size_t)gco * H + t_y0) * W + t_x0) * 4u : 0u));
        frcnn_buf_load_lds_b32(dbuf, &dy_lds[buf][c * DP], ch_ok ? t_vd : kBufOob, sd);
    };
    // Lean form for workgroups whose 64 + 64 channels all exist (every VGG layer but conv1_1): nothing per piece but the LDS address, the
    // scalar offset and the DMA itself.  Row validity is folded into one per-lane offset per halo row, once per tile (an invalid row's
    // lanes are all out of range, so its scalar offset may be anything); the x pieces run under ONE exec mask, the dy pieces after them.
    // (The general form spends ~11 instructions, 1.6 branches and a v_cndmask per piece; the issue stream is what the DMA phase costs.)
    const bool full_tile = ci0 + 64 <= Cin && co0 + 64 <= Cout;
    auto issue_lean = [&](int b, int buf) {
        if constexpr ((ABL & 1) != 0) return;
        issue_setup(b);
        uint32_t vxr[HR];
#pragma unroll
        for (int hr = 0; hr < HR; ++hr) {
            const int gy = t_y0 - PAD + hr;
            vxr[hr] = (gy >= 0 && gy < H) ? t_vx : kBufOob;
        }
        const uint32_t chan_bytes = (uint32_t)HWs * 4u;
        // 32-bit wrap-around is fine: a row above the image has no in-range lane, and row hr >= 1 adds hr * W * 4 back
        uint32_t sx = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(((ci0 + wave) * H + t_y0 - PAD) * W + t_xs) * 4u));
        uint32_t sd = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(((co0 + wave) * H + t_y0) * W + t_x0) * 4u));
        float *xl = &x_lds[buf][wave * CHP];
        float *dl = &dy_lds[buf][wave * DP];
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            uint32_t so[HR];                                       // made uniform OUTSIDE the masked region (see issue_channel)
#pragma unroll
            for (int hr = 0; hr < HR; ++hr) so[hr] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(sx + (uint32_t)(q * 4) * chan_bytes + (uint32_t)(hr * W) * 4u));
            if (lane < HP) {
#pragma unroll
                for (int hr = 0; hr < HR; ++hr) frcnn_buf_load_lds_b32(xbuf, xl + q * 4 * CHP + hr * HP, vxr[hr], so[hr]);
            }
            frcnn_buf_load_lds_b32(dbuf, dl + q * 4 * DP, t_vd, (uint32_t)__builtin_amdgcn_readfirstlane((int)(sd + (uint32_t)(q * 4) * chan_bytes)));
        }
    };
    auto issue = [&](int b, int buf) {
        if (full_tile) { issue_lean(b, buf); return; }
        issue_setup(b);
#pragma unroll 1
        for (int q = 0; q < 16; ++q) issue_channel(q, buf);
    };
    // (r03: issuing the NEXT tile's DMAs from inside this tile's MFMA stream -- two channels per pair of steps -- measured 476 us on
    // conv3_2 against 488 us with the DMAs ahead of the MFMAs and 373 us for this single-buffer form: a wave that issues
    // v_mfma_f32_32x32x2_f32 back to back leaves the other waves of its SIMD no issue slot (scripts/micro/mfma_dma_micro.hip), and its own
    // non-MFMA instructions are not free either; what pays is fewer instructions per piece, see issue_lean.)
    // (r03: the bias gradient as a side sum of this loop -- `bsum += bv` once per step -- made the step 0.33 ms SLOWER than the two
    // launches of frcnn_bias_grad_f32 it replaced: a VALU instruction between two fp32 MFMAs is not free, and every wave pays it.)
    auto compute = [&](int buf) {
        if constexpr ((ABL & 4) != 0) return;
        const float *xa = x_lds[buf] + (wci * 32 + l31) * CHP + khalf;
        const float *db = dy_lds[buf] + (wco * 32 + l31) * DP + khalf;
        // step s = (row r, pixel pair pp): one dy value and the T shifted x values feed T MFMAs; the fragments of step s+1 are
        // read before the MFMAs of step s are issued (register double buffer, two steps per trip so it is indexed statically)
        constexpr int NSTEP = WG_ROWS * 16;
        float av[2][T], bv[2];
        auto frag = [&](int s, float (&a)[T], float &bb) {
            const int r = s >> 4, pp = s & 15;
            bb = db[r * 32 + 2 * pp];
#pragma unroll
            for (int t = 0; t < T; ++t) a[t] = xa[(r + t / KS) * HP + 2 * pp + t % KS];
        };
        frag(0, av[0], bv[0]);
        // fully unrolled: every fragment address is the tile base plus a constant in the ds_read offset field (as a loop each pair of
        // steps paid four v_add_u32 and a dozen scalar instructions for its addresses)
#pragma unroll
        for (int s = 0; s < NSTEP; s += 2) {
            frag(s + 1, av[1], bv[1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][t], bv[0], acc[t], 0, 0, 0);
            if (s + 2 < NSTEP) frag(s + 2, av[0], bv[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][t], bv[1], acc[t], 0, 0, 0);
        }
    };
    if constexpr (DB) {
        if (b_begin < b_end) issue(b_begin, 0);
        int buf = 0;
        for (int b = b_begin; b < b_end; ++b) {
            frcnn_wait_vmcnt<0>();                                // tile b's pieces of THIS wave have landed ...
            frcnn_barrier_nofence();                              // ... everybody's have, and everybody is done with tile b-1's image
            if (b + 1 < b_end) issue(b + 1, buf ^ 1);
            compute(buf);
            buf ^= 1;
        }
    } else {
        for (int b = b_begin; b < b_end; ++b) {
            if (b != b_begin) frcnn_barrier_nofence();            // every wave is done reading the previous tile
            issue(b, 0);
            frcnn_wait_vmcnt<0>();
            frcnn_barrier_nofence();
            compute(0);
        }
    }
    if constexpr ((ABL & 8) != 0) { if (acc[0][0] != 12345.678f) return; }
    float *slab = slabs + (size_t)split * ((size_t)Cin * T * Cout);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            const int co = co0 + wco * 32 + l31;
            if (ci < Cin && co < Cout) slab[((size_t)ci * T + t) * Cout + co] = acc[t][r];
        }
}

// First-layer form of the weight gradient (conv1_1: Cin * 9 <= 32).  The generic kernel above pads the 3 input channels to a 64-channel
// tile and spends 21 of every 22 MFMA rows on zeros: 425 us for 2 GFLOP at 600 x 1000, more than any 44-GFLOP layer.  Here the MFMA's 32
// A rows ARE the (channel, tap) pairs -- row m = ci * 9 + tap, exactly dWp's row index -- so one MFMA per 32 output channels and pixel
// pair does all nine taps of all three channels: lane l supplies x[ci][y + ky - 1][x + kx - 1] from ITS OWN (ci, ky, kx) offset into the
// halo image (one ds_read_b32), B is the dy fragment of the generic kernel.  A workgroup (4 waves) walks tiles of 2 rows x 32 px; each
// wave takes a quarter of a tile's 32 pixel pairs with two accumulators (64 output channels); the four waves' sums meet in LDS in wave
// order (deterministic), one slab per workgroup, wgrad_reduce_kernel as before.  The job is then bound by reading dy once (154 MB).
__global__ void __launch_bounds__(256, 4)
conv1_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ slabs, int Cin, int Cout, int H, int W,
                   int xtiles, int nblocks, int splits) {
    constexpr int KS = 3, T = 9, PAD = 1;
    constexpr int HP = 32 + KS - 1, HR = WG_ROWS + KS - 1;
    constexpr int CHP = HR * HP + 1, DP = WG_ROWS * 32 + 1;
    __shared__ float x_lds[3 * CHP];
    __shared__ float dy_lds[64 * DP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int co0 = blockIdx.y * 64, split = blockIdx.x;
    const int b_begin = (int)((long long)split * nblocks / splits), b_end = (int)((long long)(split + 1) * nblocks / splits);
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)Cin * H * W * sizeof(float)));
    const frcnn_buf_t dbuf = frcnn_make_buf(dy, (uint32_t)((size_t)Cout * H * W * sizeof(float)));
    const int rows_m = Cin * T;                             // valid A rows (27)
    const int m_ci = l31 < rows_m ? l31 / T : 0, m_tap = l31 < rows_m ? l31 % T : 0;
    const int a_off = m_ci * CHP + (m_tap / KS) * HP + m_tap % KS + khalf;     // rows >= rows_m read channel 0: their D rows are never stored

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    for (int b = b_begin; b < b_end; ++b) {
        const int tx = b % xtiles, ty = b / xtiles;
        const int x0 = tx * 32, y0 = ty * WG_ROWS;
        if (b != b_begin) frcnn_barrier_nofence();                // every wave is done reading the previous tile
        // x pieces: (channel, halo row) pairs, Cin * HR of them, wave w takes w, w + 4, ...; dy pieces: channel w, w + 4, ...
        const int xs = x0 - PAD > 0 ? x0 - PAD : 0;
        const int gx = x0 - PAD + lane;
        const uint32_t vx = (lane < HP && gx >= 0 && gx < W) ? (uint32_t)(gx - xs) * 4u : kBufOob;
        const int dgy = y0 + (lane >> 5), dgx = x0 + (lane & 31);
        const uint32_t vd = (dgy < H && dgx < W) ? (uint32_t)((lane >> 5) * W + (lane & 31)) * 4u : kBufOob;
        for (int pc = wave; pc < Cin * HR; pc += 4) {
            const int c = pc / HR, hr = pc - c * HR;
            const int gy = y0 - PAD + hr;
            const bool row_ok = gy >= 0 && gy < H;
            const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((int)(row_ok ? (uint32_t)(((size_t)c * H + gy) * W + xs) * 4u : 0u));
            if (lane < HP) frcnn_buf_load_lds_b32(xbuf, &x_lds[c * CHP + hr * HP], row_ok ? vx : kBufOob, so);
        }
#pragma unroll 1
        for (int q = 0; q < 16; ++q) {
            const int c = wave + 4 * q;
            const bool ch_ok = co0 + c < Cout && y0 < H;
            const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ch_ok ? (uint32_t)(((size_t)(co0 + c) * H + y0) * W + x0) * 4u : 0u));
            frcnn_buf_load_lds_b32(dbuf, &dy_lds[c * DP], ch_ok ? vd : kBufOob, so);
        }
        frcnn_wait_vmcnt<0>();
        frcnn_barrier_nofence();
        // this wave's eight pixel pairs: steps 8 * wave .. + 7 of the tile's 32 (row r = s >> 4, pair pp = s & 15)
        const float *xa = x_lds + a_off;
        const float *d0 = dy_lds + l31 * DP + khalf, *d1 = dy_lds + (32 + l31) * DP + khalf;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int st = wave * 8 + i, r = st >> 4, pp = st & 15;
            const float av = xa[r * HP + 2 * pp], b0 = d0[r * 32 + 2 * pp], b1 = d1[r * 32 + 2 * pp];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
        }
    }
    // the four waves' partial sums, added in wave order in LDS ([32 rows][64 couts] over the dy image), then one slab per workgroup
    float *red = dy_lds;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    float *cell = red + m * 64 + j * 32 + l31;
                    *cell = (w == 0 ? 0.0f : *cell) + acc[j][r];
                }
        }
    }
    __syncthreads();
    float *slab = slabs + (size_t)split * ((size_t)rows_m * Cout);
    for (int e = tid; e < rows_m * 64; e += 256) {
        const int m = e >> 6, c = e & 63;
        if (co0 + c < Cout) slab[(size_t)m * Cout + co0 + c] = red[m * 64 + c];
    }
}

// The same weight gradient on the bf16 matrix cores with fp32-class results (conv_f32s.hip's scheme: every fp32 value carried as
// three bf16 terms, six v_mfma_f32_32x32x16_bf16 products per block, fp32 accumulation).  The PIXELS are the reduction axis, so the
// MFMA's eight consecutive k-values of a lane are eight consecutive pixels of one channel's row: the workgroup reads the fp32 NCHW
// tensors x and dy as they are (two pixels per load), splits each pair in registers and keeps pixel-contiguous bf16 rows in LDS --
// x [part][ci][4 halo rows][96 B], dy [part][co][2 rows][64 B], channel pitches 400 / 144 B = odd multiples of 16 B, so the 16 lanes
// of a ds_read_b128 group (16 channels, one pixel offset) fall on 16 distinct bank slots.  A tap's horizontal shift of +-1 pixel is
// 2 bytes -- no aligned 16-byte read exists for it -- so a lane reads the aligned eight pixels plus the dword on either side and
// builds the two shifted fragments with five v_alignbit_b32.  Per 16-pixel step a wave reads 30 fragments pieces for 54 MFMAs; each
// value is split once per workgroup and used by 32 x 9 x 6 MFMA rows, so the split costs nothing next to the matrix work; and the
// operands a step needs are 17 KB per 108 MFMAs (the forward kernel: 74 KB) -- this kernel is bound by the matrix pipe, not by
// staging.  One workgroup per CU (104 KB of LDS, 144 accumulator registers); the next tile's global loads are in flight during the
// current tile's MFMAs.  Slabs, split order and the reduce pass are the fp32 kernel's.
constexpr int kSRows = 3;                      // tile rows of the split kernel (x halo: 5 rows; 135 KB of LDS)
__global__ void __launch_bounds__(256, 1)
conv_wgrad_f32s_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ slabs, int Cin, int Cout, int H, int W,
                       int xtiles, int nblocks, int splits) {
    constexpr int T = 9, HR = kSRows + 2;
    constexpr int XROW = 96, XCH = HR * XROW + 16;             // bytes: a halo row (16 B lead-in, 34 px, pad), a channel (odd multiple of 16)
    constexpr int DROW = 64, DCH = kSRows * DROW + 16;         // a dy row (32 px), a channel
    static_assert((XCH / 16) % 2 == 1 && (DCH / 16) % 2 == 1, "channel pitches must be odd multiples of 16 bytes");
    constexpr int XPART = 64 * XCH, DPART = 64 * DCH;
    __shared__ __attribute__((aligned(16))) unsigned char x_lds[3 * XPART];
    __shared__ __attribute__((aligned(16))) unsigned char dy_lds[3 * DPART];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wci = wave & 1, wco = wave >> 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int HWs = H * W;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, split = blockIdx.z;
    const int b_begin = (int)((long long)split * nblocks / splits), b_end = (int)((long long)(split + 1) * nblocks / splits);
    const frcnn_buf_t xbuf = frcnn_make_buf(x, (uint32_t)((size_t)Cin * HWs * sizeof(float)));
    const frcnn_buf_t dbuf = frcnn_make_buf(dy, (uint32_t)((size_t)Cout * HWs * sizeof(float)));

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // staging: pixel QUADS (16 bytes of fp32 in, 8 bytes of bf16 per part out).  x: 64 channels x HR halo rows x 9 quads (px x0-2 ..
    // x0+33); dy: 64 x kSRows x 8 quads.  The quad -> (channel, row, quad) map is fixed per thread -- LDS offset, channel base and the
    // (row, column) offsets relative to the tile origin are computed ONCE; only the origin moves.
    constexpr int XN = 64 * HR * 9, DN = 64 * kSRows * 8;
    constexpr int XQ = (XN + 255) / 256, DQ = (DN + 255) / 256;
    int x_lo[XQ], x_dx[XQ], x_dr[XQ], d_lo[DQ], d_dx[DQ], d_dr[DQ];
    uint32_t x_cb[XQ], d_cb[DQ];
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        const int e = tid + 256 * q, c = e / (HR * 9), rem = e - c * (HR * 9), hr = rem / 9, pq = rem - hr * 9;
        const bool ok = e < XN && ci0 + c < Cin;
        x_lo[q] = c * XCH + hr * XROW + 12 + 8 * pq;                  // px x0 sits at byte 16 of a row
        x_dx[q] = 4 * pq - 2;
        x_dr[q] = ok ? hr - 1 : -(1 << 20);                           // an invalid quad's row is never inside the image
        x_cb[q] = (uint32_t)((size_t)(ci0 + c) * HWs) * 4u;
    }
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
        const int e = tid + 256 * q, c = e / (kSRows * 8), rem = e - c * (kSRows * 8), r = rem >> 3, pq = rem & 7;
        const bool ok = e < DN && co0 + c < Cout;
        d_lo[q] = c * DCH + r * DROW + 8 * pq;
        d_dx[q] = 4 * pq;
        d_dr[q] = ok ? r : -(1 << 20);
        d_cb[q] = (uint32_t)((size_t)(co0 + c) * HWs) * 4u;
    }
    float xv[XQ][4], dv[DQ][4];
    auto fetch = [&](int b) {
        const int tx = b % xtiles, ty = b / xtiles;
        const int x0 = tx * 32, y0 = ty * kSRows;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int gy = y0 + x_dr[q], gx = x0 + x_dx[q];
            const bool row_ok = gy >= 0 && gy < H;
            const uint32_t base = x_cb[q] + (uint32_t)(gy * W + gx) * 4u;
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[q][j] = frcnn_buf_load_f32(xbuf, (row_ok && gx + j >= 0 && gx + j < W) ? base + 4u * j : kBufOob);
        }
#pragma unroll
        for (int q = 0; q < DQ; ++q) {
            const int gy = y0 + d_dr[q], gx = x0 + d_dx[q];
            const bool row_ok = gy >= 0 && gy < H;
            const uint32_t base = d_cb[q] + (uint32_t)(gy * W + gx) * 4u;
#pragma unroll
            for (int j = 0; j < 4; ++j) dv[q][j] = frcnn_buf_load_f32(dbuf, (row_ok && gx + j < W) ? base + 4u * j : kBufOob);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            if (XN % 256 != 0 && q == XQ - 1 && tid + 256 * q >= XN) continue;
            uint32_t h0, m0, l0, h1, m1, l1;
            frcnn_split3_pair(xv[q][0], xv[q][1], h0, m0, l0);
            frcnn_split3_pair(xv[q][2], xv[q][3], h1, m1, l1);
            unsigned char *d = x_lds + x_lo[q];                       // (4-byte aligned: two dword stores per part)
            *reinterpret_cast<uint32_t *>(d) = h0; *reinterpret_cast<uint32_t *>(d + 4) = h1;
            *reinterpret_cast<uint32_t *>(d + XPART) = m0; *reinterpret_cast<uint32_t *>(d + XPART + 4) = m1;
            *reinterpret_cast<uint32_t *>(d + 2 * XPART) = l0; *reinterpret_cast<uint32_t *>(d + 2 * XPART + 4) = l1;
        }
#pragma unroll
        for (int q = 0; q < DQ; ++q) {
            if (DN % 256 != 0 && q == DQ - 1 && tid + 256 * q >= DN) continue;
            uint32_t h0, m0, l0, h1, m1, l1;
            frcnn_split3_pair(dv[q][0], dv[q][1], h0, m0, l0);
            frcnn_split3_pair(dv[q][2], dv[q][3], h1, m1, l1);
            unsigned char *d = dy_lds + d_lo[q];                      // 8-byte aligned
            *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(d + DPART) = make_uint2(m0, m1);
            *reinterpret_cast<uint2 *>(d + 2 * DPART) = make_uint2(l0, l1);
        }
    };
    auto compute = [&]() {
        const unsigned char *xa = x_lds + (wci * 32 + l31) * XCH + 16 + 16 * khalf;     // this lane's eight pixels of a 16-pixel step
        const unsigned char *db = dy_lds + (wco * 32 + l31) * DCH + 16 * khalf;
        // unit u = (row r, 16-pixel step ks, tap row ky): 18 MFMAs.  One wave per SIMD has nobody to hide an LDS round trip behind, so
        // the raw reads of unit u+1 are issued BEFORE the MFMAs of unit u (register double buffer, two units per trip so that it is
        // indexed statically); the compiler's own schedule read each fragment right before its use: 9 exposed round trips per step
        constexpr int NU = kSRows * 2 * 3;
        struct Raw { uint4 mid[3]; uint32_t lo[3], hi[3]; uint4 b[3]; };
        auto load = [&](int u, Raw &w) {
            const int sidx = u / 3, ky = u - sidx * 3, r = sidx >> 1, ks = sidx & 1;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned char *row = xa + p * XPART + (r + ky) * XROW + ks * 32;
                w.mid[p] = *reinterpret_cast<const uint4 *>(row);
                w.lo[p] = *reinterpret_cast<const uint32_t *>(row - 4);
                w.hi[p] = *reinterpret_cast<const uint32_t *>(row + 16);
                w.b[p] = *reinterpret_cast<const uint4 *>(db + p * DPART + r * DROW + ks * 32);
            }
        };
        auto mfmas = [&](int ky, const Raw &w) {
            uint4 a[3][3];                                                               // [part][kx]
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint4 mid = w.mid[p];
                const uint32_t s01 = frcnn_alignbit(mid.y, mid.x, 16), s12 = frcnn_alignbit(mid.z, mid.y, 16), s23 = frcnn_alignbit(mid.w, mid.z, 16);
                a[p][0] = make_uint4(frcnn_alignbit(mid.x, w.lo[p], 16), s01, s12, s23);   // pixels -1 .. +6: tap column 0
                a[p][1] = mid;
                a[p][2] = make_uint4(s01, s12, s23, frcnn_alignbit(w.hi[p], mid.w, 16));   // pixels +1 .. +8: tap column 2
            }
            // six products per tap; the three taps of the row take turns so that consecutive MFMAs never share an accumulator
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = frcnn_mfma_32x32x16_bf16(a[2][kx], w.b[0], acc[ky * 3 + kx]);     // l.h
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = frcnn_mfma_32x32x16_bf16(a[0][kx], w.b[2], acc[ky * 3 + kx]);     // h.l
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = frcnn_mfma_32x32x16_bf16(a[1][kx], w.b[1], acc[ky * 3 + kx]);     // m.m
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = frcnn_mfma_32x32x16_bf16(a[1][kx], w.b[0], acc[ky * 3 + kx]);     // m.h
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = frcnn_mfma_32x32x16_bf16(a[0][kx], w.b[1], acc[ky * 3 + kx]);     // h.m
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = frcnn_mfma_32x32x16_bf16(a[0][kx], w.b[0], acc[ky * 3 + kx]);     // h.h
        };
        static_assert(NU % 6 == 0, "six units per trip: the tap row index is static");
        Raw w0, w1;
        load(0, w0);
#pragma unroll 1
        for (int u = 0; u < NU; u += 6) {
#pragma unroll
            for (int v = 0; v < 6; v += 2) {
                load(u + v + 1, w1);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(v % 3, w0);
                if (u + v + 2 < NU) load(u + v + 2, w0);
                __builtin_amdgcn_sched_barrier(0);
                mfmas((v + 1) % 3, w1);
            }
        }
    };
    if (b_begin < b_end) fetch(b_begin);
    for (int b = b_begin; b < b_end; ++b) {
        if (b != b_begin) __syncthreads();                         // every wave is done reading the previous tile
        stage();
        __syncthreads();
        if (b + 1 < b_end) fetch(b + 1);                           // in flight during this tile's MFMAs
        compute();
    }
    float *slab = slabs + (size_t)split * ((size_t)Cin * T * Cout);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            const int co = co0 + wco * 32 + l31;
            if (ci < Cin && co < Cout) slab[((size_t)ci * T + t) * Cout + co] = acc[t][r];
        }
}

__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float *__restrict__ slabs, size_t n, int splits, float *__restrict__ dwp) {
    // slabs are added in split order (deterministic); four slabs' loads are in flight at a time and each thread owns four
    // consecutive weights (n is a multiple of 4: Cout % 4 == 0), so the pass streams at the memory rate instead of one 4-byte
    // load per thread per round trip
    const size_t n4 = n / 4;
    const bool out_aligned = (reinterpret_cast<uintptr_t>(dwp) & 15) == 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int q = 0;
        for (; q + 8 <= splits; q += 8) {                                      // (round 5: eight slabs' loads in flight, then four; same order of additions)
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4 *>(slabs + (size_t)(q + u) * n)[i];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; q + 4 <= splits; q += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4 *>(slabs + (size_t)(q + u) * n)[i];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; q < splits; ++q) {
            const float4 v = reinterpret_cast<const float4 *>(slabs + (size_t)q * n)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (out_aligned) reinterpret_cast<float4 *>(dwp)[i] = s;
        else { dwp[4 * i] = s.x; dwp[4 * i + 1] = s.y; dwp[4 * i + 2] = s.z; dwp[4 * i + 3] = s.w; }   // a view at an odd offset of the flat gradient buffer
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int q = 0; q < splits; ++q) s += slabs[(size_t)q * n + i];
        dwp[i] = s;
    }
}

// WeightDecay hook, then MomentumSGD (Chainer v1): g += wd*W; v = momentum*v - lr*g; W += v
__global__ void __launch_bounds__(256)
sgd_momentum_wd_kernel(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ v, size_t n, float lr, float momentum, float wd) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float wi = w[i];
        const float gi = g[i] + wd * wi;
        const float vi = momentum * v[i] - lr * gi;
        v[i] = vi;
        w[i] = wi + vi;
    }
}

__global__ void __launch_bounds__(256)
transpose_kernel(const float *__restrict__ src, int R, int Cc, float *__restrict__ dst) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) tile[i][tx] = (r0 + i < R && c0 + tx < Cc) ? src[(size_t)(r0 + i) * Cc + c0 + tx] : 0.0f;
    __syncthreads();
    for (int i = ty; i < 64; i += 4)
        if (c0 + i < Cc && r0 + tx < R) dst[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

// ------------------------------------------------------------------------------------------------
// Stage-2 (rcnn_train) losses and gradients, models/faster_rcnn.py:152-164: F.softmax_cross_entropy(cls_score, labels) (mean over
// the R sampled RoIs), F.accuracy, F.huber_loss(bbox_pred, targets, delta) summed per RoI and divided by R; gradients of
// loss_cls + loss_bbox.  One workgroup, fixed reduction tree.  out[0..2] = loss_cls, loss_bbox, cls_accuracy.
__global__ void __launch_bounds__(256)
rcnn_loss_kernel(const float *__restrict__ cls_score, const float *__restrict__ bbox_pred, const int32_t *__restrict__ labels,
                 const float *__restrict__ targets, int R, int ncls, float delta, float *__restrict__ out, float *__restrict__ dcls,
                 float *__restrict__ dbbox) {
    __shared__ double red[3][256];
    const int tid = threadIdx.x;
    double s_cls = 0.0, s_box = 0.0, s_acc = 0.0;
    const float inv_r = 1.0f / (float)(R > 0 ? R : 1);
    for (int r = tid; r < R; r += 256) {
        const float *s = cls_score + (size_t)r * ncls;
        const int lab = labels[r];
        float m = s[0];
        int arg = 0;
        for (int c = 1; c < ncls; ++c) if (s[c] > m) { m = s[c]; arg = c; }
        float z = 0.0f;
        for (int c = 0; c < ncls; ++c) z += expf(s[c] - m);
        const float logz = m + logf(z);
        s_cls += (double)(logz - s[lab]);
        s_acc += arg == lab ? 1.0 : 0.0;
        if (dcls)
            for (int c = 0; c < ncls; ++c) dcls[(size_t)r * ncls + c] = (expf(s[c] - m) / z - (c == lab ? 1.0f : 0.0f)) * inv_r;
        for (int c = 0; c < 4 * ncls; ++c) {
            const float d = bbox_pred[(size_t)r * 4 * ncls + c] - targets[(size_t)r * 4 * ncls + c];
            const float ad = fabsf(d);
            s_box += (double)(ad < delta ? 0.5f * d * d : delta * (ad - 0.5f * delta));
            if (dbbox) dbbox[(size_t)r * 4 * ncls + c] = (ad < delta ? d : (d > 0.0f ? delta : -delta)) * inv_r;
        }
    }
    red[0][tid] = s_cls; red[1][tid] = s_box; red[2][tid] = s_acc;
    __syncthreads();
    for (int q = 128; q > 0; q >>= 1) {
        if (tid < q) {
#pragma unroll
            for (int k = 0; k < 3; ++k) red[k][tid] += red[k][tid + q];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double n = R > 0 ? (double)R : 1.0;
        out[0] = (float)(red[0][0] / n);
        out[1] = (float)(red[1][0] / n);
        out[2] = (float)(red[2][0] / n);
    }
}

// y = a * b (F.dropout forward with b = the 0 / 1/(1-ratio) mask, and its backward); y may alias a
__global__ void __launch_bounds__(256)
mul_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t n, float *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = a[i] * b[i];
}

// F.dropout with the mask drawn ON THE DEVICE: element i keeps its value (scaled by 1 / (1 - ratio)) iff u(seed, i) >= ratio, u = the top 24 bits of a
// splitmix64 finaliser over (seed, i) -- a counter-based generator: stateless, order-free, reproducible for a given seed.  Writes the mask as well
// (0 or 1 / (1 - ratio)): the backward pass multiplies by it.  The reference's CPU path draws numpy.random.rand (chainer F.dropout); its GPU path
// draws from cupy's generator, so no bit-level claim exists there -- the NumPy-stream form stays the default of RCNNTrainer for the parity tests.
__global__ void __launch_bounds__(256)
dropout_kernel(const float *__restrict__ x, size_t n, float ratio, unsigned long long seed, float scale, float *__restrict__ mask, float *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = seed + 0x9e3779b97f4a7c15ull * (unsigned long long)(i + 1);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z ^= z >> 31;
        const float u = (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);          // [0, 1), 24 bits: exact in fp32
        const float m = u >= ratio ? scale : 0.0f;
        mask[i] = m;
        y[i] = x[i] * m;
    }
}

__global__ void __launch_bounds__(256)
add_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t n, float *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = a[i] + b[i];
}

// F.relu backward: g = (out > 0) ? g : 0, in place
__global__ void __launch_bounds__(256)
relu_bwd_kernel(float *__restrict__ g, const float *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) g[i] = out[i] > 0.0f ? g[i] : 0.0f;
}

// dst[i][:] = src[idx[i]][:]  (x[keep_inds]) and its adjoint dst[idx[i]][:] = src[i][:] into a zero-filled dst (indices unique)
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx, int n, int cols, float *__restrict__ dst, int scatter) {
    const size_t total = (size_t)n * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
        if (scatter) dst[(size_t)idx[r] * cols + c] = src[i];
        else dst[i] = src[(size_t)idx[r] * cols + c];
    }
}

struct WgradPlan { int xtiles, nblocks, splits, ci_tiles, co_tiles; size_t slab_floats; };

// The double-buffered 3x3 kernel (one workgroup per CU, half as many slabs) against the single-buffer one (two per CU).  Round 2 measured it
// 3.6 % slower on the training step and left it off; with the lean DMA issue and the unrolled MFMA loop of round 3 it is the faster form wherever a
// workgroup of the single-buffer launch would get few tiles or the layer has a single (ci, co) tile (profiles/r03_wgrad_micro.txt: conv1_2
// 433 -> 403 us, conv2_1 230 -> 217, conv3_1 220 -> 213, conv4_1 215 -> 208, the 38 x 63 layers 127 -> 118) and the slower one on the 44-GFLOP layers
// with >= 4 (ci, co) tiles (conv3_2 375 -> 392): picked per layer.  FRCNN_WGRAD_DB=0 / 1 forces one form (A/B hook).
static bool wgrad_double_buffered(int Cin, int Cout, int H, int W) {
    const char *e = frcnn_tune("FRCNN_WGRAD_DB");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    const int cico = frcnn_cdiv(Cin, 64) * frcnn_cdiv(Cout, 64);
    const int nblocks = frcnn_cdiv(W, 32) * frcnn_cdiv(H, WG_ROWS);
    int splits2 = frcnn_cdiv(2 * frcnn_cu_count(), cico);             // what the single-buffer launch would use
    if (splits2 > nblocks) splits2 = nblocks;
    return cico == 1 || nblocks < 14 * splits2;
}

// conv1_1 (Cin * 9 <= 32) has its own kernel; FRCNN_WGRAD_CONV1=generic keeps the generic one on it (A/B, tests compare the two)
static bool wgrad_first_layer_form(int Cin, int ks) {
    if (ks != 3 || Cin * 9 > 32) return false;
    const char *e = frcnn_tune("FRCNN_WGRAD_CONV1");
    return !(e && e[0] == 'g');
}

static WgradPlan plan_wgrad(int Cin, int Cout, int H, int W, int ks) {
    WgradPlan p;
    p.xtiles = frcnn_cdiv(W, 32);
    p.nblocks = p.xtiles * frcnn_cdiv(H, WG_ROWS);
    p.ci_tiles = frcnn_cdiv(Cin, 64);
    p.co_tiles = frcnn_cdiv(Cout, 64);
    if (wgrad_first_layer_form(Cin, ks)) {                       // four workgroups per CU, one slab each
        int s1 = frcnn_cdiv(4 * frcnn_cu_count(), p.co_tiles);
        if (s1 > p.nblocks) s1 = p.nblocks;
        p.splits = s1 < 1 ? 1 : s1;
        p.slab_floats = (size_t)Cin * ks * ks * Cout;
        return p;
    }
    // 3x3 double-buffered kernel: one workgroup per CU; the single-buffer forms (1x1, FRCNN_WGRAD_DB=0): about two per CU
    int target = (ks == 3 && wgrad_double_buffered(Cin, Cout, H, W)) ? frcnn_cu_count() : 2 * frcnn_cu_count();
    const int mul10 = frcnn_tune_int("FRCNN_WGRAD_MUL10", 0);            // A/B hook (round 6): workgroups of the launch = CUs x mul10 / 10 instead of the 1 x / 2 x rule
    if (mul10 > 0) target = frcnn_cu_count() * mul10 / 10;
    int s = frcnn_cdiv(target, p.ci_tiles * p.co_tiles);
    if (s > p.nblocks) s = p.nblocks;
    if (s < 1) s = 1;
    p.splits = s;
    p.slab_floats = (size_t)Cin * ks * ks * Cout;
    return p;
}

static AnchorsD load_anchors(const double *anchors_host, int A) {
    AnchorsD anc;
    for (int a = 0; a < 32; ++a)
        for (int c = 0; c < 4; ++c) anc.a[a][c] = a < A ? anchors_host[a * 4 + c] : 0.0;
    return anc;
}

struct AtlLayout { size_t overlaps, max_ov, gt_max, total; };
static AtlLayout atl_layout(int n_all, int G) {
    AtlLayout L;
    size_t o = 0;
    L.overlaps = o; o += frcnn_align256((size_t)n_all * G * sizeof(double));
    L.max_ov = o; o += frcnn_align256((size_t)n_all * sizeof(double));
    L.gt_max = o; o += frcnn_align256((size_t)G * sizeof(double));
    L.total = o;
    return L;
}

}  // namespace

extern "C" {

int frcnn_bbox_overlaps_f64(const double *boxes, int N, const double *query_boxes, int K, double *overlaps, void *stream) {
    if (N < 0 || K < 0 || (N > 0 && K > 0 && (!boxes || !query_boxes || !overlaps))) return FRCNN_ERR_INVALID;
    if (N == 0 || K == 0) return FRCNN_OK;
    const size_t total = (size_t)N * K;
    hipLaunchKernelGGL(bbox_overlaps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes, N, query_boxes, K,
                       overlaps);
    return frcnn_launch_status();
}

size_t frcnn_anchor_target_workspace_bytes(int A, int H, int W, int G) {
    if (A < 1 || H < 1 || W < 1 || G < 1) return 0;
    return atl_layout(A * H * W, G).total;
}

int frcnn_anchor_target(const double *anchors_host, int A, int H, int W, int feat_stride, int im_h, int im_w, const float *gt_boxes, int G,
                        int32_t *inds_inside, int32_t *n_inside, int32_t *labels, float *targets, int32_t *argmax_overlaps,
                        void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!anchors_host || !gt_boxes || !inds_inside || !n_inside || !labels || !targets || !argmax_overlaps) return FRCNN_ERR_INVALID;
    if (A < 1 || A > 32 || H < 1 || W < 1 || G < 1) return FRCNN_ERR_INVALID;
    const int n_all = A * H * W;
    const AtlLayout L = atl_layout(n_all, G);
    if (!workspace || workspace_bytes < L.total) return FRCNN_ERR_INVALID;
    char *ws = (char *)workspace;
    double *overlaps = (double *)(ws + L.overlaps), *max_ov = (double *)(ws + L.max_ov), *gt_max = (double *)(ws + L.gt_max);
    const AnchorsD anc = load_anchors(anchors_host, A);
    hipLaunchKernelGGL(atl_inside_kernel, dim3(1), dim3(1024), 0, stream, anc, A, H, W, feat_stride, im_h, im_w, inds_inside, n_inside);
    const dim3 grid(frcnn_cdiv(n_all, 256)), blk(256);
    hipLaunchKernelGGL(atl_overlap_kernel, grid, blk, 0, stream, anc, A, W, feat_stride, inds_inside, n_inside, gt_boxes, G, overlaps, max_ov,
                       argmax_overlaps);
    hipLaunchKernelGGL(atl_gtmax_kernel, dim3(G), blk, 0, stream, overlaps, n_inside, G, gt_max);
    hipLaunchKernelGGL(atl_label_kernel, grid, blk, 0, stream, anc, A, W, feat_stride, inds_inside, n_inside, gt_boxes, G, overlaps, max_ov,
                       argmax_overlaps, gt_max, 0.3, 0.7, labels, targets);
    return frcnn_launch_status();
}

int frcnn_rpn_loss(const float *rpn_cls_score, const float *rpn_bbox_pred, const int32_t *labels, const float *targets,
                   const int32_t *inds_inside, int n_inside, int A, int H, int W, float delta, float loss_lambda, float *losses,
                   float *d_cls_score, float *d_bbox_pred, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!rpn_cls_score || !rpn_bbox_pred || !losses || A < 1 || H < 1 || W < 1 || n_inside < 0) return FRCNN_ERR_INVALID;
    if (n_inside > 0 && (!labels || !targets || !inds_inside)) return FRCNN_ERR_INVALID;
    if ((d_cls_score == nullptr) != (d_bbox_pred == nullptr)) return FRCNN_ERR_INVALID;
    const size_t HW = (size_t)H * W;
    if (d_cls_score) {
        FRCNN_HIP_TRY(hipMemsetAsync(d_cls_score, 0, sizeof(float) * 2 * A * HW, stream));
        FRCNN_HIP_TRY(hipMemsetAsync(d_bbox_pred, 0, sizeof(float) * 4 * A * HW, stream));
    }
    hipLaunchKernelGGL(rpn_loss_kernel, dim3(1), dim3(1024), 0, stream, rpn_cls_score, rpn_bbox_pred, labels, targets, inds_inside, n_inside, A,
                       (int)HW, delta, loss_lambda, losses, d_cls_score, d_bbox_pred);
    return frcnn_launch_status();
}

int frcnn_rcnn_loss(const float *cls_score, const float *bbox_pred, const int32_t *labels, const float *targets, int R, int ncls, float delta,
                    float *losses, float *d_cls_score, float *d_bbox_pred, void *stream) {
    if (!cls_score || !bbox_pred || !labels || !targets || !losses || R < 1 || ncls < 2) return FRCNN_ERR_INVALID;
    if ((d_cls_score == nullptr) != (d_bbox_pred == nullptr)) return FRCNN_ERR_INVALID;
    hipLaunchKernelGGL(rcnn_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, cls_score, bbox_pred, labels, targets, R, ncls, delta, losses,
                       d_cls_score, d_bbox_pred);
    return frcnn_launch_status();
}

static int grid_for(size_t n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

int frcnn_mul_f32(const float *a, const float *b, size_t n, float *y, void *stream) {
    if (n == 0) return FRCNN_OK;
    if (!a || !b || !y) return FRCNN_ERR_INVALID;
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, y);
    return frcnn_launch_status();
}

int frcnn_dropout_f32(const float *x, size_t n, float ratio, unsigned long long seed, float *mask, float *y, void *stream) {
    if (n == 0) return FRCNN_OK;
    if (!x || !mask || !y || !(ratio >= 0.0f) || !(ratio < 1.0f)) return FRCNN_ERR_INVALID;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, ratio, seed, 1.0f / (1.0f - ratio), mask, y);
    return frcnn_launch_status();
}

int frcnn_add_f32(const float *a, const float *b, size_t n, float *y, void *stream) {
    if (n == 0) return FRCNN_OK;
    if (!a || !b || !y) return FRCNN_ERR_INVALID;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, y);
    return frcnn_launch_status();
}

int frcnn_relu_bwd_f32(float *g, const float *out, size_t n, void *stream) {
    if (n == 0) return FRCNN_OK;
    if (!g || !out) return FRCNN_ERR_INVALID;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, out, n);
    return frcnn_launch_status();
}

int frcnn_gather_rows_f32(const float *src, const int32_t *idx, int n, int cols, float *dst, void *stream) {
    if (n == 0) return FRCNN_OK;
    if (!src || !idx || !dst || n < 0 || cols < 1) return FRCNN_ERR_INVALID;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)n * cols)), dim3(256), 0, (hipStream_t)stream, src, idx, n, cols, dst, 0);
    return frcnn_launch_status();
}

int frcnn_scatter_rows_f32(const float *src, const int32_t *idx, int n, int cols, float *dst, int dst_rows, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dst || dst_rows < 1 || cols < 1 || n < 0 || (n > 0 && (!src || !idx))) return FRCNN_ERR_INVALID;
    FRCNN_HIP_TRY(hipMemsetAsync(dst, 0, sizeof(float) * (size_t)dst_rows * cols, stream));
    if (n == 0) return FRCNN_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)n * cols)), dim3(256), 0, stream, src, idx, n, cols, dst, 1);
    return frcnn_launch_status();
}

int frcnn_maxpool2x2_bwd_f32(const float *x, const float *dy, float *dx, int C, int H, int W, void *stream) {
    if (!x || !dy || !dx || C < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const size_t total = (size_t)C * OH * OW;                     // one thread per window
    if (total >= (1ull << 32)) return FRCNN_ERR_INVALID;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, dy, dx, C, H, W, OH, OW);
    return frcnn_launch_status();
}

int frcnn_maxpool2x2_bwd_idx_f32(const unsigned char *idx, const float *dy, float *dx, int C, int H, int W, void *stream) {
    if (!idx || !dy || !dx || C < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const size_t total = (size_t)C * OH * OW;
    if (total >= (1ull << 32)) return FRCNN_ERR_INVALID;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool2x2_bwd_idx_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, idx, dy, dx, C, H, W, OH, OW);
    return frcnn_launch_status();
}

static int bias_grad_parts(int C, int HW) {
    int parts = frcnn_cdiv(1024, C);                      // about four workgroups per CU in total
    const int max_parts = frcnn_cdiv(HW, 2048);
    if (parts > max_parts) parts = max_parts;
    return parts < 1 ? 1 : parts;
}

size_t frcnn_bias_grad_workspace_bytes(int C, int HW) {
    if (C < 1 || HW < 1) return 0;
    return frcnn_align256((size_t)C * bias_grad_parts(C, HW) * sizeof(float));
}

int frcnn_bias_grad_f32(const float *dy, int C, int HW, float *db, void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dy || !db || C < 1 || HW < 1) return FRCNN_ERR_INVALID;
    const int parts = bias_grad_parts(C, HW);
    if (!workspace || workspace_bytes < (size_t)C * parts * sizeof(float)) return FRCNN_ERR_INVALID;
    hipLaunchKernelGGL(bias_grad_partial_kernel, dim3(C, parts), dim3(256), 0, stream, dy, HW, parts, (float *)workspace);
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3(frcnn_cdiv(C, 256)), dim3(256), 0, stream, (const float *)workspace, C, parts, db);
    return frcnn_launch_status();
}

int frcnn_pack_conv_dgrad_w(const float *w_packed, int Cin, int Cout, int ksize, float *w_dgrad, void *stream) {
    if (!w_packed || !w_dgrad || Cin < 1 || Cout < 1 || (ksize != 1 && ksize != 3)) return FRCNN_ERR_INVALID;
    const size_t total = (size_t)Cin * Cout * ksize * ksize;
    hipLaunchKernelGGL(pack_dgrad_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_packed, Cin, Cout,
                       ksize * ksize, w_dgrad);
    return frcnn_launch_status();
}

int frcnn_pack_conv_dgrad_w_many(const frcnn_dgrad_pack_desc *layers, int n, void *stream) {
    if (!layers || n < 1 || n > 16) return FRCNN_ERR_INVALID;
    DgradPackArgs a;
    memset(&a, 0, sizeof(a));
    unsigned blocks = 0;
    for (int i = 0; i < n; ++i) {
        const frcnn_dgrad_pack_desc &d = layers[i];
        if (!d.w_packed || !d.w_dgrad || d.Cin < 1 || d.Cout < 1 || (d.ksize != 1 && d.ksize != 3)) return FRCNN_ERR_INVALID;
        a.wp[i] = d.w_packed; a.wd[i] = d.w_dgrad; a.cin[i] = d.Cin; a.cout[i] = d.Cout; a.taps[i] = d.ksize * d.ksize;
        blocks += (unsigned)(d.ksize * d.ksize * ((d.Cin + 63) / 64) * ((d.Cout + 63) / 64));
        a.block_end[i] = blocks;
    }
    a.n = n;
    hipLaunchKernelGGL(pack_dgrad_w_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return frcnn_launch_status();
}

size_t frcnn_conv_wgrad_workspace_bytes(int Cin, int Cout, int H, int W, int ksize) {
    if (Cin < 1 || Cout < 1 || H < 1 || W < 1 || (ksize != 1 && ksize != 3)) return 0;
    const WgradPlan p = plan_wgrad(Cin, Cout, H, W, ksize);
    return frcnn_align256(p.slab_floats * p.splits * sizeof(float));
}

int frcnn_conv_wgrad_f32(const float *x, const float *dy, float *dw_packed, int Cin, int Cout, int H, int W, int ksize, void *workspace,
                         size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !dy || !dw_packed || Cin < 1 || Cout < 1 || H < 1 || W < 1 || (ksize != 1 && ksize != 3)) return FRCNN_ERR_INVALID;
    if ((size_t)Cin * H * W * 4 >= (1ull << 31) || (size_t)Cout * H * W * 4 >= (1ull << 31)) return FRCNN_ERR_INVALID;
    const WgradPlan p = plan_wgrad(Cin, Cout, H, W, ksize);
    if (!workspace || workspace_bytes < p.slab_floats * p.splits * sizeof(float)) return FRCNN_ERR_INVALID;
    float *slabs = (float *)workspace;
    if (wgrad_first_layer_form(Cin, ksize)) {                  // conv1_1: the (channel, tap) pairs are the MFMA rows
        hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(p.splits, p.co_tiles), dim3(256), 0, stream, x, dy, slabs, Cin, Cout, H, W, p.xtiles, p.nblocks, p.splits);
        const size_t n1 = p.slab_floats;
        const size_t work1 = (n1 / 4 + 255) / 256 + 1;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)(work1 < 4096 ? work1 : 4096)), dim3(256), 0, stream, slabs, n1, p.splits, dw_packed);
        return frcnn_launch_status();
    }
    const dim3 grid(p.ci_tiles, p.co_tiles, p.splits);
    const bool reg = frcnn_tune("FRCNN_WGRAD_REG") != nullptr;        // A/B hook: the register-staged kernel
    if (ksize == 3 && !reg && wgrad_double_buffered(Cin, Cout, H, W)) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_dma_kernel<3, true>), grid, dim3(256), 0, stream, x, dy, slabs, Cin, Cout, H, W, p.xtiles, p.nblocks, p.splits);
    else if (ksize == 3 && !reg) {
#define FRCNN_WGRAD_LAUNCH(ABL_) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_dma_kernel<3, false, ABL_>), grid, dim3(256), 0, stream, x, dy, slabs, Cin, Cout, H, W, p.xtiles, p.nblocks, p.splits)
#ifdef FRCNN_TIMING_ABLATIONS
        const char *ae = frcnn_tune("FRCNN_WGRAD_ABL");
        const int abl = ae ? atoi(ae) : 0;
        if (abl == 1) FRCNN_WGRAD_LAUNCH(1);
        else if (abl == 4) FRCNN_WGRAD_LAUNCH(4);
        else if (abl == 8) FRCNN_WGRAD_LAUNCH(8);
        else if (abl == 5) FRCNN_WGRAD_LAUNCH(5);
        else
#endif
        FRCNN_WGRAD_LAUNCH(0);
#undef FRCNN_WGRAD_LAUNCH
    }
    else if (ksize == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_mfma_kernel<3>), grid, dim3(256), 0, stream, x, dy, slabs, Cin, Cout, H, W, p.xtiles, p.nblocks, p.splits);
    else if (!reg) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_dma_kernel<1>), grid, dim3(256), 0, stream, x, dy, slabs, Cin, Cout, H, W, p.xtiles, p.nblocks, p.splits);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_mfma_kernel<1>), grid, dim3(256), 0, stream, x, dy, slabs, Cin, Cout, H, W, p.xtiles, p.nblocks, p.splits);
    const size_t n = p.slab_floats;
    const size_t work = (n / 4 + 255) / 256 + 1;
    const int blocks = (int)(work < 4096 ? work : 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, slabs, n, p.splits, dw_packed);
    return frcnn_launch_status();
}

int frcnn_conv_wgrad_f32s(const float *x, const float *dy, float *dw_packed, int Cin, int Cout, int H, int W, void *workspace, size_t workspace_bytes,
                          void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !dy || !dw_packed || Cin < 1 || Cout < 1 || H < 1 || W < 1) return FRCNN_ERR_INVALID;
    if ((size_t)Cin * H * W * 4 >= (1ull << 31) || (size_t)Cout * H * W * 4 >= (1ull << 31)) return FRCNN_ERR_INVALID;
    WgradPlan p = plan_wgrad(Cin, Cout, H, W, 3);
    if (!workspace || workspace_bytes < p.slab_floats * p.splits * sizeof(float)) return FRCNN_ERR_INVALID;   // the fp32 kernel's workspace fits
    p.nblocks = p.xtiles * frcnn_cdiv(H, kSRows);                                   // this kernel's tiles are kSRows x 32 px
    int s = frcnn_cdiv(frcnn_cu_count(), p.ci_tiles * p.co_tiles);                   // ONE workgroup per CU
    const char *se = frcnn_tune("FRCNN_WGRAD_F32S_SPLITS");
    if (se && atoi(se) > 0) s = atoi(se);
    if (s > p.splits) s = p.splits;
    if (s > p.nblocks) s = p.nblocks;
    if (s < 1) s = 1;
    float *slabs = (float *)workspace;
    hipLaunchKernelGGL(conv_wgrad_f32s_kernel, dim3(p.ci_tiles, p.co_tiles, s), dim3(256), 0, stream, x, dy, slabs, Cin, Cout, H, W, p.xtiles, p.nblocks, s);
    const size_t n = p.slab_floats;
    const size_t work = (n / 4 + 255) / 256 + 1;
    const int blocks = (int)(work < 4096 ? work : 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, slabs, n, s, dw_packed);
    return frcnn_launch_status();
}

int frcnn_sgd_momentum_wd(float *w, const float *grad, float *velocity, size_t n, float lr, float momentum, float weight_decay, void *stream) {
    if (n == 0) return FRCNN_OK;
    if (!w || !grad || !velocity) return FRCNN_ERR_INVALID;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(sgd_momentum_wd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, grad, velocity, n, lr, momentum, weight_decay);
    return frcnn_launch_status();
}

int frcnn_transpose_f32(const float *src, int rows, int cols, float *dst, void *stream) {
    if (!src || !dst || rows < 1 || cols < 1) return FRCNN_ERR_INVALID;
    hipLaunchKernelGGL(transpose_kernel, dim3(frcnn_cdiv(cols, 64), frcnn_cdiv(rows, 64)), dim3(256), 0, (hipStream_t)stream, src, rows, cols, dst);
    return frcnn_launch_status();
}

}  // extern "C"
