/*
 * oracle/c/frcnn_oracle.c -- CPU restatement of the reference's native hot-path arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (chainer-faster-rcnn_amd/) may link,
 * load or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the reported CPU baseline.
 *
 * Each function restates one reference routine in plain scalar C, in the same
 * floating-point types and the same operation order, so results are bit-identical to the
 * reference's compiled Cython (pinned by tests/test_oracle_pinned.py against
 * oracle/_ref/*.so built from /root/reference and against tests/golden/*.npz).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/build.py).  -ffp-contract=off
 * matters: the reference's C is compiled without FMA contraction of a*b+c.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------
 * Greedy IoU NMS.  Follows /root/reference/models/cpu_nms.pyx:18-69.
 *   dets   : (n,5) float32 rows [x1,y1,x2,y2,score]                   (cpu_nms.pyx:18-23)
 *   order  : (n,) int64 = scores.argsort()[::-1], computed by the caller with NumPy so the
 *            tie order is NumPy's own                                  (cpu_nms.pyx:26)
 *   thresh : a Python float in the reference => the compare `ovr >= thresh` is done in
 *            DOUBLE on the float32 `ovr`                               (cpu_nms.pyx:18,66)
 *   keep   : out, indices into dets, in visiting order; returns count  (cpu_nms.pyx:43-48,69)
 * areas are float32: (x2 - x1 + 1) * (y2 - y1 + 1)                     (cpu_nms.pyx:25)
 * ------------------------------------------------------------------------------------- */
static inline float fmax32(float a, float b) { return a >= b ? a : b; } /* cpu_nms.pyx:12-13 */
static inline float fmin32(float a, float b) { return a <= b ? a : b; } /* cpu_nms.pyx:15-16 */

int64_t oracle_cpu_nms(const float *dets, int64_t n, const int64_t *order, double thresh, int64_t *keep)
{
    float *areas = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    uint8_t *suppressed = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    int64_t nkeep = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float *d = dets + 5 * i;
        volatile float w = d[2] - d[0] + 1.0f;
        volatile float h = d[3] - d[1] + 1.0f;
        areas[i] = w * h;
    }
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = order[_i];
        if (suppressed[i]) continue;
        keep[nkeep++] = i;
        const float ix1 = dets[5 * i + 0], iy1 = dets[5 * i + 1];
        const float ix2 = dets[5 * i + 2], iy2 = dets[5 * i + 3];
        const float iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j];
            if (suppressed[j]) continue;
            float xx1 = fmax32(ix1, dets[5 * j + 0]);
            float yy1 = fmax32(iy1, dets[5 * j + 1]);
            float xx2 = fmin32(ix2, dets[5 * j + 2]);
            float yy2 = fmin32(iy2, dets[5 * j + 3]);
            float w = fmax32(0.0f, xx2 - xx1 + 1.0f);
            float h = fmax32(0.0f, yy2 - yy1 + 1.0f);
            float inter = w * h;
            float ovr = inter / (iarea + areas[j] - inter);
            if ((double)ovr >= thresh) suppressed[j] = 1;
        }
    }
    free(areas);
    free(suppressed);
    return nkeep;
}

/* ---------------------------------------------------------------------------------------
 * Dense IoU matrix in float64.  Follows /root/reference/models/bbox.pyx:16-56.
 * boxes (N,4), query (K,4) float64 -> overlaps (N,K) float64, 0 where no overlap.
 * ------------------------------------------------------------------------------------- */
void oracle_bbox_overlaps(const double *boxes, int64_t N, const double *query, int64_t K, double *overlaps)
{
    memset(overlaps, 0, sizeof(double) * (size_t)(N * K));
    for (int64_t k = 0; k < K; ++k) {
        const double *q = query + 4 * k;
        double box_area = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);          /* bbox.pyx:34-37 */
        for (int64_t n = 0; n < N; ++n) {
            const double *b = boxes + 4 * n;
            double iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1; /* :39-42 */
            if (iw > 0) {
                double ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1; /* :44-47 */
                if (ih > 0) {
                    double ua = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + box_area - iw * ih; /* :49-53 */
                    overlaps[n * K + k] = iw * ih / ua;                    /* bbox.pyx:54 */
                }
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------
 * RoI max pooling, forward.  The reference calls chainer.functions.roi_pooling_2d
 * (/root/reference/models/faster_rcnn.py:125-126); Chainer is an un-vendored, un-pinned
 * dependency (v1.22-1.24 era), so this restates its published CPU algorithm
 * (chainer/functions/pooling/roi_pooling_2d.py, forward_cpu + _roi_pooling_slice):
 *   xs = round_half_even(x1*scale) ... ; rw = max(xe-xs+1,1); stride = rw/outw (double)
 *   bin = [floor(p*stride)+xs, ceil((p+1)*stride)+xs) clamped to [0,W]
 *   out = max over bin (first max in row-major order wins -> argmax = h*W+w),
 *   empty bin -> 0 / argmax -1 (forward_gpu's definition; forward_cpu leaves it undefined).
 *   NaN: this scan is forward_gpu's rule (the first cell seeds, a later cell wins only under a strict `>`: a NaN in the first cell stays, a NaN
 *   elsewhere never wins).  forward_cpu's numpy.max / numpy.argmax PROPAGATE a NaN instead.  The deviation is deliberate and stated in
 *   include/frcnn_hip.h (RoIPooling2D block); the two rules agree on NaN-free maps and tests/test_oracle_pinned.py pins the difference.
 * PARITY UNPINNED by the reference's tests.
 * x (N,C,H,W) f32; rois (R,5) f32 [batch,x1,y1,x2,y2]; y (R,C,outh,outw); argmax int32 or NULL.
 * ------------------------------------------------------------------------------------- */
static inline int rint_half_even(float v) { return (int)nearbyintf(v); }

void oracle_roi_pool_fwd(const float *x, int64_t C, int64_t H, int64_t W, const float *rois, int64_t R,
                         int outh, int outw, float scale, float *y, int32_t *argmax)
{
    for (int64_t r = 0; r < R; ++r) {
        const float *roi = rois + 5 * r;
        int b = (int)roi[0];
        int xs = rint_half_even(roi[1] * scale), ys = rint_half_even(roi[2] * scale);
        int xe = rint_half_even(roi[3] * scale), ye = rint_half_even(roi[4] * scale);
        int rw = xe - xs + 1 > 1 ? xe - xs + 1 : 1;
        int rh = ye - ys + 1 > 1 ? ye - ys + 1 : 1;
        double sh = 1. * rh / outh, sw = 1. * rw / outw;
        for (int ph = 0; ph < outh; ++ph) {
            int hs = (int)floor(ph * sh) + ys, he = (int)ceil((ph + 1) * sh) + ys;
            hs = hs < 0 ? 0 : (hs > H ? (int)H : hs);
            he = he < 0 ? 0 : (he > H ? (int)H : he);
            for (int pw = 0; pw < outw; ++pw) {
                int ws = (int)floor(pw * sw) + xs, we = (int)ceil((pw + 1) * sw) + xs;
                ws = ws < 0 ? 0 : (ws > W ? (int)W : ws);
                we = we < 0 ? 0 : (we > W ? (int)W : we);
                for (int64_t c = 0; c < C; ++c) {
                    const float *xc = x + ((int64_t)b * C + c) * H * W;
                    int64_t o = ((r * C + c) * outh + ph) * outw + pw;
                    if (he <= hs || we <= ws) {
                        y[o] = 0.0f;
                        if (argmax) argmax[o] = -1;
                        continue;
                    }
                    float m = xc[hs * W + ws];
                    int32_t mi = (int32_t)(hs * W + ws);
                    for (int h = hs; h < he; ++h)
                        for (int w = ws; w < we; ++w)
                            if (xc[h * W + w] > m) { m = xc[h * W + w]; mi = (int32_t)(h * W + w); }
                    y[o] = m;
                    if (argmax) argmax[o] = mi;
                }
            }
        }
    }
}

/* RoI max pooling, backward (chainer roi_pooling_2d.py backward_cpu): every (roi,c,ph,pw)
 * adds its top gradient to bottom_diff[batch,c] at its stored argmax; argmax<0 contributes
 * nothing.  Accumulation order is roi-major, as the reference's Python loop does. */
void oracle_roi_pool_bwd(const float *dy, const int32_t *argmax, const float *rois, int64_t R, int64_t C,
                         int64_t H, int64_t W, int outh, int outw, int64_t N, float *dx)
{
    memset(dx, 0, sizeof(float) * (size_t)(N * C * H * W));
    for (int64_t r = 0; r < R; ++r) {
        int b = (int)rois[5 * r];
        for (int64_t c = 0; c < C; ++c)
            for (int p = 0; p < outh * outw; ++p) {
                int64_t o = (r * C + c) * outh * outw + p;
                if (argmax[o] >= 0) dx[((int64_t)b * C + c) * H * W + argmax[o]] += dy[o];
            }
    }
}
