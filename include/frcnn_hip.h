/*
 * frcnn_hip.h -- C ABI of libfrcnn_hip.so: the MI355X (gfx950) Faster R-CNN hot path.
 *
 * Drop-in boundary for mitmul/chainer-faster-rcnn's per-image forward (and RPN training step).
 * The reference's only C FFI on this path is
 *     void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num,
 *               int boxes_dim, float nms_overlap_thresh, int device_id);      (models/gpu_nms.hpp:9-10)
 * plus the Cython entry points cpu_nms(dets, thresh) (models/cpu_nms.pyx:18) and
 * bbox_overlaps(boxes, query_boxes) (models/bbox.pyx:16); everything else on the path is a Python
 * object boundary (ProposalLayer / AnchorTargetLayer / F.roi_pooling_2d / L.Convolution2D ...).
 * Each entry point below names the reference interface it replaces.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; no torch / chainer types.
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *     never calls hipSetDevice, hipMalloc or hipFree, never synchronises unless documented;
 *   - scratch memory is caller-owned: query the size with the matching *_workspace_bytes();
 *   - returns 0 on success, FRCNN_ERR_INVALID (-1) for bad arguments, or -(1000 + hipError_t) when a
 *     launch fails.  Errors are returned, never printed-and-swallowed (contrast nms_kernel.cu:12-19).
 *   - tensors are C-contiguous, NCHW, batch 1 (the reference asserts batch==1: faster_rcnn.py:77).
 */
#ifndef FRCNN_HIP_H
#define FRCNN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRCNN_OK 0
#define FRCNN_ERR_INVALID (-1)
/* the arguments are valid but THIS entry's kernel form does not cover the shape (documented per entry, with the detour to take): -2.
 * Distinct from FRCNN_ERR_INVALID so that a caller's detour can never hide a genuinely bad argument. */
#define FRCNN_ERR_UNSUPPORTED (-2)

/* Library identification: returns the ABI version (bumped on any signature change). */
int frcnn_abi_version(void);
/* Number of HIP devices visible to the library (hipGetDeviceCount), or -(1000+hipError_t). */
int frcnn_device_count(void);

/* ---- tuning registry (csrc/frcnn_tune.h) ---------------------------------------------------------------
 * Every A/B knob of the launchers (kernel-form picks such as FRCNN_BF16_DMA, FRCNN_ROI_BWD, FRCNN_NMS_SCAN; DESIGN.md 7b lists them) lives in ONE
 * table inside the library.  It is filled once, when the library is loaded, from the FRCNN_* variables the process environment holds at that
 * moment; no entry point reads the environment afterwards (a later setenv() has no effect and cannot race a launch).  With no entry a launcher
 * takes its measured default; no key selects a CPU path.  The reference has no counterpart (its only knob on this path is the process-wide
 * cudaSetDevice of nms_kernel.cu:80-89).
 *   frcnn_set_tuning(key, value)   key must start with "FRCNN_" (47 characters at most), value at most 79 characters; value NULL removes
 *                                  the entry.  0, or FRCNN_ERR_INVALID (key / value too long, a 97th key).  Thread-safe: values are immutable
 *                                  interned strings and a set swaps a pointer, so a launch on another thread reads the old or the new value,
 *                                  never a torn one.  FRCNN_* environment entries the table cannot hold are skipped at load, not an error.
 *   frcnn_get_tuning(key, out, n)  copies the value (NUL-terminated, truncated to n) and returns its length + 1; 0 when the key is unset.
 *   frcnn_reset_tuning()           back to the load-time snapshot.
 */
int frcnn_set_tuning(const char *key, const char *value);
int frcnn_get_tuning(const char *key, char *value_out, int capacity);
int frcnn_reset_tuning(void);

/* ---- greedy IoU NMS ------------------------------------------------------------------------------
 * Replaces cpu_nms(dets, thresh) (models/cpu_nms.pyx:18-69; live caller models/proposal_layer.py:176-178,
 * second caller forward.py:54) and the dead gpu_nms/_nms (models/gpu_nms.hpp:9-10).
 *   dets    (n,5) f32 rows [x1,y1,x2,y2,score], any order (re-sorted by score descending internally,
 *           ties by ascending index, as cpu_nms.pyx:26 does with NumPy's argsort)
 *   thresh  compared in DOUBLE against the float32 IoU with `>=` (cpu_nms.pyx:18,66)
 *   max_out stop after this many survivors (<=0: all) -- the caller's keep[:post_nms_top_n]
 *   keep    (cap) int32 out, cap = max_out>0 ? min(n,max_out) : n: indices into dets in descending-score
 *           order; entries past n_keep are set to -1
 *   n_keep  (1) int32 out
 */
size_t frcnn_nms_workspace_bytes(int n);
int frcnn_nms(const float *dets, int n, double thresh, int max_out, int32_t *keep, int32_t *n_keep,
              void *workspace, size_t workspace_bytes, void *stream);

/* Batched NMS over `groups` independent problems laid out back to back ((groups, n, 5) dets,
 * (groups, n) keep, (groups) n_keep): forward.py:48-58's 20 per-class cpu_nms(thresh 0.3) calls. */
size_t frcnn_nms_batched_workspace_bytes(int groups, int n);
int frcnn_nms_batched(const float *dets, int groups, int n, double thresh, int max_out, int32_t *keep,
                      int32_t *n_keep, void *workspace, size_t workspace_bytes, void *stream);

/* The reference's own C FFI, signature unchanged (models/gpu_nms.hpp:9-10; bound by models/gpu_nms.pyx:16-31, which links against
 * this symbol as is): HOST pointers, synchronous, device memory allocated and freed inside the call, `device_id` selected for the
 * duration of the call only (the caller's current device is restored).  boxes_host: boxes_num rows of boxes_dim >= 5 floats
 * [x1,y1,x2,y2,score,..] pre-sorted by descending score (gpu_nms.pyx:25-28); keep_out (capacity boxes_num): surviving row indices.
 * Errors are reported as *num_out = -1 (never printed and swallowed).  Suppression rule = cpu_nms.pyx's `(double)iou >= thresh`
 * with thresh recovered as the shortest decimal that rounds to the float passed in (0.7f -> 0.7): see csrc/nms_host.hip. */
void _nms(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim,
          float nms_overlap_thresh, int device_id);

/* ---- ProposalLayer ---------------------------------------------------------------------------------
 * Replaces ProposalLayer.__call__ (models/proposal_layer.py:102-198): anchor enumeration (:200-221),
 * bbox_transform_inv (bbox_transform.py:41-76), clip_boxes (:79-99), filter_boxes (:102-109), fg-score
 * gather (proposal_layer.py:150-154), descending top-K sort (:156-170), cpu_nms (:176-178), top-N
 * (:189-193) -- all on device, no host round trip.
 *   rpn_cls_prob (2A,H,W) f32, fg scores in channels [A,2A); rpn_bbox_pred (4A,H,W) f32, channel a*4+c
 *   anchors_host (A,4) float64 HOST pointer (generate_anchors output; copied by value into the launch)
 *   im_h, im_w   img_info (H, W) exactly as the caller passes it (forward.py:93 passes (H,H))
 *   rois (cap,4) f32 out; probs (cap) f32 out; n_out (1) int32 out; cap = post_nms_top_n if >0 else
 *   min(A*H*W, pre_nms_top_n); rows past n_out are zero-filled so fixed-capacity consumers stay defined
 *   src_index (cap) int32 out or NULL: index of each RoI in the (H*W*A) anchor enumeration (-1 past n_out)
 * Arithmetic: the reference's operation by operation in float32 (separate multiply and add), with exp(dw), exp(dh) CORRECTLY ROUNDED (double exp, rounded
 * once).  bbox_transform.py:63-64 calls np.exp on float32, whose value depends on the NumPy build and CPU (a SIMD polynomial of up to 2.5 ulp since 1.17;
 * libm's expf before): against a correctly rounded exp the RoIs are the reference's bit for bit, against a given host's NumPy they sit within 4 ulp and an
 * NMS decision whose IoU lies within ~1e-6 of the threshold may fall the other way (tests/test_coord_margins.py).
 */
size_t frcnn_proposals_workspace_bytes(int A, int H, int W, int pre_nms_top_n);
int frcnn_proposals(const float *rpn_cls_prob, const float *rpn_bbox_pred, int A, int H, int W,
                    const double *anchors_host, int feat_stride, int im_h, int im_w, float min_size,
                    int pre_nms_top_n, int post_nms_top_n, double nms_thresh, float *rois, float *probs,
                    int32_t *n_out, int32_t *src_index, void *workspace, size_t workspace_bytes,
                    void *stream);

/* ---- RoIPooling2D ----------------------------------------------------------------------------------
 * Replaces F.roi_pooling_2d(x, rois, outh, outw, spatial_scale) (call site models/faster_rcnn.py:125-126;
 * Chainer v1 ROIPooling2D forward / backward).  x (C,H,W) f32 NCHW batch 1 (the batch index in rois is
 * ignored: the reference asserts batch==1); rois (R,5) f32 [batch,x1,y1,x2,y2]; y (R,C,outh,outw) f32;
 * argmax same shape int32 (flat h*W+w, -1 for an empty bin) or NULL (inference).  outh,outw <= 7.
 * RoI COORDINATE DOMAIN: every finite RoI with |v * spatial_scale| <= 2^24 for its four coordinates is pooled exactly as the reference's
 * arithmetic pools it -- negative corners, reversed corners (extent max(., 1)), one-point RoIs, RoIs wholly outside the map (all bins empty:
 * 0 / -1) and RoIs far larger than the map (e.g. [-1e4, -1e4, 2e4, 2e4]) included; tests/parity_cases.py roi_case holds one of each.  Beyond
 * 2^24 cells (or NaN / inf coordinates) the float -> int conversion saturates; the call still returns 0 and writes finite in-range output,
 * but which bins count as empty is unspecified (the reference's Python integers do not saturate).  The device forms cannot validate device
 * rois without a host round trip, so nothing is rejected.
 *   frcnn_roi_pool_fwd_chw  the fast path: NCHW in, channel planes resident in LDS, no transpose (falls back to
 *                           frcnn_chw_to_hwc into `workspace` + frcnn_roi_pool_fwd_hwc when one H*W plane exceeds
 *                           the LDS budget; workspace may be NULL otherwise)
 *   frcnn_roi_pool_fwd      = frcnn_roi_pool_fwd_chw with roi_cols = 5 (Chainer's (R,5) rois)
 *   frcnn_roi_pool_fwd_hwc  takes the feature map channel-last, xt (H*W, C) -- any map size;
 *                           roi_cols = 5 ([batch,x1,y1,x2,y2] rows) or 4 (ProposalLayer's bare (R,4)
 *                           output, i.e. the concat at faster_rcnn.py:123-124 folded into the read)
 * NaN RULE -- A STATED DEVIATION from Chainer's CPU path.  forward_cpu takes numpy.max / numpy.argmax over the bin, which PROPAGATE a NaN (any NaN in
 * the bin -> NaN out, argmax = the first NaN).  Every kernel here (and oracle/c/frcnn_oracle.c, which the parity tests compare with) follows
 * forward_gpu's scan instead: the bin's first cell seeds the maximum and a later cell replaces it only under a strict `>` -- a NaN in the FIRST cell
 * stays, a NaN elsewhere never wins.  The two rules agree on every NaN-free map (tests/test_oracle_pinned.py: the C restatement == the NumPy twin of
 * forward_cpu; its test_roi_pool_nan_rule_is_the_stated_deviation pins where they differ).  A map with NaNs means the step has already diverged; the
 * scan rule keeps the kernel's fast path (IEEE maxNum, v_max_f32) exact on NaN-free maps without a per-cell NaN test.
 */
size_t frcnn_roi_pool_workspace_bytes(int C, int H, int W);
int frcnn_chw_to_hwc(const float *x, int C, int H, int W, float *xt, void *stream);
int frcnn_roi_pool_fwd_hwc(const float *xt, int C, int H, int W, const float *rois, int R, int roi_cols,
                           int outh, int outw, float spatial_scale, float *y, int32_t *argmax, void *stream);
int frcnn_roi_pool_fwd_chw(const float *x, int C, int H, int W, const float *rois, int R, int roi_cols,
                           int outh, int outw, float spatial_scale, float *y, int32_t *argmax, void *workspace,
                           size_t workspace_bytes, void *stream);
/* the same pooling with the fp32 maxima written as their three bf16 terms, y_parts = [3][R][C*outh*outw] (h + m + l = the fp32
 * value, exactly): the split tensor frcnn_linear_f32s reads (cell-major kernel only: maps up to 76 x 64) */
int frcnn_roi_pool_fwd_chw_f32s(const float *x, int C, int H, int W, const float *rois, int R, int roi_cols, int outh, int outw,
                                float spatial_scale, uint16_t *y_parts, void *stream);
/* the same with the pooled values written as raw bf16 (one rounding of the fp32 maximum): the input of the bf16 FC head
 * (BASELINE config 3) without the fp32 pool5 round trip.  LDS-resident kernels only (the cell-major kernel for maps up to 76 x 64,
 * else the plane kernel while a plane fits in LDS); FRCNN_ERR_UNSUPPORTED beyond that: pool in fp32 and convert with frcnn_f32_to_bf16. */
int frcnn_roi_pool_fwd_chw_bf16(const float *x, int C, int H, int W, const float *rois, int R, int roi_cols, int outh,
                                int outw, float spatial_scale, uint16_t *y, void *stream);
/* the same pooling straight from the bf16 chain's channel-blocked map x_blk = [CP/16][H][W][16] bf16 (CP = C rounded up to 16): a
 * cell's eight channels are one 16-byte load and no fp32 NCHW copy of conv5_3 has to exist.  y = (R, C, outh, outw) fp32, or raw
 * bf16 bits when out_bf16 (values are bf16 to begin with: exact).  Cell-major kernel only (maps up to 76 x 64), else
 * FRCNN_ERR_UNSUPPORTED: frcnn_bf16_to_nchw_f32 + frcnn_roi_pool_fwd_chw then. */
int frcnn_roi_pool_fwd_blk_bf16(const uint16_t *x_blk, int C, int H, int W, const float *rois, int R, int roi_cols, int outh,
                                int outw, float spatial_scale, void *y, int out_bf16, void *stream);
int frcnn_roi_pool_fwd(const float *x, int C, int H, int W, const float *rois, int R, int outh, int outw,
                       float spatial_scale, float *y, int32_t *argmax, void *workspace,
                       size_t workspace_bytes, void *stream);
/* backward (Chainer roi_pooling_2d backward_cpu: dx[c, argmax] += dy over every (roi, c, bin) with argmax >= 0): the sums are accumulated with float atomics
 * (in LDS planes for maps up to 38 x 64 cells, in memory beyond), so their ORDER -- the reference's loop is RoI-major -- and with it the last bits of dx are
 * not fixed from run to run (measured: 1.4e-6 of the largest entry between two runs of a stage-2 step; tests/train_cases.py:check_trainers_across_image_sizes).
 * Every other reduction of the two training steps has a fixed order. */
int frcnn_roi_pool_bwd(const float *dy, const int32_t *argmax, int R, int C, int H, int W, int outh,
                       int outw, float *dx, void *stream);

/* ---- convolution stack -----------------------------------------------------------------------------
 * Replaces L.Convolution2D(ci,co,3,1,1)+F.ReLU (models/vgg16.py:39-68, region_proposal_network.py:53,117)
 * and F.MaxPooling2D(2,2) (cover_all => ceil-mode) on the batch-1 NCHW path.
 *   frcnn_pack_conv3x3_w: (Cout,Cin,3,3) Chainer layout -> packed [(ci*9+tap)][Cout] f32 (once, at load)
 *   frcnn_conv3x3_f32:    y = act(conv3x3(x, pad 1, stride 1) + b); x (Cin,H,W), y (Cout,H,W);
 *                         f32 operands, exact-f32 MFMA accumulation (v_mfma_f32_32x32x2_f32); the
 *                         workspace holds the partial tiles of the stream-K work distribution
 *   frcnn_maxpool2x2_f32: (C,H,W) -> (C,ceil(H/2),ceil(W/2))
 */
int frcnn_pack_conv3x3_w(const float *w, int Cout, int Cin, float *w_packed, void *stream);
/*   workspace contract:   the first 64 KB of the conv workspace are the tile counters of the stream-K
 *                         distribution.  Zero them ONCE after allocating the workspace
 *                         (frcnn_conv3x3_workspace_init); every launch leaves them zeroed again (the last
 *                         arriver of a split tile resets its counter), so no launch pays a memset.  Do not
 *                         hand the same scratch to other entry points, and do not share it between streams.
 */
size_t frcnn_conv3x3_workspace_bytes(int Cin, int Cout, int H, int W);
int frcnn_conv3x3_workspace_init(void *workspace, size_t workspace_bytes, void *stream);
int frcnn_conv3x3_f32(const float *x, const float *w_packed, const float *bias, float *y, int Cin,
                      int Cout, int H, int W, int relu, void *workspace, size_t workspace_bytes,
                      void *stream);
/* same, with an explicit work decomposition: cfg = tile-shape id (0..11) + 100 * mode, mode 0 = one
 * workgroup per tile, 1 = stream-K when the tile count is ragged, 2 = stream-K forced; -1 = automatic.
 * The tuning hook scripts/conv_sweep.py drives.  workspace may be NULL (then whole tiles are used). */
int frcnn_conv3x3_f32_cfg(const float *x, const float *w_packed, const float *bias, float *y, int Cin,
                          int Cout, int H, int W, int relu, int cfg, void *workspace,
                          size_t workspace_bytes, void *stream);
int frcnn_maxpool2x2_f32(const float *x, float *y, int C, int H, int W, void *stream);

/* ---- RPN 1x1 heads + the reference's 18-way softmax ------------------------------------------------
 * Replaces rpn_cls_score / F.softmax / rpn_bbox_pred (models/region_proposal_network.py:118-120).
 *   frcnn_rpn_heads_pack (once, at load): w_cls (2A,Cmid), w_bbox (4A,Cmid) in Chainer's (out,in,1,1)
 *       layout -> one stacked, zero-padded (Cmid, NP) matrix + (NP) bias, NP = frcnn_rpn_heads_padded_channels(A)
 *   frcnn_rpn_heads_f32: h (Cmid,H,W) -> raw (NP,H,W): rows [0,2A) = rpn_cls_score, rows [2A,6A) =
 *       rpn_bbox_pred (contiguous NCHW blocks, use them in place); cls_prob (2A,H,W) = softmax over ALL
 *       2A score channels (axis 1), as the reference does.  One launch for A <= 10 (both heads as a K-split 1x1
 *       convolution on fp32 MFMA + the softmax on the tile); larger A: the convolution kernel + a softmax launch.
 */
int frcnn_rpn_heads_padded_channels(int A);
int frcnn_rpn_heads_pack(const float *w_cls, const float *b_cls, const float *w_bbox, const float *b_bbox,
                         int Cmid, int A, float *w_packed, float *b_packed, void *stream);
int frcnn_rpn_heads_f32(const float *h, int Cmid, int H, int W, int A, const float *w_packed,
                        const float *b_packed, float *raw, float *cls_prob, void *stream);

/* ---- fully connected head --------------------------------------------------------------------------
 * Replaces L.Linear + F.relu (models/faster_rcnn.py:33-36,127-134): y(M,N) = act(x(M,K) @ W(N,K)^T + b).
 * bias (N) or NULL = no bias term (the training step's dW = dy^T x and dx = dy W).  Split-K partial slabs live in the caller's workspace; it is not
 * touched (and may be NULL) when bias == NULL, relu == 0 and the plan has a single slab: the GEMM then writes y itself and no combine pass runs.
 */
size_t frcnn_linear_workspace_bytes(int M, int N, int K);
int frcnn_linear_f32(const float *x, const float *w, const float *bias, float *y, int M, int N, int K,
                     int relu, void *workspace, size_t workspace_bytes, void *stream);

/* Replaces bbox_transform_inv + clip_boxes + F.softmax on the head outputs (faster_rcnn.py:175-178):
 * boxes (R,4), deltas (R,4*ncls) -> pred_boxes (R,4*ncls); cls_score (R,ncls) -> cls_prob (R,ncls). */
int frcnn_head_decode(const float *boxes, const float *deltas, const float *cls_score, int R, int ncls,
                      int im_h, int im_w, float *pred_boxes, float *cls_prob, void *stream);
/* the same for the stacked head GEMM (cls_score and bbox_pred as ONE L.Linear whose weight rows are [cls_score.W; 0...; bbox_pred.W]):
 * out (R, ld) holds the class scores in columns [0, ncls) and the 4*ncls deltas from column dcol; ld % 4 == dcol % 4 == 0 */
int frcnn_head_decode_stacked(const float *boxes, const float *out, int ld, int dcol, int R, int ncls, int im_h, int im_w,
                              float *pred_boxes, float *cls_prob, void *stream);
/* forward.py:50-53: per-class detection rows for the 20 per-class NMS problems --
 * dets (ncls-1, R, 5) = [pred_boxes[:, 4c:4c+4], cls_prob[:, c]] for c = 1..ncls-1; feed frcnn_nms_batched(thresh 0.3). */
int frcnn_class_dets(const float *cls_prob, const float *pred_boxes, int R, int ncls, float *dets, void *stream);
/* forward.py:33-45 img_preprocessing: img (H,W,C) uint8 as cv.imread returns it -> out (C,OH,OW) f32 =
 * cv.resize(float32(img) - means, fx=fy=im_scale, INTER_LINEAR) transposed to CHW; OH = round(H*im_scale), OW = round(W*im_scale)
 * are computed by the caller (the upload is 1.8 MB of uint8 instead of 7.2 MB of float32). */
int frcnn_preprocess_u8(const uint8_t *img, int H, int W, int C, const double *means_host, double im_scale, int OH, int OW,
                        float *out, void *stream);
/* the three pieces on their own: bbox_transform_inv (bbox_transform.py:41-76), clip_boxes (:79-99, in
 * place on n_boxes x 4 floats), and a row-wise softmax (F.softmax on (R,n)) */
int frcnn_bbox_transform_inv(const float *boxes, const float *deltas, int R, int ncls, float *pred_boxes,
                             void *stream);
int frcnn_clip_boxes(float *boxes, int n_boxes, int im_h, int im_w, void *stream);
int frcnn_softmax_rows(const float *scores, int R, int n, float *probs, void *stream);

/* generic form of the convolution entry: ksize 1 or 3 (stride 1, pad ksize/2), act 0 = none, 1 = ReLU,
 * 4 = ReLU then F.MaxPooling2D(2,2) (cover_all) fused into the epilogue: y is (Cout, ceil(H/2), ceil(W/2)) (ksize 3),
 * 5 = the training form of 4: as 4, plus the window cell every pooled value came from (0..3 = (row, column) scan order, first maximum) as one
 *     byte per pooled value at `mask`, reinterpreted as unsigned char[Cout][ceil(H/2)][ceil(W/2)] (frcnn_maxpool2x2_bwd_idx_f32 reads it),
 * 3 = y = relu(conv + bias + mask) (residual add), 2 = y = (mask > 0) ? conv + bias : 0  -- the input-gradient convolution of the backward pass with the producing
 * ReLU's mask fused in (mask has y's shape).  Cout % 64 == 0. */
int frcnn_conv_f32_ex(const float *x, const float *w_packed, const float *bias, const float *mask, float *y, int Cin,
                      int Cout, int H, int W, int ksize, int act, void *workspace, size_t workspace_bytes,
                      void *stream);

/* ---- fp32 convolution on the bf16 matrix cores ("split" tensors) ------------------------------------------
 * Same reference interface as frcnn_conv3x3_f32 (L.Convolution2D(ci, co, 3, 1, 1) + F.relu [+ F.max_pooling_2d(2,2)]:
 * models/vgg16.py:39-82, region_proposal_network.py:53), fp32 results: every fp32 operand is carried as three bf16 terms
 * h + m + l (exact) and a product block is six v_mfma_f32_32x32x16_bf16 (h.h, h.m, m.h, h.l, l.h, m.m; what is dropped is
 * below 2^-24 of the product), fp32 accumulation.  A split tensor is [3 parts][CP/16][H][W][16] bf16 (CP = C rounded up to 16),
 * split weights [3][CinP/16][tap][CoutP][16].  out_mode 0: y = split tensor (CoutP, H, W); 2: ReLU + 2x2 max-pool (cover_all,
 * in fp32 before the split) fused, y = split tensor (CoutP, ceil(H/2), ceil(W/2)); 1: y = (Cout,H,W) fp32 NCHW. */
int frcnn_f32s_pack_conv_w(const float *w, int Cout, int Cin, uint16_t *w_packed, void *stream);
int frcnn_f32s_from_nchw_f32(const float *x, int C, int H, int W, uint16_t *y, void *stream);
int frcnn_f32s_to_nchw_f32(const uint16_t *x, int C, int H, int W, float *y, void *stream);
int frcnn_conv3x3_f32s(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int Cin, int Cout, int H,
                       int W, int relu, int out_mode, void *stream);
/* first layer (Cin <= 3, Cout <= 64; models/vgg16.py:39 conv1_1): x = the fp32 NCHW image, w = Chainer's (Cout,Cin,3,3) fp32 as it
 * is, y = split tensor (CoutP, H, W).  A persistent launch (as many workgroups as the chip seats, striding over the tiles); the
 * outputs sit behind 32-bit buffer ranges: H * W * 128 bytes (one part) must stay below 2 GiB, else FRCNN_ERR_INVALID. */
int frcnn_conv1_f32s(const float *x, const float *w, const float *bias, uint16_t *y, int Cin, int Cout, int H, int W, int relu,
                     void *stream);
/* the same kernel in plain bf16 arithmetic (operands rounded to bf16, fp32 accumulation, y = [CoutP/16][H][W][16] bf16): the first
 * layer of the bf16 chain without the fp32 -> blocked-bf16 image conversion and without 13 padded channels of MFMA work */
int frcnn_conv1_bf16(const float *x, const float *w, const float *bias, uint16_t *y, int Cin, int Cout, int H, int W, int relu,
                     void *stream);
/* and in native fp32 arithmetic (v_mfma_f32_32x32x2_f32): conv1_1 of the fp32 chain -- x = fp32 NCHW image, w_packed = the packed
 * weights of frcnn_pack_conv3x3_w [(ci * 9 + tap)][Cout], y = (Cout, H, W) fp32 NCHW.  frcnn_conv3x3_f32 hands such layers
 * (Cin <= 3, Cout <= 64, plain bias / ReLU epilogue) to it on its own. */
int frcnn_conv1_f32(const float *x, const float *w_packed, const float *bias, float *y, int Cin, int Cout, int H, int W, int relu,
                    void *stream);
/* the same with a workspace (frcnn_conv_f32s_workspace_bytes; its first 64 KB zeroed ONCE by frcnn_conv_f32s_workspace_init --
 * every launch leaves them zero): lets launches with few tiles (38x63 maps) split their K range over several workgroups */
size_t frcnn_conv_f32s_workspace_bytes(int Cin, int Cout, int H, int W);
int frcnn_conv_f32s_workspace_init(void *workspace, size_t workspace_bytes, void *stream);
int frcnn_conv3x3_f32s_ws(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int Cin, int Cout, int H,
                          int W, int relu, int out_mode, void *workspace, size_t workspace_bytes, void *stream);

/* training forms (train_rpn.py's forward / backward through the trunk): the convolution with the result written as a
 * split tensor (y_split, may be NULL) and / or as fp32 NCHW (y_nchw, may be NULL: the weight-gradient kernel, pooling and the ReLU masks
 * read fp32), optionally masked -- y = (mask > 0) ? y : 0, mask (Cout,H,W) fp32: the input-gradient convolution with the producing
 * ReLU's mask fused in.  frcnn_f32s_pack_from_packed: the trainer's packed fp32 weights [(ci*9+tap)][co] (frcnn_pack_conv3x3_w) ->
 * split weights of the forward (dgrad 0) or of the input-gradient convolution (dgrad 1: channels swapped, taps rotated). */
int frcnn_f32s_pack_from_packed(const float *w_packed_f32, int Cin, int Cout, int dgrad, uint16_t *w_split, void *stream);
/* the same for up to 16 layers in ONE launch (a trainer re-packs every layer every step); w_split_dgrad may be NULL */
typedef struct {
    const float *w_packed_f32;
    uint16_t *w_split_fwd, *w_split_dgrad;
    int Cin, Cout;
} frcnn_f32s_pack_desc;
int frcnn_f32s_pack_many(const frcnn_f32s_pack_desc *layers, int n, void *stream);
/* first layer, training form: w = the trainers' packed fp32 weights [(ci*9+tap)][co]; y_split and (may be NULL) y_nchw (Cout,H,W) fp32 */
int frcnn_conv1_f32s_train(const float *x, const float *w_packed_f32, const float *bias, uint16_t *y_split, float *y_nchw, int Cin,
                           int Cout, int H, int W, int relu, void *stream);
int frcnn_conv3x3_f32s_train(const uint16_t *x, const uint16_t *w_packed, const float *bias, uint16_t *y_split, float *y_nchw,
                             const float *mask, int Cin, int Cout, int H, int W, int relu, void *workspace, size_t workspace_bytes,
                             void *stream);

/* the 3x3 weight gradient (frcnn_conv_wgrad_f32's arguments and result, its workspace) computed as six bf16 MFMA products of the 3-way
 * split fp32 operands: x and dy are the fp32 NCHW tensors as they are, the split happens inside the kernel */
int frcnn_conv_wgrad_f32s(const float *x, const float *dy, float *dw_packed, int Cin, int Cout, int H, int W, void *workspace,
                          size_t workspace_bytes, void *stream);

/* fully connected layers on split tensors (L.Linear + F.relu, models/faster_rcnn.py:33-36,127-134): x = [3][M][K], w = [3][N][K]
 * bf16 parts (frcnn_f32s_split of the fp32 (M,K) / (N,K) arrays), K % 32 == 0; y = (M,N) fp32, or its three parts [3][M][N] when
 * out_split (the next layer's x).  frcnn_f32s_join: parts -> fp32 (h + m + l, exact). */
int frcnn_f32s_split(const float *x, size_t n, uint16_t *y, void *stream);
int frcnn_f32s_join(const uint16_t *x, size_t n, float *y, void *stream);
size_t frcnn_linear_f32s_workspace_bytes(int M, int N, int K);
int frcnn_linear_f32s(const uint16_t *x, const uint16_t *w, const float *bias, void *y, int M, int N, int K, int relu,
                      int out_split, void *workspace, size_t workspace_bytes, void *stream);

/* ---- bf16 convolution stack (BASELINE config 3: bf16 convs / fp32 RoI) -----------------------------------
 * Same reference interface as the fp32 stack (L.Convolution2D + F.relu, F.MaxPooling2D: models/vgg16.py:39-68,
 * region_proposal_network.py:53-57); operands rounded to bf16 (nearest even), fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16.  Activations are channel-blocked [CP/16][H][W][16] bf16, CP = frcnn_bf16_padded_channels(C)
 * (multiple of 16, padding channels zero); weights packed [CinP/16][tap][CoutP][16] bf16 by frcnn_bf16_pack_conv_w from
 * Chainer's (Cout,Cin,k,k) fp32.  out_mode 0: y = [CoutP/16][H][W][16] bf16; out_mode 2: the same with
 * F.MaxPooling2D(2,2) (cover_all) fused behind the ReLU, y = [CoutP/16][ceil(H/2)][ceil(W/2)][16]; out_mode 1: y = (Cout,H,W) fp32 NCHW (what RoI pooling, the
 * 18-way softmax and the proposal kernels consume).  uint16_t = raw bf16 bits. */
int frcnn_bf16_padded_channels(int c);
int frcnn_bf16_pack_conv_w(const float *w, int Cout, int Cin, int ksize, uint16_t *w_packed, void *stream);
int frcnn_bf16_from_nchw_f32(const float *x, int C, int H, int W, uint16_t *y, void *stream);
int frcnn_bf16_to_nchw_f32(const uint16_t *x, int C, int H, int W, float *y, void *stream);
int frcnn_conv_bf16(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int Cin, int Cout, int H,
                    int W, int ksize, int relu, int out_mode, void *stream);
/* The RPN heads of the bf16 chain in one launch (models/region_proposal_network.py:118-120): h = rpn_conv_3x3's channel-blocked
 * bf16 output [CmidP/16][H][W][16], w_packed = frcnn_bf16_pack_conv_w (ksize 1) of the stacked (6A, Cmid) matrix (rpn_cls_score rows,
 * then rpn_bbox_pred rows), bias (6A) fp32 -> raw (6A, H, W) fp32 (rows [0,2A) = rpn_cls_score, [2A,6A) = rpn_bbox_pred) and
 * cls_prob (2A, H, W) = softmax over all 2A score rows.  6A <= 64 (A <= 10).  Same operands as frcnn_conv_bf16 (ksize 1, out_mode 1)
 * + frcnn_softmax_channels_f32; the fp32 additions run in a different order. */
int frcnn_rpn_heads_bf16(const uint16_t *h, int Cmid, int H, int W, int A, const uint16_t *w_packed, const float *bias, float *raw,
                         float *cls_prob, void *stream);
/* the same with a workspace, which lets launches with fewer tiles than the chip has room for (the 38x63 maps) split K
 * across workgroups (deterministic: partial tiles are summed in split order by the last arriver).  Same contract as the
 * fp32 conv workspace: its first 64 KB are tile counters -- zero them once (frcnn_conv_bf16_workspace_init), every
 * launch leaves them zeroed; convolutions only, one workspace per stream.  workspace NULL = frcnn_conv_bf16. */
size_t frcnn_conv_bf16_workspace_bytes(int Cin, int Cout, int H, int W);
int frcnn_conv_bf16_workspace_init(void *workspace, size_t workspace_bytes, void *stream);
int frcnn_conv_bf16_ws(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int Cin, int Cout, int H,
                       int W, int ksize, int relu, int out_mode, void *workspace, size_t workspace_bytes, void *stream);
/* which kernel family frcnn_conv_bf16[_ws] launches for this shape under the current environment (A/B hooks included): 0 = conv_dma_bf16_kernel
 * (or, for ksize 1 / FRCNN_BF16_DMA=0, the register-staged kernel), 901 / 902 / 903 / 909 / 910 / 911 = strip form A / B / C / D / D with direct stores / E of csrc/conv_bf16_strip.h (one wave
 * per SIMD, software-pipelined ring; D and C are default picks: DESIGN 3.8b).  Forms A, B, D give bit-identical results to 0; form C splits the K loop
 * four ways over the waves of a workgroup and sums the partial accumulators in K-way order (deterministic, fp32 rounding differs).  No launch. */
int frcnn_conv_bf16_plan(int Cin, int Cout, int H, int W, int ksize, int out_mode);
int frcnn_maxpool2x2_bf16(const uint16_t *x, uint16_t *y, int C, int H, int W, void *stream);
/* The first two layers of the bf16 chain as ONE launch (ABI v21): conv1_1 (Cin <= 3 -> 64, 3x3, pad 1) + ReLU + conv1_2 (64 -> 64) + ReLU +
 * F.max_pooling_2d(2, stride 2) -- /root/reference/models/vgg16.py:38-44 (conv1_1, conv1_2, pool1 of VGG16Prev / VGG16).  The 64-channel map
 * between the two convolutions lives in LDS only.  x (Cin, H, W) fp32 NCHW; w1 (64, Cin, 3, 3) fp32 as Chainer stores it, b1 (64);
 * w2_packed = frcnn_bf16_pack_conv_w of the (64, 64, 3, 3) weights, b2 (64); y [4][ceil(H/2)][ceil(W/2)][16] bf16 (channel-blocked).
 * Bit-identical to frcnn_conv1_bf16 followed by frcnn_conv_bf16 with out_mode 2 (same operand rounding, same accumulation order). */
int frcnn_conv1_pair_bf16(const float *x, const float *w1, const float *b1, const uint16_t *w2_packed, const float *b2,
                          uint16_t *y, int Cin, int H, int W, void *stream);
/* bf16 fully connected layer (config-3 head): y(M,N) = act(x(M,K) @ W(N,K)^T + b); x, W raw bf16 bits (frcnn_f32_to_bf16
 * converts fp32 arrays: weights once at load, activations per call), fp32 accumulation and bias; y fp32, or bf16 when
 * out_bf16 (feeding the next bf16 layer).  K % 8 == 0. */
int frcnn_f32_to_bf16(const float *x, size_t n, uint16_t *y, void *stream);
size_t frcnn_linear_bf16_workspace_bytes(int M, int N, int K);
int frcnn_linear_bf16(const uint16_t *x, const uint16_t *w, const float *bias, void *y, int M, int N, int K, int relu,
                      int out_bf16, void *workspace, size_t workspace_bytes, void *stream);
/* The same layer as a WEIGHT STREAM (ABI v23; csrc/linear_bf16.hip) -- L.Linear of /root/reference/models/faster_rcnn.py:33-36,127-134 on the bf16 line:
 * the (N, K) bf16 weight matrix is re-tiled ONCE at load time (frcnn_linear_bf16_tile_w) into 8 KB tiles [ceil(N/128)][K/32] that are the kernel's
 * swizzled LDS image of a 128-row x 32-k weight panel (rows past N zero), so the GEMM reads each weight byte once, as contiguous 8 KB runs, through a
 * five-stage LDS-DMA ring; all M <= 320 rows of x sit in one workgroup.  K % 32 == 0 (FRCNN_ERR_INVALID otherwise); tensors behind a 32-bit buffer
 * descriptor: M*K*2 and the tiled bytes below 2 GiB (FRCNN_ERR_UNSUPPORTED beyond).  Same products, fp32 accumulation, bias / ReLU / output forms as
 * frcnn_linear_bf16; split-K partial sums are added in split order (deterministic). */
size_t frcnn_linear_bf16_tiled_bytes(int N, int K);
int frcnn_linear_bf16_tile_w(const uint16_t *w, int N, int K, uint16_t *w_tiled, void *stream);
size_t frcnn_linear_bf16_tiled_workspace_bytes(int M, int N, int K);
int frcnn_linear_bf16_tiled(const uint16_t *x, const uint16_t *w_tiled, const float *bias, void *y, int M, int N, int K, int relu,
                            int out_bf16, void *workspace, size_t workspace_bytes, void *stream);
/* ---- the fp16 instantiation of the 16-bit chain (ABI v24; csrc/conv_f16.hip, conv_f16_pair.hip, linear_f16.hip) ---------------------
 * north_star: "MFMA-tiled ... conv stack (fp16/bf16 accumulate fp32)".  Every entry below is the twin of the *_bf16* entry of the same name: the SAME kernel
 * source compiled with fp16 pack / widen / v_mfma_f32_32x32x16_f16 (csrc/frcnn_intrin.h), same signature, layouts ([C/16][H][W][16] channel-blocked maps,
 * [CinP/16][tap][CoutP][16] weights, 8 KB FC tiles), workspaces, tuning keys and error codes; uint16_t arrays hold raw IEEE binary16 bits.  fp16 carries
 * 10 mantissa bits (bf16: 7) and 5 exponent bits: |v| > 65504 rounds to Inf, |v| < 6e-8 to 0 -- VGG-16's activations on mean-subtracted 8-bit images stay
 * inside that range; a caller whose maps do not must use the bf16 entries.  Replaces what the bf16 twins replace: models/vgg16.py:38-82,
 * region_proposal_network.py:53,117-120, faster_rcnn.py:33-36,125-134.  The two RoI-pooling twins (csrc/roi_f16.hip) pool into / from fp16: a maximum of fp16
 * values is an fp16 value, so pooling from the blocked fp16 map rounds nothing. */
int frcnn_roi_pool_fwd_chw_f16(const float *x, int C, int H, int W, const float *rois, int R, int roi_cols, int outh,
                                int outw, float spatial_scale, uint16_t *y, void *stream);
int frcnn_roi_pool_fwd_blk_f16(const uint16_t *x_blk, int C, int H, int W, const float *rois, int R, int roi_cols, int outh,
                                int outw, float spatial_scale, void *y, int out_f16, void *stream);
int frcnn_f16_to_nchw_f32(const uint16_t *x, int C, int H, int W, float *y, void *stream);
int frcnn_f16_padded_channels(int c);
int frcnn_f16_pack_conv_w(const float *w, int Cout, int Cin, int ksize, uint16_t *w_packed, void *stream);
int frcnn_f16_from_nchw_f32(const float *x, int C, int H, int W, uint16_t *y, void *stream);
size_t frcnn_conv_f16_workspace_bytes(int Cin, int Cout, int H, int W);
int frcnn_conv_f16_workspace_init(void *workspace, size_t workspace_bytes, void *stream);
int frcnn_conv_f16_ws(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int Cin, int Cout, int H,
                       int W, int ksize, int relu, int out_mode, void *workspace, size_t workspace_bytes, void *stream);
int frcnn_rpn_heads_f16(const uint16_t *h, int Cmid, int H, int W, int A, const uint16_t *w_packed, const float *bias, float *raw,
                         float *cls_prob, void *stream);
int frcnn_conv_f16(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int Cin, int Cout, int H,
                    int W, int ksize, int relu, int out_mode, void *stream);
int frcnn_conv_f16_plan(int Cin, int Cout, int H, int W, int ksize, int out_mode);
int frcnn_maxpool2x2_f16(const uint16_t *x, uint16_t *y, int C, int H, int W, void *stream);
int frcnn_f32_to_f16(const float *x, size_t n, uint16_t *y, void *stream);
size_t frcnn_linear_f16_workspace_bytes(int M, int N, int K);
int frcnn_linear_f16(const uint16_t *x, const uint16_t *w, const float *bias, void *y, int M, int N, int K, int relu,
                      int out_f16, void *workspace, size_t workspace_bytes, void *stream);
int frcnn_conv1_pair_f16(const float *x, const float *w1, const float *b1, const uint16_t *w2_packed, const float *b2,
                          uint16_t *y, int Cin, int H, int W, void *stream);
size_t frcnn_linear_f16_tiled_bytes(int N, int K);
int frcnn_linear_f16_tile_w(const uint16_t *w, int N, int K, uint16_t *w_tiled, void *stream);
size_t frcnn_linear_f16_tiled_workspace_bytes(int M, int N, int K);
int frcnn_linear_f16_tiled(const uint16_t *x, const uint16_t *w_tiled, const float *bias, void *y, int M, int N, int K, int relu,
                            int out_f16, void *workspace, size_t workspace_bytes, void *stream);
/* softmax over the channel axis of a (n_ch, H*W) fp32 map: the reference's F.softmax(rpn_cls_score) (region_proposal_network.py:119) */
int frcnn_softmax_channels_f32(const float *score, int n_ch, int HW, float *prob, void *stream);

/* ---- ResNet trunk pieces (models/resnet.py -> chainer ResNetLayers) ------------------
 * frcnn_conv_f32_ex with act = 3 is the bottleneck tail: y = relu(conv + bias + residual), residual passed as `mask`.
 * frcnn_im2col7x7s2_f32: the 7x7 / stride 2 / pad 3 stem as an explicit im2col (Kp >= Cin*49 rows, zero padded) so that
 *   it runs as a 1x1 convolution on the MFMA kernel; frcnn_maxpool3x3s2_f32: F.max_pooling_2d(3, stride=2), cover_all;
 * frcnn_subsample2_f32: the input view of a stride-2 1x1 convolution.  BatchNormalization (test mode) is folded into
 * the convolution weights and bias on the host at load time. */
int frcnn_im2col7x7s2_f32(const float *x, int Cin, int H, int W, int Kp, float *cols, void *stream);
int frcnn_maxpool3x3s2_f32(const float *x, float *y, int C, int H, int W, void *stream);
int frcnn_subsample2_f32(const float *x, float *y, int C, int H, int W, void *stream);

/* ---- RPN training step --------------------------------------------------------
 * frcnn_bbox_overlaps_f64: bbox_overlaps(boxes (N,4) f64, query_boxes (K,4) f64) -> (N,K) f64   (models/bbox.pyx:16-56)
 *
 * frcnn_anchor_target: AnchorTargetLayer.__call__ up to (not including) the random subsample
 *   (models/anchor_target_layer.py:66-145,183-198; keep_inside bbox_transform.py:112-130; bbox_transform :18-38).
 *   gt_boxes (G,5) f32 [x1,y1,x2,y2,cls]; out (capacity A*H*W each): inds_inside int32 ascending, n_inside (1),
 *   labels int32 in {-1,0,1} BEFORE the fg/bg subsample (:147-167 draws from NumPy's global RNG: host side),
 *   targets (n,4) f32, argmax_overlaps (n) int32.
 *
 * frcnn_rpn_loss: _calc_rpn_loss_cls + _calc_rpn_loss_bbox (models/region_proposal_network.py:160-204) and their
 *   gradients.  labels are the SUBSAMPLED labels of the n_inside inside anchors.  losses (3) f32 =
 *   [rpn_loss_cls, rpn_loss_bbox, rpn_cls_accuracy]; d_cls_score (2A,H,W), d_bbox_pred (4A,H,W) = gradients of
 *   rpn_loss_cls + loss_lambda * rpn_loss_bbox (both NULL: losses only).
 *
 * Backward pieces: frcnn_maxpool2x2_bwd_f32 (x = the pool's input), frcnn_bias_grad_f32 (db[c] = sum_p dy[c][p]),
 *   frcnn_pack_conv_dgrad_w (forward-packed (Cin*k*k,Cout) -> packed weights (Cout*k*k,Cin) of the input-gradient
 *   convolution, taps rotated 180 degrees; run it through frcnn_conv_f32_ex), frcnn_conv_wgrad_f32 (weight gradient in
 *   the forward-packed layout, MFMA, deterministic split over the pixels).
 * frcnn_sgd_momentum_wd: WeightDecay hook + MomentumSGD (train_rpn.py:165-167): g += wd*w; v = m*v - lr*g; w += v. */
int frcnn_bbox_overlaps_f64(const double *boxes, int N, const double *query_boxes, int K, double *overlaps,
                            void *stream);
size_t frcnn_anchor_target_workspace_bytes(int A, int H, int W, int G);
int frcnn_anchor_target(const double *anchors_host, int A, int H, int W, int feat_stride, int im_h, int im_w,
                        const float *gt_boxes, int G, int32_t *inds_inside, int32_t *n_inside, int32_t *labels,
                        float *targets, int32_t *argmax_overlaps, void *workspace, size_t workspace_bytes,
                        void *stream);
int frcnn_rpn_loss(const float *rpn_cls_score, const float *rpn_bbox_pred, const int32_t *labels,
                   const float *targets, const int32_t *inds_inside, int n_inside, int A, int H, int W, float delta,
                   float loss_lambda, float *losses, float *d_cls_score, float *d_bbox_pred, void *stream);
/* Stage-2 (rcnn_train) pieces, models/faster_rcnn.py:136-173: frcnn_rcnn_loss = softmax-CE (mean over the R
 * sampled RoIs) + Huber(delta) summed per RoI / R, and the gradients of their sum; losses (3) = [loss_cls, loss_bbox,
 * cls_accuracy]; labels int32 (R), targets (R,4*ncls).  frcnn_mul_f32: F.dropout forward/backward with the 0 | 1/(1-ratio)
 * mask; frcnn_relu_bwd_f32: g = (out > 0) ? g : 0 in place; frcnn_gather_rows_f32 / frcnn_scatter_rows_f32: x[keep_inds] and
 * its adjoint (dst zero-filled, dst_rows rows).  The head's L.Linear gradients are frcnn_linear_f32 calls on transposed
 * operands (frcnn_transpose_f32); RoI pooling backward is frcnn_roi_pool_bwd. */
int frcnn_rcnn_loss(const float *cls_score, const float *bbox_pred, const int32_t *labels, const float *targets, int R,
                    int ncls, float delta, float *losses, float *d_cls_score, float *d_bbox_pred, void *stream);
int frcnn_mul_f32(const float *a, const float *b, size_t n, float *y, void *stream);
int frcnn_add_f32(const float *a, const float *b, size_t n, float *y, void *stream);      /* y = a + b (y may alias a or b) */
/* F.dropout (models/faster_rcnn.py:128,131, train = True) with the mask drawn on the device: y[i] = x[i] * mask[i], mask[i] = 1/(1-ratio) iff
 * u(seed, i) >= ratio else 0, u = 24 bits of a splitmix64 hash of (seed, i) -- stateless and reproducible per seed.  The trainer's default keeps
 * the reference CPU path's numpy.random stream (host masks + frcnn_mul_f32); this entry is the throughput form (ABI v20). */
int frcnn_dropout_f32(const float *x, size_t n, float ratio, unsigned long long seed, float *mask, float *y, void *stream);
int frcnn_relu_bwd_f32(float *g, const float *out, size_t n, void *stream);
/* frcnn_gather_rows_f32 MOVES 4-byte words (plain loads and stores, no arithmetic, no canonicalisation): any 32-bit payload survives bit for bit -- the stage-2 trainer
 * carries int32 arg-max rows and packed keep / label blobs through it (an arg-max of -1 is a NaN bit pattern as a float).  This is part of the contract. */
int frcnn_gather_rows_f32(const float *src, const int32_t *idx, int n, int cols, float *dst, void *stream);
int frcnn_scatter_rows_f32(const float *src, const int32_t *idx, int n, int cols, float *dst, int dst_rows, void *stream);
int frcnn_maxpool2x2_bwd_f32(const float *x, const float *dy, float *dx, int C, int H, int W, void *stream);
/* the input-gradient convolution of the layer ABOVE a fused pool (3x3, frcnn_pack_conv_dgrad_w weights, `bias` = Cout zeros) with that pool's backward
 * pass in its epilogue: x (Cin,H,W) = dL/d(output of the layer above), argmax = the bytes frcnn_conv_f32_ex(act = 5) wrote for the layer below
 * (Cout,H,W: bits 0-1 the window cell, bit 2 "pooled value > 0" = the ReLU mask), y (Cout,H2,W2) = dL/d(pre-pool map): every value lands in its
 * window's arg-max cell, the other cells are written 0.  H2 = 2H or 2H-1, W2 likewise.  Workspace: frcnn_conv3x3_workspace_bytes. */
int frcnn_conv_dgrad_unpool_f32(const float *x, const float *w_packed, const float *bias, const unsigned char *argmax, float *y, int Cin, int Cout,
                                int H, int W, int H2, int W2, void *workspace, size_t workspace_bytes, void *stream);
/* the same from the arg-max bytes of frcnn_conv_f32_ex(act = 5) instead of the pool's input: dx (C,H,W) = dy routed to cell idx of each window */
int frcnn_maxpool2x2_bwd_idx_f32(const unsigned char *idx, const float *dy, float *dx, int C, int H, int W, void *stream);
size_t frcnn_bias_grad_workspace_bytes(int C, int HW);
int frcnn_bias_grad_f32(const float *dy, int C, int HW, float *db, void *workspace, size_t workspace_bytes, void *stream);
int frcnn_pack_conv_dgrad_w(const float *w_packed, int Cin, int Cout, int ksize, float *w_dgrad, void *stream);
/* frcnn_pack_conv_dgrad_w for up to 16 layers in one launch (the trainers re-pack every layer once per step) */
typedef struct {
    const float *w_packed;      /* (Cin*k*k, Cout) forward-packed */
    float *w_dgrad;             /* (Cout*k*k, Cin) */
    int Cin, Cout, ksize;
} frcnn_dgrad_pack_desc;
int frcnn_pack_conv_dgrad_w_many(const frcnn_dgrad_pack_desc *layers, int n, void *stream);
size_t frcnn_conv_wgrad_workspace_bytes(int Cin, int Cout, int H, int W, int ksize);
int frcnn_conv_wgrad_f32(const float *x, const float *dy, float *dw_packed, int Cin, int Cout, int H, int W, int ksize,
                         void *workspace, size_t workspace_bytes, void *stream);
int frcnn_sgd_momentum_wd(float *w, const float *grad, float *velocity, size_t n, float lr, float momentum,
                          float weight_decay, void *stream);
int frcnn_transpose_f32(const float *src, int rows, int cols, float *dst, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FRCNN_HIP_H */
