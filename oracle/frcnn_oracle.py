"""CPU oracle: a restatement of the reference's Faster R-CNN hot path (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module.  The product package (chainer-faster-rcnn_amd/) never does: it fails loudly when
the HIP library is missing instead of falling back to anything here.

Every function cites the reference file:line (relative to /root/reference) it follows.
Detection glue is restated in NumPy in the reference's own dtypes and operation order;
the two native routines (cpu_nms, bbox_overlaps) and RoI pooling are restated in C
(oracle/c/frcnn_oracle.c).  Pinning (tests/test_oracle_pinned.py):
  * glue + cpu_nms + bbox_overlaps: checked bit-for-bit against the reference's own code
    executed in the build container (oracle/ref_harness.py, oracle/_ref/*.so) and against
    tests/golden/*.npz generated from it by tests/make_golden.py -> PINNED.
  * conv / max-pool / softmax / linear / RoI pooling / losses / optimizer: these live in
    Chainer, an un-vendored and un-pinned dependency (README.md:13 "1.22.0+"); the
    reference's tests pin no value at that boundary -> PARITY UNPINNED.  They restate
    Chainer v1's published semantics with torch-CPU fp32 (conv2d / max_pool2d(ceil_mode) /
    linear) and explicit loops.  tests/test_oracle_second_witness.py holds a second,
    torch-free restatement of each (NumPy float64 loops from the same definitions) that has
    to agree with this one -- a cross-check of the restatement, not a pin to Chainer.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c(force=False):
    """gcc-compile oracle/c/frcnn_oracle.c -> oracle/_build/libfrcnn_oracle.so."""
    out_dir = os.path.join(HERE, "_build")
    so = os.path.join(out_dir, "libfrcnn_oracle.so")
    src = os.path.join(HERE, "c", "frcnn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
                               src, "-o", so, "-lm"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build_c())
        P, I64, D = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double
        L.oracle_cpu_nms.restype = I64
        L.oracle_cpu_nms.argtypes = [P, I64, P, D, P]
        L.oracle_bbox_overlaps.restype = None
        L.oracle_bbox_overlaps.argtypes = [P, I64, P, I64, P]
        L.oracle_roi_pool_fwd.restype = None
        L.oracle_roi_pool_fwd.argtypes = [P, I64, I64, I64, P, I64, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_float, P, P]
        L.oracle_roi_pool_bwd.restype = None
        L.oracle_roi_pool_bwd.argtypes = [P, P, P, I64, I64, I64, I64, ctypes.c_int, ctypes.c_int, I64, P]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------- anchors
def generate_anchors(base_size=15, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """models/generate_anchors.py:47-93.  (A,4) float64, ratio-major then scale.

    base anchor [0,0,15,15] (line 50) => w=h=16, ctr=7.5; per ratio: ws=rint(sqrt(256/r)),
    hs=rint(ws*r) (lines 77-84); per scale: ws*s, hs*s around the same centre (87-93);
    corner = ctr -/+ 0.5*(w-1) (lines 66-74).
    """
    w = h = float(base_size) + 1.0
    cx = cy = 0.5 * (w - 1)
    out = []
    for r in ratios:
        ws = np.rint(np.sqrt(w * h / r))
        hs = np.rint(ws * r)
        for s in scales:
            W_, H_ = ws * s, hs * s
            out.append([cx - 0.5 * (W_ - 1), cy - 0.5 * (H_ - 1), cx + 0.5 * (W_ - 1), cy + 0.5 * (H_ - 1)])
    return np.asarray(out, dtype=np.float64)


def generate_all_bbox(anchors, feat_h, feat_w, feat_stride=16):
    """models/proposal_layer.py:207-221.  (feat_h*feat_w*A, 4) float64, order (h, w, a), a fastest."""
    sx = np.arange(0, feat_w) * feat_stride
    sy = np.arange(0, feat_h) * feat_stride
    sx, sy = np.meshgrid(sx, sy)
    shifts = np.stack([sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()], axis=1)
    A = len(anchors)
    return (anchors.reshape(1, A, 4) + shifts.reshape(-1, 1, 4)).reshape(-1, 4)


# --------------------------------------------------------------------------- bbox transforms
def bbox_transform(ex_rois, gt_rois):
    """models/bbox_transform.py:18-38 (dtype follows the inputs: float64 in AnchorTargetLayer)."""
    ew = ex_rois[:, 2] - ex_rois[:, 0] + 1.0
    eh = ex_rois[:, 3] - ex_rois[:, 1] + 1.0
    ecx = ex_rois[:, 0] + 0.5 * ew
    ecy = ex_rois[:, 1] + 0.5 * eh
    gw = gt_rois[:, 2] - gt_rois[:, 0] + 1.0
    gh = gt_rois[:, 3] - gt_rois[:, 1] + 1.0
    gcx = gt_rois[:, 0] + 0.5 * gw
    gcy = gt_rois[:, 1] + 0.5 * gh
    return np.stack([(gcx - ecx) / ew, (gcy - ecy) / eh, np.log(gw / ew), np.log(gh / eh)], axis=1)


EXP = np.exp      # tests/test_coord_margins.py swaps this for an exp perturbed by +-k ulp (how far are the decisions from flipping?)


def bbox_transform_inv(boxes, trans):
    """models/bbox_transform.py:41-76.  Separate multiply and add (no FMA), np.exp in the input dtype,
    no clamp on dw/dh; generic over trans (N, 4*C)."""
    if boxes.shape[0] == 0:
        return np.zeros((0, trans.shape[1]), dtype=trans.dtype)
    widths = boxes[:, 2] - boxes[:, 0] + 1.0
    heights = boxes[:, 3] - boxes[:, 1] + 1.0
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    dx, dy, dw, dh = trans[:, 0::4], trans[:, 1::4], trans[:, 2::4], trans[:, 3::4]
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = EXP(dw) * widths[:, None]
    ph = EXP(dh) * heights[:, None]
    out = np.zeros(trans.shape, dtype=trans.dtype)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw
    out[:, 3::4] = pcy + 0.5 * ph
    return out


def clip_boxes(boxes, im_shape):
    """models/bbox_transform.py:79-99.  im_shape = (H, W); in place."""
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], int(im_shape[1] - 1)), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], int(im_shape[0] - 1)), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], int(im_shape[1] - 1)), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], int(im_shape[0] - 1)), 0)
    return boxes


def filter_boxes(boxes, min_size):
    """models/bbox_transform.py:102-109."""
    ws = boxes[:, 2] - boxes[:, 0] + 1
    hs = boxes[:, 3] - boxes[:, 1] + 1
    return np.where((ws >= min_size) & (hs >= min_size))[0]


def keep_inside(anchors, img_info):
    """models/bbox_transform.py:112-130.  img_info = (H, W)."""
    inds = np.where((anchors[:, 0] >= 0) & (anchors[:, 1] >= 0) &
                    (anchors[:, 2] < img_info[1]) & (anchors[:, 3] < img_info[0]))[0]
    return inds, anchors[inds]


# --------------------------------------------------------------------------- native routines
def descending_order_ties_by_index(scores):
    """The one platform-independent refinement of `scores.argsort()[::-1]`: descending score, every NaN first (where NumPy's sort puts them, reversed),
    ascending index among equal keys -- the order the HIP kernels document (include/frcnn_hip.h).  NumPy's introsort leaves the order of equal keys open."""
    sc = np.asarray(scores).ravel()
    isn = np.isnan(sc)
    return np.lexsort((np.arange(len(sc)), np.where(isn, 0.0, -sc.astype(np.float64)), ~isn))


def cpu_nms(dets, thresh, tie_rule=None):
    """models/cpu_nms.pyx:18-69 (C restatement).  Returns a Python list of indices into dets.
    tie_rule="ascending_index": visit equal scores in ascending index (see descending_order_ties_by_index) instead of in NumPy's implementation-defined order."""
    dets = np.ascontiguousarray(dets)
    if dets.dtype != np.float32 or dets.ndim != 2:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")     # Cython's own error class
    if not isinstance(thresh, float):
        raise TypeError("Argument 'thresh' has incorrect type (expected float)")
    n = dets.shape[0]
    if tie_rule == "ascending_index":
        order = np.ascontiguousarray(descending_order_ties_by_index(dets[:, 4]).astype(np.int64))
    else:
        order = np.ascontiguousarray(dets[:, 4].argsort()[::-1].astype(np.int64))   # cpu_nms.pyx:26
    keep = np.empty(max(n, 1), dtype=np.int64)
    k = _lib().oracle_cpu_nms(_p(dets), n, _p(order), float(thresh), _p(keep))
    return [int(v) for v in keep[:k]]


def cpu_nms_py(dets, thresh):
    """Pure-Python twin of cpu_nms for tiny cases (same arithmetic: float32 IoU, double compare)."""
    f = np.float32
    x1, y1, x2, y2, sc = [dets[:, i] for i in range(5)]
    areas = (x2 - x1 + f(1)) * (y2 - y1 + f(1))
    order = sc.argsort()[::-1]
    sup = np.zeros(len(dets), dtype=bool)
    keep = []
    for _i in range(len(dets)):
        i = order[_i]
        if sup[i]:
            continue
        keep.append(int(i))
        for _j in range(_i + 1, len(dets)):
            j = order[_j]
            if sup[j]:
                continue
            w = max(f(0), min(x2[i], x2[j]) - max(x1[i], x1[j]) + f(1))
            h = max(f(0), min(y2[i], y2[j]) - max(y1[i], y1[j]) + f(1))
            inter = f(w * h)
            ovr = f(inter / f(f(areas[i] + areas[j]) - inter))
            if float(ovr) >= thresh:
                sup[j] = True
    return keep


def bbox_overlaps(boxes, query_boxes):
    """models/bbox.pyx:16-56 (C restatement).  float64 in, float64 (N,K) out."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float64)
    out = np.empty((boxes.shape[0], query_boxes.shape[0]), dtype=np.float64)
    _lib().oracle_bbox_overlaps(_p(boxes), boxes.shape[0], _p(query_boxes), query_boxes.shape[0], _p(out))
    return out


# --------------------------------------------------------------------------- ProposalLayer
class ProposalParams(object):
    """Class constants of models/proposal_layer.py:51-56 and the train switch at :75-83."""
    RPN_NMS_THRESH = 0.7
    TRAIN = (12000, 2000)
    TEST = (6000, 300)
    RPN_MIN_SIZE = 16


def proposal_layer(rpn_cls_prob, rpn_bbox_pred, img_info, train=False, feat_stride=16,
                   anchor_ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32),
                   pre_nms_top_n=None, post_nms_top_n=None, nms_thresh=0.7, min_size=16,
                   return_debug=False, nms_fn=None, stage_times=None, tie_rule=None):
    """models/proposal_layer.py:102-198.  Inputs (1,2A,H,W), (1,4A,H,W) float32, img_info (1,2) int.

    tie_rule: None = the reference's own `argsort()[::-1]` (here and again inside cpu_nms: NumPy's introsort leaves the order of EQUAL scores
    implementation-defined); "ascending_index" = the one platform-independent refinement of it, which is the HIP path's documented rule -- descending
    score, every NaN first, ascending anchor index among equals, and NMS visiting the boxes in exactly that order.

    Returns (proposals (n,4) f32, fg_probs (n,1) f32) [+ a dict of intermediates].
    nms_fn: a cpu_nms(dets, thresh) callable to use instead of the C restatement (bench.py passes the reference's own
    compiled models/cpu_nms.pyx from oracle/_ref); stage_times: dict receiving seconds spent before / inside NMS.
    """
    import time as _time
    _t0 = _time.perf_counter()
    anchors = generate_anchors(ratios=anchor_ratios, scales=anchor_scales)      # :63-64
    A = len(anchors)
    pre, post = ProposalParams.TRAIN if train else ProposalParams.TEST            # :75-83
    pre = pre if pre_nms_top_n is None else pre_nms_top_n
    post = post if post_nms_top_n is None else post_nms_top_n
    prob = rpn_cls_prob[0]
    pred = rpn_bbox_pred[0]
    info = img_info[0]
    _, fh, fw = pred.shape
    all_bbox = generate_all_bbox(anchors, fh, fw, feat_stride).astype(np.float32)  # :200-205
    trans = pred.transpose(1, 2, 0).reshape(-1, 4)                                 # :138
    proposals = bbox_transform_inv(all_bbox, trans)                                # :141
    proposals = clip_boxes(proposals, info)                                        # :144
    keep0 = filter_boxes(proposals, min_size)                                      # :147
    proposals = proposals[keep0]
    fg = prob[A:].transpose(1, 2, 0).reshape(-1, 1)[keep0]                          # :152-154
    if tie_rule == "ascending_index":
        order = descending_order_ties_by_index(fg)
    else:
        order = fg.ravel().argsort()[::-1]                                         # :158-165
    if pre > 0:
        order = order[:pre]                                                        # :167-168
    proposals = proposals[order]
    fg = fg[order]
    dets = np.hstack((proposals, fg))
    _t1 = _time.perf_counter()
    if tie_rule == "ascending_index":                                              # cpu_nms.pyx:26 sorts again: the same rule there
        keep = cpu_nms(dets.astype(np.float32), float(nms_thresh), tie_rule="ascending_index")
    else:
        keep = (nms_fn or cpu_nms)(dets, float(nms_thresh))                        # :178
    if stage_times is not None:
        stage_times["decode_sort"] = stage_times.get("decode_sort", 0.0) + (_t1 - _t0)
        stage_times["nms"] = stage_times.get("nms", 0.0) + (_time.perf_counter() - _t1)
    if post > 0:
        keep = keep[:post]                                                         # :189-190
    out_p, out_s = proposals[keep], fg[keep]
    if return_debug:
        return out_p, out_s, dict(keep0=keep0, order=order, sorted_boxes=proposals, sorted_scores=fg,
                                  keep=np.asarray(keep, dtype=np.int64),
                                  src_index=keep0[order][keep] if len(keep) else np.zeros(0, np.int64))
    return out_p, out_s


# --------------------------------------------------------------------------- AnchorTargetLayer
def anchor_target_layer(feat_h, feat_w, gt_boxes, img_info, feat_stride=16, anchor_ratios=(0.5, 1, 2),
                        anchor_scales=(8, 16, 32), rng=np.random):
    """models/anchor_target_layer.py:66-198.  gt_boxes (1,G,5) f32, img_info (1,2) int.

    Returns (labels int32 (n_in,), targets f32 (n_in,4), inds_inside int64, n_all).  `rng` must offer
    NumPy's legacy `choice` so seeded runs reproduce the reference's np.random.choice draws (:153,164).
    """
    NEG, POS, FG_FRAC, BATCH = 0.3, 0.7, 0.5, 256                                   # :44-47
    gt = gt_boxes[0]
    info = img_info[0]
    anchors = generate_anchors(ratios=anchor_ratios, scales=anchor_scales)
    all_bbox = generate_all_bbox(anchors, feat_h, feat_w, feat_stride)             # float64, :109
    inds_inside, inside = keep_inside(all_bbox, info)                              # :110
    labels = np.ones((len(inds_inside),), dtype=np.int32) * -1                      # :129
    overlaps = bbox_overlaps(np.ascontiguousarray(inside, dtype=np.float64),
                             np.ascontiguousarray(gt[:, :4], dtype=np.float64))    # :183-185
    argmax = overlaps.argmax(axis=1)                                               # :189
    gt_argmax = overlaps.argmax(axis=0)                                            # :190
    max_ov = overlaps[np.arange(len(inds_inside)), argmax]                         # :192-193
    gt_max = overlaps[gt_argmax, np.arange(overlaps.shape[1])]                     # :194-195
    gt_argmax = np.where(overlaps == gt_max)[0]                                    # :196
    labels[max_ov < NEG] = 0                                                        # :135
    labels[gt_argmax] = 1                                                           # :138
    labels[max_ov >= POS] = 1                                                       # :141
    labels[max_ov < NEG] = 0                                                        # :144 (negatives clobber)
    num_fg = int(FG_FRAC * BATCH)
    fg_inds = np.where(labels == 1)[0]
    if len(fg_inds) > num_fg:                                                       # :149-155
        labels[rng.choice(fg_inds, size=int(len(fg_inds) - num_fg), replace=False)] = -1
    num_bg = BATCH - np.sum(labels == 1)
    bg_inds = np.where(labels == 0)[0]
    if len(bg_inds) > num_bg:                                                       # :158-167
        labels[rng.choice(bg_inds, size=int(len(bg_inds) - num_bg), replace=False)] = -1
    targets = bbox_transform(inside, gt[argmax]).astype(np.float32)                # :115-118
    return labels, targets, inds_inside, len(all_bbox)


# --------------------------------------------------------------------------- RoI pooling (chainer-ext)
def roi_pooling_2d(x, rois, outh=7, outw=7, spatial_scale=0.0625, return_argmax=False):
    """Call site models/faster_rcnn.py:125-126; semantics = Chainer v1 roi_pooling_2d (see the C file)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    N, C, H, W = x.shape
    R = rois.shape[0]
    y = np.empty((R, C, outh, outw), dtype=np.float32)
    am = np.empty((R, C, outh, outw), dtype=np.int32)
    _lib().oracle_roi_pool_fwd(_p(x), C, H, W, _p(rois), R, outh, outw, float(spatial_scale), _p(y), _p(am))
    return (y, am) if return_argmax else y


def roi_pooling_2d_py(x, rois, outh=7, outw=7, spatial_scale=0.0625):
    """Loop-for-loop NumPy twin of Chainer's forward_cpu (Python round() = half-to-even on the float32
    product, double strides, floor/ceil, clamp) -- used to cross-check the C restatement on small cases."""
    N, C, H, W = x.shape
    R = rois.shape[0]
    y = np.zeros((R, C, outh, outw), dtype=np.float32)
    am = -np.ones((R, C, outh, outw), dtype=np.int32)
    for r in range(R):
        idx, xmin, ymin, xmax, ymax = rois[r]
        xmin = int(round(np.float32(xmin * np.float32(spatial_scale))))
        xmax = int(round(np.float32(xmax * np.float32(spatial_scale))))
        ymin = int(round(np.float32(ymin * np.float32(spatial_scale))))
        ymax = int(round(np.float32(ymax * np.float32(spatial_scale))))
        rw = max(xmax - xmin + 1, 1)
        rh = max(ymax - ymin + 1, 1)
        sh, sw = 1. * rh / outh, 1. * rw / outw
        for ph in range(outh):
            hs = min(max(int(np.floor(ph * sh)) + ymin, 0), H)
            he = min(max(int(np.ceil((ph + 1) * sh)) + ymin, 0), H)
            if he <= hs:
                continue
            for pw in range(outw):
                ws = min(max(int(np.floor(pw * sw)) + xmin, 0), W)
                we = min(max(int(np.ceil((pw + 1) * sw)) + xmin, 0), W)
                if we <= ws:
                    continue
                d = x[int(idx), :, hs:he, ws:we].reshape(C, -1)
                y[r, :, ph, pw] = d.max(axis=1)
                a = d.argmax(axis=1)
                am[r, :, ph, pw] = (a // (we - ws) + hs) * W + (a % (we - ws) + ws)
    return y, am


def roi_pooling_2d_backward(dy, argmax, rois, x_shape):
    N, C, H, W = x_shape
    dy = np.ascontiguousarray(dy, dtype=np.float32)
    argmax = np.ascontiguousarray(argmax, dtype=np.int32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    dx = np.empty(x_shape, dtype=np.float32)
    R, _, outh, outw = dy.shape
    _lib().oracle_roi_pool_bwd(_p(dy), _p(argmax), _p(rois), R, C, H, W, outh, outw, N, _p(dx))
    return dx


# --------------------------------------------------------------------------- network (chainer-ext, torch-CPU)
VGG16_LAYERS = [  # models/vgg16.py:38-68 -- (name, cin, cout) / 'pool'
    ("conv1_1", 3, 64), ("conv1_2", 64, 64), "pool",
    ("conv2_1", 64, 128), ("conv2_2", 128, 128), "pool",
    ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), "pool",
    ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), "pool",
    ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512),
]


def init_params(seed=1, num_classes=21, n_anchors=9, roi_feat=512 * 7 * 7, dtype=np.float32):
    """Random-init weights of the reference architecture, keyed by Chainer link path.

    Trunk: He-normal (keeps activations alive through 13 ReLUs); RPN + head: Normal(0, 0.01) as
    models/faster_rcnn.py:27 and region_proposal_network.py:50; biases 0.
    """
    rs = np.random.RandomState(seed)
    p = {}
    for l in VGG16_LAYERS:
        if l == "pool":
            continue
        name, ci, co = l
        p["trunk/%s/W" % name] = (rs.randn(co, ci, 3, 3) * np.sqrt(2.0 / (ci * 9))).astype(dtype)
        p["trunk/%s/b" % name] = np.zeros(co, dtype)
    p["RPN/rpn_conv_3x3/W"] = (rs.randn(512, 512, 3, 3) * 0.01).astype(dtype)
    p["RPN/rpn_conv_3x3/b"] = np.zeros(512, dtype)
    p["RPN/rpn_cls_score/W"] = (rs.randn(2 * n_anchors, 512, 1, 1) * 0.01).astype(dtype)
    p["RPN/rpn_cls_score/b"] = np.zeros(2 * n_anchors, dtype)
    p["RPN/rpn_bbox_pred/W"] = (rs.randn(4 * n_anchors, 512, 1, 1) * 0.01).astype(dtype)
    p["RPN/rpn_bbox_pred/b"] = np.zeros(4 * n_anchors, dtype)
    p["fc6/W"] = (rs.randn(4096, roi_feat) * 0.01).astype(dtype)
    p["fc6/b"] = np.zeros(4096, dtype)
    p["fc7/W"] = (rs.randn(4096, 4096) * 0.01).astype(dtype)
    p["fc7/b"] = np.zeros(4096, dtype)
    p["cls_score/W"] = (rs.randn(num_classes, 4096) * 0.01).astype(dtype)
    p["cls_score/b"] = np.zeros(num_classes, dtype)
    p["bbox_pred/W"] = (rs.randn(4 * num_classes, 4096) * 0.01).astype(dtype)
    p["bbox_pred/b"] = np.zeros(4 * num_classes, dtype)
    return p


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


def conv2d(x, W, b, pad):
    """L.Convolution2D(ci, co, k, 1, pad): cross-correlation + bias, NCHW, fp32 [chainer-ext]."""
    import torch.nn.functional as F
    return F.conv2d(_t(x), _t(W), _t(b), stride=1, padding=pad).numpy()


def relu(x):
    return np.maximum(x, 0)


def max_pool_2x2(x):
    """F.MaxPooling2D(2, 2): cover_all=True => ceil-mode output size [chainer-ext] (vgg16.py:43)."""
    import torch.nn.functional as F
    return F.max_pool2d(_t(x), 2, 2, ceil_mode=True).numpy()


def softmax(x, axis=1):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return (e / e.sum(axis=axis, keepdims=True)).astype(x.dtype)


def vgg16_trunk(p, x, upto=None, collect=None):
    """models/vgg16.py:74-82 (VGG16Prev): conv1_1 ... relu5_3, no pool5.  `collect` (a dict) receives every layer's output:
    collect["conv3_2"] = relu3_2, collect["pool3"] = the map after the third F.MaxPooling2D."""
    n_pool = 0
    for l in VGG16_LAYERS:
        if l == "pool":
            x = max_pool_2x2(x)
            n_pool += 1
            if collect is not None:
                collect["pool%d" % n_pool] = x
        else:
            x = relu(conv2d(x, p["trunk/%s/W" % l[0]], p["trunk/%s/b" % l[0]], 1))
            if collect is not None:
                collect[l[0]] = x
            if upto == l[0]:
                break
    return x


def rpn_head(p, feat):
    """models/region_proposal_network.py:117-120.  NB: softmax over ALL 2A channels (axis 1)."""
    h = relu(conv2d(feat, p["RPN/rpn_conv_3x3/W"], p["RPN/rpn_conv_3x3/b"], 1))
    score = conv2d(h, p["RPN/rpn_cls_score/W"], p["RPN/rpn_cls_score/b"], 0)
    prob = softmax(score, axis=1)
    bbox = conv2d(h, p["RPN/rpn_bbox_pred/W"], p["RPN/rpn_bbox_pred/b"], 0)
    return h, score, prob, bbox


def linear(x, W, b):
    import torch.nn.functional as F
    return F.linear(_t(x), _t(W), _t(b)).numpy()


def rcnn_head(p, pool5, proposals, img_info):
    """models/faster_rcnn.py:127-134,175-178 (inference: dropout is identity)."""
    fc6 = relu(linear(pool5.reshape(len(pool5), -1), p["fc6/W"], p["fc6/b"]))
    fc7 = relu(linear(fc6, p["fc7/W"], p["fc7/b"]))
    cls_score = linear(fc7, p["cls_score/W"], p["cls_score/b"])
    bbox_pred = linear(fc7, p["bbox_pred/W"], p["bbox_pred/b"])
    pred_boxes = clip_boxes(bbox_transform_inv(proposals, bbox_pred), img_info[0])
    return softmax(cls_score, axis=1), pred_boxes, dict(fc6=fc6, fc7=fc7, cls_score=cls_score,
                                                        bbox_pred=bbox_pred)


def faster_rcnn_forward(p, x, img_info, return_debug=False):
    """models/faster_rcnn.py:111-134,175-178 inference path end to end."""
    layers = {} if return_debug == "layers" else None
    feat = vgg16_trunk(p, x, collect=layers)
    h, score, prob, bbox = rpn_head(p, feat)
    proposals, probs, pdbg = proposal_layer(prob, bbox, img_info, train=False, return_debug=True)
    brois = np.concatenate((np.zeros((len(proposals), 1), np.float32), proposals), axis=1)  # :123-124
    pool5 = roi_pooling_2d(feat, brois, 7, 7, 1.0 / 16)
    cls_prob, pred_boxes, dbg = rcnn_head(p, pool5, proposals, img_info)
    if return_debug:
        dbg.update(feat=feat, rpn_h=h, rpn_cls_score=score, rpn_cls_prob=prob, rpn_bbox_pred=bbox,
                   proposals=proposals, probs=probs, pool5=pool5, proposal_debug=pdbg)
        if layers is not None:
            dbg["layers"] = layers
        return cls_prob, pred_boxes, dbg
    return cls_prob, pred_boxes


# --------------------------------------------------------------------------- RPN losses (chainer-ext)
def rpn_loss_cls(rpn_cls_score, labels, inds_inside, n_all, feat_h, feat_w, n_anchors=9):
    """models/region_proposal_network.py:160-181: 2-way softmax CE over (1,2,A,H,W), ignore -1,
    mean over non-ignored [chainer-ext softmax_cross_entropy normalize=True]; accuracy ditto."""
    mapped = np.ones((n_all,), dtype=np.int32) * -1
    mapped[inds_inside] = labels
    mapped = mapped.reshape(1, feat_h, feat_w, n_anchors).transpose(0, 3, 1, 2)
    s = rpn_cls_score.reshape(1, 2, n_anchors, feat_h, feat_w).astype(np.float32)
    m = s.max(axis=1, keepdims=True)
    logp = s - m - np.log(np.exp(s - m).sum(axis=1, keepdims=True))
    valid = mapped != -1
    cnt = max(int(valid.sum()), 1)
    lab = np.where(valid, mapped, 0)
    picked = np.take_along_axis(logp, lab[:, None], axis=1)[:, 0]
    loss = -(picked * valid).sum() / cnt
    pred = s.argmax(axis=1)
    acc = float(((pred == mapped) & valid).sum()) / max(int(valid.sum()), 1)
    return np.float32(loss), np.float32(acc)


def rpn_loss_bbox(rpn_bbox_pred, targets, inds_inside, n_anchors=9, delta=3.0):
    """models/region_proposal_network.py:183-204: the (4,A,K)->(K,A,4) re-interpretation (channel =
    coord*A + a, inconsistent with ProposalLayer -- reproduced as is), Huber(delta) summed over all
    inside anchors, divided by K*A."""
    pred = rpn_bbox_pred.reshape(4, n_anchors, -1).transpose(2, 1, 0).reshape(-1, 4)
    n_bbox = pred.shape[0]
    d = pred[inds_inside].ravel().astype(np.float32) - targets.ravel().astype(np.float32)
    a = np.abs(d)
    l = np.where(a < delta, 0.5 * d * d, delta * (a - 0.5 * delta))
    return np.float32(l.sum(dtype=np.float32) / n_bbox)


def momentum_sgd_wd(W, g, v, lr=0.001, momentum=0.9, wd=0.0005):
    """train_rpn.py:165-167: WeightDecay hook then MomentumSGD [chainer-ext]: g+=wd*W; v=m*v-lr*g; W+=v."""
    g = g + np.float32(wd) * W
    v = np.float32(momentum) * v - np.float32(lr) * g
    return W + v, v


# --------------------------------------------------------------------------- backward pass (chainer-ext; torch-CPU autograd)
def conv2d_backward(x, W, b, dy, pad):
    """Gradients of L.Convolution2D(ksize, stride 1, pad) [chainer-ext]: (dx, dW, db) for upstream gradient dy."""
    import torch
    xt = _t(x).clone().requires_grad_(True)
    Wt = _t(W).clone().requires_grad_(True)
    bt = _t(b).clone().requires_grad_(True)
    y = torch.nn.functional.conv2d(xt, Wt, bt, stride=1, padding=pad)
    y.backward(_t(dy))
    return xt.grad.numpy(), Wt.grad.numpy(), bt.grad.numpy()


def max_pool_2x2_backward(x, dy):
    """F.max_pooling_2d(2, 2) (cover_all) backward: the gradient goes to the first maximum of each window."""
    import torch
    xt = _t(x).clone().requires_grad_(True)
    y = torch.nn.functional.max_pool2d(xt, 2, 2, ceil_mode=True)
    y.backward(_t(dy))
    return xt.grad.numpy()


def _torch_rpn_losses(score, bbox_pred, labels, targets, inds_inside, n_all, feat_h, feat_w, n_anchors, delta, lam):
    """models/region_proposal_network.py:160-204 with torch ops (so autograd yields the gradients)."""
    import torch
    mapped = np.ones((n_all,), dtype=np.int64) * -1
    mapped[inds_inside] = labels
    mapped = torch.from_numpy(mapped.reshape(1, feat_h, feat_w, n_anchors).transpose(0, 3, 1, 2).copy())
    s = score.reshape(1, 2, n_anchors, feat_h, feat_w)
    loss_cls = torch.nn.functional.cross_entropy(s, mapped, ignore_index=-1, reduction="sum") / max(int((mapped != -1).sum()), 1)
    pred = bbox_pred.reshape(4, n_anchors, -1).permute(2, 1, 0).reshape(-1, 4)
    d = pred[torch.from_numpy(np.asarray(inds_inside, dtype=np.int64))].reshape(-1) - torch.from_numpy(
        np.ascontiguousarray(targets, dtype=np.float32).ravel()).to(pred.dtype)
    a = d.abs()
    loss_bbox = torch.where(a < delta, 0.5 * d * d, delta * (a - 0.5 * delta)).sum() / pred.shape[0]
    return loss_cls, loss_bbox, loss_cls + lam * loss_bbox


def rpn_loss_grads(rpn_cls_score, rpn_bbox_pred, labels, targets, inds_inside, n_all, feat_h, feat_w, n_anchors=9, delta=3.0, lam=1.0):
    """-> (loss_cls, loss_bbox, d rpn_cls_score, d rpn_bbox_pred) of rpn_loss = cls + lam * bbox."""
    s = _t(rpn_cls_score).clone().requires_grad_(True)
    p = _t(rpn_bbox_pred).clone().requires_grad_(True)
    lc, lb, total = _torch_rpn_losses(s, p, labels, targets, inds_inside, n_all, feat_h, feat_w, n_anchors, delta, lam)
    total.backward()
    return np.float32(lc.item()), np.float32(lb.item()), s.grad.numpy(), p.grad.numpy()


def rpn_train_grads(p, x, labels, targets, inds_inside, n_all, delta=3.0, lam=1.0, layers=None, f64_wgrad=(), float64=False):
    """One RPN-mode forward/backward of FasterRCNN (faster_rcnn.py:110-116 -> region_proposal_network.py:116-145):
    -> (rpn_loss, {chainer link path: gradient}) for the trunk and RPN parameters, given the anchor targets.
    f64_wgrad: trunk layer names whose WEIGHT gradient is additionally accumulated in float64 from the same fp32 upstream gradient
    (returned under "<path>/W@f64"): at 600 x 1000 a conv1_x weight gradient is a 600 000-term fp32 sum, and two correct fp32
    implementations differ by more than the sum's own rounding -- the float64 value is the arbiter.
    float64=True: the WHOLE forward/backward in float64 from the same fp32 parameters, image and targets -- the arbiter of an
    end-to-end comparison between two fp32 implementations (tests/train_cases.py:check_vgg_step at 600 x 1000)."""
    import torch
    F = torch.nn.functional
    from_names = [k for k in p if k.startswith("trunk/") or k.startswith("RPN/")]
    cast = (lambda a: _t(a).double()) if float64 else _t
    tp = {k: cast(p[k]).clone().requires_grad_(True) for k in from_names}
    h = cast(x)
    layers = layers or ["conv1_1", "conv1_2", "pool", "conv2_1", "conv2_2", "pool", "conv3_1", "conv3_2", "conv3_3", "pool",
                        "conv4_1", "conv4_2", "conv4_3", "pool", "conv5_1", "conv5_2", "conv5_3"]
    taps = {}
    for l in layers:
        if l == "pool":
            h = F.max_pool2d(h, 2, 2, ceil_mode=True)
            continue
        pre = F.conv2d(h, tp["trunk/%s/W" % l], tp["trunk/%s/b" % l], padding=1)
        if l in f64_wgrad:
            pre.retain_grad()
            taps[l] = (h.detach(), pre)
        h = F.relu(pre)
    hh = F.relu(F.conv2d(h, tp["RPN/rpn_conv_3x3/W"], tp["RPN/rpn_conv_3x3/b"], padding=1))
    score = F.conv2d(hh, tp["RPN/rpn_cls_score/W"], tp["RPN/rpn_cls_score/b"])
    bbox = F.conv2d(hh, tp["RPN/rpn_bbox_pred/W"], tp["RPN/rpn_bbox_pred/b"])
    fh, fw = int(score.shape[2]), int(score.shape[3])
    A = int(score.shape[1]) // 2
    lc, lb, total = _torch_rpn_losses(score, bbox, labels, targets, inds_inside, n_all, fh, fw, A, delta, lam)
    total.backward()
    grads = {k: v.grad.numpy() for k, v in tp.items()}
    for l, (xin, pre) in taps.items():
        W = tp["trunk/%s/W" % l]
        grads["trunk/%s/W@f64" % l] = torch.nn.grad.conv2d_weight(xin.double(), tuple(W.shape), pre.grad.double(), padding=1).numpy()
    return np.float32(total.item()), grads


def rpn_train_grads_given_decisions(p, x, labels, targets, inds_inside, n_all, post_relu, pre_pool, rpn_mid, delta=3.0, lam=1.0, layers=None):
    """rpn_train_grads(float64=True) with the DISCRETE decisions of another forward pass imposed: `post_relu[name]` is that pass's
    (fp32) output of conv `name` after its ReLU -- the float64 pre-activation is multiplied by (post_relu > 0) instead of going through
    relu() -- and `pre_pool[i]` is its input of the i-th 2x2 max-pool, whose window arg-maxima (first maximum in scan order) select the
    cells of the float64 map; `rpn_mid` likewise for rpn_conv_3x3.  Every arithmetic step stays float64, so what comes out is "the
    exact gradient of the function the other pass actually evaluated": a device pass whose gradients match THIS to 1e-5 differs from
    the pure float64 pass only through the handful of ReLU signs / pool winners that fp32 rounding decides differently.
    -> (loss, grads, n_relu_flips per layer vs the float64 pass's own decisions)."""
    import torch
    F = torch.nn.functional
    from_names = [k for k in p if k.startswith("trunk/") or k.startswith("RPN/")]
    tp = {k: _t(p[k]).double().clone().requires_grad_(True) for k in from_names}
    h = _t(x).double()
    layers = layers or ["conv1_1", "conv1_2", "pool", "conv2_1", "conv2_2", "pool", "conv3_1", "conv3_2", "conv3_3", "pool",
                        "conv4_1", "conv4_2", "conv4_3", "pool", "conv5_1", "conv5_2", "conv5_3"]
    flips, ip = {}, 0
    for l in layers:
        if l == "pool":
            src = _t(pre_pool[ip]).reshape(1, *pre_pool[ip].shape[-3:])
            ip += 1
            _, idx = F.max_pool2d(src, 2, 2, ceil_mode=True, return_indices=True)
            own = F.max_pool2d(h.detach(), 2, 2, ceil_mode=True, return_indices=True)[1]
            flips["pool%d" % ip] = int((own != idx).sum())
            h = h.flatten(2).gather(2, idx.flatten(2)).reshape(idx.shape)
            continue
        pre = F.conv2d(h, tp["trunk/%s/W" % l], tp["trunk/%s/b" % l], padding=1)
        m = _t(post_relu[l]).reshape(pre.shape) > 0
        flips[l] = int((m != (pre.detach() > 0)).sum())
        h = pre * m
    pre = F.conv2d(h, tp["RPN/rpn_conv_3x3/W"], tp["RPN/rpn_conv_3x3/b"], padding=1)
    m = _t(rpn_mid).reshape(pre.shape) > 0
    flips["rpn_conv_3x3"] = int((m != (pre.detach() > 0)).sum())
    hh = pre * m
    score = F.conv2d(hh, tp["RPN/rpn_cls_score/W"], tp["RPN/rpn_cls_score/b"])
    bbox = F.conv2d(hh, tp["RPN/rpn_bbox_pred/W"], tp["RPN/rpn_bbox_pred/b"])
    fh, fw = int(score.shape[2]), int(score.shape[3])
    A = int(score.shape[1]) // 2
    lc, lb, total = _torch_rpn_losses(score, bbox, labels, targets, inds_inside, n_all, fh, fw, A, delta, lam)
    total.backward()
    return float(total.item()), {k: v.grad.numpy() for k, v in tp.items()}, flips


# --------------------------------------------------------------------------- ResNet trunk (chainer-ext)
def resnet_forward(p, x, blocks=(3, 4, 23, 3), prefix="trunk/", eps=2e-5):
    """models/resnet.py:43-45 -> chainer ResNetLayers(...)(x, ['res5'], test=True)['res5'] [chainer-ext], restated with
    torch-CPU fp32 and EXPLICIT (unfolded) test-mode BatchNormalization: conv1 7x7/2 pad 3 -> bn1 -> relu ->
    max_pooling_2d(3, stride=2) (cover_all) -> res2..res5; BottleNeckA: stride on conv1 and on the projection conv4;
    h = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + shortcut)."""
    import torch
    F = torch.nn.functional

    def bn(h, name):
        g, b, m, v = [_t(p[prefix + name + "/" + n]) for n in ("gamma", "beta", "avg_mean", "avg_var")]
        return F.batch_norm(h, m, v, g, b, training=False, eps=eps)

    def conv(h, name, stride=1, pad=0):
        return F.conv2d(h, _t(p[prefix + name + "/W"]), None, stride=stride, padding=pad)

    with torch.no_grad():
        h = F.relu(bn(conv(_t(x), "conv1", 2, 3), "bn1"))
        h = F.max_pool2d(h, 3, 2, ceil_mode=True)
        stages = [("res2", 1), ("res3", 2), ("res4", 2), ("res5", 2)]
        for (stage, stride), n in zip(stages, blocks):
            for i in range(n):
                b = "a" if i == 0 else "b%d" % i
                q = "%s/%s/" % (stage, b)
                s = stride if i == 0 else 1
                sc = bn(conv(h, q + "conv4", s), q + "bn4") if i == 0 else h
                t = F.relu(bn(conv(h, q + "conv1", s), q + "bn1"))
                t = F.relu(bn(conv(t, q + "conv2", 1, 1), q + "bn2"))
                h = F.relu(bn(conv(t, q + "conv3"), q + "bn3") + sc)
        return h.numpy()


# --------------------------------------------------------------------------- image preprocessing (cv2-ext)
def img_preprocessing(orig_img, pixel_means, max_size=1000, scale=600):
    """forward.py:33-45.  cv.resize(INTER_LINEAR) is OpenCV (absent here: parity unpinned); restated from its documented
    float path: dsize = round(size*f); source coordinate (d + 0.5)/f - 0.5, floor, clamp with zero weight at the edges,
    horizontal blend then vertical blend in float32."""
    img = orig_img.astype(np.float32, copy=True)
    img -= pixel_means
    h, w = img.shape[:2]
    im_scale = float(scale) / float(min(h, w))
    if np.round(im_scale * max(h, w)) > max_size:
        im_scale = float(max_size) / float(max(h, w))
    oh, ow = int(np.rint(h * im_scale)), int(np.rint(w * im_scale))

    def taps(n_out, n_in):
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * (1.0 / im_scale) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        a = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        a[lo] = 0; s[lo] = 0
        hi = s >= n_in - 1
        a[hi] = 0; s[hi] = n_in - 1
        return s, np.minimum(s + 1, n_in - 1), a
    sx, sx1, ax = taps(ow, w)
    sy, sy1, ay = taps(oh, h)
    rows = img[:, sx] * (np.float32(1) - ax)[None, :, None] + img[:, sx1] * ax[None, :, None]
    out = rows[sy] * (np.float32(1) - ay)[:, None, None] + rows[sy1] * ay[:, None, None]
    return out.transpose(2, 0, 1).astype(np.float32), im_scale


# --------------------------------------------------------------------------- ProposalTargetLayer + stage-2 losses
def proposal_target_layer(proposals, gt_boxes, num_classes=21, rng=np.random):
    """models/proposal_target_layer.py:84-150.  proposals (n,4) f32, gt_boxes (1,G,5) f32.
    Returns (use_gt_boxes (k,5) f32, ext_bbox_reg_targets (k,4*num_classes) f32, keep_inds (k,) int32), k <= 128.
    `rng.choice` is drawn exactly as the reference draws np.random.choice (:107-108, :121-122)."""
    FG_THRESH, BG_HI, BG_LO, ROIS, FG_FRAC = 0.5, 0.5, 0.1, 128, 0.25            # :46-50
    n_fg_rois = int(FG_FRAC * ROIS)
    assert len(proposals) > 0 and proposals.ndim == 2 and proposals.shape[1] == 4       # :62-64 (_check_data_type_forward; CHAINER_TYPE_CHECK on, the default)
    gt = gt_boxes[0]
    overlaps = bbox_overlaps(np.ascontiguousarray(proposals, dtype=np.float64),
                             np.ascontiguousarray(gt[:, :4], dtype=np.float64))   # anchor_target_layer.py:183-185
    argmax = overlaps.argmax(axis=1)
    max_ov = overlaps[np.arange(len(proposals)), argmax]
    cls_labels = gt[argmax, 4]                                                     # :92
    fg_inds = np.where(max_ov >= FG_THRESH)[0]                                     # :95
    n_fg = min(n_fg_rois, fg_inds.size)                                            # :99
    if fg_inds.size > 0:
        fg_inds = rng.choice(fg_inds, size=n_fg, replace=False)                    # :105-108
    bg_inds = np.where((max_ov < BG_HI) & (max_ov >= BG_LO))[0]                    # :111-112
    n_bg = min(ROIS - n_fg, bg_inds.size)                                          # :114-115
    if bg_inds.size > 0:
        bg_inds = rng.choice(bg_inds, size=n_bg, replace=False)                    # :119-122
    keep = np.concatenate([fg_inds, bg_inds]).astype(np.int32)                     # :126
    cls_labels = cls_labels[keep]
    cls_labels[n_fg:] = 0                                                          # :130 (a local copy: use_gt_boxes keeps the gt's label)
    props = proposals[keep]
    use_gt = gt[argmax[keep]]                                                      # :134
    targets = bbox_transform(props, use_gt)                                        # :135 (float32 in, float32 out)
    ext = np.zeros((len(keep), 4 * num_classes), dtype=np.float32)                 # :138-139
    for ind in np.where(use_gt[:, 4] > 0)[0]:                                      # :140-143
        pos = int(4 * use_gt[ind, -1])
        ext[ind, pos:pos + 4] = targets[ind]
    return use_gt, ext, keep


def _torch_rcnn_losses(cls_score, bbox_pred, labels, targets, delta):
    """models/faster_rcnn.py:152-164: softmax CE (mean over the sampled RoIs) + huber_loss(delta) summed per RoI, divided by
    the number of RoIs (F.huber_loss returns one value per row [chainer-ext]; loss_bbox.size is the row count)."""
    import torch
    loss_cls = torch.nn.functional.cross_entropy(cls_score, torch.from_numpy(np.asarray(labels, dtype=np.int64)))
    d = bbox_pred - torch.from_numpy(np.ascontiguousarray(targets, dtype=np.float32)).to(bbox_pred.dtype)
    a = d.abs()
    loss_bbox = torch.where(a < delta, 0.5 * d * d, delta * (a - 0.5 * delta)).sum() / bbox_pred.shape[0]
    return loss_cls, loss_bbox


def rcnn_loss_grads(cls_score, bbox_pred, labels, targets, delta=1.0):
    """-> (loss_cls, loss_bbox, accuracy, d cls_score, d bbox_pred) of loss_rcnn = loss_cls + loss_bbox (faster_rcnn.py:164)."""
    s = _t(cls_score).clone().requires_grad_(True)
    b = _t(bbox_pred).clone().requires_grad_(True)
    lc, lb = _torch_rcnn_losses(s, b, labels, targets, delta)
    (lc + lb).backward()
    acc = float((np.asarray(cls_score).argmax(axis=1) == np.asarray(labels)).mean())
    return np.float32(lc.item()), np.float32(lb.item()), np.float32(acc), s.grad.numpy(), b.grad.numpy()


def rcnn_train_grads(p, x, rois, keep_inds, labels, targets, mask6, mask7, layers=None, spatial_scale=1.0 / 16, delta=1.0, float64=False, head_relu=None,
                     trunk_decisions=None, roi_argmax=None):
    """One rcnn_train-mode forward/backward of FasterRCNN (faster_rcnn.py:110-173) given the proposals (rois (R,4)), the
    ProposalTargetLayer output (keep_inds, labels = use_gt_boxes[:, -1], class-wise targets) and the two dropout masks
    (values 0 or 1/(1-ratio) [chainer-ext F.dropout]).  RoI pooling is a custom autograd function over the C oracle.
    -> (loss_rcnn, {link path: gradient}) for the trunk and the four head layers (the RPN receives no gradient).
    float64: the same fp32 parameters / image / sample evaluated in float64 -- the arbiter of the full-size test (two fp32 passes through
    17 layers take a handful of different ReLU / max-pool / arg-max decisions; how far that moves a gradient is measured against this pass).
    RoI pooling then picks its arg-max cells on the float64 map itself (same scan rule) and gathers / scatters in float64.
    head_relu = (r6, r7) boolean (R, hidden) arrays: IMPOSE these ReLU decisions on fc6 / fc7 instead of taking the pass's own (the device's, read off
    its activations: a pre-activation within summation noise of zero may take the other branch there -- one flipped unit moves fc6's bias gradient by
    that unit's whole upstream gradient); the number of imposed decisions that differ from the pass's own is returned as a third value.
    trunk_decisions / roi_argmax (float64 + head_relu only): EVERY discrete decision of another forward pass imposed, as rpn_train_grads_given_decisions does for
    the RPN step -- trunk_decisions[i] for layers[i]: a boolean (C, H, W) ReLU mask for a convolution that no pool follows, None for one a pool follows, and for a
    pool a pair (winner, mask): the flat index h * W + w (int64, (C, OH, OW)) of each window's winning cell in the convolution's pre-pool map and "the pooled value is
    positive" -- ReLU and first-maximum pooling as ONE decision per window; roi_argmax: (R, C, 7, 7) int, the arg-max cell of every RoI bin (-1 = empty bin).
    What comes out is the exact float64 gradient of the function that pass evaluated; the third value is then a dict of flip counts per decision site."""
    import torch
    F = torch.nn.functional
    dt = torch.float64 if float64 else torch.float32

    class RoiPool(torch.autograd.Function):
        @staticmethod
        def forward(ctx, feat, brois):
            if not float64:
                y, am = roi_pooling_2d(feat.detach().numpy(), brois, 7, 7, spatial_scale, return_argmax=True)
                ctx.am, ctx.shape, ctx.brois = am, tuple(feat.shape), brois
                return torch.from_numpy(y)
            # float64: bin edges from the fp32 restatement (they do not depend on the map), maxima / first arg-max per bin on the float64 map
            f = feat.detach().numpy()[0]
            C, H, W = f.shape
            _, am32 = roi_pooling_2d(f.astype(np.float32)[None], brois, 7, 7, spatial_scale, return_argmax=True)
            R = len(brois)
            y = np.zeros((R, C, 7, 7), np.float64)
            am = -np.ones((R, C, 7, 7), np.int64)
            sc = np.float32(spatial_scale)
            for r in range(R):
                xs, ys = int(np.rint(np.float32(brois[r, 1]) * sc)), int(np.rint(np.float32(brois[r, 2]) * sc))
                xe, ye = int(np.rint(np.float32(brois[r, 3]) * sc)), int(np.rint(np.float32(brois[r, 4]) * sc))
                rw, rh = max(xe - xs + 1, 1), max(ye - ys + 1, 1)
                sh, sw = rh / 7.0, rw / 7.0
                for ph in range(7):
                    hs, he = min(max(int(np.floor(ph * sh)) + ys, 0), H), min(max(int(np.ceil((ph + 1) * sh)) + ys, 0), H)
                    for pw in range(7):
                        ws, we = min(max(int(np.floor(pw * sw)) + xs, 0), W), min(max(int(np.ceil((pw + 1) * sw)) + xs, 0), W)
                        if he <= hs or we <= ws:
                            continue
                        d = f[:, hs:he, ws:we].reshape(C, -1)
                        a = d.argmax(axis=1)
                        y[r, :, ph, pw] = d[np.arange(C), a]
                        am[r, :, ph, pw] = (a // (we - ws) + hs) * W + (a % (we - ws) + ws)
            assert ((am32 < 0) == (am < 0)).all()
            RoiPool.last_am = am
            ctx.am, ctx.shape = am, tuple(feat.shape)
            return torch.from_numpy(y)

        @staticmethod
        def backward(ctx, gy):
            if not float64:
                return torch.from_numpy(roi_pooling_2d_backward(np.ascontiguousarray(gy.numpy()), ctx.am, ctx.brois, ctx.shape)), None
            _, C, H, W = ctx.shape
            g = np.zeros((C, H * W), np.float64)
            gyn, am = gy.numpy(), ctx.am
            cidx = np.broadcast_to(np.arange(C)[None, :, None, None], am.shape)
            ok = am >= 0
            np.add.at(g, (cidx[ok], am[ok]), gyn[ok])
            return torch.from_numpy(g.reshape(1, C, H, W)), None

    names = [k for k in p if k.startswith("trunk/") or k.split("/")[0] in ("fc6", "fc7", "cls_score", "bbox_pred")]
    tp = {k: _t(p[k]).to(dt).clone().requires_grad_(True) for k in names}
    h = _t(x).to(dt)
    layers = layers or ["conv1_1", "conv1_2", "pool", "conv2_1", "conv2_2", "pool", "conv3_1", "conv3_2", "conv3_3", "pool",
                        "conv4_1", "conv4_2", "conv4_3", "pool", "conv5_1", "conv5_2", "conv5_3"]
    site_flips = {}
    if trunk_decisions is None:
        for l in layers:
            h = F.max_pool2d(h, 2, 2, ceil_mode=True) if l == "pool" else F.relu(F.conv2d(h, tp["trunk/%s/W" % l], tp["trunk/%s/b" % l], padding=1))
    else:
        assert float64 and head_relu is not None and len(trunk_decisions) == len(layers)
        pre, npool = None, 0
        for l, dec in zip(layers, trunk_decisions):
            if l == "pool":
                npool += 1
                winner, mask = dec
                widx = torch.from_numpy(np.ascontiguousarray(winner, dtype=np.int64))[None]
                wmask = torch.from_numpy(np.ascontiguousarray(mask, dtype=bool))[None]
                own_val, own_idx = F.max_pool2d(F.relu(pre.detach()), 2, 2, ceil_mode=True, return_indices=True)
                own_mask = own_val > 0
                # a window whose value is zero either way routes no gradient: only windows that are live on one side at least can differ
                site_flips["pool%d" % npool] = int((((own_idx != widx) & (own_mask | wmask)) | (own_mask != wmask)).sum())
                h = pre.flatten(2).gather(2, widx.flatten(2)).reshape(widx.shape) * wmask
                pre = None
                continue
            pre = F.conv2d(h, tp["trunk/%s/W" % l], tp["trunk/%s/b" % l], padding=1)
            if dec is None:
                continue                                             # ReLU + pool decided together at the pool's entry
            m = torch.from_numpy(np.ascontiguousarray(dec, dtype=bool)).reshape(pre.shape)
            site_flips[l] = int((m != (pre.detach() > 0)).sum())
            h = pre * m
    brois = np.concatenate([np.zeros((len(rois), 1), np.float32), np.asarray(rois, np.float32)], axis=1)
    if roi_argmax is None:
        pool5 = RoiPool.apply(h, brois)
    else:
        assert float64
        am = torch.from_numpy(np.ascontiguousarray(roi_argmax, dtype=np.int64))           # (R, C, 7, 7)
        with torch.no_grad():
            own = RoiPool.apply(h.detach(), brois)                                          # (float64 branch: fills RoiPool.last_am)
        site_flips["roi_argmax"] = int((RoiPool.last_am != am.numpy()).sum())
        C_ = int(h.shape[1])
        cidx = torch.arange(C_)[None, :, None, None].expand_as(am)
        pool5 = h[0].flatten(1)[cidx, am.clamp(min=0)] * (am >= 0)
    flips = 0
    pre6 = F.linear(pool5.reshape(len(rois), -1), tp["fc6/W"], tp["fc6/b"])
    if head_relu is None:
        fc6 = F.relu(pre6) * _t(mask6).to(dt)
    else:
        r6 = torch.from_numpy(np.ascontiguousarray(head_relu[0], dtype=bool))
        flips += int(((pre6.detach() > 0) != r6).sum())
        fc6 = pre6 * r6.to(dt) * _t(mask6).to(dt)
    pre7 = F.linear(fc6, tp["fc7/W"], tp["fc7/b"])
    if head_relu is None:
        fc7 = F.relu(pre7) * _t(mask7).to(dt)
    else:
        r7 = torch.from_numpy(np.ascontiguousarray(head_relu[1], dtype=bool))
        flips += int(((pre7.detach() > 0) != r7).sum())
        fc7 = pre7 * r7.to(dt) * _t(mask7).to(dt)
    cls_score = F.linear(fc7, tp["cls_score/W"], tp["cls_score/b"])
    bbox_pred = F.linear(fc7, tp["bbox_pred/W"], tp["bbox_pred/b"])
    idx = torch.from_numpy(np.asarray(keep_inds, dtype=np.int64))
    lc, lb = _torch_rcnn_losses(cls_score[idx], bbox_pred[idx], labels, targets, delta)
    total = lc + lb
    total.backward()
    if trunk_decisions is not None or roi_argmax is not None:
        site_flips["fc6_fc7_relu"] = flips
        return float(total.item()), {k: v.grad.numpy() for k, v in tp.items()}, site_flips
    if head_relu is not None:
        return float(total.item()), {k: v.grad.numpy() for k, v in tp.items()}, flips
    if float64:
        return float(total.item()), {k: v.grad.numpy() for k, v in tp.items()}
    return np.float32(total.item()), {k: v.grad.numpy() for k, v in tp.items()}
