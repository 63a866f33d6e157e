"""Full-size parity report: the device forward against the CPU oracle on the SAME image (TEST INFRASTRUCTURE ONLY).

Used by tests/test_gpu_fullsize.py (asserts on the numbers) and by bench.py's `cpu_baseline` leg (emits them as the
`parity` object of the bench line) -- the oracle forward that leg times is the one compared here, so the check costs
nothing extra.  Follows /root/reference/forward.py:92-94 -> models/faster_rcnn.py:111-178.

Tolerances (north_star): proposal indices / NMS survivors / RoI maxima bit-exact; fp32 features within 1e-3 relative
(max |device - oracle| / max |oracle| per tensor); bf16 stack within 3e-2 of the feature scale.
"""
import numpy as np

from . import frcnn_oracle as O


def rel_err(got, want):
    want = np.asarray(want, dtype=np.float32)
    scale = float(np.abs(want).max())
    return float(np.abs(np.asarray(got, dtype=np.float32) - want).max() / max(scale, 1e-30))


def nms_margins(sorted_boxes, sorted_scores, thresh=0.7, chunk=500, rows=None):
    """min |IoU - thresh| over the pairs greedy NMS can look at (cpu_nms.pyx:58-66 arithmetic, fp32) and the smallest gap between
    neighbouring sorted scores: how far the inputs are from a decision flipping under a 1-ulp perturbation.
    rows: restrict the first box of a pair to these indices (the KEPT boxes: only their rows ever suppress anything)."""
    b = np.asarray(sorted_boxes, dtype=np.float32)
    n = len(b)
    area = (b[:, 2] - b[:, 0] + np.float32(1)) * (b[:, 3] - b[:, 1] + np.float32(1))
    best = np.inf
    if rows is not None:
        rows = np.asarray(rows, dtype=np.int64)
        for i0 in range(0, len(rows), chunk):
            r = rows[i0:i0 + chunk]
            a = b[r]
            xx1 = np.maximum(a[:, None, 0], b[None, :, 0]); yy1 = np.maximum(a[:, None, 1], b[None, :, 1])
            xx2 = np.minimum(a[:, None, 2], b[None, :, 2]); yy2 = np.minimum(a[:, None, 3], b[None, :, 3])
            w = np.maximum(np.float32(0), xx2 - xx1 + np.float32(1)); h = np.maximum(np.float32(0), yy2 - yy1 + np.float32(1))
            inter = w * h
            d = np.abs((inter / (area[r, None] + area[None, :] - inter)).astype(np.float64) - thresh)
            d[np.arange(n)[None, :] <= r[:, None]] = np.inf         # a kept box only judges the boxes after it
            best = min(best, float(d.min())) if d.size else best
        s = np.asarray(sorted_scores, dtype=np.float32).ravel()
        return best, (float(np.min(-np.diff(s.astype(np.float64)))) if len(s) > 1 else float("inf"))
    for i0 in range(0, n, chunk):
        a = b[i0:i0 + chunk]
        xx1 = np.maximum(a[:, None, 0], b[None, :, 0]); yy1 = np.maximum(a[:, None, 1], b[None, :, 1])
        xx2 = np.minimum(a[:, None, 2], b[None, :, 2]); yy2 = np.minimum(a[:, None, 3], b[None, :, 3])
        w = np.maximum(np.float32(0), xx2 - xx1 + np.float32(1)); h = np.maximum(np.float32(0), yy2 - yy1 + np.float32(1))
        inter = w * h
        ovr = inter / (area[i0:i0 + chunk, None] + area[None, :] - inter)
        d = np.abs(ovr.astype(np.float64) - thresh)
        idx = np.arange(i0, min(i0 + chunk, n))
        d[idx - i0, idx] = np.inf                                   # a box against itself is never compared
        best = min(best, float(d.min()))
    s = np.asarray(sorted_scores, dtype=np.float32).ravel()
    gap = float(np.min(-np.diff(s.astype(np.float64)))) if len(s) > 1 else float("inf")
    return best, gap


def compare_forward(params, info, dbg, dev, layer_tol=1e-3, head_tol=1e-3):
    """dbg: the oracle's debug dict (faster_rcnn_forward(..., return_debug="layers")); dev: host copies of the device forward's
    `keep=True` outputs (+ "layers": {name: array}).  Returns a flat dict of numbers; `ok` is the conjunction of the bars."""
    rep = {}
    layers = {}
    for name, got in sorted(dev.get("layers", {}).items()):
        if name in dbg.get("layers", {}):
            layers[name] = rel_err(got, dbg["layers"][name])
    rep["layers_rel_err"] = {k: float("%.3g" % v) for k, v in layers.items()}
    rep["layers_worst"] = max(layers.values()) if layers else None
    rep["conv5_3_rel_err"] = rel_err(dev["feat"], dbg["feat"])
    rep["rpn_h_rel_err"] = rel_err(dev["rpn_h"], dbg["rpn_h"]) if "rpn_h" in dev else None
    rep["rpn_cls_prob_rel_err"] = rel_err(dev["rpn_cls_prob"], dbg["rpn_cls_prob"])
    rep["rpn_bbox_pred_rel_err"] = rel_err(dev["rpn_bbox_pred"], dbg["rpn_bbox_pred"])
    # ---- proposals: exact given the DEVICE's maps (isolates the proposal kernels from conv rounding) ...
    n = int(dev["n_out"][0])
    p2, s2, d2 = O.proposal_layer(dev["rpn_cls_prob"], dev["rpn_bbox_pred"], info, train=False, return_debug=True)
    rep["n_rois"] = n
    rep["proposals_index_exact_given_device_maps"] = bool(n == len(p2) and np.array_equal(dev["src_index"][:n], d2["src_index"].astype(np.int32)))
    rep["proposals_scores_exact_given_device_maps"] = bool(n == len(p2) and np.array_equal(dev["probs"][:n], s2.ravel()))
    rep["rois_max_abs_diff_given_device_maps"] = float(np.abs(dev["rois"][:n] - p2).max()) if n == len(p2) and n else None
    # ... the same with a CORRECTLY ROUNDED float32 exp in bbox_transform_inv (bbox_transform.py:63-64 calls np.exp on float32: NumPy >= 1.17 evaluates it with
    # a SIMD polynomial of up to 2.5 ulp that differs between CPUs and builds -- a third of the values differ from the rounded double result on this container's
    # AVX-512 path; the NumPy of the reference's time called libm's expf, which is correctly rounded but for rare cases).  The device computes exp in double and
    # rounds once, so HERE everything is bit for bit: indices, scores and the RoIs themselves.  Under NumPy's own exp the RoIs sit within 4 ulp and an NMS
    # decision can flip where |IoU - thresh| is below that noise (tests/test_coord_margins.py measures how far the fixtures are from it).
    exp_was = O.EXP
    O.EXP = lambda v: np.exp(np.asarray(v, np.float64)).astype(np.float32)
    try:
        p3, s3, d3 = O.proposal_layer(dev["rpn_cls_prob"], dev["rpn_bbox_pred"], info, train=False, return_debug=True)
    finally:
        O.EXP = exp_was
    rep["proposals_index_exact_given_device_maps_rounded_exp"] = bool(n == len(p3) and np.array_equal(dev["src_index"][:n], d3["src_index"].astype(np.int32)))
    # ... and with EQUAL scores ordered by the kernel's documented rule (ascending anchor index; NumPy's argsort leaves their order implementation-defined, in
    # ProposalLayer and again inside cpu_nms): fp32 softmax scores of 20 000 anchors tie exactly a dozen times per image, and now and then both boxes of a tied
    # pair survive NMS (2 of 144 size x seed x dtype cases in profiles/r06_size_sweep.txt) -- the lists then differ by that one swap
    O.EXP = lambda v: np.exp(np.asarray(v, np.float64)).astype(np.float32)
    try:
        p4, s4, d4 = O.proposal_layer(dev["rpn_cls_prob"], dev["rpn_bbox_pred"], info, train=False, return_debug=True, tie_rule="ascending_index")
    finally:
        O.EXP = exp_was
    rep["tied_scores_in_sorted_top"] = int((np.diff(d3["sorted_scores"].ravel()) == 0).sum())
    rep["rois_bit_exact_given_device_maps_rounded_exp"] = bool(n == len(p4) and np.array_equal(dev["src_index"][:n], d4["src_index"].astype(np.int32))
                                                               and np.array_equal(dev["rois"][:n], p4) and np.array_equal(dev["probs"][:n].ravel(), s4.ravel()))
    rep["min_abs_iou_minus_thresh_given_device_maps"] = nms_margins(d3["sorted_boxes"], d3["sorted_scores"])[0]
    # ... and from the IMAGE: how many of the device's RoIs are the oracle's own, position by position and as a set
    want_src = dbg["proposal_debug"]["src_index"].astype(np.int64)
    got_src = dev["src_index"][:n].astype(np.int64)
    m = min(len(want_src), len(got_src))
    rep["from_image_rois_oracle"] = int(len(want_src))
    rep["from_image_index_match_positional"] = int((want_src[:m] == got_src[:m]).sum())
    rep["from_image_index_match_set"] = int(len(np.intersect1d(want_src, got_src)))
    iou_margin, score_gap = nms_margins(dbg["proposal_debug"]["sorted_boxes"], dbg["proposal_debug"]["sorted_scores"])
    rep["min_abs_iou_minus_thresh"] = iou_margin
    rep["min_abs_iou_minus_thresh_kept_rows"] = nms_margins(dbg["proposal_debug"]["sorted_boxes"], dbg["proposal_debug"]["sorted_scores"],
                                                            rows=dbg["proposal_debug"]["keep"])[0]
    rep["min_adjacent_score_gap"] = score_gap
    # ---- RoI pooling: exact on the device's own feature map and RoIs
    rois = dev["rois"][:n]
    brois = np.concatenate([np.zeros((n, 1), np.float32), rois], 1)
    pool5 = O.roi_pooling_2d(dev["feat"], brois, 7, 7, 1.0 / 16)
    rep["pool5_exact"] = bool(np.array_equal(dev["pool5"][:n], pool5))
    # ---- head on the device's pool5
    cp, pb, hd = O.rcnn_head(params, pool5, rois, info)
    rep["fc6_rel_err"] = rel_err(dev["fc6"][:n], hd["fc6"]) if "fc6" in dev else None
    rep["fc7_rel_err"] = rel_err(dev["fc7"][:n], hd["fc7"]) if "fc7" in dev else None
    rep["cls_prob_rel_err"] = rel_err(dev["cls_prob"][:n], cp)
    rep["pred_boxes_rel_err"] = rel_err(dev["pred_boxes"][:n], pb)
    # ---- end to end from the image, when the proposal lists coincide
    if len(want_src) == len(got_src) and np.array_equal(want_src, got_src):
        rep["end_to_end_cls_prob_rel_err"] = rel_err(dev["cls_prob"][:n], dbg_cls(dbg))
    else:
        rep["end_to_end_cls_prob_rel_err"] = None
    worst_feat = max(v for v in (rep["layers_worst"], rep["conv5_3_rel_err"], rep["rpn_cls_prob_rel_err"], rep["rpn_bbox_pred_rel_err"]) if v is not None)
    # (index-exact under the platform-independent exp; under this host's NumPy exp too unless an IoU sits inside the exp's 4-ulp noise of the threshold)
    # (against NumPy's own order of equal scores: the same list, or the same list up to swaps inside groups of EQUAL scores)
    got_i, want_i = dev["src_index"][:n].astype(np.int64), d3["src_index"].astype(np.int64)
    swaps_only = bool(n == len(p3) and np.array_equal(np.sort(got_i), np.sort(want_i)) and np.array_equal(dev["probs"][:n].ravel(), s3.ravel()))
    rep["differs_from_numpy_order_only_inside_tied_scores"] = bool(swaps_only and not rep["proposals_index_exact_given_device_maps_rounded_exp"])
    exact = rep["rois_bit_exact_given_device_maps_rounded_exp"] and (rep["proposals_index_exact_given_device_maps_rounded_exp"] or swaps_only) and (
        rep["proposals_index_exact_given_device_maps"] or rep["min_abs_iou_minus_thresh_given_device_maps"] <= 4e-6 or swaps_only)
    rep["ok"] = bool(worst_feat <= layer_tol and exact and rep["pool5_exact"]
                     and rep["cls_prob_rel_err"] <= head_tol and rep["pred_boxes_rel_err"] <= head_tol)
    rep["tolerances"] = {"features_rel": layer_tol, "head_rel": head_tol, "indices": "bit-exact", "pool5": "bit-exact"}
    return rep


def dbg_cls(dbg):
    e = np.exp(dbg["cls_score"] - dbg["cls_score"].max(axis=1, keepdims=True))
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def _bf16_bits_to_f32(a):
    return (a.astype(np.uint16).astype(np.uint32) << 16).view(np.float32)


def device_forward_host(rt, model, x_dev, im_h, im_w):
    """Run model.forward_device(keep=True) with per-layer collection and bring everything to the host (bf16 tensors -- raw bits in
    int16 arrays, channel-blocked for feature maps -- come back as float32 NCHW / (R, N))."""
    collect = {}
    out = model.forward_device(x_dev, im_h, im_w, keep=True, collect=collect)
    rt = getattr(model, "rt", rt)                                     # an fp16 model's runtime decodes its 16-bit arrays as IEEE binary16
    f16 = getattr(rt, "half", "bf16") == "f16"
    rt.mem.synchronize()
    dev = {}
    for k, v in out.items():
        if v is None or not rt.mem.is_array(v):
            continue
        if rt.mem.dtype_of(v) == "i16":
            dev[k] = rt.mem.to_numpy(rt.bf16_to_nchw(v, model.RPN.mid_ch)) if v.ndim == 4 else (rt.mem.to_numpy(v).view(np.float16).astype(np.float32) if f16 else _bf16_bits_to_f32(rt.mem.to_numpy(v)))
        else:
            dev[k] = rt.mem.to_numpy(v)
    layers = {}
    for k, v in collect.items():
        if isinstance(v, tuple):                                  # bf16 stack: (channel-blocked bf16 array, channels)
            layers[k] = rt.mem.to_numpy(rt.bf16_to_nchw(v[0], v[1]))
        else:
            layers[k] = rt.mem.to_numpy(v)
    dev["layers"] = layers
    return dev
