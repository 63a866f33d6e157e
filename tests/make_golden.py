#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own code (build container only).

Uses oracle/ref_harness.py to import /root/reference/models/*.py under a stub chainer and the
reference's Cython modules built into oracle/_ref/.  The reference's tests pin no output values
(SURVEY.md section 4), so these fixtures -- inputs AND the reference's outputs -- are the
known-answer vectors both the oracle (tests -m "not gpu") and the HIP path (tests -m gpu) are
held to.  Re-run:  python tests/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def unique_scores_softmax(rs, shape):
    """18-way softmax of N(0,1) logits (region_proposal_network.py:119) with de-duplicated fg scores
    so NumPy's unstable argsort tie order cannot leak into the fixture."""
    logits = rs.randn(*shape).astype(np.float32)
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    prob = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    A = shape[1] // 2
    fg = prob[0, A:].ravel()
    while True:
        u, idx, cnt = np.unique(fg, return_index=True, return_counts=True)
        if len(u) == len(fg):
            break
        dup = np.ones(len(fg), bool)
        dup[idx] = False
        fg[dup] = np.nextafter(fg[dup], np.float32(2.0)) + (rs.rand(dup.sum()) * 1e-6).astype(np.float32)
    prob[0, A:] = fg.reshape(prob[0, A:].shape)
    return prob


def proposal_case(ns, name, fh, fw, img, train, seed, kind, pre=None, post=None):
    rs = np.random.RandomState(seed)
    if kind == "rand":      # tests/test_proposal_layer.py:25-29 recipe (rand probs, rand deltas)
        prob = rs.rand(1, 18, fh, fw).astype(np.float32)
        fg = prob[0, 9:].ravel()
        assert len(np.unique(fg)) == len(fg) or True
        # de-duplicate
        u, idx = np.unique(fg, return_index=True)
        dup = np.ones(len(fg), bool); dup[idx] = False
        k = 0
        while dup.any():
            fg[dup] = rs.rand(dup.sum()).astype(np.float32)
            u, idx = np.unique(fg, return_index=True)
            dup = np.ones(len(fg), bool); dup[idx] = False
            k += 1
        prob[0, 9:] = fg.reshape(prob[0, 9:].shape)
        pred = rs.rand(1, 36, fh, fw).astype(np.float32)
    else:                   # SURVEY.md section 8d stage-isolated bench inputs
        prob = unique_scores_softmax(rs, (1, 18, fh, fw))
        pred = (rs.randn(1, 36, fh, fw) * 0.2).astype(np.float32)
    info = np.array([img], dtype=np.int32)
    pl = ns.ProposalLayer()
    pl.train = train
    if pre is not None:
        pl._pre_nms_top_n, pl._post_nms_top_n = pre, post
    props, probs = pl(ns.Variable(prob.copy()), ns.Variable(pred.copy()), ns.Variable(info))
    # the sorted pre-NMS set, re-derived with the reference's own functions (proposal_layer.py:135-170)
    all_bbox = pl._generate_all_bbox_use_array_info(pred[0])
    trans = pred[0].transpose(1, 2, 0).reshape(-1, 4)
    dec = ns.clip_boxes(ns.bbox_transform_inv(all_bbox, trans), info[0])
    keep0 = ns.filter_boxes(dec, pl._min_size)
    fg = prob[0, 9:].transpose(1, 2, 0).reshape(-1, 1)[keep0]
    order = fg.ravel().argsort()[::-1][:pl._pre_nms_top_n]
    sorted_boxes, sorted_scores = dec[keep0][order], fg[order]
    keep = np.asarray(ns.cpu_nms(np.hstack((sorted_boxes, sorted_scores)), pl._nms_thresh), dtype=np.int64)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), rpn_cls_prob=prob, rpn_bbox_pred=pred, img_info=info,
        train=np.array(train), pre=np.array(pl._pre_nms_top_n), post=np.array(pl._post_nms_top_n),
        proposals=props, probs=probs, decoded_clipped=dec, keep0=keep0, order=order.astype(np.int64),
        sorted_boxes=sorted_boxes, sorted_scores=sorted_scores, nms_keep=keep)
    print(name, props.shape, "n_valid", len(keep0), "nms survivors", len(keep))


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = rh.load()
    # ---- anchors (generate_anchors.py:47-93)
    np.savez_compressed(os.path.join(OUT, "anchors.npz"),
                        a_8_16_32=ns.generate_anchors(ratios=(0.5, 1, 2), scales=(8, 16, 32)),
                        a_4_8_16_32=ns.generate_anchors(ratios=(0.5, 1, 2), scales=(4, 8, 16, 32)),
                        a_default=ns.generate_anchors())
    # ---- ProposalLayer
    proposal_case(ns, "proposal_14x14_train_rand", 14, 14, (224, 224), True, 11, "rand")
    proposal_case(ns, "proposal_38x63_test", 38, 63, (600, 1000), False, 0, "bench")
    proposal_case(ns, "proposal_38x63_test_HH", 38, 63, (600, 600), False, 2, "bench")      # forward.py:93 quirk
    proposal_case(ns, "proposal_38x63_train", 38, 63, (600, 1000), True, 3, "bench")
    proposal_case(ns, "proposal_38x63_cfg4_1000_300", 38, 63, (600, 1000), False, 4, "bench", 1000, 300)
    proposal_case(ns, "proposal_37x50_test", 37, 50, (600, 800), False, 5, "bench")        # test_region_proposal_network.py:18-23
    # ---- cpu_nms (cpu_nms.pyx:18-69)
    rs = np.random.RandomState(7)
    nms = {}
    for tag, n, thr in (("n6000_t07", 6000, 0.7), ("n300_t03", 300, 0.3), ("n1_t07", 1, 0.7), ("n65_t05", 65, 0.5)):
        x1 = rs.uniform(0, 900, n); y1 = rs.uniform(0, 500, n)
        w = rs.uniform(16, 400, n); h = rs.uniform(16, 300, n)
        sc = rs.permutation(n).astype(np.float64) / n
        d = np.stack([x1, y1, np.minimum(x1 + w, 999), np.minimum(y1 + h, 599), sc], 1).astype(np.float32)
        nms[tag + "_dets"] = d
        nms[tag + "_thresh"] = np.array(thr)
        nms[tag + "_keep"] = np.asarray(ns.cpu_nms(d, thr), dtype=np.int64)
    # exact-threshold semantics: `ovr >= thresh` compared in double (cpu_nms.pyx:18,66)
    edge = np.array([[0, 0, 9, 0, 0.9], [0, 0, 6, 0, 0.8],      # IoU = 7/10 -> 0.7f < 0.7  => kept at 0.7
                     [100, 0, 109, 0, 0.7], [100, 0, 104, 0, 0.6],  # IoU = 0.5 exactly      => suppressed at 0.5
                     [200, 0, 209, 0, 0.5], [200, 0, 202, 0, 0.4]],  # IoU = 3/10 -> 0.3f > 0.3 => suppressed at 0.3
                    dtype=np.float32)
    nms["edge_dets"] = edge
    for thr in (0.7, 0.5, 0.3):
        nms["edge_keep_%02d" % int(thr * 10)] = np.asarray(ns.cpu_nms(edge, thr), dtype=np.int64)
    nms["empty_keep"] = np.asarray(ns.cpu_nms(np.zeros((0, 5), np.float32), 0.7), dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "cpu_nms.npz"), **nms)
    print("cpu_nms survivors", {k: len(v) for k, v in nms.items() if k.endswith("keep") or "_keep_" in k})
    # ---- bbox_overlaps (bbox.pyx:16-56)
    rs = np.random.RandomState(8)
    b = np.sort(rs.uniform(0, 600, (500, 2, 2)), axis=1).transpose(0, 2, 1).reshape(500, 4)
    b = np.stack([b[:, 0], b[:, 2], b[:, 1], b[:, 3]], 1)
    q = b[rs.choice(500, 7, replace=False)] + rs.randint(-5, 5, (7, 4))
    np.savez_compressed(os.path.join(OUT, "bbox_overlaps.npz"), boxes=b, query=q, overlaps=ns.bbox_overlaps(b, q))
    # ---- bbox transforms (bbox_transform.py)
    rs = np.random.RandomState(9)
    xy = rs.uniform(0, 400, (200, 2))
    boxes = np.hstack([xy, xy + rs.uniform(30, 400, (200, 2))]).astype(np.float32)
    trans = (rs.randn(200, 84) * 0.3).astype(np.float32)
    inv = ns.bbox_transform_inv(boxes, trans)
    clipped = ns.clip_boxes(inv.copy(), np.array([600, 1000], np.int32))
    gtb = (boxes + rs.randint(-10, 10, boxes.shape)).astype(np.float64)
    np.savez_compressed(os.path.join(OUT, "bbox_transform.npz"), boxes=boxes, trans=trans, inv=inv, clipped=clipped,
                        filt=ns.filter_boxes(clipped[:, :4], 16), gt=gtb,
                        fwd=ns.bbox_transform(boxes.astype(np.float64), gtb))
    # ---- AnchorTargetLayer (anchor_target_layer.py:66-198)
    atl = ns.AnchorTargetLayer(16, [0.5, 1, 2], [8, 16, 32])
    gt = np.array([[[10, 10, 60, 200, 0], [50, 100, 210, 210, 1], [160, 40, 200, 70, 2]]], dtype=np.float32)  # tests/test_anchor_target_layer.py:23-27
    info = np.array([[224, 224]], dtype=np.int32)
    np.random.seed(21)
    l, t, ii, n_all = atl(14, 14, ns.Variable(gt), ns.Variable(info))
    at = dict(a_gt=gt, a_info=info, a_seed=np.array(21), a_labels=l, a_targets=t, a_inds=ii, a_nall=np.array(n_all))
    rs = np.random.RandomState(22)
    G = 5
    x1 = rs.uniform(0, 700, G); y1 = rs.uniform(0, 350, G)
    gt2 = np.stack([x1, y1, x1 + rs.uniform(32, 300, G), y1 + rs.uniform(32, 240, G), rs.randint(1, 21, G)], 1)[None].astype(np.float32)
    info2 = np.array([[600, 1000]], dtype=np.int32)
    np.random.seed(23)
    l, t, ii, n_all = atl(38, 63, ns.Variable(gt2), ns.Variable(info2))
    at.update(b_gt=gt2, b_info=info2, b_seed=np.array(23), b_labels=l, b_targets=t, b_inds=ii, b_nall=np.array(n_all))
    np.savez_compressed(os.path.join(OUT, "anchor_target.npz"), **at)
    print("anchor target:", at["a_labels"].shape, (at["a_labels"] == 1).sum(), at["b_labels"].shape,
          (at["b_labels"] == 1).sum(), (at["b_labels"] == 0).sum())


def edge_cases(ns):
    """Inputs SURVEY 8a-9 / 8a-11 name and the reference's own tests never feed: NaN scores of either sign bit, +-inf scores,
    NaN / +-inf deltas, an exp() overflow (dw = 100), and for cpu_nms a NaN score of either sign and a NaN coordinate.  Outputs are
    the reference's (ProposalLayer under the stub chainer, cpu_nms.pyx compiled in place).  One special value of a kind per case, so
    NumPy's implementation-defined order among equal keys (all NaNs compare equal) cannot leak into the fixture.  -> edge_cases.npz"""
    import warnings
    warnings.simplefilter("ignore", RuntimeWarning)
    out = {}
    fh = fw = 14
    info = np.array([[224, 224]], dtype=np.int32)
    NEG_NAN = np.array([0xFFC00000], np.uint32).view(np.float32)[0]      # what x86 makes of inf - inf
    POS_NAN = np.array([0x7FC00000], np.uint32).view(np.float32)[0]

    def base(seed):
        rs = np.random.RandomState(seed)
        return unique_scores_softmax(rs, (1, 18, fh, fw)), (rs.randn(1, 36, fh, fw) * 0.2).astype(np.float32)

    def run(tag, prob, pred, train=False):
        pl = ns.ProposalLayer()
        pl.train = train
        props, probs = pl(ns.Variable(prob.copy()), ns.Variable(pred.copy()), ns.Variable(info))
        all_bbox = pl._generate_all_bbox_use_array_info(pred[0])
        trans = pred[0].transpose(1, 2, 0).reshape(-1, 4)
        dec = ns.clip_boxes(ns.bbox_transform_inv(all_bbox, trans), info[0])
        keep0 = ns.filter_boxes(dec, pl._min_size)
        fg = prob[0, 9:].transpose(1, 2, 0).reshape(-1, 1)[keep0]
        order = fg.ravel().argsort()[::-1][:pl._pre_nms_top_n]
        keep = np.asarray(ns.cpu_nms(np.hstack((dec[keep0][order], fg[order])), pl._nms_thresh), dtype=np.int64)
        src = keep0[order][keep[:len(props)]]
        assert np.array_equal(dec[src], props), tag                   # the index chain reproduces the layer's own output
        out.update({"p_%s_prob" % tag: prob, "p_%s_pred" % tag: pred, "p_%s_train" % tag: np.array(train), "p_%s_proposals" % tag: props,
                    "p_%s_probs" % tag: probs, "p_%s_src" % tag: src.astype(np.int64), "p_%s_nvalid" % tag: np.array(len(keep0))})
        print("edge", tag, props.shape, "n_valid", len(keep0), "first src", src[:4])

    # ---- scores: one NaN of each sign, +-inf (fg channel 9 + a at (h, w); a mid-map anchor so that the box survives filter_boxes)
    for tag, vals in (("posnan", [POS_NAN]), ("negnan", [NEG_NAN]), ("infs", [np.float32(np.inf), np.float32(-np.inf)]),
                      ("negnan_inf", [NEG_NAN, np.float32(np.inf)])):
        prob, pred = base(41)
        for k, v in enumerate(vals):
            prob[0, 9 + 4 + k, 6 + k, 7] = v
        run(tag, prob, pred)
    prob, pred = base(42)
    prob[0, 9 + 1, 5, 5] = NEG_NAN
    run("negnan_train", prob, pred, train=True)
    # ---- deltas: (anchor a, h, w, coordinate, value); the touched anchors get the top scores so a wrongly kept box would show
    prob, pred = base(43)
    cases = [(4, 3, 3, 0, np.nan), (4, 3, 6, 1, np.inf), (4, 3, 9, 2, 100.0), (4, 6, 3, 3, -np.inf), (4, 6, 6, 2, np.inf),
             (4, 6, 9, 0, -np.inf), (4, 9, 3, 2, np.nan), (4, 9, 6, 3, 100.0), (4, 9, 9, 2, 88.0), (1, 7, 7, 3, -100.0), (7, 7, 7, 2, 87.0)]
    top = np.sort(prob[0, 9:].ravel())[::-1]
    for k, (a, h, w, c, v) in enumerate(cases):
        pred[0, a * 4 + c, h, w] = v
        prob[0, 9 + a, h, w] = np.nextafter(np.float32(1.0), np.float32(0)) - np.float32(k * 1e-6)
    pred[0, 4 * 4 + 0, 9, 6] = np.inf                                # dx = inf together with dh = 100: inf - inf = NaN corner
    assert len(np.unique(prob[0, 9:])) == prob[0, 9:].size
    run("deltas", prob, pred)
    # ---- cpu_nms (cpu_nms.pyx:18-69): a NaN score of either sign is ordered first by argsort()[::-1]; a NaN coordinate poisons
    # the IoU of every pair it enters through max / min helpers that are NOT symmetric in NaN (cpu_nms.pyx:12-16)
    rs = np.random.RandomState(44)
    n = 200
    x1 = rs.uniform(0, 300, n); y1 = rs.uniform(0, 300, n)
    d = np.stack([x1, y1, x1 + rs.uniform(20, 200, n), y1 + rs.uniform(20, 200, n), rs.permutation(n) / float(n)], 1).astype(np.float32)
    for tag, fn in (("negnan_score", lambda a: a.__setitem__((57, 4), NEG_NAN)), ("posnan_score", lambda a: a.__setitem__((57, 4), POS_NAN)),
                    ("nan_x1", lambda a: a.__setitem__((31, 0), np.nan)), ("nan_y2", lambda a: a.__setitem__((140, 3), np.nan)),
                    ("inf_score", lambda a: a.__setitem__((99, 4), np.inf)), ("inf_x2", lambda a: a.__setitem__((12, 2), np.inf))):
        a = d.copy()
        fn(a)
        for thr in (0.7, 0.3):
            keep = np.asarray(ns.cpu_nms(a, thr), dtype=np.int64)
            out["n_%s_dets" % tag] = a
            out["n_%s_keep_%02d" % (tag, int(thr * 10))] = keep
            print("edge nms", tag, thr, len(keep), keep[:5])
    np.savez_compressed(os.path.join(OUT, "edge_cases.npz"), **out)


def proposal_target_cases(ns):
    """ProposalTargetLayer (proposal_target_layer.py:84-150): three proposal sets built around the gt boxes (fg / bg / far)."""
    from oracle import frcnn_oracle as O
    out = {}
    for tag, n, G, seed in (("a", 300, 4, 31), ("b", 300, 1, 32), ("c", 40, 6, 33)):
        rs = np.random.RandomState(seed)
        x1 = rs.uniform(0, 800, G); y1 = rs.uniform(0, 400, G)
        gt = np.stack([x1, y1, x1 + rs.uniform(40, 190, G), y1 + rs.uniform(40, 190, G), rs.randint(1, 21, G)], 1)[None].astype(np.float32)
        k = n // 3
        base = gt[0, rs.randint(0, G, k), :4]
        fg = base + rs.uniform(-8, 8, (k, 4))
        bg = base + rs.uniform(-60, 60, (k, 4)) + np.array([40, 40, 40, 40])
        xy = rs.uniform(0, 900, (n - 2 * k, 2))
        far = np.hstack([xy, xy + rs.uniform(20, 100, (n - 2 * k, 2))])
        props = np.vstack([fg, bg, far]).astype(np.float32)
        props[:, 2:] = np.maximum(props[:, 2:], props[:, :2] + 1)
        props = props[rs.permutation(n)]
        layer = ns.ProposalTargetLayer(16, [0.5, 1, 2], [8, 16, 32], 21)
        np.random.seed(seed + 100)
        ug, ext, keep = layer(props, ns.Variable(gt))
        ug2, ext2, keep2 = O.proposal_target_layer(props, gt, rng=np.random.RandomState(seed + 100))
        assert np.array_equal(ug, ug2) and np.array_equal(ext, ext2) and np.array_equal(keep, keep2), tag
        out.update({tag + "_props": props, tag + "_gt": gt, tag + "_seed": np.array(seed + 100), tag + "_use_gt": ug, tag + "_ext": ext,
                    tag + "_keep": keep})
    np.savez_compressed(os.path.join(OUT, "proposal_target.npz"), **out)


if __name__ == "__main__":
    if "--edges" in sys.argv:          # only the edge fixture (the others regenerate identically; this avoids the churn)
        edge_cases(rh.load())
    else:
        main()
        proposal_target_cases(rh.load())
        edge_cases(rh.load())
