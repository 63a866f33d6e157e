"""How far are the path's DECISIONS from flipping when a decoded box coordinate moves by a few ulp?

bbox_transform_inv evaluates np.exp in float32 (/root/reference/models/bbox_transform.py:63-64).  NumPy's float32 exp is a SIMD
polynomial whose result depends on the host's vector ISA (1-3 ulp from the correctly rounded value); the device evaluates exp in
double and rounds once (correctly rounded).  Box coordinates can therefore differ by a few ulp between any two hosts -- and
between the reference and the device -- while every downstream DECISION (min-size filter, top-K order, NMS survivors, RoI-pooling
bin edges `rint(x/16)`) must not.  This test perturbs every exp() result by up to +-4 ulp (random sign and size per element,
several seeds), re-runs the whole ProposalLayer restatement on the reference-generated golden inputs, and asserts that the
surviving anchor indices and the RoI-pooling bin geometry are unchanged; it prints the margins.
"""
import numpy as np
import pytest

from oracle import frcnn_oracle as O
from oracle.parity import nms_margins

CASES = ["proposal_38x63_test", "proposal_38x63_test_HH", "proposal_38x63_train", "proposal_38x63_cfg4_1000_300", "proposal_37x50_test",
         "proposal_14x14_train_rand"]


def perturbed_exp(seed, k):
    rs = np.random.RandomState(seed)

    def f(v):
        e = np.exp(v)
        steps = rs.randint(-k, k + 1, size=e.shape).astype(np.int32)
        bits = e.view(np.int32) + steps
        return bits.view(np.float32) if e.dtype == np.float32 else e
    return f


def bin_geometry(rois, scale=np.float32(0.0625)):
    """The integers RoI pooling derives from a box (roi_pooling_2d.py forward_cpu: round() of the float32 product)."""
    return np.rint((rois.astype(np.float32) * scale)).astype(np.int64)


@pytest.mark.parametrize("case", CASES)
def test_decisions_survive_4ulp_exp_noise(golden, case):
    G = golden(case)
    pre, post = int(G["pre"]), int(G["post"])
    run = lambda: O.proposal_layer(G["rpn_cls_prob"], G["rpn_bbox_pred"], G["img_info"], pre_nms_top_n=pre, post_nms_top_n=post,
                                   return_debug=True)
    p0, s0, d0 = run()
    assert np.array_equal(p0, G["proposals"])                       # the unperturbed restatement is the pinned one
    geo0 = bin_geometry(p0)
    try:
        for seed in range(4):
            for k in (1, 4):
                O.EXP = perturbed_exp(seed, k)
                p1, s1, d1 = run()
                assert np.array_equal(d1["src_index"], d0["src_index"]), (case, seed, k)     # same anchors survive, same order
                assert np.array_equal(bin_geometry(p1), geo0), (case, seed, k)               # same RoI-pooling bins
                assert np.abs(p1 - p0).max() <= 4e-4                                          # the coordinates did move (by ulps)
    finally:
        O.EXP = np.exp
    # margins: distance of x/16 from the nearest rounding boundary (in feature cells), IoU from the threshold, score gaps
    frac = np.abs((p0.astype(np.float64) * 0.0625) % 1.0 - 0.5)
    unclipped = (p0 != np.rint(p0))                                  # clipped / integer coordinates do not carry exp noise
    bin_margin = float(frac[unclipped].min()) if unclipped.any() else float("inf")
    iou_margin, gap = nms_margins(d0["sorted_boxes"], d0["sorted_scores"])
    iou_kept, _ = nms_margins(d0["sorted_boxes"], d0["sorted_scores"], rows=d0["keep"])
    print("%s: min |frac(x/16) - .5| = %.3g cells (4 ulp at x=1000 is 3e-5 cells); min |IoU - 0.7| = %.3g over all pairs, %.3g over the "
          "rows of kept boxes (the comparisons greedy NMS acts on); min score gap = %.3g" % (case, bin_margin, iou_margin, iou_kept, gap))
    assert bin_margin > 4 * 1000 * 2.0 ** -23 * 0.0625 and gap > 0
