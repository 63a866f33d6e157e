"""Pin the CPU oracle (oracle/) to the reference: bit-for-bit against the golden vectors produced by
the reference's own code (tests/make_golden.py) and, when /root/reference is present, against the
reference executed live.  CPU only."""
import numpy as np
import pytest

from oracle import frcnn_oracle as O
from oracle import ref_harness as rh

PROPOSAL_CASES = ["proposal_14x14_train_rand", "proposal_38x63_test", "proposal_38x63_test_HH",
                  "proposal_38x63_train", "proposal_38x63_cfg4_1000_300", "proposal_37x50_test"]


def test_anchors(golden):
    g = golden("anchors")
    assert np.array_equal(O.generate_anchors(15, (0.5, 1, 2), (8, 16, 32)), g["a_8_16_32"])
    assert np.array_equal(O.generate_anchors(15, (0.5, 1, 2), (4, 8, 16, 32)), g["a_4_8_16_32"])
    # the live table is [-84,-40,99,55]..., not the 1-based table in the file's comment
    assert g["a_8_16_32"][0].tolist() == [-84.0, -40.0, 99.0, 55.0]


@pytest.mark.parametrize("case", PROPOSAL_CASES)
def test_proposal_layer(golden, case):
    g = golden(case)
    p, s, d = O.proposal_layer(g["rpn_cls_prob"], g["rpn_bbox_pred"], g["img_info"], train=bool(g["train"]),
                               pre_nms_top_n=int(g["pre"]), post_nms_top_n=int(g["post"]), return_debug=True)
    assert np.array_equal(d["keep0"], g["keep0"])
    assert np.array_equal(d["order"], g["order"])
    assert np.array_equal(d["sorted_boxes"], g["sorted_boxes"])
    assert np.array_equal(d["keep"], g["nms_keep"][:len(d["keep"])])
    assert p.dtype == np.float32 and np.array_equal(p, g["proposals"])
    assert np.array_equal(s, g["probs"])


@pytest.mark.parametrize("case", PROPOSAL_CASES)
def test_proposal_layer_tie_rule_is_a_refinement(golden, case):
    """`tie_rule="ascending_index"` (the HIP path's documented order of EQUAL scores) only decides what NumPy's argsort leaves open: on the
    reference-generated fixtures it returns the reference's lists, and on a constructed tie it returns the lower anchor index first."""
    g = golden(case)
    p, s, d = O.proposal_layer(g["rpn_cls_prob"], g["rpn_bbox_pred"], g["img_info"], train=bool(g["train"]), pre_nms_top_n=int(g["pre"]),
                               post_nms_top_n=int(g["post"]), return_debug=True, tie_rule="ascending_index")
    assert np.array_equal(p, g["proposals"]) and np.array_equal(s, g["probs"])
    if case == PROPOSAL_CASES[0]:
        # a constructed tie: every anchor at 0.25 but two far-apart ones at 0.9, zero deltas (the anchors themselves, clipped)
        A = g["rpn_cls_prob"].shape[1] // 2
        fh, fw = g["rpn_cls_prob"].shape[2:]
        v = np.full(fh * fw * A, 0.25, np.float32)
        lo, hi = 4, fh * fw * A - 5                      # anchor enumeration index = (y * fw + x) * A + a; both are mid-sized anchors
        v[lo] = v[hi] = 0.9
        prob = np.zeros_like(g["rpn_cls_prob"])
        prob[0, A:] = v.reshape(fh, fw, A).transpose(2, 0, 1)
        _, _, d2 = O.proposal_layer(prob, np.zeros_like(g["rpn_bbox_pred"]), g["img_info"], train=False, return_debug=True, tie_rule="ascending_index")
        src = d2["keep0"][d2["order"]]
        assert src[:2].tolist() == [lo, hi] and np.all(np.diff(src[2:]) > 0)      # the tied pair, then the 0.25 block, both in ascending index


def test_cpu_nms(golden):
    g = golden("cpu_nms")
    for tag in ("n6000_t07", "n300_t03", "n1_t07", "n65_t05"):
        keep = O.cpu_nms(g[tag + "_dets"], float(g[tag + "_thresh"]))
        assert keep == g[tag + "_keep"].tolist(), tag
    for thr in (0.7, 0.5, 0.3):
        assert O.cpu_nms(g["edge_dets"], thr) == g["edge_keep_%02d" % int(thr * 10)].tolist()
    # `ovr >= thresh` in double: IoU==0.7f survives 0.7, IoU==0.5 dies at 0.5, IoU==0.3f dies at 0.3
    assert g["edge_keep_07"].tolist() == [0, 1, 2, 3, 4, 5]
    assert g["edge_keep_05"].tolist() == [0, 2, 4, 5]
    assert g["edge_keep_03"].tolist() == [0, 2, 4]
    assert O.cpu_nms(np.zeros((0, 5), np.float32), 0.7) == []
    assert O.cpu_nms_py(g["n65_t05_dets"], 0.5) == g["n65_t05_keep"].tolist()
    with pytest.raises(ValueError):
        O.cpu_nms(g["edge_dets"].astype(np.float64), 0.7)
    with pytest.raises(TypeError):
        O.cpu_nms(g["edge_dets"], 1)


def test_edge_cases(golden):
    """The oracle on the reference-generated edge fixtures: NaN scores of either sign, +-inf scores, NaN / +-inf deltas, exp overflow,
    NaN / inf coordinates into cpu_nms (tests/make_golden.py edge_cases)."""
    import warnings
    g = golden("edge_cases")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for tag in ("posnan", "negnan", "infs", "negnan_inf", "negnan_train", "deltas"):
            p, s, d = O.proposal_layer(g["p_%s_prob" % tag], g["p_%s_pred" % tag], np.array([[224, 224]], np.int32),
                                       train=bool(g["p_%s_train" % tag]), return_debug=True)
            assert np.array_equal(p, g["p_%s_proposals" % tag]), tag
            assert np.array_equal(s.view(np.uint32), g["p_%s_probs" % tag].view(np.uint32)), tag
            assert np.array_equal(d["keep0"][d["order"]][d["keep"]], g["p_%s_src" % tag]), tag
        for tag in ("negnan_score", "posnan_score", "nan_x1", "nan_y2", "inf_score", "inf_x2"):
            for thr in (0.7, 0.3):
                want = g["n_%s_keep_%02d" % (tag, int(thr * 10))].tolist()
                assert O.cpu_nms(g["n_%s_dets" % tag], thr) == want, (tag, thr)
                assert O.cpu_nms_py(g["n_%s_dets" % tag], thr) == want, (tag, thr)


def test_bbox_overlaps(golden):
    g = golden("bbox_overlaps")
    assert np.array_equal(O.bbox_overlaps(g["boxes"], g["query"]), g["overlaps"])


def test_bbox_transforms(golden):
    g = golden("bbox_transform")
    inv = O.bbox_transform_inv(g["boxes"], g["trans"])
    assert np.array_equal(inv, g["inv"])
    clipped = O.clip_boxes(inv.copy(), np.array([600, 1000], np.int32))
    assert np.array_equal(clipped, g["clipped"])
    assert np.array_equal(O.filter_boxes(clipped[:, :4], 16), g["filt"])
    assert np.array_equal(O.bbox_transform(g["boxes"].astype(np.float64), g["gt"]), g["fwd"])
    assert O.bbox_transform_inv(np.zeros((0, 4), np.float32), np.zeros((0, 8), np.float32)).shape == (0, 8)


@pytest.mark.parametrize("tag,fh,fw", [("a", 14, 14), ("b", 38, 63)])
def test_anchor_target_layer(golden, tag, fh, fw):
    g = golden("anchor_target")
    rng = np.random.RandomState(int(g[tag + "_seed"]))
    l, t, ii, n_all = O.anchor_target_layer(fh, fw, g[tag + "_gt"], g[tag + "_info"], rng=rng)
    assert n_all == int(g[tag + "_nall"])
    assert np.array_equal(ii, g[tag + "_inds"])
    assert l.dtype == np.int32 and np.array_equal(l, g[tag + "_labels"])
    assert t.dtype == np.float32 and np.array_equal(t, g[tag + "_targets"])
    # the reference's own (weak) assertions, tests/test_anchor_target_layer.py:76-77,88
    assert len(l) == len(ii) == len(t) and set(np.unique(l)) <= {-1, 0, 1}


def test_roi_pool_c_matches_python_twin():
    rs = np.random.RandomState(3)
    x = rs.randn(1, 5, 38, 63).astype(np.float32)
    rois = np.zeros((40, 5), np.float32)
    x1 = rs.uniform(0, 900, 40); y1 = rs.uniform(0, 500, 40)
    rois[:, 1], rois[:, 2] = x1, y1
    rois[:, 3] = np.minimum(x1 + rs.uniform(0, 500, 40), 999)
    rois[:, 4] = np.minimum(y1 + rs.uniform(0, 400, 40), 599)
    rois[:8, 1:] = np.round(rois[:8, 1:] / 8) * 8        # exact .5 after *1/16 -> exercises half-to-even
    rois[8] = [0, 990, 590, 999, 599]                     # tiny RoI: many bins share one cell
    rois[9] = [0, 1200, 700, 1300, 800]                   # fully outside -> all bins empty -> 0 / -1
    y, am = O.roi_pooling_2d(x, rois, return_argmax=True)
    y2, am2 = O.roi_pooling_2d_py(x, rois)
    assert np.array_equal(y, y2) and np.array_equal(am, am2)
    assert (y[9] == 0).all() and (am[9] == -1).all()
    dy = rs.randn(*y.shape).astype(np.float32)
    dx = O.roi_pooling_2d_backward(dy, am, rois, x.shape)
    ref = np.zeros_like(x).reshape(1, 5, -1)
    for r in range(40):
        for c in range(5):
            for p in range(49):
                a = am[r, c].ravel()[p]
                if a >= 0:
                    ref[0, c, a] += dy[r, c].ravel()[p]
    assert np.array_equal(dx.reshape(1, 5, -1), ref)


def test_roi_pool_nan_rule_is_the_stated_deviation():
    """Chainer's forward_cpu (numpy.max / numpy.argmax, the NumPy twin below) PROPAGATES a NaN; the C restatement -- and every kernel, which the
    parity tests compare with it -- follows forward_gpu's scan: a NaN in a bin's first cell stays, a NaN elsewhere never wins (include/frcnn_hip.h,
    RoIPooling2D block: a stated deviation).  This test pins exactly where the two differ and that they differ nowhere else."""
    rs = np.random.RandomState(5)
    x = rs.randn(1, 2, 12, 17).astype(np.float32)
    rois = np.array([[0, 0, 0, 16 * 16 - 1, 11 * 16 - 1]], np.float32)      # the whole map: bin (ph, pw) covers rows / columns known below
    x[0, 0, 0, 0] = np.nan                                                   # the FIRST cell of bin (0, 0): both rules say NaN
    x[0, 1, 5, 8] = np.nan                                                   # an interior cell of some bin: numpy.max says NaN, the scan ignores it
    y, am = O.roi_pooling_2d(x, rois, return_argmax=True)
    y2, am2 = O.roi_pooling_2d_py(x, rois)
    assert np.isnan(y[0, 0, 0, 0]) and np.isnan(y2[0, 0, 0, 0]) and am[0, 0, 0, 0] == am2[0, 0, 0, 0] == 0
    differ = np.isnan(y2) & ~np.isnan(y)
    assert differ.sum() >= 1 and differ[0, 1].sum() == differ.sum()          # only channel 1's bins that hold the interior NaN
    for ph, pw in zip(*np.nonzero(differ[0, 1])):
        assert am2[0, 1, ph, pw] == 5 * 17 + 8                               # numpy.argmax: the NaN's position
        assert np.isfinite(y[0, 1, ph, pw])                                  # the scan: the maximum of the bin's other cells
        rest = x[0, 1].copy(); rest[5, 8] = -np.inf
        assert y[0, 1, ph, pw] == rest.reshape(-1)[am[0, 1, ph, pw]]
    same = ~differ
    assert np.array_equal(np.nan_to_num(y[same], nan=-1.0), np.nan_to_num(y2[same], nan=-1.0)) and np.array_equal(am[same], am2[same])


def test_roi_bin_edges_need_double_arithmetic():
    """Chainer's CPU path derives bin edges in Python doubles: floor(p*(rh/7.)), ceil((p+1)*(rh/7.)).
    That is NOT the exact-rational formula: 7*(29/7.) = 29.000000000000004 -> ceil = 30, so the last bin
    of a 29-cell RoI reaches one cell further.  The HIP kernel therefore evaluates the same two IEEE
    double operations (never integer division, never float) -- this test documents why."""
    assert int(np.ceil(7 * (1. * 29 / 7))) == 30 and (7 * 29 + 6) // 7 == 29
    bad = 0
    for r in range(1, 2048):
        s = 1. * r / 7
        for p in range(7):
            assert int(np.floor(p * s)) == (p * r) // 7          # floors do agree
            bad += int(np.ceil((p + 1) * s)) != ((p + 1) * r + 6) // 7
    assert bad > 0


@pytest.mark.skipif(not rh.available(), reason="reference tree not present (GPU box)")
def test_live_reference_random_sweep():
    """Beyond the committed fixtures: fresh seeds, oracle vs the reference executed right now."""
    ns = rh.load()
    for seed in range(3):
        rs = np.random.RandomState(100 + seed)
        fh, fw = rs.randint(8, 30), rs.randint(8, 40)
        prob = rs.permutation(18 * fh * fw).reshape(1, 18, fh, fw).astype(np.float32) / (18 * fh * fw)
        pred = (rs.randn(1, 36, fh, fw) * 0.3).astype(np.float32)
        info = np.array([[fh * 16, fw * 16]], np.int32)
        pl = ns.ProposalLayer()
        pl.train = bool(seed % 2)
        rp, rsn = pl(ns.Variable(prob.copy()), ns.Variable(pred.copy()), ns.Variable(info))
        op, osn = O.proposal_layer(prob, pred, info, train=bool(seed % 2))
        assert np.array_equal(rp, op) and np.array_equal(rsn, osn)
        n = 500
        x1 = rs.uniform(0, 300, n); y1 = rs.uniform(0, 300, n)
        d = np.stack([x1, y1, x1 + rs.uniform(1, 200, n), y1 + rs.uniform(1, 200, n),
                      rs.permutation(n) / float(n)], 1).astype(np.float32)
        for thr in (0.3, 0.5, 0.7):
            assert ns.cpu_nms(d, thr) == O.cpu_nms(d, thr)


def test_ref_native_so_travels():
    """oracle/_ref/*.so (built from /root/reference) must be loadable without the reference tree."""
    from oracle import build_ref
    if not build_ref.built():
        pytest.skip("oracle/_ref not built")
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "cpu_nms.npz"))
    assert rh.native("cpu_nms").cpu_nms(g["n300_t03_dets"], 0.3) == g["n300_t03_keep"].tolist()


def test_proposal_target_layer(golden):
    """oracle restatement of models/proposal_target_layer.py:84-150 vs vectors produced by the reference's own class."""
    g = golden("proposal_target")
    for tag in "abc":
        rng = np.random.RandomState(int(g[tag + "_seed"]))
        ug, ext, keep = O.proposal_target_layer(g[tag + "_props"], g[tag + "_gt"], rng=rng)
        assert keep.dtype == np.int32 and np.array_equal(keep, g[tag + "_keep"])
        assert np.array_equal(ug, g[tag + "_use_gt"]) and np.array_equal(ext, g[tag + "_ext"])
