"""RPN training-step cases shared by the CPU (host-emulated kernels, narrow trunk) and GPU (real VGG-16) suites."""
import functools
import json


def json_dumps(o):
    return json.dumps(o, sort_keys=True)


import numpy as np

from oracle import frcnn_oracle as O
import parity_cases as P

SMALL_LAYERS = [("conv1_1", 3, 64), "pool", ("conv2_1", 64, 64), ("conv2_2", 64, 64), "pool"]     # stride 4, 64 channels


def small_params(seed=1, ch=64, n_anchors=9):
    rs = np.random.RandomState(seed)
    p = {}
    for l in SMALL_LAYERS:
        if l == "pool":
            continue
        name, ci, co = l
        p["trunk/%s/W" % name] = (rs.randn(co, ci, 3, 3) * np.sqrt(2.0 / (ci * 9))).astype(np.float32)
        p["trunk/%s/b" % name] = (rs.randn(co) * 0.01).astype(np.float32)
    p["RPN/rpn_conv_3x3/W"] = (rs.randn(ch, ch, 3, 3) * np.sqrt(2.0 / (ch * 9))).astype(np.float32)
    p["RPN/rpn_conv_3x3/b"] = (rs.randn(ch) * 0.01).astype(np.float32)
    p["RPN/rpn_cls_score/W"] = (rs.randn(2 * n_anchors, ch, 1, 1) * 0.05).astype(np.float32)
    p["RPN/rpn_cls_score/b"] = (rs.randn(2 * n_anchors) * 0.01).astype(np.float32)
    p["RPN/rpn_bbox_pred/W"] = (rs.randn(4 * n_anchors, ch, 1, 1) * 0.05).astype(np.float32)
    p["RPN/rpn_bbox_pred/b"] = (rs.randn(4 * n_anchors) * 0.01).astype(np.float32)
    return p


def build_small(rt, params):
    from chainer_faster_rcnn_amd.models import FasterRCNN, VGG16Prev
    model = FasterRCNN(trunk_class=functools.partial(VGG16Prev, layers=SMALL_LAYERS), rpn_in_ch=64, rpn_mid_ch=64, feat_stride=4,
                       anchor_scales=(2, 4, 8), runtime=rt)
    model.trunk.load_params(params, "trunk/")
    model.RPN.load_params(params, "RPN/")
    model.rpn_train = True
    return model


def oracle_step(params, x, gt, info, layers, feat_stride, scales, seed, f64_wgrad=(), float64=False):
    """The oracle's forward/backward for one image: anchor targets with the SAME NumPy RNG state, then autograd."""
    names = [l if l == "pool" else l[0] for l in layers]
    h, w = x.shape[2], x.shape[3]
    for l in layers:
        if l == "pool":
            h, w = (h + 1) // 2, (w + 1) // 2
    np.random.seed(seed)
    labels, targets, inds, n_all = O.anchor_target_layer(h, w, gt, info, feat_stride=feat_stride, anchor_scales=scales)
    loss, grads = O.rpn_train_grads(params, x, labels, targets, inds, n_all, layers=names, f64_wgrad=f64_wgrad, float64=float64)
    return loss, grads


def check_small_step(rt, seed=0, im_h=40, im_w=56, conv_math="mfma"):
    """forward + AnchorTargetLayer + losses + backward on the narrow trunk vs the oracle, then the SGD update."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.train import RPNTrainer
    rs = np.random.RandomState(seed)
    params = small_params()
    x = rs.randn(1, 3, im_h, im_w).astype(np.float32)
    gt = P.gt_case(rs, 3, im_h, im_w)
    gt[0, :, 2] = np.minimum(gt[0, :, 0] + rs.uniform(8, 30, 3), im_w - 1)
    gt[0, :, 3] = np.minimum(gt[0, :, 1] + rs.uniform(8, 30, 3), im_h - 1)
    info = np.array([[im_h, im_w]], dtype=np.int32)
    model = build_small(rt, params)
    tr = RPNTrainer(model, conv_math=conv_math)
    w0 = rt.mem.to_numpy(tr.W)
    np.random.seed(123)
    out = tr.forward_backward(Variable(x), Variable(info), Variable(gt))
    want_loss, want = oracle_step(params, x, gt, info, SMALL_LAYERS, 4, (2, 4, 8), 123)
    got = tr.grads_chainer_layout()
    l = tr.losses_host(out)
    assert abs(l["rpn_loss"] - want_loss) <= 1e-4 * abs(want_loss), (l, want_loss)
    for k in sorted(want):
        scale = max(np.abs(want[k]).max(), 1e-8)
        assert np.abs(got[k] - want[k]).max() <= 1e-3 * scale, (k, np.abs(got[k] - want[k]).max(), scale)     # fp32, 1e-3 relative
    # update: W1 = W0 + v1, v1 = -lr * (g + wd * W0)   (velocity starts at 0)
    g = rt.mem.to_numpy(tr.G)
    tr.update()
    w1, v1 = O.momentum_sgd_wd(w0, g, np.zeros_like(w0))
    assert np.array_equal(rt.mem.to_numpy(tr.W), w1) and np.array_equal(rt.mem.to_numpy(tr.V), v1)
    # the links see the updated weights (views of the flat buffer), and sync_params() restores Chainer's layout
    name, link = tr.convs[1]
    seg = tr.seg[name + "/W"]
    assert np.array_equal(rt.mem.to_numpy(link.Wp), w1[seg.offset:seg.offset + seg.size].reshape(seg.shape))
    tr.sync_params()
    assert np.array_equal(rt.mem.to_numpy(link.W).reshape(link.cout, -1), rt.mem.to_numpy(link.Wp).T)
    return l


def check_vgg_step(rt, im_h=160, im_w=224, seed=0, conv_math="mfma"):
    """One RPN training step of the real VGG-16 FasterRCNN (GPU suite): loss and every gradient vs the oracle's autograd."""
    from chainer_faster_rcnn_amd import synthetic
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.models import FasterRCNN
    from chainer_faster_rcnn_amd.models.vgg16 import LAYERS
    from chainer_faster_rcnn_amd.train import RPNTrainer
    rs = np.random.RandomState(seed)
    params = synthetic.params(seed=1)
    for k in list(params):
        if k.endswith("/b") and (k.startswith("trunk/") or k.startswith("RPN/")):
            params[k] = (rs.randn(*params[k].shape) * 0.01).astype(np.float32)
    x = synthetic.image(seed=4, h=im_h, w=im_w)
    gt = P.gt_case(rs, 4, im_h, im_w)
    info = np.array([[im_h, im_w]], dtype=np.int32)
    model = FasterRCNN(runtime=rt)
    model.load_params(params)
    model.rpn_train = True
    tr = RPNTrainer(model, conv_math=conv_math)
    # Full size (600 x 1000): two fp32 implementations of a 14-layer backward pass take a handful of DIFFERENT discrete decisions (a
    # ReLU whose pre-activation is 1e-8, a max-pool tie); each flips one pixel's gradient, and every upstream gradient moves by
    # ~sqrt(flips / pixels) of its scale -- measured 1.1e-3 ... 1.3e-3 on two layers, all others < 1e-3.  So at full size every conv
    # KERNEL is judged on its own inputs -- float64 accumulation of exactly the (input, upstream gradient) pair it consumed, 1e-4 --
    # and the end-to-end comparison with the fp32 autograd gets 3e-3.  At the small size the plain 1e-3 bar applies.
    # The same happens -- rarely -- at the small size: ONE max-pool window whose two largest cells differ by an ulp routes its gradient
    # to a different cell than the oracle's pass, and because the RPN loss gradient is concentrated on 256 sampled anchors that one
    # re-routing moves conv1_1's weight gradient by percents (seen when the first layer's accumulation order changed: every
    # activation still within 2e-7, dL/dy(conv4_3) off by 0.6 of its maximum in one cell).  So at every size: each weight-gradient
    # kernel against float64 on its own inputs (1e-4), and the end-to-end bar (1e-3 / 3e-3) may only be exceeded -- up to 0.1 -- when the
    # device's own pre-pool maps contain near-tie windows (top two cells of a window within 4 ulp of each other, not equal), which
    # is reported.
    full = im_h * im_w >= 200000                          # the sizes forward.py's rescaling produces (600 x 1000, 800 x 600, 450 x 642 ...): the float64 arbiter below
    size_tag = "%dx%d" % (im_h, im_w)
    convs = tuple(l[0] for l in LAYERS if l != "pool")
    tr.keep_dy, tr.kept_dy = set(convs), {}
    np.random.seed(11)
    out = tr.forward_backward(Variable(x), Variable(info), Variable(gt))
    want_loss, want = oracle_step(params, x, gt, info, LAYERS, 16, (8, 16, 32), 11, f64_wgrad=convs[:2] if full else ())
    l = tr.losses_host(out)
    assert abs(l["rpn_loss"] - want_loss) <= 1e-4 * abs(want_loss), (l, want_loss)
    near_ties = 0
    for pm in tr.kept_dy.get("pool_inputs", []):
        a = rt.mem.to_numpy(pm)[0]
        c, h, w = a.shape
        a = a[:, :h // 2 * 2, :w // 2 * 2].reshape(c, h // 2, 2, w // 2, 2).transpose(0, 1, 3, 2, 4).reshape(c, h // 2, w // 2, 4)
        top = np.sort(a, axis=-1)
        gap = top[..., 3] - top[..., 2]
        near_ties += int(((gap > 0) & (gap <= 4 * np.spacing(top[..., 3]))).sum())
    got = tr.grads_chainer_layout()
    worst, notes = 0.0, {}
    tol = 3e-3 if full else 1e-3
    if full:
        # Full size: a float64 arbiter instead of a relaxed bar.  The oracle's autograd runs ONCE more in float64 (same fp32 parameters,
        # image, anchor targets); per gradient the device must be within max(1e-3, 2 x the distance of torch's own fp32 pass from that
        # float64 result) -- two fp32 backward passes through 14 layers take a handful of different ReLU / max-pool decisions, and how
        # far that moves a gradient is MEASURED on the reference implementation instead of argued.  No escape clause.
        _, want64 = oracle_step(params, x, gt, info, LAYERS, 16, (8, 16, 32), 11, float64=True)
        # ... and the float64 pass once more with the DEVICE's discrete decisions imposed (its ReLU signs, its max-pool winners): the
        # exact gradient of the function the device actually evaluated
        names = [lay if lay == "pool" else lay[0] for lay in LAYERS]
        linp = tr.kept_dy["layer_inputs"]
        post_relu = {n: rt.mem.to_numpy(linp[i + 1]) for i, n in enumerate(names) if n != "pool"}
        pre_pool = [rt.mem.to_numpy(linp[i]) for i, n in enumerate(names) if n == "pool"]
        hh, ww = x.shape[2], x.shape[3]
        for n in names:
            if n == "pool":
                hh, ww = (hh + 1) // 2, (ww + 1) // 2
        np.random.seed(11)
        labels, targets, inds, n_all = O.anchor_target_layer(hh, ww, gt, info, feat_stride=16, anchor_scales=(8, 16, 32))
        _, want64d, flips = O.rpn_train_grads_given_decisions(params, x, labels, targets, inds, n_all, post_relu, pre_pool,
                                                              rt.mem.to_numpy(tr.kept_dy["rpn_mid"]), layers=names)
        table = {}
        for k in sorted(want):
            if k.endswith("@f64"):
                continue
            w64 = want64[k]
            scale = max(float(np.abs(w64).max()), 1e-12)
            e_dev = float(np.abs(got[k].astype(np.float64) - w64).max() / scale)
            e_t32 = float(np.abs(want[k].astype(np.float64) - w64).max() / scale)
            e_pair = float(np.abs(got[k] - want[k]).max() / max(float(np.abs(want[k]).max()), 1e-8))
            e_given = float(np.abs(got[k].astype(np.float64) - want64d[k]).max() / max(float(np.abs(want64d[k]).max()), 1e-12))
            table[k] = {"device_vs_f64": float("%.3g" % e_dev), "torch_fp32_vs_f64": float("%.3g" % e_t32), "device_vs_torch_fp32": float("%.3g" % e_pair),
                        "device_vs_f64_given_device_decisions": float("%.3g" % e_given)}
            worst = max(worst, e_dev)
        print("\nPARITY_TABLE rpn_train_%s%s %s" % (size_tag, "_split_products" if conv_math == "split" else "", json_dumps(table)))
        print("PARITY_FLIPS rpn_train_%s%s %s" % (size_tag, "_split_products" if conv_math == "split" else "",
                                                        json_dumps({"relu_signs_or_pool_winners_that_differ_from_the_float64_pass": flips})))
        exceed = []
        for k, row in sorted(table.items()):
            # (1) the device's ARITHMETIC: against float64 under the device's own decisions -- no decision noise left, a tight bar
            assert row["device_vs_f64_given_device_decisions"] <= 1e-4, (k, row)
            # (2) end to end against the pure float64 pass: max(1e-3, 2 x torch's own fp32 distance); where that is exceeded the
            #     excess is decision flips by (1) -- listed, bounded (5e-3), and the flip counts are printed above
            if row["device_vs_f64"] > max(1e-3, 2.0 * row["torch_fp32_vs_f64"]):
                exceed.append((k, row["device_vs_f64"], row["torch_fp32_vs_f64"]))
            assert row["device_vs_f64"] <= 5e-3, (k, row)
        print("PARITY_EXCEED rpn_train_%s%s %s" % (size_tag, "_split_products" if conv_math == "split" else "",
                                                         json_dumps({"gradients_beyond_max(1e-3, 2 x torch_fp32_vs_f64)": exceed, "total_flips": int(sum(flips.values()))})))
        assert not exceed or sum(flips.values()) > 0, exceed
        for k in sorted(want):                                    # and every weight-gradient KERNEL on its own inputs, as at the small size
            if k.startswith("trunk/") and k.endswith("/W"):
                import torch
                name = k.split("/")[1]
                xin, dy = tr.kept_dy[name]
                xin, dy = rt.mem.to_numpy(xin), rt.mem.to_numpy(dy)
                ref = torch.nn.grad.conv2d_weight(torch.from_numpy(xin).double().reshape(1, xin.shape[-3], xin.shape[-2], xin.shape[-1]),
                                                  tuple(got[k].shape), torch.from_numpy(dy).double().reshape(1, dy.shape[-3], dy.shape[-2], dy.shape[-1]),
                                                  padding=1).numpy()
                kerr = float(np.abs(got[k] - ref).max() / max(np.abs(ref).max(), 1e-8))
                assert kerr <= 1e-4, (k, kerr)
        return l, worst, False
    for k in sorted(want):
        if k.endswith("@f64"):
            continue
        scale = max(np.abs(want[k]).max(), 1e-8)
        err = np.abs(got[k] - want[k]).max() / scale
        if k.startswith("trunk/") and k.endswith("/W"):
            import torch
            name = k.split("/")[1]
            xin, dy = tr.kept_dy[name]
            xin, dy = rt.mem.to_numpy(xin), rt.mem.to_numpy(dy)
            ref = torch.nn.grad.conv2d_weight(torch.from_numpy(xin).double().reshape(1, xin.shape[-3], xin.shape[-2], xin.shape[-1]),
                                              tuple(got[k].shape), torch.from_numpy(dy).double().reshape(1, dy.shape[-3], dy.shape[-2], dy.shape[-1]),
                                              padding=1).numpy()
            kerr = float(np.abs(got[k] - ref).max() / max(np.abs(ref).max(), 1e-8))
            notes[name] = {"kernel_vs_f64": float("%.2g" % kerr), "end_to_end": float("%.2g" % err)}
            if k + "@f64" in want:
                notes[name]["torch_fp32_vs_its_own_f64"] = float("%.2g" % (np.abs(want[k] - want[k + "@f64"]).max() / scale))
            assert kerr <= 1e-4, (k, kerr)
        worst = max(worst, err)
        assert err <= (tol if near_ties == 0 else 0.1), (k, err, near_ties)
    flipped = worst > tol
    if full or flipped:
        print("\n%s weight gradients: %s" % ("full-size" if full else "small-size", notes))
    if flipped:
        print("end-to-end worst %.3g > %.0e with %d near-tie max-pool window(s) in the device's pre-pool maps: a routing decision "
              "differs from the oracle's fp32 pass; every weight-gradient kernel is within 1e-4 of float64 on its own inputs" % (worst, tol, near_ties))
    return l, worst, flipped                      # flipped: the end-to-end bar was exceeded AND near-tie windows explain it (asserted above)


def small_head_params(rs, ch=64, hidden=128, ncls=21):
    p = {}
    p["fc6/W"] = (rs.randn(hidden, ch * 49) * np.sqrt(2.0 / (ch * 49))).astype(np.float32)
    p["fc6/b"] = (rs.randn(hidden) * 0.01).astype(np.float32)
    p["fc7/W"] = (rs.randn(hidden, hidden) * np.sqrt(2.0 / hidden)).astype(np.float32)
    p["fc7/b"] = (rs.randn(hidden) * 0.01).astype(np.float32)
    p["cls_score/W"] = (rs.randn(ncls, hidden) * 0.05).astype(np.float32)
    p["cls_score/b"] = (rs.randn(ncls) * 0.01).astype(np.float32)
    p["bbox_pred/W"] = (rs.randn(4 * ncls, hidden) * 0.02).astype(np.float32)
    p["bbox_pred/b"] = (rs.randn(4 * ncls) * 0.01).astype(np.float32)
    return p


def check_rcnn_step(rt, model, params, layers, x, gt, info, feat_stride, seed=0, conv_math="mfma"):
    """Stage-2 step vs the oracle: the device's own proposals, ProposalTargetLayer sample and dropout masks are handed to the
    oracle's autograd restatement; loss and every trunk / head gradient within 1e-3 relative."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.train import RCNNTrainer
    model.rcnn_train = True
    tr = RCNNTrainer(model, conv_math=conv_math)
    # dropout masks fixed up front (the proposal count is not known before the forward: draw for the capacity, slice below)
    rs = np.random.RandomState(seed)
    cap = model.RPN.proposal_layer.TEST_RPN_POST_NMS_TOP_N
    h6, h7 = int(model.fc6.W.shape[0]), int(model.fc7.W.shape[0])
    m6 = ((rs.rand(cap, h6) >= 0.5) * 2.0).astype(np.float32)
    m7 = ((rs.rand(cap, h7) >= 0.5) * 2.0).astype(np.float32)

    class Masks(object):                      # sliced to the actual RoI count on first use
        def __getitem__(self, i):
            return (m6, m7)[i][:self.n]
    # one un-timed forward to learn n (proposals are deterministic)
    feat = model.trunk(Variable(x))
    _, _, prob, bbox = model.RPN.heads(feat, want_score=False)
    n = int(rt.mem.to_numpy(model.RPN.proposal_layer.forward_device(prob, bbox, int(info[0][0]), int(info[0][1]))[2])[0])
    assert n >= 8, n
    masks = (m6[:n], m7[:n])
    np.random.seed(seed + 1)
    out = tr.forward_backward(Variable(x), Variable(info), Variable(gt), masks=masks)
    assert out["n_rois"] == n
    keep = rt.mem.to_numpy(out["keep_inds"])
    # the proposals the STEP pooled (not those of the inference-form forward above: the two forwards' fused conv + pool launches may run different decompositions,
    # so conv5_3 -- and with it a near-tied proposal -- can differ in the last bits; round 6 found 1431 head-ReLU "flips" that were RoIs of another forward)
    rois = rt.mem.to_numpy(out["rois"])[:n]
    np.random.seed(seed + 1)
    use_gt, ext, keep2 = O.proposal_target_layer(rois, gt)
    assert np.array_equal(keep, keep2)
    names = [l if l == "pool" else l[0] for l in layers]
    want_loss, want = O.rcnn_train_grads(params, x, rois, keep, use_gt[:, -1].astype(np.int64), ext, masks[0], masks[1], layers=names,
                                         spatial_scale=1.0 / feat_stride)
    l = tr.losses_host(out)
    assert abs(l["loss_rcnn"] - want_loss) <= 1e-4 * abs(want_loss), (l, want_loss)
    got = tr.grads_chainer_layout()
    # The head's backward runs on the kept rows only (their gradient rows are the only non-zero ones): the same step with every one of the n rows
    # (FRCNN_RCNN_BWD_ROWS=all, the zero-padded form) must give the same gradients up to the grouping of the sums.
    if 0 < len(keep) < n:
        from chainer_faster_rcnn_amd import tuning
        g_kept = rt.mem.to_numpy(tr.G).copy()
        with tuning.override(FRCNN_RCNN_BWD_ROWS="all"):
            np.random.seed(seed + 1)
            out_all = tr.forward_backward(Variable(x), Variable(info), Variable(gt), masks=masks)
        assert np.array_equal(rt.mem.to_numpy(out_all["keep_inds"]), keep)
        g_all = rt.mem.to_numpy(tr.G)
        for name, sg in sorted(tr.seg.items()):
            a, b = g_kept[sg.offset:sg.offset + sg.size], g_all[sg.offset:sg.offset + sg.size]
            assert np.abs(a - b).max() <= 2e-5 * max(float(np.abs(b).max()), 1e-12), (name, float(np.abs(a - b).max()), float(np.abs(b).max()))
        print("RCNN_BWD_ROWS kept %d of %d rows: gradients equal to the all-rows form within 2e-5" % (len(keep), n))
    worst = 0.0
    if x.shape[2] * x.shape[3] >= 200000:
        # Full size (600 x 1000), as for the RPN step (check_vgg_step): a float64 arbiter instead of a relaxed bar.  The oracle's autograd runs once more
        # in float64 (same fp32 parameters, image, RoIs, sample, masks); per gradient the device must be within 5e-3 of it under the device's own head ReLU
        # decisions (asserted below); the tighter figure max(1e-3, 2 x the distance of torch's own fp32 pass from that float64 result) is reported per
        # gradient (PARITY_EXCEED) -- two fp32 passes through 17 layers take a handful of different ReLU / max-pool / RoI arg-max decisions, and how far
        # that moves a gradient is MEASURED on the reference implementation.
        loss64, want64 = O.rcnn_train_grads(params, x, rois, keep, use_gt[:, -1].astype(np.int64), ext, masks[0], masks[1], layers=names,
                                            spatial_scale=1.0 / feat_stride, float64=True)
        assert abs(l["loss_rcnn"] - loss64) <= 1e-4 * abs(loss64), (l, loss64)
        # ... and once more in float64 with the DEVICE's fc6 / fc7 ReLU decisions imposed (read off its activations): of 1.2 M fc6 pre-activations a
        # couple sit within fp32 summation noise (K = 25088) of zero, and ONE unit taking the other branch moves fc6's bias gradient by that unit's whole
        # upstream gradient (measured: 1.6e-2 of the largest entry) -- decision noise, not arithmetic.  The flips are counted and must be a handful.
        a6, a7 = [rt.mem.to_numpy(a) for a in out["head_acts"]]
        _, want64h, flips = O.rcnn_train_grads(params, x, rois, keep, use_gt[:, -1].astype(np.int64), ext, masks[0], masks[1], layers=names,
                                               spatial_scale=1.0 / feat_stride, float64=True, head_relu=(a6 > 0, a7 > 0))
        # ... and a third time with EVERY discrete decision of the device's forward pass imposed (round 6): the trunk's ReLU signs and pool winners -- read off the
        # arg-max bytes of its fused conv + ReLU + pool launches, or off the maps of the two-launch form -- and the arg-max cell of every RoI bin.  What that pass
        # returns is the exact float64 gradient of the function the device evaluated: the device's ARITHMETIC is judged against it with a tight bar, and the
        # distance of the free-decision columns from it is decision noise by construction (the flips are counted per site).
        from chainer_faster_rcnn_amd.train import _PoolArg
        linp = out["layer_inputs"]
        dec = []
        for i, nme in enumerate(names):
            if nme == "pool":
                ent = linp[i]
                if isinstance(ent, _PoolArg):
                    by = rt.mem.to_numpy(ent.idx)
                    C_, OH, OW = by.shape
                    cell = (by & 3).astype(np.int64)
                    winner = (2 * np.arange(OH)[None, :, None] + (cell >> 1)) * ent.W + 2 * np.arange(OW)[None, None, :] + (cell & 1)
                    dec.append((winner, ((by >> 2) & 1).astype(bool)))
                else:
                    import torch
                    m = torch.from_numpy(rt.mem.to_numpy(ent).reshape(1, *ent.shape[-3:]))
                    val, idx = torch.nn.functional.max_pool2d(m, 2, 2, ceil_mode=True, return_indices=True)
                    dec.append((idx[0].numpy(), (val[0] > 0).numpy()))
            elif i + 1 < len(names) and names[i + 1] == "pool":
                dec.append(None)
            else:
                nxt = rt.mem.to_numpy(linp[i + 1])
                dec.append(nxt.reshape(nxt.shape[-3:]) > 0)
        am_dev = rt.mem.to_numpy(out["roi_argmax"]).reshape(n, -1, 7, 7)
        _, want64d, site_flips = O.rcnn_train_grads(params, x, rois, keep, use_gt[:, -1].astype(np.int64), ext, masks[0], masks[1], layers=names,
                                                    spatial_scale=1.0 / feat_stride, float64=True, head_relu=(a6 > 0, a7 > 0), trunk_decisions=dec, roi_argmax=am_dev)
        table = {}
        for k in sorted(want):
            w64 = want64[k]
            scale = max(float(np.abs(w64).max()), 1e-12)
            e_dev = float(np.abs(got[k].astype(np.float64) - w64).max() / scale)
            e_t32 = float(np.abs(want[k].astype(np.float64) - w64).max() / scale)
            e_giv = float(np.abs(got[k].astype(np.float64) - want64h[k]).max() / max(float(np.abs(want64h[k]).max()), 1e-12))
            e_all = float(np.abs(got[k].astype(np.float64) - want64d[k]).max() / max(float(np.abs(want64d[k]).max()), 1e-12))
            table[k] = {"device_vs_f64": float("%.3g" % e_dev), "torch_fp32_vs_f64": float("%.3g" % e_t32), "device_vs_f64_given_device_head_relu": float("%.3g" % e_giv),
                        "device_vs_f64_given_all_device_decisions": float("%.3g" % e_all)}
            worst = max(worst, e_all)
        print("\nPARITY_SITE_FLIPS rcnn_train_%dx%d %s" % (x.shape[2], x.shape[3], json_dumps(site_flips)))
        tag = "rcnn_train_%dx%d%s" % (x.shape[2], x.shape[3], "_split_products" if conv_math == "split" else "")
        print("\nPARITY_TABLE %s %s" % (tag, json_dumps(table)))
        print("PARITY_FLIPS %s %s" % (tag, json_dumps({"fc6_fc7_relu_decisions_that_differ_from_the_float64_pass": int(flips), "of": int(a6.size + a7.size)})))
        assert flips <= 16, flips
        trunk_flips = int(sum(v for k_, v in site_flips.items() if k_ != "fc6_fc7_relu"))
        assert trunk_flips <= 400, site_flips                       # (of ~40 M ReLU signs, ~2 M pool windows, ~7 M RoI bins: a handful each)
        beyond = []
        for k, row in sorted(table.items()):
            # (1) the device's ARITHMETIC: against the float64 gradient of the function it evaluated (all of its decisions imposed) -- no decision noise left, a
            #     tight bar (measured: <= 1.1e-6 on every gradient at 450 x 642, where the free-decision columns reach 6e-3)
            assert row["device_vs_f64_given_all_device_decisions"] <= 1e-4, (k, row)
            # (2) with the trunk's ReLU / max-pool / RoI arg-max decisions left free: REPORTED against max(1e-3, 2 x torch's own fp32 distance); whatever exceeds
            #     5e-3 must come with counted flips (by (1) it is decision noise), and 2e-2 bounds it
            e = row["device_vs_f64_given_device_head_relu"]
            assert e <= 5e-3 or (trunk_flips > 0 and e <= 2e-2), (k, row, site_flips)
            if e > max(1e-3, 2.0 * row["torch_fp32_vs_f64"]):
                beyond.append((k, e, row["torch_fp32_vs_f64"]))
            if flips == 0 and trunk_flips == 0:
                assert row["device_vs_f64"] <= 1e-4, (k, row)
        print("PARITY_EXCEED %s %s" % (tag, json_dumps({"gradients_beyond_max(1e-3, 2 x torch_fp32_vs_f64)_but_within_5e-3": beyond})))
        # ADVICE r05 (docstring vs assertion): what is ASSERTED is (1) above -- 1e-4 against the float64 gradient under ALL of the device's decisions -- and the
        # flip-conditioned caps of (2).  The max(1e-3, 2 x torch) figure is a REPORT: how many gradients land beyond it depends on which near-ties the kernels'
        # summation order happens to decide -- 1 entry with round 5's convolution picks, 17 (all <= 3.5e-3) with round 6's, 6e-3 at 450 x 642 (17 RoI arg-max flips).
    else:
        for k in sorted(want):
            scale = max(np.abs(want[k]).max(), 1e-8)
            err = np.abs(got[k] - want[k]).max() / scale
            worst = max(worst, err)
            assert err <= 1e-3, (k, err, scale)
    # update: same arithmetic as the oracle's MomentumSGD + WeightDecay
    w0, g = rt.mem.to_numpy(tr.W), rt.mem.to_numpy(tr.G)
    tr.update()
    w1, _ = O.momentum_sgd_wd(w0, g, np.zeros_like(w0))
    assert np.array_equal(rt.mem.to_numpy(tr.W), w1)
    return l, worst


def check_small_rcnn_step(rt, seed=0, im_h=48, im_w=64, conv_math="mfma"):
    rs = np.random.RandomState(seed)
    params = small_params()
    params.update(small_head_params(rs))
    x = rs.randn(1, 3, im_h, im_w).astype(np.float32)
    gt = P.gt_case(rs, 3, im_h, im_w)
    gt[0, :, 2] = np.minimum(gt[0, :, 0] + rs.uniform(10, 30, 3), im_w - 1)
    gt[0, :, 3] = np.minimum(gt[0, :, 1] + rs.uniform(10, 30, 3), im_h - 1)
    info = np.array([[im_h, im_w]], dtype=np.int32)
    model = build_small(rt, params)
    for n in ("fc6", "fc7", "cls_score", "bbox_pred"):
        getattr(model, n).set(params[n + "/W"], params[n + "/b"])
    model.RPN.proposal_layer.RPN_MIN_SIZE = 4
    model.RPN.proposal_layer._min_size = 4
    return check_rcnn_step(rt, model, params, SMALL_LAYERS, x, gt, info, 4, seed, conv_math=conv_math)


def check_vgg_rcnn_step(rt, im_h=160, im_w=224, seed=0, conv_math="mfma"):
    """Stage-2 step of the real VGG-16 FasterRCNN (GPU suite)."""
    from chainer_faster_rcnn_amd import synthetic
    from chainer_faster_rcnn_amd.models import FasterRCNN
    from chainer_faster_rcnn_amd.models.vgg16 import LAYERS
    rs = np.random.RandomState(seed)
    params = synthetic.params(seed=1)
    x = synthetic.image(seed=4, h=im_h, w=im_w)
    gt = P.gt_case(rs, 4, im_h, im_w)
    info = np.array([[im_h, im_w]], dtype=np.int32)
    model = FasterRCNN(runtime=rt)
    model.load_params(params)
    return check_rcnn_step(rt, model, params, LAYERS, x, gt, info, 16, seed, conv_math=conv_math)


def check_trainers_across_image_sizes(rt, sizes=((40, 56), (56, 40))):
    """train_rpn.py / train_rcnn.py feed a differently sized image every iteration (VOC: 600 x 800, 800 x 600, 600 x 901 ...): a trainer that has stepped on other
    sizes must give, on the next image, exactly what a NEW trainer gives from the same parameters -- workspaces, kept maps and cached state are functions of the
    current image only.  Both trainers on the narrow trunk; losses bit for bit, the RPN step's gradients bit for bit (every reduction of that step has a fixed
    order), the stage-2 step's to rounding (its RoI-pooling backward adds with float atomics)."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.train import RPNTrainer, RCNNTrainer
    rs = np.random.RandomState(5)
    params = small_params()
    params.update(small_head_params(rs))
    images = []
    for (h, w) in sizes:
        x = rs.randn(1, 3, h, w).astype(np.float32)
        gt = P.gt_case(rs, 3, h, w)
        gt[0, :, 2] = np.minimum(gt[0, :, 0] + rs.uniform(10, 30, 3), w - 1)
        gt[0, :, 3] = np.minimum(gt[0, :, 1] + rs.uniform(10, 30, 3), h - 1)
        images.append((x, gt, np.array([[h, w]], dtype=np.int32)))

    def rpn_trainer():
        return RPNTrainer(build_small(rt, params))

    def rcnn_trainer():
        model = build_small(rt, params)
        for n in ("fc6", "fc7", "cls_score", "bbox_pred"):
            getattr(model, n).set(params[n + "/W"], params[n + "/b"])
        model.RPN.proposal_layer.RPN_MIN_SIZE = 4
        model.RPN.proposal_layer._min_size = 4
        model.rcnn_train = True
        return RCNNTrainer(model)                                  # masks from NumPy's global stream, re-seeded before every step below

    # one long-lived instance (no update between the steps: the parameters stay a new trainer's) against one new instance per image
    for tag, make in (("rpn", rpn_trainer), ("rcnn", rcnn_trainer)):
        old = make()
        for i, (x, gt, info) in enumerate(images):
            np.random.seed(100 + i)
            lo = old.losses_host(old.forward_backward(Variable(x), Variable(info), Variable(gt)))
            g_old = rt.mem.to_numpy(old.G).copy()
            new = make()
            np.random.seed(100 + i)
            ln = new.losses_host(new.forward_backward(Variable(x), Variable(info), Variable(gt)))
            assert lo == ln, (tag, i, lo, ln)
            g_new = rt.mem.to_numpy(new.G)
            if tag == "rpn":
                assert np.array_equal(g_old, g_new), (tag, i, float(np.abs(g_old - g_new).max()))
            else:
                # the stage-2 backward pass is NOT bit-reproducible from run to run on the GPU: roi_pool_bwd_runs_kernel adds the 300 RoIs' bin gradients into its
                # LDS planes with float atomics, in whatever order its sixteen waves arrive (include/frcnn_hip.h states it; the reference's CPU loop is RoI-major) --
                # rounding-level differences (measured 1.4e-6 of the largest entry between two NEW trainers on the same image), nothing size-dependent
                assert np.abs(g_old - g_new).max() <= 2e-5 * np.abs(g_new).max(), (tag, i, float(np.abs(g_old - g_new).max()))
    return len(images)
