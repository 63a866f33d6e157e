"""The parity tests proper: libfrcnn_hip.so on a real MI355X, through the C ABI, against the oracle and the
golden vectors the reference produced.  Run with `-m gpu` on the GPU box."""
import numpy as np
import pytest

import parity_cases as P
from chainer_faster_rcnn_amd import tuning

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import chainer_faster_rcnn_amd as pkg
    return pkg.runtime.default_runtime()          # raises (no fallback) if the library or the GPU is missing


def test_library_is_the_device_build(rt):
    assert rt.lib.frcnn_device_count() >= 1 and rt.lib.frcnn_abi_version() == 24


def test_nms_golden(rt):
    P.check_nms_golden(rt)


def test_nms_edges(rt):
    P.check_nms_edges(rt)


def test_gpu_nms_reference_ffi(rt):
    """The reference's one C FFI, `_nms` (models/gpu_nms.hpp:9-10), exported with its exact signature."""
    import torch
    before = torch.cuda.current_device()
    P.check_gpu_nms_ffi(rt)
    assert torch.cuda.current_device() == before


def test_nms_random(rt):
    P.check_nms_random(rt, n=3000, seeds=(0, 1, 2))


def test_nms_above_single_wave_capacity(rt):
    """> 16384 boxes: the multi-wave scan kernel (LDS bitmap) instead of the register-resident single-wave one."""
    P.check_nms_random(rt, n=17000, seeds=(0,), thrs=(0.5,))


def test_nms_staged(rt):
    P.check_nms_staged(rt, n=5000, seeds=(0, 1))


def test_nms_chains(rt):
    P.check_nms_chains(rt, n=1500)


def test_nms_staged_strided_tail(rt):
    P.check_nms_staged_strided_tail(rt)


def test_nms_batched(rt):
    P.check_nms_batched(rt, groups=20, n=300)


@pytest.mark.parametrize("case", ["proposal_14x14_train_rand", "proposal_38x63_test", "proposal_38x63_test_HH",
                                  "proposal_38x63_train", "proposal_38x63_cfg4_1000_300", "proposal_37x50_test"])
def test_proposals_golden(rt, case):
    P.check_proposals_golden(rt, case)


def test_proposals_edge_goldens(rt):
    """NaN scores of either sign, +-inf scores, NaN / +-inf deltas, exp overflow: reference-generated fixtures."""
    P.check_proposals_edge_goldens(rt)


def test_proposals_tied_scores(rt):
    """Equal scores: ascending anchor index, in the sort and in NMS's visiting order (the oracle's tie_rule="ascending_index"); everything bit for bit."""
    P.check_proposals_tied_scores(rt)
    P.check_proposals_tied_scores(rt, 38, 63, seed=1)


def test_nms_edge_goldens(rt):
    P.check_nms_edge_goldens(rt)


def test_proposals_repeatable(rt):
    """idempotence: the same inputs give the same RoIs on every call (workspace reuse, no stale state)."""
    G = P.g("proposal_38x63_test")
    from oracle import frcnn_oracle as O
    a = [rt.mem.to_numpy(v) for v in rt.proposals(P.dev(rt, G["rpn_cls_prob"][0]), P.dev(rt, G["rpn_bbox_pred"][0]),
                                                  O.generate_anchors(), 16, 600, 1000, 16.0, 6000, 300, 0.7)]
    for _ in range(3):
        b = [rt.mem.to_numpy(v) for v in rt.proposals(P.dev(rt, G["rpn_cls_prob"][0]), P.dev(rt, G["rpn_bbox_pred"][0]),
                                                      O.generate_anchors(), 16, 600, 1000, 16.0, 6000, 300, 0.7)]
        assert all(np.array_equal(u, v) for u, v in zip(a, b))


def test_roi_pool_extreme_rois(rt):
    P.check_roi_pool_extreme_rois(rt)


def test_roi_pool_cells_kernel(rt):
    P.check_roi_pool_cells(rt)


def test_roi_pool_cells_batches(rt):
    P.check_roi_pool_cells_batches(rt)


def test_roi_pool_kernels_agree_at_full_size(rt, monkeypatch):
    """The cell-major kernel (default) and the plane kernel (FRCNN_ROI_KERNEL=planes) give the same 300 x 512 x 7 x 7 bits."""
    rs = np.random.RandomState(3)
    x, rois = P.roi_case(rs, 300, 512, 38, 63)
    a = P.host(rt, rt.roi_pool_fwd(P.dev(rt, x[0]), P.dev(rt, rois), 7, 7, 0.0625))
    tuning.set("FRCNN_ROI_KERNEL", "planes")
    b = P.host(rt, rt.roi_pool_fwd(P.dev(rt, x[0]), P.dev(rt, rois), 7, 7, 0.0625))
    assert np.array_equal(a, b)


def test_roi_pool_full_size(rt):
    P.check_roi_pool(rt, R=300, C=512, H=38, W=63)        # BASELINE config: 300 x 512 x 7 x 7
    P.check_roi_pool(rt, R=7, C=64, H=19, W=32, seed=2)   # VEC=1 path


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 8, 10, 11, 104, 105, 108, 110, 205, 210, -1])
def test_conv3x3_cfgs(rt, cfg):
    P.check_conv3x3(rt, 64, 128, 75, 125, cfg=cfg)


@pytest.mark.parametrize("cin,cout,h,w", [(3, 64, 120, 200), (64, 64, 60, 100), (128, 256, 75, 125), (512, 512, 38, 63),
                                          (256, 512, 19, 32)])
def test_conv3x3_vgg_shapes(rt, cin, cout, h, w):
    P.check_conv3x3(rt, cin, cout, h, w)


def test_conv3x3_streamk_is_deterministic(rt):
    """Split tiles are summed in piece order by whichever workgroup arrives last: repeated launches must agree
    bit for bit, and with the whole-tile schedule to rounding."""
    import numpy as np
    rs = np.random.RandomState(0)
    x = P.dev(rt, rs.randn(1, 256, 150, 250).astype(np.float32))
    w = P.dev(rt, (rs.randn(256, 256, 3, 3) * 0.02).astype(np.float32))
    b = P.dev(rt, rs.randn(256).astype(np.float32))
    wp = rt.pack_conv3x3_w(w)
    ref = P.host(rt, rt.conv3x3(x, wp, b, cfg=10))
    outs = [P.host(rt, rt.conv3x3(x, wp, b, cfg=210)) for _ in range(4)]
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    assert np.abs(outs[0] - ref).max() <= 1e-4 * np.abs(ref).max()


def test_maxpool(rt):
    P.check_maxpool(rt, 64, 75, 125)
    P.check_maxpool(rt, 8, 600, 1000)


def test_rpn_heads(rt, monkeypatch):
    P.check_rpn_heads_forms(rt, monkeypatch, Cmid=512, H=38, W=63)
    P.check_rpn_heads_forms(rt, monkeypatch, Cmid=272, H=19, W=32, A=3, seed=1)


def test_linear(rt):
    P.check_linear(rt, 300, 4096, 25088, True)       # fc6
    P.check_linear(rt, 300, 4096, 4096, True)        # fc7
    P.check_linear(rt, 300, 21, 4096, False)         # cls_score
    P.check_linear(rt, 300, 84, 4096, False)         # bbox_pred
    P.check_linear(rt, 17, 33, 100, False)
    P.check_linear(rt, 512, 1024, 128, False, bias=False)     # the weight-gradient shape (short contraction, one slab: written straight to y)
    P.check_linear(rt, 84, 416, 128, False, bias=False)
    P.check_linear(rt, 128, 256, 4096, False, bias=False)     # NULL bias with split-K slabs (the combine pass without a bias term)
    P.check_linear(rt, 40, 64, 256, True, bias=False)


def test_head_decode(rt):
    P.check_head_decode(rt, R=300)


def test_models_surface_and_end_to_end(rt):
    """The reference's call surface: FasterRCNN(trunk_class=VGG16Prev)(Variable(img), Variable(img_info))."""
    from chainer_faster_rcnn_amd import synthetic
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.models import FasterRCNN, VGG16Prev, cpu_nms
    from oracle import frcnn_oracle as O
    params = synthetic.params(seed=1)
    h, w = 224, 320
    x = synthetic.image(seed=5, h=h, w=w)
    info = np.array([[h, w]], dtype=np.int32)
    model = FasterRCNN(trunk_class=VGG16Prev, runtime=rt)
    model.rcnn_train = False
    model.rpn_train = False
    model.load_params(params)
    cls_score, bbox_pred = model(Variable(x, volatile=True), Variable(info))
    cls, boxes, dbg = O.faster_rcnn_forward(params, x, info, return_debug=True)
    out = model.forward_device(rt.mem.from_numpy(x), h, w, keep=True)
    feat = rt.mem.to_numpy(out["feat"])
    assert np.abs(feat - dbg["feat"]).max() / np.abs(dbg["feat"]).max() < 1e-3          # fp32 features, 1e-3 rel
    # proposals: exact given the device's own RPN maps
    p2, s2, d2 = O.proposal_layer(rt.mem.to_numpy(out["rpn_cls_prob"]), rt.mem.to_numpy(out["rpn_bbox_pred"]), info,
                                  train=False, return_debug=True)
    n = int(rt.mem.to_numpy(out["n_out"])[0])
    assert n == len(p2) and np.allclose(rt.mem.to_numpy(out["rois"])[:n], p2, rtol=5e-7, atol=1e-4)
    assert cls_score.data.shape == (n, 21) and tuple(bbox_pred.shape) == (n, 84)
    # head: compare on the device's own RoIs
    rois = rt.mem.to_numpy(out["rois"])[:n]
    pool5 = O.roi_pooling_2d(feat, np.concatenate([np.zeros((n, 1), np.float32), rois], 1), 7, 7, 1 / 16.)
    assert np.array_equal(rt.mem.to_numpy(out["pool5"])[:n], pool5)
    cp, pb, _ = O.rcnn_head(params, pool5, rois, info)
    assert np.allclose(rt.mem.to_numpy(out["cls_prob"])[:n], cp, rtol=1e-3, atol=1e-5)
    assert np.allclose(rt.mem.to_numpy(out["pred_boxes"])[:n], pb, rtol=1e-3, atol=1e-2)
    # forward.py:48-58 style post-processing through the cpu_nms-compatible entry point
    dets = np.hstack([pb[:, 4:8], cp[:, 1:2]]).astype(np.float32)
    assert cpu_nms(dets, 0.3, runtime=rt) == O.cpu_nms(dets, 0.3)


# ---- RPN training step (csrc/train.hip)
def test_bbox_overlaps(rt):
    P.check_bbox_overlaps(rt, N=8151, K=20)


@pytest.mark.parametrize("fh,fw,im_h,im_w,G", [(14, 14, 224, 224, 3), (38, 63, 600, 1000, 8), (38, 63, 600, 1000, 1), (37, 50, 600, 800, 40)])
def test_anchor_target(rt, fh, fw, im_h, im_w, G):
    n = P.check_anchor_target(rt, fh, fw, im_h, im_w, G, seed=G)
    if (fh, fw) == (38, 63):
        assert n == 8151                                            # SURVEY.md 8a-17 [probe]


def test_rpn_loss(rt):
    P.check_rpn_loss(rt)
    P.check_rpn_loss(rt, fh=38, fw=63, im=600, G=6, seed=3)


@pytest.mark.parametrize("cin,cout,h,w,ks", [(64, 64, 60, 100, 3), (3, 64, 120, 200, 3), (128, 256, 38, 63, 3), (512, 512, 19, 32, 3),
                                             (512, 64, 38, 63, 1)])
def test_conv_backward(rt, cin, cout, h, w, ks):
    P.check_conv_backward(rt, cin, cout, h, w, ksize=ks)


@pytest.mark.parametrize("env", [{"FRCNN_WGRAD_DB": "1"}, {"FRCNN_WGRAD_DB": "0"}])
def test_conv_wgrad_forms(rt, monkeypatch, env):
    """both forms of the 3x3 weight-gradient kernel (single- / double-buffered) forced on every layer against the oracle, incl. a map whose width is no multiple of 4 and conv1_1's three input channels"""
    for k, v in env.items():
        tuning.set(k, v)
    P.check_conv_backward(rt, 128, 128, 75, 125)
    P.check_conv_backward(rt, 3, 64, 120, 200, seed=1)
    P.check_conv_backward(rt, 256, 64, 38, 63, seed=2)


def test_conv_relu_pool_train(rt):
    P.check_conv_relu_pool_train(rt, 64, 64, 120, 200)
    P.check_conv_relu_pool_train(rt, 128, 128, 75, 125, seed=1)
    P.check_conv_relu_pool_train(rt, 512, 512, 37, 63, seed=2)


def test_conv_dgrad_unpool(rt):
    P.check_conv_dgrad_unpool(rt, 128, 64, 120, 200)
    P.check_conv_dgrad_unpool(rt, 256, 128, 75, 125, seed=1)
    P.check_conv_dgrad_unpool(rt, 512, 512, 37, 63, seed=2)


def test_pack_dgrad_many(rt):
    P.check_pack_dgrad_many(rt)


def test_maxpool_bwd(rt):
    P.check_maxpool_bwd(rt, 64, 75, 125)
    P.check_maxpool_bwd(rt, 8, 600, 1000)


def test_sgd(rt):
    P.check_sgd(rt, n=17100003)


def test_rpn_train_step_small(rt):
    import train_cases as T
    T.check_small_step(rt)


def test_rccl_single_rank_training_step(rt):
    """RCCL really executes on the one GPU a test box has: a process group of ONE rank ("nccl" = RCCL on ROCm), the trainer's three
    bucketed async all-reduces on the collective's stream and their stream waits (TorchComm(force_single_rank=True)) -- and the step
    equals the same step without a communicator bit for bit (a one-rank sum is the identity).  train_rpn.py:162-174."""
    import os
    import torch
    import torch.distributed as dist
    import train_cases as T
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.train import RPNTrainer, TorchComm
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(29500 + os.getpid() % 2000)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        assert dist.get_backend() == "nccl"
        params = T.small_params()
        rs = np.random.RandomState(3)
        x = rs.randn(1, 3, 40, 56).astype(np.float32)
        gt = P.gt_case(rs, 2, 40, 56)
        gt[0, :, 2] = np.minimum(gt[0, :, 0] + 20, 55); gt[0, :, 3] = np.minimum(gt[0, :, 1] + 20, 39)
        info = np.array([[40, 56]], dtype=np.int32)
        got = []
        for comm in (TorchComm(force_single_rank=True), None):
            tr = RPNTrainer(T.build_small(rt, params), comm=comm)
            for step in range(2):
                np.random.seed(11 + step)
                tr.step(Variable(x), Variable(info), Variable(gt))
            if comm is not None:
                assert comm.active and len(tr.buckets) == 3
            got.append(rt.mem.to_numpy(tr.W).copy())
        assert np.array_equal(got[0], got[1])
        # a plain blocking all-reduce of a device buffer through the same communicator
        buf = rt.mem.from_numpy(np.arange(1000, dtype=np.float32))
        TorchComm(force_single_rank=True).all_reduce_sum(buf)
        assert np.array_equal(rt.mem.to_numpy(buf), np.arange(1000, dtype=np.float32))
    finally:
        if created:
            dist.destroy_process_group()


def test_rpn_train_step_vgg16(rt):
    import train_cases as T
    losses, worst, flipped = T.check_vgg_step(rt)
    assert losses["rpn_loss"] > 0 and (worst <= 1e-3 or flipped)


# ---- ResNet-101 trunk and BASELINE config 4 (ResNet-101 backbone, 1000 pre-NMS / 300 post-NMS proposals)
def test_resnet_pieces(rt):
    P.check_resnet_pieces(rt)


def test_resnet101_trunk(rt):
    err = P.check_resnet(rt, blocks=(3, 4, 23, 3), im_h=224, im_w=320)
    assert err < 1e-3


def test_faster_rcnn_resnet101_config4(rt):
    """FasterRCNN(trunk_class=ResNet101, rpn_in_ch=2048, feat_stride=32) with ProposalLayer's class constants overridden to
    1000 / 300 (config 4): trunk vs the oracle's explicit-BN restatement, proposals exact given the device's RPN maps, RoI
    pooling (scale 1/32) exact, head within 1e-3."""
    from chainer_faster_rcnn_amd import synthetic
    from chainer_faster_rcnn_amd.models import FasterRCNN, ResNet101
    from oracle import frcnn_oracle as O
    h, w = 320, 480
    params = synthetic.resnet_params(101, seed=2)
    rs = np.random.RandomState(3)
    head = synthetic.params(seed=1, rpn_ch=512, roi_feat=2048 * 49)
    for k in ("fc6", "fc7", "cls_score", "bbox_pred"):
        params[k + "/W"], params[k + "/b"] = head[k + "/W"], head[k + "/b"]
    params["RPN/rpn_conv_3x3/W"] = (rs.randn(512, 2048, 3, 3) * 0.01).astype(np.float32)
    params["RPN/rpn_conv_3x3/b"] = np.zeros(512, np.float32)
    for k in ("rpn_cls_score", "rpn_bbox_pred"):
        params["RPN/%s/W" % k], params["RPN/%s/b" % k] = head["RPN/%s/W" % k], head["RPN/%s/b" % k]
    model = FasterRCNN(trunk_class=ResNet101, rpn_in_ch=2048, rpn_mid_ch=512, feat_stride=32, runtime=rt)
    model.load_params(params)
    model.RPN.proposal_layer._pre_nms_top_n, model.RPN.proposal_layer._post_nms_top_n = 1000, 300
    x = synthetic.image(seed=6, h=h, w=w) / 64.0
    info = np.array([[h, w]], dtype=np.int32)
    out = model.forward_device(rt.mem.from_numpy(x), h, w, keep=True)
    feat = rt.mem.to_numpy(out["feat"])
    want_feat = O.resnet_forward(params, x)
    assert feat.shape == want_feat.shape == (1, 2048, 10, 15)
    assert np.abs(feat - want_feat).max() / np.abs(want_feat).max() < 1e-3
    n = int(rt.mem.to_numpy(out["n_out"])[0])
    p2, s2 = O.proposal_layer(rt.mem.to_numpy(out["rpn_cls_prob"]), rt.mem.to_numpy(out["rpn_bbox_pred"]), info, train=False,
                              feat_stride=32, pre_nms_top_n=1000, post_nms_top_n=300)
    assert n == len(p2) and np.allclose(rt.mem.to_numpy(out["rois"])[:n], p2, rtol=5e-7, atol=1e-4)
    rois = rt.mem.to_numpy(out["rois"])[:n]
    pool5 = O.roi_pooling_2d(feat, np.concatenate([np.zeros((n, 1), np.float32), rois], 1), 7, 7, 1 / 32.)
    assert np.array_equal(rt.mem.to_numpy(out["pool5"])[:n], pool5)
    cp, pb, _ = O.rcnn_head(params, pool5, rois, info)
    assert np.allclose(rt.mem.to_numpy(out["cls_prob"])[:n], cp, rtol=1e-3, atol=1e-5)
    assert np.allclose(rt.mem.to_numpy(out["pred_boxes"])[:n], pb, rtol=1e-3, atol=1e-2)


# ---- bf16 convolution stack (BASELINE config 3: bf16 convs / fp32 RoI)
@pytest.mark.parametrize("cin,cout,h,w,ks", [(3, 64, 120, 200, 3), (64, 64, 60, 100, 3), (128, 256, 75, 125, 3), (512, 512, 38, 63, 3),
                                             (512, 54, 38, 63, 1)])
def test_conv_bf16(rt, cin, cout, h, w, ks):
    P.check_conv_bf16(rt, cin, cout, h, w, ksize=ks, relu=(ks == 3))


def test_conv_bf16_pool_fused(rt):
    P.check_conv_bf16_pool(rt, 64, 64, 120, 200)
    P.check_conv_bf16_pool(rt, 256, 512, 75, 125, seed=1)


@pytest.mark.parametrize("cin,cout,h,w", [(64, 64, 600, 1000), (256, 256, 150, 250), (512, 512, 38, 63), (3, 64, 600, 1000)])
def test_conv_bf16_staging_variants_bit_identical(rt, monkeypatch, cin, cout, h, w):
    """Full-size property: the LDS-DMA kernels (every ring depth / tile shape) accumulate in the register-staged kernel's
    order, so all outputs -- plain and pool-fused -- are bit-identical to it at the real VGG layer sizes."""
    rs = np.random.RandomState(11)
    x = rt.bf16_from_nchw(rt.mem.from_numpy(rs.randn(1, cin, h, w).astype(np.float32)))
    wt = rt.bf16_pack_conv_w(rt.mem.from_numpy((rs.randn(cout, cin, 3, 3) * 0.05).astype(np.float32)), 3)
    b = rt.mem.from_numpy(rs.randn(cout).astype(np.float32))
    outs = {}
    for mode in ["0", "-1", "141", "231", "132"]:              # the register-staged kernel, the default rule, and the three ring / tile shapes the rule picks from
        tuning.set("FRCNN_BF16_DMA", mode)
        outs[mode] = (rt.mem.to_numpy(rt.conv_bf16(x, wt, b, cin, cout, 3, relu=True)),
                      rt.mem.to_numpy(rt.conv_bf16(x, wt, b, cin, cout, 3, relu=True, pool=True)))
    for mode, (full, pooled) in outs.items():
        if mode == "-1" and not np.array_equal(full, outs["0"][0]):
            # the default pick of a 38 x 63 launch is strip form C (csrc/conv_bf16_strip.h): the K loop split four ways over the waves of a
            # workgroup, partial accumulators summed in K-way order -- fp32 summation-order noise.  A bf16 output may then land one rounding
            # step off at a tie, and an output whose fp32 pre-activation sits within that noise of zero may cross the ReLU (0 vs ~1e-5):
            # P.check_ksplit_words proves every differing word is one of the two, from the fp32 pre-activations of BOTH kernels.
            assert (h, w) == (38, 63)
            pre = {}
            for m2 in ("0", "-1"):
                tuning.set("FRCNN_BF16_DMA", m2)
                pre[m2] = rt.mem.to_numpy(rt.conv_bf16(x, wt, b, cin, cout, 3, relu=False, out_f32_nchw=True))[0]
            P.check_ksplit_words("staging[%d-%d-%d-%d] full" % (cin, cout, h, w), full, outs["0"][0], pre["-1"], pre["0"], cout)
            # the fused-pool launch of this size is NOT form C (odd tile rows): conv_dma_bf16_kernel's pick, the register-staged kernel's chain
            assert rt.lib.frcnn_conv_bf16_plan(cin, cout, h, w, 3, 0) == 903 and rt.lib.frcnn_conv_bf16_plan(cin, cout, h, w, 3, 2) == 0
            assert np.array_equal(pooled, outs["0"][1])
            continue
        assert np.array_equal(full, outs["0"][0]) and np.array_equal(pooled, outs["0"][1]), mode


@pytest.mark.parametrize("cin,cout,h,w,expect,expect_pooled", [(256, 256, 150, 250, 910, 910),      # conv3_2 / conv3_3 (pooled): form D, direct stores / LDS epilogue
                                                               (512, 512, 75, 125, 910, 910),      # conv4_2 / conv4_3 (pooled)
                                                               (128, 128, 300, 500, 910, 910),     # conv2_2 (pooled): 8 K-chunks, the shortest loop of the rule
                                                               (512, 512, 38, 63, 903, 0),         # conv5_x, rpn_conv: form C; its pooled launch stays on conv_dma_bf16_kernel
                                                               (64, 64, 300, 500, 0, 0)])          # 4 K-chunks: outside the rule
def test_conv_bf16_default_picks_vs_oracle(rt, cin, cout, h, w, expect, expect_pooled):
    """The strip forms at the VGG layer sizes where the default rule really launches them, against the oracle (not against another HIP kernel)."""
    P.check_conv_bf16_default_pick(rt, cin, cout, h, w, expect, expect_pooled)


@pytest.mark.parametrize("h,w,cin,rw", [(600, 1000, 3, None), (75, 101, 3, None), (24, 64, 1, None)])
def test_conv1_pair_bf16(rt, h, w, cin, rw):
    """conv1_1 + conv1_2 + pool1 as one launch (csrc/conv_bf16_pair.hip; the bf16 chain's default first launch): bit for bit the two-launch chain, and
    within one rounding of the oracle -- at the real 600 x 1000 image (1600 tiles on 256 persistent workgroups) and on ragged / odd sizes."""
    P.check_conv1_pair_bf16(rt, h, w, Cin=cin, rw=rw)


@pytest.mark.parametrize("form", [903, 909, 910])
def test_conv_bf16_strip_forms(rt, form):
    """The strip forms (csrc/conv_bf16_strip.h; D -- 910, with 909's epilogue under the fused pool -- and C = 903 are default picks) against conv_dma_bf16_kernel at VGG layer sizes: bit-identical
    where one accumulation chain per output is kept, fp32 summation-order noise for the K-split form (profiles/r03_conv_bf16_strip_micro.txt holds
    the same comparison from the torch-free harness on all ten layer shapes)."""
    P.check_conv_bf16_strip(rt, form, 256, 256, 150, 250, pool=form != 903)
    P.check_conv_bf16_strip(rt, form, 512, 512, 38, 63, seed=1)
    P.check_conv_bf16_strip(rt, form, 128, 54, 75, 125, seed=2)      # ragged: 54 couts of 64, 75 rows = 7.5 tiles of 10


def test_research_forms_are_not_in_the_product_library(rt):
    """The measured-and-not-adopted kernel forms (csrc/conv_bf16_res.h, strip A / B / E / 907 / 908, the unpicked ring depths, the one-wave conv1 pair) are
    compiled into research builds only (-DFRCNN_TUNING_FORMS: scripts/micro, the test emulator); the product library refuses them instead of substituting."""
    rs = np.random.RandomState(0)
    x = rt.bf16_from_nchw(rt.mem.from_numpy(rs.randn(1, 64, 24, 40).astype(np.float32)))
    wt = rt.bf16_pack_conv_w(rt.mem.from_numpy((rs.randn(64, 64, 3, 3) * 0.05).astype(np.float32)), 3)
    b = rt.mem.from_numpy(rs.randn(64).astype(np.float32))
    for mode in ("901", "902", "907", "908", "911", "921", "321", "224"):
        with tuning.override(FRCNN_BF16_DMA=mode):
            if mode[0] == "9":
                assert rt.lib.frcnn_conv_bf16_plan(64, 64, 24, 40, 3, 0) == -1          # the plan query says the same
            with pytest.raises(ValueError):
                rt.conv_bf16(x, wt, b, 64, 64, 3, relu=True)


@pytest.mark.parametrize("split,mode", [("2", None), ("4", None), ("2", "231")])
def test_conv_bf16_split_k(rt, monkeypatch, split, mode):
    tuning.set("FRCNN_BF16_SPLIT", split)
    if mode:
        tuning.set("FRCNN_BF16_DMA", mode)
    P.check_conv_bf16(rt, 512, 512, 38, 63)              # the shape split-K exists for: 160 tiles, 32 chunks
    P.check_conv_bf16_pool(rt, 256, 512, 75, 125, seed=1)


def test_maxpool_bf16(rt):
    P.check_maxpool_bf16(rt, 64, 75, 125)
    P.check_maxpool_bf16(rt, 128, 300, 500)


def test_vgg16_bf16_forward(rt):
    err = P.check_vgg_bf16_forward(rt, 224, 320)
    assert err < 3e-2


def test_detections_postprocess(rt):
    P.check_detections(rt, R=300)


def test_linear_bf16(rt):
    P.check_linear_bf16(rt, 300, 4096, 25088, True)      # fc6
    P.check_linear_bf16(rt, 300, 4096, 4096, True)       # fc7
    P.check_linear_bf16(rt, 300, 84, 4096, False)        # bbox_pred
    P.check_linear_bf16(rt, 17, 33, 104, False)


def test_f16_instantiation_of_the_16_bit_chain(rt):
    """The fp16 twins of the 16-bit chain (csrc/conv_f16.hip, conv_f16_pair.hip, linear_f16.hip) at the layer sizes where the default rule launches each kernel
    form -- the bf16 parity checks on a runtime in fp16 mode, the oracle fed fp16-rounded operands."""
    r16 = rt.with_half("f16")
    with P.half_format("f16"):
        P.check_conv_bf16(r16, 256, 256, 150, 250)                 # conv3_2: strip form D
        P.check_conv_bf16(r16, 512, 512, 38, 63)                   # conv5_x: strip form C (K split over the waves)
        P.check_conv_bf16(r16, 64, 128, 300, 500)                  # conv2_1: conv_dma_bf16_kernel
        P.check_conv_bf16(r16, 512, 54, 38, 63, ksize=1, relu=False)
        P.check_conv_bf16_pool(r16, 128, 128, 60, 100)
        P.check_maxpool_bf16(r16, 64, 75, 125)
        P.check_conv1_pair_bf16(r16, 120, 200)
        P.check_rpn_heads_bf16(r16, 512, 38, 63)
        P.check_linear_bf16_tiled(r16, 300, 4096, 25088, True)     # fc6
        P.check_linear_bf16_tiled(r16, 300, 116, 4096, False)
        P.check_linear_bf16(r16, 300, 4096, 4096, True)
        P.check_roi_pool_blk_bf16(r16, 300, 512, 38, 63)           # roi_f16.hip at the benchmark size
        P.check_roi_pool(r16, R=40, C=64, H=38, W=63)


def test_vgg16_f16_forward(rt):
    """The whole forward pass with fp16 convolutions and head at 224 x 320: trunk within 4e-3 of the fp32 oracle's feature scale (bf16: 3e-2)."""
    with P.half_format("f16"):
        err = P.check_vgg_bf16_forward(rt, 224, 320, dtype="f16", feat_tol=4e-3, head_tol=3e-3)
    print("PARITY vgg16 f16 forward 224x320: conv5_3 rel err %.2e" % err)


def test_linear_bf16_tiled(rt):
    """The weight-stream kernel on pre-tiled weights (csrc/linear_bf16.hip), the bf16 line's default head."""
    P.check_linear_bf16_tiled(rt, 300, 4096, 25088, True)      # fc6
    P.check_linear_bf16_tiled(rt, 300, 4096, 4096, True)       # fc7
    P.check_linear_bf16_tiled(rt, 300, 116, 4096, False)       # cls_score || bbox_pred, stacked
    P.check_linear_bf16_tiled(rt, 128, 4096, 4096, True, seed=2)   # the stage-2 backward's row count: MT 3
    P.check_linear_bf16_tiled(rt, 17, 33, 96, False)
    P.check_linear_bf16_tiled(rt, 700, 256, 32 * 40, False, seed=3)  # three row blocks


@pytest.mark.parametrize("cin,cout,h,w", [(64, 64, 120, 200), (128, 128, 60, 100), (256, 256, 150, 250), (512, 512, 75, 125)])
def test_conv_relu_pool_fused(rt, cin, cout, h, w):
    P.check_conv_relu_pool(rt, cin, cout, h, w)


def test_img_preprocessing(rt):
    P.check_preprocess(rt, 375, 500)          # a VOC-sized image -> 600 x 800
    P.check_preprocess(rt, 333, 1000, seed=1)


# ---- BASELINE.json full sizes through size-independent properties (the oracle is not needed at these sizes)
def test_conv_full_size_linearity_and_spot_values(rt):
    """conv1_2 at 600x1000 (153.6 MB per tensor): conv is linear in x without bias/ReLU, and individual output pixels equal
    the direct 576-term dot product (float64) at image corners, edges and the interior."""
    rs = np.random.RandomState(0)
    x1 = rs.randn(1, 64, 600, 1000).astype(np.float32)
    x2 = rs.randn(1, 64, 600, 1000).astype(np.float32)
    w = (rs.randn(64, 64, 3, 3) * 0.06).astype(np.float32)
    zero = P.dev(rt, np.zeros(64, np.float32))
    wp = rt.pack_conv3x3_w(P.dev(rt, w))
    y1 = P.host(rt, rt.conv3x3(P.dev(rt, x1), wp, zero, relu=False))
    y2 = P.host(rt, rt.conv3x3(P.dev(rt, x2), wp, zero, relu=False))
    y3 = P.host(rt, rt.conv3x3(P.dev(rt, (2 * x1 - x2)), wp, zero, relu=False))
    assert np.abs(y3 - (2 * y1 - y2)).max() <= 2e-5 * np.abs(y1).max() * 4
    xp = np.pad(x1[0], ((0, 0), (1, 1), (1, 1))).astype(np.float64)
    for (co, py, px) in [(0, 0, 0), (63, 599, 999), (17, 0, 999), (40, 599, 0), (5, 300, 31), (5, 300, 32), (33, 3, 4), (62, 597, 993)]:
        want = float((xp[:, py:py + 3, px:px + 3] * w[co].astype(np.float64)).sum())
        assert abs(y1[0, co, py, px] - want) <= 1e-4 * max(abs(want), 1.0), (co, py, px)


def test_nms_train_size_properties(rt):
    """12000 boxes (ProposalLayer train mode): greedy-NMS invariants checked on the host from the result alone -- survivors
    are in descending score order, no two survivors overlap >= thresh, every dropped box overlaps an earlier-scored survivor."""
    rs = np.random.RandomState(1)
    n = 12000
    xy = rs.uniform(0, 900, (n, 2))
    wh = rs.uniform(16, 300, (n, 2))
    scores = rs.permutation(n).astype(np.float32) / n
    dets = np.hstack([xy, xy + wh, scores[:, None]]).astype(np.float32)
    keep, n_keep = rt.nms(P.dev(rt, dets), 0.7)
    k = P.host(rt, keep)[:int(P.host(rt, n_keep)[0])]
    assert len(k) > 100 and np.all(np.diff(dets[k, 4]) < 0)

    def iou(a, b):
        iw = np.maximum(0, np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]) + 1)
        ih = np.maximum(0, np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]) + 1)
        inter = (iw * ih).astype(np.float32)
        area = lambda z: (z[:, 2] - z[:, 0] + 1) * (z[:, 3] - z[:, 1] + 1)
        return inter / (area(a)[:, None] + area(b)[None, :] - inter)
    K = dets[k]
    m = iou(K, K)
    np.fill_diagonal(m, 0)
    assert (m.astype(np.float64) >= 0.7).sum() == 0
    dropped = np.setdiff1d(np.arange(n), k)
    sup = (iou(dets[dropped], K).astype(np.float64) >= 0.7) & (K[None, :, 4] > dets[dropped, 4][:, None])
    assert sup.any(axis=1).all()


# ---- stage-2 (rcnn_train) step: SURVEY.md 8f rank 2
def test_rcnn_train_step_small(rt):
    import train_cases as T
    T.check_small_rcnn_step(rt)


def test_rcnn_train_step_vgg16(rt):
    import train_cases as T
    losses, worst = T.check_vgg_rcnn_step(rt)
    assert losses["loss_rcnn"] > 0 and worst <= 1e-3


def test_rcnn_train_step_split_products(rt):
    """RCNNTrainer(conv_math="split") on the small network and on VGG-16: the stage-2 step's trunk convolutions as split products."""
    import train_cases as T
    T.check_small_rcnn_step(rt, conv_math="split")
    losses, worst = T.check_vgg_rcnn_step(rt, conv_math="split")
    assert losses["loss_rcnn"] > 0 and worst <= 1e-3


def test_conv_workspace_self_cleaning(rt):
    P.check_conv_workspace_self_cleaning(rt)


def test_forwards_in_flight_match_serial(rt):
    """graph.ForwardsInFlight: two captured forwards on two HIP streams, four DIFFERENT images submitted back to back -- every image's outputs equal the serial
    captured forward's for that image bit for bit (separate workspaces per instance: nothing is shared but the read-only weights)."""
    import torch
    import chainer_faster_rcnn_amd as pkg
    from chainer_faster_rcnn_amd import synthetic
    from chainer_faster_rcnn_amd.graph import CapturedForward, ForwardsInFlight
    from chainer_faster_rcnn_amd.models import FasterRCNN
    params = synthetic.params(seed=1)
    h, w = 160, 224
    imgs = [rt.mem.from_numpy(synthetic.image(seed=10 + i, h=h, w=w)) for i in range(4)]

    def make_model(r):
        m = FasterRCNN(runtime=r)
        m.load_params(params)
        return m
    serial = CapturedForward(make_model(rt), imgs[0], h, w)
    want = []
    for x in imgs:
        o = serial.replay(x)
        torch.cuda.synchronize()
        want.append({k: o[k].clone() for k in ("rois", "cls_prob", "pred_boxes", "n_out")})
    fl = ForwardsInFlight(make_model, lambda: pkg.runtime.Runtime(rt.lib, pkg.runtime.TorchDeviceMemory(str(rt.mem.device))), imgs[0], h, w, n=2, probe_steps=4)
    got = []
    for rnd in range(2):                              # two images in flight, then the next two
        handles = [fl.submit(imgs[2 * rnd + i]) for i in range(2)]
        for slot, out in handles:
            fl.wait(slot)
            got.append({k: out[k].clone() for k in ("rois", "cls_prob", "pred_boxes", "n_out")})
    for i in range(4):
        for k in want[i]:
            assert torch.equal(got[i][k], want[i][k]), (i, k)


def test_captured_forward_matches_eager(rt):
    """hipGraph replay (graph.CapturedForward) == eager launches, bit for bit, also after the input buffer is replaced."""
    from chainer_faster_rcnn_amd import synthetic
    from chainer_faster_rcnn_amd.graph import CapturedForward
    from chainer_faster_rcnn_amd.models import FasterRCNN
    model = FasterRCNN(runtime=rt)
    model.load_params(synthetic.params(seed=1))
    h, w = 224, 320
    x0, x1 = [rt.mem.from_numpy(synthetic.image(seed=s, h=h, w=w)) for s in (3, 4)]
    cap = CapturedForward(model, x0, h, w)
    for x in (x0, x1, x0):
        want = {k: rt.mem.to_numpy(v).copy() for k, v in model.forward_device(x, h, w).items()}
        got = {k: rt.mem.to_numpy(v).copy() for k, v in cap.replay(x).items()}
        for k in want:
            assert np.array_equal(want[k], got[k]), k


def test_empty_proposals_pipeline(rt):
    P.check_empty_proposals_pipeline(rt)


def test_conv_f32s_split_bf16(rt):
    """fp32 convolution as six bf16 MFMA products of 3-way split operands (csrc/conv_f32s.hip) at VGG channel counts."""
    P.check_conv_f32s(rt, 16, 64, 9, 37)
    P.check_conv_f32s(rt, 3, 64, 61, 97, seed=1)
    P.check_conv_f32s(rt, 64, 64, 75, 125, seed=2)
    P.check_conv_f32s(rt, 256, 256, 38, 63, seed=3)
    P.check_conv_f32s(rt, 512, 512, 19, 32, seed=4)
    P.check_conv_f32s(rt, 48, 80, 21, 30, relu=False, seed=5)


def test_f32s_pipeline_small(rt):
    P.check_f32s_pipeline_small(rt, im_h=90, im_w=131)


def test_conv1_f32s_first_layer(rt):
    P.check_conv1_f32s(rt, 3, 64, 75, 203)
    P.check_conv1_f32s(rt, 3, 64, 600, 1000, seed=2)
    P.check_conv1_f32s(rt, 1, 24, 37, 65, relu=False, seed=1)


def test_linear_f32s(rt):
    P.check_linear_f32s(rt, 37, 116, 96, relu=False)
    P.check_linear_f32s(rt, 300, 4096, 4096, relu=True, seed=1)      # fc7
    P.check_linear_f32s(rt, 300, 512, 25088, relu=True, seed=2)      # fc6's K
    P.check_linear_f32s(rt, 300, 116, 4096, relu=False, seed=3)      # the stacked cls_score / bbox_pred head


def test_roi_pool_from_blocked_bf16(rt):
    P.check_roi_pool_blk_bf16(rt, 300, 512, 38, 63)
    P.check_roi_pool_blk_bf16(rt, 40, 24, 12, 17, seed=1)
    P.check_roi_pool_blk_bf16(rt, 64, 136, 75, 64, seed=2)


def test_rpn_heads_bf16_fused(rt):
    P.check_rpn_heads_bf16(rt, 512, 38, 63)
    P.check_rpn_heads_bf16(rt, 208, 19, 32, A=3, seed=1)


def test_conv1_f32_first_layer(rt, monkeypatch):
    P.check_conv1_f32(rt, monkeypatch, 3, 64, 75, 203)
    P.check_conv1_f32(rt, monkeypatch, 3, 64, 600, 1000, seed=2)
    P.check_conv1_f32(rt, monkeypatch, 1, 24, 37, 65, relu=False, seed=1)


def test_conv1_bf16_first_layer(rt):
    P.check_conv1_bf16(rt, 3, 64, 75, 203)
    P.check_conv1_bf16(rt, 3, 64, 600, 1000, seed=2)


def test_rpn_train_step_split_products(rt):
    """RPNTrainer(conv_math="split"): forward and input-gradient convolutions as six bf16 MFMA products of 3-way split operands -- the
    same bars as the fp32-MFMA step (loss 1e-4, every gradient 1e-3 of the oracle's autograd)."""
    import train_cases as T
    T.check_small_step(rt, conv_math="split")
    losses, worst, flipped = T.check_vgg_step(rt, conv_math="split")
    assert losses["rpn_loss"] > 0 and (worst <= 1e-3 or flipped)


def test_f32s_weight_packs(rt):
    P.check_f32s_weight_packs(rt)


def test_conv_wgrad_f32s(rt):
    P.check_conv_wgrad_f32s(rt, 64, 64, 5, 37)
    P.check_conv_wgrad_f32s(rt, 3, 64, 61, 97, seed=1)
    P.check_conv_wgrad_f32s(rt, 128, 256, 75, 125, seed=2)
    P.check_conv_wgrad_f32s(rt, 512, 512, 38, 63, seed=3)


def test_trainers_across_image_sizes(rt):
    """A differently sized image every iteration (what train_rpn.py / train_rcnn.py feed): each step equals a new trainer's, bit for bit."""
    import train_cases as T
    assert T.check_trainers_across_image_sizes(rt, sizes=((48, 64), (64, 48), (41, 57), (48, 64), (200, 150), (48, 64))) == 6


def test_nms_random_box_sets(rt):
    """Random box sets, sparse to crowded, with and without tied scores, three thresholds: the reference's keep lists."""
    P.check_nms_random_box_sets(rt)
    P.check_nms_random_box_sets(rt, sizes=(1000, 4097, 12000), seeds=(0,))


def test_anchor_target_empty_cases(rt):
    """No ground-truth box / no anchor inside the image: the reference's ValueError."""
    P.check_anchor_target_empty_cases(rt)
