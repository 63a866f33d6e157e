"""Kernel LOGIC tests on the host emulator (tests/hipemu): the unmodified .hip sources compiled for the CPU,
checked against the oracle.  These are CPU tests of indexing / wave collectives / MFMA fragment layouts /
barriers; the parity tests proper run the same checks on the real library under -m gpu."""
import sys
import os

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
import parity_cases as P  # noqa: E402
from chainer_faster_rcnn_amd import tuning  # noqa: E402


@pytest.fixture(scope="module")
def rt():
    from emu_runtime import emu_runtime
    return emu_runtime()


def test_emu_exports_every_declared_symbol(rt):
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "frcnn_hip.h")).read()
    names = set(re.findall(r"\b(frcnn_[a-z0-9_]+)\s*\(", hdr))
    for n in names:
        assert hasattr(rt.lib, n), n


def test_nms_golden_small(rt):
    P.check_nms_golden(rt, tags=("n300_t03", "n1_t07", "n65_t05"))


def test_nms_edges(rt):
    P.check_nms_edges(rt)


def test_nms_random(rt):
    P.check_nms_random(rt, n=700, seeds=(0,), thrs=(0.3, 0.7))


def test_gpu_nms_reference_ffi(rt):
    """`_nms` (models/gpu_nms.hpp:9-10) host-pointer wrapper: H2D, frcnn_nms, D2H, threshold recovery, error reporting."""
    P.check_gpu_nms_ffi(rt, tags=("n300_t03", "n65_t05", "n1_t07"))


def test_nms_staged(rt):
    P.check_nms_staged(rt, n=1200, seeds=(0,))


def test_nms_chains(rt):
    P.check_nms_chains(rt, n=150)


def test_nms_staged_strided_tail(rt):
    P.check_nms_staged_strided_tail(rt)


def test_nms_batched(rt):
    P.check_nms_batched(rt, groups=3, n=150)


def test_proposals_14x14(rt):
    P.check_proposals_golden(rt, "proposal_14x14_train_rand")


def test_proposals_edge_goldens(rt):
    P.check_proposals_edge_goldens(rt)


def test_proposals_tied_scores(rt):
    """Equal scores: ascending anchor index, in the sort and in NMS's visiting order (the oracle's tie_rule="ascending_index"); everything bit for bit."""
    P.check_proposals_tied_scores(rt)


def test_nms_edge_goldens(rt):
    P.check_nms_edge_goldens(rt)


def test_roi_pool_extreme_rois(rt):
    P.check_roi_pool_extreme_rois(rt)


def test_roi_pool_cells_kernel(rt):
    P.check_roi_pool_cells(rt)


def test_roi_pool_output_forms_when_the_cell_kernel_declines(rt):
    """frcnn_roi_pool_fwd_chw_f32s / _blk_bf16 exist on the cell-major kernel only and return FRCNN_ERR_UNSUPPORTED when it declines (a map beyond its
    76 x 64 LDS image); the runtime wrappers then take the header's documented detour -- fp32 pooling + an exact conversion -- instead of
    raising (callers guarded on the map size only by convention).  Same bits either way."""
    calls, orig = [], rt.roi_pool_fwd_chw
    rt.roi_pool_fwd_chw = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        P.check_roi_pool_blk_bf16(rt, 9, 24, 20, 70)                 # 70 cells per row: beyond the image
        rs = np.random.RandomState(7)
        x, rois = P.roi_case(rs, 9, 11, 19, 70)
        want = P.O.roi_pooling_2d(x, rois, 7, 7, 0.0625)
        ysp = rt.roi_pool_fwd_chw_f32s(P.dev(rt, x[0]), P.dev(rt, np.ascontiguousarray(rois[:, 1:])), 7, 7, 0.0625)
        assert np.array_equal(P.host(rt, rt.f32s_join(ysp)).reshape(want.shape), want)
    finally:
        rt.roi_pool_fwd_chw = orig
    assert len(calls) == 3                                           # blocked map: fp32 and bf16 outputs; split tensor: once


def test_roi_pool_cells_batches(rt):
    P.check_roi_pool_cells_batches(rt)


def test_roi_pool(rt):
    P.check_roi_pool(rt, R=9, C=128, H=12, W=17)
    P.check_roi_pool(rt, R=5, C=64, H=38, W=63, seed=1)     # VEC=1 path (C % 128 != 0)
    P.check_roi_pool(rt, R=40, C=8, H=12, W=17, seed=2)     # 3 RoI groups x 1 channel group, ragged last batch
    P.check_roi_pool(rt, R=6, C=6, H=70, W=90, seed=3)      # 3 planes per group (odd sizes: scalar copies)
    P.check_roi_pool(rt, R=37, C=7, H=12, W=17, seed=4)     # odd channel count: the backward's one-channel workgroups; more RoIs than one wave step


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 8, 10, 11, 30, 35, 34, 36, 46, 37, 38, 39])
def test_conv3x3_cfg(rt, cfg):
    P.check_conv3x3(rt, 8, 128, 9, 37, cfg=cfg)


@pytest.mark.parametrize("cfg", [201, 205, 210, 104, 110, 1205, 2205, 2210, 1010, 2010, 230, 235, 1235, 236, 234, 206, 246, 238, 237, 239])
def test_conv3x3_streamk(rt, cfg):
    """stream-K work distribution: the emulated chip has 3 CUs, so tiles split unevenly into 2..4 pieces and the
    last-arriver fix-up (partial slots, tickets, piece-ordered sum) is exercised."""
    P.check_conv3x3(rt, 24, 128, 9, 70, cfg=cfg)
    P.check_conv3x3(rt, 8, 64 if cfg % 100 not in (1, 4, 37, 38, 39) else 128, 21, 33, cfg=cfg, seed=1)     # 128-cout tiles need Cout % 128 == 0


def test_conv3x3_cin3_and_norelu(rt):
    P.check_conv3x3(rt, 3, 64, 9, 33, cfg=0)               # conv1_1: K = 27, padded to a 4-channel chunk
    P.check_conv3x3(rt, 5, 64, 5, 70, cfg=3, relu=False)   # odd Cin, width > 2 tiles


def test_maxpool(rt):
    P.check_maxpool(rt, 3, 7, 9)
    P.check_maxpool(rt, 2, 8, 6)


def test_rpn_heads(rt, monkeypatch):
    P.check_rpn_heads_forms(rt, monkeypatch, Cmid=128, H=5, W=15)        # 75 px: two whole pixel tiles + 11; waves 0-3 hold channels
    P.check_rpn_heads_forms(rt, monkeypatch, Cmid=272, H=3, W=7, A=3, seed=1)   # two load batches per wave, the last one ragged; A = 3


def test_linear(rt):
    P.check_linear(rt, 70, 140, 256, True)
    P.check_linear(rt, 9, 21, 64, False)
    P.check_linear(rt, 130, 84, 128, False)
    P.check_linear(rt, 9, 21, 72, False, seed=3)          # K % 32 != 0: the register-staged kernel (odd-pitch LDS image)
    P.check_linear(rt, 100, 130, 96, True, seed=4)        # AM = 5, ragged M and N, three panels
    P.check_linear(rt, 130, 84, 128, False, bias=False)   # NULL bias, one slab: the GEMM writes y itself (the weight-gradient product)
    P.check_linear(rt, 40, 64, 512, True, seed=5, bias=False)     # NULL bias through the combine pass


def test_head_decode(rt):
    P.check_head_decode(rt)


# ---- RPN training step kernels (csrc/train.hip)
def test_bbox_overlaps(rt):
    P.check_bbox_overlaps(rt, N=300, K=5)


def test_anchor_target(rt):
    P.check_anchor_target(rt, 14, 14, 224, 224, 3)              # the reference's own test geometry (tests/test_anchor_target_layer.py:18-28)
    P.check_anchor_target(rt, 19, 32, 300, 500, 6, seed=1)


def test_rpn_loss(rt):
    P.check_rpn_loss(rt)


def test_conv_backward(rt):
    P.check_conv_backward(rt, 64, 64, 9, 37)
    P.check_conv_backward(rt, 3, 64, 7, 33, seed=1)             # conv1_1: channel padding, no input gradient
    P.check_conv_backward(rt, 64, 64, 5, 40, ksize=1, seed=2)   # the RPN heads' 1x1


@pytest.mark.parametrize("env", [{"FRCNN_WGRAD_DB": "1"}, {"FRCNN_WGRAD_DB": "0"}])
def test_conv_wgrad_forms(rt, monkeypatch, env):
    """A/B forms of the 3x3 weight-gradient kernel: the single-buffer form (two workgroups per CU) and the double-buffered one (one per CU) forced on every layer.  Several tiles per workgroup (the emulated chip has 3 CUs) and ragged borders."""
    for k, v in env.items():
        tuning.set(k, v)
    P.check_conv_backward(rt, 64, 64, 9, 70)
    P.check_conv_backward(rt, 3, 64, 7, 33, seed=1)


def test_conv1_wgrad_first_layer_form(rt, monkeypatch):
    """conv1_1's weight gradient: the (channel, tap)-row kernel and the generic kernel (FRCNN_WGRAD_CONV1=generic) against the oracle;
    several tiles per workgroup, ragged right / bottom borders, one and two input channels as well"""
    for cin, h, w in ((3, 7, 33), (3, 12, 70), (1, 5, 40), (2, 9, 64)):
        P.check_conv_backward(rt, cin, 64, h, w, seed=cin)
    P.check_conv_backward(rt, 3, 128, 6, 37, seed=5)
    tuning.set("FRCNN_WGRAD_CONV1", "generic")
    P.check_conv_backward(rt, 3, 64, 7, 33, seed=1)


def test_conv_relu_pool_train(rt):
    P.check_conv_relu_pool_train(rt, 8, 64, 9, 37)              # odd H and W: clipped windows on both edges
    P.check_conv_relu_pool_train(rt, 64, 128, 8, 70, seed=1)     # the 128-cout tiles (four accumulators per wave)


def test_conv_dgrad_unpool(rt):
    P.check_conv_dgrad_unpool(rt, 64, 64, 9, 37)                 # odd pre-pool height and width
    P.check_conv_dgrad_unpool(rt, 64, 128, 8, 66, seed=1)         # even sizes, the 128-cout tiles


def test_pack_dgrad_many(rt):
    P.check_pack_dgrad_many(rt)


def test_maxpool_bwd(rt):
    P.check_maxpool_bwd(rt, 3, 7, 9)
    P.check_maxpool_bwd(rt, 2, 8, 6)


def test_sgd(rt):
    P.check_sgd(rt, n=5000)


# ---- ResNet trunk (BASELINE config 4)
def test_resnet_pieces(rt):
    P.check_resnet_pieces(rt)


def test_resnet_tiny(rt):
    P.check_resnet(rt, blocks=(2, 1, 1, 1), im_h=40, im_w=70)     # one `a` + one `b` block, all four stages, odd sizes


# ---- bf16 convolution stack (BASELINE config 3)
def test_conv_bf16(rt):
    P.check_conv_bf16(rt, 16, 64, 9, 37)
    P.check_conv_bf16(rt, 3, 64, 7, 33, seed=1)                    # conv1_1: channels padded 3 -> 16
    P.check_conv_bf16(rt, 32, 54, 5, 40, ksize=1, relu=False, seed=2)   # the stacked RPN heads: 54 real couts of 64


def test_conv_bf16_pool_fused(rt):
    P.check_conv_bf16_pool(rt, 16, 64, 9, 37)            # odd H, W: clipped windows
    P.check_conv_bf16_pool(rt, 16, 128, 8, 64, seed=1)


def test_maxpool_bf16(rt):
    P.check_maxpool_bf16(rt, 16, 7, 9)
    P.check_maxpool_bf16(rt, 32, 8, 6, seed=1)


def test_detections_postprocess(rt):
    P.check_detections(rt, R=60)


def test_linear_bf16(rt):
    P.check_linear_bf16(rt, 70, 140, 256, True)
    P.check_linear_bf16(rt, 9, 21, 72, False)              # K % 64 != 0: the register-staged kernel
    P.check_linear_bf16(rt, 100, 130, 128, True, seed=2)   # LDS-DMA kernel, AM = 5, ragged M and N, two panels
    P.check_linear_bf16(rt, 20, 40, 64, False, seed=3)     # AM = 1, one panel


def test_linear_bf16_tiled(rt):
    """The weight-stream kernel (csrc/linear_bf16.hip) on the host emulator: splits shorter and longer than the ring (prologue / steady state / tail
    each taken), ragged M and N, all three row-tile forms, two row blocks."""
    P.check_linear_bf16_tiled(rt, 300, 140, 32 * 12, True)         # MT 5 (320 rows), 12 chunks: prologue + seven steady-state iterations + tail; N ragged (two column blocks)
    P.check_linear_bf16_tiled(rt, 100, 130, 32 * 3, False, seed=2)  # MT 3, three chunks: shorter than the ring
    P.check_linear_bf16_tiled(rt, 20, 40, 32, True, seed=3)         # MT 1, ONE chunk
    P.check_linear_bf16_tiled(rt, 330, 128, 32 * 6, False, seed=4)  # two row blocks, exactly NS + 1 chunks


def test_linear_bf16_tiled_with_late_landing(rt, monkeypatch):
    """The same with every LDS-DMA piece landing only at the wait that covers it (HIPEMU_DMA_DEFER): a miscounted vmcnt reads a stage before it has arrived."""
    monkeypatch.setenv("HIPEMU_DMA_DEFER", "1")
    P.check_linear_bf16_tiled(rt, 300, 128, 32 * 8, True)
    P.check_linear_bf16_tiled(rt, 70, 128, 32 * 9, False, seed=5)


def test_f16_instantiation_of_the_16_bit_chain(rt):
    """csrc/conv_f16.hip, conv_f16_pair.hip, linear_f16.hip: the bf16 kernel sources compiled with fp16 pack / widen / MFMA (north_star: "fp16/bf16 accumulate
    fp32").  The bf16 parity checks, run on a runtime whose *_bf16 methods resolve to the *_f16* entry points and an oracle fed fp16-rounded operands."""
    r16 = rt.with_half("f16")
    assert r16.half == "f16" and rt.half == "bf16" and r16.lib is rt.lib
    with P.half_format("f16"):
        x = np.array([1.0, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 65504.0, 70000.0, 6e-8, -2.5], np.float32)     # ties to even; overflow -> Inf; the smallest subnormal
        with np.errstate(over="ignore"):
            want_bits = x.astype(np.float16).view(np.int16)
        assert np.array_equal(P.host(rt, r16.to_bf16(P.dev(rt, x))), want_bits)
        P.check_conv_bf16(r16, 19, 70, 9, 37)                     # ragged channels, conv_dma kernel
        P.check_conv_bf16(r16, 32, 64, 7, 11, ksize=1, relu=False)
        P.check_conv_bf16_pool(r16, 16, 64, 9, 37)
        P.check_maxpool_bf16(r16, 32, 9, 13)
        P.check_linear_bf16(r16, 70, 140, 256, True)
        P.check_linear_bf16_tiled(r16, 100, 130, 32 * 9, True, seed=2)
        P.check_rpn_heads_bf16(r16, 64, 7, 9)
        P.check_conv1_pair_bf16(r16, 12, 40)
        P.check_conv_bf16_strip(r16, 910, 128, 64, 20, 64)        # form D, direct stores
        P.check_roi_pool_blk_bf16(r16, 9, 16, 12, 17)             # csrc/roi_f16.hip: pooling straight from the blocked fp16 map (fp32 and fp16 outputs)
        P.check_roi_pool(r16, R=12, C=16, H=12, W=17)             # ... and the fp32-in / fp16-out form among check_roi_pool's output forms


def test_conv_relu_pool_fused(rt):
    P.check_conv_relu_pool(rt, 8, 64, 9, 37)          # odd H and W: clipped windows on both edges
    P.check_conv_relu_pool(rt, 16, 128, 8, 64, seed=1)


def test_img_preprocessing(rt):
    P.check_preprocess(rt, 37, 50)            # scale 600/37: upsampling, both clamps
    P.check_preprocess(rt, 60, 200, seed=1)   # max_size rule: scale 1000/200


def test_conv_bf16_eight_row_tiles(rt, monkeypatch):
    tuning.set("FRCNN_BF16_RP", "4")                        # force the 8-wave / 8-row decomposition
    P.check_conv_bf16(rt, 16, 64, 13, 37, seed=3)
    P.check_conv_bf16(rt, 32, 128, 8, 33, seed=4)


@pytest.mark.parametrize("mode", ["321", "231", "141", "132", "222", "0", "223", "233", "323", "224", "324", "124", "133"])
def test_conv_bf16_lds_dma(rt, monkeypatch, mode):
    """Every LDS-DMA staging variant of the 3x3 bf16 kernel (lane-linear swizzled LDS image, NS-stage ring) and the
    register-staged kernel ("0"): same results.  The default picks 141 / 231 by launch size."""
    tuning.set("FRCNN_BF16_DMA", mode)
    P.check_conv_bf16(rt, 16, 64, 9, 37)                  # one chunk
    P.check_conv_bf16(rt, 3, 64, 7, 33, seed=1)
    P.check_conv_bf16(rt, 80, 128, 13, 70, seed=2)        # five chunks: the ring wraps; three x tiles, ragged rows
    P.check_conv_bf16_pool(rt, 48, 64, 9, 37, seed=3)


@pytest.mark.parametrize("mode", ["901", "902", "903", "900", "907", "908", "909", "910"])
def test_conv_bf16_strip_forms(rt, monkeypatch, mode):
    """The strip forms of the 3x3 bf16 kernel (csrc/conv_bf16_strip.h: one wave per SIMD, software-pipelined ring; D = 909 and C = 903 are default
    picks, the others selectable) against the oracle like every other staging variant -- one stage, the rings wrapping (5 and 12 stages),
    several x / y / cout tiles, ragged edges, 54 couts of 64 (form C: a 32-cout tile that is half padding), the fused pool."""
    tuning.set("FRCNN_BF16_DMA", mode)
    P.check_conv_bf16(rt, 64, 64, 9, 37)                   # forms A, B: 4 stages; C: one stage of four K ways
    if mode in ("903", "907"):                             # one chunk cannot be split over K ways: the explicit form refuses, 900 falls back
        with pytest.raises(Exception):
            P.check_conv_bf16(rt, 3, 64, 7, 33, seed=1)
    else:
        P.check_conv_bf16(rt, 3, 64, 7, 33, seed=1)
    P.check_conv_bf16(rt, 192, 54, 23, 70, seed=2)         # 12 chunks: every ring wraps; two row blocks (A), three (B), five (C)
    if mode not in ("903", "907"):
        P.check_conv_bf16_pool(rt, 80, 128, 22, 37, seed=3)   # odd tile rows rule the pool out for form C, five chunks the two-way K split


@pytest.mark.parametrize("form", [901, 902, 903, 907, 909, 910, 911])
def test_conv_bf16_strip_same_as_default(rt, form):
    P.check_conv_bf16_strip(rt, form, 128, 96, 21, 45, seed=4)
    P.check_conv_bf16_strip(rt, form, 64, 64, 12, 64, pool=form != 903, seed=5)
    if form == 911:                                           # form E: eight waves, 20-row tiles: two row blocks + a ragged third, 12 chunks (the ring wraps)
        P.check_conv_bf16_strip(rt, form, 128, 64, 45, 40, pool=True, seed=6)


def test_vgg16_bf16_trunk_through_the_strip_picks(rt):
    """The full-width bf16 trunk, conv1_1 ... conv4_1, at 22 x 37 through the model class: on the emulated three-CU chip the default rule sends
    five of those ten layers (two of them pool-fused) through strip form D; same map bit for bit with the rule off, within the bf16 bar of
    the fp32 oracle."""
    assert P.check_vgg_bf16_trunk(rt, 22, 37) < 3e-2


def test_default_rule_picks_form_c_on_a_chip_of_eight_cus():
    """The default rule's second branch -- a launch too small for form D that ONE round of form C covers -- cannot trigger on the three-CU
    emulated chip; a subprocess with HIPEMU_CUS=8 (the CU count is cached per process) runs a 256 -> 64 convolution on a 10 x 64 map:
    the plan says form C, and the launch -- default picks, no hook -- matches the oracle like every other variant."""
    import subprocess
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from emu_runtime import emu_runtime; import parity_cases as P\n"
            "rt = emu_runtime()\n"
            "assert rt.lib.frcnn_conv_bf16_plan(256, 64, 10, 64, 3, 0) == 903, rt.lib.frcnn_conv_bf16_plan(256, 64, 10, 64, 3, 0)\n"
            "assert rt.lib.frcnn_conv_bf16_plan(256, 64, 10, 64, 3, 2) == 0\n"
            "assert rt.lib.frcnn_conv_bf16_plan(256, 256, 10, 64, 3, 0) == 910\n"
            "P.check_conv_bf16(rt, 256, 64, 10, 64, seed=3)\n"
            "P.check_conv_bf16_strip(rt, 903, 256, 64, 10, 64, seed=3)\n"
            "print('ok')\n") % (os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"), os.path.dirname(os.path.abspath(__file__)),
                                 os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, HIPEMU_CUS="8")
    for k in ("FRCNN_BF16_DMA", "FRCNN_BF16_STRIP", "FRCNN_BF16_SPLIT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


def test_ticketed_fixups_do_not_depend_on_arrival_order(rt, monkeypatch):
    """The launches whose workgroups hand partial results to each other through a ticket (the fp32 convolution's stream-K pieces, the bf16
    convolution's split-K, the split-K FC layers; the last arriver sums the pieces in piece order) give bit-identical results whichever
    workgroup arrives last: the emulator runs workgroups first to last by default and last to first with HIPEMU_BLOCK_ORDER=reverse."""
    rs = np.random.RandomState(3)
    x = rs.randn(1, 128, 9, 70).astype(np.float32)               # 8 K-chunks: two bf16 splits of four
    w = (rs.randn(64, 128, 3, 3) * 0.05).astype(np.float32)
    b = (rs.randn(64) * 0.1).astype(np.float32)
    xm, wm = rs.randn(70, 512).astype(np.float32), (rs.randn(140, 512) * 0.05).astype(np.float32)
    bm = (rs.randn(140) * 0.1).astype(np.float32)

    def run():
        out = [P.host(rt, rt.conv3x3(P.dev(rt, x), rt.pack_conv3x3_w(P.dev(rt, w)), P.dev(rt, b), relu=True))]
        tuning.set("FRCNN_BF16_SPLIT", "2")
        out.append(P.host(rt, rt.conv_bf16(rt.bf16_from_nchw(P.dev(rt, x)), rt.bf16_pack_conv_w(P.dev(rt, w), 3), P.dev(rt, b), 128, 64, 3, relu=True)))
        tuning.set("FRCNN_BF16_SPLIT", None)
        out.append(P.host(rt, rt.linear(P.dev(rt, xm), P.dev(rt, wm), P.dev(rt, bm), relu=True)))
        return out
    first = run()
    monkeypatch.setenv("HIPEMU_BLOCK_ORDER", "reverse")
    for a, c in zip(first, run()):
        assert a.shape == c.shape and np.array_equal(a, c)


def test_lds_dma_kernels_with_late_landing(rt, monkeypatch):
    """Every kernel family that stages through LDS-DMA (buffer_load ... lds + counted s_waitcnt vmcnt + fence-less barriers), once more with
    the emulator landing each piece at the LATEST legal moment -- the wait that covers it (HIPEMU_DMA_DEFER=1, tests/hipemu/hip/hip_runtime.h)
    -- instead of at issue.  By default the emulator lands a piece the moment it is issued, which catches a stage refilled too early but is
    blind to a fragment read before its stage's wait and to a wait that allows one stage too many in flight (changing the strip kernel's
    hand-over to `wait_allow(allow + 1)` passes every other test here and fails this one); a ring protocol has to hold at both ends."""
    monkeypatch.setenv("HIPEMU_DMA_DEFER", "1")
    for form in (901, 902, 903, 910):                                   # strip forms: 8 chunks, every ring wraps (3, 4, 2, 2 stages)
        P.check_conv_bf16_strip(rt, form, 128, 96, 21, 45, seed=4)
    P.check_conv_bf16_strip(rt, 909, 64, 64, 12, 64, pool=True, seed=5)
    for mode in ("231", "321", "141", "224"):                           # conv_dma_bf16_kernel: two / three-stage rings, single stage, 16-row tiles
        tuning.set("FRCNN_BF16_DMA", mode)
        P.check_conv_bf16(rt, 80, 128, 13, 70, seed=2)
    tuning.set("FRCNN_BF16_DMA", None)
    P.check_conv3x3(rt, 48, 64, 9, 70, seed=1)                          # fp32 MFMA convolution (conv.hip), stream-K pieces included
    P.check_conv_f32s(rt, 192, 64, 5, 33, seed=6)                       # split products (conv_f32s.hip)
    P.check_conv_backward(rt, 64, 64, 9, 70)                            # weight / input gradients (train.hip)
    P.check_linear(rt, 70, 140, 256, True)                              # fp32 FC (gemm.hip)
    P.check_linear_bf16(rt, 100, 130, 128, True, seed=2)                # bf16 FC
    P.check_linear_f32s(rt, 40, 96, 192, True, seed=1)                  # split-product FC


def test_conv_bf16_strip_forms_on_ragged_shapes(rt):
    """A seeded sweep of awkward launches through the strip forms that are default picks (D = 909, C = 903) and the two one-workgroup
    forms: maps smaller than a tile, a single row / column, channel counts that fill neither a 16-channel block nor a 32- / 64-cout
    tile, odd sizes under the fused pool -- each against conv_dma_bf16_kernel (bit-identical, or summation-order noise for the K split)."""
    rs = np.random.RandomState(321)
    for t in range(8):
        form = int(rs.choice([901, 902, 903, 909, 910]))
        kways = 4 if form == 903 else 1
        cin = int(rs.choice([16, 24, 40, 64, 100])) if kways == 1 else int(rs.choice([49, 64, 120, 128]))      # 903: chunks a multiple of 4
        cout, h, w = int(rs.choice([1, 20, 33, 64, 70])), int(rs.randint(1, 24)), int(rs.randint(1, 70))
        P.check_conv_bf16_strip(rt, form, cin, cout, h, w, pool=bool(rs.randint(2)) and form != 903, seed=t)


@pytest.mark.parametrize("split,mode", [("2", None), ("4", None), ("2", "224"), ("4", "223")])
def test_conv_bf16_split_k(rt, monkeypatch, split, mode):
    """Split-K form of the bf16 3x3 kernel: partial tiles through the workspace, last arriver sums in split order."""
    tuning.set("FRCNN_BF16_SPLIT", split)
    if mode:
        tuning.set("FRCNN_BF16_DMA", mode)
    P.check_conv_bf16(rt, 128, 64, 9, 37, seed=5)         # 8 chunks: 2 or 4 splits of >= 2 chunks ... (the picker wants >= 4 per split)
    P.check_conv_bf16(rt, 256, 128, 6, 40, seed=6)        # 16 chunks
    P.check_conv_bf16_pool(rt, 256, 64, 8, 33, seed=7)


def test_conv_workspace_self_cleaning(rt):
    P.check_conv_workspace_self_cleaning(rt)


def test_empty_proposals_pipeline(rt):
    P.check_empty_proposals_pipeline(rt)


def test_conv_f32s_split_bf16(rt):
    """fp32 convolution as six bf16 MFMA products of 3-way split operands (csrc/conv_f32s.hip)."""
    P.check_conv_f32s(rt, 16, 64, 9, 37)                  # one chunk, ragged rows / columns, odd sizes through the fused pool
    P.check_conv_f32s(rt, 3, 64, 6, 34, seed=1)           # conv1_1: channels padded 3 -> 16; two x tiles
    P.check_conv_f32s(rt, 48, 80, 5, 30, relu=False, seed=2)   # three chunks; 80 couts: a ragged second cout tile
    P.check_conv_f32s(rt, 1, 21, 1, 1, seed=3)            # one pixel, one channel, 21 couts
    P.check_conv_f32s(rt, 17, 33, 3, 65, relu=False, seed=4)   # 17 -> 32 padded channels, three x tiles, fewer rows than a tile


@pytest.mark.parametrize("split", ["2", "3"])
def test_conv_f32s_split_k(rt, monkeypatch, split):
    """Few-tile launches split their K range over several workgroups (ticket + deterministic fix-up by the last one)."""
    tuning.set("FRCNN_F32S_SPLIT", split)
    P.check_conv_f32s(rt, 192, 64, 5, 33, seed=6)          # 12 chunks: 2 or 3 splits of >= 4
    if split == "2":
        tuning.set("FRCNN_F32S_XCD", "1")          # the XCD-aware (pixel tile, split) enumeration (off by default)
        P.check_conv_f32s(rt, 192, 128, 5, 33, seed=6)     # two cout tiles: two XCDs per tile


def test_f32s_pipeline_small(rt):
    P.check_f32s_pipeline_small(rt)


def test_conv1_f32s_first_layer(rt):
    P.check_conv1_f32s(rt, 3, 64, 11, 70)                  # two x tiles (64 + 6 px), three y tiles, ragged rows
    P.check_conv1_f32s(rt, 1, 24, 5, 33, relu=False, seed=1)   # one channel (K = 9), 24 couts: one block, padded to 32


def test_first_layer_heads_and_blocked_pooling_on_ragged_shapes(rt, monkeypatch):
    """A seeded sweep of awkward shapes (maps smaller than a tile, channel counts that fill neither a block of 32 nor of 16, a single
    pixel) through the kernels added late in round 2: the first-layer kernel in its three arithmetic forms, both fused RPN-heads
    launches, RoI pooling from the channel-blocked map."""
    rs = np.random.RandomState(123)
    for t in range(5):
        cin, cout = int(rs.randint(1, 4)), int(rs.choice([1, 7, 16, 31, 32, 33, 48, 64]))
        h, w = int(rs.randint(1, 10)), int(rs.randint(1, 80))
        if cout % 4 == 0:
            P.check_conv1_f32(rt, monkeypatch, cin, cout, h, w, relu=bool(rs.randint(2)), seed=t)
        P.check_conv1_f32s(rt, cin, cout, h, w, relu=bool(rs.randint(2)), seed=t)
        P.check_conv1_bf16(rt, cin, cout, h, w, seed=t)
    for t in range(3):
        cmid, h, w, a = int(rs.choice([16, 48, 100, 130])), int(rs.randint(1, 6)), int(rs.randint(1, 40)), int(rs.choice([1, 3, 9, 10]))
        P.check_rpn_heads_forms(rt, monkeypatch, Cmid=cmid, H=h, W=w, A=a, seed=t)
        P.check_rpn_heads_bf16(rt, cmid, h, w, A=a, seed=t)
    for t in range(3):
        c, h, w, r = int(rs.choice([1, 9, 24, 33])), int(rs.randint(2, 40)), int(rs.randint(2, 64)), int(rs.randint(1, 40))
        P.check_roi_pool_blk_bf16(rt, r, c, h, w, seed=t)


def test_roi_pool_from_blocked_bf16(rt):
    P.check_roi_pool_blk_bf16(rt, 40, 24, 12, 17)               # 24 channels: the second block's upper half is padding
    P.check_roi_pool_blk_bf16(rt, 9, 40, 50, 40, seed=1)         # the 76-row image, five channel groups (the last one in block 2, lower half)


def test_rpn_heads_bf16_fused(rt):
    P.check_rpn_heads_bf16(rt, 128, 5, 15)                  # 8 chunks: waves 0-1 hold a batch each; 75 px = two tiles + 11
    P.check_rpn_heads_bf16(rt, 208, 3, 7, A=3, seed=1)      # 13 chunks (a ragged last batch), 18 outputs: CoutP = 32, one cout block


def test_conv1_f32_first_layer(rt, monkeypatch):
    P.check_conv1_f32(rt, monkeypatch, 3, 64, 11, 70)                      # ragged right edge (70 = 64 + 6: a partial group of four)
    P.check_conv1_f32(rt, monkeypatch, 3, 64, 6, 67, seed=2)               # W % 4 != 0: unaligned 16-byte stores, 3-px tail
    P.check_conv1_f32(rt, monkeypatch, 1, 24, 5, 33, relu=False, seed=1)   # one channel, one cout block, no ReLU
    tuning.set("FRCNN_CONV1_GRID", "2")                            # the strided tile loop
    P.check_conv1_f32(rt, monkeypatch, 3, 64, 11, 70, seed=3)


@pytest.mark.parametrize("grid", ["1", "4"])
def test_conv1_persistent_tile_loop(rt, monkeypatch, grid):
    """The first-layer kernel is a persistent launch: with fewer workgroups than tiles every workgroup strides over several tiles
    (next halo prefetched under the current tile's units) -- forced here on a six-tile image; 4 workgroups = an uneven split."""
    tuning.set("FRCNN_CONV1_GRID", grid)
    P.check_conv1_f32s(rt, 3, 64, 11, 70, seed=3)
    P.check_conv1_bf16(rt, 3, 64, 11, 70, seed=3)


def test_linear_f32s(rt):
    P.check_linear_f32s(rt, 37, 116, 96, relu=False)            # AM = 3, one ragged N block (the stacked head: 116 columns)
    P.check_linear_f32s(rt, 100, 160, 320, relu=True, seed=1)   # AM = 5, two N blocks, split-K


def test_conv1_bf16_first_layer(rt):
    P.check_conv1_bf16(rt, 3, 64, 11, 70)
    P.check_conv1_bf16(rt, 1, 24, 5, 33, seed=1)


def test_f32s_weight_packs(rt):
    P.check_f32s_weight_packs(rt)


def test_conv_wgrad_f32s(rt):
    P.check_conv_wgrad_f32s(rt, 64, 64, 5, 37)               # two x tiles, three row tiles (ragged), one (ci, co) tile
    P.check_conv_wgrad_f32s(rt, 3, 80, 4, 33, seed=1)         # conv1_1's 3 input channels; 80 couts: a ragged second co tile


@pytest.mark.parametrize("h,w,cin,rw", [(24, 64, 3, None), (13, 37, 3, None), (30, 33, 1, None), (17, 70, 2, None), (41, 100, 3, None),
                                        (24, 64, 3, 6), (13, 37, 3, 6), (30, 33, 1, 4), (17, 70, 3, 4)])
def test_conv1_pair_bf16(rt, h, w, cin, rw):
    """conv1_1 + conv1_2 + pool as one launch (csrc/conv_bf16_pair.hip): whole tiles, ragged rows / columns, odd sizes (ceil-mode windows with one row /
    column), one input channel, several tiles per workgroup (the emulator seats few workgroups: the persistent loop wraps) -- for the default
    producer / consumer form (rw None) and for the one-wave-per-SIMD form at both tile heights."""
    P.check_conv1_pair_bf16(rt, h, w, Cin=cin, rw=rw)


def test_vgg16_bf16_trunk_with_the_conv1_pair_launch(rt, monkeypatch):
    """The bf16 trunk's default first launch (conv1_1 + conv1_2 + pool1 fused, csrc/conv_bf16_pair.hip) through the model class, conv1_1 ... conv2_2 at full
    width: the same map bit for bit as the three-entry form it replaces, and a per-layer collection still sees conv1_1's own map."""
    import functools
    from chainer_faster_rcnn_amd import synthetic
    from chainer_faster_rcnn_amd.models import VGG16Prev
    from chainer_faster_rcnn_amd.models.vgg16 import LAYERS
    layers = LAYERS[:[l[0] if l != "pool" else None for l in LAYERS].index("conv2_2") + 1]
    trunk = VGG16Prev(layers=layers, runtime=rt, conv_dtype="bf16")
    trunk.load_params(synthetic.params(seed=1), "trunk/")
    x = rt.mem.from_numpy(synthetic.image(seed=2, h=24, w=40))
    outs = {}
    for flag in ("1", "0"):                                      # (read at every call)
        tuning.set("FRCNN_BF16_CONV1_PAIR", flag)
        assert trunk.conv1_pair_applies() == (flag == "1")
        outs[flag] = rt.mem.to_numpy(trunk(x)).copy()
    assert np.array_equal(outs["1"], outs["0"]) and np.abs(outs["1"]).max() > 0
    tuning.set("FRCNN_BF16_CONV1_PAIR", "1")
    col = {}
    trunk(x, collect=col)
    assert "conv1_1" in col and "pool1" in col


def test_nms_random_box_sets(rt):
    """Random box sets, sparse to crowded, with and without tied scores, three thresholds: the reference's keep lists."""
    P.check_nms_random_box_sets(rt)


def test_anchor_target_empty_cases(rt):
    """No ground-truth box / no anchor inside the image: the reference's ValueError."""
    P.check_anchor_target_empty_cases(rt)
