"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the product never
falls back when the GPU is absent, the models/ mirror keeps the reference's names and error behaviour."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    return sorted(set(re.findall(r"\b(frcnn_[a-z0-9_]+)\s*\(", hdr)))


def test_device_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    so = ge.build()                                   # hipcc cross-compiles gfx950 without a GPU
    import ctypes
    lib = ctypes.CDLL(so)
    for name in _declared():
        assert hasattr(lib, name), name
    assert hasattr(lib, "_nms")                         # the reference's own C FFI (models/gpu_nms.hpp:9-10), same name and signature
    hdr = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    ref = "void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float nms_overlap_thresh, int device_id);"
    norm = lambda t: re.sub(r"\s+", "", t.replace("* ", "*").replace(" *", "*"))
    assert norm(ref) in norm(hdr)
    import chainer_faster_rcnn_amd as pkg
    assert sorted(pkg._lib.SIGNATURES) == _declared()   # the binding table mirrors the header one to one
    assert lib.frcnn_abi_version() == 24


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import chainer_faster_rcnn_amd as pkg
    with pytest.raises(pkg._lib.FrcnnError):
        pkg.runtime.default_runtime()
    from chainer_faster_rcnn_amd.models import cpu_nms
    with pytest.raises(pkg._lib.FrcnnError):
        cpu_nms(np.zeros((3, 5), np.float32), 0.7)


def test_product_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "chainer-faster-rcnn_amd")
    for d, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), os.path.join(d, f)
                assert "hipemu" not in src, os.path.join(d, f)


def test_models_surface_matches_reference_names():
    from chainer_faster_rcnn_amd import models
    for name in ("ProposalLayer", "cpu_nms", "gpu_nms", "roi_pooling_2d", "ROIPooling2D", "generate_anchors",
                 "bbox_transform_inv", "clip_boxes", "VGG16Prev", "VGG16", "RegionProposalNetwork", "FasterRCNN"):
        assert hasattr(models, name)
    PL = models.ProposalLayer
    assert (PL.RPN_NMS_THRESH, PL.TRAIN_RPN_PRE_NMS_TOP_N, PL.TRAIN_RPN_POST_NMS_TOP_N, PL.TEST_RPN_PRE_NMS_TOP_N,
            PL.TEST_RPN_POST_NMS_TOP_N, PL.RPN_MIN_SIZE) == (0.7, 12000, 2000, 6000, 300, 16)


def test_proposal_layer_mirror_on_emulated_kernels(golden):
    """Same call as the reference's tests/test_proposal_layer.py:20-33 (train-mode default, Variables in)."""
    from emu_runtime import emu_runtime
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.models import ProposalLayer, generate_anchors
    rt = emu_runtime()
    G = golden("proposal_14x14_train_rand")
    pl = ProposalLayer(runtime=rt)
    assert pl.train and pl._num_anchors == 9 and (pl._pre_nms_top_n, pl._post_nms_top_n) == (12000, 2000)
    assert np.array_equal(pl._anchors, golden("anchors")["a_8_16_32"])
    assert np.array_equal(generate_anchors(), golden("anchors")["a_default"])
    p, s = pl(Variable(G["rpn_cls_prob"]), Variable(G["rpn_bbox_pred"]), Variable(G["img_info"]))
    assert p.shape == G["proposals"].shape and s.shape == G["probs"].shape
    assert np.allclose(p, G["proposals"], rtol=5e-7, atol=1e-4) and np.array_equal(s, G["probs"])
    pl.train = False
    assert (pl._pre_nms_top_n, pl._post_nms_top_n) == (6000, 300)
    with pytest.raises(AssertionError):       # the reference's type checks (proposal_layer.py:85-100)
        pl(Variable(G["rpn_cls_prob"][:, :10]), Variable(G["rpn_bbox_pred"]), Variable(G["img_info"]))
    with pytest.raises(AssertionError):
        pl(G["rpn_cls_prob"], Variable(G["rpn_bbox_pred"]), Variable(G["img_info"]))      # not a Variable


def test_cpu_nms_mirror_errors_and_result(golden):
    from emu_runtime import emu_runtime
    from chainer_faster_rcnn_amd.models import cpu_nms
    rt = emu_runtime()
    G = golden("cpu_nms")
    assert cpu_nms(G["n65_t05_dets"], 0.5, runtime=rt) == G["n65_t05_keep"].tolist()
    with pytest.raises(ValueError):
        cpu_nms(G["n65_t05_dets"].astype(np.float64), 0.5, runtime=rt)      # Cython: "Buffer dtype mismatch"
    with pytest.raises(TypeError):
        cpu_nms(G["n65_t05_dets"], 1, runtime=rt)                           # thresh must be a Python float
    assert rt.lib.frcnn_nms(None, -1, 0.5, 0, None, None, None, 0, None) == -1   # FRCNN_ERR_INVALID, not a crash


def test_roi_pooling_function_object(golden):
    from emu_runtime import emu_runtime
    from chainer_faster_rcnn_amd.models import ROIPooling2D, roi_pooling_2d
    from oracle import frcnn_oracle as O
    rt = emu_runtime()
    rs = np.random.RandomState(0)
    x = rs.randn(1, 64, 10, 12).astype(np.float32)
    rois = np.array([[0, 0, 0, 100, 90], [0, 32, 16, 150, 140]], np.float32)
    y = roi_pooling_2d(x, rois, 7, 7, 1 / 16., runtime=rt)
    want, am = O.roi_pooling_2d(x, rois, return_argmax=True)
    assert np.array_equal(y, want)
    f = ROIPooling2D(7, 7, 1 / 16., runtime=rt)
    y2, = f.forward((x, rois))
    gx, none = f.backward((x, rois), (np.ones_like(y2),))
    assert none is None and np.array_equal(f.argmax_data, am)
    assert np.allclose(gx, O.roi_pooling_2d_backward(np.ones_like(want), am, rois, x.shape))
    with pytest.raises(ValueError):
        roi_pooling_2d(x, rois[:, :4], 7, 7, 1 / 16., runtime=rt)


def test_split_tensor_entry_points_reject_bad_arguments():
    """The fp32-on-bf16-matrix-cores entry points (csrc/conv_f32s.hip, train.hip) fail with FRCNN_ERR_INVALID, not with a launch, on
    arguments outside their contract (checked on the host-compiled library: no GPU needed)."""
    import ctypes
    from emu_runtime import emu_runtime
    rt = emu_runtime()
    L, m = rt.lib, rt.mem
    buf = m.empty((4096,), "f32")
    p = m.ptr(buf)
    INVALID = -1 if not hasattr(L, "FRCNN_ERR_INVALID") else L.FRCNN_ERR_INVALID
    bad = [
        L.frcnn_conv3x3_f32s(p, p, p, p, 16, 16, 4, 4, 1, 5, None),                      # out_mode out of range
        L.frcnn_conv3x3_f32s(p, p, p, p, 16, 16, 4, 4, 0, 2, None),                      # fused pool needs the ReLU
        L.frcnn_conv3x3_f32s(None, p, p, p, 16, 16, 4, 4, 1, 0, None),
        L.frcnn_conv1_f32s(p, p, p, p, 4, 64, 4, 4, 1, None),                            # first-layer kernel: Cin <= 3
        L.frcnn_conv1_f32s(p, p, p, p, 3, 65, 4, 4, 1, None),                            # ... and Cout <= 64
        L.frcnn_linear_f32s(p, p, p, p, 4, 8, 40, 0, 0, p, 4096 * 4, None),              # K % 32 != 0
        L.frcnn_linear_f32s(p, p, p, p, 4, 8, 64, 0, 0, None, 0, None),                  # no workspace
        L.frcnn_conv3x3_f32s_train(p, p, p, None, None, None, 16, 16, 4, 4, 1, None, 0, None),    # neither output
        L.frcnn_conv_wgrad_f32s(p, p, p, 16, 16, 4, 4, None, 0, None),                   # no workspace
        L.frcnn_f32s_pack_many(None, 1, None),
        L.frcnn_roi_pool_fwd_chw_f32s(p, 8, 100, 100, p, 4, 4, 7, 7, ctypes.c_float(0.0625), p, None),   # map too large for the cell kernel
    ]
    assert all(rc != 0 for rc in bad), bad


def test_tuning_registry_abi():
    """frcnn_set_tuning / frcnn_get_tuning / frcnn_reset_tuning (ABI v22): the library's knobs live in one table filled from the FRCNN_* environment when the
    library is loaded; the environment is never read again (a later os.environ change has no effect), keys must start with FRCNN_, values are bounded, reset goes
    back to the load-time snapshot; the Python registry (chainer_faster_rcnn_amd.tuning) keeps its own copy in step."""
    import ctypes
    import chainer_faster_rcnn_amd as pkg
    tuning = pkg.tuning
    lib = pkg._lib.bind(pkg._lib.LIB_PATH)

    def get(key):
        buf = ctypes.create_string_buffer(96)
        n = lib.frcnn_get_tuning(key.encode(), buf, 96)
        return None if n == 0 else buf.value.decode()
    assert get("FRCNN_TEST_KNOB") is None
    os.environ["FRCNN_TEST_KNOB"] = "from-the-environment-after-load"
    try:
        assert get("FRCNN_TEST_KNOB") is None                                   # no entry point reads the environment after load
        assert lib.frcnn_conv_bf16_plan(256, 256, 150, 250, 3, 0) == 910
    finally:
        del os.environ["FRCNN_TEST_KNOB"]
    tuning.set("FRCNN_TEST_KNOB", "7")
    assert get("FRCNN_TEST_KNOB") == "7" and tuning.get("FRCNN_TEST_KNOB") == "7"
    with tuning.override(FRCNN_TEST_KNOB="8", FRCNN_BF16_STRIP="0"):
        assert get("FRCNN_TEST_KNOB") == "8" and lib.frcnn_conv_bf16_plan(256, 256, 150, 250, 3, 0) == 0
    assert get("FRCNN_TEST_KNOB") == "7" and lib.frcnn_conv_bf16_plan(256, 256, 150, 250, 3, 0) == 910
    assert lib.frcnn_set_tuning(b"NOT_OURS", b"1") == -1 and lib.frcnn_set_tuning(b"FRCNN_" + b"K" * 60, b"1") == -1
    assert lib.frcnn_set_tuning(b"FRCNN_TEST_KNOB", b"v" * 200) == -1 and get("FRCNN_TEST_KNOB") == "7"
    buf = ctypes.create_string_buffer(2)
    assert lib.frcnn_get_tuning(b"FRCNN_TEST_KNOB", buf, 2) == 2 and buf.value == b"7"
    tuning.set("FRCNN_TEST_KNOB", None)
    assert get("FRCNN_TEST_KNOB") is None
    tuning.set("FRCNN_TEST_KNOB", "9")
    tuning.reset()
    assert get("FRCNN_TEST_KNOB") is None and tuning.get("FRCNN_TEST_KNOB") is None
    with pytest.raises(ValueError):
        tuning.set("PATH", "x")


def test_import_survives_frcnn_variables_the_library_cannot_hold():
    """an unrelated FRCNN_* variable -- a long path, the space-separated sweep lists scripts/conv_bf16_sweep.py reads, a
    60-character name -- is skipped by the library's load-time snapshot and must be skipped, not raised on, by tuning.register(); a knob that
    fits still arrives.  A fresh interpreter, because the snapshot is taken at import."""
    import subprocess
    import sys
    env = dict(os.environ)
    env["FRCNN_BF16_DMAS"] = " ".join(str(900 + i) for i in range(40))            # 199 characters
    env["FRCNN_SOME_OUTPUT_DIRECTORY"] = "/tmp/" + "x" * 300
    env["FRCNN_" + "K" * 60] = "1"
    env["FRCNN_BF16_STRIP"] = "0"
    code = ("import ctypes, chainer_faster_rcnn_amd as pkg\n"
            "lib = pkg._lib.bind(pkg._lib.LIB_PATH)\n"
            "buf = ctypes.create_string_buffer(96)\n"
            "assert lib.frcnn_get_tuning(b'FRCNN_BF16_STRIP', buf, 96) == 2 and buf.value == b'0'\n"
            "assert lib.frcnn_get_tuning(b'FRCNN_BF16_DMAS', buf, 96) == 0 and pkg.tuning.get('FRCNN_BF16_DMAS') is None\n"
            "pkg.tuning.reset()\n"
            "assert pkg.tuning.get('FRCNN_BF16_STRIP') == '0'\n"
            "print('ok')\n")
    out = subprocess.run([sys.executable, "-W", "error", "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]


def test_conv_bf16_plan_of_the_vgg16_chain(monkeypatch):
    """frcnn_conv_bf16_plan (launch-free; a CU count of 256 is assumed where no device is visible): the default picks of the bf16 chain at
    600 x 1000 -- strip form D where a launch has >= 8 K-chunks and >= one 64-cout x 10-row x 32-px tile per CU, form C on the 38 x 63
    launches (not under the fused pool: five tile rows), conv_dma_bf16_kernel elsewhere -- and the hooks that switch the rule off."""
    import chainer_faster_rcnn_amd as pkg
    tuning = pkg.tuning
    lib = pkg._lib.bind(pkg._lib.LIB_PATH)
    for k in ("FRCNN_BF16_DMA", "FRCNN_BF16_STRIP", "FRCNN_BF16_SPLIT", "FRCNN_BF16_RP", "FRCNN_BF16_DMA_DEFAULT"):
        tuning.set(k, None)
    want = {(64, 64, 600, 1000, 2): 0, (64, 128, 300, 500, 0): 0, (128, 128, 300, 500, 2): 910, (128, 256, 150, 250, 0): 910, (256, 256, 150, 250, 0): 910,
            (256, 256, 150, 250, 2): 910, (256, 512, 75, 125, 0): 910, (512, 512, 75, 125, 2): 910, (512, 512, 38, 63, 0): 903, (512, 512, 38, 63, 2): 0,
            (3, 64, 600, 1000, 0): 0, (512, 512, 10, 14, 0): 0}               # 910 = form D (direct stores where the launch has no fused pool)
    for (ci, co, h, w, om), form in want.items():
        assert lib.frcnn_conv_bf16_plan(ci, co, h, w, 3, om) == form, (ci, co, h, w, om)
    assert lib.frcnn_conv_bf16_plan(512, 54, 38, 63, 1, 1) == 0                     # 1x1: the register-staged kernel
    assert lib.frcnn_conv_bf16_plan(512, 512, 38, 63, 5, 0) == -1                   # FRCNN_ERR_INVALID
    tuning.set("FRCNN_BF16_STRIP", "0")
    assert lib.frcnn_conv_bf16_plan(256, 256, 150, 250, 3, 0) == 0
    tuning.set("FRCNN_BF16_STRIP", None)
    tuning.set("FRCNN_BF16_DMA", "909")
    assert lib.frcnn_conv_bf16_plan(256, 256, 150, 250, 3, 0) == 909
    tuning.set("FRCNN_BF16_DMA", "901")                                              # form A: research builds only (-DFRCNN_TUNING_FORMS) -- refused, not substituted
    assert lib.frcnn_conv_bf16_plan(256, 256, 150, 250, 3, 0) == -1
    tuning.set("FRCNN_BF16_DMA", "921")
    assert lib.frcnn_conv_bf16_plan(64, 64, 600, 1000, 3, 0) == -1
    tuning.set("FRCNN_BF16_DMA", "141")
    assert lib.frcnn_conv_bf16_plan(256, 256, 150, 250, 3, 0) == 0
