"""Run the reference's OWN Python hot-path files under a stub `chainer` (TEST INFRASTRUCTURE ONLY).

Chainer / CuPy / OpenCV are not installable here, but the reference's detection glue
(models/proposal_layer.py, bbox_transform.py, generate_anchors.py, anchor_target_layer.py,
proposal_target_layer.py) only touches a tiny part of the chainer API.  This module
fabricates that surface in `sys.modules`, restores the NumPy aliases the reference still
uses (np.float / np.int, removed in NumPy 1.24), plugs in the Cython modules built by
oracle/build_ref.py, and imports the reference modules *from /root/reference* (never
copied).  It only works inside the build container; the GPU box has no /root/reference and
relies on the fixtures in tests/golden/ produced by tests/make_golden.py.

Only tests/make_golden.py, tests/ and bench.py's cpu_baseline leg may import this.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np

REF = os.environ.get("FRCNN_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REF, "models"))


def _restore_numpy_aliases():
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int


class Variable(object):
    """Duck-typed stand-in for chainer.Variable (v1): .data plus shape/dtype/ndim."""

    def __init__(self, data, volatile=False):
        self.data = data

    shape = property(lambda s: s.data.shape)
    dtype = property(lambda s: s.data.dtype)
    ndim = property(lambda s: s.data.ndim)

    def __len__(self):
        return len(self.data)


class _Dev(object):
    id = -1

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _install_stub():
    if "chainer" in sys.modules and getattr(sys.modules["chainer"], "_frcnn_stub", False):
        return
    ch = types.ModuleType("chainer")
    ch._frcnn_stub = True
    ch.Variable = Variable
    cuda = types.ModuleType("chainer.cuda")

    class _CupyNd(object):
        pass
    cupy = types.ModuleType("cupy")
    cupy.ndarray = _CupyNd
    cuda.cupy = cupy
    cuda.available = False

    def get_array_module(*args):
        return np
    cuda.get_array_module = get_array_module
    cuda.get_device_from_array = lambda *a: _Dev()
    cuda.to_cpu = lambda x, *a, **k: x
    cuda.to_gpu = lambda x, *a, **k: x
    ch.cuda = cuda
    sys.modules["chainer"] = ch
    sys.modules["chainer.cuda"] = cuda


def _load_native(name):
    import sysconfig
    path = os.path.join(HERE, "_ref", name + sysconfig.get_config_var("EXT_SUFFIX"))
    spec = importlib.util.spec_from_file_location("models." + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def native(name):
    """The reference's Cython module (`cpu_nms` or `bbox`) from oracle/_ref — works on the GPU box too."""
    _restore_numpy_aliases()
    return _load_native(name)


_cache = {}


def load():
    """Returns a namespace with the reference's hot-path callables."""
    if "ns" in _cache:
        return _cache["ns"]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    from oracle import build_ref
    build_ref.build()
    _restore_numpy_aliases()
    _install_stub()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # `models` here is the REFERENCE's package (/root/reference/models/__init__.py)
    models = importlib.import_module("models")
    assert os.path.realpath(os.path.dirname(models.__file__)).startswith(os.path.realpath(REF))
    sys.modules["models.cpu_nms"] = _load_native("cpu_nms")
    sys.modules["models.bbox"] = _load_native("bbox")
    g = types.ModuleType("models.gpu_nms")          # dead code in the reference (proposal_layer.py:180-187)
    g.gpu_nms = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("gpu_nms is dead code"))
    sys.modules["models.gpu_nms"] = g
    ns = types.SimpleNamespace()
    ns.Variable = Variable
    ns.cpu_nms = sys.modules["models.cpu_nms"].cpu_nms
    ns.bbox_overlaps = sys.modules["models.bbox"].bbox_overlaps
    ns.generate_anchors = importlib.import_module("models.generate_anchors").generate_anchors
    bt = importlib.import_module("models.bbox_transform")
    ns.bbox_transform = bt.bbox_transform
    ns.bbox_transform_inv = bt.bbox_transform_inv
    ns.clip_boxes = bt.clip_boxes
    ns.filter_boxes = bt.filter_boxes
    ns.keep_inside = bt.keep_inside
    ns.ProposalLayer = importlib.import_module("models.proposal_layer").ProposalLayer
    ns.AnchorTargetLayer = importlib.import_module("models.anchor_target_layer").AnchorTargetLayer
    ns.ProposalTargetLayer = importlib.import_module("models.proposal_target_layer").ProposalTargetLayer
    _cache["ns"] = ns
    return ns
