"""chainer-faster-rcnn_amd: the MI355X-native Faster R-CNN hot path behind the reference's models/ surface.

  csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/frcnn_hip.h) -> libfrcnn_hip.so
  _lib.py    ctypes binding of that ABI (no fallback: raises if the library or the GPU is missing)
  tuning.py  the tuning registry (frcnn_set_tuning): A/B knobs are set through the ABI, never read from the environment at launch time
  runtime.py device memory / stream plumbing (PyTorch-ROCm) + typed wrappers over each entry point
  graph.py   hipGraph capture / replay of the inference forward (CapturedForward)
  train.py   the RPN training step (forward, anchor targets, losses, backward, one all-reduce, fused SGD update)
  models/    host-side mirror of the reference's models/ package (ProposalLayer, AnchorTargetLayer,
             cpu_nms, bbox, roi_pooling_2d, VGG16Prev, RegionProposalNetwork, FasterRCNN)
"""
from . import _lib  # noqa: F401
from . import tuning  # noqa: F401
from . import runtime  # noqa: F401

__version__ = "0.1.0"
