#!/usr/bin/env python
"""Timing ablations of roi_pool_cells_kernel on the benchmark's own map and RoIs (FRCNN_ROI_DBG bits: 1 arrival order instead of
longest-first, 2 prologue only, 4 no scan, 8 no output).  Needs a tuning build of the library (FRCNN_TIMING_ABLATIONS=1 python
chainer-faster-rcnn_amd/csrc/build.py --force: see scripts/r02_gpu_l.sh); the shipped build ignores FRCNN_ROI_DBG.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402
from chainer_faster_rcnn_amd import tuning as _tuning  # noqa: E402  (knobs go through frcnn_set_tuning, not the environment)
from chainer_faster_rcnn_amd import synthetic  # noqa: E402
from chainer_faster_rcnn_amd.models import FasterRCNN  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prop_bench import graph_us  # noqa: E402


def main():
    rt = pkg.runtime.default_runtime()
    model = FasterRCNN(runtime=rt)
    model.load_params(synthetic.params(seed=1))
    out = model.forward_device(rt.mem.from_numpy(synthetic.image(seed=0)), 600, 1000, keep=True)
    feat, rois = out["feat"], out["rois"]
    R, C = int(rois.shape[0]), int(feat.shape[1])
    ys = [rt.mem.empty((R, C, 7, 7), "f32") for _ in range(10)]

    def seq():
        for i in range(10):
            rt.roi_pool_fwd_chw(feat, rois, 7, 7, 1 / 16., out=ys[i])
    # A/B against another build of roi_pool.hip (scripts/_ab/libroi_old.so, built by hand from an earlier revision), same pointers
    import ctypes
    old_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", "libroi_old.so")
    old = ctypes.CDLL(old_path) if os.path.exists(old_path) else None
    m = rt.mem
    H, W = int(feat.shape[2]), int(feat.shape[3])

    def seq_old():
        for i in range(10):
            old.frcnn_roi_pool_fwd_chw(m.ptr(feat), C, H, W, m.ptr(rois), R, int(rois.shape[1]), 7, 7, ctypes.c_float(1 / 16.), m.ptr(ys[i]),
                                       None, None, ctypes.c_size_t(0), m.stream())
    for rep in range(2):
        if old is not None:
            print("previous revision: %.2f us" % graph_us(seq_old, 10))
        for dbg in (0, 1, 2, 4, 8, 12):
            _tuning.set("FRCNN_ROI_DBG", str(dbg))
            print("FRCNN_ROI_DBG=%2d: %.2f us" % (dbg, graph_us(seq, 10)))


if __name__ == "__main__":
    main()
