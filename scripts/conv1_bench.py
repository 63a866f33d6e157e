#!/usr/bin/env python
"""Timing of the first-layer kernels (conv1_f32s_kernel, csrc/conv_f32s.hip: split output, plain bf16 output, the trainers' dual
output) on the 600x1000 image, over workgroups-per-CU settings of the persistent launch.  hipGraph of 8 launches, HIP events.
GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402
from chainer_faster_rcnn_amd import tuning as _tuning  # noqa: E402  (knobs go through frcnn_set_tuning, not the environment)
from prop_bench import graph_us  # noqa: E402


def main():
    rt = pkg.runtime.default_runtime()
    rs = np.random.RandomState(0)
    h, w = 600, 1000
    x = rt.mem.from_numpy((rs.randn(1, 3, h, w) * 60).astype(np.float32))
    wt = rt.mem.from_numpy((rs.randn(64, 3, 3, 3) * 0.27).astype(np.float32))
    wp = rt.pack_conv3x3_w(wt)
    b = rt.mem.from_numpy(np.zeros(64, np.float32))
    forms = {"bf16 (77 MB out)": (lambda: rt.conv1_bf16(x, wt, b, relu=True), 77e6 + 7.2e6),
             "split (230 MB out)": (lambda: rt.conv1_f32s(x, wt, b, relu=True), 230e6 + 7.2e6),
             "split + fp32 NCHW (384 MB out)": (lambda: rt.conv1_f32s_train(x, wp, b, 64, relu=True), 384e6 + 7.2e6)}
    for name, (fn, nbytes) in forms.items():
        for per_cu in os.environ.get("PER_CU", "0,1,2,3,4,6,8").split(","):
            if per_cu == "0":
                _tuning.set("FRCNN_CONV1_WGS_PER_CU", None)
            else:
                _tuning.set("FRCNN_CONV1_WGS_PER_CU", per_cu)

            def f():
                for _ in range(8):
                    fn()
            us = graph_us(f, 8, replays=10)
            print("%-32s workgroups/CU %-8s %7.1f us  %5.2f TB/s" % (name, per_cu if per_cu != "0" else "default", us, nbytes / us / 1e6), flush=True)
    _tuning.set("FRCNN_CONV1_WGS_PER_CU", None)


if __name__ == "__main__":
    main()
