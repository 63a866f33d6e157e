#!/usr/bin/env python
"""Per-layer timing of the split-product weight gradient (csrc/train.hip conv_wgrad_f32s_kernel) against the fp32 MFMA kernel on the
VGG-16 layer shapes of a 600x1000 image.  hipGraph of 4 launches each.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402
from chainer_faster_rcnn_amd import tuning as _tuning  # noqa: E402  (knobs go through frcnn_set_tuning, not the environment)
from prop_bench import graph_us  # noqa: E402

SHAPES = [("conv1_1", 3, 64, 600, 1000), ("conv1_2", 64, 64, 600, 1000), ("conv2_1", 64, 128, 300, 500), ("conv2_2", 128, 128, 300, 500),
          ("conv3_1", 128, 256, 150, 250), ("conv3_2", 256, 256, 150, 250), ("conv4_1", 256, 512, 75, 125), ("conv4_2", 512, 512, 75, 125),
          ("conv5_1", 512, 512, 38, 63)]


def main():
    rt = pkg.runtime.default_runtime()
    rs = np.random.RandomState(0)
    only = sys.argv[1:] or None
    for name, ci, co, h, w in SHAPES:
        if only and name not in only:
            continue
        x = rt.mem.from_numpy(np.maximum(rs.randn(1, ci, h, w), 0).astype(np.float32))
        dy = rt.mem.from_numpy((rs.randn(1, co, h, w) * 0.1).astype(np.float32))
        out = rt.mem.empty((ci * 9, co), "f32")
        res = {}
        for sp in os.environ.get("SPLITS", "0").split(","):
            if sp != "0":
                _tuning.set("FRCNN_WGRAD_F32S_SPLITS", sp)
            else:
                _tuning.set("FRCNN_WGRAD_F32S_SPLITS", None)
            res[sp] = graph_us(lambda: [rt.conv_wgrad_f32s(x, dy, out=out) for _ in range(4)], 4, replays=8)
        us_n = graph_us(lambda: [rt.conv_wgrad(x, dy, 3, out=out) for _ in range(4)], 4, replays=8)
        gf = 2.0 * ci * co * 9 * h * w / 1e9
        print("%-8s %4d->%4d %4dx%-4d  split %s us (best %.1f TFLOP/s fp32-equivalent)   native %7.1f us (%5.1f TFLOP/s)" %
              (name, ci, co, h, w, " ".join("%s:%.1f" % kv for kv in res.items()), gf / min(res.values()) * 1e3, us_n, gf / us_n * 1e3), flush=True)
        del x, dy
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
