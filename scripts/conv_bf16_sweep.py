#!/usr/bin/env python
"""Time the bf16 conv kernel on the VGG-16 layer shapes at 600x1000 (TFLOP/s).  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402
from chainer_faster_rcnn_amd import tuning as _tuning  # noqa: E402  (knobs go through frcnn_set_tuning, not the environment)

SHAPES = [("conv1_1", 3, 64, 600, 1000), ("conv1_2", 64, 64, 600, 1000), ("conv2_1", 64, 128, 300, 500),
          ("conv2_2", 128, 128, 300, 500), ("conv3_1", 128, 256, 150, 250), ("conv3_2", 256, 256, 150, 250),
          ("conv4_1", 256, 512, 75, 125), ("conv4_2", 512, 512, 75, 125), ("conv5_1", 512, 512, 38, 63)]


def main():
    """FRCNN_BF16_ABLS="0 1 2 3 4 7": repeat the sweep with each timing ablation of the 3x3 kernel (see conv_bf16.hip)."""
    for abl in os.environ.get("FRCNN_BF16_ABLS", "0").split():
        for dma in os.environ.get("FRCNN_BF16_DMAS", "").split() or [os.environ.get("FRCNN_BF16_DMA", "-1")]:
            _tuning.set("FRCNN_BF16_ABL", abl)
            _tuning.set("FRCNN_BF16_DMA", dma)
            print("abl", abl, "dma", dma, end="  ")
            sweep()


def sweep():
    rt = pkg.runtime.default_runtime()
    blk_a = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    blk_b = torch.empty_like(blk_a)
    out = []
    for name, ci, co, h, w in SHAPES:
        rs = np.random.RandomState(0)
        x = rt.bf16_from_nchw(rt.mem.from_numpy(rs.randn(1, ci, h, w).astype(np.float32)))
        wt = rt.bf16_pack_conv_w(rt.mem.from_numpy((rs.randn(co, ci, 3, 3) * 0.05).astype(np.float32)), 3)
        b = rt.mem.from_numpy(np.zeros(co, np.float32))
        fn = lambda: rt.conv_bf16(x, wt, b, ci, co, 3, relu=True)
        for _ in range(20):
            fn()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            for _ in range(20):
                blk_b.copy_(blk_a)                       # queue ahead of the host so launch cost is hidden
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        ms = float(np.median(ts))
        out.append("%s %.0f us %.0f TF" % (name, ms * 1e3, 2.0 * h * w * co * ci * 9 / (ms * 1e-3) / 1e12))
    print("  ".join(out))


if __name__ == "__main__":
    main()
