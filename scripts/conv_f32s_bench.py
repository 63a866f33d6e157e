#!/usr/bin/env python
"""Per-layer timing of the split-bf16 fp32 convolution (csrc/conv_f32s.hip) against the native fp32 MFMA kernel on the VGG-16
layer shapes of a 600x1000 image.  hipGraph of 4 launches each, HIP events.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402
from chainer_faster_rcnn_amd import tuning as _tuning  # noqa: E402  (knobs go through frcnn_set_tuning, not the environment)
from prop_bench import graph_us  # noqa: E402

SHAPES = [("conv1_1", 3, 64, 600, 1000, False), ("conv1_2", 64, 64, 600, 1000, True), ("conv2_1", 64, 128, 300, 500, False),
          ("conv2_2", 128, 128, 300, 500, True), ("conv3_1", 128, 256, 150, 250, False), ("conv3_2", 256, 256, 150, 250, False),
          ("conv3_3", 256, 256, 150, 250, True), ("conv4_1", 256, 512, 75, 125, False), ("conv4_2", 512, 512, 75, 125, False),
          ("conv4_3", 512, 512, 75, 125, True), ("conv5_1", 512, 512, 38, 63, False)]


def main():
    rt = pkg.runtime.default_runtime()
    rs = np.random.RandomState(0)
    only = sys.argv[1:] or None
    tot_s = tot_n = 0.0
    for name, ci, co, h, w, pool in SHAPES:
        if only and name not in only:
            continue
        x = rt.mem.from_numpy(rs.randn(1, ci, h, w).astype(np.float32))
        wt = rt.mem.from_numpy((rs.randn(co, ci, 3, 3) * np.sqrt(2.0 / (ci * 9))).astype(np.float32))
        b = rt.mem.from_numpy(np.zeros(co, np.float32))
        xs = rt.f32s_from_nchw(x)
        ws = rt.f32s_pack_conv_w(wt)
        wn = rt.pack_conv3x3_w(wt)

        def f_split():
            for _ in range(4):
                rt.conv3x3_f32s(xs, ws, b, ci, co, relu=True, pool=pool)

        def f_native():
            for _ in range(4):
                if pool:
                    rt.conv_ex(x, wn, b, 3, act=4)
                else:
                    rt.conv3x3(x, wn, b, relu=True)
        for abl in os.environ.get("ABLS", "").split(","):
            if abl:
                _tuning.set("FRCNN_F32S_ABL", abl)
                print("   ablation %s: %.1f us" % (abl, graph_us(f_split, 4, replays=10)))
        _tuning.set("FRCNN_F32S_ABL", "0")
        if os.environ.get("XCD_AB"):
            _tuning.set("FRCNN_F32S_XCD", "0")
            print("   linear tile order: %.1f us" % graph_us(f_split, 4, replays=10))
            _tuning.set("FRCNN_F32S_XCD", None)
        us_s = graph_us(f_split, 4, replays=10)
        us_n = graph_us(f_native, 4, replays=10)
        gf = 2.0 * ci * co * 9 * h * w / 1e9
        mult = 4 if name.startswith("conv5") else 1            # conv5_1..3 + rpn_conv_3x3 share the shape
        tot_s += us_s * mult
        tot_n += us_n * mult
        print("%-8s %4d->%4d %4dx%-4d pool=%d  split %7.1f us (%6.1f TFLOP/s fp32-equivalent, %5.1f %% of the bf16 MFMA peak)   native %7.1f us (%6.1f TFLOP/s)  x%.2f"
              % (name, ci, co, h, w, pool, us_s, gf / us_s * 1e3, 6 * gf / us_s * 1e3 / 2500 * 100, us_n, gf / us_n * 1e3, us_n / us_s), flush=True)
        del x, xs
        torch.cuda.empty_cache()
    print("14-launch chain: split %.0f us, native %.0f us" % (tot_s, tot_n))


if __name__ == "__main__":
    main()
