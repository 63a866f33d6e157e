#!/usr/bin/env python
"""Time every conv work-decomposition (cfg id of frcnn_conv3x3_f32_cfg) on every VGG-16 / RPN layer shape at
600x1000 and print TFLOP/s.  GPU only.  Used to choose pick_conv_config() in csrc/conv.hip."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402

SHAPES = [("conv1_1", 3, 64, 600, 1000), ("conv1_2", 64, 64, 600, 1000), ("conv2_1", 64, 128, 300, 500),
          ("conv2_2", 128, 128, 300, 500), ("conv3_1", 128, 256, 150, 250), ("conv3_2", 256, 256, 150, 250),
          ("conv4_1", 256, 512, 75, 125), ("conv4_2", 512, 512, 75, 125), ("conv5_1", 512, 512, 38, 63)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", type=int, nargs="*", default=[0, 1, 2, 3, 4, 5, 8, 10, 11, 104, 105, 108, 110, 111, 205, 210, -1])
    ap.add_argument("--layers", nargs="*", default=None)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rt = pkg.runtime.default_runtime()
    res = {}
    for name, ci, co, h, w in SHAPES:
        if a.layers and name not in a.layers:
            continue
        rs = np.random.RandomState(0)
        x = rt.mem.from_numpy(rs.randn(1, ci, h, w).astype(np.float32))
        wt = rt.mem.from_numpy((rs.randn(co, ci, 3, 3) * 0.05).astype(np.float32))
        b = rt.mem.from_numpy(np.zeros(co, np.float32))
        wp = rt.pack_conv3x3_w(wt)
        y = rt.mem.empty((1, co, h, w), "f32")
        flops = 2.0 * h * w * co * ci * 9
        cfgs = []
        for cfg in a.cfgs:
            try:
                rt.conv3x3(x, wp, b, relu=True, out=y, cfg=cfg)
                cfgs.append(cfg)
            except ValueError:
                continue
        for _ in range(30):                                   # clocks and caches warm before anything is timed
            rt.conv3x3(x, wp, b, relu=True, out=y, cfg=cfgs[0])
        samples = {c: [] for c in cfgs}
        for _ in range(a.rounds):                             # interleaved rounds: variants see the same chip state
            for cfg in cfgs:
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    rt.conv3x3(x, wp, b, relu=True, out=y, cfg=cfg)
                e1.record()
                torch.cuda.synchronize()
                samples[cfg].append(e0.elapsed_time(e1) / a.iters)
        row = {c: round(flops / (float(np.median(v)) * 1e-3) / 1e12, 1) for c, v in samples.items()}
        res[name] = row
        print(name, " ".join("%d:%.1f" % (k, v) for k, v in row.items()), flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"))


if __name__ == "__main__":
    main()
