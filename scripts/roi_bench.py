#!/usr/bin/env python
"""Time RoIPooling2D forward alone on the benchmark's own feature map and RoIs (600x1000 synthetic image):
algorithmic bytes (4.90 MB map + 30.11 MB output + rois) / HIP-event time.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402
from chainer_faster_rcnn_amd import tuning as _tuning  # noqa: E402  (knobs go through frcnn_set_tuning, not the environment)
from chainer_faster_rcnn_amd import synthetic  # noqa: E402
from chainer_faster_rcnn_amd.models import FasterRCNN  # noqa: E402


def timeit(fn, iters=8, rounds=7):
    """GPU time per call: the calls are queued behind a ~1 ms blocker so the host's launch cost (tens of us per
    Python call) is hidden and the two events bracket back-to-back kernels."""
    blk_a = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    blk_b = torch.empty_like(blk_a)
    for _ in range(5):
        fn()
    out = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        for _ in range(40):
            blk_b.copy_(blk_a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    return float(np.median(out))


def main():
    rt = pkg.runtime.default_runtime()
    model = FasterRCNN(runtime=rt)
    model.load_params(synthetic.params(seed=1))
    out = model.forward_device(rt.mem.from_numpy(synthetic.image(seed=0)), 600, 1000, keep=True)
    feat, rois = out["feat"], out["rois"]
    C, H, W = [int(v) for v in feat.shape[1:]]
    R = int(rois.shape[0])
    y = rt.mem.empty((R, C, 7, 7), "f32")
    nbytes = (C * H * W + R * C * 49) * 4 + R * 16
    _tuning.set("FRCNN_ROI_KERNEL", "cells")
    us = timeit(lambda: rt.roi_pool_fwd_chw(feat, rois, 7, 7, 1 / 16., out=y))
    print("cells (NCHW, b128) %.1f us  %.0f GB/s  (%.1f %% of 8 TB/s)" % (us, nbytes / us / 1e3, nbytes / us / 1e3 / 80))
    yb = rt.roi_pool_fwd_chw_bf16(feat, rois, 7, 7, 1 / 16.)
    us = timeit(lambda: rt.roi_pool_fwd_chw_bf16(feat, rois, 7, 7, 1 / 16.))
    print("cells, bf16 out    %.1f us" % us)
    _tuning.set("FRCNN_ROI_KERNEL", "planes")
    us = timeit(lambda: rt.roi_pool_fwd_chw(feat, rois, 7, 7, 1 / 16., out=y))
    print("planes (NCHW)      %.1f us  %.0f GB/s  (%.1f %% of 8 TB/s)" % (us, nbytes / us / 1e3, nbytes / us / 1e3 / 80))
    _tuning.set("FRCNN_ROI_KERNEL", "cells")
    us = timeit(lambda: rt.roi_pool_fwd_chw(feat, rois, 7, 7, 1 / 16., want_argmax=True, out=y))
    print("planes + argmax    %.1f us" % us)
    xt = rt.chw_to_hwc(feat)
    us = timeit(lambda: rt.roi_pool_fwd_hwc(xt, C, H, W, rois, 7, 7, 1 / 16., out=y))
    print("channel-last gather %.1f us (+ transpose)" % us)
    us = timeit(lambda: y.copy_(y))  # noqa
    z = torch.empty_like(y)
    us = timeit(lambda: z.copy_(y))
    print("30 MB d2d copy     %.1f us (read+write)" % us)


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def sweep():
    rt = pkg.runtime.default_runtime()
    model = FasterRCNN(runtime=rt)
    model.load_params(synthetic.params(seed=1))
    out = model.forward_device(rt.mem.from_numpy(synthetic.image(seed=0)), 600, 1000, keep=True)
    feat, rois = out["feat"], out["rois"]
    for R in (16, 64, 128, 200, 300):
        r = rois[:R].contiguous()
        y = rt.mem.empty((R, 512, 7, 7), "f32")
        us = timeit(lambda: rt.roi_pool_fwd_chw(feat, r, 7, 7, 1 / 16., out=y))
        print("R=%d  %.1f us" % (R, us))
    tiny = rois[:300].clone()
    tiny[:, 2] = tiny[:, 0] + 15
    tiny[:, 3] = tiny[:, 1] + 15                      # 1-2 cells per RoI: fixed costs only
    y = rt.mem.empty((300, 512, 7, 7), "f32")
    print("R=300 tiny RoIs %.1f us" % timeit(lambda: rt.roi_pool_fwd_chw(feat, tiny, 7, 7, 1 / 16., out=y)))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sweep":
    sweep()
