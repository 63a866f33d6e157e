#!/usr/bin/env python
"""Time the fp32 FC kernel on the RCNN-head shapes (TFLOP/s); FRCNN_LINEAR_NODMA=1 selects the register-staged kernel.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402
from chainer_faster_rcnn_amd import tuning as _tuning  # noqa: E402  (knobs go through frcnn_set_tuning, not the environment)


def main():
    rt = pkg.runtime.default_runtime()
    blk_a = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    blk_b = torch.empty_like(blk_a)
    variants = ("dma", "reg", "dma", "reg")
    if "--splits" in sys.argv:                 # round 6: split-K factor sweep of the LDS-DMA kernel (FRCNN_LINEAR_F32_SPLITS)
        variants = tuple("s%s" % v for v in sys.argv[sys.argv.index("--splits") + 1].split(",")) * 2
    for variant in variants:
        if variant.startswith("s"):
            _tuning.set("FRCNN_LINEAR_NODMA", None)
            _tuning.set("FRCNN_LINEAR_F32_SPLITS", None if variant == "s0" else variant[1:])
            rt._ws.pop("linear", None)
        elif variant == "reg":
            _tuning.set("FRCNN_LINEAR_NODMA", "1")
        else:
            _tuning.set("FRCNN_LINEAR_NODMA", None)
        out = []
        for name, M, N, K in [("fc6", 300, 4096, 25088), ("fc7", 300, 4096, 4096), ("bbox", 300, 84, 4096)]:
            rs = np.random.RandomState(0)
            x = rt.mem.from_numpy(rs.randn(M, K).astype(np.float32))
            w = rt.mem.from_numpy((rs.randn(N, K) * 0.01).astype(np.float32))
            b = rt.mem.from_numpy(np.zeros(N, np.float32))
            y = rt.mem.empty((M, N), "f32")
            fn = lambda: rt.linear(x, w, b, relu=True, out=y)
            for _ in range(10):
                fn()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                for _ in range(20):
                    blk_b.copy_(blk_a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5)
            ms = float(np.median(ts))
            out.append("%s %.0f us %.1f TF" % (name, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12))
        print(variant, "  ".join(out), flush=True)


if __name__ == "__main__":
    main()
