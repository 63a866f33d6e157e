#!/usr/bin/env python
"""Error of the fp32 MFMA kernels and of the split-product kernels against FLOAT64 results of the same fp32 operands (max |err| / max |ref|):
convolution forward, weight gradient, fully connected layer, at VGG-16 shapes.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402


def rel(a, ref):
    return float(np.abs(a - ref).max() / np.abs(ref).max())


def main():
    rt = pkg.runtime.default_runtime()
    rs = np.random.RandomState(0)
    rows = []
    for name, ci, co, h, w in (("conv2_2", 128, 128, 75, 125), ("conv3_2", 256, 256, 75, 125), ("conv4_2", 512, 512, 38, 63)):
        x = np.maximum(rs.randn(1, ci, h, w), 0).astype(np.float32)
        wt = (rs.randn(co, ci, 3, 3) * np.sqrt(2.0 / (ci * 9))).astype(np.float32)
        b = (rs.randn(co) * 0.1).astype(np.float32)
        dy = (rs.randn(1, co, h, w) * 0.1).astype(np.float32)
        t = lambda a: torch.from_numpy(a).double()
        ref = torch.nn.functional.conv2d(t(x), t(wt), t(b), padding=1).numpy()
        refw = torch.nn.grad.conv2d_weight(t(x), (co, ci, 3, 3), t(dy), padding=1).numpy().reshape(co, ci * 9).T
        torch32 = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(b), padding=1).numpy()
        d = rt.mem.from_numpy
        nat = rt.mem.to_numpy(rt.conv3x3(d(x), rt.pack_conv3x3_w(d(wt)), d(b), relu=False))
        spl = rt.mem.to_numpy(rt.conv3x3_f32s(rt.f32s_from_nchw(d(x)), rt.f32s_pack_conv_w(d(wt)), d(b), ci, co, relu=False, out_f32_nchw=True))
        natw = rt.mem.to_numpy(rt.conv_wgrad(d(x), d(dy), 3))
        splw = rt.mem.to_numpy(rt.conv_wgrad_f32s(d(x), d(dy)))
        rows.append((name + " forward (K = %d)" % (ci * 9), rel(nat, ref), rel(spl, ref), rel(torch32, ref)))
        rows.append((name + " weight gradient (K = %d px)" % (h * w), rel(natw, refw), rel(splw, refw), None))
    for name, m, n, k in (("fc7", 300, 4096, 4096), ("fc6 (N = 512 of 4096)", 300, 512, 25088)):
        x = np.maximum(rs.randn(m, k), 0).astype(np.float32)
        wt = (rs.randn(n, k) / np.sqrt(k)).astype(np.float32)
        b = (rs.randn(n) * 0.1).astype(np.float32)
        ref = x.astype(np.float64) @ wt.astype(np.float64).T + b
        d = rt.mem.from_numpy
        nat = rt.mem.to_numpy(rt.linear(d(x), d(wt), d(b)))
        spl = rt.mem.to_numpy(rt.linear_f32s(rt.f32s_split(d(x)), rt.f32s_split(d(wt)), d(b)))
        rows.append((name + " (K = %d)" % k, rel(nat, ref), rel(spl, ref), rel((torch.from_numpy(x) @ torch.from_numpy(wt).T + torch.from_numpy(b)).numpy(), ref)))
    print("%-44s %12s %12s %12s" % ("max |err| / max |float64 result|", "fp32 MFMA", "split bf16x6", "torch CPU fp32"))
    for r in rows:
        print("%-44s %12.2e %12.2e %12s" % (r[0], r[1], r[2], "%.2e" % r[3] if r[3] is not None else "-"))


if __name__ == "__main__":
    main()
