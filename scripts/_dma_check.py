import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import chainer_faster_rcnn_amd as pkg
rt = pkg.runtime.default_runtime()
rs = np.random.RandomState(0)
for (ci, co, h, w) in [(64, 64, 600, 1000), (256, 256, 150, 250), (512, 512, 75, 125), (512, 512, 38, 63), (3, 64, 37, 70)]:
    x = rt.bf16_from_nchw(rt.mem.from_numpy(rs.randn(1, ci, h, w).astype(np.float32)))
    wt = rt.bf16_pack_conv_w(rt.mem.from_numpy((rs.randn(co, ci, 3, 3) * 0.05).astype(np.float32)), 3)
    b = rt.mem.from_numpy(rs.randn(co).astype(np.float32))
    outs = []
    for mode in ["0", "231", "141", "-1"]:
        os.environ["FRCNN_BF16_DMA"] = mode
        outs.append(rt.mem.to_numpy(rt.conv_bf16(x, wt, b, ci, co, 3, relu=True)))
        outs.append(rt.mem.to_numpy(rt.conv_bf16(x, wt, b, ci, co, 3, relu=True, pool=True)))
    torch.cuda.synchronize()
    print(ci, co, h, w, "bit-equal", [bool(np.array_equal(outs[0], outs[2 * i])) and bool(np.array_equal(outs[1], outs[2 * i + 1])) for i in range(1, 4)], flush=True)
os.environ["FRCNN_BF16_DMA"] = "0"
