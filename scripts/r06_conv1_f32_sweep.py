"""Round 6: workgroups per CU of the first-layer kernel's fp32 form (frcnn_conv1_f32: conv1_1 of the contract line, 154 MB written), hipGraph of 8 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import chainer_faster_rcnn_amd as pkg
from chainer_faster_rcnn_amd import tuning
from prop_bench import graph_us
rt = pkg.runtime.default_runtime()
rs = np.random.RandomState(0)
x = rt.mem.from_numpy((rs.randn(1, 3, 600, 1000) * 60).astype(np.float32))
wp = rt.pack_conv3x3_w(rt.mem.from_numpy((rs.randn(64, 3, 3, 3) * 0.27).astype(np.float32)))
b = rt.mem.from_numpy(np.zeros(64, np.float32))
outs = [rt.mem.empty((1, 64, 600, 1000), "f32") for _ in range(3)]
for rep in range(2):
    for per_cu in ("0", "1", "2", "3", "4", "5", "6", "8"):
        tuning.set("FRCNN_CONV1_WGS_PER_CU", None if per_cu == "0" else per_cu)
        st = {"i": 0}
        def f():
            for _ in range(8):
                rt.conv3x3(x, wp, b, relu=True, out=outs[st["i"] % 3]); st["i"] += 1
        us = graph_us(f, 8, replays=10)
        print("conv1_1 fp32 form, workgroups/CU %-7s %6.1f us  %5.2f TB/s" % (per_cu if per_cu != "0" else "default", us, (153.6e6 + 7.2e6) / us / 1e6), flush=True)
