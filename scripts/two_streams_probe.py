#!/usr/bin/env python
"""Two images in flight per GPU (two model instances, two captured graphs, two HIP streams): throughput against one graph on one stream, with the outputs of the
concurrent replays compared with the serial ones.  GPU only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402
from chainer_faster_rcnn_amd import synthetic  # noqa: E402
from chainer_faster_rcnn_amd.graph import CapturedForward  # noqa: E402
from chainer_faster_rcnn_amd.models import FasterRCNN  # noqa: E402


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    n_inst = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    params = synthetic.params(seed=1)
    x = [None] * n_inst
    graphs, rts = [], []
    for i in range(n_inst):
        rt = pkg.runtime.Runtime(pkg._lib.load(), pkg.runtime.TorchDeviceMemory("cuda:0"))
        m = FasterRCNN(runtime=rt, conv_dtype=dtype, head_dtype=dtype)
        m.load_params(params)
        x[i] = rt.mem.from_numpy(synthetic.image(seed=i, h=600, w=1000))
        for _ in range(3):
            m.forward_device(x[i], 600, 1000)
        torch.cuda.synchronize()
        graphs.append(CapturedForward(m, x[i], 600, 1000, warmup=1))
        rts.append(rt)
    torch.cuda.synchronize()
    ref = []
    for g in graphs:                                  # serial reference outputs
        o = g.replay()
        torch.cuda.synchronize()
        ref.append({k: o[k].clone() for k in ("rois", "cls_prob", "pred_boxes", "n_out")})
    streams = [torch.cuda.Stream() for _ in range(n_inst)]

    def timed(fn, steps):
        fn(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        return steps / (time.perf_counter() - t0)

    def serial(k):
        for i in range(k):
            graphs[0].graph.replay()

    def rr(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % n_inst]):
                graphs[i % n_inst].graph.replay()
    for rep in range(3):
        a = timed(serial, 1000)
        b = timed(rr, 1000)
        print("%s: one graph, one stream %.1f img/s; %d instances round-robin on %d streams %.1f img/s (x %.3f)" % (dtype, a, n_inst, n_inst, b, b / a))
    ok = True
    for i, g in enumerate(graphs):
        for k in ("rois", "cls_prob", "pred_boxes", "n_out"):
            ok = ok and bool(torch.equal(g.out[k], ref[i][k]))
    print("outputs after the concurrent replays identical to the serial ones:", ok)


if __name__ == "__main__":
    main()
