#!/usr/bin/env python
"""Time the proposal pipeline (decode -> tile sort -> rank -> NMS mask -> NMS scan) alone on the benchmark's own RPN maps
(600x1000 synthetic image).  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split; prints the end-to-end
HIP-event time per call of a hipGraph holding 8 calls.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402
from chainer_faster_rcnn_amd import tuning as _tuning  # noqa: E402  (knobs go through frcnn_set_tuning, not the environment)
from chainer_faster_rcnn_amd import synthetic  # noqa: E402
from chainer_faster_rcnn_amd.models import FasterRCNN  # noqa: E402


def graph_us(fn, per, replays=40):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (replays * per)


def main():
    rt = pkg.runtime.default_runtime()
    model = FasterRCNN(runtime=rt)
    model.load_params(synthetic.params(seed=1))
    out = model.forward_device(rt.mem.from_numpy(synthetic.image(seed=0)), 600, 1000, keep=True)
    prob, bbox = out["rpn_cls_prob"], out["rpn_bbox_pred"]
    pl = model.RPN.proposal_layer
    rois, probs, n_out, src = pl.forward_device(prob, bbox, 600, 1000, want_index=True)
    # where in the score order does the 300th survivor sit? (how many 64-box chunks the sequential scan has to visit)
    from oracle import frcnn_oracle as O
    p2, s2, d2 = O.proposal_layer(rt.mem.to_numpy(prob), rt.mem.to_numpy(bbox), np.array([[600, 1000]], np.int32), return_debug=True)
    keep = d2["keep"]
    print("n_valid_sorted=%d kept=%d last kept position=%d (chunk %d of %d)" % (len(d2["sorted_boxes"]), len(keep), keep[-1], keep[-1] // 64,
                                                                             (len(d2["sorted_boxes"]) + 63) // 64))
    print("kept per 256-box super-chunk:", np.bincount(np.asarray(keep) // 256).tolist())

    def seq():
        for _ in range(8):
            pl.forward_device(prob, bbox, 600, 1000)
    for scan in ("0", "1"):
        _tuning.set("FRCNN_NMS_SCAN", scan)
        print("FRCNN_NMS_SCAN=%s: %.1f us per proposals call (test mode 6000 -> 300)" % (scan, graph_us(seq, 8)))
    _tuning.set("FRCNN_NMS_SCAN", "0")
    pl.train = True

    def seq_train():
        for _ in range(4):
            pl.forward_device(prob, bbox, 600, 1000)
    print("train mode (12000 -> 2000): %.1f us per call" % graph_us(seq_train, 4, 20))


if __name__ == "__main__":
    main()
