#!/usr/bin/env python
"""Time the ResNet-101 trunk alone and the config-4 FasterRCNN (ResNet-101, 1000/300 proposals) at 600x1000 (hipGraph replay).  GPU only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402
from chainer_faster_rcnn_amd import synthetic  # noqa: E402
from chainer_faster_rcnn_amd.graph import CapturedForward  # noqa: E402
from chainer_faster_rcnn_amd.models import FasterRCNN, ResNet101  # noqa: E402


def main():
    rt = pkg.runtime.default_runtime()
    h, w = 600, 1000
    params = synthetic.resnet_params(101, seed=2)
    rs = np.random.RandomState(3)
    head = synthetic.params(seed=1, rpn_ch=512, roi_feat=2048 * 49)
    for k in ("fc6", "fc7", "cls_score", "bbox_pred"):
        params[k + "/W"], params[k + "/b"] = head[k + "/W"], head[k + "/b"]
    params["RPN/rpn_conv_3x3/W"] = (rs.randn(512, 2048, 3, 3) * 0.01).astype(np.float32)
    params["RPN/rpn_conv_3x3/b"] = np.zeros(512, np.float32)
    for k in ("rpn_cls_score", "rpn_bbox_pred"):
        params["RPN/%s/W" % k], params["RPN/%s/b" % k] = head["RPN/%s/W" % k], head["RPN/%s/b" % k]
    model = FasterRCNN(trunk_class=ResNet101, rpn_in_ch=2048, rpn_mid_ch=512, feat_stride=32, runtime=rt)
    model.load_params(params)
    model.RPN.proposal_layer._pre_nms_top_n, model.RPN.proposal_layer._post_nms_top_n = 1000, 300
    x = rt.mem.from_numpy(synthetic.image(seed=6, h=h, w=w) / 64.0)
    for _ in range(3):
        model.trunk(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        model.trunk(x)
    e1.record()
    torch.cuda.synchronize()
    print("resnet101 trunk (eager, GPU time incl. host gaps): %.2f ms" % (e0.elapsed_time(e1) / 5))
    cap = CapturedForward(model, x, h, w)
    for _ in range(5):
        cap.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        cap.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("config 4 (ResNet-101 FasterRCNN, 1000/300) hipGraph replay: %.2f ms/img = %.1f img/s" % (ms, 1e3 / ms))


if __name__ == "__main__":
    main()
