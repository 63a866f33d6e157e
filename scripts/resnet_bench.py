#!/usr/bin/env python
"""BASELINE.json configs[3]: ResNet-101 backbone inference on one MI355X, 1000 pre-NMS / 300 post-NMS proposals, 600 x 1000, fp32
(models/resnet.py:11-45 wired literally: res5 -> RPN(2048) -> RoI pooling at 1/32 -> fc6(2048*49) ...).  One JSON record:
whole-forward hipGraph replay time, the trunk alone as its own hipGraph (HIP events on the launch stream) priced against the fp32
MFMA peak with the trunk's algorithmic FLOPs, per-stage HIP events of eager forwards (fc6 = 300 x 100352 x 4096 called out), and the
comparison of res5 / proposals with nothing (parity lives in tests/test_gpu_fullsize.py::test_resnet101_config4_600x1000).
GPU only.  Usage: python scripts/resnet_bench.py [> profiles/r03_bench_resnet101.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chainer_faster_rcnn_amd as pkg  # noqa: E402
from chainer_faster_rcnn_amd import synthetic  # noqa: E402
from chainer_faster_rcnn_amd.graph import CapturedForward  # noqa: E402
from chainer_faster_rcnn_amd.models import FasterRCNN, ResNet101  # noqa: E402
import bench  # noqa: E402  (EventTimer, graph_time_us, the peaks)


def resnet_trunk_flops(h, w, blocks=(3, 4, 23, 3)):
    """Algorithmic multiply-add FLOPs (2 per MAC; BN folded, bias / ReLU / pooling excluded) of chainer's ResNetLayers up to res5."""
    def cdiv(a, b):
        return -(-a // b)
    per = {}
    h1, w1 = (h + 2 * 3 - 7) // 2 + 1, (w + 2 * 3 - 7) // 2 + 1                       # conv1 7x7 / 2, pad 3
    per["conv1"] = 2.0 * 3 * 49 * 64 * h1 * w1
    hh, ww = cdiv(h1 - 3, 2) + 1, cdiv(w1 - 3, 2) + 1                                 # max_pooling_2d(3, stride 2), cover_all
    cin = 64
    for stage, (n, mid, stride) in enumerate(zip(blocks, (64, 128, 256, 512), (1, 2, 2, 2))):
        out = mid * 4
        tot = 0.0
        for i in range(n):
            s = stride if i == 0 else 1
            ho, wo = cdiv(hh, s), cdiv(ww, s)                                           # 1x1 / s, pad 0
            tot += 2.0 * cin * mid * ho * wo                                            # conv1 1x1 (stride on it: chainer's BottleNeckA)
            tot += 2.0 * mid * 9 * mid * ho * wo                                        # conv2 3x3
            tot += 2.0 * mid * out * ho * wo                                            # conv3 1x1
            if i == 0:
                tot += 2.0 * cin * out * ho * wo                                        # conv4: the projection shortcut
            cin, hh, ww = out, ho, wo
        per["res%d" % (stage + 2)] = tot
    return per, (hh, ww)


def cpu_baseline_resnet(params, x, samples=3, warmups=1):
    """configs[3] on this box's host cores, bounded (seconds per image): the oracle's ResNet-101 trunk (torch-CPU fp32, BatchNormalization in test mode,
    models/resnet.py:11-45) -> RPN head -> the pinned ProposalLayer restatement at 1000 / 300 with the reference's own cpu_nms where loadable -> RoI
    pooling at 1/32 (C restatement) -> fc6 (300 x 100352 x 4096) / fc7 / cls / bbox.  `warmups` untimed + `samples` timed forwards, median."""
    from oracle import frcnn_oracle as O
    O.build_c()
    nms_fn = None
    try:
        from oracle import ref_harness
        nms_fn = ref_harness.native("cpu_nms").cpu_nms
    except Exception as e:
        print("oracle/_ref cpu_nms not loadable (%s): timing the C restatement" % (e,), file=sys.stderr)
    vgg = synthetic.params(seed=1)
    cores = bench.pick_cpu_threads(torch, O, vgg)
    info = np.array([[600, 1000]], dtype=np.int32)
    times, stages = [], {}
    for it in range(warmups + samples):
        t0 = time.perf_counter()
        feat = O.resnet_forward(params, x)
        t1 = time.perf_counter()
        _, _, prob, bbox = O.rpn_head(params, feat)
        t2 = time.perf_counter()
        proposals, _ = O.proposal_layer(prob, bbox, info, train=False, feat_stride=32, pre_nms_top_n=1000, post_nms_top_n=300, nms_fn=nms_fn)
        t3 = time.perf_counter()
        pool5 = O.roi_pooling_2d(feat, np.concatenate([np.zeros((len(proposals), 1), np.float32), proposals], 1), 7, 7, 1 / 32.)
        t4 = time.perf_counter()
        O.rcnn_head(params, pool5, proposals, info)
        t5 = time.perf_counter()
        if it >= warmups:
            times.append(t5 - t0)
            for k, v in (("trunk", t1 - t0), ("rpn_head", t2 - t1), ("proposals", t3 - t2), ("roi_pool", t4 - t3), ("head", t5 - t4)):
                stages.setdefault(k, []).append(v * 1e3)
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "img/s", "cores": cores, "kind": "reference-native" if nms_fn is not None else "port",
            "sample": "%d full 600x1000 ResNet-101 forwards after %d warm-up, median; torch-CPU fp32 on `cores` threads standing in for Chainer's CPU "
                      "convolutions / linears, pinned NumPy ProposalLayer (1000 / 300), NMS = %s, C restatement of RoI pooling"
                      % (samples, warmups, "the reference's own cpu_nms.pyx (oracle/_ref)" if nms_fn is not None else "the C restatement"),
            "ms_per_image": med * 1e3, "stages_ms": {k: round(float(np.median(v)), 2) for k, v in stages.items()}}


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
    x_host = synthetic.image(seed=6, h=h, w=w) / 64.0
    x = rt.mem.from_numpy(x_host)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 1.0:                                          # clock ramp, untimed
        model.forward_device(x, h, w)
        torch.cuda.synchronize()
    timer = bench.EventTimer(torch)
    for _ in range(10):
        timer.begin()
        model.forward_device(x, h, w, timer=timer)
        timer.end()
    torch.cuda.synchronize()
    stages = timer.averages_ms()
    trunk_ms = bench.graph_time_us(torch, lambda: model.trunk(x), 1, 100) / 1e3
    cap = CapturedForward(model, x, h, w)
    for _ in range(5):
        cap.replay()
    torch.cuda.synchronize()
    steps = 100
    t0 = time.perf_counter()
    for _ in range(steps):
        out = cap.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # two images in flight per GPU (graph.ForwardsInFlight): the ResNet trunk is 104 small launches, many of them far from filling the chip
    two = None
    try:
        from chainer_faster_rcnn_amd.graph import ForwardsInFlight

        def make_model(rt_i):
            m_ = FasterRCNN(trunk_class=ResNet101, rpn_in_ch=2048, rpn_mid_ch=512, feat_stride=32, runtime=rt_i)
            m_.load_params(params)
            m_.RPN.proposal_layer._pre_nms_top_n, m_.RPN.proposal_layer._post_nms_top_n = 1000, 300
            return m_
        fl = ForwardsInFlight(make_model, lambda: pkg.runtime.Runtime(rt.lib, pkg.runtime.TorchDeviceMemory(str(rt.mem.device))), x, h, w, n=2)
        for _ in range(10):
            fl.submit()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2 * steps):
            fl.submit()
        torch.cuda.synchronize()
        two_rate = 2 * steps / (time.perf_counter() - t1)
        ref = cap.replay()
        torch.cuda.synchronize()
        same = all(bool(torch.equal(g.out[k], ref[k])) for g in fl.slots for k in ("rois", "cls_prob", "pred_boxes", "n_out"))
        two = {"img_s_two_images_in_flight": two_rate, "outputs_identical_to_the_serial_graph": same,
               "stream_set_probe_img_s": {"best": round(max(fl.probe.values()), 1), "worst": round(min(fl.probe.values()), 1)}}
    except Exception as e:
        two = {"error": repr(e)}
        torch.cuda.synchronize()
    per, (fh, fw) = resnet_trunk_flops(h, w)
    trunk_flops = sum(per.values())
    rpn_flops = 2.0 * 2048 * 9 * 512 * fh * fw
    fc6_flops = 2.0 * 300 * 2048 * 49 * 4096
    rec = {"metric": "images/sec ResNet-101 Faster R-CNN 600x1000", "value": 1e3 / ms, "unit": "img/s", "n_gpus": 1, "steps": steps, "warmup": 5,
           "ms_per_step": ms, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "ResNet-101 backbone inference, 1xMI355X, batch 1, 1000 pre-NMS / 300 post-NMS proposals, fp32 "
                                  "(BASELINE.json configs[3]); res5 at stride 32, RoI pooling at 1/32, fc6 over 2048 x 7 x 7",
                      "image": "1x3x600x1000", "launch": "hipGraph replay", "n_rois_last_step": int(out["n_out"].cpu()[0]), "feature_map": [2048, fh, fw]},
           "roofline": {"bound": "mfma", "kernel": "conv_mfma_f32_kernel (ResNet-101 trunk: 7x7/2 stem as im2col + 1x1, 33 bottlenecks of 1x1 / 3x3 / 1x1 "
                                                     "with folded BatchNormalization and the residual epilogue): 104 convolution launches",
                        "achieved": trunk_flops / (trunk_ms * 1e-3) / 1e12, "peak": bench.PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": trunk_flops / (trunk_ms * 1e-3) / 1e12 / bench.PEAK_F32_MFMA_TFLOPS, "traffic": None,
                        "trunk_ms": trunk_ms, "trunk_ms_source": "HIP events around 100 replays of a hipGraph holding exactly the trunk's launches",
                        "algorithmic_gflop_trunk": trunk_flops / 1e9, "algorithmic_gflop_by_stage": {k: v / 1e9 for k, v in per.items()}},
           "stages_ms": {k: round(v, 4) for k, v in stages.items()},
           "fc6": {"shape": "300 x 100352 x 4096", "ms": stages.get("fc6"), "tflops": fc6_flops / (stages["fc6"] * 1e-3) / 1e12 if stages.get("fc6") else None,
                   "weight_mb": 100352 * 4096 * 4 / 1e6},
           "rpn_conv_3x3": {"gflop": rpn_flops / 1e9}, "two_images_in_flight": two}
    if "--no-cpu-baseline" not in sys.argv:
        try:
            rec["cpu_baseline"] = cpu_baseline_resnet(params, x_host)
        except Exception as e:
            rec["cpu_baseline"] = {"error": repr(e)}
    else:
        rec["cpu_baseline"] = None
    bench.emit_json_line(rec)


if __name__ == "__main__":
    main()
