#!/usr/bin/env python
"""Benchmark of the Faster R-CNN hot path on MI355X (BASELINE.json metric: images/sec, VGG16, 600x1000).

  python bench.py --gpus N --steps K --warmup W
N>1: the driver launches `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`; a plain
`python bench.py --gpus N` re-launches itself that way (one rank per GPU; with fewer GPUs than ranks the ranks share GPUs over
gloo -- a functional smoke of the N>1 path, said so in the line).

A "step" = one full inference forward of one synthetic 600x1000 image per GPU (trunk -> RPN -> proposals ->
NMS -> RoI pooling -> FC head -> decode), inputs and weights resident in HBM before the timed region.
Workload = BASELINE.json configs[1]: "VGG16 inference, 1xMI355X, batch 1, 300 proposals post-NMS, fp32".
Images shard one per GPU with no collective on the data path (weak scaling).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel family (the MFMA conv3x3: 14 launches per image) --
algorithmic FLOPs / HIP-event time of a hipGraph holding exactly those launches, on the launch stream -- against the dense MFMA
peak of the dtype (MI355X_MICROARCH.md).  `nms_roi` prices RoI pooling against the HBM peak and reports proposals+NMS us/img, both
from hipGraphs of back-to-back launches (kernel time only) next to the in-pipeline stage events.  `cpu_baseline` times the CPU
path of forward.py on this box's host cores per SURVEY 8(d): torch-CPU fp32 convs/linears (stand-in for Chainer's CPU im2col+GEMM),
the reference's own compiled cpu_nms (oracle/_ref) inside the pinned NumPy restatement of proposal_layer.py, the C restatement of
Chainer's RoI pooling; per-stage ms, median of >= 5 images.  `parity` compares the device forward with that very oracle forward.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: Peak BF16 MFMA, dense
SUSTAINED_BF16_MFMA_TFLOPS_RANDOM = 1867.0   # measured, round 3: every SIMD issuing only bf16 MFMAs on random operands (profiles/r03_mfma_peak_micro.txt)
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak BW (spec)
IM_H, IM_W = 600, 1000
DEFAULT_STEPS = 250               # ~1 s of timed region at 3.9 ms / step: long enough for an outside observer (rocm-smi) to see it


class EventTimer(object):
    """Records a HIP event (torch.cuda.Event on the current stream = the stream our kernels launch on) at
    every stage boundary; durations are read after the timed region."""

    def __init__(self, torch):
        self.torch = torch
        self.steps = []
        self.cur = None

    def begin(self):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        self.cur = [("_start", e)]

    def mark(self, name):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        self.cur.append((name, e))

    def end(self):
        self.steps.append(self.cur)
        self.cur = None

    def averages_ms(self):
        acc = {}
        for st in self.steps:
            for (n0, e0), (n1, e1) in zip(st[:-1], st[1:]):
                acc.setdefault(n1, []).append(e0.elapsed_time(e1))
        return {k: float(np.mean(v)) for k, v in acc.items()}


def pmc_traffic(dtype):
    """HBM bytes per conv launch from the committed PMC passes of this same command (scripts/gpu_evidence.sh + scripts/gpu_traffic.py: FETCH_SIZE and
    WRITE_SIZE in separate rocprofv3 --pmc runs); counters cannot be read from inside the process, so the bench line carries
    the last measured figure and names its source, or null when no PMC pass exists for this dtype."""
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    cands = {"f32": ["r06_hbm_traffic_pmc.json", "r05_hbm_traffic_pmc.json", "r04_hbm_traffic_pmc.json", "r03_hbm_traffic_pmc.json", "r02_hbm_traffic_pmc.json", "r01_hbm_traffic_pmc.json"],
             "bf16": ["r06_hbm_traffic_pmc_bf16.json", "r05_hbm_traffic_pmc_bf16.json", "r04_hbm_traffic_pmc_bf16.json", "r03_hbm_traffic_pmc_bf16.json", "r02_hbm_traffic_pmc_bf16.json"],
             "f32s": ["r03_hbm_traffic_pmc_f32s.json", "r02_hbm_traffic_pmc_f32s.json"], "f16": []}[dtype]      # (no PMC pass of the fp16 twins: same kernels, same bytes as bf16)
    kernel = {"f32": "conv_mfma_f32_kernel", "bf16": "conv_bf16_kernel", "f32s": "conv_f32s_kernel", "f16": "conv_bf16_kernel"}[dtype]
    for name in cands:
        path = os.path.join(prof, name)
        if not os.path.exists(path):
            continue
        try:
            s = json.load(open(path))["_summary"][kernel]
            src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % name
            if dtype == "bf16" and name.startswith(("r02_", "r03_")):
                # that pass measured conv_dma_bf16_kernel on every layer; since then eleven of the fourteen launches run as strip forms D / C
                # (csrc/conv_bf16_strip.h, DESIGN 3.8b: larger tiles, fewer weight re-reads) -- no PMC pass of those picks exists yet
                src += "; measured BEFORE the strip-form default picks of late round 3 (conv_dma_bf16_kernel on every layer): an upper bound for today's launches"
            return s["hbm_bytes_per_launch"], src
        except (KeyError, ValueError):
            continue
    return None, None


def conv_algorithmic_bytes(model_layers, h, w, esize=4):
    """Bytes each 3x3 conv launch must move once: input + output (after a fused pool where there is one) + weights + bias."""
    out, prev = {}, None
    for l in model_layers:
        if l == "pool":
            h2, w2 = (h + 1) // 2, (w + 1) // 2
            if prev is not None:        # VGG16Prev.fuse_pool: the pooled map is what the conv launch writes
                out[prev[0]] -= esize * prev[2] * (h * w - h2 * w2)
            h, w = h2, w2
        else:
            name, ci, co = l
            out[name] = esize * (ci * h * w + co * h * w) + 4 * (9 * ci * co + co)
            prev = l
    out["rpn_conv_3x3"] = esize * (512 * h * w * 2) + 4 * (9 * 512 * 512 + 512)
    return out


def conv_flops(model_layers, h, w):
    """Algorithmic FLOPs (2*MAC) of every 3x3 conv at a h x w input; bias/ReLU/pool excluded."""
    out = {}
    for l in model_layers:
        if l == "pool":
            h, w = (h + 1) // 2, (w + 1) // 2
        else:
            name, ci, co = l
            out[name] = 2.0 * h * w * co * ci * 9
    out["rpn_conv_3x3"] = 2.0 * h * w * 512 * 512 * 9
    return out, (h, w)


def pick_cpu_threads(torch, O, params):
    """batch-1 convs do not scale to hundreds of threads: pick the thread count that is fastest on one mid-size layer (conv3_2 at
    150x250) -- reported as `cores` by the CPU baselines."""
    xs = np.random.RandomState(0).randn(1, 256, 150, 250).astype(np.float32)
    best = None
    for nt in sorted(set([8, 16, 32, 64, os.cpu_count() or 1])):
        if nt > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(nt)
        O.conv2d(xs, params["trunk/conv3_2/W"], params["trunk/conv3_2/b"], 1)
        t0 = time.perf_counter()
        O.conv2d(xs, params["trunk/conv3_2/W"], params["trunk/conv3_2/b"], 1)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
    torch.set_num_threads(best[1])
    return best[1]


def train_flops(model_layers, h, w, stage2=False, n_rois=300, bwd_rows=128):
    """Algorithmic FLOPs (2 per MAC) of one training step at h x w: every 3x3 convolution forward, its input gradient (conv1_1 has none: the image
    needs no gradient) and its weight gradient.  RPN mode (train_rpn.py): trunk + rpn_conv_3x3 in all three passes (the two 1x1 heads, 0.6 GFLOP,
    are left out).  Stage 2 (train_rcnn.py): the trunk in all three passes, rpn_conv_3x3 forward only (proposals carry no gradient), fc6 / fc7
    forward over n_rois rows and backward (input + weight gradients) over ProposalTargetLayer's bwd_rows kept rows."""
    fl, _ = conv_flops(model_layers, h, w)
    total = sum(fl.values())
    rpn = fl["rpn_conv_3x3"]
    first = fl[model_layers[0][0]]
    if not stage2:
        return {"forward": total, "input_gradients": total - first, "weight_gradients": total}
    trunk = total - rpn
    fc = lambda rows: 2.0 * rows * (512 * 49 * 4096 + 4096 * 4096)
    return {"forward": trunk + rpn + fc(n_rois), "input_gradients": trunk - first + fc(bwd_rows), "weight_gradients": trunk + fc(bwd_rows)}


def cpu_train_baseline(params, x, gt, stage2, samples=3, warmups=1, rank_seed=0):
    """The reference's training step on this box's host cores, bounded: `warmups` untimed + `samples` timed steps (seconds each), median.
    RPN mode (train_rpn.py:140-182): AnchorTargetLayer + the train-mode ProposalLayer the reference runs and discards + forward / losses / backward
    (torch-CPU autograd standing in for Chainer's) + WeightDecay / MomentumSGD over every parameter.  Stage 2 (train_rcnn.py:35-78): trunk + RPN +
    ProposalLayer without gradient, ProposalTargetLayer, RoI pooling (C restatement) + head with dropout + losses + backward + update.
    The detection glue is the oracle's pinned restatement (bit-for-bit the reference's classes on the golden vectors: tests/test_oracle_pinned.py);
    cpu_nms / bbox_overlaps are the reference's own Cython, compiled in place into oracle/_ref, where loadable."""
    import torch
    from oracle import frcnn_oracle as O
    O.build_c()
    nms_fn, native = None, False
    try:
        from oracle import ref_harness
        nms_fn = ref_harness.native("cpu_nms").cpu_nms
        native = True
    except Exception as e:
        print("oracle/_ref cpu_nms not loadable (%s): timing the C restatement" % (e,), file=sys.stderr)
    cores = pick_cpu_threads(torch, O, params)
    info = np.array([[IM_H, IM_W]], dtype=np.int32)
    rng = np.random.RandomState(rank_seed)
    names = [k for k in params if k.startswith("trunk/") or (k.split("/")[0] in ("fc6", "fc7", "cls_score", "bbox_pred") if stage2 else k.startswith("RPN/"))]
    P = {k: params[k].copy() for k in params}
    V = {k: np.zeros_like(P[k]) for k in names}
    times, stages = [], {}
    for it in range(warmups + samples):
        st = {}
        t0 = time.perf_counter()
        dup = 0.0
        if not stage2:
            fh, fw = (IM_H + 15) // 16, (IM_W + 15) // 16
            labels, targets, inds, n_all = O.anchor_target_layer(fh, fw, gt, info, rng=rng)
            st["anchor_targets"] = time.perf_counter() - t0
            t1 = time.perf_counter()
            loss, grads = O.rpn_train_grads(P, x, labels, targets, inds, n_all)
            st["forward_backward"] = time.perf_counter() - t1
            t1 = time.perf_counter()
            # the train-mode ProposalLayer (12000 / 2000) the reference runs inside the step and discards (region_proposal_network.py:121-124)
            with torch.no_grad():
                feat = O.vgg16_trunk(P, x)
                dup = time.perf_counter() - t1                        # a second trunk forward this composition needs and the reference does not
                _, _, prob, bbox = O.rpn_head(P, feat)
            t2 = time.perf_counter()
            O.proposal_layer(prob, bbox, info, train=True, nms_fn=nms_fn)
            st["proposal_layer_train_mode"] = time.perf_counter() - t2
        else:
            with torch.no_grad():
                feat = O.vgg16_trunk(P, x)
                dup = time.perf_counter() - t0
                _, _, prob, bbox = O.rpn_head(P, feat)
            t1 = time.perf_counter()
            proposals, _ = O.proposal_layer(prob, bbox, info, train=True, nms_fn=nms_fn)
            st["proposal_layer"] = time.perf_counter() - t1
            t1 = time.perf_counter()
            use_gt, ext, keep = O.proposal_target_layer(proposals, gt, rng=rng)
            st["proposal_target_layer"] = time.perf_counter() - t1
            t1 = time.perf_counter()
            m6 = (rng.rand(len(proposals), 4096) >= 0.5).astype(np.float32) * 2.0     # F.dropout's masks, drawn on the host as chainer's CPU path does
            m7 = (rng.rand(len(proposals), 4096) >= 0.5).astype(np.float32) * 2.0
            st["dropout_masks"] = time.perf_counter() - t1
            t1 = time.perf_counter()
            loss, grads = O.rcnn_train_grads(P, x, proposals, keep, use_gt[:, -1].astype(np.int32), ext, m6, m7)
            st["forward_backward"] = time.perf_counter() - t1
        t1 = time.perf_counter()
        for k in names:       # O.momentum_sgd_wd's arithmetic, in place as chainer's CPU update rule does it (NumPy, single-threaded: g += wd * W; v = m v - lr g; W += v)
            g = np.array(grads[k], dtype=np.float32)
            g += np.float32(0.0005) * P[k]
            V[k] *= np.float32(0.9)
            g *= np.float32(0.001)
            V[k] -= g
            P[k] += V[k]
        st["update"] = time.perf_counter() - t1
        total = time.perf_counter() - t0 - dup
        if it >= warmups:
            times.append(total)
            for k, v in st.items():
                stages.setdefault(k, []).append(v * 1e3)
            stages.setdefault("second_trunk_forward_not_counted", []).append(dup * 1e3)
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "img/s", "cores": cores, "kind": "reference-native" if native else "port",
            "sample": "%d %s training steps at 600x1000 after %d warm-up, median (seconds per step: bounded on purpose); torch-CPU fp32 autograd on `cores` "
                      "threads standing in for Chainer's CPU convolutions / linears, the pinned NumPy restatements of AnchorTargetLayer / ProposalLayer / "
                      "ProposalTargetLayer, cpu_nms + bbox_overlaps = %s; this composition evaluates the trunk forward twice (once without autograd, for the "
                      "proposals) where the reference evaluates it once: the second one is timed and NOT counted"
                      % (samples, "stage-2 (train_rcnn.py)" if stage2 else "RPN (train_rpn.py)", warmups,
                         "the reference's own Cython compiled in place (oracle/_ref)" if native else "the C restatement"),
            "ms_per_step": med * 1e3, "stages_ms": {k: round(float(np.median(v)), 2) for k, v in stages.items()}, "loss_last_step": float(loss)}


def train_roofline(flops, step_ms, fwd_bwd_ms, profile):
    """`roofline` of a training line: the step's convolution (+ FC) FLOPs over the step time against the fp32 MFMA peak.  The kernels are many (forward,
    input-gradient, weight-gradient forms on two streams), so `achieved` is priced on the WHOLE step (a lower bound of the kernels' own fraction);
    the same FLOPs over the forward + backward stage alone (HIP events) and the rocprofv3 per-kernel sums of the same command are beside it."""
    total = sum(flops.values())
    ach = total / (step_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "conv_mfma_f32_kernel + conv_wgrad_* + conv_dgrad forms + linear_dma_f32_kernel: every MFMA launch of the step, priced together",
            "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None,
            "algorithmic_gflop_per_step": total / 1e9, "algorithmic_gflop": {k: v / 1e9 for k, v in flops.items()},
            "basis": "whole step (ms_per_step): forward + backward + all-reduce + update; the MFMA kernels' own fraction is higher",
            "frac_of_forward_backward_stage": (total / (fwd_bwd_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS) if fwd_bwd_ms else None,
            "rocprof_kernel_sums": profile}


def cpu_baseline(params, x, samples):
    """forward.py's CPU path on this box (SURVEY 8(d) / BASELINE.md 4): warm-up 2, then `samples` (>= 5) full 600x1000 images,
    per-stage medians.  Returns (the bench-line object, the oracle's debug dict of the LAST image for the parity block)."""
    import torch
    from oracle import frcnn_oracle as O
    info = np.array([[IM_H, IM_W]], dtype=np.int32)
    O.build_c()
    nms_fn, nms_kind = None, "C restatement of models/cpu_nms.pyx (oracle/c/frcnn_oracle.c)"
    try:
        from oracle import ref_harness
        nms_fn = ref_harness.native("cpu_nms").cpu_nms
        nms_kind = "the reference's own models/cpu_nms.pyx, compiled in place into oracle/_ref (single-threaded, as in the reference)"
    except Exception as e:                                          # oracle/_ref absent: the restatement is the baseline, and says so
        print("oracle/_ref cpu_nms not loadable (%s): timing the C restatement" % (e,), file=sys.stderr)
    pick_cpu_threads(torch, O, params)
    samples = max(int(samples), 5)
    stages, totals, dbg = {}, [], None
    for it in range(2 + samples):
        st = {}
        t0 = time.perf_counter()
        layers = {}
        feat = O.vgg16_trunk(params, x, collect=layers)
        t1 = time.perf_counter()
        h, score, prob, bbox = O.rpn_head(params, feat)
        t2 = time.perf_counter()
        pt = {}
        proposals, probs, pdbg = O.proposal_layer(prob, bbox, info, train=False, return_debug=True, nms_fn=nms_fn, stage_times=pt)
        t3 = time.perf_counter()
        brois = np.concatenate((np.zeros((len(proposals), 1), np.float32), proposals), axis=1)
        pool5 = O.roi_pooling_2d(feat, brois, 7, 7, 1.0 / 16)
        t4 = time.perf_counter()
        cls_prob, pred_boxes, hd = O.rcnn_head(params, pool5, proposals, info)
        t5 = time.perf_counter()
        if it < 2:
            continue
        st = {"trunk": t1 - t0, "rpn_head": t2 - t1, "proposals_decode_sort": pt["decode_sort"], "nms": pt["nms"],
              "roi_pool": t4 - t3, "head": t5 - t4}
        for k, v in st.items():
            stages.setdefault(k, []).append(v * 1e3)
        totals.append(t5 - t0)
        dbg = dict(hd, feat=feat, rpn_h=h, rpn_cls_score=score, rpn_cls_prob=prob, rpn_bbox_pred=bbox, proposals=proposals, probs=probs,
                   pool5=pool5, proposal_debug=pdbg, layers=layers)
    med = float(np.median(totals))
    res = {"value": 1.0 / med, "unit": "img/s", "cores": torch.get_num_threads(),
           "kind": "reference-native" if nms_fn is not None else "port",
           "sample": "%d full 600x1000 forwards after 2 warm-ups, median; torch-CPU fp32 convs/linears on `cores` threads (stand-in for Chainer's "
                     "CPU im2col+GEMM), the pinned NumPy restatement of proposal_layer.py, NMS = %s, C restatement of Chainer's RoI pooling"
                     % (samples, nms_kind),
           "ms_per_image": med * 1e3, "stages_ms": {k: round(float(np.median(v)), 3) for k, v in stages.items()}}
    return res, dbg


def launch_command(gpus, port, argv):
    """The command `python bench.py --gpus N` turns itself into when no launcher is around it -- exactly the driver's form: one rank per GPU of ONE node,
    rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(gpus)), "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become `torch.distributed.run --nproc-per-node N bench.py ...`."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(launch_command(args.gpus, port, sys.argv[1:]), env=env))


def rank_placement(environ, gpus, n_dev):
    """(rank, local_rank, world, device index, backend, shared_note) of this process from the launcher's environment.  One rank per GPU: device =
    LOCAL_RANK over RCCL.  Fewer GPUs than ranks (a 1-GPU box): the ranks share GPUs and talk over gloo -- RCCL refuses two ranks on one device --
    which exercises the N > 1 code path (sharding, barriers, bucketed all-reduce) but is NOT a scaling measurement."""
    world = int(environ.get("WORLD_SIZE", "1"))
    rank = int(environ.get("RANK", "0"))
    local_rank = int(environ.get("LOCAL_RANK", "0"))
    if gpus > 1 and world != gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (gpus, world))
    backend = environ.get("FRCNN_DIST_BACKEND", "nccl")
    n_dev = max(int(n_dev), 1)
    shared_note = None
    if world > n_dev:
        backend = "gloo"
        shared_note = "%d ranks on %d GPU(s), gloo: functional smoke of the N>1 path, not a scaling number" % (world, n_dev)
    device = local_rank if backend == "nccl" else local_rank % n_dev
    if device >= n_dev:
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, n_dev))
    return rank, local_rank, world, device, backend, shared_note


def graph_time_us(torch, fn, launches_per_replay, replays):
    """Average duration of ONE launch sequence `fn` (it must enqueue `launches_per_replay` repetitions): HIP events on the launch
    stream around `replays` replays of a hipGraph holding them -- kernel time only, no host in the interval."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn()
    for _ in range(2):
        g.replay()
    # three batches of `replays`, the median batch: one transient (a clock dip, another process's burst) inside a single interval once turned the 8 us RoI
    # figure behind the NMS scan into 49 us (round 6, gpurun_out/r06h) while every other line of the same call read 8.0
    batches = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        batches.append(e0.elapsed_time(e1) * 1e3 / (replays * launches_per_replay))
    return sorted(batches)[1]


def f32s_conv_chain(rt, model, x, bf16=False):
    """The 14 convolution launches of the f32s (or bf16 / f16) model: first layer straight from the fp32 image, fused pools, rpn_conv_3x3."""
    rt = model.rt                                  # the model's own runtime: an fp16 model computes its 16-bit chain through the *_f16* entry points
    def chain():
        tr_ = model.trunk
        h, n_l = None, len(tr_.layers)
        pair = bf16 and tr_.conv1_pair_applies()
        for idx, l in enumerate(tr_.layers):
            if pair and idx < 3:                                  # conv1_1 + conv1_2 + pool1 as one launch (csrc/conv_bf16_pair.hip): 13 launches then
                if idx == 0:
                    l1, l2 = tr_.links[tr_.layers[0][0]], tr_.links[tr_.layers[1][0]]
                    h = rt.conv1_pair_bf16(x, l1.W, l1.b, l2.Wb, l2.b)
                continue
            if l == "pool":
                continue
            link = tr_.links[l[0]]
            pool = idx + 1 < n_l and tr_.layers[idx + 1] == "pool"
            if h is None:
                h = rt.conv1_bf16(x, link.W, link.b, relu=True) if bf16 else rt.conv1_f32s(x, link.W, link.b, relu=True)
            else:
                h = link.bf16(h, relu=True, pool=pool) if bf16 else link.f32s(h, relu=True, pool=pool)
        return model.RPN.rpn_conv_3x3.bf16(h, relu=True) if bf16 else model.RPN.rpn_conv_3x3.f32s(h, relu=True, out_f32_nchw=True)
    return chain


def split_variant(args, torch, rt, params, x, dbg, flops_total):
    """The SAME fp32 network with its 3x3 convolutions computed as six bf16 MFMA products of 3-way split fp32 operands (fp32
    accumulation; csrc/conv_f32s.hip), timed like the contract line (K hipGraph replays) and checked against the same oracle
    forward.  Reported next to the native-fp32-MFMA contract line, not instead of it."""
    from chainer_faster_rcnn_amd.graph import CapturedForward
    from chainer_faster_rcnn_amd.models import FasterRCNN
    model = FasterRCNN(runtime=rt, conv_dtype="f32s", head_dtype="f32s")
    model.load_params(params)
    for _ in range(max(args.warmup, 3)):
        model.forward_device(x, IM_H, IM_W)
    torch.cuda.synchronize()
    graph = CapturedForward(model, x, IM_H, IM_W, warmup=1)
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"what": ("fp32 tensors and results; every 3x3 convolution = six v_mfma_f32_32x32x16_bf16 products of 3-way split fp32 operands "
                    "(h.h, h.m, m.h, h.l, l.h, m.m; dropped terms < 2^-24 of a product), fp32 accumulation, and the four fully connected layers "
                    "likewise; proposals / RoI pooling / decode as in the contract line.  `python bench.py --dtype f32s` prints this variant as its own line."),
           "value": args.steps / dt, "unit": "img/s", "ms_per_step": dt / args.steps * 1e3, "steps": args.steps, "launch": "hipGraph replay"}
    try:
        conv_ms = graph_time_us(torch, f32s_conv_chain(rt, model, x), 1, max(args.steps, 100)) / 1e3
        out.update(conv_ms_per_image=conv_ms, conv_algorithmic_tflops=flops_total / (conv_ms * 1e-3) / 1e12,
                   mfma_executed_tflops=6 * flops_total / (conv_ms * 1e-3) / 1e12,
                   frac_of_bf16_mfma_peak=6 * flops_total / (conv_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS)
    except Exception as e:
        print("f32s conv-chain graph failed (%s)" % (e,), file=sys.stderr)
        torch.cuda.synchronize()
    if dbg is not None:
        try:
            from oracle import parity
            info = np.array([[IM_H, IM_W]], dtype=np.int32)
            rep = parity.compare_forward(params, info, dbg, parity.device_forward_host(rt, model, x, IM_H, IM_W), layer_tol=1e-3, head_tol=1e-3)
            out["parity"] = {k: rep[k] for k in ("ok", "layers_worst", "conv5_3_rel_err", "rpn_cls_prob_rel_err", "rpn_bbox_pred_rel_err",
                                                 "proposals_index_exact_given_device_maps", "from_image_index_match_positional", "pool5_exact",
                                                 "cls_prob_rel_err", "pred_boxes_rel_err", "end_to_end_cls_prob_rel_err", "tolerances")}
        except Exception as e:
            out["parity"] = {"ok": False, "error": repr(e)}
    return out


def feed_variant(torch, rt, graph, steps, n_images=8, check_oracle=True):
    """The same captured forward with the INPUT inside the timed region (VERDICT r04 missing #4; /root/reference/forward.py:33-45, 85-94: cv.imread ->
    img_preprocessing -> model): a different uint8 HWC image every step, pinned host memory -> H2D on a copy stream (1.8 MB instead of the 7.2 MB
    fp32 tensor) -> frcnn_preprocess_u8 (mean subtraction, resize at im_scale 1.0, HWC -> CHW) straight into the graph's input buffer -> graph replay.
    Double-buffered in time, not in memory: the preprocess kernel is the ONLY reader of the uint8 device buffer, so image k+1's copy starts as soon as
    step k's preprocess has run (an event) and rides under step k's forward.  A secondary figure: the contract line's `value` keeps its input resident."""
    from chainer_faster_rcnn_amd.postprocess import PIXEL_MEANS
    dev = rt.mem.device
    rs = np.random.RandomState(123)
    host = [torch.from_numpy(rs.randint(0, 256, size=(IM_H, IM_W, 3), dtype=np.uint8)).pin_memory() for _ in range(n_images)]
    u8 = torch.empty((IM_H, IM_W, 3), dtype=torch.uint8, device=dev)
    means = np.asarray(PIXEL_MEANS, dtype=np.float64).ravel()
    main = torch.cuda.current_stream(dev)
    copies = [torch.cuda.Stream(device=dev) for _ in range(4)]        # candidates: a copy stream that shares a hardware queue with the compute stream serialises with it
    copy = copies[0]
    consumed, landed = torch.cuda.Event(), torch.cuda.Event()
    report = {}
    if check_oracle:                                          # the fed tensor is the oracle's img_preprocessing of the same pixels
        from oracle import frcnn_oracle as O
        u8.copy_(host[0])
        rt.preprocess_u8(u8, means, 1.0, (IM_H, IM_W), out=graph.x)
        torch.cuda.synchronize()
        want, scale = O.img_preprocessing(host[0].numpy(), PIXEL_MEANS)
        got = rt.mem.to_numpy(graph.x)[0]
        report["fed_image_vs_oracle_img_preprocessing_max_abs"] = float(np.abs(got - want).max()) if got.shape == want.shape else "shape %r vs %r" % (got.shape, want.shape)
        report["im_scale"] = float(scale)

    # the host stays at most four images ahead of the device (a serving loop reads its results anyway): with unbounded run-ahead -- hundreds of graph launches,
    # copies and cross-stream events queued -- the same loop measured 1420 instead of 1553-1567 img/s on the bf16 line (sync every 1 ... 16 images: all the same)
    sync_every = 4

    def run(k_steps):
        with torch.cuda.stream(copy):
            u8.copy_(host[0], non_blocking=True)
            landed.record(copy)
        for k in range(k_steps):
            main.wait_event(landed)
            rt.preprocess_u8(u8, means, 1.0, (IM_H, IM_W), out=graph.x)
            consumed.record(main)
            with torch.cuda.stream(copy):
                copy.wait_event(consumed)
                u8.copy_(host[(k + 1) % n_images], non_blocking=True)
                landed.record(copy)
            graph.graph.replay()
            if sync_every and k % sync_every == sync_every - 1:
                consumed.synchronize()                        # the host stays at most `sync_every` images ahead of the device
        main.wait_event(landed)
    probe = []
    for cand in copies:                                       # (ForwardsInFlight's lesson: which streams share one of the runtime's hardware queues is not knowable up front)
        copy = cand
        run(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(30)
        torch.cuda.synchronize()
        probe.append(30 / (time.perf_counter() - t0))
    copy = copies[int(np.argmax(probe))]
    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    report["copy_stream_probe_img_s"] = [round(v, 1) for v in probe]
    report.update({"img_s_with_feed": steps / dt, "ms_per_step_with_feed": dt / steps * 1e3, "steps": steps, "distinct_images": n_images,
                   "host_runs_ahead_by_at_most": sync_every,
                   "what": "per step: pinned uint8 HWC image (1.8 MB) -> H2D on a copy stream under the previous forward -> frcnn_preprocess_u8 into the graph's input -> graph replay"})
    return report


def two_streams_variant(torch, pkg, rt, params, dtype, x, steps, graph_a):
    """TWO images in flight per GPU (chainer_faster_rcnn_amd.graph.ForwardsInFlight: two model instances with their own workspaces and captured graphs on two HIP
    streams, image k on slot k % 2): image k's proposal / RoI / head stages overlap image k + 1's convolutions.  Throughput of a serving loop, not the latency of
    one image; a secondary figure next to the one-image-at-a-time contract line.  The outputs of the concurrent replays are compared with the serial graph's."""
    from chainer_faster_rcnn_amd.graph import ForwardsInFlight
    from chainer_faster_rcnn_amd.models import FasterRCNN

    def make_model(rt_i):
        m = FasterRCNN(runtime=rt_i, conv_dtype=dtype, head_dtype=dtype)
        m.load_params(params)
        return m
    fl = ForwardsInFlight(make_model, lambda: pkg.runtime.Runtime(rt.lib, pkg.runtime.TorchDeviceMemory(str(rt.mem.device))), x, IM_H, IM_W, n=2)
    for _ in range(20):
        fl.submit()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fl.submit()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ref = graph_a.replay(x)
    torch.cuda.synchronize()
    same = all(bool(torch.equal(g.out[k], ref[k])) for g in fl.slots for k in ("rois", "cls_prob", "pred_boxes", "n_out"))
    # ... and the serving loop whole: two in flight AND the input inside the timed region (a different pinned uint8 image per step -> H2D -> frcnn_preprocess_u8 ->
    # forward, all on the slot's stream, stream-ordered behind the slot's previous image; the host stays at most eight images ahead)
    fed, fed_probe = None, None
    try:
        from chainer_faster_rcnn_amd.postprocess import PIXEL_MEANS
        rs = np.random.RandomState(321)
        host = [torch.from_numpy(rs.randint(0, 256, size=(IM_H, IM_W, 3), dtype=np.uint8)).pin_memory() for _ in range(8)]
        means = np.asarray(PIXEL_MEANS, dtype=np.float64).ravel()

        def run_fed(k_steps):
            for k in range(k_steps):
                fl.submit_u8(host[k % len(host)], means, 1.0)     # (a slot's copy / preprocess / replay are stream-ordered behind its previous image: no hazard)
                if k % 8 == 7:
                    fl.wait()                                     # the host stays at most eight images ahead
        fed_probe = fl.reprobe(lambda k: fl.submit_u8(host[k % len(host)], means, 1.0))          # the copies bring other queues into play: pick the stream pair again
        run_fed(10)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_fed(steps)
        torch.cuda.synchronize()
        fed = steps / (time.perf_counter() - t1)
    except Exception as e:
        fed = repr(e)
        torch.cuda.synchronize()
    return {"img_s_two_images_in_flight": steps / dt, "img_s_two_images_in_flight_with_feed": fed,
            "fed_stream_set_probe_img_s": (sorted((round(v, 1) for v in fed_probe.values()), reverse=True) if isinstance(fed, float) and fed_probe else None), "ms_per_image": dt / steps * 1e3, "steps": steps, "outputs_identical_to_the_serial_graph": same,
            "stream_set_probe_img_s": {"best": round(max(fl.probe.values()), 1), "worst": round(min(fl.probe.values()), 1), "sets": len(fl.probe),
                                       "why": "streams that share one of the runtime's hardware queues run one after the other: the constructor keeps the pair that overlaps"},
            "what": "two model instances (own workspaces + captured graphs, same weights), image k on HIP stream k % 2: image k's proposal / RoI / head stages overlap image k + 1's convolutions"}


def bf16_variant(args, torch, rt, params, x, dbg, flops_total, half="bf16"):
    """BASELINE.json configs[2] on this GPU ("bf16 convs / fp32 RoI", 1 image per GPU): the bf16 chain (csrc/conv_bf16.hip: operands
    rounded to bf16, fp32 accumulation on v_mfma_f32_32x32x16_bf16, bf16 FC head; proposals, RoI pooling, decode in fp32), timed like
    the contract line (K hipGraph replays), its conv chain priced against the dense bf16 MFMA peak, and compared with the same oracle
    forward.  Bit-exact proposal indices FROM THE IMAGE are claimed for the fp32 lines only: a bf16 trunk moves conv5_3 by ~1e-2.
    half = "f16": the same chain in its fp16 instantiation (csrc/conv_f16.hip: v_mfma_f32_32x32x16_f16, 10 mantissa bits; north_star's "fp16/bf16 accumulate
    fp32"): timing and parity only (no feed / two-in-flight repeats)."""
    from chainer_faster_rcnn_amd.graph import CapturedForward
    from chainer_faster_rcnn_amd.models import FasterRCNN
    model = FasterRCNN(runtime=rt, conv_dtype=half, head_dtype=half)
    model.load_params(params)
    for _ in range(max(args.warmup, 3)):
        model.forward_device(x, IM_H, IM_W)
    torch.cuda.synchronize()
    graph = CapturedForward(model, x, IM_H, IM_W, warmup=1)
    graph.replay()
    torch.cuda.synchronize()
    steps = max(args.steps, 100)
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    feed = None
    if not args.no_feed_variant and half == "bf16":
        try:
            feed = feed_variant(torch, rt, graph, steps, check_oracle=False)
        except Exception as e:
            feed = {"error": repr(e)}
            torch.cuda.synchronize()
    out = {"what": ("BASELINE.json configs[2] per GPU: the 13 trunk convolutions, rpn_conv_3x3, the RPN heads and the four FC layers with bf16 operands "
                    "and fp32 accumulation (v_mfma_f32_32x32x16_bf16); proposals / RoI pooling / decode in fp32.  `python bench.py --dtype bf16` "
                    "prints this configuration as its own line."),
           "value": steps / dt, "unit": "img/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "launch": "hipGraph replay"}
    if half == "f16":
        out["what"] = ("the 16-bit chain of BASELINE.json configs[2] in its fp16 instantiation (north_star: fp16/bf16 accumulate fp32): the same kernel sources compiled "
                       "with fp16 pack / widen / v_mfma_f32_32x32x16_f16 (csrc/conv_f16.hip, conv_f16_pair.hip, linear_f16.hip), same layouts and schedule; RoI pooling takes "
                       "the fp32 kernel between two conversions.  `python bench.py --dtype f16` prints this configuration as its own line.")
    if feed is not None:
        out["with_feed"] = feed
    if not args.no_two_streams_variant and half == "bf16":
        try:
            import chainer_faster_rcnn_amd as _pkg
            graph.replay(x)
            out["two_images_in_flight"] = two_streams_variant(torch, _pkg, rt, params, "bf16", x, 2 * steps, graph)
        except Exception as e:
            out["two_images_in_flight"] = {"error": repr(e)}
            torch.cuda.synchronize()
    try:
        conv_ms = graph_time_us(torch, f32s_conv_chain(rt, model, x, bf16=True), 1, max(args.steps, 100)) / 1e3
        out.update(conv_ms_per_image=conv_ms, conv_tflops=flops_total / (conv_ms * 1e-3) / 1e12,
                   frac_of_bf16_mfma_peak=flops_total / (conv_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS)
        # a second reading, not a replacement: what every SIMD of this chip sustains issuing nothing but bf16 MFMAs on RANDOM operands
        # (scripts/micro/mfma_peak_micro.hip -> profiles/r03_mfma_peak_micro.txt: 1856-1878 TFLOP/s, against 2385-2467 on constant operands:
        # the matrix datapath is power-limited on real data) -- a measured constant of round 3, not re-measured by this run
        out["frac_of_sustained_bf16_mfma_rate_on_random_operands"] = out["conv_tflops"] / SUSTAINED_BF16_MFMA_TFLOPS_RANDOM
        out["sustained_rate_source"] = "profiles/r03_mfma_peak_micro.txt (1867 TFLOP/s = the mean of 1856 and 1878; DESIGN 3.8b)"
    except Exception as e:
        print("bf16 conv-chain graph failed (%s)" % (e,), file=sys.stderr)
        torch.cuda.synchronize()
    try:                                                     # which kernel family each 3x3 launch of the chain runs (launch-free query of the library)
        names = {0: "conv_dma_bf16_kernel", 901: "strip A", 902: "strip B", 903: "strip C (K split over waves)", 909: "strip D", 910: "strip D", 921: "resident R", 922: "resident R2"}
        picks, h, w = {}, IM_H, IM_W
        from chainer_faster_rcnn_amd.models.vgg16 import LAYERS as _layers
        for l in _layers:
            if l == "pool":
                h, w = (h + 1) // 2, (w + 1) // 2
            else:
                picks[l[0]] = names.get(int(rt.lib.frcnn_conv_bf16_plan(int(l[1]), int(l[2]), h, w, 3, 0)), "?") if l[1] > 3 else "conv1_f32s_kernel (first layer)"
        picks["rpn_conv_3x3"] = names.get(int(rt.lib.frcnn_conv_bf16_plan(512, 512, h, w, 3, 0)), "?")
        if model.trunk.conv1_pair_applies():                  # conv1_1 + conv1_2 + pool1 are ONE launch (csrc/conv_bf16_pair.hip): 13 conv launches per image
            picks[_layers[0][0]] = picks[_layers[1][0]] = "conv1_pair_pc_bf16_kernel (conv1_1 + conv1_2 + pool1 in one launch)"
        out["conv_kernel_picks"] = picks
    except Exception as e:
        out["conv_kernel_picks"] = {"error": repr(e)}
    if dbg is not None:
        try:
            from oracle import parity
            info = np.array([[IM_H, IM_W]], dtype=np.int32)
            tol = 3e-2 if half == "bf16" else 4e-3
            rep = parity.compare_forward(params, info, dbg, parity.device_forward_host(rt, model, x, IM_H, IM_W), layer_tol=tol, head_tol=tol)
            out["parity"] = {k: rep.get(k) for k in ("ok", "layers_worst", "conv5_3_rel_err", "rpn_cls_prob_rel_err", "rpn_bbox_pred_rel_err",
                                                     "proposals_index_exact_given_device_maps", "from_image_index_match_positional",
                                                     "from_image_index_match_set", "pool5_exact", "cls_prob_rel_err", "pred_boxes_rel_err", "tolerances")}
            out["parity"]["claim"] = ("stage by stage, given the device's own maps: proposal indices and pool5 exact; from the IMAGE a bf16 trunk does not "
                                      "reproduce the fp32 oracle's proposal indices (see from_image_index_match_*): that claim is made for the fp32 lines only"
                                      if half == "bf16" else
                                      "stage by stage, given the device's own maps: proposal indices and pool5 exact; from the IMAGE the fp16 trunk stays within ~1.5e-3 of the "
                                      "fp32 oracle's conv5_3 (bf16: 1.1e-2) and reproduces its proposal SET up to a box or two, but not every position (north_star's 1e-3 / "
                                      "bit-exact-from-image bar is met by the fp32 lines only)")
        except Exception as e:
            out["parity"] = {"ok": False, "error": repr(e)}
    return out


def emit_json_line(obj):
    """The contract's ONE JSON line, as the LAST thing on stdout: RCCL writes a version banner through C stdio when its first
    communicator comes up, which a buffered stdout flushes at exit -- after the line.  So: print, flush, then point file descriptor 1
    at stderr for whatever the C runtime still holds."""
    print(json.dumps(obj))
    sys.stdout.flush()
    try:
        os.dup2(2, 1)
    except OSError:
        pass


def gather_rank_times(torch, dist, dt, world):
    """MAX over ranks of the timed interval (the contract's value) and every rank's own interval (rank 0 reports the spread)."""
    if dist is None or world == 1:
        return dt, [dt]
    t = torch.tensor([dt], device="cuda", dtype=torch.float64)
    if dist.get_backend() == "gloo":
        t = t.cpu()
    ts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(ts, t)
    per = [float(v.item()) for v in ts]
    return max(per), per


def per_rank_block(per_rank, steps):
    ms = [v / steps * 1e3 for v in per_rank]
    return {"ms_per_step": [round(v, 4) for v in ms], "min_ms": round(min(ms), 4), "max_ms": round(max(ms), 4), "spread_ms": round(max(ms) - min(ms), 4)}


def train_mode(args, torch, dist, rt, model, x, rank, world, barrier, shared_note, params=None, x_host=None):
    """BASELINE.json configs[4]: train_rpn.py's step -- forward, ProposalLayer (train top-N, discarded, as the reference runs it),
    anchor targets, losses, backward, the all-reduce of the flat gradient buffer (RCCL), fused MomentumSGD+WD -- one synthetic
    VOC-shaped image per GPU per step."""
    from chainer_faster_rcnn_amd.train import RPNTrainer, TorchComm
    model.rpn_train = True
    # --dtype f32s in train mode: forward and input-gradient convolutions as bf16x6 split products (weight gradients on the fp32 kernel)
    conv_math = "split" if args.dtype == "f32s" else "mfma"
    tr = RPNTrainer(model, comm=TorchComm(force_single_rank=args.dist_world1) if dist is not None else None, run_proposal_layer=not args.no_train_proposals, conv_math=conv_math)
    rs = np.random.RandomState(rank)
    G = 4
    w, h = rs.uniform(32, 400, G), rs.uniform(32, 400, G)
    x1, y1 = rs.uniform(0, IM_W - 1 - w), rs.uniform(0, IM_H - 1 - h)
    gt = np.stack([x1, y1, x1 + w, y1 + h, rs.randint(1, 21, G)], axis=1).astype(np.float32)[None]
    gt_dev = rt.mem.from_numpy(gt)
    info = np.array([[IM_H, IM_W]], dtype=np.int32)
    np.random.seed(rank)
    ev = {}

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.setdefault(name, []).append(e)

    # untimed clock-ramp preamble (see the inference mode) -- a FIXED number of steps: every step holds collectives, so all ranks
    # must run the same count (a wall-clock loop would let them diverge and deadlock the first all-reduce)
    for _ in range(max(1, int(round(args.ramp_seconds / 0.0125)))):
        out = tr.step(x, info, gt_dev)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = tr.step(x, info, gt_dev)
    if getattr(getattr(tr, "comm", None), "trace", None) is not None:
        tr.comm.trace.clear()                                       # host-side comm milliseconds of the timed steps only
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mark("start")
        out = tr.forward_backward(x, info, gt_dev)
        mark("fwd_bwd")
        tr.all_reduce()
        mark("all_reduce")
        tr.update()
        mark("update")
    barrier()
    dt = time.perf_counter() - t0
    # the same K steps without the (discarded) ProposalLayer launch sequence, for the record
    other_ms = None
    if not args.no_train_proposals:
        tr.run_proposal_layer = False
        for _ in range(2):
            tr.step(x, info, gt_dev)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            tr.step(x, info, gt_dev)
        barrier()
        other_ms = (time.perf_counter() - t1) / args.steps * 1e3
    dt, per_rank = gather_rank_times(torch, dist, dt, world)
    if rank == 0:
        st = {k: float(np.mean([a.elapsed_time(b) for a, b in zip(ev[p], ev[k])])) for p, k in
              (("start", "fwd_bwd"), ("fwd_bwd", "all_reduce"), ("all_reduce", "update"))}
        from chainer_faster_rcnn_amd.models.vgg16 import LAYERS
        step_ms = dt / args.steps * 1e3
        roof = train_roofline(train_flops(LAYERS, IM_H, IM_W), step_ms, st["fwd_bwd"],
                              "profiles/r06_train_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --mode train`)")
        if conv_math == "split":
            roof["basis"] += "; the forward / input-gradient products run on the bf16 pipes (six per fp32 product): the fraction is of the fp32 MFMA peak and may exceed what fp32 MFMAs could do"
        cpu = None
        if world == 1 and not args.no_cpu_baseline and params is not None:
            try:
                cpu = cpu_train_baseline(params, x_host, gt, stage2=False, samples=3, warmups=1, rank_seed=rank)
            except Exception as e:
                cpu = {"error": repr(e)}
        emit_json_line({"metric": "images/sec RPN training step VGG16 600x1000", "value": world * args.steps / dt, "unit": "img/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32s" if conv_math == "split" else "f32", "data": "synthetic",
                          "config": {"workload": "train_rpn.py end-to-end RPN training step, 1 image per GPU, all-reduce of the flat fp32 gradient "
                                                 "buffer in 3 buckets overlapped with the backward pass (BASELINE.json configs[4])" +
                                                 ("; forward and input-gradient convolutions as six bf16 MFMA products of 3-way split fp32 operands, "
                                                  "weight gradients on the fp32 MFMA kernel" if conv_math == "split" else ""),
                                     "conv_math": conv_math,
                                     "proposal_layer_in_step": (not args.no_train_proposals),
                                     "grad_buffer_mb": tr.n_flat * 4 / 1e6, "global_batch": world, "ranks_share_gpus": shared_note},
                          "ms_per_step_without_proposal_layer": other_ms, "ramp_seconds": args.ramp_seconds,
                          "per_rank": per_rank_block(per_rank, args.steps),
                          "dist": {"backend": (dist.get_backend() if dist is not None else None), "world_size": world,
                                   "single_rank_process_group": bool(args.dist_world1 and world == 1),
                                   "comm_host_ms_per_step": ({k: round(v[1] / max(args.steps, 1), 4) for k, v in tr.comm.trace.items()}
                                                             if getattr(getattr(tr, "comm", None), "trace", None) else None)},
                          "roofline": roof,
                          "cpu_baseline": cpu if world == 1 else "not run: ranks > 1 (the N = 1 line carries it)",
                          "stages_ms": st, "losses": tr.losses_host(out)})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def train_rcnn_mode(args, torch, dist, rt, model, x, rank, world, barrier, shared_note, params=None, x_host=None):
    """Stage 2 of the reference's alternating schedule (train_rcnn.py:35-78; models/faster_rcnn.py:110-173 with rcnn_train = True): trunk -> RPN
    proposals (no gradient) -> RoI pooling with arg-max -> fc6 / fc7 + dropout -> cls_score / bbox_pred -> ProposalTargetLayer -> losses ->
    backward through the head, RoI pooling and the trunk -> all-reduce -> MomentumSGD + WeightDecay over trunk + head (548 MB of parameters).
    One synthetic VOC-shaped image per GPU per step; per-stage HIP events through the trainer's stage hook."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.train import RCNNTrainer, TorchComm
    model.rpn_train, model.rcnn_train = False, True
    conv_math = "split" if args.dtype == "f32s" else "mfma"
    tr = RCNNTrainer(model, comm=TorchComm(force_single_rank=args.dist_world1) if dist is not None else None, conv_math=conv_math,
                     dropout_rng=args.dropout_rng, dropout_seed=rank)
    rs = np.random.RandomState(rank)
    G = 4
    w, h = rs.uniform(32, 400, G), rs.uniform(32, 400, G)
    x1, y1 = rs.uniform(0, IM_W - 1 - w), rs.uniform(0, IM_H - 1 - h)
    gt = np.stack([x1, y1, x1 + w, y1 + h, rs.randint(1, 21, G)], axis=1).astype(np.float32)[None]
    gt, info = Variable(gt), Variable(np.array([[IM_H, IM_W]], dtype=np.int32))       # ProposalTargetLayer keeps the reference's type checks
    np.random.seed(rank)
    ev = []

    def hook(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev[-1].append((name, e, time.perf_counter()))

    for _ in range(max(2, args.warmup)):                              # a fixed count (collectives inside): see train_mode
        out = tr.step(x, info, gt)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev.append([])
        tr.stage_hook = hook
        out = tr.forward_backward(x, info, gt)
        tr.stage_hook = None
        tr.all_reduce()
        hook("all_reduce")
        tr.update()
        hook("update")
    barrier()
    dt = time.perf_counter() - t0
    dt, per_rank = gather_rank_times(torch, dist, dt, world)
    if rank == 0:
        names = [n for n, _, _ in ev[0]]
        st = {names[i]: float(np.mean([s[i - 1][1].elapsed_time(s[i][1]) for s in ev])) for i in range(1, len(names))}
        # the host's side of the same boundaries: how long the Python thread took to ENQUEUE each stage (a stage whose GPU time equals its enqueue
        # time is bound by the launch rate of the un-captured step, not by its kernels)
        st_host = {names[i]: float(np.mean([(s[i][2] - s[i - 1][2]) * 1e3 for s in ev])) for i in range(1, len(names))}
        from chainer_faster_rcnn_amd.models.vgg16 import LAYERS
        step_ms = dt / args.steps * 1e3
        roof = train_roofline(train_flops(LAYERS, IM_H, IM_W, stage2=True, n_rois=int(out["n_rois"]), bwd_rows=int(out["keep_inds"].shape[0])), step_ms,
                              sum(v for k, v in st.items() if k not in ("all_reduce", "update")),
                              "profiles/r06_train_rcnn_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --mode train-rcnn`)")
        cpu = None
        if world == 1 and not args.no_cpu_baseline and params is not None:
            try:
                cpu = cpu_train_baseline(params, x_host, gt.data, stage2=True, samples=3, warmups=1, rank_seed=rank)
            except Exception as e:
                cpu = {"error": repr(e)}
        emit_json_line({"metric": "images/sec Fast R-CNN (stage 2) training step VGG16 600x1000", "value": world * args.steps / dt, "unit": "img/s",
                        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32s" if conv_math == "split" else "f32", "data": "synthetic",
                        "config": {"workload": "train_rcnn.py stage-2 training step (trunk + RoI head; RPN proposals without gradient), 1 image per GPU, "
                                               "all-reduce of the flat fp32 gradient buffer in 3 buckets (SURVEY 8f-2; not a BASELINE.json config)",
                                   "conv_math": conv_math, "grad_buffer_mb": tr.n_flat * 4 / 1e6, "global_batch": world, "n_rois_last_step": int(out["n_rois"]),
                                   "head_backward_rows_last_step": int(out["keep_inds"].shape[0]),
                                   "head_backward": "on ProposalTargetLayer's kept rows only (every other row of the head's output gradient is exactly zero: "
                                                    "faster_rcnn.py:155-160); FRCNN_RCNN_BWD_ROWS=all is the zero-padded form over all n_rois rows",
                                   "ranks_share_gpus": shared_note,
                                   "dropout_rng": args.dropout_rng,
                                   "host_in_step": ("dropout masks (2 x n_rois x 4096 floats from NumPy's global RNG, as chainer's CPU path draws them: ~7 ms) and "
                                                    if args.dropout_rng == "numpy" else "dropout masks drawn by the dropout kernel (counter-based hash; no host work); ") +
                                                   "ProposalTargetLayer's subsample is host work inside the timed step (under the head's forward pass), after one device->host read of "
                                                   "the RoI count, the RoIs and their float64 IoU matrix"},
                        "per_rank": per_rank_block(per_rank, args.steps),
                        "dist": {"backend": (dist.get_backend() if dist is not None else None), "world_size": world},
                        "roofline": roof,
                        "cpu_baseline": cpu if world == 1 else "not run: ranks > 1",
                        "stages_ms": {k: round(v, 4) for k, v in st.items()}, "sum_of_stages_ms": round(sum(st.values()), 4),
                        "host_enqueue_ms": {k: round(v, 4) for k, v in st_host.items()},
                        "losses": tr.losses_host(out)})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default %d for inference, 40 for --mode train)" % DEFAULT_STEPS)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ramp-seconds", type=float, default=1.0, help="untimed clock-ramp preamble before the warm-up steps")
    ap.add_argument("--cpu-samples", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16-variant", action="store_true", help="skip the bf16_config3 block (BASELINE configs[2] per GPU) of the default line")
    ap.add_argument("--no-stage-events", action="store_true")
    ap.add_argument("--no-two-streams-variant", action="store_true",
                    help="skip the secondary measurement with two images in flight per GPU (a second model instance replaying its graph on a second stream)")
    ap.add_argument("--no-feed-variant", action="store_true",
                    help="skip the secondary measurement with the input inside the timed region (uint8 image -> H2D -> frcnn_preprocess_u8 -> forward, a different image per step)")
    ap.add_argument("--no-split-variant", action="store_true",
                    help="f32 inference line only: skip the second measurement with the convolutions computed as bf16x6 split products")
    ap.add_argument("--no-train-proposals", action="store_true",
                    help="--mode train: skip the ProposalLayer(12000/2000) launch sequence the reference runs and discards in every RPN step")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the forward as ONE captured hipGraph in the timed region (auto = on: the ~45 launches of a bf16 step are "
                         "shorter than the host can issue them, and the f32 step, GPU-bound in eager mode on a quiet host, lost up to "
                         "10 % of wall clock to host jitter on some boxes; off = eager launches)")
    ap.add_argument("--dtype", choices=["f32", "f32s", "bf16", "f16"], default="f32",
                    help="f32 = BASELINE.json configs[1] (the contract line); bf16 = configs[2]: bf16 convolutions, fp32 RoI / head; f16 = the fp16 instantiation of the "
                         "same 16-bit chain (north_star: fp16/bf16 accumulate fp32): same kernels and rate, 10 mantissa bits")
    ap.add_argument("--dist-world1", action="store_true",
                    help="with one rank: still create the process group (nccl = RCCL) and run every collective of the N > 1 path through it")
    ap.add_argument("--dropout-rng", choices=["device", "numpy"], default="device",
                    help="--mode train-rcnn: where F.dropout's masks are drawn (numpy = the reference CPU path's random stream, on the host)")
    ap.add_argument("--mode", choices=["infer", "train", "train-rcnn"], default="infer",
                    help="infer = BASELINE.json configs[1] (the contract line); train = configs[4], the RPN training step; train-rcnn = the stage-2 step of train_rcnn.py (SURVEY 8f-2)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = DEFAULT_STEPS if args.mode == "infer" else 40

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    import torch
    rank, _, world, local_rank, backend, shared_note = rank_placement(os.environ, args.gpus, torch.cuda.device_count())
    if rank != 0:
        os.dup2(2, 1)                                              # only rank 0 owns stdout (the launcher merges the ranks' streams)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.dist_world1:
        # --dist-world1: a process group of ONE rank -- RCCL initialisation, the bucketed async all-reduces on the collective stream and
        # their stream waits all really execute on the one GPU a development box has (no peer: the sums are identities)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import chainer_faster_rcnn_amd as pkg
    from chainer_faster_rcnn_amd import synthetic
    from chainer_faster_rcnn_amd.models import FasterRCNN
    from chainer_faster_rcnn_amd.models.vgg16 import LAYERS
    rt = pkg.runtime.Runtime(pkg._lib.load(), pkg.runtime.TorchDeviceMemory("cuda:%d" % local_rank))
    params = synthetic.params(seed=1)
    # f32s: the fp32 network of configs[1] with every 3x3 convolution computed as six bf16 MFMA products of 3-way split fp32
    # operands (fp32 accumulation; csrc/conv_f32s.hip), the four fully connected layers likewise; proposals, RoI pooling, decode as in f32
    model = FasterRCNN(runtime=rt, conv_dtype=args.dtype, head_dtype=args.dtype)
    model.load_params(params)
    x_host = synthetic.image(seed=rank, h=IM_H, w=IM_W)          # every rank its own image (1 img / GPU)
    x = rt.mem.from_numpy(x_host)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode == "train":
        return train_mode(args, torch, dist, rt, model, x, rank, world, barrier, shared_note, params, x_host)
    if args.mode == "train-rcnn":
        return train_rcnn_mode(args, torch, dist, rt, model, x, rank, world, barrier, shared_note, params, x_host)

    use_graph = args.graph in ("on", "auto")
    # Untimed preamble before the W warm-up steps: the first forwards of a process run at idle clocks (DVFS needs a few hundred
    # ms of load to settle; 5 steps are 20 ms), which showed up as 0.71 vs 0.79 roofline fractions between otherwise identical
    # runs.  Not part of W, K or the timed region.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        model.forward_device(x, IM_H, IM_W)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        model.forward_device(x, IM_H, IM_W)
    timer = None if args.no_stage_events else EventTimer(torch)
    if use_graph:
        # Stage events come from an eager pass (they cannot be read out of a replayed graph); the TIMED region below is
        # K replays of one captured hipGraph of the whole forward: no Python, no per-launch host cost inside it.
        if timer:
            for _ in range(min(20, max(3, args.steps // 3))):
                timer.begin()
                model.forward_device(x, IM_H, IM_W, timer=timer)
                timer.end()
        torch.cuda.synchronize()
        try:
            from chainer_faster_rcnn_amd.graph import CapturedForward
            graph = CapturedForward(model, x, IM_H, IM_W, warmup=1)
            out = graph.replay()
        except Exception as e:                                    # never lose the run to the capture: time eager launches instead
            print("hipGraph capture failed (%s): timing eager launches" % (e,), file=sys.stderr)
            torch.cuda.synchronize()
            use_graph, graph = False, None
    if use_graph:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        barrier()
        dt = time.perf_counter() - t0
    else:
        eager_timer = timer if args.graph == "off" else None      # after a failed capture the stage events already exist
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if eager_timer:
                eager_timer.begin()
            out = model.forward_device(x, IM_H, IM_W, timer=eager_timer)
            if eager_timer:
                eager_timer.end()
        barrier()
        dt = time.perf_counter() - t0
    n_rois = int(out["n_out"].cpu()[0])
    feed_main = None
    if rank == 0 and world == 1 and use_graph and not args.no_feed_variant:
        try:
            feed_main = feed_variant(torch, rt, graph, max(args.steps, 50), check_oracle=not args.no_cpu_baseline)
        except Exception as e:
            feed_main = {"error": repr(e)}
            torch.cuda.synchronize()
        graph.replay(x)                                           # the resident image back in the graph's input (parity below reads its outputs)
    two_main = None
    if rank == 0 and world == 1 and use_graph and not args.no_two_streams_variant:
        try:
            graph.replay(x)
            torch.cuda.synchronize()
            two_main = two_streams_variant(torch, pkg, rt, params, args.dtype, x, max(args.steps, 50), graph)
        except Exception as e:
            two_main = {"error": repr(e)}
            torch.cuda.synchronize()
    # roofline of the dominant kernel: the 14 conv launches (13 trunk convs with their fused pools + rpn_conv_3x3) as their own
    # hipGraph, replays bracketed by HIP events on the launch stream -- kernel time only, whatever the host is doing
    conv_chain_ms, iso = None, {}
    iso_replays = max(args.steps, 100)
    if rank == 0 and args.graph != "off" and not args.no_stage_events:
        try:
            if args.dtype == "f32":
                def conv_chain():
                    return model.RPN.rpn_conv_3x3(model.trunk(x), relu=True)
            elif args.dtype == "f32s":
                conv_chain = f32s_conv_chain(rt, model, x)
            else:
                conv_chain = f32s_conv_chain(rt, model, x, bf16=True)
            conv_chain_ms = graph_time_us(torch, conv_chain, 1, iso_replays) / 1e3
        except Exception as e:
            print("conv-chain graph failed (%s): roofline from the per-stage events" % (e,), file=sys.stderr)
            torch.cuda.synchronize()
        # RoI pooling and the proposal pipeline in isolation: 8 back-to-back launches per replay, the RoI outputs rotating over 10 buffers
        # (301 MB > the 256 MB Infinity Cache, so the writes cannot all be absorbed on-die)
        try:
            feat = model.trunk(x)
            _, _, prob, bbox = model.RPN.heads(feat, want_score=False, x_bf16=getattr(model.trunk, "feat_bf16", None) if args.dtype in ("bf16", "f16") else None,
                                               x_split=getattr(model.trunk, "feat_split", None) if args.dtype == "f32s" else None)
            rois, _, _ = model.RPN.proposal_layer.forward_device(prob, bbox, IM_H, IM_W)
            outs = [rt.mem.empty((int(rois.shape[0]), 512, 7, 7), "f32") for _ in range(10)]
            state = {"i": 0}

            def roi_seq():
                for _ in range(8):
                    rt.roi_pool_fwd_chw(feat, rois, 7, 7, 1.0 / 16, out=outs[state["i"] % 10])
                    state["i"] += 1
            iso["roi_pool_us"] = graph_time_us(torch, roi_seq, 8, max(iso_replays // 4, 25))
            # the training forms (models/faster_rcnn.py:125-126 in rcnn_train mode): forward with argmax_data and backward, 65.1 MB each
            ams = [rt.mem.empty((int(rois.shape[0]), 512, 7, 7), "i32") for _ in range(10)]
            dxs = [rt.mem.empty(tuple(int(v) for v in feat.shape), "f32") for _ in range(4)]

            def roi_am_seq():
                for _ in range(8):
                    rt.roi_pool_fwd_chw(feat, rois, 7, 7, 1.0 / 16, want_argmax=True, out=outs[state["i"] % 10], out_argmax=ams[state["i"] % 10])
                    state["i"] += 1
            iso["roi_pool_argmax_us"] = graph_time_us(torch, roi_am_seq, 8, max(iso_replays // 4, 25))

            def roi_bwd_seq():
                for _ in range(8):
                    rt.roi_pool_bwd(outs[state["i"] % 10], ams[state["i"] % 10], 512, int(feat.shape[-2]), int(feat.shape[-1]), out=dxs[state["i"] % 4])
                    state["i"] += 1
            iso["roi_pool_bwd_us"] = graph_time_us(torch, roi_bwd_seq, 8, max(iso_replays // 4, 25))
            del outs, ams, dxs

            def prop_seq():
                for _ in range(8):
                    model.RPN.proposal_layer.forward_device(prob, bbox, IM_H, IM_W)
            iso["proposals_nms_us"] = graph_time_us(torch, prop_seq, 8, max(iso_replays // 4, 25))

            # RoI pooling IN the pipeline: eight times (proposal pipeline -> RoI pooling of ITS OWN rois) in one graph, minus the proposal pipeline's own
            # figure = what the RoI launch costs behind the NMS scan it depends on (launch latency + prologue are not hidden by anything there)
            outs2 = [rt.mem.empty((int(rois.shape[0]), 512, 7, 7), "f32") for _ in range(4)]

            def prop_roi_seq():
                for _ in range(8):
                    r_, _, _ = model.RPN.proposal_layer.forward_device(prob, bbox, IM_H, IM_W)
                    rt.roi_pool_fwd_chw(feat, r_, 7, 7, 1.0 / 16, out=outs2[state["i"] % 4])
                    state["i"] += 1
            iso["proposals_plus_roi_us"] = graph_time_us(torch, prop_roi_seq, 8, max(iso_replays // 4, 25))
            del outs2
        except Exception as e:
            print("isolated RoI / proposal graphs failed (%s)" % (e,), file=sys.stderr)
            torch.cuda.synchronize()
    dt, per_rank = gather_rank_times(torch, dist, dt, world)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * args.steps / dt
        res = {"metric": "images/sec VGG16 Faster R-CNN 600x1000", "value": value, "unit": "img/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": ("VGG16 inference, 1xMI355X per image, batch 1, 300 proposals post-NMS, fp32 "
                                       "(BASELINE.json configs[1]); 1 image per GPU per step") if args.dtype == "f32" else
                                      ("VGG16 inference, 1xMI355X per image, batch 1, 300 proposals post-NMS, fp32 tensors and results "
                                       "(BASELINE.json configs[1]); the 14 3x3 convolutions and the 4 fully connected layers run as six bf16 MFMA products of "
                                       "3-way split fp32 operands with fp32 accumulation (dropped terms < 2^-24 of a product)") if args.dtype == "f32s" else
                                      ("VGG16 inference, data-parallel 1 img/GPU, bf16 convs + bf16 FC head (fp32 accumulate) / fp32 proposals, "
                                       "RoI pooling, decode (BASELINE.json configs[2])") if args.dtype == "bf16" else
                                      ("VGG16 inference, data-parallel 1 img/GPU, fp16 convs + fp16 FC head (fp32 accumulate) / fp32 proposals, RoI pooling, decode: "
                                       "BASELINE.json configs[2]'s 16-bit chain in its fp16 instantiation (north_star: fp16/bf16 accumulate fp32; the same kernels compiled "
                                       "with v_mfma_f32_32x32x16_f16: csrc/conv_f16.hip)"),
                          "image": "1x3x600x1000", "global_batch": world, "launch": "hipGraph replay" if use_graph else "eager", "parallelism": "dp%d (images sharded, no collective)" % world,
                          "n_rois_last_step": n_rois, "ranks_share_gpus": shared_note,
                          "timed_region": ("K replays of ONE captured hipGraph of the whole forward (image -> cls_prob / boxes) on ONE image already resident in HBM: "
                                           "no H2D inside the region, the same pixels every step") if use_graph else
                                          "K eager forwards on one image already resident in HBM (no H2D inside the region, the same pixels every step)"},
               "ramp_seconds": args.ramp_seconds,
               "per_rank": per_rank_block(per_rank, args.steps)}
        if world > 1:
            res["cpu_baseline"] = "not run: ranks > 1 (the N = 1 line carries it)"
        if timer:
            avg = timer.averages_ms()
            flops, (fh, fw) = conv_flops(LAYERS, IM_H, IM_W)
            conv_ms = sum(avg[k] for k in flops)
            conv_src = "sum of the 14 per-stage HIP-event intervals"
            if conv_chain_ms is not None:
                conv_ms, conv_src = conv_chain_ms, ("HIP events around %d replays of a hipGraph holding exactly the 14 conv launches (no host "
                                                    "gaps inside the interval; the per-stage figures below come from eager launches)" % iso_replays)
            conv_tf = sum(flops.values()) / (conv_ms * 1e-3) / 1e12
            peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
            alg_tf = conv_tf
            if args.dtype == "f32s":
                conv_tf *= 6.0                                     # the MFMA work actually executed: six bf16 products per algorithmic product
            traffic, traffic_src = pmc_traffic(args.dtype)
            res["roofline"] = {"bound": "mfma", "achieved": conv_tf, "peak": peak, "unit": "TFLOP/s",
                               "frac": conv_tf / peak, "traffic": traffic, "traffic_source": traffic_src,
                               "kernel": ("conv_f32s_kernel (14 launches/image: 13 VGG-16 convs + rpn_conv_3x3)" if args.dtype == "f32s" else
                                          "conv1_f32s_kernel<..F32> (conv1_1, native fp32) + conv_mfma_f32_kernel (13 launches: conv1_2 ... conv5_3 + rpn_conv_3x3): 14 launches/image" if args.dtype == "f32" else
                                          ("bf16 conv chain: conv1_pair_pc_bf16_kernel (conv1_1 + conv1_2 + pool1) + conv_dma_bf16_kernel / conv_strip_bf16_kernel, 13 launches/image"
                                           if model.trunk.conv1_pair_applies() else
                                           "bf16 conv chain: conv1_f32s_kernel + conv_dma_bf16_kernel / conv_strip_bf16_kernel, 14 launches/image")),
                               "algorithmic_tflops": alg_tf,
                               "algorithmic_gflop_per_image": sum(flops.values()) / 1e9, "conv_ms_per_image": conv_ms, "conv_ms_source": conv_src,
                               "algorithmic_bytes_per_launch": sum(conv_algorithmic_bytes(LAYERS, IM_H, IM_W, {"f32": 4, "f32s": 6, "bf16": 2, "f16": 2}[args.dtype]).values()) / 14.0}
            if args.dtype == "f16":
                res["roofline"]["kernel"] += (" -- the fp16 instantiation (csrc/conv_f16.hip / conv_f16_pair.hip: the same kernel sources and names compiled with "
                                              "v_mfma_f32_32x32x16_f16)")
            if args.dtype == "f32s":
                res["roofline"]["note"] = ("achieved / peak count the bf16 MFMA flops executed (6 per algorithmic product) against the dense bf16 peak; "
                                           "algorithmic_tflops is the fp32 convolution work per second (fp32 MFMA peak: %.1f)" % PEAK_F32_MFMA_TFLOPS)
            roi_bytes = (512 * fh * fw + 300 * 512 * 49) * 4 + 300 * 16
            res["stages_ms"] = {k: round(v, 4) for k, v in avg.items()}
            res["stage_events"] = {"where": ("HIP events on the launch stream around every stage of %d eager forwards run immediately before the "
                                             "timed graph replays (events cannot be read out of a replayed graph)" % len(timer.steps)) if args.graph != "off"
                                   else "HIP events on the launch stream inside the timed region",
                                   "sum_of_stages_ms": round(sum(avg.values()), 4), "timed_ms_per_step": round(ms_per_step, 4)}
            res["per_layer_tflops"] = {k: round(flops[k] / (avg[k] * 1e-3) / 1e12, 2) for k in flops}        # algorithmic
            roi_us = iso.get("roi_pool_us", avg["roi_pool"] * 1e3)
            res["nms_roi"] = {"proposals_nms_us": iso.get("proposals_nms_us", avg["proposals"] * 1e3), "roi_pool_us": roi_us,
                              "source": ("HIP events around hipGraphs of 8 back-to-back launches (kernel time; RoI outputs rotate over 10 buffers = 301 MB)"
                                         if "roi_pool_us" in iso else "per-stage HIP events of eager launches"),
                              "roi_pool_us_behind_nms_in_one_graph": (iso["proposals_plus_roi_us"] - iso["proposals_nms_us"]) if "proposals_plus_roi_us" in iso and "proposals_nms_us" in iso else None,
                              "proposals_nms_us_in_pipeline_stage_event": avg["proposals"] * 1e3,
                              "roi_pool_us_in_pipeline_stage_event": avg["roi_pool"] * 1e3,
                              "roi_pool_algorithmic_mb": roi_bytes / 1e6,
                              "roi_pool_gbps": roi_bytes / (roi_us * 1e-6) / 1e9,
                              "roi_pool_frac_of_hbm_peak": roi_bytes / (roi_us * 1e-6) / 1e9 / PEAK_HBM_GBPS}
            # the RoI launch where it actually runs -- behind the NMS scan inside one graph (its 2.7 us launch latency overlaps the scan's tail): the same
            # algorithmic bytes over (graph of [proposal pipeline -> RoI pooling of its own RoIs] - graph of the pipeline alone); the isolated figure stays beside it
            ing = res["nms_roi"]["roi_pool_us_behind_nms_in_one_graph"]
            res["nms_roi"]["roi_pool_frac_of_hbm_peak_behind_nms_in_one_graph"] = (roi_bytes / (ing * 1e-6) / 1e9 / PEAK_HBM_GBPS) if ing else None
            # training forms: map read + y AND argmax_data written (forward), dy and argmax_data read + dx written (backward)
            train_bytes = (512 * fh * fw + 2 * 300 * 512 * 49) * 4 + 300 * 16
            for key, tag in (("roi_pool_argmax_us", "roi_pool_fwd_argmax"), ("roi_pool_bwd_us", "roi_pool_bwd")):
                if key in iso:
                    res["nms_roi"][tag + "_us"] = iso[key]
                    res["nms_roi"][tag + "_algorithmic_mb"] = train_bytes / 1e6
                    res["nms_roi"][tag + "_frac_of_hbm_peak"] = train_bytes / (iso[key] * 1e-6) / 1e9 / PEAK_HBM_GBPS
        if feed_main is not None:
            res["with_feed"] = feed_main
        if two_main is not None:
            res["two_images_in_flight"] = two_main
        dbg = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"], dbg = cpu_baseline(params, x_host, args.cpu_samples)
            except Exception as e:                                 # the measured line is never lost to the checker's side (a missing oracle/_ref, a host without gcc ...)
                res["cpu_baseline"] = {"value": None, "error": repr(e)}
                print("cpu_baseline failed: %r" % (e,), file=sys.stderr)
            try:
                if dbg is None:
                    raise RuntimeError("no cpu_baseline forward to compare with")
                from oracle import parity
                info = np.array([[IM_H, IM_W]], dtype=np.int32)
                tol = 3e-2 if args.dtype == "bf16" else (4e-3 if args.dtype == "f16" else 1e-3)
                rep = parity.compare_forward(params, info, dbg, parity.device_forward_host(rt, model, x, IM_H, IM_W), layer_tol=tol, head_tol=tol)
                rep["against"] = "the cpu_baseline forward of this run (same image, same weights); /root/reference/forward.py:92-94"
                res["parity"] = rep
            except Exception as e:
                res["parity"] = {"ok": False, "error": repr(e)}
        if world == 1 and args.dtype == "f32" and not args.no_bf16_variant:
            try:
                from chainer_faster_rcnn_amd.models.vgg16 import LAYERS as _L3
                res["bf16_config3"] = bf16_variant(args, torch, rt, params, x, dbg, sum(conv_flops(_L3, IM_H, IM_W)[0].values()))
            except Exception as e:
                res["bf16_config3"] = {"error": repr(e)}
            try:
                res["f16_config3"] = bf16_variant(args, torch, rt, params, x, dbg, sum(conv_flops(_L3, IM_H, IM_W)[0].values()), half="f16")
            except Exception as e:
                res["f16_config3"] = {"error": repr(e)}
                torch.cuda.synchronize()
        if world == 1 and args.dtype == "f32" and not args.no_split_variant:
            try:
                from chainer_faster_rcnn_amd.models.vgg16 import LAYERS as _L
                res["f32_split_products"] = split_variant(args, torch, rt, params, x, dbg, sum(conv_flops(_L, IM_H, IM_W)[0].values()))
            except Exception as e:
                res["f32_split_products"] = {"error": repr(e)}
                torch.cuda.synchronize()
        # the secondary figures INSIDE `roofline` (the driver's stored record keeps `roofline` verbatim and only the names of the other blocks)
        if "roofline" in res:
            nr, b3, sp = res.get("nms_roi") or {}, res.get("bf16_config3") or {}, res.get("f32_split_products") or {}
            h3 = res.get("f16_config3") or {}
            sec = {"roi_pool_us": nr.get("roi_pool_us"), "roi_pool_frac_of_hbm_peak": nr.get("roi_pool_frac_of_hbm_peak"),
                   "roi_pool_us_in_pipeline": nr.get("roi_pool_us_in_pipeline_stage_event"),
                   "roi_pool_us_behind_nms_in_one_graph": nr.get("roi_pool_us_behind_nms_in_one_graph"),
                   "roi_pool_frac_of_hbm_peak_behind_nms_in_one_graph": nr.get("roi_pool_frac_of_hbm_peak_behind_nms_in_one_graph"),
                   "roi_pool_fwd_argmax_us": nr.get("roi_pool_fwd_argmax_us"), "roi_pool_fwd_argmax_frac_of_hbm_peak": nr.get("roi_pool_fwd_argmax_frac_of_hbm_peak"),
                   "roi_pool_bwd_us": nr.get("roi_pool_bwd_us"), "roi_pool_bwd_frac_of_hbm_peak": nr.get("roi_pool_bwd_frac_of_hbm_peak"),
                   "proposals_nms_us": nr.get("proposals_nms_us"),
                   "f16_img_s": h3.get("value"), "f16_conv5_3_rel_err": (h3.get("parity") or {}).get("conv5_3_rel_err"),
                   "f16_from_image_index_match_set": (h3.get("parity") or {}).get("from_image_index_match_set"),
                   "f16_from_image_index_match_positional": (h3.get("parity") or {}).get("from_image_index_match_positional"),
                   "bf16_conv5_3_rel_err": (b3.get("parity") or {}).get("conv5_3_rel_err"),
                   "bf16_from_image_index_match_set": (b3.get("parity") or {}).get("from_image_index_match_set"),
                   "bf16_img_s": b3.get("value"), "bf16_ms_per_step": b3.get("ms_per_step"), "bf16_conv_ms_per_image": b3.get("conv_ms_per_image"),
                   "bf16_conv_frac_of_bf16_mfma_peak": b3.get("frac_of_bf16_mfma_peak"),
                   "f32s_img_s": sp.get("value"), "f32s_ms_per_step": sp.get("ms_per_step"),
                   "img_s_with_feed": (res.get("with_feed") or {}).get("img_s_with_feed"),
                   "bf16_img_s_with_feed": (b3.get("with_feed") or {}).get("img_s_with_feed"),
                   "img_s_two_images_in_flight": (res.get("two_images_in_flight") or {}).get("img_s_two_images_in_flight"),
                   "bf16_img_s_two_images_in_flight": (b3.get("two_images_in_flight") or {}).get("img_s_two_images_in_flight"),
                   "img_s_two_images_in_flight_with_feed": (res.get("two_images_in_flight") or {}).get("img_s_two_images_in_flight_with_feed"),
                   "bf16_img_s_two_images_in_flight_with_feed": (b3.get("two_images_in_flight") or {}).get("img_s_two_images_in_flight_with_feed")}
            res["roofline"]["secondary"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in sec.items() if v is not None}
        emit_json_line(res)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
