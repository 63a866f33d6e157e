"""The RPN training step of train_rpn.py (train_rpn.py:140-182 + chainer's StandardUpdater / ParallelUpdater and
MomentumSGD(lr=0.001) + WeightDecay(0.0005)) as one object over the HIP kernels.

    trainer = RPNTrainer(model)                       # model: models.FasterRCNN with parameters loaded
    out = trainer.step(x, img_info, gt_boxes)         # forward, AnchorTargetLayer, losses, backward, all-reduce, update

What runs where:
  forward         trunk + rpn_conv_3x3 + heads on the MFMA conv kernel (csrc/conv.hip), activations kept in HBM
  targets         AnchorTargetLayer (device: csrc/train.hip; host: the NumPy-RNG subsample, as in the reference)
  losses          frcnn_rpn_loss: softmax-CE + Huber and their gradients w.r.t. the two head outputs
  backward        per conv: weight gradient (MFMA, pixels as the reduction axis), bias gradient, input gradient = the forward
                  kernel on 180-degree-rotated weights with the ReLU mask fused into the epilogue; max-pool: gather
  data parallel   all_reduce(SUM) of the flat fp32 gradient buffer (RCCL over xGMI; gloo on CPU), launched per bucket (three
                  contiguous tail ranges) while the backward pass is still running -- see RPNTrainer._plan_buckets; i.e. the
                  gradients are summed over replicas exactly as ParallelUpdater's addgrads does, then every rank applies
                  the identical update (replaces gather-to-main + copyparams broadcast)
  update          one fused MomentumSGD + WeightDecay launch over the flat parameter / gradient / velocity buffers

Trained parameters are the trunk and the RPN (in rpn_train mode FasterRCNN.__call__ returns before the head:
models/faster_rcnn.py:115-116).  Convolution weights live in the kernels' packed layout (Cin*9, Cout) while training;
`sync_params()` writes them back to Chainer's (Cout, Cin, 3, 3) arrays for snapshots.
ProposalLayer runs inside the training step exactly where the reference runs it (train-mode top-N 12000 / 2000,
region_proposal_network.py:123-126) and its result is discarded, as there; `run_proposal_layer=False` skips it (same gradients).
"""

import numpy as np

from . import tuning as _tuning
from .chainer_compat import unwrap
from .models.anchor_target_layer import AnchorTargetLayer


class _PoolArg(object):
    """What the backward pass of a pool FUSED into its convolution needs: the arg-max byte of every window and the pre-pool size."""

    def __init__(self, idx, H, W):
        self.idx, self.H, self.W = idx, int(H), int(W)


def trunk_forward(model, x, fuse_pools=True):
    """Trunk forward keeping every layer's input (the backward pass needs them) -> (feat, inputs).  A convolution that is followed by a
    pool runs conv + ReLU + pool as ONE launch (csrc/conv.hip, act 5) and keeps a byte per window instead of the pre-pool map: the
    pool's entry in `inputs` is then a _PoolArg (fuse_pools=False / FRCNN_TRAIN_FUSE_POOL=0: two launches, the pre-pool map kept --
    the tests that impose the device's decisions on a float64 pass read it)."""
    rt = model.rt
    layers = model.trunk.layers
    fuse = fuse_pools and _tuning.get("FRCNN_TRAIN_FUSE_POOL") != "0"
    inputs, h, skip = [], x, None
    for idx, l in enumerate(layers):
        if l == "pool":
            if skip is not None:                                      # applied inside the previous convolution
                inputs.append(skip)
                skip = None
                continue
            inputs.append(h)
            h = rt.maxpool2x2(h)
            continue
        inputs.append(h)
        link = model.trunk.links[l[0]]
        if fuse and idx + 1 < len(layers) and layers[idx + 1] == "pool" and int(link.cin) > 3 and int(link.cout) % 64 == 0:
            H, W = int(h.shape[2]), int(h.shape[3])
            h, arg = rt.conv_relu_pool_train(h, link.Wp, link.b)
            skip = _PoolArg(arg, H, W)
        else:
            h = link(h, relu=True)
    return h, inputs


def _grad_stream(rt, *arrays):
    import contextlib
    if _tuning.get("FRCNN_TRAIN_STREAMS") == "1":
        return contextlib.nullcontext()
    return rt.mem.aux_stream("grad", *arrays)


def trunk_backward(trainer, layer_inputs, g):
    """Backward through [(layer, input)] in reverse, starting from g = dL/d(output of the last layer) ALREADY masked by that
    layer's ReLU.  Writes weight / bias gradients into trainer.grad and re-packs the input-gradient weights."""
    rt = trainer.rt
    first = trainer.convs[0][0]
    links = dict(trainer.convs)
    unpool = _tuning.get("FRCNN_TRAIN_FUSE_POOL") != "0"
    skip_pool = False
    for pos in range(len(layer_inputs) - 1, -1, -1):
        l, xin = layer_inputs[pos]
        if l == "pool":
            if skip_pool:                                             # done in the epilogue of the input-gradient convolution above
                skip_pool = False
                continue
            if isinstance(xin, _PoolArg):                             # the pool ran inside its convolution: route by the kept bytes
                g = rt.maxpool2x2_bwd_idx(xin.idx, g, xin.H, xin.W)
                continue
            if getattr(trainer, "keep_dy", None) is not None:        # tests: the pre-pool maps (near-tie windows = where two fp32 passes may route differently)
                trainer.kept_dy.setdefault("pool_inputs", []).append(xin)
            g = rt.maxpool2x2_bwd(xin, g)
            continue
        name = l[0]
        keep = getattr(trainer, "keep_dy", None)
        if keep is not None and name in keep:                     # tests: the (input, upstream gradient) pair a weight gradient was computed from
            trainer.kept_dy[name] = (xin, g.clone() if hasattr(g, "clone") else g.copy())
        # this layer's weight / bias gradient on the gradient stream: it needs (xin, g) only, so it runs NEXT TO the input-gradient
        # convolution below (FRCNN_TRAIN_STREAMS=1: one stream, A/B hook)
        with _grad_stream(rt, xin, g):
            rt.conv_wgrad(xin, g, 3, out=trainer.grad[name + "/W"])
            rt.bias_grad(g, out=trainer.grad[name + "/b"])
            if hasattr(trainer, "_grads_ready"):
                trainer._grads_ready(name)                        # data parallel: a finished bucket starts its all-reduce now
        if name != first:                                         # the image needs no gradient
            if not getattr(trainer, "_dgrad_packed", False):          # (RPNTrainer re-packs every layer in one launch per step)
                rt.pack_conv_dgrad_w(links[name].Wp, 3, out=trainer.wd[name])
            below = layer_inputs[pos - 1] if pos > 0 else None
            if unpool and below is not None and below[0] == "pool" and isinstance(below[1], _PoolArg):
                # the layer below ran conv + ReLU + pool as one launch: its ReLU mask (pooled > 0) and the pool's routing both sit in
                # its arg-max bytes, and this convolution writes dL/d(pre-pool map) directly (csrc/conv.hip, act 6)
                pa = below[1]
                g = rt.conv_dgrad_unpool(g, trainer.wd[name], trainer.zero_bias, pa.idx, pa.H, pa.W)
                skip_pool = True
            else:
                g = rt.conv_ex(g, trainer.wd[name], trainer.zero_bias, 3, act=2, mask=xin)
    rt.mem.join_aux_stream("grad")
    return g


def _conv_dims(trainer, name):
    link = dict(trainer.convs)[name]
    return int(link.cin), int(link.cout)


def trunk_forward_split(trainer, x):
    """trunk_forward with the 3x3 convolutions (all but the 3-channel first one) computed as six bf16 MFMA products of 3-way split
    fp32 operands (csrc/conv_f32s.hip): every convolution leaves its result as fp32 NCHW (what the weight-gradient kernel, pooling
    and the ReLU masks read) and, when a convolution follows, as the split tensor that one reads.  -> (feat, inputs, feat_split)."""
    model, rt = trainer.model, trainer.rt
    layers = model.trunk.layers
    inputs, h, hs = [], x, None
    for idx, l in enumerate(layers):
        inputs.append(h)
        if l == "pool":
            h, hs = rt.maxpool2x2(h), None
            continue
        name, link = l[0], model.trunk.links[l[0]]
        if int(link.cin) <= 3:                                        # conv1_1: the first-layer kernel, fp32 NCHW image in
            if idx == 0 and int(link.cout) <= 64 and idx + 1 < len(layers) and layers[idx + 1] != "pool":
                hs, h = rt.conv1_f32s_train(h, link.Wp, link.b, link.cout, relu=True)
            else:
                h, hs = link(h, relu=True), None
            continue
        if hs is None:
            hs = rt.f32s_from_nchw(h)
        conv_next = idx + 1 == len(layers) or layers[idx + 1] != "pool"        # the last map feeds rpn_conv_3x3
        hs, h = rt.conv3x3_f32s_train(hs, trainer.ws_fwd[name], link.b, link.cin, link.cout, relu=True, want_split=conv_next)
    if hs is None:
        hs = rt.f32s_from_nchw(h)                                     # (trunks that end in a pool or in the first layer)
    return h, inputs, hs


def trunk_backward_split(trainer, layer_inputs, g):
    """trunk_backward with the input-gradient convolutions on split tensors; weight and bias gradients as before (fp32 NCHW)."""
    rt = trainer.rt
    first = trainer.convs[0][0]
    links = dict(trainer.convs)
    names = [l if l == "pool" else l[0] for l, _ in layer_inputs]
    gs = None
    for pos in range(len(layer_inputs) - 1, -1, -1):
        l, xin = layer_inputs[pos]
        if l == "pool":
            if getattr(trainer, "keep_dy", None) is not None:
                trainer.kept_dy.setdefault("pool_inputs", []).append(xin)
            g, gs = rt.maxpool2x2_bwd(xin, g), None
            continue
        name = l[0]
        keep = getattr(trainer, "keep_dy", None)
        if keep is not None and name in keep:
            trainer.kept_dy[name] = (xin, g.clone() if hasattr(g, "clone") else g.copy())
        with _grad_stream(rt, xin, g):                               # next to the input-gradient convolution below (see trunk_backward)
            rt.conv_wgrad_f32s(xin, g, out=trainer.grad[name + "/W"])  # split products too: fp32 NCHW in, the split happens in the kernel
            rt.bias_grad(g, out=trainer.grad[name + "/b"])
            if hasattr(trainer, "_grads_ready"):
                trainer._grads_ready(name)
        if name == first:
            continue                                                  # the image needs no gradient
        cin, cout = _conv_dims(trainer, name)
        if cin <= 3:
            continue
        if gs is None:
            gs = rt.f32s_from_nchw(g)
        below = names[pos - 1] if pos > 0 else None                   # who consumes dL/d(input): a convolution with an input gradient of its own?
        below_conv = below is not None and below != "pool" and below != first and _conv_dims(trainer, below)[0] > 3
        gs, g = rt.conv3x3_f32s_train(gs, trainer.ws_dgrad[name], trainer.zero_bias, cout, cin, relu=False, want_split=below_conv, mask=xin)
    rt.mem.join_aux_stream("grad")
    return g


class _Seg(object):
    def __init__(self, name, shape, offset):
        self.name, self.shape, self.offset = name, tuple(shape), offset
        self.size = int(np.prod(shape))


class _ParamArena(object):
    """A trainer keeps its parameters in ONE flat buffer (self.W) and re-points the links' arrays at windows of it, so a single
    fused launch updates everything.  Two trainers on one model (the reference's rpn -> rcnn -> rpn alternation, train.py; the
    hidden RCNNTrainer of FasterRCNN.__call__) would orphan each other's windows: before every step a trainer therefore checks
    that each link still points INTO its buffer and re-adopts the link's current values if not (velocities are kept)."""

    def _adopt(self, key, current):
        seg = self.seg[key]
        v = self.rt.mem.view(self.W, seg.offset, seg.shape)
        if not self.rt.mem.within(current, self.W):
            v[...] = current
        return v

    def _ensure_adopted(self):
        raise NotImplementedError


class _BucketedAllReduce(_ParamArena):
    """Data parallel: the gradient all-reduce as a few contiguous TAIL buckets of the flat buffer, each launched asynchronously
    the moment the layer that completes it has its gradient kernels enqueued (the buffer is laid out in forward order and the
    backward pass fills it from the end), so the exchange runs under the rest of the backward pass.  Same element-wise sums over
    ranks as one all-reduce.  Few large buckets, not many small ones: xGMI is point-to-point, a ring collective is per-link bound,
    and every extra collective adds its latency."""

    def _plan_buckets(self, names, n_buckets=3):
        """names: parameter groups in FORWARD (= buffer) order; every group's gradients precede, in time, those of the groups
        before it.  Buckets are (closing group, start, end) in backward order and tile [0, n_flat)."""
        total, end = self.n_flat, self.n_flat
        target = total / float(n_buckets)
        self.buckets = []
        for i in range(len(names) - 1, -1, -1):
            start = self.seg[names[i] + "/W"].offset
            if end - start >= target or i == 0:
                self.buckets.append((names[i], start if i else 0, end))
                end = start
        self._closing = {b[0]: k for k, b in enumerate(self.buckets)}
        self._works = []

    def _grads_ready(self, name):
        k = self._closing.get(name)
        if k is None or self.comm is None or not getattr(self.comm, "active", self.comm.world_size > 1):
            return
        _, start, end = self.buckets[k]
        self._works.append(self.comm.all_reduce_sum_async(self.G[start:end]))

    def _drain(self):
        for w in self._works:
            self.comm.wait(w)
        self._works = []

    def all_reduce(self):
        """Sum the gradient buffer over the replicas (ParallelUpdater: grads are added, not averaged).  The buckets were launched
        during the backward pass (_grads_ready); this waits for them -- the update must see every sum."""
        if self.comm is None or not getattr(self.comm, "active", self.comm.world_size > 1):
            return
        complete = len(self._works) == len(self.buckets)
        self._drain()
        if not complete:                                          # forward_backward was not the producer (e.g. a hand-filled G)
            self.comm.all_reduce_sum(self.G)


class RPNTrainer(_BucketedAllReduce):
    def __init__(self, model, lr=0.001, momentum=0.9, weight_decay=0.0005, comm=None, run_proposal_layer=True, conv_math="mfma"):
        """conv_math: "mfma" = forward and input-gradient convolutions on the fp32 MFMA kernel; "split" = the same fp32 convolutions as
        six bf16 MFMA products of 3-way split operands (csrc/conv_f32s.hip), the 3x3 weight gradients likewise (csrc/train.hip
        conv_wgrad_f32s_kernel)."""
        self.model, self.rt = model, model.rt
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.comm = comm
        self.conv_math = conv_math
        self.run_proposal_layer = run_proposal_layer
        self.proposals = None
        rt = self.rt
        rpn = model.RPN
        self.atl = AnchorTargetLayer(rpn.proposal_layer._feat_stride, runtime=rt)
        self.atl._anchors = rpn.proposal_layer._anchors
        self.atl._num_anchors = rpn.proposal_layer._num_anchors
        self.layers = model.trunk.layers
        self.convs = [(l[0], model.trunk.links[l[0]]) for l in self.layers if l != "pool"] + [("rpn_conv_3x3", rpn.rpn_conv_3x3)]
        # ---- one flat buffer each for parameters, gradients, velocities; every segment 256-byte aligned
        segs, off = [], 0
        def add(name, shape):
            nonlocal off
            s = _Seg(name, shape, off)
            segs.append(s)
            off += (s.size + 63) // 64 * 64
            return s
        self.seg = {}
        for name, link in self.convs:
            self.seg[name + "/W"] = add(name + "/W", link.Wp.shape)
            self.seg[name + "/b"] = add(name + "/b", link.b.shape)
        wp, bp, self.A = rpn._heads_packed
        self.seg["heads/W"] = add("heads/W", wp.shape)
        self.seg["heads/b"] = add("heads/b", bp.shape)
        self.n_flat = off
        self.W = rt.mem.zeros((off,), "f32")
        self.G = rt.mem.zeros((off,), "f32")
        self.V = rt.mem.zeros((off,), "f32")
        self._ensure_adopted()
        self.grad = {k: rt.mem.view(self.G, s.offset, s.shape) for k, s in self.seg.items()}
        # weights of the input-gradient convolutions (re-packed from the current weights every step)
        self.wd = {name: rt.mem.empty((int(link.Wp.shape[1]) * 9, int(link.Wp.shape[0]) // 9), "f32") for name, link in self.convs[1:]}
        self.wd_heads = rt.mem.empty((int(wp.shape[1]), int(wp.shape[0])), "f32")
        self.zero_bias = rt.mem.zeros((512,), "f32")
        if conv_math == "split":                                      # split weights of the forward / input-gradient convolutions (re-packed every step)
            pad = rt.bf16_pad
            big = [(n, l) for n, l in self.convs if int(l.cin) > 3]
            self.ws_fwd = {n: rt.mem.empty((3, pad(l.cin) // 16, 9, pad(l.cout), 16), "i16") for n, l in big}
            self.ws_dgrad = {n: rt.mem.empty((3, pad(l.cout) // 16, 9, pad(l.cin), 16), "i16") for n, l in big}
        self._draw = None
        self.iteration = 0
        self._plan_buckets([n for n, _ in self.convs])       # heads follow rpn_conv_3x3 in the buffer and precede it in time

    def _ensure_adopted(self):
        rpn = self.model.RPN
        for name, link in self.convs:
            link.Wp = self._adopt(name + "/W", link.Wp)
            link.b = self._adopt(name + "/b", link.b)
            link._adopted = True
        wp, bp, A = rpn._heads_packed
        rpn._heads_packed = (self._adopt("heads/W", wp), self._adopt("heads/b", bp), A)
        rpn._heads_adopted = True

    # ------------------------------------------------------------------
    def forward_backward(self, x, img_info, gt_boxes):
        """Fills self.G with this replica's gradients; returns dict(loss, loss_cls, loss_bbox, accuracy) (device scalars)."""
        rt, model, rpn = self.rt, self.model, self.model.RPN
        self._drain()                                              # a previous backward whose sums were never consumed
        self._ensure_adopted()
        x = rt.asarray(unwrap(x), "f32")
        im_h, im_w = rpn.proposal_layer._img_hw(img_info)
        # ---- targets first, on their own stream: AnchorTargetLayer depends on the ground truth and the map size only, and its fg / bg
        #      subsample runs on the HOST (NumPy's global RNG, the reference's call sequence) -- a device round trip.  Python runs a whole
        #      step ahead of the GPU, so here the round trip overlaps the previous step's backward pass; after the forward pass (where
        #      the reference has it) it drained the stream and left the GPU idle while the host sampled (0.4 ms per step).  One draw
        #      per step either way: the RNG sequence is the reference's.
        H, W = int(x.shape[2]), int(x.shape[3])
        for l in self.layers:
            if l == "pool":
                H, W = (H + 1) // 2, (W + 1) // 2
        with rt.mem.early_stream(unwrap(gt_boxes)):                   # its own stream: it must not wait for the previous step's backward
            labels, targets, inds, n_in, _ = self.atl.forward_device(H, W, gt_boxes, im_h, im_w)      # (host ground truth; a DEVICE array is waited for)
        rt.mem.join_early_stream(labels, targets, inds)
        if self.conv_math == "split":
            # split weights of every forward / input-gradient convolution from the current (packed fp32) weights: one launch
            rt.f32s_pack_many([(l.Wp, self.ws_fwd[n], self.ws_dgrad[n], l.cin, l.cout) for n, l in self.convs if int(l.cin) > 3])
            feat, inputs, feat_split = trunk_forward_split(self, x)
            link = rpn.rpn_conv_3x3
            _, mid = rt.conv3x3_f32s_train(feat_split, self.ws_fwd["rpn_conv_3x3"], link.b, link.cin, link.cout, relu=True, want_split=False)
        else:
            # weights of every input-gradient convolution (rotated / transposed copies of the current packed weights): one launch
            if _tuning.get("FRCNN_DGRAD_PACK") != "each":          # (=each: A/B hook, one launch per layer inside the backward pass)
                with _grad_stream(rt):                                # on the gradient stream: under the forward pass, joined before the backward pass
                    rt.pack_conv_dgrad_w_many([(l.Wp, self.wd[n], 3) for n, l in self.convs[1:]] + [(rpn._heads_packed[0], self.wd_heads, 1)])
                self._dgrad_packed = True
            feat, inputs = trunk_forward(model, x, fuse_pools=getattr(self, "keep_dy", None) is None)     # keeps every layer's input
            mid = rpn.rpn_conv_3x3(feat, relu=True)
        score, prob, bbox = rt.rpn_heads(mid, rpn._heads_packed)
        if getattr(self, "keep_dy", None) is not None:               # tests: every activation map a ReLU / max-pool decision was taken on
            self.kept_dy["rpn_mid"] = mid
            self.kept_dy["layer_inputs"] = list(inputs) + [feat]
        if self.run_proposal_layer:
            # region_proposal_network.py:123-126: `proposals, probs = self.proposal_layer(...)` in train mode (12000 -> NMS -> 2000);
            # nothing downstream consumes it in rpn_train mode (faster_rcnn.py:115-116 returns the loss) -- kept for inspection.
            # Its sequential NMS pass keeps one wave busy for ~0.5 ms: it runs on a second stream, under the backward pass.
            with rt.mem.side_stream(prob, bbox):
                self.proposals = rpn.proposal_layer.forward_device(prob, bbox, im_h, im_w)
        A = self.A
        assert (H, W) == (int(feat.shape[2]), int(feat.shape[3]))
        NP = int(rpn._heads_packed[0].shape[1])
        # ---- losses
        if self._draw is None or tuple(self._draw.shape) != (NP, H, W):
            self._draw = rt.mem.zeros((NP, H, W), "f32")             # rows >= 6A (channel padding) stay zero
        draw = self._draw
        losses, _, _ = rt.rpn_loss(score[0], bbox[0], labels, targets, inds, n_in, A, H, W, rpn._delta, rpn._loss_lambda,
                                   d_score=draw[:2 * A], d_bbox=draw[2 * A:6 * A])
        # ---- backward: heads (one 1x1 convolution over the stacked cls|bbox matrix)
        rt.mem.join_aux_stream("grad")                               # the re-packed input-gradient weights are ready
        with _grad_stream(rt, mid, draw):
            rt.conv_wgrad(mid, draw, 1, out=self.grad["heads/W"])
            rt.bias_grad(draw, out=self.grad["heads/b"])
        if not getattr(self, "_dgrad_packed", False):
            rt.pack_conv_dgrad_w(rpn._heads_packed[0], 1, out=self.wd_heads)
        g = rt.conv_ex(draw.reshape(1, NP, H, W), self.wd_heads, self.zero_bias, 1, act=2, mask=mid)
        # ---- rpn_conv_3x3, then the trunk in reverse
        (trunk_backward_split if self.conv_math == "split" else trunk_backward)(self, list(zip(self.layers, inputs)) + [(("rpn_conv_3x3", 0, 0), feat)], g)
        self._dgrad_packed = False
        if self.run_proposal_layer:
            rt.mem.join_side_stream()
        return dict(losses=losses)

    def update(self):
        self._ensure_adopted()
        self.rt.sgd_momentum_wd(self.W, self.G, self.V, self.lr, self.momentum, self.weight_decay)
        if hasattr(self.model, "mark_params_updated"):
            self.model.mark_params_updated(self)
        self.iteration += 1

    def step(self, x, img_info, gt_boxes):
        out = self.forward_backward(x, img_info, gt_boxes)
        self.all_reduce()
        self.update()
        return out

    def losses_host(self, out):
        l = self.rt.mem.to_numpy(out["losses"])
        return dict(rpn_loss_cls=float(l[0]), rpn_loss_bbox=float(l[1]), rpn_cls_accuracy=float(l[2]),
                    rpn_loss=float(l[0] + self.model.RPN._loss_lambda * l[1]))

    # ------------------------------------------------------------------
    def sync_params(self):
        """Packed training weights -> Chainer-layout arrays on the links (for snapshots / inference objects)."""
        rt = self.rt
        for name, link in self.convs:
            co, ci = link.cout, link.cin
            link.W = rt.transpose(link.Wp).reshape(co, ci, 3, 3)
        wp, bp, A = self.model.RPN._heads_packed
        wt = rt.transpose(wp)                                        # (NP, Cmid)
        rpn = self.model.RPN
        rpn.rpn_cls_score["W"], rpn.rpn_cls_score["b"] = wt[:2 * A], bp[:2 * A]
        rpn.rpn_bbox_pred["W"], rpn.rpn_bbox_pred["b"] = wt[2 * A:6 * A], bp[2 * A:6 * A]

    def flat_to_chainer_layout(self, flat):
        """A flat buffer in this trainer's segment layout (gradients, velocities, parameters) -> {link path: host array in
        Chainer's layout}: packed (Cin*9, Cout) -> (Cout, Cin, 3, 3); the stacked heads -> rpn_cls_score / rpn_bbox_pred."""
        rt, out = self.rt, {}
        host = rt.mem.to_numpy(flat)
        def seg(key):
            sg = self.seg[key]
            return host[sg.offset:sg.offset + sg.size].reshape(sg.shape)
        for name, link in self.convs:
            prefix = "RPN/" if name == "rpn_conv_3x3" else "trunk/"
            out[prefix + name + "/W"] = np.ascontiguousarray(seg(name + "/W").T).reshape(link.cout, link.cin, 3, 3)
            out[prefix + name + "/b"] = seg(name + "/b").copy()
        A = self.A
        gw, gb = seg("heads/W").T, seg("heads/b")
        out["RPN/rpn_cls_score/W"], out["RPN/rpn_cls_score/b"] = np.ascontiguousarray(gw[:2 * A]).reshape(2 * A, -1, 1, 1), gb[:2 * A].copy()
        out["RPN/rpn_bbox_pred/W"], out["RPN/rpn_bbox_pred/b"] = np.ascontiguousarray(gw[2 * A:6 * A]).reshape(4 * A, -1, 1, 1), gb[2 * A:6 * A].copy()
        return out

    def chainer_layout_to_flat(self, arrays, flat):
        """Inverse of flat_to_chainer_layout: write {link path: array} into `flat` (missing keys leave their segment untouched)."""
        rt = self.rt
        host = rt.mem.to_numpy(flat).copy()
        def put(key, val):
            sg = self.seg[key]
            host[sg.offset:sg.offset + sg.size] = np.asarray(val, dtype=np.float32).reshape(-1)
        for name, link in self.convs:
            prefix = "RPN/" if name == "rpn_conv_3x3" else "trunk/"
            if prefix + name + "/W" in arrays:
                put(name + "/W", np.asarray(arrays[prefix + name + "/W"], dtype=np.float32).reshape(link.cout, -1).T)
            if prefix + name + "/b" in arrays:
                put(name + "/b", arrays[prefix + name + "/b"])
        A = self.A
        sw, sb = self.seg["heads/W"], self.seg["heads/b"]
        if "RPN/rpn_cls_score/W" in arrays and "RPN/rpn_bbox_pred/W" in arrays:
            w = np.zeros((sw.shape[1], sw.shape[0]), np.float32)                       # (NP, Cmid): rows past 6A are channel padding
            w[:2 * A] = np.asarray(arrays["RPN/rpn_cls_score/W"], dtype=np.float32).reshape(2 * A, -1)
            w[2 * A:6 * A] = np.asarray(arrays["RPN/rpn_bbox_pred/W"], dtype=np.float32).reshape(4 * A, -1)
            put("heads/W", w.T)
            b = np.zeros((sb.shape[0],), np.float32)
            b[:2 * A], b[2 * A:6 * A] = arrays["RPN/rpn_cls_score/b"], arrays["RPN/rpn_bbox_pred/b"]
            put("heads/b", b)
        flat[...] = rt.mem.from_numpy(host)

    def grads_chainer_layout(self):
        """{link path: gradient in Chainer's layout} (host arrays) -- for tests and inspection."""
        return self.flat_to_chainer_layout(self.G)


class RCNNTrainer(_BucketedAllReduce):
    """Stage-2 step of train_rcnn.py (train_rcnn.py:35-78; models/faster_rcnn.py:110-173 with rcnn_train = True): trunk -> RPN
    proposals (test-mode ProposalLayer, no gradient) -> RoI pooling -> fc6/fc7 with dropout -> cls_score / bbox_pred ->
    ProposalTargetLayer sampling -> softmax-CE + Huber(1) on the sampled rows -> backward through the head (the four L.Linear
    as GEMMs on transposed operands), RoI pooling (arg-max scatter), the trunk -> MomentumSGD + WeightDecay over trunk + head.

    Dropout masks are drawn on the host from NumPy's global RNG in the order chainer's CPU path draws them (fc6's mask, fc7's
    mask, then ProposalTargetLayer's two np.random.choice calls), so a seeded run follows the reference's random stream.
    """

    HEAD = ("fc6", "fc7", "cls_score", "bbox_pred")

    def __init__(self, model, lr=0.001, momentum=0.9, weight_decay=0.0005, dropout_ratio=0.5, comm=None, conv_math="mfma", dropout_rng="numpy",
                 dropout_seed=0):
        """dropout_rng: "numpy" (default) draws both masks on the host from NumPy's global stream exactly as chainer's CPU F.dropout does (two
        np.random.rand calls of n_rois x 4096 values per step: ~7 ms of host time at 300 RoIs, bench.py --mode train-rcnn); "device" draws them
        in the dropout kernel from a counter-based hash of (dropout_seed, step, layer) -- the throughput form, no host work, no H2D.
        conv_math: as in RPNTrainer -- "mfma" = the trunk's forward / input-gradient / weight-gradient convolutions on the fp32 MFMA
        kernels, "split" = the same fp32 convolutions as six bf16 MFMA products of 3-way split operands (csrc/conv_f32s.hip)."""
        from .models.proposal_target_layer import ProposalTargetLayer
        assert conv_math in ("mfma", "split")
        self.conv_math = conv_math
        self.model, self.rt = model, model.rt
        self.lr, self.momentum, self.weight_decay, self.dropout_ratio, self.comm = lr, momentum, weight_decay, dropout_ratio, comm
        assert dropout_rng in ("numpy", "device")
        self.dropout_rng, self.dropout_seed = dropout_rng, int(dropout_seed)
        rt = self.rt
        self.ptl = ProposalTargetLayer(model._feat_stride, num_classes=model._num_classes, runtime=rt)
        self.layers = model.trunk.layers
        self.convs = [(l[0], model.trunk.links[l[0]]) for l in self.layers if l != "pool"]
        segs, off = {}, 0
        def add(name, shape):
            nonlocal off
            segs[name] = _Seg(name, shape, off)
            off += (segs[name].size + 63) // 64 * 64
        for name, link in self.convs:
            add(name + "/W", link.Wp.shape)
            add(name + "/b", link.b.shape)
        for n in self.HEAD:
            lin = getattr(model, n)
            add(n + "/W", lin.W.shape)
            add(n + "/b", lin.b.shape)
        self.seg, self.n_flat = segs, off
        self.W, self.G, self.V = rt.mem.zeros((off,), "f32"), rt.mem.zeros((off,), "f32"), rt.mem.zeros((off,), "f32")
        self._ensure_adopted()
        self.grad = {k: rt.mem.view(self.G, sg.offset, sg.shape) for k, sg in segs.items()}
        self.wd = {name: rt.mem.empty((int(link.Wp.shape[1]) * 9, int(link.Wp.shape[0]) // 9), "f32") for name, link in self.convs[1:]}
        self.zero_bias = rt.mem.zeros((max(512, max(int(getattr(model, n).W.shape[1]) for n in self.HEAD)),), "f32")
        if conv_math == "split":                                      # split weights of the forward / input-gradient convolutions (re-packed every step)
            pad = rt.bf16_pad
            big = [(n, l) for n, l in self.convs if int(l.cin) > 3]
            self.ws_fwd = {n: rt.mem.empty((3, pad(l.cin) // 16, 9, pad(l.cout), 16), "i16") for n, l in big}
            self.ws_dgrad = {n: rt.mem.empty((3, pad(l.cout) // 16, 9, pad(l.cin), 16), "i16") for n, l in big}
        self.iteration = 0
        # the head (fc6: 411 MB of gradients) is complete before the trunk's backward starts: its bucket rides under all of it
        self._plan_buckets([n for n, _ in self.convs] + list(self.HEAD))

    def _ensure_adopted(self):
        for name, link in self.convs:
            link.Wp = self._adopt(name + "/W", link.Wp)
            link.b = self._adopt(name + "/b", link.b)
            link._adopted = True
        for n in self.HEAD:
            lin = getattr(self.model, n)
            lin.W = self._adopt(n + "/W", lin.W)
            lin.b = self._adopt(n + "/b", lin.b)
            lin._adopted = True

    # ------------------------------------------------------------------ one L.Linear backward: GEMMs on transposed operands
    def _linear_backward(self, name, x, dy, need_dx=True):
        """x (M,K) the layer's input, dy (M,N): fills grad[name/W] (N,K), grad[name/b]; returns dx (M,K) or None."""
        rt = self.rt
        lin = getattr(self.model, name)
        M, K = int(x.shape[0]), int(np.prod(x.shape[1:]))
        N = int(dy.shape[1])
        Mp = (M + 3) // 4 * 4                                        # the GEMM contracts over M here: pad it to a multiple of 4
        if Mp == M:                                                  # (the 128 kept rows of a full sample: nothing to pad, four launches fewer)
            xpad, dpad = x.reshape(M, K), dy
        else:
            xpad, dpad = rt.mem.zeros((Mp, K), "f32"), rt.mem.zeros((Mp, N), "f32")
            xpad[:M] = x.reshape(M, K)
            dpad[:M] = dy
        dyT, xT = rt.transpose(dpad), rt.transpose(xpad)             # (N, Mp), (K, Mp)
        rt.linear(dyT, xT, None, out=self.grad[name + "/W"])                        # dW = dy^T x (no bias term: a single K slab goes straight into the gradient)
        rt.bias_grad(dyT.reshape(1, N, 1, Mp), out=self.grad[name + "/b"])          # db = column sums of dy
        if not need_dx:
            return None
        if N % 64 == 0 and K % 64 == 0 and M <= 4096 and _tuning.get("FRCNN_LINEAR_DX", "conv") == "conv":
            # dx = dy W with W (N, K) READ AS STORED: a 1x1 convolution whose "image" is W itself -- N channels of K "pixels" -- and whose
            # packed weights (Cin, Cout) are dy^T padded to 64 output channels: y[m][k] = sum_n dy[m][n] W[n][k].  No transposed copy of W
            # (fc6: 411 MB read + written every step), and W is streamed exactly once.
            Mc = (M + 63) // 64 * 64
            wh = 1
            for c in (256, 224, 128, 64):                          # a map shape for the kernel's 2-D tiles; any factorisation of K is the same sum
                if K % c == 0:
                    wh = c
                    break
            if Mc == Mp:
                dycT = dyT                                           # already (N, Mc)
            else:
                dyc = rt.mem.zeros((Mc, N), "f32")
                dyc[:M] = dy
                dycT = rt.transpose(dyc)
            dx = rt.conv_ex(lin.W.reshape(1, N, K // wh, wh), dycT, self.zero_bias[:Mc], ksize=1, act=0)
            return dx.reshape(Mc, K)[:M]
        Np = (N + 3) // 4 * 4                                        # dx = dy W contracts over N: pad it to a multiple of 4 as well
        if Np != N:
            dyn, wn = rt.mem.zeros((M, Np), "f32"), rt.mem.zeros((Np, K), "f32")
            dyn[:, :N] = dy
            wn[:N] = lin.W
        else:
            dyn, wn = dy, lin.W
        return rt.linear(dyn, rt.transpose(wn), None)                               # W^T is (K, Np)

    def forward_backward(self, x, img_info, gt_boxes, masks=None):
        """Fills self.G; returns dict(losses (3,) device [loss_cls, loss_bbox, cls_accuracy], n_rois, keep_inds)."""
        rt, model = self.rt, self.model
        stage = getattr(self, "stage_hook", None) or (lambda name: None)      # bench.py --mode train-rcnn: a HIP event per stage boundary
        self._drain()                                              # a previous backward whose sums were never consumed
        self._ensure_adopted()
        x = rt.asarray(unwrap(x), "f32")
        im_h, im_w = model.RPN.proposal_layer._img_hw(img_info)
        # the gt boxes go up FIRST and without blocking the host: everything up to the one host read behind the ProposalLayer is then enqueued while
        # the trunk still runs (a pageable copy issued where the boxes are needed stalled the host -- and, behind it, the GPU -- in the middle of the step)
        gt_host = unwrap(gt_boxes)
        gt_host = np.ascontiguousarray(rt.mem.to_numpy(gt_host) if rt.mem.is_array(gt_host) else np.asarray(gt_host), dtype=np.float32)
        gt_dev64 = rt.mem.from_numpy_async(np.ascontiguousarray(gt_host[0][:, :4], dtype=np.float64))
        stage("start")
        if self.conv_math == "split":
            big = [(n, l) for n, l in self.convs if int(l.cin) > 3]
            if big:
                rt.f32s_pack_many([(l.Wp, self.ws_fwd[n], self.ws_dgrad[n], l.cin, l.cout) for n, l in big])
            feat, inputs, _ = trunk_forward_split(self, x)
        else:
            # weights of every input-gradient convolution (rotated / transposed copies of the current packed weights): ONE launch on the gradient
            # stream, under the forward pass -- as in RPNTrainer (one launch per layer inside the backward pass was 12 x 17 us on its critical path)
            if _tuning.get("FRCNN_DGRAD_PACK") != "each":
                with _grad_stream(rt):
                    rt.pack_conv_dgrad_w_many([(l.Wp, self.wd[n], 3) for n, l in self.convs[1:]])
                self._dgrad_packed = True
            feat, inputs = trunk_forward(model, x, fuse_pools=getattr(self, "keep_dy", None) is None)
        C, H, W = [int(v) for v in feat.shape[1:]]
        stage("trunk_fwd")
        _, _, prob, bbox = model.RPN.heads(feat, want_score=False)
        rois, _, n_out = model.RPN.proposal_layer.forward_device(prob, bbox, im_h, im_w)      # RPN.train is False in rcnn_train mode
        # ProposalTargetLayer needs the RoIs and the gt boxes only: its float64 IoU matrix is enqueued here and comes back with the RoI count (the
        # one host read the step needs anyway), so the host-side sampling further down overlaps the head's forward pass instead of waiting for it
        if self.ptl.type_check_enable:
            self.ptl._check_data_type_forward(rois, gt_boxes)
        ov_dev = self.ptl.overlaps_device(rois, gt_dev64)
        fetch = rt.mem.to_numpy_many_async([ov_dev, rois, n_out])                            # one device -> host copy, one host round trip
        scale = 1.0 / (1.0 - self.dropout_ratio)
        on_device = masks is None and self.dropout_rng == "device"

        def collect():
            ov_host, rois_host, n_host = fetch()
            n = int(n_host[0])
            assert n > 0, "the ProposalLayer returned no RoI (proposal_target_layer.py:52 asserts the same)"
            return n, rois_host[:n], ov_host[:n]

        def head_forward(rois, n):
            """RoI pooling -> fc6 -> dropout -> fc7 -> dropout -> cls_score / bbox_pred on n rows."""
            pool5, argmax = rt.roi_pool_fwd_chw(feat, rois, 7, 7, model._spatial_scale, want_argmax=True)
            stage("roi_pool_fwd")
            pool5 = pool5.reshape(n, -1)
            a6 = model.fc6(pool5, relu=True)
            if on_device:                                            # one launch: mask drawn, stored and applied
                # one counter per FORWARD (two draws each) -- not per update: forward_backward twice without update() must not reuse its masks -- and
                # the data-parallel rank folded in: every rank draws its own masks from one dropout_seed
                # the stream is a function of (dropout_seed, rank, iteration, forward-within-iteration) -- a trainer rebuilt mid-run (resume: `iteration`
                # comes back from the snapshot) continues the mask sequence instead of replaying step 0's; the within-iteration index resets when update() advances
                # `iteration`
                if getattr(self, "_dropout_iter", None) != self.iteration:
                    self._dropout_iter, self._dropout_fwd = self.iteration, 0
                fwd = self._dropout_fwd
                self._dropout_fwd += 1
                rank = int(getattr(getattr(self, "comm", None), "rank", 0) or 0)
                base = (self.dropout_seed * 0x100000001b3 + rank * 0x9E3779B97F4A7C15 + (int(self.iteration) << 20) + 2 * fwd) & 0xFFFFFFFFFFFFFFFF
                d6, m6 = rt.dropout(a6, self.dropout_ratio, base)
            else:
                if masks is None:                                    # F.dropout [chainer-ext]: mask = (rand >= ratio) * 1/(1-ratio)
                    m6 = ((np.random.rand(*a6.shape) >= self.dropout_ratio) * scale).astype(np.float32)
                else:
                    m6 = masks[0]
                m6 = rt.asarray(m6, "f32")
                d6 = rt.mul(a6, m6)
            a7 = model.fc7(d6, relu=True)
            if on_device:
                d7, m7 = rt.dropout(a7, self.dropout_ratio, (base + 1) & 0xFFFFFFFFFFFFFFFF)
            else:
                m7 = ((np.random.rand(*a7.shape) >= self.dropout_ratio) * scale).astype(np.float32) if masks is None else masks[1]
                m7 = rt.asarray(m7, "f32")
                d7 = rt.mul(a7, m7)
            cls_score, bbox_pred = model.cls_score(d7), model.bbox_pred(d7)
            stage("head_fwd")
            return pool5, argmax, a6, d6, m6, a7, d7, m7, cls_score, bbox_pred

        if on_device and _tuning.get("FRCNN_RCNN_HEAD_FWD", "early") == "early":
            # Device-drawn masks: nothing on the host has to know the RoI count before the head's forward pass, so it is enqueued at the
            # ProposalLayer's CAPACITY (rows past the count are zero boxes -- detect.hip defines them -- and a row's results, its mask included,
            # do not depend on the row count) BEFORE the host waits for its copy: the round trip, the sampling and the enqueueing of the losses
            # and the head's backward pass all run under 0.65 ms of GPU work instead of in front of it.  (The NumPy-stream masks are
            # (n, 4096) draws in the reference's order: that form reads the count first.)
            stage("rpn_proposals")
            cap = int(rois.shape[0])
            fw = head_forward(rois, cap)
            n, rois_host, ov_host = collect()
            if n < cap:
                fw = tuple(t[:n] for t in fw)
            rois = rois[:n]
        else:
            n, rois_host, ov_host = collect()
            rois = rois[:n]
            stage("rpn_proposals")
            fw = head_forward(rois, n)
        pool5, argmax, a6, d6, m6, a7, d7, m7, cls_score, bbox_pred = fw
        # host work under the head's forward pass: np.random.choice in the reference's call order (after the two dropout draws)
        use_gt, ext, keep = self.ptl.sample(np.ascontiguousarray(rois_host, dtype=np.float32), gt_host[0], overlaps=ov_host)
        k = int(keep.shape[0])
        blob = np.empty((k * (2 + ext.shape[1]),), dtype=np.float32)                         # keep | labels | targets: ONE host -> device copy
        blob[:k] = np.ascontiguousarray(keep, dtype=np.int32).view(np.float32)
        blob[k:2 * k] = use_gt[:, -1].astype(np.int32).view(np.float32)                      # faster_rcnn.py:153
        blob[2 * k:] = ext.reshape(-1)
        blob = rt.mem.from_numpy_async(blob)                                                 # the host keeps running ahead of the head's forward pass
        keep, labels = rt.mem.bitcast(blob[:k], "i32"), rt.mem.bitcast(blob[k:2 * k], "i32")
        ext = blob[2 * k:].reshape(k, -1)
        losses, dcs, dbp = rt.rcnn_loss(rt.gather_rows(cls_score, keep), rt.gather_rows(bbox_pred, keep), labels, ext, model._rcnn_delta)
        stage("targets_loss")
        # ---- backward: head.  Only the sampled rows carry a gradient (faster_rcnn.py:155-160: the losses see cls_score[keep_inds] and
        # bbox_pred[keep_inds]; get_item's adjoint leaves every other row of the 300 EXACTLY zero), so the four L.Linear backward passes, the
        # dropout / ReLU adjoints and the RoI-pooling scatter run on the k <= 128 kept rows: the same sums without their zero terms
        # (fc6's two 61.7-GFLOP GEMMs become two of 26.3).  FRCNN_RCNN_BWD_ROWS=all runs them over all n rows (A/B; the zero-padded form).
        if 0 < k < n and _tuning.get("FRCNN_RCNN_BWD_ROWS", "kept") != "all":
            rows = k
            dcls, dbb = dcs, dbp
            d7b, m7b = rt.gather_rows(d7, keep), rt.gather_rows(m7, keep)
            d6b, m6b = rt.gather_rows(d6, keep), rt.gather_rows(m6, keep)
            pool5b, argmaxb = rt.gather_rows(pool5, keep), rt.gather_rows(argmax.reshape(n, -1), keep)
        else:
            rows = n
            dcls, dbb = rt.scatter_rows(dcs, keep, n), rt.scatter_rows(dbp, keep, n)
            d7b, m7b, d6b, m6b, pool5b, argmaxb = d7, m7, d6, m6, pool5, argmax
        # dropout then ReLU, backwards: g * mask where relu(.) > 0.  d = relu(.) * mask is positive exactly where both are, and where the mask is zero
        # the product already is (for finite g): the ReLU test reads d (no third row set to gather)
        g7 = rt.add(self._linear_backward("cls_score", d7b, dcls), self._linear_backward("bbox_pred", d7b, dbb))
        g7 = rt.relu_bwd_(rt.mul(g7, m7b, out=g7), d7b)
        g6 = self._linear_backward("fc7", d6b, g7)
        g6 = rt.relu_bwd_(rt.mul(g6, m6b, out=g6), d6b)
        stage("head_bwd_small")
        gp = self._linear_backward("fc6", pool5b, g6)
        stage("fc6_bwd")
        for n_ in reversed(self.HEAD):                             # every head gradient is enqueued: a bucket closed by a head layer starts now
            self._grads_ready(n_)
        # ---- RoI pooling (arg-max scatter) and the trunk; feat = relu(conv5_3): mask before entering conv5_3's backward
        gfeat = rt.relu_bwd_(rt.roi_pool_bwd(gp.reshape(rows, C, 7, 7), argmaxb.reshape(rows, C, 7, 7), C, H, W), feat)
        stage("roi_pool_bwd")
        rt.mem.join_aux_stream("grad")                               # the re-packed input-gradient weights are ready
        (trunk_backward_split if self.conv_math == "split" else trunk_backward)(self, list(zip(self.layers, inputs)), gfeat)
        self._dgrad_packed = False
        stage("trunk_bwd")
        # rois: the (n, 4) proposals THIS step pooled -- a parity test must hand the oracle these, not the proposals of a second, inference-form forward: the
        # fused conv + ReLU + pool launches of the training forward (act 5) and of the inference forward (act 4) may run different decompositions (round 6:
        # pick_conv_config), i.e. conv5_3 agrees to the last bits but one, and near-tied proposals can then differ
        return dict(losses=losses, n_rois=n, keep_inds=keep, masks=(m6, m7), rois=rois, head_acts=(a6, a7), layer_inputs=list(inputs) + [feat], roi_argmax=argmax)       # head_acts: relu(fc6), relu(fc7) before dropout (the parity tests read the device's ReLU decisions off them)
        # layer_inputs / roi_argmax: every discrete decision of this step's forward pass (a fused pool's entry is its _PoolArg: the winning cell and the ReLU bit of
        # every window; an unfused layer's successor input is its post-ReLU map; the arg-max cell of every RoI bin) -- the float64 arbiter of the parity tests imposes them

    def update(self):
        self._ensure_adopted()
        self.rt.sgd_momentum_wd(self.W, self.G, self.V, self.lr, self.momentum, self.weight_decay)
        if hasattr(self.model, "mark_params_updated"):
            self.model.mark_params_updated(self)
        self.iteration += 1

    def step(self, x, img_info, gt_boxes, masks=None):
        out = self.forward_backward(x, img_info, gt_boxes, masks)
        self.all_reduce()
        self.update()
        return out

    def losses_host(self, out):
        l = self.rt.mem.to_numpy(out["losses"])
        return dict(loss_cls=float(l[0]), loss_bbox=float(l[1]), cls_accuracy=float(l[2]), loss_rcnn=float(l[0] + l[1]))

    def flat_to_chainer_layout(self, flat):
        rt, out = self.rt, {}
        host = rt.mem.to_numpy(flat)
        def seg(key):
            sg = self.seg[key]
            return host[sg.offset:sg.offset + sg.size].reshape(sg.shape)
        for name, link in self.convs:
            out["trunk/" + name + "/W"] = np.ascontiguousarray(seg(name + "/W").T).reshape(link.cout, link.cin, 3, 3)
            out["trunk/" + name + "/b"] = seg(name + "/b").copy()
        for n in self.HEAD:
            out[n + "/W"], out[n + "/b"] = seg(n + "/W").copy(), seg(n + "/b").copy()
        return out

    def chainer_layout_to_flat(self, arrays, flat):
        rt = self.rt
        host = rt.mem.to_numpy(flat).copy()
        def put(key, val):
            sg = self.seg[key]
            host[sg.offset:sg.offset + sg.size] = np.asarray(val, dtype=np.float32).reshape(-1)
        for name, link in self.convs:
            if "trunk/" + name + "/W" in arrays:
                put(name + "/W", np.asarray(arrays["trunk/" + name + "/W"], dtype=np.float32).reshape(link.cout, -1).T)
            if "trunk/" + name + "/b" in arrays:
                put(name + "/b", arrays["trunk/" + name + "/b"])
        for n in self.HEAD:
            for sfx in ("/W", "/b"):
                if n + sfx in arrays:
                    put(n + sfx, arrays[n + sfx])
        flat[...] = rt.mem.from_numpy(host)

    def grads_chainer_layout(self):
        return self.flat_to_chainer_layout(self.G)

    def sync_params(self):
        for name, link in self.convs:
            link.W = self.rt.transpose(link.Wp).reshape(link.cout, link.cin, 3, 3)


class TorchComm(object):
    """torch.distributed as the collective layer: backend "nccl" is RCCL on ROCm (xGMI), "gloo" on CPU.

    gloo with DEVICE tensors (ranks sharing one GPU: the functional smoke of the N > 1 path on a 1-GPU box) is staged by hand
    through pinned host memory: the device -> host copy of a bucket rides on a copy stream behind the kernels that fill it, the
    reduction itself runs on host tensors when the trainer waits, the sum goes back with one async copy.  (Handing gloo the device
    tensor makes every call synchronise the whole device and poll: 2 s per step for three buckets, profiles/r02_bench_2rank_gloo_train.json.)
    FRCNN_COMM_TRACE=1 collects host milliseconds per call kind in `self.trace`."""

    def __init__(self, force_single_rank=False):
        import os
        import time
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self._time = torch, dist, time
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        # one rank normally skips the exchange; force_single_rank runs it anyway (bench.py --dist-world1, the RCCL smoke test)
        self.active = dist.is_initialized() and (self.world_size > 1 or force_single_rank)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.backend = dist.get_backend() if dist.is_initialized() else None
        self._pinned, self._copy_stream = None, None
        self.trace = {} if _tuning.get("FRCNN_COMM_TRACE") == "1" else None

    def _note(self, kind, t0):
        if self.trace is not None:
            e = self.trace.setdefault(kind, [0, 0.0])
            e[0] += 1
            e[1] += (self._time.perf_counter() - t0) * 1e3

    def _staged(self, t):
        return self.backend == "gloo" and t.is_cuda

    def _stage_out(self, t):
        """Device slice -> pinned host slice on the copy stream, ordered behind everything already enqueued on the current stream."""
        torch = self.torch
        n = t.numel()
        if self._pinned is None:
            self._pinned = {}
        key = (t.data_ptr(), n)                                    # a trainer's buckets are fixed slices of its gradient buffer
        host = self._pinned.get(key)
        if host is None:
            host = self._pinned[key] = torch.empty(n, dtype=t.dtype).pin_memory()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=t.device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(t.device))
        done = torch.cuda.Event()
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            host.copy_(t.reshape(-1), non_blocking=True)
            done.record(self._copy_stream)
        return ("staged", t, host, done)

    def all_reduce_sum(self, buf):
        t0 = self._time.perf_counter()
        t = buf if isinstance(buf, self.torch.Tensor) else self.torch.from_numpy(buf)      # NumPy buffers are reduced in place
        if self._staged(t):
            self.wait(self._stage_out(t))
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self._note("all_reduce_sum", t0)

    def all_reduce_sum_async(self, buf):
        """Launch the all-reduce of a (contiguous) slice now and return a handle: with RCCL it is enqueued behind the kernels
        already on the current stream and runs on the collective's stream, under whatever the caller launches next."""
        t0 = self._time.perf_counter()
        t = buf if isinstance(buf, self.torch.Tensor) else self.torch.from_numpy(buf)
        w = self._stage_out(t) if self._staged(t) else self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, async_op=True)
        self._note("launch", t0)
        return w

    def wait(self, work):
        t0 = self._time.perf_counter()
        if isinstance(work, tuple) and work[0] == "staged":
            _, t, host, done = work
            done.synchronize()                                     # the bucket is on the host
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM)  # gloo, host tensors: blocks this thread only
            t.reshape(-1).copy_(host, non_blocking=True)           # back on the current stream: later kernels see the sum
        else:
            work.wait()                                # RCCL: the current stream waits for the collective (no host block); gloo: blocks
        self._note("wait", t0)
