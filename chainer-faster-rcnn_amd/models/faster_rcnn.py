"""FasterRCNN assembly with the reference's interface (models/faster_rcnn.py:19-178): trunk -> RPN ->
RoI pooling -> fc6/fc7 -> cls_score / bbox_pred -> decode + clip + softmax, every stage a HIP kernel of
libfrcnn_hip.so.  `model(x, img_info)` returns (softmax(cls_score) (R,21), pred_boxes (R,84)) and leaves
`rpn_proposals` / `rpn_probs` on the object, as the reference does (:119-120,175-178).

`forward_device()` is the sync-free form the benchmark drives: fixed capacity R = post_nms_top_n rows
(rows past n_out are zero boxes), so nothing between the image upload and the final read-back waits on
the host.
"""
import os

import numpy as np

from ..chainer_compat import Variable, is_variable, kind, unwrap
from .. import tuning
from ..runtime import default_runtime
from .region_proposal_network import RegionProposalNetwork
from .vgg16 import VGG16Prev


class Linear(object):
    """L.Linear: W (out, in), b (out)."""

    def __init__(self, rt, dtype="f32"):
        self.rt, self.dtype = rt, dtype
        self.W = self.b = self.Wb = None

    def set(self, W, b):
        W = self.rt.asarray(np.ascontiguousarray(W, dtype=np.float32) if isinstance(W, np.ndarray) else W, "f32")
        b = self.rt.asarray(np.ascontiguousarray(b, dtype=np.float32) if isinstance(b, np.ndarray) else b, "f32")
        if getattr(self, "_adopted", False) and self.W is not None and tuple(self.W.shape) == tuple(W.shape):
            self.W[...] = W                                # windows of a trainer's flat buffer: write through them (see Conv3x3.set)
            self.b[...] = b
        else:
            self.W, self.b = W, b
        self.refresh_bf16()

    def refresh_bf16(self):
        if self.dtype == "bf16":
            self.Wb = self.rt.to_bf16(self.W)              # raw bf16 bits, (out, in): K-contiguous for the MFMA B operand
            # the weight-stream layout of csrc/linear_bf16.hip (8 KB tiles = the kernel's LDS image), once per load; FRCNN_LINEAR_BF16=dma keeps the
            # round-5 kernel on the row-major bits (A/B)
            K = int(np.prod(self.W.shape[1:]))
            # (a layer with fewer than 256 outputs -- the stacked cls_score || bbox_pred GEMM -- is ONE column block of the stream kernel: 12 workgroups; the
            # round-5 kernel's 32 are faster there: 11.8 vs 13.9 us, profiles/r06_linear_bf16_micro.txt)
            N = int(self.W.shape[0])
            self.Wt = self.rt.linear_bf16_tile_w(self.Wb) if (K % 32 == 0 and N >= 256 and tuning.get("FRCNN_LINEAR_BF16") != "dma") else None
        elif self.dtype == "f32s":
            self.Ws = self.rt.f32s_split(self.W)           # the three bf16 terms of every fp32 weight, (3, out, in)

    def __call__(self, x, relu=False):
        return self.rt.linear(x, self.W, self.b, relu=relu)

    def bf16(self, x_bits, relu=False, out_bf16=False):
        if getattr(self, "Wt", None) is not None:
            return self.rt.linear_bf16_tiled(x_bits, self.Wt, int(self.W.shape[0]), self.b, relu=relu, out_bf16=out_bf16)
        return self.rt.linear_bf16(x_bits, self.Wb, self.b, relu=relu, out_bf16=out_bf16)

    def f32s(self, x_parts, relu=False, out_split=False):
        """The fp32 layer as six bf16 MFMA products of the 3-way split operands (csrc/conv_f32s.hip); x (3,M,K) split tensor."""
        return self.rt.linear_f32s(x_parts, self.Ws, self.b, relu=relu, out_split=out_split)


class FasterRCNN(object):
    type_check_enable = int(os.environ.get('CHAINER_TYPE_CHECK', '1')) != 0

    def __init__(self, trunk_class=VGG16Prev, rpn_in_ch=512, rpn_mid_ch=512, feat_stride=16, anchor_ratios=(0.5, 1, 2),
                 anchor_scales=(8, 16, 32), num_classes=21, loss_lambda=1, rpn_delta=3, rcnn_delta=1, runtime=None,
                 conv_dtype="f32", head_dtype="f32"):
        """conv_dtype: "f32" = BASELINE config 2 (fp32 everywhere); "bf16" = config 3 (trunk + RPN convolutions in bf16 on
        v_mfma_f32_32x32x16_bf16; proposals and RoI pooling in fp32).  head_dtype: the four L.Linear layers of the RCNN
        head, "f32" or "bf16" (bf16 operands, fp32 accumulation; box decoding and the class softmax stay fp32).  "f32s" (either): the
        fp32 layers computed as six bf16 MFMA products of 3-way split fp32 operands, fp32 accumulation (csrc/conv_f32s.hip) -- fp32
        tensors and fp32-class results on the bf16 matrix cores."""
        self.rt = runtime or default_runtime()
        # "f16" (either): the 16-bit chain's fp16 instantiation (north_star: "fp16/bf16 accumulate fp32"; csrc/conv_f16.hip ...) -- the same layouts, kernels and
        # schedule as "bf16" with 10-bit-mantissa operands: the model runs its bf16 code path on a runtime whose *_bf16 methods resolve to the *_f16* entry points.
        self.half = "f16" if "f16" in (conv_dtype, head_dtype) else "bf16"
        if self.half == "f16":
            if "bf16" in (conv_dtype, head_dtype):
                raise ValueError("conv_dtype / head_dtype: one 16-bit format per model (bf16 or f16)")
            self.rt = self.rt.with_half("f16")
            conv_dtype = "bf16" if conv_dtype == "f16" else conv_dtype
            head_dtype = "bf16" if head_dtype == "f16" else head_dtype
        self.conv_dtype, self.head_dtype = conv_dtype, head_dtype
        self.trunk = trunk_class(runtime=self.rt, conv_dtype=conv_dtype) if conv_dtype != "f32" else trunk_class(runtime=self.rt)
        self.RPN = RegionProposalNetwork(rpn_in_ch, rpn_mid_ch, feat_stride, anchor_ratios, anchor_scales, num_classes,
                                         loss_lambda, rpn_delta, runtime=self.rt, conv_dtype=conv_dtype)
        self.fc6, self.fc7 = Linear(self.rt, head_dtype), Linear(self.rt, head_dtype)
        self.cls_score, self.bbox_pred = Linear(self.rt, head_dtype), Linear(self.rt, head_dtype)
        self._feat_stride = feat_stride
        self._num_classes = num_classes
        self._rcnn_delta = rcnn_delta
        self.RPN.train = False                               # faster_rcnn.py:42
        self._rcnn_train = False
        self._spatial_scale = 1. / feat_stride
        self.rpn_proposals = self.rpn_probs = None

    # mode switches, faster_rcnn.py:48-74
    @property
    def rcnn_train(self):
        return self._rcnn_train

    @rcnn_train.setter
    def rcnn_train(self, val):
        self._rcnn_train = val
        if val:
            self.RPN.train = not val
        self.trunk.train = bool(self.rcnn_train or self.rpn_train)

    @property
    def rpn_train(self):
        return self.RPN.train

    @rpn_train.setter
    def rpn_train(self, val):
        self.RPN.train = val
        if val:
            self._rcnn_train = not val
        self.trunk.train = bool(self.rcnn_train or self.rpn_train)

    def load_params(self, params):
        """params: dict keyed by Chainer link path ('trunk/conv1_1/W', 'RPN/rpn_conv_3x3/b', 'fc6/W', ...) --
        the key scheme of the reference's .npz snapshots (forward.py:29)."""
        self.trunk.load_params(params, "trunk/")
        self.RPN.load_params(params, "RPN/")
        for name in ("fc6", "fc7", "cls_score", "bbox_pred"):
            getattr(self, name).set(params[name + "/W"], params[name + "/b"])
        self._stack_head()

    def _stack_head(self):
        """Inference runs cls_score and bbox_pred (faster_rcnn.py:35-36) as ONE L.Linear: weight rows [cls_score.W; zero rows up to
        the next multiple of 32; bbox_pred.W], so a row of the output is [scores | pad | deltas] and one GEMM launch replaces two
        (every output column is still its own dot product over fc7).  The trainers keep using the two separate layers."""
        rt, ncls = self.rt, self._num_classes
        Wc, bc = rt.mem.to_numpy(self.cls_score.W), rt.mem.to_numpy(self.cls_score.b)
        Wb, bb = rt.mem.to_numpy(self.bbox_pred.W), rt.mem.to_numpy(self.bbox_pred.b)
        self._head_dcol = (ncls + 31) // 32 * 32
        n = self._head_dcol + 4 * ncls
        W = np.zeros((n, Wc.shape[1]), np.float32)
        b = np.zeros((n,), np.float32)
        W[:ncls], b[:ncls] = Wc, bc
        W[self._head_dcol:], b[self._head_dcol:] = Wb, bb
        self.head_out = Linear(rt, self.head_dtype)
        self.head_out.set(W, b)
        self._head_dirty = False

    def mark_params_updated(self, trainer=None):
        """The trainers call this after an optimizer update (and serializers.load_npz after a load): everything DERIVED from the
        parameters -- the stacked inference head, the bf16 copies of the convolution / RPN-head / FC weights -- is rebuilt on the
        next inference.  `trainer`: the trainer holding the live packed weights (its sync_params() writes them back to Chainer's
        layout first)."""
        self._head_dirty = True
        self._derived_dirty = True
        if trainer is not None:
            # EVERY trainer that has updated this model holds live packed weights for its own parameter set (after an rpn -> rcnn
            # alternation the RPNTrainer still owns rpn_conv_3x3 and the RPN heads, the RCNNTrainer the trunk and the FC head):
            # remember them all, most recent last, so that later syncs win where two trainers share the trunk
            # A NEW trainer of a class that is already registered supersedes the old one (a script that builds a trainer per stage or
            # epoch kept every one's W / G / V arenas alive, ~1.6 GB each for VGG-16, and re-synced them all on every snapshot): it has adopted
            # every parameter of that set into its own arena (_ParamArena._adopt copies the links' current windows), so the old trainer's packed
            # weights are stale copies from then on.  Not a weak reference: a trainer the caller dropped WITHOUT a successor still holds the only
            # packed weights its links point into until they are synced -- detach_trainer() is the explicit way out.
            trs = [t for t in getattr(self, "_trainers", []) if t is not trainer and type(t) is not type(trainer)]
            trs.append(trainer)
            self._trainers = trs
            self._last_trainer = trainer

    def detach_trainer(self, trainer):
        """Write `trainer`'s packed weights back to the links' Chainer-layout arrays and forget it.  Only its gradient and velocity arenas go with it: the links'
        packed weights / biases stay VIEWS into the trainer's parameter arena (≈ 550 MB for the RCNN trainer), which therefore lives until another trainer adopts
        the links or the model is dropped."""
        trainer.sync_params()
        self._trainers = [t for t in getattr(self, "_trainers", []) if t is not trainer]
        if getattr(self, "_last_trainer", None) is trainer:
            self._last_trainer = self._trainers[-1] if self._trainers else None
        self._head_dirty = self._derived_dirty = True

    def sync_trainers(self):
        """Packed training weights of every trainer that has updated the model -> (co,ci,3,3) / (out,in) arrays on the links,
        oldest first (the most recent trainer's view of shared parameters wins: they share one arena, so it is the same view)."""
        for tr in getattr(self, "_trainers", []):
            tr.sync_params()

    def _refresh_derived(self):
        self.sync_trainers()                                 # packed training weights -> (co,ci,3,3) / (out,in) arrays on the links
        if self.conv_dtype in ("bf16", "f32s"):
            for link in getattr(self.trunk, "links", {}).values():
                link.refresh_bf16()
            self.RPN.rpn_conv_3x3.refresh_bf16()
            self.RPN.refresh_heads_bf16()
        for name in ("fc6", "fc7", "cls_score", "bbox_pred"):
            getattr(self, name).refresh_bf16()
        self._derived_dirty = False

    def _check_data_type_forward(self, x, img_info, gt_boxes):
        assert x.shape[0] == 1
        assert kind(x) == 'f'
        assert is_variable(x)
        assert tuple(img_info.shape) == (1, 2)
        assert kind(img_info) in 'iu'
        assert is_variable(img_info)
        if gt_boxes is not None:
            assert gt_boxes.shape[0] == 1 and gt_boxes.shape[1] > 0 and gt_boxes.shape[2] == 5
            assert kind(gt_boxes) == 'f' and is_variable(gt_boxes)

    def forward_device(self, x, im_h, im_w, keep=False, timer=None, collect=None):
        """Sync-free inference.  Returns dict(cls_prob (R,ncls), pred_boxes (R,4*ncls), rois (R,4), probs (R,),
        n_out (1,) int32) -- all device arrays, R = post_nms_top_n; rows >= n_out are padding.
        `timer.mark(name)` (optional) is called after every stage: bench.py records a HIP event there."""
        rt = self.rt
        if getattr(self, "_derived_dirty", False):
            self._refresh_derived()
        if getattr(self, "_head_dirty", True):
            self._stack_head()
        mark = timer.mark if timer else (lambda name: None)
        # bf16 chain: RoI pooling reads the channel-blocked bf16 map itself (a cell's eight channels are one 16-byte load), so the fp32
        # NCHW copy of conv5_3 is only made when somebody asks for the intermediate maps
        blk_pool = self.conv_dtype == "bf16" and not keep and collect is None and hasattr(self.trunk, "_call_bf16")
        if blk_pool:
            self.trunk.skip_nchw = True
        try:
            feat = self.trunk(x, timer=timer, collect=collect) if collect is not None else self.trunk(x, timer=timer)
        finally:
            if blk_pool:
                self.trunk.skip_nchw = False
        x_bf16 = getattr(self.trunk, "feat_bf16", None) if self.conv_dtype == "bf16" else None
        if feat is None:
            C, H, W = self.trunk.feat_shape[1:]
            if H > 76 or W > 64:                             # beyond the cell-major kernel's LDS image: pool from an fp32 NCHW copy
                feat = rt.bf16_to_nchw(x_bf16, C)
        else:
            C, H, W = [int(v) for v in feat.shape[1:]]
        x_split = getattr(self.trunk, "feat_split", None) if self.conv_dtype == "f32s" else None
        rpn_h, score, prob, bbox = self.RPN.heads(feat, want_score=False, timer=timer, x_bf16=x_bf16, x_split=x_split)
        if keep:
            rois, probs, n_out, src_index = self.RPN.proposal_layer.forward_device(prob, bbox, im_h, im_w, want_index=True)
        else:
            rois, probs, n_out = self.RPN.proposal_layer.forward_device(prob, bbox, im_h, im_w)
        mark("proposals")
        pool5_split = None
        if self.head_dtype == "f32s" and not keep and H <= 76 and W <= 64 and feat is not None:
            pool5 = pool5_split = rt.roi_pool_fwd_chw_f32s(feat, rois, 7, 7, self._spatial_scale)   # fp32 maxima, stored as their three bf16 terms
            pool5_bits = None
        elif self.head_dtype == "bf16" and not keep:
            if feat is None:
                pool5 = rt.roi_pool_fwd_blk_bf16(x_bf16, C, rois, 7, 7, self._spatial_scale, out_bf16=True)
            else:
                pool5 = rt.roi_pool_fwd_chw_bf16(feat, rois, 7, 7, self._spatial_scale)   # pooled in fp32, stored as bf16 bits
            pool5_bits = pool5
        else:
            if feat is None:
                pool5 = rt.roi_pool_fwd_blk_bf16(x_bf16, C, rois, 7, 7, self._spatial_scale, out_bf16=False)
            else:
                pool5 = rt.roi_pool_fwd_chw(feat, rois, 7, 7, self._spatial_scale)    # rois (R,4): concat (:123-124) folded in
            pool5_bits = None
        mark("roi_pool")
        if self.head_dtype == "f32s":
            if pool5_split is None:
                pool5_split = rt.f32s_split(pool5.reshape(int(pool5.shape[0]), -1))
            fc6 = self.fc6.f32s(pool5_split, relu=True, out_split=True)
            mark("fc6")
            fc7 = self.fc7.f32s(fc6, relu=True, out_split=True)
            mark("fc7")
            head = self.head_out.f32s(fc7)
            if keep:
                fc6, fc7 = rt.f32s_join(fc6), rt.f32s_join(fc7)
        elif self.head_dtype == "bf16":
            if pool5_bits is None:
                pool5_bits = rt.to_bf16(pool5.reshape(int(pool5.shape[0]), -1))
            fc6 = self.fc6.bf16(pool5_bits, relu=True, out_bf16=True)
            mark("fc6")
            fc7 = self.fc7.bf16(fc6, relu=True, out_bf16=True)
            mark("fc7")
            head = self.head_out.bf16(fc7)
        else:
            fc6 = self.fc6(pool5, relu=True)        # dropout is the identity in inference (faster_rcnn.py:127-128)
            mark("fc6")
            fc7 = self.fc7(fc6, relu=True)
            mark("fc7")
            head = self.head_out(fc7)
        pred_boxes, cls_prob = rt.head_decode_stacked(rois, head, self._num_classes, self._head_dcol, im_h, im_w)
        mark("head_out")
        out = dict(cls_prob=cls_prob, pred_boxes=pred_boxes, rois=rois, probs=probs, n_out=n_out)
        if keep:
            ncls, d0 = self._num_classes, self._head_dcol
            out.update(feat=feat, rpn_h=rpn_h, rpn_cls_prob=prob, rpn_bbox_pred=bbox, src_index=src_index, pool5=pool5, fc6=fc6, fc7=fc7,
                       cls_score=rt.mem.from_numpy(np.ascontiguousarray(rt.mem.to_numpy(head)[:, :ncls])),
                       bbox_pred=rt.mem.from_numpy(np.ascontiguousarray(rt.mem.to_numpy(head)[:, d0:d0 + 4 * ncls])))
        return out

    def __call__(self, x, img_info, gt_boxes=None):
        if self.type_check_enable:
            self._check_data_type_forward(x, img_info, gt_boxes)
        if self.rpn_train and gt_boxes is not None:                  # faster_rcnn.py:115-116: RPN training mode returns rpn_loss
            return self.RPN(Variable(self.trunk(x)), img_info, gt_boxes)
        if self.rcnn_train and gt_boxes is not None:                 # faster_rcnn.py:136-166: returns loss_rcnn
            from ..train import RCNNTrainer                          # the forward half of the stage-2 step (dropout, ProposalTargetLayer, losses)
            if getattr(self, "_rcnn_stepper", None) is None:
                self._rcnn_stepper = RCNNTrainer(self)
            out = self._rcnn_stepper.forward_backward(x, img_info, gt_boxes)
            l = self._rcnn_stepper.losses_host(out)
            self.loss_cls, self.loss_bbox, self.cls_accuracy = l["loss_cls"], l["loss_bbox"], l["cls_accuracy"]
            return Variable(np.float32(l["loss_rcnn"]), name='loss_rcnn')
        im_h, im_w = self.RPN.proposal_layer._img_hw(img_info)
        out = self.forward_device(x, im_h, im_w)
        n = int(self.rt.mem.to_numpy(out["n_out"])[0])
        self.rpn_proposals = out["rois"][:n]
        self.rpn_probs = out["probs"][:n].reshape(n, 1)
        return Variable(out["cls_prob"][:n]), out["pred_boxes"][:n]
