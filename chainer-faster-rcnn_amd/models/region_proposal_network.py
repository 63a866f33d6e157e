"""RegionProposalNetwork with the reference's interface (models/region_proposal_network.py:16-204):
rpn_conv_3x3 (+ReLU) -> rpn_cls_score / 18-way softmax / rpn_bbox_pred -> ProposalLayer.
Inference path on device; children keep Chainer's names (rpn_conv_3x3, rpn_cls_score, rpn_bbox_pred)."""
import os

import numpy as np

from .. import tuning as _tuning
from ..chainer_compat import Variable, is_variable, kind, unwrap
from ..runtime import default_runtime
from .proposal_layer import ProposalLayer
from .vgg16 import Conv3x3


class RegionProposalNetwork(object):
    type_check_enable = int(os.environ.get('CHAINER_TYPE_CHECK', '1')) != 0

    def __init__(self, in_ch=512, mid_ch=512, feat_stride=16, anchor_ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32),
                 num_classes=21, loss_lambda=1., delta=3, runtime=None, conv_dtype="f32"):
        self.rt = runtime or default_runtime()
        self.conv_dtype = conv_dtype
        self.n_anchors = len(anchor_ratios) * len(anchor_scales)
        self.mid_ch = mid_ch
        self.rpn_conv_3x3 = Conv3x3(self.rt, in_ch, mid_ch, conv_dtype)
        self.rpn_cls_score = dict(W=None, b=None)       # (2A, mid, 1, 1)
        self.rpn_bbox_pred = dict(W=None, b=None)       # (4A, mid, 1, 1)
        self.proposal_layer = ProposalLayer(feat_stride, anchor_ratios, anchor_scales, runtime=self.rt)
        self._loss_lambda = loss_lambda
        self._delta = delta
        self.anchor_target_layer = None
        self._train = True
        self.train = True                               # region_proposal_network.py:64

    @property
    def train(self):
        return self._train

    @train.setter
    def train(self, val):
        self._train = val
        self.proposal_layer.train = val                 # :71-74

    def load_params(self, params, prefix="RPN/"):
        rt = self.rt
        self.rpn_conv_3x3.set(params[prefix + "rpn_conv_3x3/W"], params[prefix + "rpn_conv_3x3/b"])
        for name, store in (("rpn_cls_score", self.rpn_cls_score), ("rpn_bbox_pred", self.rpn_bbox_pred)):
            W = np.ascontiguousarray(params[prefix + name + "/W"], dtype=np.float32)
            store["W"] = rt.asarray(W.reshape(W.shape[0], -1), "f32")
            store["b"] = rt.asarray(np.ascontiguousarray(params[prefix + name + "/b"], dtype=np.float32), "f32")
        packed = rt.rpn_heads_pack(self.rpn_cls_score["W"], self.rpn_cls_score["b"], self.rpn_bbox_pred["W"], self.rpn_bbox_pred["b"])
        if getattr(self, "_heads_adopted", False):      # windows of a trainer's flat parameter buffer: write through them
            self._heads_packed[0][...] = packed[0]
            self._heads_packed[1][...] = packed[1]
        else:
            self._heads_packed = packed
        self.refresh_heads_bf16()

    def refresh_heads_bf16(self):
        """The two 1x1 heads as ONE bf16 1x1 convolution: cls (2A) rows then bbox (4A) rows -- re-derived from the fp32 head weights."""
        if self.conv_dtype != "bf16":
            return
        rt = self.rt
        W = np.concatenate([rt.mem.to_numpy(self.rpn_cls_score["W"]).reshape(-1, self.mid_ch),
                            rt.mem.to_numpy(self.rpn_bbox_pred["W"]).reshape(-1, self.mid_ch)], 0).astype(np.float32)
        b = np.concatenate([rt.mem.to_numpy(self.rpn_cls_score["b"]), rt.mem.to_numpy(self.rpn_bbox_pred["b"])], 0).astype(np.float32)
        self._heads_bf16 = (rt.bf16_pack_conv_w(rt.asarray(np.ascontiguousarray(W[:, :, None, None]), "f32"), 1), rt.asarray(b, "f32"))

    def _check_data_type_forward(self, x, img_info, gt_boxes):
        assert x.shape[0] == 1
        assert kind(x) == 'f'
        assert tuple(img_info.shape) == (1, 2)
        assert kind(img_info) in 'iu'
        assert is_variable(x) and is_variable(img_info)
        if gt_boxes is not None:
            assert gt_boxes.shape[0] == 1 and gt_boxes.shape[2] == 5 and kind(gt_boxes) == 'f'

    def heads(self, x, want_score=True, timer=None, x_bf16=None, x_split=None):
        """(h, rpn_cls_score, rpn_cls_prob, rpn_bbox_pred) -- region_proposal_network.py:117-120.  x_bf16: the same map as the
        channel-blocked bf16 array the bf16 trunk produced (skips re-converting the fp32 copy)."""
        if self.conv_dtype == "bf16":
            return self._heads_bf16_path(None if x_bf16 is not None else self.rt.asarray(unwrap(x), "f32"), timer, x_bf16)
        if self.conv_dtype == "f32s":           # the fp32 convolution on split tensors (csrc/conv_f32s.hip), fp32 NCHW out; heads as in fp32
            xs = x_split if x_split is not None else self.rt.f32s_from_nchw(self.rt.asarray(unwrap(x), "f32"))
            h = self.rpn_conv_3x3.f32s(xs, relu=True, out_f32_nchw=True)
        else:
            h = self.rpn_conv_3x3(self.rt.asarray(unwrap(x), "f32"), relu=True)
        if timer:
            timer.mark("rpn_conv_3x3")
        score, prob, bbox = self.rt.rpn_heads(h, self._heads_packed)
        if timer:
            timer.mark("rpn_heads")
        return h, score, prob, bbox

    def _heads_bf16_path(self, x, timer, x_bf16=None):
        """x = fp32 NCHW feature map whose values are bf16-representable (it came out of the bf16 trunk): back to channel-last
        bf16 (exact), rpn_conv_3x3 in bf16, both heads as one bf16 1x1 convolution writing fp32 NCHW, softmax in fp32."""
        rt, A = self.rt, self.n_anchors
        h = self.rpn_conv_3x3.bf16(x_bf16 if x_bf16 is not None else rt.bf16_from_nchw(x), relu=True)
        if timer:
            timer.mark("rpn_conv_3x3")
        wb, bb = self._heads_bf16
        if 6 * A <= 64 and _tuning.get("FRCNN_RPN_HEADS", "")[:1] != "c":
            score, prob, bbox = rt.rpn_heads_bf16(h, wb, bb, self.mid_ch, A)                      # one launch (csrc/conv_bf16.hip)
        else:
            raw = rt.conv_bf16(h, wb, bb, self.mid_ch, 6 * A, 1, relu=False, out_f32_nchw=True)   # (1, 6A, H, W) fp32
            score, bbox = raw[:, :2 * A], raw[:, 2 * A:]
            prob = rt.softmax_channels(score[0])
        if timer:
            timer.mark("rpn_heads")
        return h, score, prob, bbox

    def __call__(self, x, img_info, gt_boxes=None):
        if self.type_check_enable:
            self._check_data_type_forward(x, img_info, gt_boxes)
        _, score, prob, bbox = self.heads(x, want_score=True)
        if self.train and gt_boxes is not None:
            # region_proposal_network.py:127-158: anchor targets, the two losses, rpn_loss = cls + lambda * bbox.  (The
            # reference also runs ProposalLayer here and discards the result; the backward pass and the update live in
            # chainer_faster_rcnn_amd.train.RPNTrainer, which plays chainer's optimizer.update(lossfun).)
            if self.anchor_target_layer is None:
                from .anchor_target_layer import AnchorTargetLayer
                self.anchor_target_layer = AnchorTargetLayer(self.proposal_layer._feat_stride, runtime=self.rt)
                self.anchor_target_layer._anchors = self.proposal_layer._anchors
                self.anchor_target_layer._num_anchors = self.proposal_layer._num_anchors
            feat_h, feat_w = int(prob.shape[2]), int(prob.shape[3])
            im_h, im_w = self.proposal_layer._img_hw(img_info)
            labels, targets, inds, n_in, _ = self.anchor_target_layer.forward_device(feat_h, feat_w, gt_boxes, im_h, im_w)
            losses = self.rt.rpn_loss(score[0], bbox[0], labels, targets, inds, n_in, self.n_anchors, feat_h, feat_w, self._delta,
                                      self._loss_lambda, want_grad=False)
            l = self.rt.mem.to_numpy(losses)
            self.rpn_loss_cls, self.rpn_loss_bbox, self.rpn_cls_accuracy = float(l[0]), float(l[1]), float(l[2])
            return Variable(np.float32(l[0] + self._loss_lambda * l[1]), name='rpn_loss')
        return self.proposal_layer(Variable(prob), Variable(bbox), img_info)
