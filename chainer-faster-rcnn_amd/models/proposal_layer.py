"""ProposalLayer with the reference's interface (models/proposal_layer.py:31-221), running as one fused
device pipeline (csrc/detect.hip via frcnn_proposals): anchors -> decode -> clip -> min-size filter ->
descending top-K sort -> greedy NMS -> top-N, with no host round trip.

Same class constants, same `train` switch (12000/2000 vs 6000/300), same call signature
`layer(rpn_cls_prob, rpn_bbox_pred, img_info) -> (proposals (n,4), fg_probs (n,1))` and the same
type-check assertions (gated by CHAINER_TYPE_CHECK, proposal_layer.py:58,85-100).
"""
import os

import numpy as np

from ..chainer_compat import is_variable, kind, unwrap
from ..runtime import default_runtime
from .generate_anchors import generate_anchors


class ProposalLayer(object):
    RPN_NMS_THRESH = 0.7
    TRAIN_RPN_PRE_NMS_TOP_N = 12000
    TRAIN_RPN_POST_NMS_TOP_N = 2000
    TEST_RPN_PRE_NMS_TOP_N = 6000
    TEST_RPN_POST_NMS_TOP_N = 300
    RPN_MIN_SIZE = 16

    type_check_enable = int(os.environ.get('CHAINER_TYPE_CHECK', '1')) != 0

    def __init__(self, feat_stride=16, anchor_ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32), runtime=None):
        self._feat_stride = feat_stride
        self._anchors = generate_anchors(ratios=anchor_ratios, scales=anchor_scales)
        self._num_anchors = len(self._anchors)
        self._nms_thresh = float(self.RPN_NMS_THRESH)
        self._min_size = self.RPN_MIN_SIZE
        self._rt = runtime
        self._train = True
        self.train = self._train          # constructed in train mode, as the reference is (:68-69)

    @property
    def rt(self):
        if self._rt is None:
            self._rt = default_runtime()
        return self._rt

    @property
    def train(self):
        return self._train

    @train.setter
    def train(self, value):
        self._train = value
        if value:
            self._pre_nms_top_n = self.TRAIN_RPN_PRE_NMS_TOP_N
            self._post_nms_top_n = self.TRAIN_RPN_POST_NMS_TOP_N
        else:
            self._pre_nms_top_n = self.TEST_RPN_PRE_NMS_TOP_N
            self._post_nms_top_n = self.TEST_RPN_POST_NMS_TOP_N

    def _check_data_type_forward(self, rpn_cls_prob, rpn_bbox_pred, img_info):
        assert rpn_cls_prob.shape[0] == 1
        assert rpn_cls_prob.shape[1] == 2 * self._num_anchors
        assert len(rpn_cls_prob.shape) == 4
        assert kind(rpn_cls_prob) == 'f'
        assert is_variable(rpn_cls_prob)          # a Variable (anything carrying .data)
        assert rpn_bbox_pred.shape[0] == 1
        assert rpn_bbox_pred.shape[1] == 4 * self._num_anchors
        assert len(rpn_bbox_pred.shape) == 4
        assert kind(rpn_bbox_pred) == 'f'
        assert is_variable(rpn_bbox_pred)
        assert tuple(img_info.shape) == (1, 2)
        assert kind(img_info) in 'iu'
        assert is_variable(img_info)

    def _img_hw(self, img_info):
        info = unwrap(img_info)
        if self.rt.mem.is_array(info):
            info = self.rt.mem.to_numpy(info)
        info = np.asarray(info)
        return int(info[0][0]), int(info[0][1])

    def forward_device(self, rpn_cls_prob, rpn_bbox_pred, im_h, im_w, want_index=False):
        """Fixed-capacity, sync-free form: device (cap,4) rois, (cap,) probs, (1,) n_out [, (cap,) index]."""
        rt = self.rt
        prob = rt.asarray(unwrap(rpn_cls_prob), "f32")
        pred = rt.asarray(unwrap(rpn_bbox_pred), "f32")
        return rt.proposals(prob[0], pred[0], self._anchors, self._feat_stride, im_h, im_w, float(self._min_size),
                            self._pre_nms_top_n, self._post_nms_top_n, self._nms_thresh, want_index=want_index)

    def __call__(self, rpn_cls_prob, rpn_bbox_pred, img_info):
        if self.type_check_enable:
            self._check_data_type_forward(rpn_cls_prob, rpn_bbox_pred, img_info)
        im_h, im_w = self._img_hw(img_info)           # img_info.data[0] = (H, W) as passed (forward.py:93 passes (H,H))
        rois, probs, n_out = self.forward_device(rpn_cls_prob, rpn_bbox_pred, im_h, im_w)
        n = int(self.rt.mem.to_numpy(n_out)[0])       # the only sync: the reference returns exact-length arrays
        return rois[:n], probs[:n].reshape(n, 1)

    # ---- helpers other reference code reads (tests/test_anchor_target_layer.py:37-39) ----
    def _generate_all_bbox(self, feat_h, feat_w):
        """(feat_h*feat_w*A, 4) float64, order (h, w, a) with a fastest (proposal_layer.py:207-221)."""
        s = self._feat_stride
        ys, xs = np.meshgrid(np.arange(feat_h) * s, np.arange(feat_w) * s, indexing="ij")
        shifts = np.stack([xs, ys, xs, ys], axis=-1).reshape(-1, 1, 4)
        return (shifts + self._anchors[None]).reshape(-1, 4)

    def _generate_all_bbox_use_array_info(self, rpn_bbox_pred):
        _, feat_h, feat_w = rpn_bbox_pred.shape
        return self._generate_all_bbox(int(feat_h), int(feat_w)).astype(np.float32)
