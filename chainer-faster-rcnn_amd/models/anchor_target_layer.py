"""AnchorTargetLayer with the reference's interface (models/anchor_target_layer.py:15-198): a ProposalLayer subclass
whose `__call__(feat_h, feat_w, gt_boxes, img_info)` returns `(bbox_labels, bbox_reg_targets, inds_inside, n_all_bbox)`.

keep_inside, the float64 IoU matrix (bbox.pyx), the arg-max bookkeeping, the label rules (including "negatives clobber
positives", :144-145) and bbox_transform run on the device (csrc/train.hip via frcnn_anchor_target).  The random fg/bg
subsample (:147-167) is host code, as it is in the reference (which copies the indices to the CPU for
`np.random.choice`): it draws from NumPy's global RNG with the reference's exact call sequence, so a seeded run
reproduces the reference's draws.
"""
import os

import numpy as np

from ..chainer_compat import is_variable, kind, unwrap
from .proposal_layer import ProposalLayer


class AnchorTargetLayer(ProposalLayer):
    RPN_NEGATIVE_OVERLAP = 0.3
    RPN_POSITIVE_OVERLAP = 0.7
    RPN_FG_FRACTION = 0.5
    RPN_BATCHSIZE = 256

    type_check_enable = int(os.environ.get('CHAINER_TYPE_CHECK', '1')) != 0

    def __init__(self, feat_stride=16, anchor_ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32), runtime=None):
        super(AnchorTargetLayer, self).__init__(feat_stride, anchor_ratios, anchor_scales, runtime=runtime)

    def _check_data_type_forward(self, gt_boxes, img_info):
        assert gt_boxes.shape[0] == 1
        assert gt_boxes.shape[2] == 5
        assert kind(gt_boxes) == 'f'
        assert is_variable(gt_boxes)
        assert tuple(img_info.shape) == (1, 2)
        assert kind(img_info) in 'iu'
        assert is_variable(img_info)

    def subsample(self, labels):
        """anchor_target_layer.py:147-167 on a host int32 array, in place; same np.random.choice calls as the reference."""
        num_fg = int(self.RPN_FG_FRACTION * self.RPN_BATCHSIZE)
        fg_inds = np.where(labels == 1)[0]
        if len(fg_inds) > num_fg:
            disable_inds = np.random.choice(fg_inds, size=int(len(fg_inds) - num_fg), replace=False)
            labels[disable_inds] = -1
        num_bg = self.RPN_BATCHSIZE - np.sum(labels == 1)
        bg_inds = np.where(labels == 0)[0]
        if len(bg_inds) > num_bg:
            disable_inds = np.random.choice(bg_inds, size=int(len(bg_inds) - num_bg), replace=False)
            labels[disable_inds] = -1
        return labels

    def forward_device(self, feat_h, feat_w, gt_boxes, im_h, im_w):
        """-> (labels (n,) i32, targets (n,4) f32, inds_inside (n,) i32, n_inside int, n_all int); device arrays."""
        rt = self.rt
        gt = rt.asarray(unwrap(gt_boxes), "f32")
        gt = gt[0] if len(gt.shape) == 3 else gt
        # anchor_target_layer.py:188-190: `overlaps.argmax(axis=1)` / `argmax(axis=0)` on an image without a ground-truth box, or one too small for any anchor
        # to lie inside it, is NumPy's "attempt to get argmax of an empty sequence" -- the reference raises, so does this
        if int(gt.shape[0]) == 0:
            raise ValueError("attempt to get argmax of an empty sequence")
        inds, n_in, labels, targets, _ = rt.anchor_target(self._anchors, int(feat_h), int(feat_w), self._feat_stride, im_h, im_w, gt)
        n = int(rt.mem.to_numpy(n_in)[0])
        if n == 0:
            raise ValueError("attempt to get argmax of an empty sequence")
        host_labels = self.subsample(rt.mem.to_numpy(labels[:n]))
        labels = rt.mem.from_numpy(host_labels)
        return labels, targets[:n], inds[:n], n, self._num_anchors * int(feat_h) * int(feat_w)

    def __call__(self, feat_h, feat_w, gt_boxes, img_info):
        if self.type_check_enable:
            self._check_data_type_forward(gt_boxes, img_info)
        im_h, im_w = self._img_hw(img_info)
        labels, targets, inds, _, n_all = self.forward_device(feat_h, feat_w, gt_boxes, im_h, im_w)
        return labels, targets, inds, n_all
