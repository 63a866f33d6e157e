"""ProposalTargetLayer with the reference's interface (models/proposal_target_layer.py:20-150): an AnchorTargetLayer subclass
whose `__call__(proposals, gt_boxes)` returns `(use_gt_boxes (k,5), ext_bbox_reg_targets (k, 4*num_classes), keep_inds (k,))`,
k <= ROIS_PER_IMAGE = 128.

The float64 IoU matrix runs on the device (frcnn_bbox_overlaps_f64).  The fg/bg sampling is host code, as it is in the
reference (which copies the indices to the CPU for `np.random.choice`): it draws from NumPy's global RNG with the reference's
exact call sequence.  Like the reference, the class labels that reach the loss are `use_gt_boxes[:, -1]`, i.e. the matched
gt's label even for the background samples (the zeroing at :130 only touches a local copy).
"""
import os

import numpy as np

from ..chainer_compat import is_variable, kind, unwrap
from .anchor_target_layer import AnchorTargetLayer


class ProposalTargetLayer(AnchorTargetLayer):
    FG_THRESH = 0.5
    BG_THRESH_HI = 0.5
    BG_THRESH_LO = 0.1
    ROIS_PER_IMAGE = 128
    FG_FRACTION = 0.25

    type_check_enable = int(os.environ.get('CHAINER_TYPE_CHECK', '1')) != 0

    def __init__(self, feat_stride=16, anchor_ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32), num_classes=21, runtime=None):
        super(ProposalTargetLayer, self).__init__(feat_stride, anchor_ratios, anchor_scales, runtime=runtime)
        self._num_classes = num_classes
        self._n_fg_rois = int(self.FG_FRACTION * self.ROIS_PER_IMAGE)

    def _check_data_type_forward(self, proposals, gt_boxes):
        assert len(proposals) > 0
        assert len(proposals.shape) == 2 and proposals.shape[1] == 4
        assert kind(proposals) == 'f'
        assert len(gt_boxes.shape) == 3 and gt_boxes.shape[0] == 1 and gt_boxes.shape[2] == 5
        assert kind(gt_boxes) == 'f'
        assert is_variable(gt_boxes)

    def overlaps_device(self, proposals_dev, gt):
        """The float64 IoU matrix of DEVICE proposals (n,4) f32 against gt -- host (G,5), or (G,4) float64 already on the device -- (:88-91), enqueued on the current stream and left on the
        device: RCNNTrainer issues it right behind the ProposalLayer and reads it back together with the RoI count, so the sampling below runs on
        the host while the head's forward pass runs on the GPU."""
        rt = self.rt
        if not rt.mem.is_array(gt):                                  # (a caller that uploaded the (G,4) float64 boxes ahead of time passes them as they are)
            gt = rt.asarray(np.ascontiguousarray(gt[:, :4], dtype=np.float64), "f64")
        return rt.bbox_overlaps(rt.mem.astype(proposals_dev, "f64"), gt)

    def sample(self, proposals, gt, overlaps=None):
        """Host part (:84-148) on NumPy arrays: proposals (n,4) f32, gt (G,5) f32 -> (use_gt_boxes, ext_targets, keep_inds).
        overlaps: the (n,G) float64 IoU matrix when the caller has already fetched it (overlaps_device)."""
        rt = self.rt
        ov = overlaps if overlaps is not None else rt.mem.to_numpy(
            rt.bbox_overlaps(rt.asarray(np.ascontiguousarray(proposals, dtype=np.float64), "f64"),
                             rt.asarray(np.ascontiguousarray(gt[:, :4], dtype=np.float64), "f64")))
        argmax = ov.argmax(axis=1)
        max_ov = ov[np.arange(len(proposals)), argmax]
        fg_inds = np.where(max_ov >= self.FG_THRESH)[0]
        n_fg = min(self._n_fg_rois, fg_inds.size)
        if fg_inds.size > 0:
            fg_inds = np.random.choice(fg_inds, size=n_fg, replace=False)
        bg_inds = np.where((max_ov < self.BG_THRESH_HI) & (max_ov >= self.BG_THRESH_LO))[0]
        n_bg = min(self.ROIS_PER_IMAGE - n_fg, bg_inds.size)
        if bg_inds.size > 0:
            bg_inds = np.random.choice(bg_inds, size=n_bg, replace=False)
        keep = np.concatenate([fg_inds, bg_inds]).astype(np.int32)
        props = proposals[keep]
        use_gt = gt[argmax[keep]]
        # bbox_transform(proposals, use_gt_boxes) in float32 (bbox_transform.py:18-38)
        ew = props[:, 2] - props[:, 0] + 1.0; eh = props[:, 3] - props[:, 1] + 1.0
        ecx = props[:, 0] + 0.5 * ew; ecy = props[:, 1] + 0.5 * eh
        gw = use_gt[:, 2] - use_gt[:, 0] + 1.0; gh = use_gt[:, 3] - use_gt[:, 1] + 1.0
        gcx = use_gt[:, 0] + 0.5 * gw; gcy = use_gt[:, 1] + 0.5 * gh
        t = np.vstack(((gcx - ecx) / ew, (gcy - ecy) / eh, np.log(gw / ew), np.log(gh / eh))).transpose()
        ext = np.zeros((len(keep), 4 * self._num_classes), dtype=np.float32)
        ind = np.where(use_gt[:, 4] > 0)[0]                           # :136-142, the loop over the rows as one indexed assignment
        if ind.size:
            pos = (4 * use_gt[ind, -1]).astype(np.int64)
            ext[ind[:, None], pos[:, None] + np.arange(4)] = t[ind]
        return use_gt, ext, keep

    def __call__(self, proposals, gt_boxes):
        if self.type_check_enable:
            self._check_data_type_forward(proposals, gt_boxes)
        rt = self.rt
        props = rt.mem.to_numpy(proposals) if rt.mem.is_array(proposals) else np.asarray(proposals)
        gt = unwrap(gt_boxes)
        gt = rt.mem.to_numpy(gt) if rt.mem.is_array(gt) else np.asarray(gt)
        use_gt, ext, keep = self.sample(np.ascontiguousarray(props, dtype=np.float32), np.ascontiguousarray(gt[0], dtype=np.float32))
        return rt.mem.from_numpy(use_gt), rt.mem.from_numpy(ext), rt.mem.from_numpy(keep)
