"""Synthetic VOC-shaped inputs and random-init weights of the reference architecture:
there is no network for datasets or checkpoints, so the benchmark, smoke test and end-to-end tests use these.

Weights are keyed by Chainer link path, exactly the key scheme of the reference's .npz snapshots
(forward.py:29): 'trunk/conv1_1/W' (co,ci,3,3), 'RPN/rpn_conv_3x3/b', 'fc6/W' (out,in), ...
"""
import numpy as np

from .models.vgg16 import LAYERS

PIXEL_MEANS = np.array([102.9801, 115.9465, 122.7717], dtype=np.float64)     # forward.py:22


def image(seed=0, h=600, w=1000):
    """(1,3,h,w) float32: uniform(0,255) minus the BGR pixel means, as forward.py:34-45 produces."""
    rs = np.random.RandomState(seed)
    x = rs.uniform(0, 255, (1, 3, h, w)) - PIXEL_MEANS.reshape(1, 3, 1, 1)
    return x.astype(np.float32)


def params(seed=1, num_classes=21, n_anchors=9, rpn_ch=512, roi_feat=512 * 7 * 7):
    """Trunk: He-normal (conv1_1 additionally divided by 64 so activations are O(1) rather than O(pixel
    value) and the RPN's exp() stays finite); RPN and head: Normal(0, 0.01) as the reference initialises
    them (models/faster_rcnn.py:27, region_proposal_network.py:50); biases 0."""
    rs = np.random.RandomState(seed)
    p = {}
    for l in LAYERS:
        if l == "pool":
            continue
        name, ci, co = l
        w = rs.randn(co, ci, 3, 3) * np.sqrt(2.0 / (ci * 9))
        if name == "conv1_1":
            w = w / 64.0
        p["trunk/%s/W" % name] = w.astype(np.float32)
        p["trunk/%s/b" % name] = np.zeros(co, np.float32)
    p["RPN/rpn_conv_3x3/W"] = (rs.randn(rpn_ch, 512, 3, 3) * 0.01).astype(np.float32)
    p["RPN/rpn_conv_3x3/b"] = np.zeros(rpn_ch, np.float32)
    p["RPN/rpn_cls_score/W"] = (rs.randn(2 * n_anchors, rpn_ch, 1, 1) * 0.01).astype(np.float32)
    p["RPN/rpn_cls_score/b"] = np.zeros(2 * n_anchors, np.float32)
    p["RPN/rpn_bbox_pred/W"] = (rs.randn(4 * n_anchors, rpn_ch, 1, 1) * 0.01).astype(np.float32)
    p["RPN/rpn_bbox_pred/b"] = np.zeros(4 * n_anchors, np.float32)
    p["fc6/W"] = (rs.randn(4096, roi_feat) * 0.01).astype(np.float32)
    p["fc6/b"] = np.zeros(4096, np.float32)
    p["fc7/W"] = (rs.randn(4096, 4096) * 0.01).astype(np.float32)
    p["fc7/b"] = np.zeros(4096, np.float32)
    p["cls_score/W"] = (rs.randn(num_classes, 4096) * 0.01).astype(np.float32)
    p["cls_score/b"] = np.zeros(num_classes, np.float32)
    p["bbox_pred/W"] = (rs.randn(4 * num_classes, 4096) * 0.001).astype(np.float32)
    p["bbox_pred/b"] = np.zeros(4 * num_classes, np.float32)
    return p


def load_npz(path, model):
    """serializers.load_npz(path, model) for the reference's snapshot format (forward.py:29); see serializers.py."""
    from .serializers import load_npz as _load
    return _load(path, model)


def resnet_params(n_layers=101, seed=2, blocks=None, prefix="trunk/"):
    """Random-init ResNet trunk in chainer's ResNetLayers naming: He-normal convolutions (no bias) and BatchNormalization
    statistics drawn so that test-mode activations stay O(1) through 100 layers."""
    from .models.resnet import BLOCKS, conv_specs
    rs = np.random.RandomState(seed)
    p = {}
    for conv, bn, ci, co, k in conv_specs(tuple(blocks) if blocks is not None else BLOCKS[n_layers]):
        gain = 0.5 if conv.endswith("conv3") or conv.endswith("conv4") else 1.0      # the two summands of a block: keep the sum O(1)
        p[prefix + conv + "/W"] = (rs.randn(co, ci, k, k) * gain * np.sqrt(2.0 / (ci * k * k))).astype(np.float32)
        p[prefix + bn + "/gamma"] = rs.uniform(0.5, 1.0, co).astype(np.float32)
        p[prefix + bn + "/beta"] = (rs.randn(co) * 0.1).astype(np.float32)
        p[prefix + bn + "/avg_mean"] = (rs.randn(co) * 0.1).astype(np.float32)
        p[prefix + bn + "/avg_var"] = rs.uniform(0.5, 1.5, co).astype(np.float32)
    return p
