"""`load_npz(path, model)` / `save_npz(path, model)` for the reference's snapshot format (forward.py:29
`serializers.load_npz('data/VGG16_faster_rcnn_final.model', model)`, train_rpn.py:101-109 snapshot_object): a NumPy .npz whose
keys are chainer link paths -- `trunk/conv1_1/W` (co,ci,3,3), `RPN/rpn_cls_score/b`, `fc6/W` (out,in), ....
"""
import numpy as np


def namedparams(model):
    """Yield (link path, device array in chainer's layout) for every parameter the model holds."""
    rt = model.rt
    for name, link in model.trunk.links.items():
        yield "trunk/%s/W" % name, link.W
        yield "trunk/%s/b" % name, link.b
    rpn = model.RPN
    yield "RPN/rpn_conv_3x3/W", rpn.rpn_conv_3x3.W
    yield "RPN/rpn_conv_3x3/b", rpn.rpn_conv_3x3.b
    for n, store in (("rpn_cls_score", rpn.rpn_cls_score), ("rpn_bbox_pred", rpn.rpn_bbox_pred)):
        W = store["W"]
        yield "RPN/%s/W" % n, W.reshape(int(W.shape[0]), int(W.shape[1]), 1, 1)
        yield "RPN/%s/b" % n, store["b"]
    for n in ("fc6", "fc7", "cls_score", "bbox_pred"):
        lin = getattr(model, n)
        if lin.W is not None:
            yield n + "/W", lin.W
            yield n + "/b", lin.b
    del rt


def save_npz(path, model, trainer=None):
    """Write the model's parameters; pass the RPNTrainer after training so the packed weights are synced back first."""
    if trainer is not None:
        trainer.sync_params()
    rt = model.rt
    np.savez(path, **{k: rt.mem.to_numpy(rt.mem.contiguous(v)) for k, v in namedparams(model)})


def load_npz(path, model):
    with np.load(path) as f:
        params = {k: f[k] for k in f.files}
    model.trunk.load_params(params, "trunk/")
    model.RPN.load_params(params, "RPN/")
    for n in ("fc6", "fc7", "cls_score", "bbox_pred"):
        if n + "/W" in params:
            getattr(model, n).set(params[n + "/W"], params[n + "/b"])
    # everything derived from the parameters (the stacked inference head above all: forward_device() would otherwise keep the OLD
    # cls_score / bbox_pred rows next to the new trunk) is rebuilt now; links adopted by a trainer were written through in place
    model._last_trainer = None
    if hasattr(model, "_stack_head") and all(getattr(model, n).W is not None for n in ("cls_score", "bbox_pred")):
        model._stack_head()
    return model
