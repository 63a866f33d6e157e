"""`load_npz(path, model)` / `save_npz(path, model)` for the reference's snapshot format (forward.py:29
`serializers.load_npz('data/VGG16_faster_rcnn_final.model', model)`, train_rpn.py:101-109 snapshot_object): a NumPy .npz whose
keys are chainer link paths -- `trunk/conv1_1/W` (co,ci,3,3), `RPN/rpn_cls_score/b`, `fc6/W` (out,in), ....

`save_trainer_npz` / `load_trainer_npz` cover train_rpn.py:101-105 `extensions.snapshot()`: chainer v1 serialises the trainer as
`updater/model:main/<link path>` (the parameters), `updater/optimizer:main/<link path>/v` (MomentumSGD's velocity, in the
parameter's shape), `updater/optimizer:main/t`, `updater/optimizer:main/epoch`, `updater/iteration` [chainer-ext: Trainer.serialize
-> StandardUpdater.serialize -> Optimizer.serialize].  Iterator / extension / trigger entries of such a file are the reference's
control plane (out of scope): ignored on load, not written on save.
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
    if hasattr(model, "sync_trainers"):
        model.sync_trainers()          # every trainer that has updated the model (an rpn -> rcnn alternation leaves two)
    if trainer is not None:
        trainer.sync_params()
    rt = model.rt
    with open(path, "wb") as f:        # chainer's save_npz writes through a file object: the name is kept as given (`..._final.model`)
        np.savez(f, **{k: rt.mem.to_numpy(rt.mem.contiguous(v)) for k, v in namedparams(model)})


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
    model._trainers = []
    if hasattr(model, "_stack_head") and all(getattr(model, n).W is not None for n in ("cls_score", "bbox_pred")):
        model._stack_head()
    return model


TRAINER_MODEL = "updater/model:main/"
TRAINER_OPT = "updater/optimizer:main/"


def save_trainer_npz(path, trainer):
    """Resumable snapshot of a training run (train_rpn.py:101-105): parameters + momentum velocities + iteration count."""
    model, rt = trainer.model, trainer.rt
    if hasattr(model, "sync_trainers"):
        model.sync_trainers()          # parameters another trainer of the same model owns (the RPN after an rpn -> rcnn alternation)
    trainer.sync_params()
    d = {TRAINER_MODEL + k: rt.mem.to_numpy(rt.mem.contiguous(v)) for k, v in namedparams(model)}
    for k, v in trainer.flat_to_chainer_layout(trainer.V).items():
        d[TRAINER_OPT + k + "/v"] = v
    d[TRAINER_OPT + "t"] = np.asarray(trainer.iteration, dtype=np.int32)
    d[TRAINER_OPT + "epoch"] = np.asarray(0, dtype=np.int32)
    d["updater/iteration"] = np.asarray(trainer.iteration, dtype=np.int32)
    with open(path, "wb") as f:
        np.savez(f, **d)


def load_trainer_npz(path, trainer):
    """Resume: parameters into the model (written THROUGH the trainer's windows), velocities into trainer.V, iteration count."""
    import os
    import tempfile
    with np.load(path) as f:
        arrays = {k: f[k] for k in f.files}
    params = {k[len(TRAINER_MODEL):]: v for k, v in arrays.items() if k.startswith(TRAINER_MODEL)}
    fd, tmp = tempfile.mkstemp(suffix=".npz")
    os.close(fd)
    try:
        np.savez(tmp, **params)
        load_npz(tmp, trainer.model)
    finally:
        os.remove(tmp)
    trainer._ensure_adopted()
    vel = {k[len(TRAINER_OPT):-2]: v for k, v in arrays.items() if k.startswith(TRAINER_OPT) and k.endswith("/v")}
    trainer.chainer_layout_to_flat(vel, trainer.V)
    if "updater/iteration" in arrays:
        trainer.iteration = int(arrays["updater/iteration"])
    return trainer
