#!/usr/bin/env python
"""Hand-build the two snapshot fixtures in CHAINER's on-disk scheme, with NumPy only (no code of this repo's serializer):

  tests/golden/chainer_model_snapshot_small.npz    what `extensions.snapshot_object(model, ...)` / `serializers.save_npz(path, model)`
                                                   writes (/root/reference/train_rpn.py:106-109, forward.py:29): numpy.savez of
                                                   {link path without the leading slash: parameter array in chainer's layout}
  tests/golden/chainer_trainer_snapshot_small.npz  what `extensions.snapshot()` writes (train_rpn.py:101-105): the trainer tree --
                                                   updater/model:main/<path>, updater/optimizer:main/<path>/v (MomentumSGD state),
                                                   updater/optimizer:main/t|epoch, updater/iteration, plus iterator / extension /
                                                   trigger entries (control plane: readers of this repo must ignore them)

for the narrow model the CPU tests use (tests/train_cases.py: SMALL_LAYERS trunk, 64-channel RPN, 32-unit head).  Chainer itself is
not installable here, so the key scheme follows its v1 sources as documented [chainer-ext]:
Link.serialize / Chain.serialize (child name + '/' prefix), Optimizer.serialize (t, epoch, then state[key] under the parameter's
path), StandardUpdater.serialize ('iterator:<name>', 'optimizer:<name>', 'model:<name>', 'iteration'), Trainer.serialize
('updater', 'stop_trigger', 'extensions', 'extension_triggers').   Re-run: python tests/make_chainer_snapshot.py
"""
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LAYERS = [("conv1_1", 3, 64), ("conv2_1", 64, 64), ("conv2_2", 64, 64)]


def model_arrays(rs):
    p = {}
    for name, ci, co in LAYERS:
        p["trunk/%s/W" % name] = (rs.randn(co, ci, 3, 3) * np.sqrt(2.0 / (ci * 9))).astype(np.float32)     # L.Convolution2D: (out, in, kh, kw)
        p["trunk/%s/b" % name] = (rs.randn(co) * 0.01).astype(np.float32)
    p["RPN/rpn_conv_3x3/W"] = (rs.randn(64, 64, 3, 3) * 0.06).astype(np.float32)
    p["RPN/rpn_conv_3x3/b"] = (rs.randn(64) * 0.01).astype(np.float32)
    p["RPN/rpn_cls_score/W"] = (rs.randn(18, 64, 1, 1) * 0.05).astype(np.float32)
    p["RPN/rpn_cls_score/b"] = (rs.randn(18) * 0.01).astype(np.float32)
    p["RPN/rpn_bbox_pred/W"] = (rs.randn(36, 64, 1, 1) * 0.05).astype(np.float32)
    p["RPN/rpn_bbox_pred/b"] = (rs.randn(36) * 0.01).astype(np.float32)
    p["fc6/W"] = (rs.randn(32, 64 * 49) * 0.02).astype(np.float32)                                         # L.Linear: (out, in)
    p["fc6/b"] = (rs.randn(32) * 0.01).astype(np.float32)
    p["fc7/W"] = (rs.randn(32, 32) * 0.2).astype(np.float32)
    p["fc7/b"] = (rs.randn(32) * 0.01).astype(np.float32)
    p["cls_score/W"] = (rs.randn(21, 32) * 0.05).astype(np.float32)
    p["cls_score/b"] = (rs.randn(21) * 0.01).astype(np.float32)
    p["bbox_pred/W"] = (rs.randn(84, 32) * 0.02).astype(np.float32)
    p["bbox_pred/b"] = (rs.randn(84) * 0.01).astype(np.float32)
    return p


def main():
    rs = np.random.RandomState(20170401)
    model = model_arrays(rs)
    with open(os.path.join(OUT, "chainer_model_snapshot_small.npz"), "wb") as f:
        np.savez_compressed(f, **model)
    tr = {"updater/model:main/" + k: v for k, v in model_arrays(rs).items()}
    for k, v in list(tr.items()):
        path = k[len("updater/model:main/"):]
        if path.startswith("trunk/") or path.startswith("RPN/"):                      # rpn_train mode: only trunk + RPN have optimizer state
            tr["updater/optimizer:main/" + path + "/v"] = (rs.randn(*v.shape) * 1e-3).astype(np.float32)
    tr["updater/optimizer:main/t"] = np.asarray(37, dtype=np.int32)
    tr["updater/optimizer:main/epoch"] = np.asarray(0, dtype=np.int32)
    tr["updater/iteration"] = np.asarray(37, dtype=np.int32)
    # control plane of the reference's trainer (ignored by this repo's reader)
    tr["updater/iterator:main/current_position"] = np.asarray(37, dtype=np.int64)
    tr["updater/iterator:main/epoch"] = np.asarray(0, dtype=np.int64)
    tr["updater/iterator:main/is_new_epoch"] = np.asarray(False)
    tr["extensions/LogReport/_trigger/_previous_iteration"] = np.asarray(30, dtype=np.int64)
    tr["extension_triggers/snapshot/_previous_iteration"] = np.asarray(0, dtype=np.int64)
    with open(os.path.join(OUT, "chainer_trainer_snapshot_small.npz"), "wb") as f:
        np.savez_compressed(f, **tr)
    print("wrote", sorted(model)[:3], "...", len(tr), "trainer entries")


if __name__ == "__main__":
    main()
