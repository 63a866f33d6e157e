"""Round 6 debugging aid: the stage-2 step's head activations under round 5's convolution picks and under round 6's, same inputs, same masks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import chainer_faster_rcnn_amd as pkg
from chainer_faster_rcnn_amd import synthetic, tuning
from chainer_faster_rcnn_amd.chainer_compat import Variable
from chainer_faster_rcnn_amd.models import FasterRCNN
from chainer_faster_rcnn_amd.train import RCNNTrainer

rt = pkg.runtime.default_runtime()
params = synthetic.params(seed=1)
x = synthetic.image(seed=0, h=600, w=1000)
rs = np.random.RandomState(0)
gt = np.array([[[100, 100, 400, 300, 3], [500, 200, 900, 550, 7], [50, 400, 300, 580, 11], [600, 50, 800, 180, 5]]], np.float32)
info = np.array([[600, 1000]], np.int32)
m6 = ((rs.rand(300, 4096) >= 0.5) * 2.0).astype(np.float32)
m7 = ((rs.rand(300, 4096) >= 0.5) * 2.0).astype(np.float32)
res = {}
for pk in ("5", None, "6"):
    tuning.set("FRCNN_CONV_PICK", pk)
    model = FasterRCNN(runtime=rt)
    model.load_params(params)
    model.rcnn_train = True
    tr = RCNNTrainer(model, dropout_rng="numpy")
    np.random.seed(5)
    out = tr.forward_backward(Variable(x), Variable(info), Variable(gt), masks=(m6, m7))
    rt.mem.synchronize()
    n = int(out["n_rois"])
    a6, a7 = [rt.mem.to_numpy(a) for a in out["head_acts"]]
    G = rt.mem.to_numpy(tr.G).copy()
    res[pk] = (n, a6, a7, G, rt.mem.to_numpy(out["keep_inds"]))
    print("pick", pk, "n", n, "a6", a6.shape, float(np.abs(a6).max()), "keep", len(res[pk][4]))
base = res["5"]
for pk in (None, "6"):
    n, a6, a7, G, keep = res[pk]
    d6 = np.abs(a6 - base[1]); d7 = np.abs(a7 - base[2])
    rows6 = np.where(d6.max(axis=1) > 1e-4 * np.abs(base[1]).max())[0]
    print("pick", pk, "vs 5: a6 max diff", float(d6.max()), "rows with > 1e-4 rel:", rows6[:20], len(rows6), "a7 max diff", float(d7.max()),
          "relu decisions that differ a6/a7:", int(((a6 > 0) != (base[1] > 0)).sum()), int(((a7 > 0) != (base[2] > 0)).sum()),
          "G max rel diff", float(np.abs(G - base[3]).max() / np.abs(base[3]).max()), "same keep", bool(np.array_equal(keep, base[4])))
