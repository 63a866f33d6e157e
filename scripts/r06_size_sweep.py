"""Round 6 sweep: the integer-exact stages (proposal pipeline, RoI pooling) of the fp32 forward against the oracle ON THE DEVICE'S OWN MAPS, over image sizes and seeds:
indices / scores / RoIs bit for bit under a correctly rounded exp with equal scores in ascending anchor index (the kernel's documented rule), pool5 bit for bit; beside it the
index lists under NumPy's own order of equal scores and under the host's NumPy exp."""
import sys
import numpy as np
import chainer_faster_rcnn_amd as pkg
from chainer_faster_rcnn_amd import synthetic
from chainer_faster_rcnn_amd.models import FasterRCNN
from oracle import frcnn_oracle as O
from oracle import parity

rt = pkg.runtime.default_runtime()
params = synthetic.params(seed=1)
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
model = FasterRCNN(runtime=rt, conv_dtype=dtype, head_dtype=dtype)
model.load_params(params)
sizes = [(600, 1000), (800, 600), (600, 901), (450, 642), (600, 800), (1000, 600), (224, 224), (600, 600), (562, 1000), (600, 999), (601, 903), (333, 500)]
bad = 0
for (h, w) in sizes:
    for seed in range(seed0, seed0 + 3):
        x = synthetic.image(seed=seed, h=h, w=w)
        info = np.array([[h, w]], dtype=np.int32)
        dev = parity.device_forward_host(rt, model, rt.mem.from_numpy(x), h, w)
        n = int(dev["n_out"][0])
        p2, s2, d2 = O.proposal_layer(dev["rpn_cls_prob"], dev["rpn_bbox_pred"], info, train=False, return_debug=True)
        O.EXP = lambda v: np.exp(np.asarray(v, np.float64)).astype(np.float32)
        p3, s3, d3 = O.proposal_layer(dev["rpn_cls_prob"], dev["rpn_bbox_pred"], info, train=False, return_debug=True, tie_rule="ascending_index")
        p5, s5, d5 = O.proposal_layer(dev["rpn_cls_prob"], dev["rpn_bbox_pred"], info, train=False, return_debug=True)
        O.EXP = np.exp
        got = dev["src_index"][:n].astype(np.int64)
        ex_r = n == len(p3) and np.array_equal(got, d3["src_index"]) and np.array_equal(dev["rois"][:n], p3) and np.array_equal(dev["probs"][:n].ravel(), s3.ravel())
        ex_h = n == len(p2) and np.array_equal(got, d2["src_index"])
        brois = np.concatenate([np.zeros((n, 1), np.float32), dev["rois"][:n]], 1)
        pool_ok = np.array_equal(dev["pool5"][:n], O.roi_pooling_2d(dev["feat"], brois, 7, 7, 1.0 / 16))
        cp, pb, hd = O.rcnn_head(params, dev["pool5"][:n], dev["rois"][:n], info)
        herr = float(np.abs(dev["cls_prob"][:n] - cp).max() / max(np.abs(cp).max(), 1e-9))
        ex_np = n == len(p5) and np.array_equal(got, d5["src_index"])
        flag = "" if (ex_r and pool_ok) else "   <-- NOT EXACT"
        bad += 0 if (ex_r and pool_ok) else 1
        print("%s %4dx%-4d seed %d: n %3d  rounded exp + index ties: bit-exact %s | rounded exp, NumPy's tie order: index-exact %s | host exp, NumPy's tie order: index-exact %s | pool5 exact %s | cls_prob rel %.2e%s" % (dtype, h, w, seed, n, ex_r, ex_np, ex_h, pool_ok, herr, flag))
print("NOT EXACT cases:", bad)
