"""Round 6 probe: where the device's proposals differ from the oracle's ProposalLayer run on the DEVICE's own maps (800 x 600, fp32), are the scores tied?"""
import sys
import numpy as np
import chainer_faster_rcnn_amd as pkg
from chainer_faster_rcnn_amd import synthetic
from chainer_faster_rcnn_amd.models import FasterRCNN
from oracle import frcnn_oracle as O
from oracle import parity

im_h, im_w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 600)
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
rt = pkg.runtime.default_runtime()
params = synthetic.params(seed=1)
x = synthetic.image(seed=seed, h=im_h, w=im_w)
info = np.array([[im_h, im_w]], dtype=np.int32)
model = FasterRCNN(runtime=rt, conv_dtype=dtype, head_dtype=dtype)
model.load_params(params)
dev = parity.device_forward_host(rt, model, rt.mem.from_numpy(x), im_h, im_w)
n = int(dev["n_out"][0])
p2, s2, d2 = O.proposal_layer(dev["rpn_cls_prob"], dev["rpn_bbox_pred"], info, train=False, return_debug=True)
got, want = dev["src_index"][:n].astype(np.int64), d2["src_index"].astype(np.int64)
print("n", n, len(want), "equal", np.array_equal(got, want))
bad = np.nonzero(got[:min(n, len(want))] != want[:min(n, len(want))])[0]
print("differing positions", bad.tolist())
A = 9
fg = dev["rpn_cls_prob"][0][A:].transpose(1, 2, 0).ravel()
for i in bad[:10]:
    print(" pos", i, "device src", got[i], "score bits", hex(fg[got[i]].view(np.uint32)), "| oracle src", want[i], "score bits", hex(fg[want[i]].view(np.uint32)),
          "| device probs", hex(dev["probs"][i].view(np.uint32).ravel()[0]))
order = d2["order"]
ss = d2["sorted_scores"].ravel()
ties = np.nonzero(ss[1:] == ss[:-1])[0]
print("tied adjacent pairs in the oracle's sorted top-%d: %d; first positions %s" % (len(ss), len(ties), ties[:10].tolist()))
# the kernel's documented rule: descending score, ascending index among equals == a stable argsort of -score
k0 = d2["keep0"]
fgk = fg[k0] if False else dev["rpn_cls_prob"][0][A:].transpose(1, 2, 0).reshape(-1)[k0]
stable = np.argsort(-fgk.astype(np.float64), kind="stable")[:len(order)]
print("oracle order == stable descending order:", np.array_equal(order, stable), "; positions where they differ:", np.nonzero(order != stable)[0][:10].tolist())
# the kernel's exp is the correctly rounded one (double exp, rounded once); NumPy's float32 exp is a SIMD polynomial that is not: re-run the oracle with it
O.EXP = lambda v: np.exp(np.asarray(v, np.float64)).astype(np.float32)
p3, s3, d3 = O.proposal_layer(dev["rpn_cls_prob"], dev["rpn_bbox_pred"], info, train=False, return_debug=True)
O.EXP = np.exp
print("with a correctly rounded exp: index-exact", np.array_equal(got, d3["src_index"].astype(np.int64)), "rois bit-exact", np.array_equal(dev["rois"][:n], p3),
      "scores bit-exact", np.array_equal(dev["probs"][:n].ravel(), s3.ravel()))
print("NumPy exp vs correctly rounded exp on the deltas: differing values", int((np.exp(dev["rpn_bbox_pred"]) != np.exp(dev["rpn_bbox_pred"].astype(np.float64)).astype(np.float32)).sum()), "of", dev["rpn_bbox_pred"].size)

want3 = d3["src_index"].astype(np.int64)
bad3 = np.nonzero(got[:min(n, len(want3))] != want3[:min(n, len(want3))])[0]
print("rounded exp: differing positions", bad3[:12].tolist(), "of", n, len(want3))
for i in bad3[:6]:
    print(" pos", i, "device src", got[i], "score", hex(fg[got[i]].view(np.uint32)), "| oracle src", want3[i], "score", hex(fg[want3[i]].view(np.uint32)))
# is the device's list the oracle's under the kernel's tie rule (descending score, ascending index among equals)?
k0 = d3["keep0"]
fgk = dev["rpn_cls_prob"][0][A:].transpose(1, 2, 0).reshape(-1)[k0]
stable = np.argsort(-fgk.astype(np.float64), kind="stable")[:6000]
O.EXP = lambda v: np.exp(np.asarray(v, np.float64)).astype(np.float32)
anchors = O.generate_anchors()
all_bbox = O.generate_all_bbox(anchors, dev["rpn_bbox_pred"].shape[2], dev["rpn_bbox_pred"].shape[3], 16).astype(np.float32)
props = O.clip_boxes(O.bbox_transform_inv(all_bbox, dev["rpn_bbox_pred"][0].transpose(1, 2, 0).reshape(-1, 4)), info[0])
O.EXP = np.exp
pk = props[k0][stable]
dets = np.hstack((pk, fgk[stable][:, None])).astype(np.float32)
keep = O.cpu_nms(dets, 0.7)[:300]
src_stable = k0[stable][keep]
print("device == oracle under the ascending-index tie rule:", np.array_equal(got, src_stable), "; ties among the oracle's sorted scores:", int((np.diff(d3["sorted_scores"].ravel()) == 0).sum()))
