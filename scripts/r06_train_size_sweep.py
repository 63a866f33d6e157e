"""Round 6 sweep: the two training steps' full-size parity checks (tests/train_cases.py: loss and every gradient against the oracle's autograd, float64 arbiter) at image
sizes other than 600 x 1000."""
import sys, os, time, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import chainer_faster_rcnn_amd as pkg
import train_cases as T

rt = pkg.runtime.default_runtime()
cases = [(800, 600, 1), (600, 901, 2), (450, 642, 3)] if len(sys.argv) < 2 else [(600, 800, 4), (562, 1000, 5), (600, 600, 6), (1000, 600, 7)]
bad = 0
for (h, w, seed) in cases:
    for name, fn in (("rpn", T.check_vgg_step), ("rcnn", T.check_vgg_rcnn_step)):
        t0 = time.time()
        try:
            fn(rt, im_h=h, im_w=w, seed=seed)
            print("TRAIN-SWEEP %s step %dx%d seed %d: ok (%.0f s)" % (name, h, w, seed, time.time() - t0), flush=True)
        except Exception:
            bad += 1
            print("TRAIN-SWEEP %s step %dx%d seed %d: FAILED" % (name, h, w, seed), flush=True)
            traceback.print_exc()
print("TRAIN-SWEEP failures:", bad)
