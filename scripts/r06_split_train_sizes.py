import sys
sys.path.insert(0, "tests")
import chainer_faster_rcnn_amd as pkg
import train_cases as T
rt = pkg.runtime.default_runtime()
for (h, w, seed) in ((800, 600, 1), (450, 642, 3)):
    try:
        print("split rpn", h, w, T.check_vgg_step(rt, im_h=h, im_w=w, seed=seed, conv_math="split")[:2])
    except AssertionError as e:
        print("split rpn", h, w, "ASSERT", str(e)[:400])
