import sys, os
sys.path.insert(0, "tests")
import chainer_faster_rcnn_amd as pkg
import train_cases as T
rt = pkg.runtime.default_runtime()
h, w, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
try:
    print(T.check_vgg_rcnn_step(rt, im_h=h, im_w=w, seed=seed))
except AssertionError as e:
    print("ASSERT", str(e)[:600])
