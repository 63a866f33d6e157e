"""Round 6 sweep: frcnn_anchor_target against the oracle's AnchorTargetLayer (pinned to the reference class by fixtures) over map sizes, ground-truth counts and seeds."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import chainer_faster_rcnn_amd as pkg
import parity_cases as P

rt = pkg.runtime.default_runtime()
n = bad = 0
for (im_h, im_w) in [(600, 1000), (800, 600), (600, 901), (450, 642), (224, 224), (600, 600), (562, 1000), (333, 500), (1000, 600), (96, 128)]:
    fh, fw = im_h, im_w
    for _ in range(4):
        fh, fw = (fh + 1) // 2, (fw + 1) // 2
    for G in (1, 2, 7, 40):
        for seed in range(3):
            n += 1
            try:
                P.check_anchor_target(rt, fh, fw, im_h, im_w, G, seed=seed)
            except ValueError as e:                                  # (96 x 128: no anchor inside the image -- the reference's own ValueError, mirrored by the model class)
                print("%dx%d G %d seed %d: the oracle raises %s" % (im_h, im_w, G, seed, e))
            except AssertionError as e:
                bad += 1
                print("MISMATCH %dx%d (map %dx%d) G %d seed %d: %s" % (im_h, im_w, fh, fw, G, seed, str(e)[:200]))
print("anchor-target sweep: %d cases, %d mismatches" % (n, bad))
