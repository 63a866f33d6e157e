"""Round 6 sweep: frcnn_nms / frcnn_nms_batched / `_nms` against the reference's cpu_nms arithmetic (the oracle's C restatement, pinned to cpu_nms.pyx by fixtures) on random
box sets of many sizes, densities and thresholds -- with scores drawn from few values (ties everywhere: the ascending-index rule) and from a continuum."""
import numpy as np
import chainer_faster_rcnn_amd as pkg
from oracle import frcnn_oracle as O

rt = pkg.runtime.default_runtime()
bad = n_cases = 0
for n in (1, 2, 63, 64, 65, 127, 129, 1000, 4097, 6000, 12000):
    for seed in range(3):
        rs = np.random.RandomState(1000 * seed + n)
        for dens, tied in ((0.3, False), (3.0, False), (3.0, True), (30.0, True)):
            span = max(60.0, np.sqrt(n / dens) * 40.0)
            xy = rs.uniform(0, span, (n, 2))
            wh = rs.uniform(8, 120, (n, 2))
            sc = rs.choice(np.linspace(0.05, 0.95, 7), n) if tied else rs.uniform(0, 1, n)
            d = np.hstack([xy, xy + wh, sc[:, None]]).astype(np.float32)
            for thr in (0.3, 0.5, 0.7):
                want = O.cpu_nms(d, thr, tie_rule="ascending_index")
                keep, cnt = rt.nms(rt.mem.from_numpy(d), thr)
                k = int(rt.mem.to_numpy(cnt)[0])
                got = rt.mem.to_numpy(keep)[:k].tolist()
                n_cases += 1
                if got != want:
                    bad += 1
                    print("MISMATCH n %d seed %d dens %.1f tied %s thr %.1f: %d vs %d kept, first difference at %s" % (
                        n, seed, dens, tied, thr, len(got), len(want), next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), "length")))
                if len(np.unique(d[:, 4])) == n:                    # no two scores equal (fp32 uniforms DO collide from a few thousand values on): NumPy's own order gives the same list
                    assert O.cpu_nms(d, thr) == want
print("nms sweep: %d cases, %d mismatches" % (n_cases, bad))
