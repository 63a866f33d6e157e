"""Base anchor table (host side, once at construction) -- interface of the reference's
models/generate_anchors.py:47 `generate_anchors(base_size=15, ratios, scales)`.

Closed form of the enumeration: the reference window [0,0,base,base] has side s = base+1 and centre
c = base/2; for ratio r the rounded size is w_r = rint(sqrt(s*s/r)), h_r = rint(w_r*r) (round-half-even,
as np.rint), each scaled by every `scale`, all sharing the centre c.  Output rows are ratio-major, then
scale; corners are c -/+ (size-1)/2.  NB: with base_size=15 this yields [-84,-40,99,55] ..., not the
1-based table quoted in the reference file's comment.
"""
import numpy as np


def generate_anchors(base_size=15, ratios=(0.5, 1, 2), scales=(4, 8, 16, 32)):
    ratios = np.asarray(ratios, dtype=np.float64).reshape(-1, 1)
    scales = np.asarray(scales, dtype=np.float64).reshape(1, -1)
    side = float(base_size) + 1.0
    centre = 0.5 * (side - 1.0)
    w_r = np.rint(np.sqrt(side * side / ratios))
    h_r = np.rint(w_r * ratios)
    half_w = 0.5 * ((w_r * scales).ravel() - 1.0)
    half_h = 0.5 * ((h_r * scales).ravel() - 1.0)
    return np.stack([centre - half_w, centre - half_h, centre + half_w, centre + half_h], axis=1)
