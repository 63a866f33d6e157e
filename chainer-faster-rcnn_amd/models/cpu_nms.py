"""`cpu_nms(dets, thresh)` with the reference's signature and error behaviour
(models/cpu_nms.pyx:18; callers: models/proposal_layer.py:176-178, forward.py:54) -- executed by the
HIP NMS (csrc/detect.hip via frcnn_nms), not on the CPU.  The name is kept so `from models.cpu_nms import
cpu_nms as nms` (forward.py:12) keeps working.
"""
import numpy as np

from ..runtime import default_runtime


def cpu_nms(dets, thresh, max_out=0, runtime=None):
    """dets: (n,5) float32 [x1,y1,x2,y2,score] (NumPy array or device tensor); thresh: Python float.
    Returns the list of kept indices into `dets`, highest score first (bit-identical to cpu_nms.pyx)."""
    rt = runtime or default_runtime()
    if not isinstance(thresh, float):
        raise TypeError("Argument 'thresh' has incorrect type (expected float, got %s)" % type(thresh).__name__)
    if isinstance(dets, np.ndarray):
        if dets.dtype != np.float32:
            raise ValueError("Buffer dtype mismatch, expected 'float32_t' but got '%s'" % dets.dtype)
        if dets.ndim != 2:
            raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % dets.ndim)
    d = rt.asarray(dets, "f32")
    if d.shape[1] != 5:
        raise ValueError("dets must be (n, 5)")
    keep, n_keep = rt.nms(d, thresh, max_out)
    n = int(rt.mem.to_numpy(n_keep)[0])
    return [int(v) for v in rt.mem.to_numpy(keep)[:n]]
