"""`gpu_nms(dets, thresh, device_id=0)` -- the reference's models/gpu_nms.pyx:16-31, statement for statement, over the C symbol
`_nms` it binds (models/gpu_nms.hpp:9-10), which libfrcnn_hip.so exports with that exact signature (csrc/nms_host.hip).
The Cython file itself links against the library unchanged; this module is the same binding through ctypes."""
import ctypes

import numpy as np

from .. import _lib


def gpu_nms(dets, thresh, device_id=0, lib=None):
    if not isinstance(dets, np.ndarray) or dets.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")             # np.ndarray[np.float32_t, ndim=2] dets
    if dets.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % dets.ndim)
    lib = lib or _lib.load()
    boxes_num, boxes_dim = int(dets.shape[0]), int(dets.shape[1])
    num_out = ctypes.c_int(0)
    keep = np.zeros(boxes_num, dtype=np.int32)                                       # gpu_nms.pyx:21-22
    scores = dets[:, 4]
    order = scores.argsort()[::-1]                                                   # :25-26
    sorted_dets = np.ascontiguousarray(dets[order, :])                               # :27-28
    lib._nms(keep.ctypes.data_as(ctypes.c_void_p), ctypes.byref(num_out), sorted_dets.ctypes.data_as(ctypes.c_void_p), boxes_num,
             boxes_dim, ctypes.c_float(thresh), int(device_id))
    if num_out.value < 0:                                                            # the original printed and swallowed errors
        raise _lib.FrcnnError("_nms failed (device_id=%d)" % device_id)
    keep = keep[:num_out.value]
    return list(order[keep])
