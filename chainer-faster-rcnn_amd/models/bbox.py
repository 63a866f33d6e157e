"""`bbox_overlaps(boxes, query_boxes)` with the reference's signature and dtype contract (models/bbox.pyx:16-56):
float64 (N,4) x float64 (K,4) -> float64 (N,K), executed by frcnn_bbox_overlaps_f64 (csrc/train.hip)."""
import numpy as np

from ..runtime import default_runtime


def bbox_overlaps(boxes, query_boxes, runtime=None):
    rt = runtime or default_runtime()
    for a in (boxes, query_boxes):
        if isinstance(a, np.ndarray):
            if a.dtype != np.float64:
                raise ValueError("Buffer dtype mismatch, expected 'DTYPE_t' but got '%s'" % a.dtype)
            if a.ndim != 2:
                raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % a.ndim)
    return rt.bbox_overlaps(rt.asarray(boxes, "f64"), rt.asarray(query_boxes, "f64"))
