"""Box decoding / clipping with the reference's function names (models/bbox_transform.py:41-99), executed by
the HIP kernels of csrc/head.hip.  Inside ProposalLayer / FasterRCNN the same arithmetic is fused into the
proposal and head-decode kernels; these stand-alone forms serve callers that use the functions directly."""
from ..chainer_compat import unwrap
from ..runtime import default_runtime


def bbox_transform_inv(boxes, trans, runtime=None):
    """boxes (N,4), trans (N,4*C) float32 -> (N,4*C) device array (bbox_transform.py:41-76)."""
    rt = runtime or default_runtime()
    b = rt.asarray(unwrap(boxes), "f32")
    t = rt.asarray(unwrap(trans), "f32")
    if int(b.shape[0]) == 0:
        return rt.mem.zeros((0, int(t.shape[1])), "f32")        # the reference's empty-input guard (:48-49)
    return rt.bbox_transform_inv(b, t)


def clip_boxes(boxes, im_shape, runtime=None):
    """In-place clip of (N,4*C) boxes to x in [0,W-1], y in [0,H-1]; im_shape = (H, W) (bbox_transform.py:79-99)."""
    rt = runtime or default_runtime()
    b = rt.asarray(unwrap(boxes), "f32")
    im = unwrap(im_shape)
    im = rt.mem.to_numpy(im) if rt.mem.is_array(im) else im
    return rt.clip_boxes_(b, int(im[0]), int(im[1]))
