"""forward.py:48-58 (`draw_result` without the drawing): per class 1..20, greedy NMS at `nms_thresh` over the 300 scored
boxes and the confidence cut -- the step immediately downstream of FasterRCNN.__call__.

On the device the 20 problems are ONE batched launch sequence of the NMS kernels (frcnn_class_dets +
frcnn_nms_batched); only the kept rows come back to the host.
"""
import numpy as np

from .runtime import default_runtime


def detections(cls_prob, pred_boxes, nms_thresh=0.3, conf=0.8, im_scale=1.0, runtime=None):
    """cls_prob (R,ncls), pred_boxes (R,4*ncls): device arrays / NumPy.  Returns {cls_id: (k,5) float32 rows
    [x1,y1,x2,y2,score]} for cls_id in 1..ncls-1, boxes divided by im_scale, rows in descending-score (NMS keep) order,
    score >= conf -- exactly the rows forward.py draws."""
    rt = runtime or default_runtime()
    cp = rt.asarray(cls_prob, "f32")
    pb = rt.asarray(pred_boxes, "f32")
    R, ncls = int(cp.shape[0]), int(cp.shape[1])
    out = {}
    if R == 0:
        return {c: np.zeros((0, 5), np.float32) for c in range(1, ncls)}
    dets = rt.class_dets(cp, pb)
    dets = dets if rt.mem.is_array(dets) and getattr(dets, "is_contiguous", lambda: True)() else rt.mem.contiguous(dets)
    keep, n_keep = rt.nms_batched(dets, float(nms_thresh))
    dets_h, keep_h, n_h = rt.mem.to_numpy(dets), rt.mem.to_numpy(keep), rt.mem.to_numpy(n_keep)
    for c in range(1, ncls):
        d = dets_h[c - 1][keep_h[c - 1][:int(n_h[c - 1])]]
        d = d[d[:, -1] >= conf].copy()
        d[:, :4] /= im_scale
        out[c] = d
    return out


PIXEL_MEANS = np.array([[[102.9801, 115.9465, 122.7717]]])        # forward.py:22 (BGR)


def img_preprocessing(orig_img, pixel_means=PIXEL_MEANS, max_size=1000, scale=600, runtime=None):
    """forward.py:33-45 with the reference's signature: uint8 HWC image -> ((1? no: C,H,W) float32 device array, im_scale).
    The scale rule is the reference's host arithmetic; mean subtraction, the bilinear resize and the HWC->CHW transpose
    run in one device kernel on the uint8 upload.  Returns (img (C,OH,OW) device f32, im_scale)."""
    rt = runtime or default_runtime()
    h, w = orig_img.shape[0], orig_img.shape[1]
    im_size_min, im_size_max = min(h, w), max(h, w)
    im_scale = float(scale) / float(im_size_min)
    if np.round(im_scale * im_size_max) > max_size:
        im_scale = float(max_size) / float(im_size_max)
    oh, ow = int(np.rint(h * im_scale)), int(np.rint(w * im_scale))           # cv.resize: dsize = round(size * f), ties to even
    dev = rt.asarray(np.ascontiguousarray(orig_img, dtype=np.uint8), "u8") if isinstance(orig_img, np.ndarray) else orig_img
    out = rt.preprocess_u8(dev, np.asarray(pixel_means, dtype=np.float64).ravel(), im_scale, (oh, ow))
    return out[0], im_scale
