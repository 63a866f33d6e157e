"""`roi_pooling_2d(x, rois, outh, outw, spatial_scale)` -- the call the reference makes at
models/faster_rcnn.py:125-126 (chainer.functions.roi_pooling_2d), on the HIP kernels of csrc/roi_pool.hip.

`ROIPooling2D` keeps Chainer's Function shape: forward((x, rois)) retains `argmax_data`; backward returns
(bottom_diff, None)."""
from ..chainer_compat import unwrap
from ..runtime import default_runtime


class ROIPooling2D(object):
    def __init__(self, outh, outw, spatial_scale, runtime=None):
        self.outh, self.outw, self.spatial_scale = int(outh), int(outw), float(spatial_scale)
        self.rt = runtime or default_runtime()
        self.argmax_data = None
        self._bottom_shape = None

    def forward(self, inputs, train=True):
        x, rois = [unwrap(a) for a in inputs]
        rt = self.rt
        x = rt.asarray(x, "f32")
        rois = rt.asarray(rois, "f32")
        if x.ndim != 4 or int(x.shape[0]) != 1:
            raise ValueError("roi_pooling_2d: x must be (1, C, H, W) (the reference asserts batch size 1)")
        if rois.ndim != 2 or int(rois.shape[1]) != 5:
            raise ValueError("roi_pooling_2d: rois must be (R, 5) [batch, x1, y1, x2, y2]")
        self._bottom_shape = tuple(int(v) for v in x.shape)
        if train:
            y, self.argmax_data = rt.roi_pool_fwd(x, rois, self.outh, self.outw, self.spatial_scale, want_argmax=True)
        else:
            y = rt.roi_pool_fwd(x, rois, self.outh, self.outw, self.spatial_scale)
        return y,

    def backward(self, inputs, grad_outputs):
        if self.argmax_data is None:
            raise RuntimeError("backward before a training-mode forward")
        _, C, H, W = self._bottom_shape
        gy = self.rt.asarray(unwrap(grad_outputs[0]), "f32")
        return self.rt.roi_pool_bwd(gy, self.argmax_data, C, H, W), None


def roi_pooling_2d(x, rois, outh, outw, spatial_scale, runtime=None):
    return ROIPooling2D(outh, outw, spatial_scale, runtime).forward((x, rois), train=False)[0]
