"""VGG-16 trunk with the reference's layer list (models/vgg16.py:38-82, class VGG16Prev): 13 x
[conv3x3 pad 1 + ReLU], 4 x max-pool 2x2 (ceil mode), stops after relu5_3 -- on the fp32 MFMA conv kernel
(csrc/conv.hip).  Parameters keep Chainer's link paths (`conv1_1/W` (co,ci,3,3), `conv1_1/b`)."""
import numpy as np

from .. import tuning as _tuning
from ..chainer_compat import unwrap
from ..runtime import default_runtime

LAYERS = [
    ("conv1_1", 3, 64), ("conv1_2", 64, 64), "pool",
    ("conv2_1", 64, 128), ("conv2_2", 128, 128), "pool",
    ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), "pool",
    ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), "pool",
    ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512),
]


class Conv3x3(object):
    """L.Convolution2D(ci, co, 3, 1, 1): holds W (co,ci,3,3) + b on device and the kernel's packed copy."""

    def __init__(self, rt, cin, cout, conv_dtype="f32"):
        self.rt, self.cin, self.cout, self.conv_dtype = rt, cin, cout, conv_dtype
        self.W = self.b = self.Wp = self.Wb = self.Ws = None

    def set(self, W, b):
        rt = self.rt
        W = np.ascontiguousarray(W, dtype=np.float32) if isinstance(W, np.ndarray) else W
        assert tuple(W.shape) == (self.cout, self.cin, 3, 3), (tuple(W.shape), self.cout, self.cin)
        self.W = rt.asarray(W, "f32")
        b = rt.asarray(np.ascontiguousarray(b, dtype=np.float32) if isinstance(b, np.ndarray) else b, "f32")
        Wp = rt.pack_conv3x3_w(self.W)
        if getattr(self, "_adopted", False):                  # Wp / b are windows of a trainer's flat parameter buffer: write THROUGH them,
            self.Wp[...] = Wp                                 # or the trainer would keep updating an orphaned copy (load_npz after training)
            self.b[...] = b
        else:
            self.Wp, self.b = Wp, b
        self.refresh_bf16()

    def refresh_bf16(self):
        if self.conv_dtype == "bf16":
            self.Wb = self.rt.bf16_pack_conv_w(self.W, 3)     # [tap][CoutP][CinP] bf16 (csrc/conv_bf16.hip)
        elif self.conv_dtype == "f32s":
            self.Ws = self.rt.f32s_pack_conv_w(self.W)        # three bf16 terms per fp32 weight (csrc/conv_f32s.hip)

    def __call__(self, x, relu=True, out=None, cfg=-1):
        return self.rt.conv3x3(x, self.Wp, self.b, relu=relu, out=out, cfg=cfg)

    def relu_pool(self, x):
        """conv + ReLU + the following F.MaxPooling2D(2,2) in one launch (the pool lives in the conv kernel's epilogue)."""
        return self.rt.conv_ex(x, self.Wp, self.b, 3, act=4)

    def bf16(self, x_blk, relu=True, out_f32_nchw=False, pool=False):
        """x [CinP/16][H][W][16] bf16 (channel-blocked) -> [CoutP/16][H][W][16] bf16 (or fp32 NCHW)."""
        return self.rt.conv_bf16(x_blk, self.Wb, self.b, self.cin, self.cout, 3, relu=relu, out_f32_nchw=out_f32_nchw, pool=pool)


    def f32s(self, x_split, relu=True, out_f32_nchw=False, pool=False):
        """x split tensor [3][CinP/16][H][W][16] -> split tensor (or fp32 NCHW): the fp32 convolution as six bf16 MFMA products."""
        return self.rt.conv3x3_f32s(x_split, self.Ws, self.b, self.cin, self.cout, relu=relu, out_f32_nchw=out_f32_nchw, pool=pool)


class VGG16Prev(object):
    def __init__(self, train=False, runtime=None, layers=None, conv_dtype="f32"):
        self.rt = runtime or default_runtime()
        self.train = train
        # "f32" (BASELINE config 2: fp32 MFMA), "f32s" (the same fp32 convolutions computed as six bf16 MFMA products of 3-way
        # split operands -- csrc/conv_f32s.hip) or "bf16" (config 3)
        self.conv_dtype = conv_dtype
        self.fuse_pool = True        # inference: conv -> ReLU -> pool as one launch (the trainer keeps the pre-pool maps instead)
        self.layers = list(layers) if layers is not None else LAYERS     # (tests build narrow / shallow variants)
        self.links = {}
        for l in self.layers:
            if l != "pool":
                self.links[l[0]] = Conv3x3(self.rt, l[1], l[2], conv_dtype)
                setattr(self, l[0], self.links[l[0]])

    def load_params(self, params, prefix="trunk/"):
        for name, link in self.links.items():
            link.set(params[prefix + name + "/W"], params[prefix + name + "/b"])

    def namedparams(self, prefix="trunk/"):
        for name, link in self.links.items():
            yield prefix + name + "/W", link.W
            yield prefix + name + "/b", link.b

    def __call__(self, x, timer=None, collect=None):
        """`collect` (a dict, optional) receives every launch's output under the layer's name -- `pool<n>` for a convolution
        whose ReLU + max-pool ran fused (the pre-pool map never exists then); the full-size parity tests read it."""
        rt = self.rt
        h = rt.asarray(unwrap(x), "f32")
        assert h.ndim == 4 and int(h.shape[0]) == 1, "batch size 1 (models/faster_rcnn.py:77)"
        if self.conv_dtype == "bf16":
            return self._call_bf16(h, timer, collect)
        if self.conv_dtype == "f32s":
            return self._call_f32s(h, timer, collect)
        n_pool, skip = 0, False
        for idx, l in enumerate(self.layers):
            if l == "pool":
                n_pool += 1
                if skip:                                                  # already applied inside the previous conv
                    skip = False
                    continue
                h = rt.maxpool2x2(h)
                if timer:
                    timer.mark("pool%d" % n_pool)
                if collect is not None:
                    collect["pool%d" % n_pool] = h
            else:
                fuse = self.fuse_pool and idx + 1 < len(self.layers) and self.layers[idx + 1] == "pool" and l[2] % 64 == 0
                h = self.links[l[0]].relu_pool(h) if fuse else self.links[l[0]](h, relu=True)
                skip = fuse
                if timer:
                    timer.mark(l[0])
                if collect is not None:
                    collect["pool%d" % (n_pool + 1) if fuse else l[0]] = h
        return h


    def conv1_pair_applies(self):
        """conv1_1 (<= 3 -> 64), conv1_2 (64 -> 64), pool -- VGG-16's first three entries -- run as one launch on the bf16 chain
        (FRCNN_BF16_CONV1_PAIR=0: the two-launch form, for A/B measurements)."""
        import os
        L = self.layers
        return (self.conv_dtype == "bf16" and self.fuse_pool and _tuning.get("FRCNN_BF16_CONV1_PAIR", "1") != "0" and len(L) >= 3 and L[0] != "pool"
                and L[1] != "pool" and L[2] == "pool" and L[0][1] <= 3 and L[0][2] == 64 and L[1][1] == 64 and L[1][2] == 64
                and not getattr(self, "generic_first_layer", False))

    def _call_bf16(self, x, timer, collect=None):
        """bf16 chain: fp32 NCHW image -> channel-blocked bf16 -> 13 bf16 convs / 4 pools -> conv5_3 back as fp32 NCHW."""
        rt = self.rt
        h = None                             # converted lazily: a first layer with <= 3 input channels reads the fp32 NCHW image itself
        n_pool, cout, skip = 0, int(x.shape[1]), False
        pair = self.conv1_pair_applies() and collect is None            # (a per-layer collection wants conv1_1's map: the two-launch form then)
        for idx, l in enumerate(self.layers):
            if pair and idx < 3:
                # conv1_1 + ReLU + conv1_2 + ReLU + pool1 as ONE launch (csrc/conv_bf16_pair.hip): the 64-channel map between them stays in LDS
                if idx == 0:
                    l1, l2 = self.links[self.layers[0][0]], self.links[self.layers[1][0]]
                    h = rt.conv1_pair_bf16(x, l1.W, l1.b, l2.Wb, l2.b)
                    n_pool, cout = 1, int(self.layers[1][2])
                    if timer:
                        timer.mark(self.layers[0][0])                   # (the whole launch is booked on conv1_2: conv1_1 has no launch of its own)
                        timer.mark(self.layers[1][0])
                continue
            if l == "pool":
                n_pool += 1
                if skip:
                    skip = False
                    continue
                if h is None:
                    h = rt.bf16_from_nchw(x)
                h = rt.maxpool2x2_bf16(h)
                if timer:
                    timer.mark("pool%d" % n_pool)
                if collect is not None:
                    collect["pool%d" % n_pool] = (h, cout)
            else:
                fuse = self.fuse_pool and idx + 1 < len(self.layers) and self.layers[idx + 1] == "pool"
                link = self.links[l[0]]
                if h is None and not fuse and l[1] <= 3 and l[2] <= 64 and not getattr(self, "generic_first_layer", False):
                    h = rt.conv1_bf16(x, link.W, link.b, relu=True)            # conv1_1: straight from the fp32 NCHW image
                else:
                    if h is None:
                        h = rt.bf16_from_nchw(x)
                    h = link.bf16(h, relu=True, pool=fuse)
                skip = fuse
                cout = l[2]
                if timer:
                    timer.mark(l[0])
                if collect is not None:                      # channel-blocked bf16 arrays: (array, channels)
                    collect["pool%d" % (n_pool + 1) if fuse else l[0]] = (h, cout)
        self.feat_bf16 = h                   # the channel-blocked bf16 map itself: the RPN's bf16 conv takes it as is
        self.feat_shape = (1, cout, int(h.shape[1]), int(h.shape[2]))
        if getattr(self, "skip_nchw", False):
            return None                      # the caller pools straight from the blocked map (FasterRCNN.forward_device)
        feat = rt.bf16_to_nchw(h, cout)
        if timer:
            timer.mark("to_nchw")
        return feat


    def _call_f32s(self, x, timer, collect=None):
        """fp32 chain on split tensors: fp32 NCHW image -> three bf16 terms per value (exact) -> 13 convolutions, each an fp32
        convolution (six bf16 MFMA products, fp32 accumulation, bias / ReLU / max-pool in fp32, result split again) -> conv5_3 back
        as fp32 NCHW (h + m + l: exact)."""
        rt = self.rt
        h = None                             # the image is split lazily: a first layer with <= 3 input channels reads it as fp32 NCHW
        feat = None
        n_pool, cout, skip = 0, int(x.shape[1]), False
        as_nchw = lambda t, c: rt.f32s_to_nchw(t, c)
        for idx, l in enumerate(self.layers):
            if l == "pool":
                n_pool += 1
                if skip:
                    skip = False
                    continue
                if h is None:
                    h = rt.f32s_from_nchw(x)
                h = rt.f32s_from_nchw(rt.maxpool2x2(as_nchw(h, cout)))        # unfused pool (tests only): through fp32 NCHW
                if timer:
                    timer.mark("pool%d" % n_pool)
                if collect is not None:
                    collect["pool%d" % n_pool] = as_nchw(h, cout)
            else:
                fuse = self.fuse_pool and idx + 1 < len(self.layers) and self.layers[idx + 1] == "pool"
                link = self.links[l[0]]
                if h is None and not fuse and l[1] <= 3 and l[2] <= 64 and not getattr(self, "generic_first_layer", False):
                    h = rt.conv1_f32s(x, link.W, link.b, relu=True)           # conv1_1: straight from the fp32 NCHW image
                else:
                    if h is None:
                        h = rt.f32s_from_nchw(x)
                    if idx + 1 == len(self.layers) and not fuse:
                        # the last convolution writes its result twice: the split tensor (rpn_conv_3x3's input) and fp32 NCHW (RoI pooling)
                        h, feat = rt.conv3x3_f32s_train(h, link.Ws, link.b, l[1], l[2], relu=True, want_split=True, want_nchw=True)
                    else:
                        h = link.f32s(h, relu=True, pool=fuse)
                skip = fuse
                cout = l[2]
                if timer:
                    timer.mark(l[0])
                if collect is not None:
                    collect["pool%d" % (n_pool + 1) if fuse else l[0]] = as_nchw(h, cout)
        self.feat_split = h                  # the split tensor itself: the RPN's convolution takes it as is
        if feat is None:                     # (a trunk that ends in a pool or in its first layer)
            feat = rt.f32s_to_nchw(h, cout)
            if timer:
                timer.mark("to_nchw")
        return feat


VGG16 = VGG16Prev   # the reference's default trunk_class needs a caffemodel download; same network
