"""ResNet trunk with the reference's interface (models/resnet.py:11-45: `ResNet(n_layers)`, `model(x)` -> the `res5`
activation of chainer's ResNetLayers): conv1 7x7/2 + BN + ReLU, max-pool 3x3/2, res2..res5 of Caffe-style bottlenecks
(stride on the first 1x1 of a stage, projection shortcut in block `a`), BatchNormalization in TEST mode.

Wiring chosen for Faster R-CNN (the reference never instantiates FasterRCNN with it): the literal one --
`res5` (1,2048,H/32,W/32), so `FasterRCNN(trunk_class=ResNet101, rpn_in_ch=2048, feat_stride=32)`.

Every convolution runs on the fp32 MFMA kernel (csrc/conv.hip): 3x3 as is, 1x1 as the KS=1 instantiation, the 7x7 stem
as an explicit im2col + 1x1, a stride-2 1x1 as subsample + 1x1; BN is folded into weights and bias at load
(W' = W * gamma/sqrt(var+eps), b' = beta - mean * gamma/sqrt(var+eps), eps = 2e-5: chainer's default); the bottleneck
tail relu(conv3 + shortcut) is fused into conv3's epilogue.  Parameters keep chainer's link paths
(`conv1/W`, `bn1/gamma|beta|avg_mean|avg_var`, `res3/a/conv1/W`, `res3/b1/bn2/gamma`, ...).
Train-mode BatchNormalization (batch statistics) is not implemented: the trunk is inference-only.
"""
import numpy as np

from ..chainer_compat import unwrap
from ..runtime import default_runtime

BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
STAGES = [("res2", 64, 64, 256, 1), ("res3", 256, 128, 512, 2), ("res4", 512, 256, 1024, 2), ("res5", 1024, 512, 2048, 2)]
BN_EPS = 2e-5


def block_names(n):
    return ["a"] + ["b%d" % i for i in range(1, n)]


def conv_specs(blocks):
    """[(link path of the conv, link path of its BN, cin, cout, ksize)] in execution order."""
    out = [("conv1", "bn1", 3, 64, 7)]
    for (stage, cin, mid, cout, _), n in zip(STAGES, blocks):
        for b in block_names(n):
            i = cin if b == "a" else cout
            p = "%s/%s/" % (stage, b)
            out += [(p + "conv1", p + "bn1", i, mid, 1), (p + "conv2", p + "bn2", mid, mid, 3), (p + "conv3", p + "bn3", mid, cout, 1)]
            if b == "a":
                out.append((p + "conv4", p + "bn4", i, cout, 1))
    return out


class _FoldedConv(object):
    def __init__(self, rt, W, bn, ksize, conv_bias=None):
        """W (co,ci,k,k) [+ an optional convolution bias: chainer's ResNetLayers creates conv1 WITH one] + its BN statistics ->
        packed (ci*k*k [padded], co) weights and a (co,) bias on device:  bn(conv(x) + b) = s*conv(x) + beta + (b - mean)*s."""
        gamma, beta, mean, var = [np.asarray(v, dtype=np.float64) for v in bn]
        s = gamma / np.sqrt(var + BN_EPS)
        if conv_bias is not None:
            mean = mean - np.asarray(conv_bias, dtype=np.float64)
        Wf = (np.asarray(W, dtype=np.float64) * s[:, None, None, None]).astype(np.float32)
        co = Wf.shape[0]
        packed = np.ascontiguousarray(Wf.reshape(co, -1).T)                 # (ci*k*k, co): the kernels' layout
        if ksize == 7:                                                      # im2col rows padded to a multiple of 8
            kp = (packed.shape[0] + 7) // 8 * 8
            packed = np.concatenate([packed, np.zeros((kp - packed.shape[0], co), np.float32)], 0)
        self.Wp = rt.mem.from_numpy(packed)
        self.b = rt.mem.from_numpy((beta - mean * s).astype(np.float32))
        self.ksize = ksize


class ResNet(object):
    def __init__(self, n_layers=101, runtime=None, blocks=None):
        self.rt = runtime or default_runtime()
        self.blocks = tuple(blocks) if blocks is not None else BLOCKS[n_layers]
        self.train = False
        self.convs = {}

    def load_params(self, params, prefix="trunk/"):
        for conv, bn, ci, co, k in conv_specs(self.blocks):
            W = params[prefix + conv + "/W"]
            assert tuple(W.shape) == (co, ci, k, k), (conv, tuple(W.shape))
            stats = [params[prefix + bn + "/" + n] for n in ("gamma", "beta", "avg_mean", "avg_var")]
            self.convs[conv] = _FoldedConv(self.rt, W, stats, k, conv_bias=params.get(prefix + conv + "/b"))

    def _conv(self, name, x, act=1, residual=None):
        c = self.convs[name]
        return self.rt.conv_ex(x, c.Wp, c.b, 1 if c.ksize == 7 else c.ksize, act=act, mask=residual)

    def __call__(self, x, timer=None):
        if self.train:
            raise NotImplementedError("train-mode BatchNormalization (batch statistics) is not part of this path")
        rt = self.rt
        h = rt.asarray(unwrap(x), "f32")
        assert h.ndim == 4 and int(h.shape[0]) == 1, "batch size 1 (models/faster_rcnn.py:77)"
        h = self._conv("conv1", rt.im2col7x7s2(h, int(self.convs["conv1"].Wp.shape[0])))      # conv1 + bn1 + relu
        h = rt.maxpool3x3s2(h)
        for (stage, _, _, _, stride), n in zip(STAGES, self.blocks):
            for b in block_names(n):
                p = "%s/%s/" % (stage, b)
                xin = rt.subsample2(h) if (b == "a" and stride == 2) else h                # stride sits on the first 1x1 (and the shortcut)
                shortcut = self._conv(p + "conv4", xin, act=0) if b == "a" else h
                t = self._conv(p + "conv1", xin)
                t = self._conv(p + "conv2", t)
                h = self._conv(p + "conv3", t, act=3, residual=shortcut)                   # relu(bn3(conv3) + shortcut)
            if timer:
                timer.mark(stage)
        return h


def ResNet50(runtime=None, **kw):
    return ResNet(50, runtime=runtime, **kw)


def ResNet101(runtime=None, **kw):
    return ResNet(101, runtime=runtime, **kw)


def ResNet152(runtime=None, **kw):
    return ResNet(152, runtime=runtime, **kw)
