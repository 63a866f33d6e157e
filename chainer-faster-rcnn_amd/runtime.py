"""Runtime = the C-ABI library + a device-memory provider + the thin typed wrappers over each entry point.

PyTorch-ROCm is used for device memory and streams only (torch.empty on cuda, data_ptr, current stream);
all arithmetic happens inside libfrcnn_hip.so.
"""
import ctypes

import numpy as np

from . import _lib
from . import tuning as _tuning

_NP = {"f32": np.float32, "i32": np.int32, "u8": np.uint8, "f64": np.float64, "i64": np.int64, "i16": np.int16}


class TorchDeviceMemory(object):
    """Device arrays are torch tensors on an MI355X; the stream is torch's current HIP stream."""

    def __init__(self, device="cuda:0"):
        import torch
        if not torch.cuda.is_available():
            raise _lib.FrcnnError("no HIP device visible to PyTorch: the MI355X path cannot run (no CPU fallback)")
        self.torch = torch
        self.device = torch.device(device)
        self._dt = {"f32": torch.float32, "i32": torch.int32, "u8": torch.uint8, "f64": torch.float64,
                    "i64": torch.int64, "i16": torch.int16}

    def empty(self, shape, dtype="f32"):
        return self.torch.empty(shape, dtype=self._dt[dtype], device=self.device)

    def zeros(self, shape, dtype="f32"):
        return self.torch.zeros(shape, dtype=self._dt[dtype], device=self.device)

    def from_numpy(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def from_numpy_async(self, a):
        """Host -> device through a pinned staging buffer, enqueued on the current stream WITHOUT blocking the host (a pageable copy waits for
        everything the stream still has queued: RCNNTrainer's targets would stall the host behind the head's forward pass).  The pinned block
        returns to torch's host allocator only after the copy has run (it records the stream)."""
        return self.torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(self.device, non_blocking=True)

    def to_numpy(self, t):
        return t.detach().cpu().numpy()

    def to_numpy_many(self, arrays):
        """Several small device arrays through ONE device -> host copy (one host round trip instead of one per array): their bytes are
        concatenated on the device; pass the widest element type first so the host views stay aligned."""
        return self.to_numpy_many_async(arrays)()

    def to_numpy_many_async(self, arrays):
        """The same copy, enqueued now on the current stream; returns a function that waits for THAT COPY (an event behind it, not the stream: work
        enqueued afterwards keeps running) and hands out the NumPy arrays.  Every call stages through its own pinned block (torch's host allocator
        recycles them), so several copies may be outstanding."""
        torch = self.torch
        dev = torch.cat([self.contiguous(a).reshape(-1).view(torch.uint8) for a in arrays])
        nbytes = int(dev.numel())
        pin = torch.empty((nbytes,), dtype=torch.uint8, pin_memory=True)      # pinned: a pageable destination goes through a bounce copy
        pin.copy_(dev, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        shapes = [(tuple(a.shape), a.numel() * a.element_size(), _NP[self.dtype_of(a)]) for a in arrays]

        def collect():
            done.synchronize()
            host = pin.numpy().copy()
            out, off = [], 0
            for shape, nb, dt in shapes:
                out.append(host[off:off + nb].view(dt).reshape(shape))
                off += nb
            return out
        return collect

    def is_array(self, a):
        return isinstance(a, self.torch.Tensor) and a.is_cuda

    def contiguous(self, t):
        return t if t.is_contiguous() else t.contiguous()

    def bitcast(self, t, dtype):
        """The same bytes as another element type of the same size (no copy)."""
        return t.view(self._dt[dtype])

    def astype(self, t, dtype):
        """A contiguous copy in another element type (a device-side cast on the current stream)."""
        return t.to(self._dt[dtype]).contiguous()

    def view(self, flat, offset, shape):
        """A contiguous window of a flat buffer, reshaped (shares memory)."""
        n = int(np.prod(shape))
        return flat[offset:offset + n].view(*shape)

    def within(self, a, flat):
        """True when array `a` is a window of the flat buffer `flat` (the trainers' parameter arenas)."""
        lo = flat.data_ptr()
        return lo <= a.data_ptr() and a.data_ptr() + a.numel() * a.element_size() <= lo + flat.numel() * flat.element_size()

    def ptr(self, t):
        if t is None:
            return None
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError("expected a contiguous device tensor")
        return ctypes.c_void_p(t.data_ptr())

    def stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        self.torch.cuda.current_stream(self.device).synchronize()

    def side_stream(self, *arrays):
        """Context manager: launches inside it go to a second HIP stream that starts after everything already enqueued on the
        current one (the trainers run the discarded ProposalLayer of an RPN step there, under the backward pass).  `arrays` are
        marked as used by that stream so the caching allocator does not hand their memory out early.  join_side_stream() makes the
        current stream wait for it."""
        torch = self.torch
        if getattr(self, "_side", None) is None:
            # the side stream carries work nobody waits for soon (the discarded train-mode ProposalLayer of an RPN step: 1.2 ms of NMS launches
            # under a 7 ms backward pass).  It has the DEFAULT priority, like the main and gradient streams: the device offers nothing below it
            # (`priority_range()` = (0, -1) and PyTorch clamps to at most 0; measured in round 4) -- its workgroups simply share the CUs.
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(torch.cuda.current_stream(self.device))
        for a in arrays:
            a.record_stream(self._side)
        return torch.cuda.stream(self._side)

    def join_side_stream(self):
        if getattr(self, "_side", None) is not None:
            self.torch.cuda.current_stream(self.device).wait_stream(self._side)

    def aux_stream(self, name, *arrays):
        """Context manager: a named second stream that starts after everything already enqueued on the current one (side_stream's
        semantics, any number of them).  The trainers put a layer's weight / bias gradient there so that it runs NEXT TO the layer's
        input-gradient convolution on the current stream instead of in front of it: tails, ramps and the memory-bound helpers (slab
        reductions, bias sums) of one chain fill what the other leaves idle.  join_aux_stream(name) makes the current stream wait."""
        torch = self.torch
        if not hasattr(self, "_aux"):
            self._aux = {}
        st = self._aux.get(name)
        if st is None:
            st = self._aux[name] = torch.cuda.Stream(device=self.device)
        st.wait_stream(torch.cuda.current_stream(self.device))
        for a in arrays:
            if a is not None and hasattr(a, "record_stream"):
                a.record_stream(st)
        return torch.cuda.stream(st)

    def join_aux_stream(self, name):
        st = getattr(self, "_aux", {}).get(name)
        if st is not None:
            self.torch.cuda.current_stream(self.device).wait_stream(st)

    def early_stream(self, *after):
        """Context manager: a stream that does NOT wait for the current one -- for work that depends on nothing already enqueued (the
        trainers' AnchorTargetLayer: ground truth in, labels out, with a host round trip in the middle) so that its host part overlaps
        whatever the GPU is still executing.  join_early_stream(*outputs) makes the current stream wait for it and marks the
        outputs as used there."""
        torch = self.torch
        if getattr(self, "_early", None) is None:
            self._early = torch.cuda.Stream(device=self.device)
        for a in after:
            # a DEVICE input produced on the current stream (ground truth scaled / augmented on the GPU): the early stream waits for
            # what is enqueued there now and the allocator keeps the array's memory until the early stream is done with it
            if hasattr(a, "record_stream"):
                self._early.wait_stream(torch.cuda.current_stream(self.device))
                a.record_stream(self._early)
        return torch.cuda.stream(self._early)

    def join_early_stream(self, *outputs):
        cur = self.torch.cuda.current_stream(self.device)
        cur.wait_stream(self._early)
        for a in outputs:
            if a is not None and hasattr(a, "record_stream"):
                a.record_stream(cur)

    def dtype_of(self, t):
        return {v: k for k, v in self._dt.items()}[t.dtype]


class _HalfLib(object):
    """The library seen through a 16-bit operand format: with half == "f16" every *_bf16* entry point resolves to its *_f16* twin (csrc/conv_f16.hip,
    conv_f16_pair.hip, linear_f16.hip, roi_f16.hip: the same kernel sources compiled with fp16 pack / widen / MFMA; include/frcnn_hip.h).  The
    first-layer entry (frcnn_conv1_bf16) has no fp16 twin: Runtime.conv1_bf16 composes it from the generic kernel instead."""

    def __init__(self, lib, half):
        self._lib, self._half = lib, half

    def __getattr__(self, name):
        if self._half == "f16" and "bf16" in name:
            return getattr(self._lib, name.replace("bf16", "f16"))
        return getattr(self._lib, name)


class Runtime(object):
    def __init__(self, lib, mem, half="bf16"):
        self.lib = lib
        self.mem = mem
        self._ws = {}
        assert half in ("bf16", "f16")
        self.half = half                       # the 16-bit operand format the *_bf16 methods below compute in
        self.hlib = _HalfLib(lib, half)

    def with_half(self, half):
        """The same runtime (library, memory, workspaces) computing its 16-bit chain in `half` ("bf16" or "f16"): the *_bf16 methods of the returned object
        call the *_f16* entry points when half == "f16"; arrays of raw 16-bit values then hold IEEE binary16 bits."""
        if half == self.half:
            return self
        r = Runtime.__new__(type(self))
        r.__dict__.update(self.__dict__)
        r.half, r.hlib = half, _HalfLib(self.lib, half)
        return r

    # ------------------------------------------------------------------ helpers
    def workspace(self, tag, nbytes, init=None):
        """Caller-owned scratch, grown on demand; `init(ws)` runs once per (re)allocation (the conv workspace's counter page)."""
        w = self._ws.get(tag)
        if w is None or w.shape[0] < nbytes:
            w = self.mem.empty((max(int(nbytes), 256),), "u8")
            self._ws[tag] = w
            if init is not None:
                init(w)
        return w

    def _conv_workspace(self, ci, co, H, W):
        m, L = self.mem, self.lib
        return self.workspace("conv3x3", L.frcnn_conv3x3_workspace_bytes(ci, co, H, W),
                              init=lambda w: _lib.check(L.frcnn_conv3x3_workspace_init(m.ptr(w), w.shape[0], m.stream()),
                                                        "frcnn_conv3x3_workspace_init"))

    def asarray(self, a, dtype="f32"):
        """Accept device arrays, NumPy arrays or anything with `.data` (chainer.Variable-like)."""
        if hasattr(a, "data") and not isinstance(a, np.ndarray) and not self.mem.is_array(a):
            a = a.data
        if self.mem.is_array(a):
            if self.mem.dtype_of(a) != dtype:
                raise ValueError("expected dtype %s, got %s" % (dtype, self.mem.dtype_of(a)))
            return self.mem.contiguous(a)
        a = np.asarray(a)
        if a.dtype != _NP[dtype]:
            raise ValueError("expected dtype %s, got %s" % (dtype, a.dtype))
        return self.mem.from_numpy(a)

    # ------------------------------------------------------------------ NMS
    def nms(self, dets, thresh, max_out=0):
        """-> (keep (cap,) i32 device, n_keep (1,) i32 device)"""
        m, L = self.mem, self.lib
        n = int(dets.shape[0])
        cap = min(n, max_out) if max_out > 0 else n
        keep = m.empty((max(cap, 1),), "i32")
        n_keep = m.empty((1,), "i32")
        wsb = L.frcnn_nms_workspace_bytes(n)
        ws = self.workspace("nms", wsb)
        _lib.check(L.frcnn_nms(m.ptr(dets) if n else None, n, float(thresh), int(max_out), m.ptr(keep), m.ptr(n_keep),
                               m.ptr(ws), ws.shape[0], m.stream()), "frcnn_nms")
        return keep, n_keep

    def nms_batched(self, dets, thresh, max_out=0):
        """dets (G,n,5) -> (keep (G,cap) i32, n_keep (G,) i32)"""
        m, L = self.mem, self.lib
        G, n = int(dets.shape[0]), int(dets.shape[1])
        cap = min(n, max_out) if max_out > 0 else n
        keep = m.empty((G, max(cap, 1)), "i32")
        n_keep = m.empty((G,), "i32")
        ws = self.workspace("nmsb", L.frcnn_nms_batched_workspace_bytes(G, n))
        _lib.check(L.frcnn_nms_batched(m.ptr(dets), G, n, float(thresh), int(max_out), m.ptr(keep), m.ptr(n_keep),
                                       m.ptr(ws), ws.shape[0], m.stream()), "frcnn_nms_batched")
        return keep, n_keep

    # ------------------------------------------------------------------ proposals
    def proposals(self, cls_prob, bbox_pred, anchors, feat_stride, im_h, im_w, min_size, pre_nms_top_n,
                  post_nms_top_n, nms_thresh, want_index=False):
        """cls_prob (2A,H,W), bbox_pred (4A,H,W) device f32; anchors (A,4) float64 host.
        -> rois (cap,4), probs (cap,), n_out (1,) [, src_index (cap,)] device arrays."""
        m, L = self.mem, self.lib
        A = int(anchors.shape[0])
        _, H, W = [int(v) for v in bbox_pred.shape]
        n = A * H * W
        m_max = min(n, pre_nms_top_n) if pre_nms_top_n > 0 else n
        cap = min(post_nms_top_n, m_max) if post_nms_top_n > 0 else m_max
        rois = m.empty((cap, 4), "f32")
        probs = m.empty((cap,), "f32")
        n_out = m.empty((1,), "i32")
        src = m.empty((cap,), "i32") if want_index else None
        ws = self.workspace("proposals", L.frcnn_proposals_workspace_bytes(A, H, W, int(pre_nms_top_n)))
        anc = np.ascontiguousarray(anchors, dtype=np.float64)
        _lib.check(L.frcnn_proposals(m.ptr(cls_prob), m.ptr(bbox_pred), A, H, W, anc.ctypes.data_as(ctypes.c_void_p),
                                     int(feat_stride), int(im_h), int(im_w), float(min_size), int(pre_nms_top_n),
                                     int(post_nms_top_n), float(nms_thresh), m.ptr(rois), m.ptr(probs), m.ptr(n_out),
                                     m.ptr(src), m.ptr(ws), ws.shape[0], m.stream()), "frcnn_proposals")
        return (rois, probs, n_out, src) if want_index else (rois, probs, n_out)

    # ------------------------------------------------------------------ RoI pooling
    def roi_pool_fwd(self, x, rois, outh, outw, scale, want_argmax=False):
        m, L = self.mem, self.lib
        C, H, W = [int(v) for v in x.shape[-3:]]
        R = int(rois.shape[0])
        y = m.empty((R, C, outh, outw), "f32")
        am = m.empty((R, C, outh, outw), "i32") if want_argmax else None
        ws = self.workspace("roi_pool", L.frcnn_roi_pool_workspace_bytes(C, H, W))
        _lib.check(L.frcnn_roi_pool_fwd(m.ptr(x), C, H, W, m.ptr(rois), R, outh, outw, float(scale), m.ptr(y),
                                        m.ptr(am), m.ptr(ws), ws.shape[0], m.stream()), "frcnn_roi_pool_fwd")
        return (y, am) if want_argmax else y

    def roi_pool_fwd_chw(self, x, rois, outh, outw, scale, want_argmax=False, out=None, out_argmax=None):
        """x (.., C, H, W) NCHW; rois (R,5) [batch,x1,y1,x2,y2] or (R,4) [x1,y1,x2,y2] (ProposalLayer's output)."""
        m, L = self.mem, self.lib
        C, H, W = [int(v) for v in x.shape[-3:]]
        R = int(rois.shape[0])
        y = out if out is not None else m.empty((R, C, outh, outw), "f32")
        am = (out_argmax if out_argmax is not None else m.empty((R, C, outh, outw), "i32")) if want_argmax else None
        ws = self.workspace("roi_pool", L.frcnn_roi_pool_workspace_bytes(C, H, W))
        _lib.check(L.frcnn_roi_pool_fwd_chw(m.ptr(x), C, H, W, m.ptr(rois), R, int(rois.shape[1]), outh, outw, float(scale),
                                            m.ptr(y), m.ptr(am), m.ptr(ws), ws.shape[0], m.stream()), "frcnn_roi_pool_fwd_chw")
        return (y, am) if want_argmax else y

    def roi_pool_fwd_chw_bf16(self, x, rois, outh, outw, scale):
        """The same pooling with the result written as raw bf16 bits, flattened (R, C*outh*outw): the bf16 FC head's input."""
        m, L = self.mem, self.hlib
        C, H, W = [int(v) for v in x.shape[-3:]]
        R = int(rois.shape[0])
        y = m.empty((R, C * outh * outw), "i16")
        _lib.check(L.frcnn_roi_pool_fwd_chw_bf16(m.ptr(x), C, H, W, m.ptr(rois), R, int(rois.shape[1]), outh, outw, float(scale),
                                                 m.ptr(y), m.stream()), "frcnn_roi_pool_fwd_chw_bf16")
        return y

    def roi_pool_fwd_blk_bf16(self, x_blk, C, rois, outh, outw, scale, out_bf16=False):
        """RoI pooling straight from the bf16 chain's channel-blocked map [CP/16][H][W][16] -> (R, C, outh, outw) fp32, or raw bf16
        bits (R, C*outh*outw) for the bf16 FC head.  Maps up to 76 x 64 (the cell-major kernel's LDS image)."""
        m, L = self.mem, self.hlib
        H, W = int(x_blk.shape[1]), int(x_blk.shape[2])
        R = int(rois.shape[0])
        y = m.empty((R, int(C) * outh * outw), "i16") if out_bf16 else m.empty((R, int(C), outh, outw), "f32")
        rc = L.frcnn_roi_pool_fwd_blk_bf16(m.ptr(x_blk), int(C), H, W, m.ptr(rois), R, int(rois.shape[1]), outh, outw, float(scale),
                                           m.ptr(y), int(bool(out_bf16)), m.stream())
        if rc == _lib.ERR_UNSUPPORTED:
            # the cell-major kernel declined (a map beyond its LDS image, or FRCNN_ROI_KERNEL=planes): the documented detour of
            # include/frcnn_hip.h -- the map as fp32 NCHW, the fp32 pooling (which has its own fallbacks), one rounding-free conversion
            # back (a maximum of bf16 values is a bf16 value) -- same device kernels, same results, no CPU path
            y32 = self.roi_pool_fwd_chw(self.bf16_to_nchw(x_blk, int(C)), rois, outh, outw, scale)
            return self.to_bf16(y32.reshape(R, -1)) if out_bf16 else y32
        _lib.check(rc, "frcnn_roi_pool_fwd_blk_bf16")
        return y

    def roi_pool_fwd_chw_f32s(self, x, rois, outh, outw, scale):
        """The same pooling with the fp32 maxima written as their three bf16 terms, (3, R, C*outh*outw): the split FC head's input."""
        m, L = self.mem, self.lib
        C, H, W = [int(v) for v in x.shape[-3:]]
        R = int(rois.shape[0])
        y = m.empty((3, R, C * outh * outw), "i16")
        rc = L.frcnn_roi_pool_fwd_chw_f32s(m.ptr(x), C, H, W, m.ptr(rois), R, int(rois.shape[1]), outh, outw, float(scale), m.ptr(y), m.stream())
        if rc == _lib.ERR_UNSUPPORTED:
            # the cell-major kernel declined (see roi_pool_fwd_blk_bf16): pool in fp32, then split -- the same three terms (the split is exact)
            return self.f32s_split(self.roi_pool_fwd_chw(x, rois, outh, outw, scale).reshape(R, -1))
        _lib.check(rc, "frcnn_roi_pool_fwd_chw_f32s")
        return y

    def chw_to_hwc(self, x):
        m, L = self.mem, self.lib
        C, H, W = [int(v) for v in x.shape[-3:]]
        xt = m.empty((H * W, C), "f32")
        _lib.check(L.frcnn_chw_to_hwc(m.ptr(x), C, H, W, m.ptr(xt), m.stream()), "frcnn_chw_to_hwc")
        return xt

    def roi_pool_fwd_hwc(self, xt, C, H, W, rois, outh, outw, scale, want_argmax=False, out=None):
        m, L = self.mem, self.lib
        R = int(rois.shape[0])
        y = out if out is not None else m.empty((R, C, outh, outw), "f32")
        am = m.empty((R, C, outh, outw), "i32") if want_argmax else None
        _lib.check(L.frcnn_roi_pool_fwd_hwc(m.ptr(xt), C, H, W, m.ptr(rois), R, int(rois.shape[1]), outh, outw,
                                            float(scale), m.ptr(y), m.ptr(am), m.stream()), "frcnn_roi_pool_fwd_hwc")
        return (y, am) if want_argmax else y

    def roi_pool_bwd(self, dy, argmax, C, H, W, out=None):
        m, L = self.mem, self.lib
        R, _, outh, outw = [int(v) for v in dy.shape]
        dx = out if out is not None else m.empty((1, C, H, W), "f32")
        _lib.check(L.frcnn_roi_pool_bwd(m.ptr(dy), m.ptr(argmax), R, C, H, W, outh, outw, m.ptr(dx), m.stream()),
                   "frcnn_roi_pool_bwd")
        return dx

    # ------------------------------------------------------------------ convolution stack
    def pack_conv3x3_w(self, w):
        m, L = self.mem, self.lib
        co, ci = int(w.shape[0]), int(w.shape[1])
        wp = m.empty((ci * 9, co), "f32")
        _lib.check(L.frcnn_pack_conv3x3_w(m.ptr(w), co, ci, m.ptr(wp), m.stream()), "frcnn_pack_conv3x3_w")
        return wp

    def conv3x3(self, x, w_packed, bias, relu=True, out=None, cfg=-1):
        m, L = self.mem, self.lib
        ci, H, W = [int(v) for v in x.shape[-3:]]
        co = int(w_packed.shape[1])
        assert int(w_packed.shape[0]) == ci * 9
        y = out if out is not None else m.empty((1, co, H, W), "f32")
        ws = self._conv_workspace(ci, co, H, W)
        _lib.check(L.frcnn_conv3x3_f32_cfg(m.ptr(x), m.ptr(w_packed), m.ptr(bias), m.ptr(y), ci, co, H, W,
                                           int(bool(relu)), int(cfg), m.ptr(ws), ws.shape[0], m.stream()),
                   "frcnn_conv3x3_f32")
        return y

    def maxpool2x2(self, x, out=None):
        m, L = self.mem, self.lib
        C, H, W = [int(v) for v in x.shape[-3:]]
        y = out if out is not None else m.empty((1, C, (H + 1) // 2, (W + 1) // 2), "f32")
        _lib.check(L.frcnn_maxpool2x2_f32(m.ptr(x), m.ptr(y), C, H, W, m.stream()), "frcnn_maxpool2x2_f32")
        return y

    def rpn_heads_pack(self, w_cls, b_cls, w_bbox, b_bbox):
        """Chainer-layout head weights -> (w_packed (Cmid,NP), b_packed (NP,), A); once, at load time."""
        m, L = self.mem, self.lib
        A, Cmid = int(w_cls.shape[0]) // 2, int(w_cls.shape[1])
        NP = L.frcnn_rpn_heads_padded_channels(A)
        wp, bp = m.empty((Cmid, NP), "f32"), m.empty((NP,), "f32")
        _lib.check(L.frcnn_rpn_heads_pack(m.ptr(w_cls), m.ptr(b_cls), m.ptr(w_bbox), m.ptr(b_bbox), Cmid, A, m.ptr(wp),
                                          m.ptr(bp), m.stream()), "frcnn_rpn_heads_pack")
        return wp, bp, A

    def rpn_heads(self, h, packed):
        """-> (rpn_cls_score (1,2A,H,W), rpn_cls_prob (1,2A,H,W), rpn_bbox_pred (1,4A,H,W)); score and bbox are
        views of one (NP,H,W) buffer."""
        m, L = self.mem, self.lib
        wp, bp, A = packed
        C, H, W = [int(v) for v in h.shape[-3:]]
        raw = m.empty((int(wp.shape[1]), H, W), "f32")
        prob = m.empty((1, 2 * A, H, W), "f32")
        _lib.check(L.frcnn_rpn_heads_f32(m.ptr(h), C, H, W, A, m.ptr(wp), m.ptr(bp), m.ptr(raw), m.ptr(prob), m.stream()),
                   "frcnn_rpn_heads_f32")
        return raw[:2 * A].reshape(1, 2 * A, H, W), prob, raw[2 * A:6 * A].reshape(1, 4 * A, H, W)

    # ------------------------------------------------------------------ head
    def linear(self, x, w, bias, relu=False, out=None):
        """bias None: no bias term (include/frcnn_hip.h)."""
        m, L = self.mem, self.lib
        M, K = int(x.shape[0]), int(np.prod(x.shape[1:]))
        N = int(w.shape[0])
        assert int(w.shape[1]) == K
        y = out if out is not None else m.empty((M, N), "f32")
        ws = self.workspace("linear", L.frcnn_linear_workspace_bytes(M, N, K))
        _lib.check(L.frcnn_linear_f32(m.ptr(x), m.ptr(w), m.ptr(bias), m.ptr(y), M, N, K, int(bool(relu)), m.ptr(ws),
                                      ws.shape[0], m.stream()), "frcnn_linear_f32")
        return y

    def head_decode(self, boxes, deltas, cls_score, im_h, im_w):
        m, L = self.mem, self.lib
        R, ncls = int(cls_score.shape[0]), int(cls_score.shape[1])
        pred = m.empty((R, 4 * ncls), "f32")
        prob = m.empty((R, ncls), "f32")
        _lib.check(L.frcnn_head_decode(m.ptr(boxes), m.ptr(deltas), m.ptr(cls_score), R, ncls, int(im_h), int(im_w),
                                       m.ptr(pred), m.ptr(prob), m.stream()), "frcnn_head_decode")
        return pred, prob


    def head_decode_stacked(self, boxes, out, ncls, dcol, im_h, im_w):
        """`out` (R, ld): class scores in columns [0, ncls), deltas from column dcol -- the stacked cls_score/bbox_pred GEMM."""
        m, L = self.mem, self.lib
        R, ld = int(out.shape[0]), int(out.shape[1])
        pred = m.empty((R, 4 * ncls), "f32")
        prob = m.empty((R, ncls), "f32")
        _lib.check(L.frcnn_head_decode_stacked(m.ptr(boxes), m.ptr(out), ld, int(dcol), R, int(ncls), int(im_h), int(im_w),
                                               m.ptr(pred), m.ptr(prob), m.stream()), "frcnn_head_decode_stacked")
        return pred, prob

    def preprocess_u8(self, img, means, im_scale, out_hw, out=None):
        """img (H,W,C) uint8 device array -> (1,C,OH,OW) float32 (forward.py:33-45 on the device); `out`: an existing array of that shape (a captured
        graph's input buffer) to write into."""
        m, L = self.mem, self.lib
        H, W, C = [int(v) for v in img.shape]
        OH, OW = int(out_hw[0]), int(out_hw[1])
        if out is None:
            out = m.empty((1, C, OH, OW), "f32")
        assert tuple(int(v) for v in out.shape) == (1, C, OH, OW)
        mh = np.ascontiguousarray(means, dtype=np.float64).ravel()
        _lib.check(L.frcnn_preprocess_u8(m.ptr(img), H, W, C, mh.ctypes.data_as(ctypes.c_void_p), float(im_scale), OH, OW, m.ptr(out),
                                         m.stream()), "frcnn_preprocess_u8")
        return out

    def class_dets(self, cls_prob, pred_boxes):
        """(R,ncls), (R,4*ncls) -> (ncls-1, R, 5) per-class [x1,y1,x2,y2,score] rows (forward.py:50-53)."""
        m, L = self.mem, self.lib
        R, ncls = int(cls_prob.shape[0]), int(cls_prob.shape[1])
        dets = m.empty((ncls - 1, max(R, 1), 5), "f32")
        _lib.check(L.frcnn_class_dets(m.ptr(cls_prob), m.ptr(pred_boxes), R, ncls, m.ptr(dets), m.stream()), "frcnn_class_dets")
        return dets[:, :R]

    def bbox_transform_inv(self, boxes, deltas):
        m, L = self.mem, self.lib
        R, c4 = int(deltas.shape[0]), int(deltas.shape[1])
        pred = m.empty((R, c4), "f32")
        _lib.check(L.frcnn_bbox_transform_inv(m.ptr(boxes), m.ptr(deltas), R, c4 // 4, m.ptr(pred), m.stream()),
                   "frcnn_bbox_transform_inv")
        return pred

    def clip_boxes_(self, boxes, im_h, im_w):
        m, L = self.mem, self.lib
        n = int(np.prod(boxes.shape)) // 4
        _lib.check(L.frcnn_clip_boxes(m.ptr(boxes), n, int(im_h), int(im_w), m.stream()), "frcnn_clip_boxes")
        return boxes

    def softmax_rows(self, scores):
        m, L = self.mem, self.lib
        R, n = int(scores.shape[0]), int(scores.shape[1])
        out = m.empty((R, n), "f32")
        _lib.check(L.frcnn_softmax_rows(m.ptr(scores), R, n, m.ptr(out), m.stream()), "frcnn_softmax_rows")
        return out


    # ------------------------------------------------------------------ training step (csrc/train.hip)
    def conv_ex(self, x, w_packed, bias, ksize=3, act=1, mask=None, out=None):
        """Generic conv entry: act 0 none / 1 ReLU / 2 masked by (mask > 0) (the input-gradient convolution)."""
        m, L = self.mem, self.lib
        ci, H, W = [int(v) for v in x.shape[-3:]]
        co = int(w_packed.shape[1])
        assert int(w_packed.shape[0]) == ci * ksize * ksize
        oh, ow = ((H + 1) // 2, (W + 1) // 2) if act == 4 else (H, W)      # act 4: ReLU + 2x2 max-pool fused
        y = out if out is not None else m.empty((1, co, oh, ow), "f32")
        ws = self._conv_workspace(ci, co, H, W)
        _lib.check(L.frcnn_conv_f32_ex(m.ptr(x), m.ptr(w_packed), m.ptr(bias), m.ptr(mask), m.ptr(y), ci, co, H, W, int(ksize),
                                       int(act), m.ptr(ws), ws.shape[0], m.stream()), "frcnn_conv_f32_ex")
        return y

    # ------------------------------------------------------------------ fp32 convolution on split (3 x bf16) tensors (csrc/conv_f32s.hip)
    def f32s_pack_conv_w(self, w):
        m, L = self.mem, self.lib
        co, ci = int(w.shape[0]), int(w.shape[1])
        wp = m.empty((3, self.bf16_pad(ci) // 16, 9, self.bf16_pad(co), 16), "i16")
        _lib.check(L.frcnn_f32s_pack_conv_w(m.ptr(w), co, ci, m.ptr(wp), m.stream()), "frcnn_f32s_pack_conv_w")
        return wp

    def f32s_from_nchw(self, x):
        m, L = self.mem, self.lib
        C, H, W = [int(v) for v in x.shape[-3:]]
        y = m.empty((3, self.bf16_pad(C) // 16, H, W, 16), "i16")
        _lib.check(L.frcnn_f32s_from_nchw_f32(m.ptr(x), C, H, W, m.ptr(y), m.stream()), "frcnn_f32s_from_nchw_f32")
        return y

    def f32s_to_nchw(self, x, C):
        m, L = self.mem, self.lib
        H, W = int(x.shape[2]), int(x.shape[3])
        y = m.empty((1, int(C), H, W), "f32")
        _lib.check(L.frcnn_f32s_to_nchw_f32(m.ptr(x), int(C), H, W, m.ptr(y), m.stream()), "frcnn_f32s_to_nchw_f32")
        return y

    def f32s_split(self, x):
        """fp32 array -> its three bf16 terms, shape (3,) + x.shape (raw bits in an int16 array); h + m + l == x exactly."""
        m, L = self.mem, self.lib
        y = m.empty((3,) + tuple(int(v) for v in x.shape), "i16")
        _lib.check(L.frcnn_f32s_split(m.ptr(x), int(np.prod(x.shape)), m.ptr(y), m.stream()), "frcnn_f32s_split")
        return y

    def f32s_join(self, parts):
        m, L = self.mem, self.lib
        y = m.empty(tuple(int(v) for v in parts.shape[1:]), "f32")
        _lib.check(L.frcnn_f32s_join(m.ptr(parts), int(np.prod(parts.shape[1:])), m.ptr(y), m.stream()), "frcnn_f32s_join")
        return y

    def linear_f32s(self, x_parts, w_parts, bias, relu=False, out_split=False):
        """x (3,M,K), w (3,N,K) split tensors, bias (N,) fp32 -> (M,N) fp32, or its parts (3,M,N) when out_split."""
        m, L = self.mem, self.lib
        M, K = int(x_parts.shape[1]), int(x_parts.shape[2])
        N = int(w_parts.shape[1])
        assert int(w_parts.shape[2]) == K and int(x_parts.shape[0]) == 3 and int(w_parts.shape[0]) == 3
        y = m.empty((3, M, N), "i16") if out_split else m.empty((M, N), "f32")
        ws = self.workspace("linear", L.frcnn_linear_f32s_workspace_bytes(M, N, K))
        _lib.check(L.frcnn_linear_f32s(m.ptr(x_parts), m.ptr(w_parts), m.ptr(bias), m.ptr(y), M, N, K, int(bool(relu)), int(bool(out_split)),
                                       m.ptr(ws), ws.shape[0], m.stream()), "frcnn_linear_f32s")
        return y

    def conv1_f32s(self, x, w, bias, relu=True):
        """First layer: x (1,Cin<=3,H,W) fp32 NCHW, w (Cout<=64,Cin,3,3) fp32 -> split tensor [3][CoutP/16][H][W][16]."""
        m, L = self.mem, self.lib
        cin, H, W = [int(v) for v in x.shape[-3:]]
        cout = int(w.shape[0])
        y = m.empty((3, self.bf16_pad(cout) // 16, H, W, 16), "i16")
        _lib.check(L.frcnn_conv1_f32s(m.ptr(x), m.ptr(w), m.ptr(bias), m.ptr(y), cin, cout, H, W, int(bool(relu)), m.stream()), "frcnn_conv1_f32s")
        return y

    def f32s_pack_from_packed(self, w_packed_f32, cin, cout, dgrad=False, out=None):
        """The trainer's packed fp32 weights (cin*9, cout) -> split weights of the forward / input-gradient convolution."""
        m, L = self.mem, self.lib
        ki, ko = (cout, cin) if dgrad else (cin, cout)
        wp = out if out is not None else m.empty((3, self.bf16_pad(ki) // 16, 9, self.bf16_pad(ko), 16), "i16")
        _lib.check(L.frcnn_f32s_pack_from_packed(m.ptr(w_packed_f32), int(cin), int(cout), int(bool(dgrad)), m.ptr(wp), m.stream()),
                   "frcnn_f32s_pack_from_packed")
        return wp

    def f32s_pack_many(self, layers):
        """layers: [(packed fp32 weights (cin*9, cout), split fwd weights, split dgrad weights or None, cin, cout)], at most 16: ONE launch."""
        import ctypes

        class Desc(ctypes.Structure):
            _fields_ = [("w", ctypes.c_void_p), ("fwd", ctypes.c_void_p), ("dgr", ctypes.c_void_p), ("cin", ctypes.c_int), ("cout", ctypes.c_int)]
        m, L = self.mem, self.lib
        val = lambda t: None if t is None else m.ptr(t).value
        arr = (Desc * len(layers))(*[Desc(val(w), val(f), val(d), int(ci), int(co)) for (w, f, d, ci, co) in layers])
        _lib.check(L.frcnn_f32s_pack_many(ctypes.cast(arr, ctypes.c_void_p), len(layers), m.stream()), "frcnn_f32s_pack_many")

    def conv3x3_f32s_train(self, x, w_packed, bias, cin, cout, relu=True, want_split=True, want_nchw=True, mask=None):
        """Training forms of the split convolution: returns (split tensor or None, fp32 NCHW or None); mask (1,Cout,H,W) fp32."""
        m, L = self.mem, self.lib
        H, W = int(x.shape[2]), int(x.shape[3])
        assert int(x.shape[0]) == 3 and int(x.shape[1]) * 16 == self.bf16_pad(cin) and int(w_packed.shape[1]) * 16 == self.bf16_pad(cin)
        ys = m.empty((3, self.bf16_pad(cout) // 16, H, W, 16), "i16") if want_split else None
        yn = m.empty((1, int(cout), H, W), "f32") if want_nchw else None
        ws = self.workspace("conv_f32s", L.frcnn_conv_f32s_workspace_bytes(int(cin), int(cout), H, W),
                            init=lambda w: _lib.check(L.frcnn_conv_f32s_workspace_init(m.ptr(w), w.shape[0], m.stream()),
                                                      "frcnn_conv_f32s_workspace_init"))
        _lib.check(L.frcnn_conv3x3_f32s_train(m.ptr(x), m.ptr(w_packed), m.ptr(bias), m.ptr(ys), m.ptr(yn), m.ptr(mask), int(cin), int(cout), H, W,
                                              int(bool(relu)), m.ptr(ws), ws.shape[0], m.stream()), "frcnn_conv3x3_f32s_train")
        return ys, yn

    def conv1_f32s_train(self, x, w_packed_f32, bias, cout, relu=True):
        """First layer, training form: packed fp32 weights (cin*9, cout) -> (split tensor, fp32 NCHW)."""
        m, L = self.mem, self.lib
        cin, H, W = [int(v) for v in x.shape[-3:]]
        ys = m.empty((3, self.bf16_pad(cout) // 16, H, W, 16), "i16")
        yn = m.empty((1, int(cout), H, W), "f32")
        _lib.check(L.frcnn_conv1_f32s_train(m.ptr(x), m.ptr(w_packed_f32), m.ptr(bias), m.ptr(ys), m.ptr(yn), cin, int(cout), H, W, int(bool(relu)),
                                            m.stream()), "frcnn_conv1_f32s_train")
        return ys, yn

    def conv1_bf16(self, x, w, bias, relu=True):
        """First layer of the bf16 chain: x (1,Cin<=3,H,W) fp32 NCHW, w (Cout<=64,Cin,3,3) fp32 -> [CoutP/16][H][W][16] bf16."""
        if self.half == "f16":                 # no fp16 twin of the first-layer kernel (csrc/conv_f32s.hip): the generic convolution on the blocked image, Cin padded to 16
            return self.conv_bf16(self.bf16_from_nchw(x), self.bf16_pack_conv_w(w, 3), bias, int(x.shape[-3]), int(w.shape[0]), 3, relu=relu)
        m, L = self.mem, self.lib
        cin, H, W = [int(v) for v in x.shape[-3:]]
        cout = int(w.shape[0])
        y = m.empty((self.bf16_pad(cout) // 16, H, W, 16), "i16")
        _lib.check(L.frcnn_conv1_bf16(m.ptr(x), m.ptr(w), m.ptr(bias), m.ptr(y), cin, cout, H, W, int(bool(relu)), m.stream()), "frcnn_conv1_bf16")
        return y

    def conv1_pair_bf16(self, x, w1, b1, w2_packed, b2):
        """conv1_1 + ReLU + conv1_2 + ReLU + 2x2 max-pool of the bf16 chain in one launch: x (1,Cin<=3,H,W) fp32 NCHW, w1 (64,Cin,3,3) fp32,
        w2_packed [4][9][64][16] bf16 -> [4][ceil(H/2)][ceil(W/2)][16] bf16."""
        m, L = self.mem, self.hlib
        cin, H, W = [int(v) for v in x.shape[-3:]]
        assert int(w1.shape[0]) == 64 and int(w1.shape[1]) == cin and tuple(int(v) for v in w2_packed.shape) == (4, 9, 64, 16)
        y = m.empty((4, (H + 1) // 2, (W + 1) // 2, 16), "i16")
        _lib.check(L.frcnn_conv1_pair_bf16(m.ptr(x), m.ptr(w1), m.ptr(b1), m.ptr(w2_packed), m.ptr(b2), m.ptr(y), cin, H, W, m.stream()),
                   "frcnn_conv1_pair_bf16")
        return y

    def conv3x3_f32s(self, x, w_packed, bias, cin, cout, relu=True, out_f32_nchw=False, pool=False):
        """x split tensor [3][CinP/16][H][W][16] -> split tensor (optionally ReLU + 2x2 max-pooled), or (1,Cout,H,W) fp32."""
        m, L = self.mem, self.lib
        H, W = int(x.shape[2]), int(x.shape[3])
        assert int(x.shape[0]) == 3 and int(x.shape[1]) * 16 == self.bf16_pad(cin) and int(w_packed.shape[1]) * 16 == self.bf16_pad(cin)
        if pool:
            y = m.empty((3, self.bf16_pad(cout) // 16, (H + 1) // 2, (W + 1) // 2, 16), "i16")
        else:
            y = m.empty((1, int(cout), H, W), "f32") if out_f32_nchw else m.empty((3, self.bf16_pad(cout) // 16, H, W, 16), "i16")
        mode = 2 if pool else (1 if out_f32_nchw else 0)
        ws = self.workspace("conv_f32s", L.frcnn_conv_f32s_workspace_bytes(int(cin), int(cout), H, W),
                            init=lambda w: _lib.check(L.frcnn_conv_f32s_workspace_init(m.ptr(w), w.shape[0], m.stream()),
                                                      "frcnn_conv_f32s_workspace_init"))
        _lib.check(L.frcnn_conv3x3_f32s_ws(m.ptr(x), m.ptr(w_packed), m.ptr(bias), m.ptr(y), int(cin), int(cout), H, W, int(bool(relu)), mode,
                                           m.ptr(ws), ws.shape[0], m.stream()), "frcnn_conv3x3_f32s_ws")
        return y

    # ------------------------------------------------------------------ bf16 convolution stack (raw bf16 bits live in int16 arrays)
    def bf16_pad(self, c):
        return (int(c) + 15) // 16 * 16

    def bf16_pack_conv_w(self, w, ksize=3):
        m, L = self.mem, self.hlib
        co, ci = int(w.shape[0]), int(w.shape[1])
        wp = m.empty((self.bf16_pad(ci) // 16, ksize * ksize, self.bf16_pad(co), 16), "i16")
        _lib.check(L.frcnn_bf16_pack_conv_w(m.ptr(w), co, ci, int(ksize), m.ptr(wp), m.stream()), "frcnn_bf16_pack_conv_w")
        return wp

    def bf16_from_nchw(self, x):
        m, L = self.mem, self.hlib
        C, H, W = [int(v) for v in x.shape[-3:]]
        y = m.empty((self.bf16_pad(C) // 16, H, W, 16), "i16")          # channel-blocked: [C/16][H][W][16]
        _lib.check(L.frcnn_bf16_from_nchw_f32(m.ptr(x), C, H, W, m.ptr(y), m.stream()), "frcnn_bf16_from_nchw_f32")
        return y

    def bf16_to_nchw(self, x, C):
        m, L = self.mem, self.hlib
        H, W = int(x.shape[1]), int(x.shape[2])
        y = m.empty((1, int(C), H, W), "f32")
        _lib.check(L.frcnn_bf16_to_nchw_f32(m.ptr(x), int(C), H, W, m.ptr(y), m.stream()), "frcnn_bf16_to_nchw_f32")
        return y

    def conv_bf16(self, x, w_packed, bias, cin, cout, ksize=3, relu=True, out_f32_nchw=False, pool=False):
        """x [CinP/16][H][W][16] bf16 -> [CoutP/16][H][W][16] bf16, or (1,Cout,H,W) fp32 when out_f32_nchw."""
        m, L = self.mem, self.hlib
        H, W = int(x.shape[1]), int(x.shape[2])
        assert int(x.shape[0]) * 16 == self.bf16_pad(cin) and int(w_packed.shape[0]) * 16 == self.bf16_pad(cin)
        if pool:                                         # ReLU + 2x2 ceil-mode max-pool fused into the epilogue (out_mode 2)
            y = m.empty((self.bf16_pad(cout) // 16, (H + 1) // 2, (W + 1) // 2, 16), "i16")
        else:
            y = m.empty((1, int(cout), H, W), "f32") if out_f32_nchw else m.empty((self.bf16_pad(cout) // 16, H, W, 16), "i16")
        mode = 2 if pool else int(bool(out_f32_nchw))
        ws = self.workspace("conv_bf16", L.frcnn_conv_bf16_workspace_bytes(int(cin), int(cout), H, W),
                            init=lambda w: _lib.check(L.frcnn_conv_bf16_workspace_init(m.ptr(w), w.shape[0], m.stream()),
                                                      "frcnn_conv_bf16_workspace_init"))
        _lib.check(L.frcnn_conv_bf16_ws(m.ptr(x), m.ptr(w_packed), m.ptr(bias), m.ptr(y), int(cin), int(cout), H, W, int(ksize),
                                        int(bool(relu)), mode, m.ptr(ws), ws.shape[0], m.stream()), "frcnn_conv_bf16_ws")
        return y

    def maxpool2x2_bf16(self, x):
        m, L = self.mem, self.hlib
        CB, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        C = CB * 16
        y = m.empty((CB, (H + 1) // 2, (W + 1) // 2, 16), "i16")
        _lib.check(L.frcnn_maxpool2x2_bf16(m.ptr(x), m.ptr(y), C, H, W, m.stream()), "frcnn_maxpool2x2_bf16")
        return y

    def to_bf16(self, x):
        """fp32 array -> raw bf16 bits (int16 array of the same shape), round to nearest even."""
        m, L = self.mem, self.hlib
        y = m.empty(tuple(int(v) for v in x.shape), "i16")
        _lib.check(L.frcnn_f32_to_bf16(m.ptr(x), int(np.prod(x.shape)), m.ptr(y), m.stream()), "frcnn_f32_to_bf16")
        return y

    def linear_bf16(self, x, w, bias, relu=False, out_bf16=False):
        """x (M,K) bf16 bits, w (N,K) bf16 bits, bias (N,) fp32 -> (M,N) fp32 (or bf16 bits)."""
        m, L = self.mem, self.hlib
        M, K = int(x.shape[0]), int(np.prod(x.shape[1:]))
        N = int(w.shape[0])
        assert int(np.prod(w.shape[1:])) == K
        y = m.empty((M, N), "i16" if out_bf16 else "f32")
        ws = self.workspace("linear", L.frcnn_linear_bf16_workspace_bytes(M, N, K))
        _lib.check(L.frcnn_linear_bf16(m.ptr(x), m.ptr(w), m.ptr(bias), m.ptr(y), M, N, K, int(bool(relu)), int(bool(out_bf16)), m.ptr(ws),
                                       ws.shape[0], m.stream()), "frcnn_linear_bf16")
        return y

    def linear_bf16_tile_w(self, w_bits):
        """(N, K) bf16 bits -> the weight-stream layout of frcnn_linear_bf16_tiled (8 KB tiles [ceil(N/128)][K/32], the kernel's swizzled LDS image);
        once, at load time.  K % 32 == 0."""
        m, L = self.mem, self.hlib
        N, K = int(w_bits.shape[0]), int(np.prod(w_bits.shape[1:]))
        nbytes = L.frcnn_linear_bf16_tiled_bytes(N, K)
        if nbytes == 0:
            raise ValueError("frcnn_linear_bf16_tile_w: K = %d is not a multiple of 32" % K)
        wt = m.empty((nbytes // 2,), "i16")
        _lib.check(L.frcnn_linear_bf16_tile_w(m.ptr(w_bits), N, K, m.ptr(wt), m.stream()), "frcnn_linear_bf16_tile_w")
        return wt

    def linear_bf16_tiled(self, x, w_tiled, N, bias, relu=False, out_bf16=False):
        """x (M,K) bf16 bits, w_tiled = linear_bf16_tile_w of the (N,K) weights, bias (N,) fp32 -> (M,N) fp32 (or bf16 bits)."""
        m, L = self.mem, self.hlib
        M, K = int(x.shape[0]), int(np.prod(x.shape[1:]))
        y = m.empty((M, N), "i16" if out_bf16 else "f32")
        ws = self.workspace("linear", L.frcnn_linear_bf16_tiled_workspace_bytes(M, N, K))
        _lib.check(L.frcnn_linear_bf16_tiled(m.ptr(x), m.ptr(w_tiled), m.ptr(bias), m.ptr(y), M, N, K, int(bool(relu)), int(bool(out_bf16)), m.ptr(ws),
                                             ws.shape[0], m.stream()), "frcnn_linear_bf16_tiled")
        return y

    def rpn_heads_bf16(self, h_blk, w_packed, bias, cmid, A):
        """h_blk [CmidP/16][H][W][16] bf16, stacked bf16-packed 1x1 weights, (6A,) fp32 bias -> (rpn_cls_score (1,2A,H,W),
        rpn_cls_prob (1,2A,H,W), rpn_bbox_pred (1,4A,H,W)) fp32: both heads and the softmax in one launch."""
        m, L = self.mem, self.hlib
        H, W = int(h_blk.shape[1]), int(h_blk.shape[2])
        raw = m.empty((1, 6 * A, H, W), "f32")
        prob = m.empty((1, 2 * A, H, W), "f32")
        _lib.check(L.frcnn_rpn_heads_bf16(m.ptr(h_blk), int(cmid), H, W, int(A), m.ptr(w_packed), m.ptr(bias), m.ptr(raw), m.ptr(prob),
                                          m.stream()), "frcnn_rpn_heads_bf16")
        return raw[:, :2 * A], prob, raw[:, 2 * A:]

    def softmax_channels(self, score):
        """(n_ch, H, W) fp32 -> softmax over the channel axis."""
        m, L = self.mem, self.lib
        n_ch, H, W = [int(v) for v in score.shape[-3:]]
        prob = m.empty((1, n_ch, H, W), "f32")
        _lib.check(L.frcnn_softmax_channels_f32(m.ptr(score), n_ch, H * W, m.ptr(prob), m.stream()), "frcnn_softmax_channels_f32")
        return prob

    # ------------------------------------------------------------------ ResNet trunk pieces
    def im2col7x7s2(self, x, Kp):
        m, L = self.mem, self.lib
        C, H, W = [int(v) for v in x.shape[-3:]]
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        cols = m.empty((1, Kp, OH, OW), "f32")
        _lib.check(L.frcnn_im2col7x7s2_f32(m.ptr(x), C, H, W, int(Kp), m.ptr(cols), m.stream()), "frcnn_im2col7x7s2_f32")
        return cols

    def maxpool3x3s2(self, x):
        m, L = self.mem, self.lib
        C, H, W = [int(v) for v in x.shape[-3:]]
        y = m.empty((1, C, (H - 2) // 2 + 1, (W - 2) // 2 + 1), "f32")
        _lib.check(L.frcnn_maxpool3x3s2_f32(m.ptr(x), m.ptr(y), C, H, W, m.stream()), "frcnn_maxpool3x3s2_f32")
        return y

    def subsample2(self, x):
        m, L = self.mem, self.lib
        C, H, W = [int(v) for v in x.shape[-3:]]
        y = m.empty((1, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), "f32")
        _lib.check(L.frcnn_subsample2_f32(m.ptr(x), m.ptr(y), C, H, W, m.stream()), "frcnn_subsample2_f32")
        return y

    def bbox_overlaps(self, boxes, query_boxes):
        m, L = self.mem, self.lib
        N, K = int(boxes.shape[0]), int(query_boxes.shape[0])
        out = m.zeros((N, K), "f64")
        _lib.check(L.frcnn_bbox_overlaps_f64(m.ptr(boxes) if N else None, N, m.ptr(query_boxes) if K else None, K,
                                             m.ptr(out) if N * K else None, m.stream()), "frcnn_bbox_overlaps_f64")
        return out

    def anchor_target(self, anchors, H, W, feat_stride, im_h, im_w, gt_boxes):
        """gt_boxes (G,5) f32 device.  -> (inds_inside, n_inside (1,), labels (pre-subsample), targets, argmax): capacity A*H*W."""
        m, L = self.mem, self.lib
        A, G = int(anchors.shape[0]), int(gt_boxes.shape[0])
        n_all = A * H * W
        inds, n_in = m.empty((n_all,), "i32"), m.empty((1,), "i32")
        labels, targets, argmax = m.empty((n_all,), "i32"), m.empty((n_all, 4), "f32"), m.empty((n_all,), "i32")
        ws = self.workspace("anchor_target", L.frcnn_anchor_target_workspace_bytes(A, H, W, G))
        anc = np.ascontiguousarray(anchors, dtype=np.float64)
        _lib.check(L.frcnn_anchor_target(anc.ctypes.data_as(ctypes.c_void_p), A, int(H), int(W), int(feat_stride), int(im_h), int(im_w),
                                         m.ptr(gt_boxes), G, m.ptr(inds), m.ptr(n_in), m.ptr(labels), m.ptr(targets), m.ptr(argmax),
                                         m.ptr(ws), ws.shape[0], m.stream()), "frcnn_anchor_target")
        return inds, n_in, labels, targets, argmax

    def rpn_loss(self, score, bbox_pred, labels, targets, inds, n_inside, A, H, W, delta=3.0, loss_lambda=1.0, want_grad=True,
                 d_score=None, d_bbox=None):
        """-> (losses (3,) device [cls, bbox, accuracy], d_score (2A,H,W), d_bbox (4A,H,W))"""
        m, L = self.mem, self.lib
        losses = m.empty((3,), "f32")
        if want_grad:
            d_score = d_score if d_score is not None else m.empty((2 * A, H, W), "f32")
            d_bbox = d_bbox if d_bbox is not None else m.empty((4 * A, H, W), "f32")
        _lib.check(L.frcnn_rpn_loss(m.ptr(score), m.ptr(bbox_pred), m.ptr(labels), m.ptr(targets), m.ptr(inds), int(n_inside), A, H, W,
                                    float(delta), float(loss_lambda), m.ptr(losses), m.ptr(d_score) if want_grad else None,
                                    m.ptr(d_bbox) if want_grad else None, m.stream()), "frcnn_rpn_loss")
        return (losses, d_score, d_bbox) if want_grad else losses

    def rcnn_loss(self, cls_score, bbox_pred, labels, targets, delta=1.0, want_grad=True):
        """(R,ncls), (R,4ncls), labels (R,) i32, targets (R,4ncls) -> losses (3,) [, d_cls_score, d_bbox_pred]."""
        m, L = self.mem, self.lib
        R, ncls = int(cls_score.shape[0]), int(cls_score.shape[1])
        losses = m.empty((3,), "f32")
        ds = m.empty((R, ncls), "f32") if want_grad else None
        db = m.empty((R, 4 * ncls), "f32") if want_grad else None
        _lib.check(L.frcnn_rcnn_loss(m.ptr(cls_score), m.ptr(bbox_pred), m.ptr(labels), m.ptr(targets), R, ncls, float(delta), m.ptr(losses),
                                     m.ptr(ds), m.ptr(db), m.stream()), "frcnn_rcnn_loss")
        return (losses, ds, db) if want_grad else losses

    def mul(self, a, b, out=None):
        m, L = self.mem, self.lib
        y = out if out is not None else m.empty(tuple(int(v) for v in a.shape), "f32")
        _lib.check(L.frcnn_mul_f32(m.ptr(a), m.ptr(b), int(np.prod(a.shape)), m.ptr(y), m.stream()), "frcnn_mul_f32")
        return y

    def dropout(self, x, ratio, seed):
        """F.dropout with a device-drawn mask -> (y, mask); mask holds 0 or 1/(1-ratio) (the backward pass multiplies by it)."""
        m, L = self.mem, self.lib
        shape = tuple(int(v) for v in x.shape)
        y, mask = m.empty(shape, "f32"), m.empty(shape, "f32")
        _lib.check(L.frcnn_dropout_f32(m.ptr(x), int(np.prod(shape)), float(ratio), int(seed) & (2 ** 64 - 1), m.ptr(mask), m.ptr(y), m.stream()),
                   "frcnn_dropout_f32")
        return y, mask

    def add(self, a, b, out=None):
        m, L = self.mem, self.lib
        y = out if out is not None else m.empty(tuple(int(v) for v in a.shape), "f32")
        _lib.check(L.frcnn_add_f32(m.ptr(a), m.ptr(b), int(np.prod(a.shape)), m.ptr(y), m.stream()), "frcnn_add_f32")
        return y

    def relu_bwd_(self, g, out):
        m, L = self.mem, self.lib
        _lib.check(L.frcnn_relu_bwd_f32(m.ptr(g), m.ptr(out), int(np.prod(g.shape)), m.stream()), "frcnn_relu_bwd_f32")
        return g

    def gather_rows(self, src, idx):
        """dst[i] = src[idx[i]]: rows of 4-byte elements moved as they are (fp32, or the int32 arg-max rows of the RoI pooling)."""
        m, L = self.mem, self.lib
        n, cols = int(idx.shape[0]), int(np.prod(src.shape[1:]))
        dt = m.dtype_of(src)
        assert dt in ("f32", "i32")
        dst = m.empty((n,) + tuple(int(v) for v in src.shape[1:]), dt)
        _lib.check(L.frcnn_gather_rows_f32(m.ptr(src), m.ptr(idx), n, cols, m.ptr(dst), m.stream()), "frcnn_gather_rows_f32")
        return dst

    def scatter_rows(self, src, idx, dst_rows):
        m, L = self.mem, self.lib
        n, cols = int(idx.shape[0]), int(np.prod(src.shape[1:]))
        dst = m.empty((int(dst_rows),) + tuple(int(v) for v in src.shape[1:]), "f32")
        _lib.check(L.frcnn_scatter_rows_f32(m.ptr(src), m.ptr(idx), n, cols, m.ptr(dst), int(dst_rows), m.stream()), "frcnn_scatter_rows_f32")
        return dst

    def maxpool2x2_bwd(self, x, dy, out=None):
        m, L = self.mem, self.lib
        C, H, W = [int(v) for v in x.shape[-3:]]
        dx = out if out is not None else m.empty((1, C, H, W), "f32")
        _lib.check(L.frcnn_maxpool2x2_bwd_f32(m.ptr(x), m.ptr(dy), m.ptr(dx), C, H, W, m.stream()), "frcnn_maxpool2x2_bwd_f32")
        return dx

    def conv_relu_pool_train(self, x, w_packed, bias):
        """conv3x3 + ReLU + F.MaxPooling2D(2,2) in one launch, training form: (pooled map (1,Cout,OH,OW), arg-max bytes (Cout,OH,OW) u8)
        -- what the pool's backward pass needs instead of the pre-pool map, which is never written."""
        m, L = self.mem, self.lib
        ci, H, W = [int(v) for v in x.shape[-3:]]
        co = int(w_packed.shape[1])
        oh, ow = (H + 1) // 2, (W + 1) // 2
        y = m.empty((1, co, oh, ow), "f32")
        idx = m.empty((co, oh, ow), "u8")
        ws = self._conv_workspace(ci, co, H, W)
        _lib.check(L.frcnn_conv_f32_ex(m.ptr(x), m.ptr(w_packed), m.ptr(bias), m.ptr(idx), m.ptr(y), ci, co, H, W, 3, 5, m.ptr(ws), ws.shape[0],
                                       m.stream()), "frcnn_conv_f32_ex")
        return y, idx

    def conv_dgrad_unpool(self, dy, w_dgrad, zero_bias, idx, H2, W2):
        """input-gradient convolution of the layer above a fused pool + that pool's backward pass: dy (1,Cin,H,W), idx (Cout,H,W) u8 from
        conv_relu_pool_train of the layer below -> dL/d(pre-pool map) (1,Cout,H2,W2)."""
        m, L = self.mem, self.lib
        ci, H, W = [int(v) for v in dy.shape[-3:]]
        co = int(w_dgrad.shape[1])
        assert int(w_dgrad.shape[0]) == ci * 9 and tuple(int(v) for v in idx.shape) == (co, H, W)
        dx = m.empty((1, co, int(H2), int(W2)), "f32")
        ws = self._conv_workspace(ci, co, H, W)
        _lib.check(L.frcnn_conv_dgrad_unpool_f32(m.ptr(dy), m.ptr(w_dgrad), m.ptr(zero_bias), m.ptr(idx), m.ptr(dx), ci, co, H, W, int(H2), int(W2), m.ptr(ws),
                                                 ws.shape[0], m.stream()), "frcnn_conv_dgrad_unpool_f32")
        return dx

    def maxpool2x2_bwd_idx(self, idx, dy, H, W):
        m, L = self.mem, self.lib
        C = int(idx.shape[0])
        dx = m.empty((1, C, int(H), int(W)), "f32")
        _lib.check(L.frcnn_maxpool2x2_bwd_idx_f32(m.ptr(idx), m.ptr(dy), m.ptr(dx), C, int(H), int(W), m.stream()), "frcnn_maxpool2x2_bwd_idx_f32")
        return dx

    def bias_grad(self, dy, out=None):
        m, L = self.mem, self.lib
        C = int(dy.shape[-3])
        HW = int(dy.shape[-2]) * int(dy.shape[-1])
        db = out if out is not None else m.empty((C,), "f32")
        ws = self.workspace("bias_grad", L.frcnn_bias_grad_workspace_bytes(C, HW))
        _lib.check(L.frcnn_bias_grad_f32(m.ptr(dy), C, HW, m.ptr(db), m.ptr(ws), ws.shape[0], m.stream()), "frcnn_bias_grad_f32")
        return db

    def pack_conv_dgrad_w(self, w_packed, ksize=3, out=None):
        """(Cin*k*k, Cout) forward-packed -> (Cout*k*k, Cin): weights of the input-gradient convolution."""
        m, L = self.mem, self.lib
        co = int(w_packed.shape[1])
        ci = int(w_packed.shape[0]) // (ksize * ksize)
        wd = out if out is not None else m.empty((co * ksize * ksize, ci), "f32")
        _lib.check(L.frcnn_pack_conv_dgrad_w(m.ptr(w_packed), ci, co, int(ksize), m.ptr(wd), m.stream()), "frcnn_pack_conv_dgrad_w")
        return wd

    def pack_conv_dgrad_w_many(self, layers):
        """layers: [(forward-packed weights (cin*k*k, cout), dgrad weights (cout*k*k, cin), ksize)], at most 16: ONE launch."""
        import ctypes

        class Desc(ctypes.Structure):
            _fields_ = [("w", ctypes.c_void_p), ("wd", ctypes.c_void_p), ("cin", ctypes.c_int), ("cout", ctypes.c_int), ("ks", ctypes.c_int)]
        m, L = self.mem, self.lib
        descs = []
        for wp, wd, ks in layers:
            co = int(wp.shape[1])
            ci = int(wp.shape[0]) // (ks * ks)
            assert tuple(wd.shape) == (co * ks * ks, ci)
            descs.append(Desc(m.ptr(wp).value, m.ptr(wd).value, ci, co, int(ks)))
        arr = (Desc * len(descs))(*descs)
        _lib.check(L.frcnn_pack_conv_dgrad_w_many(ctypes.cast(arr, ctypes.c_void_p), len(descs), m.stream()), "frcnn_pack_conv_dgrad_w_many")

    def conv_wgrad(self, x, dy, ksize=3, out=None):
        """dW in the forward-packed layout (Cin*k*k, Cout)."""
        m, L = self.mem, self.lib
        ci, H, W = [int(v) for v in x.shape[-3:]]
        co = int(dy.shape[-3])
        dw = out if out is not None else m.empty((ci * ksize * ksize, co), "f32")
        ws = self.workspace("wgrad", L.frcnn_conv_wgrad_workspace_bytes(ci, co, H, W, int(ksize)))
        _lib.check(L.frcnn_conv_wgrad_f32(m.ptr(x), m.ptr(dy), m.ptr(dw), ci, co, H, W, int(ksize), m.ptr(ws), ws.shape[0], m.stream()),
                   "frcnn_conv_wgrad_f32")
        return dw

    def conv_wgrad_f32s(self, x, dy, out=None):
        """The 3x3 weight gradient as bf16x6 split products (csrc/train.hip conv_wgrad_f32s_kernel); same layout as conv_wgrad."""
        m, L = self.mem, self.lib
        ci, H, W = [int(v) for v in x.shape[-3:]]
        co = int(dy.shape[-3])
        dw = out if out is not None else m.empty((ci * 9, co), "f32")
        ws = self.workspace("wgrad", L.frcnn_conv_wgrad_workspace_bytes(ci, co, H, W, 3))
        _lib.check(L.frcnn_conv_wgrad_f32s(m.ptr(x), m.ptr(dy), m.ptr(dw), ci, co, H, W, m.ptr(ws), ws.shape[0], m.stream()),
                   "frcnn_conv_wgrad_f32s")
        return dw

    def sgd_momentum_wd(self, w, grad, velocity, lr, momentum, weight_decay):
        m, L = self.mem, self.lib
        n = int(np.prod(w.shape))
        _lib.check(L.frcnn_sgd_momentum_wd(m.ptr(w), m.ptr(grad), m.ptr(velocity), n, float(lr), float(momentum), float(weight_decay),
                                           m.stream()), "frcnn_sgd_momentum_wd")

    def transpose(self, src, out=None):
        m, L = self.mem, self.lib
        R, C = int(src.shape[0]), int(src.shape[1])
        dst = out if out is not None else m.empty((C, R), "f32")
        _lib.check(L.frcnn_transpose_f32(m.ptr(src), R, C, m.ptr(dst), m.stream()), "frcnn_transpose_f32")
        return dst


_default = None


def default_runtime():
    """The MI355X runtime.  Raises (never falls back) if the HIP library or the GPU is missing."""
    global _default
    if _default is None:
        _default = Runtime(_lib.load(), TorchDeviceMemory())
    return _default
