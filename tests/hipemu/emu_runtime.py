"""Runtime over the host-emulated kernels (tests/hipemu/_build/libfrcnn_emu.so): "device" arrays are NumPy
arrays.  Test infrastructure only -- lives under tests/, never imported by the product package."""
import ctypes
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

_NP = {"f32": np.float32, "i32": np.int32, "u8": np.uint8, "f64": np.float64, "i64": np.int64, "i16": np.int16}


class HostMemory(object):
    def empty(self, shape, dtype="f32"):
        return np.full(shape, 0x7f if dtype == "u8" else -12345, dtype=_NP[dtype])   # poisoned, not zero

    def zeros(self, shape, dtype="f32"):
        return np.zeros(shape, dtype=_NP[dtype])

    def from_numpy(self, a):
        return np.array(a, order="C", copy=True)

    def from_numpy_async(self, a):
        return np.array(a, order="C", copy=True)

    def to_numpy(self, a):
        return np.array(a, copy=True)

    def to_numpy_many(self, arrays):
        return [np.array(a, copy=True) for a in arrays]

    def to_numpy_many_async(self, arrays):
        out = [np.array(a, copy=True) for a in arrays]
        return lambda: out

    def is_array(self, a):
        return isinstance(a, np.ndarray)

    def contiguous(self, a):
        return np.ascontiguousarray(a)

    def astype(self, a, dtype):
        return np.ascontiguousarray(a, dtype=_NP[dtype])

    def bitcast(self, a, dtype):
        return a.view(_NP[dtype])

    def view(self, flat, offset, shape):
        n = int(np.prod(shape))
        return flat[offset:offset + n].reshape(shape)

    def within(self, a, flat):
        lo = flat.ctypes.data
        return lo <= a.ctypes.data and a.ctypes.data + a.nbytes <= lo + flat.nbytes

    def ptr(self, a):
        if a is None:
            return None
        assert a.flags["C_CONTIGUOUS"]
        return ctypes.c_void_p(a.ctypes.data)

    def stream(self):
        return None

    def synchronize(self):
        pass

    def side_stream(self, *arrays):
        import contextlib
        return contextlib.nullcontext()

    def join_side_stream(self):
        pass

    def aux_stream(self, name, *arrays):                 # runtime.py: a second stream next to the current one; the host runs in order
        import contextlib
        return contextlib.nullcontext()

    def join_aux_stream(self, name):
        pass

    def early_stream(self, *after):                      # runtime.py: the side stream waits for the producers of `after`; the host runs in order
        import contextlib
        return contextlib.nullcontext()

    def join_early_stream(self, *outputs):
        pass

    def dtype_of(self, a):
        return {np.dtype(v): k for k, v in _NP.items()}[a.dtype]


_rt = None


def emu_runtime(sources=None):
    global _rt
    if _rt is None:
        import build_emu
        so = build_emu.build()
        pkg = importlib.import_module("chainer_faster_rcnn_amd")
        lib = pkg._lib.bind(so)
        assert lib.frcnn_device_count() == 0          # proves this is the emulator, not a device build
        _rt = pkg.runtime.Runtime(lib, HostMemory())
    return _rt
