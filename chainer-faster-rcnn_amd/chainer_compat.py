"""Duck-typed stand-ins for the few chainer objects the reference's call sites construct.

Chainer itself is not a dependency.  `Variable` mirrors what forward.py:92-93 / train_rpn.py build
(`chainer.Variable(array, volatile=True)`): an object with `.data` (+ shape/dtype/ndim), which is all the
reference's layers read (proposal_layer.py:127-132).
"""
import numpy as np


class Variable(object):
    def __init__(self, data, volatile=False, name=None):
        self.data = data
        self.volatile = volatile
        self.name = name

    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def ndim(self):
        return len(self.data.shape)

    @property
    def dtype(self):
        d = self.data.dtype
        return d if isinstance(d, np.dtype) else np.dtype(str(d).replace("torch.", ""))

    def __len__(self):
        return self.data.shape[0]


def is_variable(a):
    """True for chainer.Variable-like carriers (an object with .data that is not itself an array)."""
    return hasattr(a, "data") and not isinstance(a, np.ndarray) and not _is_tensor(a)


def unwrap(a):
    """Variable-like -> the array it carries."""
    return a.data if is_variable(a) else a


def _is_tensor(a):
    return type(a).__module__.startswith("torch")


def kind(a):
    """'f' / 'i' / 'u' / 'b' for NumPy arrays and torch tensors alike."""
    a = unwrap(a)
    if isinstance(a, np.ndarray):
        return a.dtype.kind
    s = str(a.dtype)
    return "f" if "float" in s or "bfloat" in s else ("u" if "uint" in s else ("b" if "bool" in s else "i"))
