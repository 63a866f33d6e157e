"""Import alias: the package directory is named `chainer-faster-rcnn_amd` (not a valid Python identifier),
so `import chainer_faster_rcnn_amd` resolves here and executes the real package in place."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "chainer-faster-rcnn_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
exec(compile(open(__file__).read(), __file__, "exec"))
