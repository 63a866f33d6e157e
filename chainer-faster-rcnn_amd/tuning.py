"""Tuning registry of the package: the Python face of csrc/frcnn_tune.h.

The library snapshots the FRCNN_* environment variables ONCE, when it is loaded; after that a knob changes only
through `frcnn_set_tuning` (include/frcnn_hip.h).  This module keeps the host-side copy of the same table (the few
knobs the Python wrappers read themselves -- e.g. FRCNN_TRAIN_FUSE_POOL -- come from here, not from os.environ) and
forwards every `set` to each library `_lib.bind()` has opened.  Defaults (no entry) are the measured picks; no key
selects a CPU path.
"""
import contextlib
import os

_snapshot = {k: v for k, v in os.environ.items() if k.startswith("FRCNN_")}      # taken at import, like the library's at load
_table = dict(_snapshot)
_libs = []


def _push(lib, key, value):
    rc = lib.frcnn_set_tuning(key.encode(), None if value is None else str(value).encode())
    if rc != 0:
        raise ValueError("frcnn_set_tuning(%r, %r) -> %d" % (key, value, rc))


def register(lib):
    """Called by _lib.bind(): bring a freshly opened library to this table's state.  The library took its own snapshot of the
    environment when it was loaded (possibly later than this module's): every entry of the table is pushed, and FRCNN_* variables
    that are in the environment now but not in the table are cleared, so both sides hold the same picture."""
    for k in os.environ:
        if k.startswith("FRCNN_") and k not in _table:
            _push(lib, k, None)
    for k, v in _table.items():
        _push(lib, k, v)
    _libs.append(lib)


def get(key, default=None):
    return _table.get(key, default)


def set(key, value):                       # noqa: A001 (the registry's verb)
    """value None removes the entry (= the default pick)."""
    if not key.startswith("FRCNN_"):
        raise ValueError("tuning keys start with FRCNN_: %r" % (key,))
    for lib in _libs:
        _push(lib, key, value)
    if value is None:
        _table.pop(key, None)
    else:
        _table[key] = str(value)


def reset():
    """Back to the load-time snapshot, here and in every bound library."""
    for lib in _libs:
        lib.frcnn_reset_tuning()
    _table.clear()
    _table.update(_snapshot)


@contextlib.contextmanager
def override(**kv):
    old = {k: _table.get(k) for k in kv}
    try:
        for k, v in kv.items():
            set(k, v)
        yield
    finally:
        for k, v in old.items():
            set(k, v)
