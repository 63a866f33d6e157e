"""Tuning registry of the package: the Python face of csrc/frcnn_tune.h.

The library snapshots the FRCNN_* environment variables ONCE, when it is loaded; after that a knob changes only
through `frcnn_set_tuning` (include/frcnn_hip.h).  This module keeps the host-side copy of the same table (the few
knobs the Python wrappers read themselves -- e.g. FRCNN_TRAIN_FUSE_POOL -- come from here, not from os.environ) and
forwards every `set` to each library `_lib.bind()` has opened.  Defaults (no entry) are the measured picks; no key
selects a CPU path.
"""
import contextlib
import os

_KEY_MAX, _VAL_MAX, _SLOTS = 47, 79, 96          # csrc/abi.hip: kTuneKey - 1, kTuneVal - 1, kTuneSlots


def _holdable(k, v):
    """What the library's own load-time snapshot keeps (abi.hip tune_snapshot_env skips the rest silently): an FRCNN_* variable with
    an over-long name or value -- a path, a space-separated sweep list for scripts/ -- is not a knob of the library."""
    return len(k.encode()) <= _KEY_MAX and len(str(v).encode()) <= _VAL_MAX


_snapshot = {}
for _k, _v in os.environ.items():                # taken at import, like the library's at load
    if _k.startswith("FRCNN_") and _holdable(_k, _v) and len(_snapshot) < _SLOTS:
        _snapshot[_k] = _v
_table = dict(_snapshot)
_libs = []


def _push(lib, key, value, strict=True):
    rc = lib.frcnn_set_tuning(key.encode(), None if value is None else str(value).encode())
    if rc != 0:
        if strict:
            raise ValueError("frcnn_set_tuning(%r, %r) -> %d" % (key, value, rc))
        import warnings
        warnings.warn("tuning: the library does not hold %s=%r (frcnn_set_tuning -> %d); entry skipped" % (key, value, rc))
        _table.pop(key, None)
        _snapshot.pop(key, None)


def register(lib):
    """Called by _lib.bind(): bring a freshly opened library to this table's state.  The library took its own snapshot of the
    environment when it was loaded (possibly later than this module's): every entry of the table is pushed, and FRCNN_* variables
    that are in the environment now but not in the table are cleared, so both sides hold the same picture.  Never raises on an
    environment entry the library rejects (an unrelated FRCNN_* variable must not make the import fail) -- only an
    explicit set() does."""
    for k in os.environ:
        if k.startswith("FRCNN_") and k not in _table and _holdable(k, ""):
            _push(lib, k, None, strict=False)
    for k, v in list(_table.items()):
        _push(lib, k, v, strict=False)
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
