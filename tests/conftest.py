import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(autouse=True)
def _tuning_back_to_defaults():
    """A/B knobs are set through the tuning registry (chainer_faster_rcnn_amd.tuning -> frcnn_set_tuning), never through the
    environment; every test starts and ends on the load-time snapshot."""
    yield
    mod = sys.modules.get("chainer_faster_rcnn_amd.tuning")
    if mod is not None:
        mod.reset()
