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


@pytest.fixture(scope="session", autouse=True)
def _built_product_library():
    """Test infrastructure only: if the in-tree liblenv_hip.so is missing (fresh checkout), compile it once with hipcc
    (the product itself never builds or falls back on its own: _lib.lib() raises when the library is absent)."""
    from learning_environments_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
        _lib.build()
    yield
