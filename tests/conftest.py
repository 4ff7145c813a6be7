import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "face-diffusion-model_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library is built in-tree (git-ignored); build it if this checkout does not have it yet."""
    from fdm_amd import _lib
    so = os.path.join(ROOT, "oracle", "_build", "liboracle_c.so")
    if not os.path.exists(_lib.LIB_PATH) or not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    yield
