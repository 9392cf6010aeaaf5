import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture
def gemm_arith():
    """Selects the arithmetic of pk2_gemm_f32 for one test ("f32" | "bf16x3") and restores the process default after it."""
    from pykaldi2_amd import _lib
    L = _lib.lib()
    before = L.pk2_gemm_get_arith()

    def select(name):
        _lib.check(L.pk2_gemm_set_arith({"f32": 0, "bf16x3": 1}[name]))
    yield select
    L.pk2_gemm_set_arith(before)
