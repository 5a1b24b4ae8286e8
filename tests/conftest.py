import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(autouse=True)
def _restore_debug_switches():
    """The diagnostic switches behind inr_debug_set are process-global: whatever a test flipped (even one that died between
    set and reset) is back at its default before the next test starts."""
    yield
    from mri_super_resolution_amd import _lib
    if _lib._LIB is not None:
        _lib._LIB.inr_debug_reset()


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        return cache[name]

    return load


def strided_sample(a, k=97):
    """The sampling rule of oracle/gen_golden.py: tensors of <= 512 elements whole, else every k-th element."""
    flat = np.asarray(a).reshape(-1)
    return flat if flat.size <= 512 else flat[::k]
