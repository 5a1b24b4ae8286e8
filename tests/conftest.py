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


# Collection order: arithmetic parity against the oracle / the reference fixtures first, host logic next, the multi-process
# orchestration files last -- under `pytest -x` a driver-level failure can then never hide a kernel-vs-oracle test (round 4: one
# red test in test_gpu_sharded.py kept all of test_gpu_split_gemm.py from running).
_ORDER = ("test_gpu_parity", "test_gpu_split_gemm", "test_gpu_hp_variants", "test_gpu_autograd_hp", "test_gpu_metrics", "test_gpu_pia",
          "test_gpu_rams", "test_gpu_drivers", "test_gpu_entrypoints", "test_gpu_cfg4", "test_gpu_bench_contract")
_LAST = ("test_gpu_sharded", "test_gpu_nccl")


def _file_rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _ORDER:
        return _ORDER.index(name)
    if name in _LAST:
        return 1000 + _LAST.index(name)
    return 500


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_file_rank)           # (stable: the order inside a file, and of the files not named above, is kept)


@pytest.fixture(autouse=True)
def _seed_global_generators(request):
    """Every test starts from generators seeded by its own node id: a test that draws from torch's / numpy's GLOBAL generator
    (or lets the product do so: `seed=None` paths follow superresDWI.py:105-118, which draws unseeded) sees the same numbers on
    every box and in every collection order."""
    import zlib

    import torch
    seed = zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF
    np.random.seed(seed)
    torch.manual_seed(seed)            # (seeds the device generators too when a GPU is present)
    yield


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
