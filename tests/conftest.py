"""pytest configuration: `gpu` marker, import paths, shared synthetic indices."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs (a host <-> kernel hand-shake gone wrong spins forever) must not take the box with it: every gpu test
    gets a hard limit.  method="thread": the limit fires even while the main thread sits inside a C call (os._exit)."""
    try:
        import pytest_timeout  # noqa: F401
    except Exception:
        return
    for it in items:
        if it.get_closest_marker("gpu") and not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(420, method="thread"))


def _make(N, D, dtype, R, m, Q, seed):
    from bang_amd import synth
    return synth.make_index(N, D, dtype, R, m, Q, K=10, n_clusters=32, seed=seed, device="cpu", pq_iters=4)


@pytest.fixture(scope="session")
def small_f32():
    """C1-like plumbing case, shrunk: float32, D=128, m=32 (4 dims per chunk)."""
    return _make(3000, 128, "float", 64, 32, 48, 11)


@pytest.fixture(scope="session")
def small_u8():
    """SIFT1B-like layout, shrunk: uint8, D=128, m=70 (chunks of 2 and 1 dims, rows not dword aligned)."""
    return _make(4000, 128, "uint8", 64, 70, 64, 12)


@pytest.fixture(scope="session")
def small_deep():
    """DEEP100M-like layout, shrunk: float32, D=96, m=74."""
    return _make(2500, 96, "float", 64, 74, 40, 13)


@pytest.fixture(scope="session")
def small_i8():
    """int8 vectors, D=64, m=16 (4 dims per chunk), degree bound 32 (ragged lists)."""
    return _make(2000, 64, "int8", 32, 16, 32, 14)


@pytest.fixture(scope="session")
def libbang():
    import bang_amd
    bang_amd.build()
    return bang_amd.lib()
