import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def pkg():
    """The product package.  Its directory name (presto-1_amd) is not a Python identifier, hence importlib."""
    return importlib.import_module("presto-1_amd")


@pytest.fixture(autouse=True)
def _seed_offset(monkeypatch):
    """TGPU_TEST_SEED_OFFSET=k (default 0 = the committed, deterministic inputs) shifts every integer seed the tests pass to
    numpy's default_rng: the same parity tests over different random inputs, for soak runs on the GPU box"""
    off = int(os.environ.get("TGPU_TEST_SEED_OFFSET", "0"))
    if off:
        import numpy as np

        real = np.random.default_rng
        monkeypatch.setattr(np.random, "default_rng", lambda seed=None, *a, **k: real(seed + off if isinstance(seed, int) else seed, *a, **k))
    yield
