import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def assert_close(a, b, rtol=1e-5, atol_scale=1e-6, what=""):
    """allclose(rtol, atol = atol_scale * max|ref|) -- the float parity bar of BASELINE.json
    (1e-5 relative), SURVEY.md 7 'hard parts' 2."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    atol = atol_scale * max(float(np.abs(b).max()), 1e-30)
    err = np.abs(a - b)
    bad = err > atol + rtol * np.abs(b)
    assert not bad.any(), f"{what}: {int(bad.sum())} elements off, max abs err {err.max():.3e} (atol {atol:.3e})"
