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
    log = os.environ.get("BGNN_BARS_LOG")        # tools: one JSON line per comparison = the measured error against ITS bar and against the default bar
    if log and a.size:
        import json
        d_atol = 1e-6 * max(float(np.abs(b).max()), 1e-30)
        with open(log, "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "what": what, "rtol": rtol, "atol_scale": atol_scale,
                                "max_abs_err": float(err.max()), "max_ref": float(np.abs(b).max()),
                                "worst_over_this_bar": float((err / (atol + rtol * np.abs(b))).max()),
                                "worst_over_default_bar": float((err / (d_atol + 1e-5 * np.abs(b))).max())}) + "\n")
    assert not bad.any(), f"{what}: {int(bad.sum())} elements off, max abs err {err.max():.3e} (atol {atol:.3e})"
