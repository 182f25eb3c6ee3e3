import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "sparse-solvers_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)


# (No process-wide engine switch: every test runs the shipped defaults unless it sets an option itself.  The tests of
# test_gpu_parity.py that examine ONE of the engines behind the screened form on a dictionary large enough for that form to
# take the signal pin option "screen_single" = 0 through the fixture `engine_forms` there.)


def note(test, **facts):
    """Counts a test observed but does not assert on exactly (diverged paths, tie re-runs, stuck signals):
    printed (pytest -s / the failure report shows them) and appended to gpurun_out/test_notes.jsonl, which
    comes back from the GPU box with the run."""
    import json
    line = json.dumps({"test": test, **{k: (v.item() if hasattr(v, "item") else v) for k, v in facts.items()}})
    print("[note] " + line)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_notes.jsonl"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "homotopy_golden.npz")
    g = np.load(path)
    cases = {}
    for key in g.files:
        name, field = key.split("/")
        cases.setdefault(name, {})[field] = g[key]
    return cases


def make_gaussian_problem(seed, m, n, k, dtype, scale=None):
    """SURVEY §8d recipe: A ~ N(0,1)/sqrt(m), k positive coefficients 1+|N(0,1)|,
    y = A x0 computed in float64 then cast."""
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((m, n), dtype=np.float32 if np.dtype(dtype) == np.float32 else np.float64)
         / np.sqrt(m)).astype(dtype)
    x0 = np.zeros(n)
    sup = np.sort(rng.choice(n, size=k, replace=False))
    x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
    y = (A.astype(np.float64) @ x0).astype(dtype)
    return A, y, x0, sup
