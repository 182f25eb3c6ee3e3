"""GPU tests of IRLS through the C-ABI (ss_hip_irls_*), the C++14 API (ss::irls<T>) and the
pybind11 module (sparsesolvers.Irls): parity with the CPU oracle and the reference's own IRLS tests
(src/solvers/irls_test.cpp, bindings/python/tests/test_binding.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import oracle  # noqa: E402
import ref_cases  # noqa: E402
from test_oracle_irls import IRLS_PERMUTATIONS, _golden, irls_as_solver  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sship():
    import sship as mod
    assert mod.device_count() >= 1, "no HIP device visible"
    return mod


def hip_irls(sship):
    def fn(A, y, tol, max_iter):
        with sship.Irls(A) as h:
            return h.solve(np.asarray(y, dtype=A.dtype), tol, max_iter)
    return fn


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(24, 10, 2), (64, 20, 3), (300, 120, 6), (1000, 300, 8)])
def test_irls_vs_oracle(sship, shape, dtype):
    M, N, k = shape
    rng = np.random.default_rng(77 + M)
    A = (rng.normal(0.0, 0.05, size=(M, N)) + np.eye(M, N)).astype(dtype)
    x0 = np.zeros(N, dtype)
    x0[rng.choice(N, k, replace=False)] = (1.0 + rng.random(k)).astype(dtype)
    y = (A.astype(np.float64) @ x0.astype(np.float64)).astype(dtype)
    # The reweighting spreads the weights over many orders of magnitude, so rounding differences of
    # the Newton step (summation order) grow with N and the iteration count.  Yardstick for fp32: the
    # same algorithm in fp64 — the device must be as close to it as the fp32 oracle is.
    A64, y64 = A.astype(np.float64), y.astype(np.float64)
    with sship.Irls(A) as h:
        for it in (1, 2, 4):
            xo, ito, eo, spdo = oracle.irls(A, y, 0.01, it)
            xg, itg, eg, spdg = h.solve(y, 0.01, it)
            assert itg == ito and spdg == spdo, (it, itg, ito)
            scale = np.abs(xo).max()
            if dtype == np.float64:
                assert np.abs(xg - xo).max() <= 1e-9 * scale, it
                assert abs(eg - eo) <= 1e-9 * max(1e-3, abs(eo))
            else:
                xd = oracle.irls(A64, y64, 0.01, it)[0]
                err_ref = np.abs(xo.astype(np.float64) - xd).max()
                err_dev = np.abs(xg.astype(np.float64) - xd).max()
                assert err_dev <= max(10 * err_ref, 1e-5 * scale), (it, err_dev, err_ref)


@pytest.mark.parametrize("name", sorted(_golden().keys()))
def test_irls_golden(sship, name):
    g = _golden()[name]
    A, y, tol = g["A"], g["y"], float(g["tol"])
    atol = 1e-5 if A.dtype == np.float32 else 1e-11
    with sship.Irls(A) as h:
        for it in (1, 2, 3):
            x, iters, err, spd = h.solve(y, tol, it)
            assert not spd and 1 <= iters <= it
            assert np.abs(x.astype(np.float64) - g["x%d" % it]).max() <= atol, (name, it)


def test_irls_layouts_and_errors(sship):
    rng = np.random.default_rng(5)
    M, N = 40, 12
    A = (rng.normal(0.0, 0.05, size=(M, N)) + np.eye(M, N)).astype(np.float32)
    y = A[:, 3].copy()
    ref = None
    for view in (A, np.asfortranarray(A), np.ascontiguousarray(np.pad(A, ((0, 0), (0, 5))))[:, :N]):
        with sship.Irls(view) as h:
            x, it, e, spd = h.solve(y, 0.01, 20)
        assert int(np.argmax(x)) == 3
        if ref is None:
            ref = x
        else:
            assert np.array_equal(x, ref)             # the device layout does not depend on the host view
    with pytest.raises(sship.SsHipError):
        sship.Irls(np.ones((3, 5), np.float32))          # underdetermined systems are not supported
    with sship.Irls(A) as h:
        with pytest.raises(sship.SsHipError):
            h.solve(y, 0.01, 0)
        with pytest.raises(TypeError):
            h.solve(y.astype(np.float64), 0.01, 5)
    with sship.Homotopy(A) as h:                         # the two context families do not mix
        import ctypes
        err = ctypes.create_string_buffer(256)
        out = np.empty(N, np.float32)
        it, e, spd = ctypes.c_uint32(0), ctypes.c_double(0), ctypes.c_int(0)
        rc = sship.lib().ss_hip_irls_solve_f32(h._h, y.ctypes.data, 1, ctypes.c_float(0.01), 5, out.ctypes.data, 1,
                                               ctypes.byref(it), ctypes.byref(e), ctypes.byref(spd), err, len(err))
        assert rc != 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ref_irls_smoke(sship, dtype):
    ref_cases.smoke(irls_as_solver(hip_irls(sship)), dtype)
    ref_cases.smoke_column_subset(irls_as_solver(hip_irls(sship)), dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ref_irls_noisy_signal(sship, dtype):
    ref_cases.noisy_signal(irls_as_solver(hip_irls(sship)), dtype)


@pytest.mark.parametrize("cfg", IRLS_PERMUTATIONS)
def test_ref_irls_permutations(sship, cfg):
    M, N, dtype, skip, seed = cfg
    ref_cases.permutations(irls_as_solver(hip_irls(sship)), M, N, dtype, .1, .1, skip, seed=seed)


def test_python_module_irls(sship):
    """sparsesolvers.Irls / IrlsReport (binding.cpp:133-146): names, defaults, tuple shape"""
    import sparsesolvers as ss
    N = 10
    A = np.eye(N, dtype=np.float64)
    for n in range(N):
        x, info = ss.Irls(A).solve(A[:, n].copy(), 0.001, N)
        assert np.array_equal(x, A[:, n])
        assert info.iter == 1 and info.solution_error == 0.0 and info.spd_failure is False
    rep = ss.IrlsReport()
    rep.iter, rep.solution_error, rep.spd_failure = 3, 0.5, True
    assert (rep.iter, rep.solution_error, rep.spd_failure) == (3, 0.5, True)
    A32 = (np.eye(20, 8) + 0.01).astype(np.float32)
    x, info = ss.Irls(A32).solve(A32[:, 2].copy(), tolerance=0.01)       # default max_iterations = 100
    assert x.dtype == np.float32 and int(np.argmax(x)) == 2
    with pytest.raises(RuntimeError):
        ss.Irls(A32).solve(A32[:, 2].astype(np.float64))
