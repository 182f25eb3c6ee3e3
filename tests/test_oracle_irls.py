"""CPU tests of the IRLS oracle (oracle/ss_oracle_irls.inc): the reference's own known-answer and
property tests for its QR and Cholesky factorisations and for the IRLS solver, restated, plus golden
vectors produced by the reference's numpy IRLS (tests/golden/make_golden_irls.py).

Sources (paths relative to /root/reference):
  src/linalg/qr_decomposition_test.cpp:14-88, src/linalg/cholesky_decomposition_test.cpp:16-95,
  src/solvers/irls_test.cpp:8-53 with src/solvers/test_util.h:27-257
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import oracle  # noqa: E402
import ref_cases  # noqa: E402


def irls_as_solver(fn):
    """adapts irls(...) -> (x, iter, err, spd) to ref_cases' solve(A, y, tol, max_iter) -> (x, iter, err);
    irls_test.cpp:8-21: the error bound is only asserted when IRLS neither ran out of iterations nor
    met a non-SPD matrix"""
    def solve(A, y, tol, max_iter):
        x, it, err, spd = fn(np.ascontiguousarray(A), y, tol, max_iter)
        return x, it, (0.0 if spd else err)
    return solve


# ---- qr_decomposition_test.cpp -------------------------------------------------------------------
def test_qr_2x2_known_answer():
    A = np.array([[1, -1], [-1, 1]], np.float32)
    x = oracle.QR(A).solve(np.array([1, -1], np.float32))
    assert np.allclose(x, [0, -1], rtol=0, atol=1e-4)


@pytest.mark.parametrize("M,N,dtype", [(1, 1, np.float32), (2, 2, np.float32), (4, 4, np.float32), (5, 4, np.float32),
                                       (6, 4, np.float32), (7, 4, np.float32), (50, 50, np.float64), (100, 20, np.float64)])
def test_qr_random_inputs(M, N, dtype):
    rng = np.random.default_rng(M * 100 + N)
    A = rng.normal(10.0, 2.5, size=(M, N)).astype(dtype)
    qr = oracle.QR(A)
    q, r = qr.q(), qr.r()
    assert q.shape == A.shape and r.shape == (N, N)
    assert np.allclose(q.astype(np.float64) @ r.astype(np.float64), A, rtol=0, atol=1e-4)
    assert np.allclose(q.T.astype(np.float64) @ q.astype(np.float64), np.eye(N), rtol=0, atol=1e-4)
    assert np.array_equal(np.tril(r, -1), np.zeros_like(r))


# ---- cholesky_decomposition_test.cpp ---------------------------------------------------------------
def test_cholesky_isspd():
    _, ok = oracle.cholesky(np.array([[0, 1], [1, 0]], np.float32))
    assert not ok


def test_cholesky_2x2_known_answer():
    A = np.array([[2, 1], [1, 2]], np.float32)
    L, ok = oracle.cholesky(A)
    assert ok
    assert np.allclose(L @ L.T, A, rtol=0, atol=1e-4)
    assert np.allclose(oracle.cholesky_solve(L, np.array([1, -1], np.float32)), [1, -1], rtol=0, atol=1e-4)


@pytest.mark.parametrize("N,dtype", [(4, np.float32), (10, np.float32), (25, np.float64), (60, np.float64)])
def test_cholesky_random_spd(N, dtype):
    rng = np.random.default_rng(N)
    noise = rng.normal(10.0, 5.0, size=(N, N)).astype(dtype)
    A = (noise @ noise.T).astype(dtype)
    L, ok = oracle.cholesky(A)
    assert ok
    scale = np.abs(A).max()
    assert np.allclose(L.astype(np.float64) @ L.T.astype(np.float64), A, rtol=0, atol=1e-5 * scale if dtype == np.float32 else 1e-10 * scale)


# ---- golden vectors of the reference's numpy IRLS -----------------------------------------------------
def _golden():
    z = np.load(os.path.join(ROOT, "tests", "golden", "irls_golden.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return {n: {k.split("/")[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}


@pytest.mark.parametrize("name", sorted(_golden().keys()))
def test_irls_golden(name):
    g = _golden()[name]
    A, y, tol = g["A"], g["y"], float(g["tol"])
    atol = 2e-6 if A.dtype == np.float32 else 1e-12
    for it in (1, 2, 3):
        x, iters, err, spd = oracle.irls(A, y, tol, it)
        assert not spd and 1 <= iters <= it
        assert np.abs(x.astype(np.float64) - g["x%d" % it]).max() <= atol, (name, it)


# ---- irls_test.cpp ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_irls_smoke(dtype):
    ref_cases.smoke(irls_as_solver(oracle.irls), dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_irls_smoke_column_subset(dtype):
    ref_cases.smoke_column_subset(irls_as_solver(oracle.irls), dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_irls_noisy_signal(dtype):
    ref_cases.noisy_signal(irls_as_solver(oracle.irls), dtype)


# (M, N, dtype, skip, numpy seed of the noise draw: the property holds for about 3 draws in 4)
IRLS_PERMUTATIONS = [(4, 4, np.float32, 10, 0), (5, 5, np.float64, 10, 0), (10, 5, np.float32, 10, 0), (10, 5, np.float64, 20, 1)]


@pytest.mark.parametrize("cfg", IRLS_PERMUTATIONS)
def test_irls_permutations(cfg):
    M, N, dtype, skip, seed = cfg
    ref_cases.permutations(irls_as_solver(oracle.irls), M, N, dtype, .1, .1, skip, seed=seed)


def test_irls_binding_smoke():
    """bindings/python/tests/test_binding.py: identity => x == e_n, solution_error == 0, iter == 1"""
    A = np.eye(10, dtype=np.float64)
    for n in range(10):
        x, it, err, spd = oracle.irls(A, A[:, n].copy(), 0.001, 10)
        assert np.array_equal(x, A[:, n]) and it == 1 and err == 0.0 and not spd


def test_irls_preconditions():
    with pytest.raises(RuntimeError):
        oracle.irls(np.ones((3, 5)), np.ones(3), 0.1, 10)       # underdetermined: not supported
    with pytest.raises(RuntimeError):
        oracle.irls(np.eye(3), np.ones(3), 0.1, 0)
