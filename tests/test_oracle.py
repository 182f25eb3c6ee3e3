"""Pins the CPU oracle (oracle/ss_oracle.c) to the reference:
  (a) golden vectors produced by the reference's own numpy implementation
      (tests/golden/make_golden.py -> homotopy_golden.npz);
  (b) the literal known-answer tests of the reference
      (src/linalg/online_inverse_test.cpp, src/linalg/rank_index_test.cpp);
  (c) the reference's property tests for the Homotopy path (tests/ref_cases.py).
CPU only — no GPU needed.
"""
import numpy as np
import pytest

import oracle
import ref_cases
from conftest import make_gaussian_problem

F32_EPS = float(np.finfo(np.float32).eps)
F64_EPS = float(np.finfo(np.float64).eps)


def o_solve(A, y, tol, max_iter, flags=oracle.SPARSE_NOTRANS):
    return oracle.homotopy(A, y, tol, max_iter, flags=flags)


def o_solve_default(A, y):
    # binding defaults: tolerance = eps(T)*10, max_iterations = 100 (binding.cpp:94-95)
    eps = F32_EPS if A.dtype == np.float32 else F64_EPS
    return oracle.homotopy(A, np.asarray(y, dtype=A.dtype), eps * 10, 100)


# ---------------------------------------------------------------- (a) golden vectors

GOLDEN_TOL = {
    # relative to max|x|.  The reference's numpy code accumulates in float64 even for
    # float32 inputs (homotopy.py:144), so the f32 rows measure f32-vs-f64 arithmetic.
    "gauss_f64_40x120_k4": 1e-10, "gauss_f64_96x384_k8": 1e-10,
    "gauss_f32_64x256_k6": 1e-5, "gauss_f32_128x512_k10": 1e-5,
    "removal_f64_24x64_seed1000": 1e-10,
    # 20 active columns of a 24-row matrix: condition number ~1e2, f32 vs f64
    "removal_f32_24x64_seed1000": 5e-4,
    "readme_toy_f64_10x10": 1e-10, "main_py_5x5_f32": 1e-5,
}


@pytest.mark.parametrize("flags", [0, oracle.SPARSE_NOTRANS])
@pytest.mark.parametrize("name", sorted(GOLDEN_TOL))
def test_golden(golden, name, flags):
    g = golden[name]
    A, y, tol = g["A"], g["y"], float(g["tol"])
    x, it, err, tr = oracle.homotopy(A, y, tol, 4000, flags=flags, trace=True)
    xr = g["x"]
    assert it == int(g["iters"])
    assert err <= tol
    # recovered support: bit-exact
    assert np.array_equal(np.nonzero(x)[0], np.nonzero(xr)[0])
    # coefficients
    assert np.abs(x - xr).max() / np.abs(xr).max() <= GOLDEN_TOL[name]
    # homotopy path: same breakpoints.  The LAST toggle is excluded: on the final
    # segment lambda -> 0, so every off-support candidate of find_max_gamma ties at
    # gamma = lambda and the winner is decided by rounding; it never reaches x
    # (x += gamma*d uses the OLD support's direction, homotopy-cpu.cpp:252).
    assert np.array_equal(tr["idx"][:-1], g["path_idx"][:-1])
    assert np.array_equal(tr["added"], g["path_add"])
    rtol = max(1e-4 if A.dtype == np.float32 else 1e-9, 10 * GOLDEN_TOL[name])
    assert np.allclose(tr["gamma"], g["path_gamma"], rtol=rtol, atol=0)


def test_dense_and_sparse_notrans_are_bit_identical():
    A, y, _, _ = make_gaussian_problem(5, 96, 700, 9, np.float32)
    xa, ita, ea = oracle.homotopy(A, y, 1e-3, 100, flags=0)
    xb, itb, eb = oracle.homotopy(A, y, 1e-3, 100, flags=oracle.SPARSE_NOTRANS)
    assert ita == itb and ea == eb
    assert np.array_equal(xa, xb)


def test_layout_independence():
    """row-major, padded row-major and column-major views give identical results
    (the three layouts of SURVEY §4 / test_binding.py)."""
    A, y, _, _ = make_gaussian_problem(6, 64, 300, 7, np.float64)
    base = oracle.homotopy(A, y, 1e-8, 100)
    padded = np.zeros((64, 340))
    padded[:, 20:320] = A
    colmajor = np.asfortranarray(A)
    for view in (padded[:, 20:320], colmajor, np.ascontiguousarray(A.T).T):
        out = oracle.homotopy(view, y, 1e-8, 100)
        assert out[1] == base[1] and np.array_equal(out[0], base[0])


def test_first_step_sign_quirk():
    """homotopy-cpu.cpp:223-227: the first direction is seeded with sign(|c[idx]|) = +1.
    With a negative leading correlation the default (bug-for-bug) path differs from
    the strict-sign path; with a positive one they coincide."""
    A, y, x0, sup = make_gaussian_problem(7, 64, 256, 5, np.float64)
    a = oracle.homotopy(A, y, 1e-8, 50)
    b = oracle.homotopy(A, y, 1e-8, 50, flags=oracle.SPARSE_NOTRANS | oracle.STRICT_SIGN)
    assert np.array_equal(a[0], b[0])
    # negative signal: strict mode recovers -x0, bug-for-bug mode does not in k iterations
    bq = oracle.homotopy(A, -y, 1e-8, 50, flags=oracle.SPARSE_NOTRANS | oracle.STRICT_SIGN)
    assert np.allclose(bq[0], -x0, atol=1e-8)
    aq = oracle.homotopy(A, -y, 1e-8, 50)
    assert not np.allclose(aq[0], -x0, atol=1e-3) or aq[1] > bq[1]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_exact_tie_and_the_opt_in_guard(dtype):
    """homotopy-cpu.cpp:143-153: an off-support column that attains lambda exactly has t = 0 and the strict
    `t > 0` skips it for good.  A = I, y = e_0 + e_1 (+ 0.5 e_5) makes that tie in exact arithmetic: the
    reference-faithful default wanders and runs out of iterations; TIE_GUARD (a restatement of the HIP
    path's opt-in option, not of the reference) lets the tied column in by a zero-length step."""
    n = 32
    A = np.eye(n, dtype=dtype)
    y = np.zeros(n, dtype=dtype)
    y[0] = y[1] = 1.0
    y[5] = 0.5
    x, it, err, tr = oracle.homotopy(A, y, 1e-3, 6, trace=True)
    assert it == 6 and err == 1.0 and x[1] == 0.0 and 1 not in tr["idx"]
    x, it, err, tr = oracle.homotopy(A, y, 1e-3, 6, flags=oracle.SPARSE_NOTRANS | oracle.TIE_GUARD, trace=True)
    assert it == 3 and err == 0.0 and np.array_equal(x, y)
    assert tr["idx"][1] == 1 and tr["gamma"][1] == np.finfo(dtype).tiny
    # the guard only acts on an exact tie: elsewhere the flag changes nothing
    A2, y2, x0, sup = make_gaussian_problem(7, 64, 256, 5, dtype)
    a = oracle.homotopy(A2, y2, 1e-6, 50)
    b = oracle.homotopy(A2, y2, 1e-6, 50, flags=oracle.SPARSE_NOTRANS | oracle.TIE_GUARD)
    assert a[1] == b[1] and np.array_equal(a[0], b[0])


def test_zero_on_removal_flag_only_touches_leaving_columns(golden):
    """ZERO_ON_REMOVAL restates the HIP path's opt-in option: the removal goldens of the reference's numpy
    solver are met with and without it (the residue is below every tolerance), and on a removal-free path
    the flag changes nothing"""
    g = golden["removal_f64_24x64_seed1000"]
    A, y, tol = g["A"], g["y"], float(g["tol"])
    a = oracle.homotopy(A, y, tol, 4000)
    b = oracle.homotopy(A, y, tol, 4000, flags=oracle.SPARSE_NOTRANS | oracle.ZERO_ON_REMOVAL)
    assert a[1] == b[1] == int(g["iters"])
    assert np.abs(a[0] - b[0]).max() <= 1e-12 * np.abs(a[0]).max()
    A2, y2, x0, sup = make_gaussian_problem(7, 64, 256, 5, np.float64)
    c = oracle.homotopy(A2, y2, 1e-8, 50)
    d = oracle.homotopy(A2, y2, 1e-8, 50, flags=oracle.SPARSE_NOTRANS | oracle.ZERO_ON_REMOVAL)
    assert c[1] == d[1] and np.array_equal(c[0], d[0])


def test_cblas_timing_leg_agrees_with_the_fixed_order_loops():
    """SS_ORACLE_CBLAS (bench.py's CPU baseline: the GEMVs through a dlopen'd CBLAS like the reference's
    blas_wrapper.cpp:33-66) walks the same path as the fixed-order loops; the parity tests never use it"""
    desc = oracle.load_cblas()
    if desc is None:
        pytest.skip("no CBLAS on this host")
    for dtype, tol, rtol in ((np.float32, 1e-3, 1e-5), (np.float64, 1e-9, 1e-12)):
        A, y, x0, sup = make_gaussian_problem(5, 256, 2048, 12, dtype)
        for V in (A, np.asfortranarray(A)):
            a = oracle.homotopy(V, y, tol, 60, flags=0)
            b = oracle.homotopy(V, y, tol, 60, flags=oracle.CBLAS)
            assert a[1] == b[1] and np.array_equal(np.nonzero(a[0])[0], np.nonzero(b[0])[0])
            assert np.abs(a[0] - b[0]).max() <= rtol * np.abs(a[0]).max()


def test_preconditions_are_errors():
    """asserts of homotopy-cpu.cpp:193-199 become error returns."""
    A = np.eye(4, dtype=np.float32)
    y = np.ones(4, dtype=np.float32)
    with pytest.raises(RuntimeError):
        oracle.homotopy(A, y, 1e-3, 0)           # max_iter > 0
    with pytest.raises(RuntimeError):
        oracle.homotopy(A, y, 1.0, 5)            # tol < 1
    with pytest.raises(RuntimeError):
        oracle.homotopy(A, y, F32_EPS / 2, 5)    # tol >= eps


# ------------------------------------------------ (b) literal known-answer tests

def test_square_permute_2():
    """online_inverse_test.cpp:13-35"""
    A = np.array([[1, 2], [3, 4]], np.float32)
    e = np.array([[4, 3], [2, 1]], np.float32)
    t = oracle.square_permute(A, 0, 1)
    assert np.array_equal(t, e)
    assert np.array_equal(oracle.square_permute(t, 1, 0), A)


def test_square_permute_3():
    """online_inverse_test.cpp:37-78"""
    A = np.arange(1, 10, dtype=np.float32).reshape(3, 3)
    e12 = np.array([[1, 3, 2], [7, 9, 8], [4, 6, 5]], np.float32)
    t = oracle.square_permute(A, 1, 2)
    assert np.array_equal(t, e12)
    assert np.array_equal(oracle.square_permute(t, 2, 1), A)
    e02 = np.array([[5, 6, 4], [8, 9, 7], [2, 3, 1]], np.float32)
    t = oracle.square_permute(A, 0, 2)
    assert np.array_equal(t, e02)
    assert np.array_equal(oracle.square_permute(t, 2, 0), A)


def test_square_permute_4():
    """online_inverse_test.cpp:80-124"""
    A = np.arange(1, 17, dtype=np.float32).reshape(4, 4)
    e13 = np.array([[1, 3, 4, 2], [9, 11, 12, 10], [13, 15, 16, 14], [5, 7, 8, 6]], np.float32)
    t = oracle.square_permute(A, 1, 3)
    assert np.array_equal(t, e13)
    assert np.array_equal(oracle.square_permute(t, 3, 1), A)
    e12 = np.array([[1, 3, 2, 4], [9, 11, 10, 12], [5, 7, 6, 8], [13, 15, 14, 16]], np.float32)
    t = oracle.square_permute(A, 1, 2)
    assert np.array_equal(t, e12)
    assert np.array_equal(oracle.square_permute(t, 2, 1), A)
    for dt in (np.float64,):
        assert np.array_equal(oracle.square_permute(A.astype(dt), 1, 3), e13.astype(dt))


def test_erase_last_rowcol():
    """online_inverse_test.cpp:126-152"""
    A = np.arange(1, 10, dtype=np.float32).reshape(3, 3)
    t = oracle.erase_last_rowcol(A)
    assert np.array_equal(t, np.array([[1, 2], [4, 5]], np.float32))
    assert np.array_equal(oracle.erase_last_rowcol(t), np.array([[1]], np.float32))
    B = np.arange(1, 7, dtype=np.float32).reshape(2, 3)
    assert np.array_equal(oracle.erase_last_rowcol(B), np.array([[1, 2]], np.float32))


def test_insert_last_rowcol():
    """online_inverse_test.cpp:154-184"""
    assert np.array_equal(oracle.insert_last_rowcol(np.array([[1]], np.float32)),
                          np.array([[1, 0], [0, 0]], np.float32))
    A = np.arange(1, 10, dtype=np.float32).reshape(3, 3)
    e = np.array([[1, 2, 3, 0], [4, 5, 6, 0], [7, 8, 9, 0], [0, 0, 0, 0]], np.float32)
    assert np.array_equal(oracle.insert_last_rowcol(A), e)
    B = np.array([[1, 2, 3]], np.float32)
    assert np.array_equal(oracle.insert_last_rowcol(B),
                          np.array([[1, 2, 3, 0], [0, 0, 0, 0]], np.float32))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_online_inverse_identity(dtype):
    """online_inverse_test.cpp:186-217"""
    K = 10
    A = np.eye(K, dtype=dtype)
    inv = oracle.OnlineColumnInverse(K, dtype)
    for k in range(K):
        inv.insert(k, A[:, k])
        assert np.allclose(inv.inverse(), np.eye(k + 1), rtol=0, atol=1e-4)
    for k in range(K - 1, 0, -1):
        inv.remove(k)
        assert np.allclose(inv.inverse(), np.eye(k), rtol=0, atol=1e-4)
    inv.remove(0)
    assert inv.N() == 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_online_inverse_random_order(dtype):
    """insert/remove at arbitrary ranks keeps inv == (A_S^T A_S)^-1 for the sorted subset
    (the maths of docs/algorithms/online-matrix-inverse/src.tex:109-142)."""
    rng = np.random.default_rng(4)
    m, n = 40, 12
    A = rng.standard_normal((m, n)).astype(dtype)
    inv = oracle.OnlineColumnInverse(m, dtype)
    ri = oracle.RankIndex()
    active = []
    tol = 2e-3 if dtype == np.float32 else 1e-10
    for step in range(40):
        j = int(rng.integers(n))
        if j in active:
            r = ri.rank_of(j)
            ri.erase(j)
            inv.remove(r)
            active.remove(j)
        else:
            r = ri.insert(j)
            inv.insert(r, A[:, j])
            active.append(j)
        S = sorted(active)
        assert inv.N() == len(S)
        if S:
            As = A[:, S].astype(np.float64)
            want = np.linalg.inv(As.T @ As)
            assert np.allclose(inv.inverse(), want, rtol=tol, atol=tol * np.abs(want).max())


def test_rank_index_insert():
    """rank_index_test.cpp:5-29"""
    r = oracle.RankIndex()
    a, b, c = ord("a"), ord("b"), ord("c")
    assert r.size() == 0
    assert r.insert(a) == 0 and r.size() == 1
    assert r.rank_of(a) == 0 and r.rank_of(c) == -1
    assert r.insert(c) == 1 and r.size() == 2
    assert r.rank_of(a) == 0 and r.rank_of(c) == 1
    assert r.insert(b) == 1 and r.size() == 3
    assert (r.rank_of(a), r.rank_of(b), r.rank_of(c)) == (0, 1, 2)
    assert r.insert(b) == 1 and r.size() == 3      # duplicate: no-op, existing rank


def test_rank_index_erase():
    """rank_index_test.cpp:31-77"""
    r = oracle.RankIndex()
    a, b, c, d, z = (ord(ch) for ch in "abcdz")
    for it in (a, d, b, c):
        r.insert(it)
    assert [r.rank_at(i) for i in range(4)] == [a, b, c, d]
    assert r.size() == 4
    assert r.erase(z) is False
    assert r.erase(b) is True
    assert r.size() == 3
    assert (r.rank_of(a), r.rank_of(b), r.rank_of(c), r.rank_of(d)) == (0, -1, 1, 2)
    assert r.erase(a) is True
    assert (r.rank_of(a), r.rank_of(b), r.rank_of(c), r.rank_of(d)) == (-1, -1, 0, 1)
    assert r.erase(d) is True
    assert (r.rank_of(c), r.rank_of(d)) == (0, -1)
    assert r.erase(c) is True
    assert r.size() == 0 and r.rank_of(c) == -1


# ------------------------------------------- (c) the reference's property tests

@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ref_smoke(dtype):
    ref_cases.smoke(o_solve, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ref_smoke_column_subset(dtype):
    ref_cases.smoke_column_subset(o_solve, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ref_noisy_signal(dtype):
    ref_cases.noisy_signal(o_solve, dtype)


@pytest.mark.parametrize("shape", [(100, 25), (25, 100)])
def test_ref_noisy_patterns(shape):
    ref_cases.noisy_patterns(o_solve, shape[0], shape[1])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cfg", [(10, 10, .1, .1, 10), (25, 10, .1, .1, 50), (10, 25, .05, .05, 50)])
def test_ref_permutations(cfg, dtype):
    M, N, sn, an, skip = cfg
    ref_cases.permutations(o_solve, M, N, dtype, sn, an, skip)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ref_binding_smoke(dtype):
    ref_cases.binding_smoke(o_solve_default, dtype)


def test_ref_binding_layouts():
    ref_cases.binding_row_subset(o_solve_default)
    ref_cases.binding_col_subset(o_solve_default)
    ref_cases.binding_transpose(o_solve_default)


# ------------------------------------------------------------- GEMV helpers

@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gemv_matches_numpy(dtype):
    rng = np.random.default_rng(9)
    A = rng.standard_normal((130, 1037)).astype(dtype)
    v = rng.standard_normal(130).astype(dtype)
    x = rng.standard_normal(1037).astype(dtype)
    tol = 2e-5 if dtype == np.float32 else 1e-12
    for view in (A, np.asfortranarray(A)):
        c = oracle.gemv_t(view, v)
        yv = oracle.gemv_n(view, x)
        assert np.allclose(c, A.astype(np.float64).T @ v, rtol=tol, atol=tol * 10)
        assert np.allclose(yv, A.astype(np.float64) @ x, rtol=tol, atol=tol * 30)
    assert np.array_equal(oracle.gemv_t(A, v), oracle.gemv_t(np.asfortranarray(A), v))
    assert np.array_equal(oracle.gemv_n(A, x), oracle.gemv_n(np.asfortranarray(A), x))


# ------------------------------------------------------------- OMP (unpinned by the reference)

@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_omp_oracle_vs_sklearn(dtype):
    """There is no OMP in the reference (include/ss/ss.h:60-64), so the oracle's OMP is
    cross-checked against scikit-learn's orthogonal_mp instead of reference fixtures."""
    from sklearn.linear_model import orthogonal_mp
    A, y, x0, sup = make_gaussian_problem(21, 128, 600, 9, dtype)
    tol = 1e-4 if dtype == np.float32 else 1e-9
    x, it, err, picks = oracle.omp(A, y, tol, 50)
    assert it == 9 and err <= tol
    assert np.array_equal(np.sort(picks), sup)
    xs = orthogonal_mp(A.astype(np.float64), y.astype(np.float64), n_nonzero_coefs=it)
    assert np.array_equal(np.nonzero(x)[0], np.nonzero(xs)[0])
    assert np.abs(x - xs).max() <= (1e-5 if dtype == np.float32 else 1e-12) * np.abs(xs).max()
    # budget-limited run: first k picks of the same greedy sequence
    x3, it3, err3, picks3 = oracle.omp(A, y, tol, 3)
    assert it3 == 3 and np.array_equal(picks3, picks[:3]) and err3 > tol
