"""The reference's own test-suite for the Homotopy path, restated as
solver-agnostic checks.  Every function takes `solve(A, y, tol, max_iter) ->
(x, iter, solution_error)` so that the same cases run against the CPU oracle
(tests/test_oracle.py) and against the HIP path through the C-ABI
(tests/test_gpu_parity.py).

Source of each case (paths relative to /root/reference):
  smoke / column-subset / noisy-signal / noisy-patterns / permutations
      src/solvers/test_util.h:27-257, src/solvers/homotopy_test.cpp:8-61
  binding cases   bindings/python/tests/test_binding.py:9-68
The reference seeds xtensor's RNG (not reproducible outside xtensor); the cases
here use numpy's default_rng with fixed seeds and assert the same properties.
"""
import numpy as np


def check_report(iters, err, tol, max_iter):
    """homotopy_test.cpp:8-21"""
    assert 1 <= iters <= max_iter
    if iters < max_iter:
        assert err <= tol


def smoke(solve, dtype):
    """test_util.h:27-55: A = I5, y = e_n  =>  x == y exactly."""
    N = 5
    A = np.eye(N, dtype=dtype)
    for n in range(N):
        y = np.zeros(N, dtype=dtype)
        y[n] = 1
        x, it, err = solve(A, y, 0.001, N)
        check_report(it, err, 0.001, N)
        assert np.array_equal(x, y)


def smoke_column_subset(solve, dtype):
    """test_util.h:57-92: solver on a strided view (columns 5..9 of a 5x10 buffer, lda = 10)."""
    M, N = 5, 10
    rng = np.random.default_rng(0)
    data = np.zeros((M, N), dtype=dtype)
    data[:, 0:M - 1] = rng.uniform(0.0, 0.1, size=(M, M - 1))
    data[:, M:N] = np.eye(M, dtype=dtype)
    ident = data[:, M:N]
    assert ident.strides[0] == N * data.itemsize
    for n in range(M):
        y = np.ascontiguousarray(ident[:, n])
        x, it, err = solve(ident, y, 0.001, N)
        assert np.array_equal(x, y)


def noisy_signal(solve, dtype):
    """test_util.h:94-126: A = I50, noise level == tolerance => exactly one entry above it."""
    N, NOISE = 50, 0.01
    rng = np.random.default_rng(0)
    A = np.eye(N, dtype=dtype)
    for n in range(N):
        y = rng.uniform(0.0, NOISE, size=N).astype(dtype)
        y[n] += dtype(1.0 - 0.5 * NOISE)
        x, it, err = solve(A, y, NOISE, N)
        check_report(it, err, NOISE, N)
        assert int((x > NOISE).sum()) == 1


def noisy_patterns(solve, M, N, dtype=np.float32, noise_level=0.1, signal_level=1.0, cols=None):
    """test_util.h:136-197 (homotopy_test.cpp:42-46 runs 100x25 and 25x100, f32)."""
    ERROR = 0.1 * noise_level
    rng = np.random.default_rng(0)
    noise = rng.normal(0.5, noise_level, size=(M, N)).astype(dtype)
    signal = rng.normal(0.5, noise_level, size=M).astype(dtype)
    signal[0::2] += dtype(signal_level)
    signal /= np.abs(signal).sum()
    for n in (range(N) if cols is None else cols):
        hay = noise.copy()
        hay[0::2, n] = signal_level
        hay /= np.abs(hay).sum(axis=0)          # ss::norm_l1, norms.h:22-27
        x, it, err = solve(hay, signal, ERROR, N)
        check_report(it, err, ERROR, N)
        assert int(np.argmax(x)) == n
        assert int((x > ERROR).sum()) == 1
        recon = hay.astype(np.float64) @ x.astype(np.float64)   # ss::reconstruct_signal
        assert np.allclose(recon, signal, rtol=0.0, atol=5 * ERROR)


def _next_permutation(v):
    """std::next_permutation on a list (in place); returns False on wrap-around."""
    i = len(v) - 2
    while i >= 0 and v[i] >= v[i + 1]:
        i -= 1
    if i < 0:
        v.reverse()
        return False
    j = len(v) - 1
    while v[j] <= v[i]:
        j -= 1
    v[i], v[j] = v[j], v[i]
    v[i + 1:] = reversed(v[i + 1:])
    return True


def _permute(v, n):
    for _ in range(n):
        _next_permutation(v)


def permutations(solve, M, N, dtype, signal_noise, sensing_noise, skip, seed=0):
    """test_util.h:204-257 (homotopy_test.cpp:48-61, irls_test.cpp:40-53).  The property holds for
    most but not all noise draws (the reference fixes xtensor's seed); `seed` picks the numpy draw."""
    rng = np.random.default_rng(seed)
    ERROR = signal_noise + sensing_noise
    colbuff = [float(i) for i in range(1, M + 1)]
    _permute(colbuff, skip)
    A = rng.normal(0.0, sensing_noise, size=(M, N)).astype(dtype)
    col = list(colbuff)
    for n in range(N):
        A[:, n] += np.asarray(col, dtype=dtype)
        _permute(col, skip)
    for n in range(N):
        y = (np.asarray(colbuff) + rng.normal(0.0, signal_noise, size=M)).astype(dtype)
        x, it, err = solve(A, y, ERROR, N)
        check_report(it, err, ERROR, N)
        assert int(np.argmax(x)) == n, (n, x)
        _permute(colbuff, skip)


# ---- bindings/python/tests/test_binding.py ---------------------------------------

def binding_smoke(solve_default, dtype):
    """test_binding.py:9-20 with the binding's DEFAULT tolerance (eps*10) and max_iterations (100)."""
    N = 5
    A = np.identity(N, dtype=dtype)
    for n in range(N - 1):
        y = np.zeros(N, dtype=dtype)
        y[n] = 1
        x, it, err = solve_default(A, y)
        assert np.array_equal(x, y)
        assert err == 0
        assert it == 1


def binding_row_subset(solve_default):
    """test_binding.py:31-42"""
    rng = np.random.default_rng(1)
    A = rng.random((10, 5)) * 0.1
    A_sub = A[:5, :]
    A_sub[:, 0] = 1
    x, it, err = solve_default(A_sub, np.ones(5))
    assert len(x) == 5
    assert np.count_nonzero(x) == 1


def binding_col_subset(solve_default):
    """test_binding.py:44-56: A[:, 2:] is row-major with lda > n."""
    rng = np.random.default_rng(2)
    A = rng.random((10, 5)) * 0.1
    A[:, 0] = 1
    A[:, 3] = 1
    A_sub = A[:, 2:]
    x, it, err = solve_default(A_sub, np.ones(10))
    assert len(x) == 3
    assert int(np.argmax(x)) == 1


def binding_transpose(solve_default):
    """test_binding.py:58-68: A.T is column-major (stride0 == 1)."""
    rng = np.random.default_rng(3)
    A = rng.random((5, 10)) * 0.1
    A[3, :] = 1
    x, it, err = solve_default(A.T, np.ones(10))
    assert len(x) == 5
    assert int(np.argmax(x)) == 3
