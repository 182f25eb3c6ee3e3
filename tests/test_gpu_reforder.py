"""The reference-order engine (option "engine" = 3, csrc/reforder.hip) against the CPU oracle, BIT FOR BIT.

The oracle (oracle/ss_oracle_impl.inc) fixes the arithmetic the reference leaves to an unpinned OpenBLAS: eight
partial sums per dot product, term r to partial r & 7, combined ((0+1)+(2+3))+((4+5)+(6+7)), products and sums
separately rounded.  The reference-order engine states the same order on the device, so here nothing is
"within tolerance": coefficients, iteration counts, lambda and the whole breakpoint trace (column, insert /
remove, step length) must be identical words — on well-posed problems, on paths with removals and re-insertions
(where the fast engines may legitimately differ by rounding luck), on exact ties, on the reference's own
ill-conditioned test matrices.  It is also the arbiter the fast engines hand a signal to when their scan meets an
exact tie (option "tie_rerun"): those results are the oracle's by construction.
Nothing here reads /root/reference."""
import numpy as np
import pytest

import oracle
import ref_cases
from conftest import make_gaussian_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sship():
    import sship as mod
    assert mod.device_count() >= 1, "no HIP device visible"
    return mod


def assert_bitwise(h, A, y, tol, max_iter, flags=0, tag=None):
    """solve on the device (engine 3, trace on) and in the oracle: every word equal"""
    xo, ito, eo, tro = oracle.homotopy(A, y, tol, max_iter, flags=flags, trace=True)
    xg, itg, eg = h.solve(y, tol, max_iter)
    trg = h.trace()
    assert itg == ito, (tag, itg, ito)
    assert np.array_equal(trg["idx"], tro["idx"]), (tag, "columns of the path")
    assert np.array_equal(trg["added"], tro["added"]), (tag, "insert / remove")
    assert np.array_equal(trg["gamma"], tro["gamma"]), (tag, "step lengths")
    # (lambda: the device records it at the START of iteration t, the oracle after iteration t)
    assert np.array_equal(trg["c_inf"][1:], tro["c_inf"][:-1]), (tag, "lambda")
    assert eg == eo, (tag, eg, eo)
    assert np.array_equal(xg, xo), (tag, "coefficients", int((xg != xo).sum()))
    return xo, ito, tro


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(64, 256, 6), (128, 1000, 10), (300, 1500, 20), (512, 4096, 24), (1000, 5000, 60)])
def test_gaussian_bitwise(sship, shape, dtype):
    m, n, k = shape
    A, y, x0, sup = make_gaussian_problem(100 + m, m, n, k, dtype)
    tol = 1e-3 if dtype == np.float32 else 1e-9
    with sship.Homotopy(A) as h:
        h.set_option("engine", 3)
        h.set_option("trace", 1)
        xo, ito, _ = assert_bitwise(h, A, y, tol, 4 * k)
        assert np.array_equal(np.nonzero(np.abs(xo) > 1e-4)[0], sup)
        # and the sweep on its own: c = A^T y
        c, _ = h.gemv_t(y)
        assert np.array_equal(c, oracle.gemv_t(A, y))
        # the direct form of the sweep (no LDS staging of the dictionary): the same sums, and the same path
        h.set_option("ro_staged", 0)
        c, _ = h.gemv_t(y)
        assert np.array_equal(c, oracle.gemv_t(A, y))
        assert_bitwise(h, A, y, tol, 4 * k, tag="direct sweep")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(1, 1), (1, 5), (5, 1), (3, 200), (200, 3), (255, 127), (257, 129), (7, 300), (1023, 511)])
def test_ragged_shapes_bitwise(sship, shape, dtype):
    """row counts that are not multiples of 8 (the last partial sums get fewer terms), of the 256-row padding,
    column counts off the 256-column padding"""
    m, n = shape
    rng = np.random.default_rng(31 * m + n)
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(dtype)
    k = max(1, min(m, n) // 8)
    x0 = np.zeros(n)
    x0[rng.choice(n, k, replace=False)] = 1 + np.abs(rng.standard_normal(k))
    y = (A.astype(np.float64) @ x0).astype(dtype)
    tol = 1e-3 if dtype == np.float32 else 1e-9
    with sship.Homotopy(A) as h:
        h.set_option("engine", 3)
        h.set_option("trace", 1)
        assert_bitwise(h, A, y, tol, 3 * k + 4, tag=shape)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("flags", [0, oracle.ZERO_ON_REMOVAL | oracle.TIE_GUARD])
def test_removal_paths_bitwise(sship, dtype, flags):
    """small m relative to k: columns leave the support and come back.  In reference mode a leaving column keeps
    its rounding residue and a re-inserted one may bounce (gamma ~ 1e-18) until the budget is spent — on these
    paths the fast engines and the oracle may part by rounding luck; this engine may not, in either mode."""
    removals = reinsertions = exhausted = 0
    for seed in range(1000, 1024):
        rng = np.random.default_rng(seed)
        m, n, k = 24, 64, 10
        A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(dtype)
        x0 = np.zeros(n)
        x0[rng.choice(n, k, replace=False)] = 1 + np.abs(rng.standard_normal(k))
        y = (A.astype(np.float64) @ x0).astype(dtype)
        tol = 1e-4 if dtype == np.float32 else 1e-6
        with sship.Homotopy(A) as h:
            h.set_option("engine", 3)
            h.set_option("trace", 1)
            if flags:
                h.set_option("tie_guard", 1)
                h.set_option("zero_on_removal", 1)
            _, ito, tr = assert_bitwise(h, A, y, tol, 200, flags=flags, tag=seed)
        gone = set()
        for i, a in zip(tr["idx"], tr["added"]):
            if a == 0:
                removals += 1
                gone.add(int(i))
            elif int(i) in gone:
                reinsertions += 1
        exhausted += ito >= 200
    assert removals >= 10 and reinsertions >= 2, (removals, reinsertions, exhausted)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_one_pass_schedule_and_the_second_sweep(sship, dtype):
    """The engine runs ONE pass over A per iteration: the direction is built from the signs of c - gamma q and the
    signs of the re-computed correlations are checked afterwards (k_ro_check).  Which sweeps run is a schedule, not
    arithmetic: with every check forced to fail (the direction rebuilt, p = A d and q = A^T p formed again) the words
    are the same; and on well-posed problems no iteration needs the second sweep."""
    for m, n, k in [(128, 1000, 10), (512, 4096, 24)]:
        A, y, x0, sup = make_gaussian_problem(500 + m, m, n, k, dtype)
        tol = 1e-3 if dtype == np.float32 else 1e-9
        with sship.Homotopy(A) as h:
            h.set_option("engine", 3)
            h.set_option("trace", 1)
            h.reset_stats()
            _, ito, _ = assert_bitwise(h, A, y, tol, 4 * k, tag=("one pass", m))
            assert h.stats()["ro_resweeps"] == 0
            h.set_option("ro_force_resweep", 1)
            h.reset_stats()
            assert_bitwise(h, A, y, tol, 4 * k, tag=("forced second sweep", m))
            assert h.stats()["ro_resweeps"] == ito - 1        # (iteration 0 carries the seed's sign: no check)
    # removal paths with the second sweep forced
    for seed in range(1000, 1008):
        rng = np.random.default_rng(seed)
        m, n, k = 24, 64, 10
        A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(dtype)
        x0 = np.zeros(n)
        x0[rng.choice(n, k, replace=False)] = 1 + np.abs(rng.standard_normal(k))
        y = (A.astype(np.float64) @ x0).astype(dtype)
        with sship.Homotopy(A) as h:
            h.set_option("engine", 3)
            h.set_option("trace", 1)
            h.set_option("ro_force_resweep", 1)
            assert_bitwise(h, A, y, 1e-4 if dtype == np.float32 else 1e-6, 200, tag=("forced", seed))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lockstep_slots_bitwise(sship, dtype):
    """Batches in engine 3 run in lock-step, up to 8 (fp32) / 4 (fp64) signals per pass over A (k_ro_sweep_t<NB, NS>): every
    signal's words are those of its own solve — the oracle's — whatever the others in its group do (different
    sparsities: they finish in different rounds; a group of one at the end), dense output and compact records alike."""
    import sharding
    m, n = 128, 1000
    B = 11
    rng = np.random.default_rng(77)
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(dtype)
    Y = np.zeros((B, m), dtype)
    ks = [3, 12, 7, 1, 16, 9, 5, 14, 2, 11, 6]
    for b in range(B):
        x0 = np.zeros(n)
        x0[rng.choice(n, ks[b], replace=False)] = (1 + np.abs(rng.standard_normal(ks[b]))) * rng.choice([-1.0, 1.0], ks[b])
        Y[b] = (A.astype(np.float64) @ x0).astype(dtype)
    Y[3] *= 0                                                # a zero signal: ends at the first pick
    tol = 1e-3 if dtype == np.float32 else 1e-9
    want = [oracle.homotopy(A, Y[b], tol, 64) for b in range(B)]
    with sship.Homotopy(A) as h:
        h.set_option("engine", 3)
        for slots in (8, 6, 5, 4, 3, 1):                      # (fp64 contexts carry at most 4 per pass whatever is asked)
            h.set_option("ro_slots", slots)
            X, it, err = h.solve_batch(Y, tol, 64)
            for b in range(B):
                xo, ito, eo = want[b][:3]
                assert it[b] == ito and err[b] == eo, (slots, b, it[b], ito)
                assert np.array_equal(X[b], xo), (slots, b)
        h.set_option("ro_slots", 4)
        rec = h.solve_batch_compact(Y, tol, 64, kmax=80)
        recs = sharding.unpack_records(rec, 80, dtype)
        for b in range(B):
            xo, ito, eo = want[b][:3]
            xr = np.zeros(n, dtype)
            xr[recs[b]["idx"]] = recs[b]["val"]
            assert recs[b]["iter"] == ito and recs[b]["err"] == eo and np.array_equal(xr, xo), b
        # and with the second sweep forced in every iteration of every slot
        h.set_option("ro_force_resweep", 1)
        X, it, err = h.solve_batch(Y, tol, 64)
        for b in range(B):
            assert it[b] == want[b][1] and np.array_equal(X[b], want[b][0]), ("forced", b)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_exact_tie_bitwise_and_rerun(sship, dtype):
    """A = I, y = e_0 + e_1 + e_5 / 2: column 1 attains lambda exactly after column 0 entered, its candidate is
    t = 0 and the strict t > 0 of the reference skips it for good.  Engine 3 follows the oracle word for word;
    the fast engines notice the exact tie, hand the signal over (tie_reruns) and return the same words."""
    n = 32
    A = np.eye(n, dtype=dtype)
    y = np.zeros(n, dtype=dtype)
    y[0] = y[1] = 1.0
    y[5] = 0.5
    with sship.Homotopy(A) as h:
        h.set_option("trace", 1)
        h.set_option("engine", 3)
        xo, ito, tro = assert_bitwise(h, A, y, 1e-3, 6)
        assert ito == 6 and xo[1] == 0.0
        for engine in (0, 1):
            h.set_option("engine", engine)
            h.reset_stats()
            xg, itg, eg = h.solve(y, 1e-3, 6)
            assert h.stats()["tie_reruns"] == 1, engine
            assert itg == ito and np.array_equal(xg, xo), engine
            assert np.array_equal(h.trace()["idx"], tro["idx"]), engine
            # switched off: the engine follows its own path and nothing is re-run
            h.set_option("tie_rerun", 0)
            h.reset_stats()
            h.solve(y, 1e-3, 6)
            assert h.stats()["tie_reruns"] == 0
            h.set_option("tie_rerun", 1)
        # a batch: the tied signal is re-run, its neighbours are not
        Y = np.stack([y, np.roll(y, 7) * dtype(0.75) + np.roll(y, 3) * dtype(0.125), y])
        Y[1, 20] = 2.0
        h.set_option("engine", 1)
        h.reset_stats()
        X, iters, errs = h.solve_batch(Y, 1e-3, 6)
        for b in range(3):
            xb, itb, eb = oracle.homotopy(A, Y[b], 1e-3, 6)
            assert itb == iters[b] and np.array_equal(X[b], xb), b
        if dtype == np.float32:
            # the same through the batched Gram forms (G = A^T A; the subset form, then the lock-step form): the tie is met in
            # the FIRST round there (lambda is still the first pick's), flagged, and the signal re-run
            h.set_option("batch_min", 2)
            h.set_option("batch_gram_min", 2)
            for subset in (1, 0):
                h.set_option("batch_subset", subset)
                h.reset_stats()
                X, iters, errs = h.solve_batch(Y, 1e-3, 6)
                st = h.stats()
                assert st["tie_reruns"] == 3, (subset, st["tie_reruns"])     # (the middle signal has an exact tie of its own: 0.75, 0.75)
                assert (st["subset_signals"] > 0) == bool(subset)
                for b in range(3):
                    xb, itb, eb = oracle.homotopy(A, Y[b], 1e-3, 6)
                    assert itb == iters[b] and np.array_equal(X[b], xb), (subset, b)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cfg", [(10, 10, .1, .1, 10), (25, 10, .1, .1, 50), (10, 25, .05, .05, 50)])
def test_ref_permutations_bitwise(sship, cfg, dtype):
    """the reference's `permutations` matrices (test_util.h:204-257; cond ~ 800, ||A^T y|| ~ 5e3): rounding-chaotic
    paths in fp32 — the case the Gram-form guard exists for — must still be the oracle's words, and the
    reference's property (argmax x = the planted column) must hold"""
    M, N, sn, an, skip = cfg
    checked = []

    def solve(A, y, tol, max_iter):
        with sship.Homotopy(A) as h:
            h.set_option("engine", 3)
            h.set_option("trace", 1)
            y = np.ascontiguousarray(y, dtype=A.dtype)
            assert_bitwise(h, A, y, tol, max_iter, tag=cfg)
            checked.append(1)
            return h.solve(y, tol, max_iter)

    ref_cases.permutations(solve, M, N, dtype, sn, an, skip)
    assert len(checked) == N


def test_layouts_and_strides_bitwise(sship):
    """padded row-major, column-major and device-resident matrices, strided y / x: the device copy is the same"""
    import torch
    A, y, _, _ = make_gaussian_problem(7, 96, 700, 9, np.float32)
    xo, ito, eo = oracle.homotopy(A, y, 1e-3, 60)
    padded = np.zeros((96, 760), dtype=np.float32)
    padded[:, 30:730] = A
    for v in (padded[:, 30:730], np.asfortranarray(A), torch.from_numpy(A).to("cuda:0")):
        with sship.Homotopy(v) as h:
            h.set_option("engine", 3)
            xg, itg, eg = h.solve(y, 1e-3, 60)
        assert itg == ito and eg == eo and np.array_equal(xg, xo)


def test_mid_size_sweep_bitwise_and_timing(sship):
    """m = 8192 (the configs[1] row count), 4096 columns: the sweep alone and a bounded solve"""
    A, y, _, _ = make_gaussian_problem(11, 8192, 4096, 32, np.float32)
    with sship.Homotopy(A) as h:
        h.set_option("engine", 3)
        c, ms = h.gemv_t(y, repeats=3)
        assert np.array_equal(c, oracle.gemv_t(A, y))
        h.set_option("trace", 1)
        assert_bitwise(h, A, y, 1e-3, 12)
