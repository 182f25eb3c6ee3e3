"""GPU tests of the screened form of one fp32 signal (csrc/screen.hip; run with `-m gpu`): the subset solve on the subset's
own Gram matrix, every state of its path certified against all columns by one pass over the fp16 copy of A.

What must hold: (1) a certified signal is the oracle's — equal iterations, identical support, coefficients within 1e-5,
the same breakpoints; (2) a signal the form cannot certify, or whose path leaves the form's common path, is solved again by
the default engine and is THAT engine's result bit for bit; (3) the certificate is a bound, not a guess: the fp16 product
plus its error term really dominates the fp32 correlation of every column and state.
Nothing here reads /root/reference.
"""
import numpy as np
import pytest

import oracle
from conftest import make_gaussian_problem, note
from test_gpu_parity import assert_parity, significant_support, MODES, set_mode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sship():
    import sship as mod
    assert mod.device_count() >= 1, "no HIP device visible"
    return mod


SHAPES = [(1024, 8192, 16), (1024, 8192, 40), (768, 2048, 20), (2048, 16384, 48), (4096, 8192, 64), (1536, 6000, 30),
          (1024, 1000, 12)]
CERTIFIED_SHAPES = [(1024, 8192, 16), (4096, 8192, 64), (1536, 6000, 30)]


@pytest.mark.parametrize("first16", [2, 1, 0])
@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("shape", SHAPES)
def test_screened_form_vs_oracle(sship, shape, mode, first16):
    """option screen_single = 2 (any shape the form can run on): well-posed Gaussian problems are certified and equal the
    oracle — iterations, support, coefficients, breakpoints — in both modes, with the first pass (A^T y) over the fp8 copy
    (2: option screen_first8, the default where the padded row count is a multiple of 1024), over the half-precision copy
    (1: option screen_first16, multiples of 512) and over the fp32 dictionary (0)."""
    m, n, k = shape
    A, y, x0, sup = make_gaussian_problem(9100 + m + k, m, n, k, np.float32)
    with sship.Homotopy(A) as h:
        flags = set_mode(h, mode)
        h.set_option("screen_single", 2)
        h.set_option("screen_first16", 1 if first16 else 0)
        h.set_option("screen_first8", 1 if first16 == 2 else 0)
        h.set_option("trace", 1)
        h.reset_stats()
        xg, itg, eg = h.solve(y, 1e-3, 4 * k)
        trg = h.trace()
        st = h.stats()
    xo, ito, eo, tro = oracle.homotopy(A, y, 1e-3, 4 * k, flags=flags, trace=True)
    assert st["screen_signals"] + st["screen_redone"] == 1
    why = {k_: v for k_, v in st.items() if k_.startswith("why_") and v}
    note("test_screened_form_vs_oracle", shape=list(shape), mode=mode, certified=st["screen_signals"], resident=st["screen_resident"],
         headroom=st["screen_headroom"], why=why)
    # well-posed shapes the form MUST certify (a form that always handed back would pass everything else here), by the resident kernel
    if shape in CERTIFIED_SHAPES:
        assert st["screen_signals"] == 1 and st["screen_resident"] == 1 and not why, (shape, st["screen_headroom"], why)
    assert_parity(xg, itg, eg, xo, ito, eo, np.float32)
    assert np.array_equal(significant_support(xg, 1e-4), sup)
    assert np.array_equal(trg["idx"][:-1], tro["idx"][:-1])
    assert np.array_equal(trg["added"][:-1], tro["added"][:-1])
    assert np.allclose(trg["gamma"][:-1], tro["gamma"][:-1], rtol=1e-3, atol=1e-6)
    if st["screen_signals"]:
        assert 0.0 < st["screen_headroom"] < 1.0


def test_screened_form_hands_back_what_it_cannot_certify(sship):
    """Signals the form must not report: a support column outside the 448 largest |c0| (dense support in few rows), paths
    with removals in reference mode, more columns than the form holds.  Each is solved again by the default engine and is
    that engine's result bit for bit."""
    cases = [(400, 9000, 30, 0.0), (512, 4096, 100, 0.0), (300, 2000, 40, 0.05), (1024, 8192, 80, 0.0)]
    redone = 0
    for ci, (m, n, k, noise) in enumerate(cases):
        rng = np.random.default_rng(9200 + ci)
        A, y, x0, sup = make_gaussian_problem(9200 + ci, m, n, k, np.float32)
        if noise:
            y = (y + noise * rng.standard_normal(m)).astype(np.float32)
        with sship.Homotopy(A) as h:
            h.set_option("screen_single", 2)
            h.set_option("screen_first16", ci % 2)
            h.reset_stats()
            x1, it1, e1 = h.solve(y, 1e-3, 3 * k)
            st = h.stats()
            h.set_option("screen_single", 0)
            x0_, it0, e0 = h.solve(y, 1e-3, 3 * k)
        assert st["screen_signals"] + st["screen_redone"] == 1
        redone += st["screen_redone"]
        if st["screen_redone"]:
            assert it1 == it0 and e1 == e0 and np.array_equal(x1, x0_), (m, n, k, "a handed-back signal is not the default engine's")
        else:
            xo, ito, eo = oracle.homotopy(A, y, 1e-3, 3 * k)
            assert_parity(x1, it1, e1, xo, ito, eo, np.float32)
    note("test_screened_form_hands_back", redone=redone, cases=len(cases))
    assert redone >= 2


@pytest.mark.parametrize("first16", [1, 0])
def test_screened_form_crowded_first_state(sship, first16):
    """More than 448 columns within a few percent of lambda_0 (600 noisy copies of one atom): whatever subset is chosen,
    columns left out reach the bound of state 0 — with the first pass in half precision that is the state-0 certificate
    (T + eps_0 against 0.875 lambda_0), with the fp32 first pass the certificate of state 1 (or the subset's path meets a tie
    of its own view first).  The signal goes back to the default engine and is its result bit for bit."""
    m, n, k = 1024, 8192, 12
    rng = np.random.default_rng(9250)
    A, y, x0, sup = make_gaussian_problem(9250, m, n, k, np.float32)
    u = A[:, 7].copy()
    A[:, 1000:1600] = u[:, None] + (2e-3 / np.sqrt(m)) * rng.standard_normal((m, 600)).astype(np.float32)
    y = (y + 6.0 * u).astype(np.float32)
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        h.set_option("screen_first16", first16)
        h.reset_stats()
        x1, it1, e1 = h.solve(y, 1e-3, 3 * k)
        st = h.stats()
        h.set_option("screen_single", 0)
        x0_, it0, e0 = h.solve(y, 1e-3, 3 * k)
    note("test_screened_form_crowded_first_state", first16=first16, certified=st["screen_signals"], redone=st["screen_redone"])
    assert st["screen_signals"] == 0 and st["screen_redone"] == 1
    assert it1 == it0 and e1 == e0 and np.array_equal(x1, x0_)


def test_screened_form_on_a_context_with_gram_matrix(sship):
    """A context that holds G = A^T A (here: option gram_full_after) used to send single signals through the subset form on G;
    with the half-precision first pass the screened form is the faster of the two and takes them (option screen_single = 0
    leaves them to the form on G).  Either way the oracle's result."""
    m, n, k = 1024, 8192, 20
    A, y, x0, sup = make_gaussian_problem(9260, m, n, k, np.float32)
    xo, ito, eo = oracle.homotopy(A, y, 1e-3, 4 * k)
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 0)
        h.set_option("gram_full_after", 1)
        x0_, it0, e0 = h.solve(y, 1e-3, 4 * k)
        assert h.stats()["gram_full_builds"] == 1
        h.reset_stats()
        xg, itg, eg = h.solve(y, 1e-3, 4 * k)                     # the subset form on G
        stg = h.stats()
        h.set_option("screen_single", 2)
        h.reset_stats()
        xs, its, es = h.solve(y, 1e-3, 4 * k)                     # the screened form, G or not
        sts = h.stats()
    assert stg["subset_signals"] == 1 and stg["screen_signals"] == 0
    assert sts["screen_signals"] == 1 and sts["subset_signals"] == 0
    for (x_, it_, e_) in ((x0_, it0, e0), (xg, itg, eg), (xs, its, es)):
        assert_parity(x_, it_, e_, xo, ito, eo, np.float32)
        assert np.array_equal(significant_support(x_, 1e-4), sup)


def test_screen_certificate_is_a_bound(sship):
    """The certificate of csrc/screen.hip, recomputed on the host in float64 from the solve's own breakpoints: for every state
    of the path and every column outside the 448-column subset, |a_i . r_k| in float64 must be below lambda_k by at least
    the margin whenever the device certified the signal — the fp16 pass plus its error term may be loose, never wrong."""
    m, n, k = 2048, 16384, 40
    A, y, x0, sup = make_gaussian_problem(9300, m, n, k, np.float32)
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        h.set_option("trace", 1)
        h.reset_stats()
        xg, itg, eg = h.solve(y, 1e-3, 4 * k)
        tr = h.trace()
        st = h.stats()
        c0, _ = h.gemv_t(y)
    assert st["screen_signals"] == 1
    A64 = A.astype(np.float64)
    sub = np.sort(np.argsort(-np.abs(c0), kind="stable")[:448])
    outside = np.ones(n, dtype=bool)
    outside[sub] = False
    # replay the path in float64 from the trace: x after every breakpoint
    x = np.zeros(n)
    S = []
    worst = 0.0
    for t in range(1, len(tr["idx"])):
        # state before toggle t: lambda = c_inf[t], support S (entry 0 is the first pick)
        if t == 1:
            S = [int(tr["idx"][0])]
        As = A64[:, S]
        r = y.astype(np.float64) - As @ x[S]
        c = A64.T @ r
        lam = float(tr["c_inf"][t])
        if t > 1:
            worst = max(worst, float(np.abs(c[outside]).max() / lam))
            assert np.abs(c[outside]).max() <= 0.875 * lam + 1e-6
        # the step: direction on S from the float64 normal equations, length gamma[t]
        sg = np.sign(c[S])
        d = np.linalg.solve(As.T @ As, sg)
        x[S] += float(tr["gamma"][t]) * d
        j = int(tr["idx"][t])
        if tr["added"][t]:
            S.append(j)
        else:
            S.remove(j)
    note("test_screen_certificate_is_a_bound", worst_ratio=worst, headroom=st["screen_headroom"])
    assert worst <= st["screen_headroom"] + 1e-3         # the device's figure (|c~| + eps) / bound dominates |c| / (0.875 lambda)... loosely


@pytest.mark.parametrize("first16", [2, 1, 0])
@pytest.mark.parametrize("shape", [(1024, 16384, 24), (1536, 9000, 40), (2048, 16384, 60), (1024, 12000, 16), (4096, 16384, 104)])
def test_screened_form_fp64_vs_oracle(sship, shape, first16):
    """fp64: the path is solved by the fp64 engine on a sub-dictionary (the 2048 columns with the largest |c0| — ranked by the
    half-precision first pass, option screen_first16, or by the fp64 sweep —, a context of its own, every state logged) and
    certified against all columns by the fp16 pass.  Certified signals equal the oracle within
    1e-10; what the form hands back (a support column outside the sub-dictionary: the sub-solve wanders) is the default
    engine's result bit for bit.  (The last shape logs 97 .. 128 states: the four-tile form of the screening pass.)"""
    m, n, k = shape
    A, y, x0, sup = make_gaussian_problem(9400 + m + k, m, n, k, np.float64)
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        h.set_option("screen_first16", 1 if first16 else 0)      # (2: the ranking pass over the fp8 copy, 1: over the fp16 copy, 0: the fp64 sweep)
        h.set_option("screen_first8", 1 if first16 == 2 else 0)
        h.reset_stats()
        xg, itg, eg = h.solve(y, 1e-9, 4 * k)
        st = h.stats()
        h.set_option("screen_single", 0)
        xd, itd, ed = h.solve(y, 1e-9, 4 * k)
    xo, ito, eo = oracle.homotopy(A, y, 1e-9, 4 * k)
    note("test_screened_form_fp64_vs_oracle", shape=list(shape), certified=st["screen_signals"], redone=st["screen_redone"],
         headroom=st["screen_headroom"])
    assert st["screen_signals"] + st["screen_redone"] == 1
    assert_parity(xg, itg, eg, xo, ito, eo, np.float64)
    if st["screen_redone"]:
        assert itg == itd and eg == ed and np.array_equal(xg, xd)
    else:
        assert 0.0 < st["screen_headroom"] < 1.0
        assert np.abs(xg - xd).max() <= 1e-12 * np.abs(xd).max()


def test_screened_form_fp64_hands_back(sship):
    """A support column outside the 2048-column sub-dictionary: the sub-context's solve wanders beyond the states the form can
    certify and the signal is handed back — the default engine's result, bit for bit, and the oracle's."""
    m, n, k = 2048, 16384, 60
    rng = np.random.default_rng(5002)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    sup = np.sort(rng.choice(n, k, replace=False))
    x0 = np.zeros(n)
    x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
    y = A @ x0
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        h.set_option("screen_rescue", 0)                  # (the rescue would find the column and certify: test_rescue_in_the_fp64_resident_tier)
        h.reset_stats()
        xg, itg, eg = h.solve(y, 1e-9, 4 * k)
        st = h.stats()
        h.set_option("screen_rescue", 1)
        h.reset_stats()
        xr, itr, er = h.solve(y, 1e-9, 4 * k)
        str_ = h.stats()
        h.set_option("screen_single", 0)
        xd, itd, ed = h.solve(y, 1e-9, 4 * k)
    xo, ito, eo = oracle.homotopy(A, y, 1e-9, 4 * k)
    note("test_screened_form_fp64_hands_back", certified=st["screen_signals"], redone=st["screen_redone"], with_rescue=dict(certified=str_["screen_signals"],
         rescued=str_["screen_rescued"], redone=str_["screen_redone"]))
    assert st["screen_signals"] + st["screen_redone"] == 1
    assert itg == itd and eg == ed and np.array_equal(xg, xd)
    assert_parity(xg, itg, eg, xo, ito, eo, np.float64)
    assert str_["screen_signals"] + str_["screen_redone"] == 1
    assert_parity(xr, itr, er, xo, ito, eo, np.float64)


@pytest.mark.parametrize("first16", [1, 0])
@pytest.mark.parametrize("shape", [(1024, 16384, 24), (1536, 9000, 40), (2048, 16384, 60)])
def test_screened_form_fp64_omp(sship, shape, first16):
    """OMP (ss::omp<double>; no reference implementation: pinned against this library's default OMP engine and numpy's least
    squares on the planted support) through the fp64 screened form: the sub-context runs k_la_omp, the certificate is that no
    column outside the sub-dictionary reaches the pick's |c|."""
    m, n, k = shape
    seed = {24: 1, 40: 3, 60: 2}[k]
    rng = np.random.default_rng(5000 + seed)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    sup = np.sort(rng.choice(n, k, replace=False))
    x0 = np.zeros(n)
    x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
    y = A @ x0
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        h.set_option("screen_first16", first16)
        h.reset_stats()
        xg, itg, eg = h.solve_omp(y, 1e-9, 4 * k)
        st = h.stats()
        h.set_option("screen_single", 0)
        xd, itd, ed = h.solve_omp(y, 1e-9, 4 * k)
    note("test_screened_form_fp64_omp", shape=list(shape), certified=st["screen_signals"], redone=st["screen_redone"], headroom=st["screen_headroom"])
    assert st["screen_signals"] + st["screen_redone"] == 1
    assert itg == itd == k and np.array_equal(np.nonzero(xg)[0], sup)
    if st["screen_redone"]:
        assert eg == ed and np.array_equal(xg, xd)
    else:
        assert np.abs(xg - xd).max() <= 1e-12 * np.abs(xd).max()
    ls = np.linalg.lstsq(A[:, sup], y, rcond=None)[0]
    assert np.abs(xg[sup] - ls).max() <= 1e-10 * np.abs(ls).max()


@pytest.mark.parametrize("first16", [2, 0])
@pytest.mark.parametrize("shape", [(1024, 8192, 16), (2048, 16384, 48), (1536, 6000, 30)])
def test_screened_form_fp32_omp(sship, shape, first16):
    """OMP in fp32 (ss::omp<float>; no reference implementation: pinned against this library's default OMP engine and numpy's least
    squares on the planted support) through the screened form: k_res_solve<float, OMP> on the 448 best-ranked columns, the certificate
    is that no column outside the subset reaches the pick's |c| at any state."""
    m, n, k = shape
    rng = np.random.default_rng(5100 + k)
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float32)
    sup = np.sort(rng.choice(n, k, replace=False))
    x0 = np.zeros(n, np.float32)
    x0[sup] = (1.0 + np.abs(rng.standard_normal(k))).astype(np.float32)
    y = (A.astype(np.float64) @ x0).astype(np.float32)
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        h.set_option("screen_first16", 1 if first16 else 0)
        h.reset_stats()
        xg, itg, eg = h.solve_omp(y, 1e-3, 4 * k)
        st = h.stats()
        h.set_option("screen_single", 0)
        xd, itd, ed = h.solve_omp(y, 1e-3, 4 * k)
    note("test_screened_form_fp32_omp", shape=list(shape), first=first16, certified=st["screen_signals"], redone=st["screen_redone"],
         headroom=st["screen_headroom"], why={k_: v for k_, v in st.items() if k_.startswith("why_") and v})
    assert st["screen_signals"] + st["screen_redone"] == 1
    assert st["screen_signals"] == 1 and st["screen_resident"] == 1, "a well-posed OMP problem must be certified"
    assert itg == itd == k and np.array_equal(np.nonzero(xg)[0], sup)
    assert np.abs(xg - xd).max() <= 2e-5 * np.abs(xd).max()
    ls = np.linalg.lstsq(A[:, sup].astype(np.float64), y.astype(np.float64), rcond=None)[0]
    assert np.abs(xg[sup] - ls).max() <= 2e-5 * np.abs(ls).max()


def test_screened_form_random_problems(sship):
    """tools/stress_screen.py in small: random shapes, fp32 and fp64, signed coefficients (the reference's first-step sign quirk
    derails those paths), noise, both modes.  Whatever the form certifies must be the oracle's; whatever it hands back must be the
    default engine's bit for bit.  (This is the test that found the two holes of the first certificate: a bound against max |c|
    instead of lambda_prev - gamma_prev on derailed paths, and the uncertified last regular step of a noisy path.)"""
    rng = np.random.default_rng(20261004)
    cert = redone = 0
    for case in range(45):
        f64 = case % 3 == 2
        dt = np.float64 if f64 else np.float32
        m = int(rng.choice([512, 768, 1024, 1536, 2048]))
        n = int(rng.choice([8192, 12000, 16384] if f64 else [2048, 4096, 8192, 16384]))
        k = int(rng.integers(4, max(5, m // 24)))
        signed = bool(rng.integers(0, 2))
        noise = float(rng.choice([0.0, 0.0, 1e-4, 1e-2]))
        fixes = bool(rng.integers(0, 2))
        A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(dt)
        sup = np.sort(rng.choice(n, k, replace=False))
        x0 = np.zeros(n)
        x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
        if signed:
            x0[sup] *= rng.choice([-1.0, 1.0], k)
        y = A.astype(np.float64) @ x0
        if noise:
            y = y + noise * rng.standard_normal(m)
        y = y.astype(dt)
        tol = 1e-9 if f64 else 1e-3
        budget = 3 * k + 8
        flags = oracle.SPARSE_NOTRANS | ((oracle.ZERO_ON_REMOVAL | oracle.TIE_GUARD) if fixes else 0)
        with sship.Homotopy(A) as h:
            if fixes:
                h.set_option("tie_guard", 1)
                h.set_option("zero_on_removal", 1)
            h.set_option("screen_single", 2)
            h.reset_stats()
            x, it, e = h.solve(y, tol, budget)
            st = h.stats()
            h.set_option("screen_single", 0)
            xd, itd, ed = h.solve(y, tol, budget)
        tag = (case, m, n, k, signed, noise, fixes)
        assert st["screen_signals"] + st["screen_redone"] == 1, tag
        if st["screen_redone"]:
            redone += 1
            assert it == itd and np.array_equal(x, xd), tag
            continue
        cert += 1
        xo, ito, eo = oracle.homotopy(A, y, tol, budget, flags=flags)
        lim = 1e-10 if f64 else 1e-5
        reld = np.abs(xd.astype(np.float64) - xo).max() / np.abs(xo).max()
        assert it == ito, tag
        assert np.array_equal(significant_support(x, 100 * lim), significant_support(xo, 100 * lim)), tag
        assert np.abs(x.astype(np.float64) - xo).max() <= max(lim, 3 * reld) * np.abs(xo).max(), tag
        # (headroom >= 1: the half-precision certificate left columns open and the exact re-check decided them)
        assert 0.0 < st["screen_headroom"] < 1.0 or st["screen_recheck"] == 1, tag
    note("test_screened_form_random_problems", certified=cert, handed_back=redone)
    assert cert >= 8 and redone >= 8


@pytest.mark.parametrize("n", [98304, 140000])
def test_screened_form_wide_dictionary(sship, n):
    """More than 65536 columns: the one-slot selection walks c0 in chunks (k_sub_select1w).  Against the oracle; and on a
    dictionary with 3000 identical columns among the largest |c0| — thousands of equal magnitudes at the selection's threshold:
    its ordered walk — against the default engine."""
    m, k = 1024, 14
    A, y, x0, sup = make_gaussian_problem(9500 + n % 1000, m, n, k, np.float32)
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        h.reset_stats()
        xg, itg, eg = h.solve(y, 1e-3, 4 * k)
        st = h.stats()
    xo, ito, eo = oracle.homotopy(A, y, 1e-3, 4 * k)
    assert st["screen_signals"] + st["screen_redone"] == 1
    assert_parity(xg, itg, eg, xo, ito, eo, np.float32)
    note("test_screened_form_wide_dictionary", n=n, certified=st["screen_signals"], headroom=st["screen_headroom"])
    if n == 98304:
        assert st["screen_signals"] == 1           # (a well-posed wide dictionary must be certified, not merely handed back correctly)
    # 3000 copies of one column whose |c0| sits between the support's and the noise's
    rng = np.random.default_rng(n)
    dup = np.sort(rng.choice(n, 3000, replace=False))
    dup = dup[~np.isin(dup, sup)]
    v = rng.standard_normal(m).astype(np.float32)
    v /= np.linalg.norm(v)
    c0 = A.T @ y
    target = 0.5 * (np.sort(np.abs(c0[sup]))[0] + np.sort(np.abs(c0))[-k - 1])      # between the weakest support column and the strongest other
    r = y / np.linalg.norm(y)
    w = v - (v @ r) * r
    w /= np.linalg.norm(w)
    a = float(target / np.linalg.norm(y))
    col = (a * r + np.sqrt(max(0.0, 1.0 - a * a)) * w).astype(np.float32)
    A2 = A.copy()
    A2[:, dup] = col[:, None]
    with sship.Homotopy(A2) as h:
        h.set_option("screen_single", 2)
        h.reset_stats()
        x1, it1, e1 = h.solve(y, 1e-3, 4 * k)
        st2 = h.stats()
        h.set_option("screen_single", 0)
        x0_, it0, e0 = h.solve(y, 1e-3, 4 * k)
    assert st2["screen_signals"] + st2["screen_redone"] == 1
    if st2["screen_redone"]:
        assert it1 == it0 and np.array_equal(x1, x0_)
    else:
        assert it1 == it0 and np.array_equal(np.nonzero(x1)[0], np.nonzero(x0_)[0])
        assert np.abs(x1 - x0_).max() <= 1e-4 * np.abs(x0_).max()


@pytest.mark.parametrize("B", [5, 40, 70])
def test_screened_batch_form(sship, B):
    """Batches without G in the screened form (option batch_screen, chunks of 64, four slots per workgroup of the screening
    launch; B not a multiple of four, B beyond one chunk): every signal against the oracle; a signal the form hands back
    (here: one whose support is too dense for the subset) is the default engine's."""
    m, n, k = 1024, 8192, 12
    rng = np.random.default_rng(9600 + B)
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float32)
    Y = np.empty((B, m), np.float32)
    sups = []
    for b in range(B):
        kb = k if b != 3 else 90                       # (signal 3: more columns than the form holds)
        sup = np.sort(rng.choice(n, kb, replace=False))
        x0 = np.zeros(n)
        x0[sup] = 1.0 + np.abs(rng.standard_normal(kb))
        Y[b] = (A.astype(np.float64) @ x0).astype(np.float32)
        sups.append(sup)
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        h.reset_stats()
        X, its, errs = h.solve_batch(Y, 1e-3, 200)
        st = h.stats()
        h.set_option("batch_screen", 0)
        h.set_option("screen_single", 0)
        x3, it3, e3 = h.solve(Y[3], 1e-3, 200)
    note("test_screened_batch_form", B=B, certified=st["screen_signals"], redone=st["screen_redone"], headroom=st["screen_headroom"])
    assert st["screen_signals"] + st["screen_redone"] + st["tie_reruns"] == B and st["screen_redone"] >= 1
    assert its[3] == it3 and np.array_equal(X[3], x3)
    for b in range(B):
        if b == 3:
            continue
        xo, ito, eo = oracle.homotopy(A, Y[b], 1e-3, 200)
        assert_parity(X[b], int(its[b]), float(errs[b]), xo, ito, eo, np.float32)
        assert np.array_equal(significant_support(X[b], 1e-4), sups[b])


@pytest.mark.parametrize("B", [3, 9])
def test_screened_form_compact_records_and_strides(sship, B):
    """The compact records {K, iter, err, idx, val} and strided host vectors through the screened forms (B = 3: one screened
    solve per signal; B = 9: the screened batch form) carry what the dense solves return."""
    import sharding
    m, n, k = 1024, 8192, 10
    rng = np.random.default_rng(9700 + B)
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float32)
    Y = np.empty((B, m), np.float32)
    for b in range(B):
        sup = np.sort(rng.choice(n, k, replace=False))
        x0 = np.zeros(n)
        x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
        Y[b] = (A.astype(np.float64) @ x0).astype(np.float32)
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        h.reset_stats()
        rec = h.solve_batch_compact(Y, 1e-3, 100, kmax=32)
        st = h.stats()
        X, its, errs = h.solve_batch(Y, 1e-3, 100)
        # a strided signal and a strided solution vector through the single-signal screened form
        ybig = np.zeros(2 * m, np.float32)
        ybig[::2] = Y[0]
        xbig = np.full(3 * n, -7.0, np.float32)
        h.solve(ybig[::2], 1e-3, 100, out=xbig[::3])
    assert st["screen_signals"] == B
    recs = sharding.unpack_records(rec, 32, np.float32)
    for b in range(B):
        r = recs[b]
        nz = np.nonzero(X[b])[0]
        assert r["K"] == len(nz) and r["iter"] == its[b] and r["err"] == errs[b]
        assert np.array_equal(r["idx"][:r["K"]], nz) and np.array_equal(r["val"][:r["K"]], X[b][nz])
    assert np.array_equal(xbig[::3], X[0]) or np.abs(xbig[::3] - X[0]).max() <= 1e-5 * np.abs(X[0]).max()
    assert np.all(xbig[1::3] == -7.0) and np.all(xbig[2::3] == -7.0)


@pytest.mark.parametrize("screen", [0, 2])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_api_surface_with_and_without_the_screened_form(sship, dtype, screen):
    """What a drop-in user touches — strided host y / x, device-resident y / x, the breakpoint trace, compact records of a small
    batch — gives the oracle's answer whether the context solves in the screened form (option screen_single = 2: forced on this
    shape; fp64: its resident tier) or in the engine behind it (0); with 2 every signal must really have been certified."""
    import torch
    import sharding
    m, n, k = 2048, 16384, 16          # (well-posed enough for the fp64 resident tier's 256 columns to hold every signal's support)
    A, y, x0, sup = make_gaussian_problem(9300 + np.dtype(dtype).itemsize, m, n, k, dtype)
    tol = 1e-3 if dtype == np.float32 else 1e-9
    xo, ito, eo, tro = oracle.homotopy(A, y, tol, 4 * k, trace=True)
    B = 5
    rng = np.random.default_rng(9350)
    Y = np.empty((B, m), dtype)
    for b in range(B):
        sb = np.sort(rng.choice(n, k, replace=False))
        xb = np.zeros(n)
        xb[sb] = 1.0 + np.abs(rng.standard_normal(k))
        Y[b] = (A.astype(np.float64) @ xb).astype(dtype)
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", screen)
        h.set_option("trace", 1)
        h.reset_stats()
        ybuf = np.zeros(3 * m, dtype)
        ybuf[::3] = y
        xbuf = np.full(2 * n, -7.0, dtype)
        _, it1, e1 = h.solve(ybuf[::3], tol, 4 * k, out=xbuf[::2])
        tr = h.trace()
        yd = torch.from_numpy(y).to("cuda:0")
        xd = torch.full((n,), -3.0, dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda:0")
        _, it2, e2 = h.solve(yd, tol, 4 * k, out=xd)
        st = h.stats()
        h.set_option("trace", 0)
        rec = h.solve_batch_compact(Y, tol, 4 * k, kmax=48)
        stb = h.stats()
        X, its, errs = h.solve_batch(Y, tol, 4 * k)
    assert_parity(xbuf[::2], it1, e1, xo, ito, eo, dtype)
    assert np.all(xbuf[1::2] == -7.0)
    assert np.array_equal(tr["idx"][:-1], tro["idx"][:-1]) and np.array_equal(tr["added"][:-1], tro["added"][:-1])
    assert np.allclose(tr["gamma"][:-1], tro["gamma"][:-1], rtol=1e-3 if dtype == np.float32 else 1e-8, atol=1e-6 if dtype == np.float32 else 0)
    assert it2 == it1 and e2 == e1 and np.array_equal(xd.cpu().numpy(), xbuf[::2])
    note("test_api_surface_with_and_without_the_screened_form", dtype=np.dtype(dtype).name, screen=screen, certified=st["screen_signals"],
         resident=st["screen_resident"], batch_certified=stb["screen_signals"] - st["screen_signals"])
    if screen == 2:
        assert st["screen_signals"] == 2 and st["screen_redone"] == 0 and st["screen_resident"] == 2
        assert stb["screen_signals"] - st["screen_signals"] == B and stb["screen_redone"] == 0, {k_: v for k_, v in stb.items() if k_.startswith(("why_", "screen_"))}
    else:
        assert stb["screen_signals"] == 0
    recs = sharding.unpack_records(rec, 48, dtype)
    for b in range(B):
        r = recs[b]
        nz = np.nonzero(X[b])[0]
        assert r["K"] == len(nz) and r["iter"] == its[b] and r["err"] == errs[b]
        assert np.array_equal(r["idx"][:r["K"]], nz) and np.array_equal(r["val"][:r["K"]], X[b][nz])
        xb_o, itb_o, eb_o = oracle.homotopy(A, Y[b], tol, 4 * k)
        assert_parity(X[b], its[b], errs[b], xb_o, itb_o, eb_o, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_screened_form_tie_goes_to_the_arbiter(sship, dtype):
    """An exact tie met by the subset solve: A = [I | Gaussian], y = e_0 + e_1 + e_5 / 2.  Column 0 enters (left-most arg-max), column 1
    then ATTAINS lambda and the reference's strict t > 0 skips it for good (homotopy-cpu.cpp:143-153).  The screened form does not
    decide such a signal — a tie is a tie of the subset's view —: it goes to the engine behind it, which meets the tie over all columns
    and asks the reference-order engine; what comes back is the oracle's result, word for word."""
    m, n = 1024, 16384
    rng = np.random.default_rng(9360)
    A = np.empty((m, n), dtype)
    A[:, :m] = np.eye(m, dtype=dtype)
    A[:, m:] = (rng.standard_normal((m, n - m)) / np.sqrt(m)).astype(dtype)
    y = np.zeros(m, dtype)
    y[0] = y[1] = 1.0
    y[5] = 0.5
    tol, max_iter = 1e-3, 12
    xo, ito, eo, tro = oracle.homotopy(A, y, tol, max_iter, trace=True)
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        h.set_option("trace", 1)
        h.reset_stats()
        xg, itg, eg = h.solve(y, tol, max_iter)
        trg = h.trace()
        st = h.stats()
    note("test_screened_form_tie_goes_to_the_arbiter", dtype=np.dtype(dtype).name, tie_reruns=st["tie_reruns"], why_tie=st["why_tie"],
         certified=st["screen_signals"], redone=st["screen_redone"], tier2=st["screen_tier2"])
    assert st["tie_reruns"] == 1 and st["why_tie"] >= 1 and st["screen_signals"] == 0
    assert itg == ito and eg == eo and np.array_equal(xg, xo)
    assert np.array_equal(trg["idx"], tro["idx"]) and np.array_equal(trg["added"], tro["added"]) and np.array_equal(trg["gamma"], tro["gamma"])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_exact_recheck_of_the_columns_the_certificate_leaves_open(sship, dtype):
    """Noisy signals whose noise floor sits just below the tolerance: at the late states the columns of the floor come within the
    certificate's margin of lambda, and at the state the path ends in they exceed the subset's own ||c||_inf.  The half-precision pass
    cannot vouch for them; instead of handing the signal back the library decides exactly those columns in fp32 with the reference's
    predicates (k_scr_recheck).  What must hold: the re-checked signals are the oracle's — iterations, support, coefficients, and the
    reported ||c||_inf (which is then a column's OUTSIDE the subset: merged by the re-check) — and with option screen_recheck = 0 the
    same signals come back from the default engine with the same answer."""
    m, n, k, tol = (1024, 8192, 16, 1e-3) if dtype == np.float32 else (2048, 16384, 16, 1e-3)
    went = 0
    for seed in range(6):
        rng = np.random.default_rng(9800 + seed)
        A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(dtype)
        sup = np.sort(rng.choice(n, k, replace=False))
        x0 = np.zeros(n)
        x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
        y = (A.astype(np.float64) @ x0 + 1.8e-4 * rng.standard_normal(m)).astype(dtype)
        xo, ito, eo = oracle.homotopy(A, y, tol, 6 * k)
        with sship.Homotopy(A) as h:
            h.set_option("screen_single", 2)
            h.reset_stats()
            xg, itg, eg = h.solve(y, tol, 6 * k)
            st = h.stats()
            h.set_option("screen_recheck", 0)
            h.reset_stats()
            x0_, it0, e0 = h.solve(y, tol, 6 * k)
            st0 = h.stats()
        note("test_exact_recheck_of_the_columns_the_certificate_leaves_open", dtype=np.dtype(dtype).name, seed=seed, certified=st["screen_signals"], rechecked=st["screen_recheck"],
             headroom=st["screen_headroom"], without_recheck_certified=st0["screen_signals"], iters=itg, err=eg)
        assert st["screen_signals"] + st["screen_redone"] == 1 and st0["screen_recheck"] == 0
        assert_parity(xg, itg, eg, xo, ito, eo, dtype)
        assert_parity(x0_, it0, e0, xo, ito, eo, dtype)
        if st["screen_recheck"]:
            went += 1
            assert st["screen_signals"] == 1 and st["screen_headroom"] >= 1.0 and st0["screen_signals"] == 0
    assert went >= 1, "no signal went through the exact re-check: the test does not reach what it is for"


@pytest.mark.parametrize("B", [4, 9, 37])
def test_fp64_batch_in_the_resident_tier(sship, B):
    """fp64 batches of four signals or more: the signals of a chunk (32) share ONE ranking pass over the fp16 copy of A, run their
    paths side by side in as many workgroups (csrc/resident.hip) and are certified one screening pass each; against the oracle
    signal by signal (iterations, support, coefficients within 1e-10), dense output and compact records, one signal made
    uncertifiable on purpose (its support beyond the 144 positions a path may take: it must come back from the tiers behind, still the
    oracle's)."""
    import sharding
    m, n, k = 2048, 16384, 16
    rng = np.random.default_rng(9900 + B)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    Y = np.empty((B, m))
    ks = []
    for b in range(B):
        kb = 150 if b == 2 else k                       # (slot 2: more columns than the resident kernel holds positions)
        sb = np.sort(rng.choice(n, kb, replace=False))
        xb = np.zeros(n)
        xb[sb] = 1.0 + np.abs(rng.standard_normal(kb))
        Y[b] = A @ xb
        ks.append(kb)
    budget = 4 * 150
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        h.reset_stats()
        X, its, errs = h.solve_batch(Y, 1e-9, budget)
        st = h.stats()
        rec = h.solve_batch_compact(Y, 1e-9, budget, kmax=200)
    note("test_fp64_batch_in_the_resident_tier", B=B, certified=st["screen_signals"], resident=st["screen_resident"], tier2=st["screen_tier2"],
         redone=st["screen_redone"], why={k_: v for k_, v in st.items() if k_.startswith("why_") and v})
    assert st["screen_resident"] >= B - 1 and st["screen_tier2"] >= 1 and st["why_positions"] >= 1
    recs = sharding.unpack_records(rec, 200, np.float64)
    for b in range(B):
        xo, ito, eo = oracle.homotopy(A, Y[b], 1e-9, budget)
        assert_parity(X[b], its[b], errs[b], xo, ito, eo, np.float64)
        r = recs[b]
        nz = np.nonzero(X[b])[0]
        assert r["K"] == len(nz) and r["iter"] == its[b]
        assert np.array_equal(r["idx"][:r["K"]], nz) and np.abs(r["val"][:r["K"]] - X[b][nz]).max() <= 1e-12 * np.abs(X[b]).max()


def test_bench_measures_traffic_in_the_run():
    """bench.py's roofline.traffic: two child runs of tools/pmc_probe.py under `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE). Where the
    profiler is at hand the figure must be the passes' algorithmic bytes to within 2 % (1.0004 x and 1.0071 x on an MI355X); where
    it is not (no rocprofv3, or this test itself runs under a profiler) the function says so by returning None and bench.py
    replays profiles/traffic.json, labelled."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    got = bench.measure_traffic_live(timeout_s=120.0)
    note("test_bench_measures_traffic_in_the_run", got=got)
    if got is None:
        pytest.skip("no rocprofv3 --pmc run possible here")
    alg_first = 8192 * 65536 * 1 + 8192 * 4 + 65536 * 4          # (the ranking pass reads the fp8 copy: k_scr_first8)
    alg_screen = 8192 * 65536 * 2 + 96 * 8192 * 2 + 65536 * 4
    assert 0.99 * alg_first <= got["first16"] <= 1.02 * alg_first
    assert 0.99 * alg_screen <= got["screen"] <= 1.02 * alg_screen


def test_fp8_ranking_pass_stays_inside_its_error_model(sship):
    """The ranking pass over the fp8 (e4m3) copy of A reports nothing, but state 0 is certified from ITS bound: every column it left
    out has |c0| <= T + eps_0 with eps_0 = 2^-4 x 1.02 ||a|| ||y|| + the flush term.  Here the bound is checked from outside: on
    dictionaries with entries over many binades (so that the subnormal range of the format is used too) every signal the form certifies
    with the fp8 pass must be the oracle's — and the same signals with the fp16 pass and the fp32 sweep give the same answer."""
    import torch
    rng = np.random.default_rng(9900)
    m, n, k = 2048, 16384, 24
    scale = np.exp2(rng.integers(-14, 1, size=(m, n))).astype(np.float32)          # entries over fifteen binades
    A = (rng.standard_normal((m, n)).astype(np.float32) * scale)
    A /= np.linalg.norm(A, axis=0, keepdims=True)
    outs = {}
    sup = np.sort(rng.choice(n, k, replace=False))
    x0 = np.zeros(n, np.float32)
    x0[sup] = 1.0 + np.abs(rng.standard_normal(k)).astype(np.float32)
    y = (A.astype(np.float64) @ x0).astype(np.float32)
    for first in (2, 1, 0):
        with sship.Homotopy(A) as h:
            h.set_option("screen_single", 2)
            h.set_option("screen_first16", 1 if first else 0)
            h.set_option("screen_first8", 1 if first == 2 else 0)
            h.reset_stats()
            outs[first] = h.solve(y, 1e-3, 4 * k) + (h.stats(),)
    xo, ito, eo = oracle.homotopy(A, y, 1e-3, 4 * k)
    note("test_fp8_ranking_pass", certified={f: o[3]["screen_signals"] for f, o in outs.items()}, headroom={f: o[3]["screen_headroom"] for f, o in outs.items()})
    for first, (x, it, e, st) in outs.items():
        assert_parity(x, it, e, xo, ito, eo, np.float32)
    assert outs[2][3]["screen_signals"] == 1 and outs[1][3]["screen_signals"] == 1
    # (certified by either pass, the path is the subset solve's: the same bits)
    assert np.array_equal(outs[2][0], outs[1][0]) and outs[2][1] == outs[1][1]


def test_rescue_of_a_signal_whose_planted_column_the_ranking_missed(sship):
    """Crowded supports (k / m = 4 %): a planted column with a small coefficient has |c0| below the 448th noise column and is ranked out
    of the subset — the subset's path then runs out of positions, or the certificate finds the column far above a state.  Option
    screen_rescue (default): the declined solve's log is scanned for such columns (the certificate pass over its early states) and the
    form runs once more with them in the subset.  Every answer must be the oracle's, rescued or not; with the option off the same
    signals go back to the default engine."""
    rescued = declined_without = 0
    for seed in range(6):
        m, n, k = 1024, 8192, 39
        A, y, x0, sup = make_gaussian_problem(9700 + seed, m, n, k, np.float32)
        xo, ito, eo = oracle.homotopy(A, y, 1e-3, 3 * k + 8)
        with sship.Homotopy(A) as h:
            h.set_option("screen_single", 2)
            h.reset_stats()
            x1, it1, e1 = h.solve(y, 1e-3, 3 * k + 8)
            st = h.stats()
            h.set_option("screen_rescue", 0)
            h.reset_stats()
            x2, it2, e2 = h.solve(y, 1e-3, 3 * k + 8)
            st2 = h.stats()
        note("test_rescue", seed=seed, certified=st["screen_signals"], rescued=st["screen_rescued"], tried=st["screen_rescue_tried"], redone=st["screen_redone"],
             why={k_: v for k_, v in st.items() if k_.startswith("why_") and v}, without=dict(certified=st2["screen_signals"], redone=st2["screen_redone"]))
        assert_parity(x1, it1, e1, xo, ito, eo, np.float32)
        assert_parity(x2, it2, e2, xo, ito, eo, np.float32)
        assert st["screen_signals"] + st["screen_redone"] == 1 and st2["screen_signals"] + st2["screen_redone"] == 1
        assert st2["screen_rescued"] == 0 and st2["screen_rescue_tried"] == 0
        assert st["screen_rescued"] <= st["screen_rescue_tried"] <= 1
        rescued += st["screen_rescued"]
        declined_without += st2["screen_redone"]
        if st["screen_rescued"]:
            assert st2["screen_redone"] == 1, "a rescue happened on a signal the form certifies without it"
    assert declined_without >= 2, "the test's signals are not crowded enough to be declined"
    assert rescued >= 1, "no declined signal was rescued"


def test_rescue_in_the_fp64_resident_tier(sship):
    """The same in fp64: the resident tier's subset is 256 columns; a crowded support leaves planted columns out of it, the tier's path goes
    astray, its log names them, the tier runs once more with them (ss_hip_stats::screen_rescued) instead of handing the signal to the
    sub-dictionary tier.  Every answer is the oracle's to 1e-10."""
    rescued = 0
    for seed in range(4):
        m, n, k = 1024, 12000, 40
        A, y, x0, sup = make_gaussian_problem(9800 + seed, m, n, k, np.float64)
        xo, ito, eo = oracle.homotopy(A, y, 1e-9, 3 * k + 8)
        with sship.Homotopy(A) as h:
            h.set_option("screen_single", 2)
            h.reset_stats()
            x1, it1, e1 = h.solve(y, 1e-9, 3 * k + 8)
            st = h.stats()
            h.set_option("screen_rescue", 0)
            h.reset_stats()
            x2, it2, e2 = h.solve(y, 1e-9, 3 * k + 8)
            st2 = h.stats()
        note("test_rescue_fp64", seed=seed, certified=st["screen_signals"], resident=st["screen_resident"], rescued=st["screen_rescued"], tried=st["screen_rescue_tried"],
             tier2=st["screen_tier2"], why={k_: v for k_, v in st.items() if k_.startswith("why_") and v},
             without=dict(certified=st2["screen_signals"], resident=st2["screen_resident"], tier2=st2["screen_tier2"]))
        assert_parity(x1, it1, e1, xo, ito, eo, np.float64)
        assert_parity(x2, it2, e2, xo, ito, eo, np.float64)
        assert st2["screen_rescued"] == 0 and st2["screen_rescue_tried"] == 0
        assert st["screen_rescued"] <= st["screen_rescue_tried"] <= 1
        if st["screen_rescued"]:
            assert st["screen_resident"] == 1 and st2["screen_tier2"] >= 1, "a rescue happened on a signal the tier certifies without it"
        rescued += st["screen_rescued"]
    assert rescued >= 1, "no declined signal was rescued"
