#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the REFERENCE's own numpy
implementation of the Homotopy solver.

Run in the authoring container only (the reference does not travel to the GPU box):

    python tests/golden/make_golden.py [/root/reference]

It imports /root/reference/bindings/python/reference/{homotopy,common,
update_inverse_columns}.py as a library, runs `homotopy.solve(A, y, N_iter, tol)`
on seeded inputs and stores inputs + outputs (data only) in homotopy_golden.npz.

What is recorded per case
  A, y, tol            inputs (dtype of A is the case's dtype)
  x                    the reference's solution (the reference accumulates in
                       float64 whatever the dtype of A: homotopy.py:144)
  iters                number of homotopy iterations the reference ran before its
                       tolerance break; equals ss::homotopy_report::iter of the
                       C++ loop (homotopy-cpu.cpp:236-272) for the same inputs
  path_idx/add/gamma   the breakpoints: toggled column, insert(1)/remove(0) and
                       step length of every find_max_gamma call (homotopy.py:41-99),
                       entry 0 being the initial argmax pick

Case selection: the C++ solver seeds its first direction with sign(|c[idx]|) = +1
(homotopy-cpu.cpp:223-227) while the numpy reference uses sign(c[idx])
(homotopy.py:158-162); the two agree only when the leading correlation is
positive, so every case here has positive coefficients / a positive leading
correlation (asserted below).
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, os.path.join(REF, "bindings", "python", "reference"))

import homotopy as ref_homotopy  # noqa: E402  (the reference, used as a library)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "homotopy_golden.npz")


def run_reference(A, y, tol, n_iter=4000):
    """Returns x, iters, path (idx, add, gamma) using the reference implementation."""
    path = []
    orig = ref_homotopy.find_max_gamma

    def spy(A_, y_, x_, d_, c_inf_, lam_):
        g, idx, add = orig(A_, y_, x_, d_, c_inf_, lam_)
        path.append((int(idx), 1 if add else 0, float(g)))
        return g, idx, add

    ref_homotopy.find_max_gamma = spy
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            x = ref_homotopy.solve(A, y, n_iter, tol)
    finally:
        ref_homotopy.find_max_gamma = orig
    iters = buf.getvalue().count("iteration ")
    # the reference broke out on tolerance => it made exactly `iters` gamma searches
    assert iters < n_iter, "reference did not converge; golden would be ill-defined"
    assert len(path) == iters, (len(path), iters)
    c0 = A.T.astype(np.float64) @ y.astype(np.float64)
    first = int(np.argmax(np.abs(c0)))
    assert c0[first] > 0, "leading correlation must be positive (C++ first-step quirk)"
    full = [(first, 1, 0.0)] + path
    return (np.asarray(x, dtype=np.float64), iters,
            np.array([p[0] for p in full], np.uint32),
            np.array([p[1] for p in full], np.uint8),
            np.array([p[2] for p in full], np.float64))


def gaussian_case(seed, m, n, k, dtype, tol):
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(dtype)
    x0 = np.zeros(n)
    sup = rng.choice(n, size=k, replace=False)
    x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
    y = (A.astype(np.float64) @ x0).astype(dtype)
    return A, y, tol


def find_removal_case(dtype, tol):
    """Search seeds for a small problem whose homotopy path contains a removal."""
    for seed in range(1000, 1400):
        rng = np.random.default_rng(seed)
        m, n, k = 24, 64, 10
        A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(dtype)
        x0 = np.zeros(n)
        sup = rng.choice(n, size=k, replace=False)
        x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
        y = (A.astype(np.float64) @ x0).astype(dtype)
        try:
            x, iters, pidx, padd, pgam = run_reference(A, y, tol, n_iter=400)
        except AssertionError:
            continue
        if (padd == 0).any() and iters < 200:
            # keep only well-separated paths: no near-tie between breakpoints
            return seed, A, y
    raise RuntimeError("no removal case found")


def main():
    cases = {}

    def add(name, A, y, tol):
        x, iters, pidx, padd, pgam = run_reference(A, y, tol)
        cases[name] = dict(A=A, y=y, tol=np.float64(tol), x=x, iters=np.uint32(iters),
                           path_idx=pidx, path_add=padd, path_gamma=pgam)
        print("%-28s m=%-4d n=%-4d dtype=%s iters=%d nnz=%d removals=%d" % (
            name, A.shape[0], A.shape[1], A.dtype, iters, np.count_nonzero(x),
            int((padd == 0).sum())))

    add("gauss_f64_40x120_k4", *gaussian_case(11, 40, 120, 4, np.float64, 1e-6))
    add("gauss_f32_64x256_k6", *gaussian_case(12, 64, 256, 6, np.float32, 1e-3))
    add("gauss_f64_96x384_k8", *gaussian_case(13, 96, 384, 8, np.float64, 1e-8))
    add("gauss_f32_128x512_k10", *gaussian_case(14, 128, 512, 10, np.float32, 1e-3))

    seed, A, y = find_removal_case(np.float64, 1e-6)
    add("removal_f64_24x64_seed%d" % seed, A, y, 1e-6)
    add("removal_f32_24x64_seed%d" % seed, A.astype(np.float32), y.astype(np.float32), 1e-3)

    # README toy (README.md:18-28), seeded; fp64, tolerance 0.1
    rng = np.random.default_rng(0)
    N = 10
    A = rng.normal(loc=0.025, scale=0.025, size=(N, N)) + np.identity(N)
    sig = np.zeros(N)
    sig[2] = 1
    add("readme_toy_f64_10x10", A, sig, 0.1)

    # the reference's own 5x5 fixture (bindings/python/reference/main.py:19-33)
    A5 = np.array([[0.25, 0.25, 0.29, 0.15, 0.14],
                   [0.20, 0.15, 0.02, 0.16, 0.27],
                   [0.15, 0.16, 0.29, 0.07, 0.09],
                   [0.12, 0.25, 0.07, 0.25, 0.28],
                   [0.20, 0.17, 0.29, 0.25, 0.14]], dtype=np.float32)
    b5 = np.asarray([0.27, 0.12, 0.25, 0.02, 0.27], dtype=np.float32)
    add("main_py_5x5_f32", A5, b5, 0.05)

    flat = {}
    for name, d in cases.items():
        for k, v in d.items():
            flat["%s/%s" % (name, k)] = v
    np.savez_compressed(OUT, **flat)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
