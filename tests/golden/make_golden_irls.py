#!/usr/bin/env python3
"""Generate tests/golden/irls_golden.npz from the REFERENCE's own numpy implementation of IRLS
(bindings/python/reference/irls.py), imported as a library.  Authoring container only:

    python tests/golden/make_golden_irls.py [/root/reference]

Per case: A (M x N, M >= N), y, tol and the reference's solution x (normalised to sum 1,
irls.py:88) after exactly 1, 2 and 3 iterations (x1, x2, x3).  The numpy reference and the C++
solver (irls-cpu.cpp:63-124) share the Newton step, the threshold and the reweighting but differ
in their stopping rules (irls.py:68-84 also stops when the weights stop changing; the C++ loop
goes on until the second largest coefficient falls under the threshold or a Cholesky pivot
vanishes), so the path is pinned iteration by iteration, with the iteration budget as the stop.
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, os.path.join(REF, "bindings", "python", "reference"))

import irls as ref_irls  # noqa: E402  (the reference, used as a library)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "irls_golden.npz")


def run_reference(A, y, n_iter, tol):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        return ref_irls.solve(A, y, n_iter, tol)


def main():
    rng = np.random.default_rng(20260)
    cases = {}
    specs = [("f64_20x8", 20, 8, np.float64, 2), ("f64_40x16", 40, 16, np.float64, 3),
             ("f64_30x30", 30, 30, np.float64, 1), ("f32_24x10", 24, 10, np.float32, 2),
             ("f32_64x20", 64, 20, np.float32, 2)]
    for name, M, N, dt, k in specs:
        A = (rng.normal(0.0, 0.05, size=(M, N)) + np.eye(M, N)).astype(dt)
        x0 = np.zeros(N, dt)
        sup = rng.choice(N, k, replace=False)
        x0[sup] = (1.0 + rng.random(k)).astype(dt)
        y = (A.astype(np.float64) @ x0.astype(np.float64)).astype(dt)
        tol = 0.01
        xs = [np.asarray(run_reference(A, y, it, tol), np.float64) for it in (1, 2, 3)]
        cases[name] = dict(A=A, y=y, tol=np.float64(tol), x1=xs[0], x2=xs[1], x3=xs[2],
                           support=np.sort(sup).astype(np.int64))
        print(name, "support", np.sort(sup), "x3_ref", np.round(xs[2], 4))
    flat = {}
    for name, c in cases.items():
        for k2, v in c.items():
            flat[name + "/" + k2] = v
    np.savez_compressed(OUT, **flat)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
