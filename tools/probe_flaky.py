#!/usr/bin/env python3
"""GPU stress probe: the reference-style tiny cases, many times, per engine form."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sship
import ref_cases

def factory(opts, log):
    def solve(A, y, tol, max_iter):
        with sship.Homotopy(A) as h:
            for k, v in opts.items():
                h.set_option(k, v)
            h.set_option("trace", 1)
            r = h.solve(np.asarray(y, dtype=A.dtype), tol, max_iter)
            log[:] = [r[1], r[2], h.trace(), h.stats()]
            return r
    return solve

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
modes = {"sweep": {"engine": 0}, "la0": {"engine": 1, "la_fused": 0}, "la1": {"engine": 1, "la_fused": 1}, "la2": {"engine": 1, "la_fused": 2}}
for name, opts in modes.items():
    fails = 0
    t0 = time.time()
    for r in range(reps):
        log = []
        for fn, args in ((ref_cases.permutations, (10, 25, np.float32, .05, .05, 50)),
                         (ref_cases.noisy_signal, (np.float32,)),
                         (ref_cases.noisy_signal, (np.float64,)),
                         (ref_cases.permutations, (25, 10, np.float32, .1, .1, 50))):
            try:
                if fn is ref_cases.permutations:
                    fn(factory(opts, log), args[0], args[1], args[2], args[3], args[4], args[5])
                else:
                    fn(factory(opts, log), args[0])
            except AssertionError as e:
                fails += 1
                print(name, "rep", r, fn.__name__, args, "FAILED:", str(e)[:100].replace("\n", " "), "iter/err", log[:2], flush=True)
                if log:
                    print("    trace idx", log[2]["idx"][:10], "added", log[2]["added"][:10], "gamma", log[2]["gamma"][:6], flush=True)
    print(name, "fails", fails, "of", reps * 4, "%.1f s" % (time.time() - t0), flush=True)
