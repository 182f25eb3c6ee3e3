"""Developer probe: configs[2] batch (4096 signals, Gram form) with different numbers of scan workgroups per slot."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, ROOT)
import sship
from bench import survey_matrix, make_batch, K_SPARSE, TOL, MAX_ITER, KMAX_RECORD
A = torch.from_numpy(survey_matrix()).to("cuda:0")
with sship.Homotopy(A) as h:
    Yb, supb, coefb = make_batch(A, 4242, 4096, K_SPARSE, torch)
    rec = torch.zeros((4096, h.record_bytes(KMAX_RECORD)), dtype=torch.uint8, device="cuda:0")
    ref = None
    for sb in (0, 0, 32, 16, 8, 4, 16, 0):
        h.set_option("scan_blocks", sb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.solve_batch_compact(Yb, TOL, MAX_ITER, kmax=KMAX_RECORD, out=rec)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        r = rec.cpu().numpy().copy()
        same = ref is None or bool(np.array_equal(r, ref))
        ref = r if ref is None else ref
        print("scan_blocks %2d: %.3f s = %.0f signals/s, records equal to the first run: %s" % (sb, dt, 4096 / dt, same), flush=True)
