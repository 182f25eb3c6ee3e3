"""Where the subset form's time goes at configs[2] (one 4096-signal batch, profiling on): SS_HIP_SUB_DBG switches parts of
k_sub_verify off (1: no loads of G, 2: no chains, 4: no predicates) — timing only, the results are then meaningless."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, "sparse-solvers_amd/python"); sys.path.insert(0, ".")
import torch, sship
import bench as B
dev = torch.device("cuda:0")
A_host = B.survey_matrix()
A = torch.from_numpy(A_host).to(dev)
Bx = int(os.environ.get("PROBE_B", "4096"))
Yb, supb, coefb = B.make_batch(A, 4242, Bx, B.K_SPARSE, torch)
out = {}
with sship.Homotopy(A) as h:
    rb = h.record_bytes(96)
    rec = torch.zeros((Bx, rb), dtype=torch.uint8, device=dev)
    h.solve_batch_compact(Yb, B.TOL, B.MAX_ITER, kmax=96, out=rec); torch.cuda.synchronize()
    for dbg in os.environ.get("PROBE_DBG", "0,1,2,4,7").split(","):
        os.environ["SS_HIP_SUB_DBG"] = dbg
        h.reset_stats(); h.set_profiling(True)
        t = time.perf_counter(); h.solve_batch_compact(Yb, B.TOL, B.MAX_ITER, kmax=96, out=rec); torch.cuda.synchronize()
        dt = time.perf_counter() - t
        st = h.stats()
        out[dbg] = dict(seconds=dt, solve_ms=st["sub_solve_ms"], verify_ms=st["sub_verify_ms"], accepted=int(st["subset_signals"]), redone=int(st["subset_redone"]))
        print(dbg, out[dbg], flush=True)
json.dump(out, open("gpurun_out/probe_subverify.json", "w"))
