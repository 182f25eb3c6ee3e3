"""Why engine 3 is slower inside bench.py than alone: the same context after a few engine-1 solves."""
import sys, time
import numpy as np
sys.path.insert(0, "sparse-solvers_amd/python"); sys.path.insert(0, ".")
import torch, sship
import bench as B
dev = torch.device("cuda:0")
A = torch.from_numpy(B.survey_matrix()).to(dev)
sigs = [B.make_signal(A, 1235 + s, B.K_SPARSE, torch) for s in range(4)]
x = torch.zeros(B.N, device=dev)
def t3(h, tag):
    h.set_option("engine", 3)
    _, ms = h.gemv_t(sigs[0][0], 5)
    h.solve(sigs[0][0], 1e-3, 256, out=x); torch.cuda.synchronize()
    t = time.perf_counter(); h.solve(sigs[1][0], 1e-3, 256, out=x); torch.cuda.synchronize()
    print(tag, "sweep ms", round(ms, 4), "solve ms", round((time.perf_counter() - t) * 1e3, 2), flush=True)
with sship.Homotopy(A) as h:
    t3(h, "fresh context:")
    h.set_option("engine", 1)
    for s in range(4): h.solve(sigs[s][0], 1e-3, 256, out=x)
    torch.cuda.synchronize()
    t3(h, "after engine-1 solves:")
    h.set_option("engine", 1)
    h.set_profiling(True); h.solve(sigs[0][0], 1e-3, 256, out=x); h.set_profiling(False)
    t3(h, "after a profiled solve:")
    Y4 = torch.stack([s[0] for s in sigs]).contiguous(); X4 = torch.zeros((4, B.N), device=dev)
    h.set_option("engine", 3); h.solve_batch(Y4, 1e-3, 256, out=X4)
    t3(h, "after a 4-slot batch:")
    # what bench.py does before its engine-3 extra: OMP solves on the context, a second solver through the drop-in module
    h.set_option("engine", 1)
    XO = torch.zeros((2, B.N), device=dev)
    for s in range(2): h.solve_omp(sigs[s][0], 1e-3, B.K_SPARSE, out=XO[s])
    torch.cuda.synchronize()
    t3(h, "after OMP solves:")
    h.set_option("engine", 1)
    import sparsesolvers
    solver = sparsesolvers.Homotopy(A.cpu().numpy())
    solver.solve(sigs[0][0].cpu().numpy(), tolerance=1e-3, max_iterations=256)
    t3(h, "with a second solver alive:")
    del solver
    t3(h, "after deleting it:")
