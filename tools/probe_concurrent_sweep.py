"""Developer probe: does a kernel that merely RUNS on another stream slow the 32-column pass down?"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship
rng = np.random.default_rng(0)
A = (rng.standard_normal((8192, 65536), dtype=np.float32) / np.sqrt(8192)).astype(np.float32)
cols = rng.choice(65536, 32, replace=False).astype(np.uint32)
side = torch.cuda.Stream()
with sship.Homotopy(A) as h:
    for v in (8, 0, 2):
        h.set_option("sweep32_variant", v)
        h.gram_cols(cols, 2)
        _, alone = h.gram_cols(cols, 1)
        res = []
        for rep in range(3):
            with torch.cuda.stream(side):
                torch.cuda._sleep(int(6e6))            # one spinning workgroup, a few ms
            time.sleep(0.0005)
            _, ms = h.gram_cols(cols, 1)
            torch.cuda.synchronize()
            res.append(ms)
        print("variant %d: alone %.3f ms, beside a spinning 1-workgroup kernel on another stream %s" % (v, alone, ["%.3f" % r for r in res]))
