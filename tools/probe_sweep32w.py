"""Developer probe: the 32-column lookahead pass in its tilings, alone on the chip (mean of back-to-back launches)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship
rng = np.random.default_rng(0)
A = (rng.standard_normal((8192, 65536), dtype=np.float32) / np.sqrt(8192)).astype(np.float32)
cols = rng.choice(65536, 32, replace=False).astype(np.uint32)
with sship.Homotopy(A) as h:
    for v in (0, 3, 8, 9, 1, 2):
        h.set_option("sweep32_variant", v)
        h.gram_cols(cols, 2)
        _, ms1 = h.gram_cols(cols, 1)
        _, ms = h.gram_cols(cols, 10)
        print("variant %d: single %.3f ms, mean of 10 back-to-back %.3f ms" % (v, ms1, ms))
