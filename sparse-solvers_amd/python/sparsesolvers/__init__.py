"""Drop-in `sparsesolvers` package for the Homotopy path, MI355X-native.

    import sparsesolvers as ss
    x, info = ss.Homotopy(A).solve(signal, tolerance=0.1)

Mirrors bindings/python/sparsesolvers/__init__.py:1 of the reference (re-exports the
binding).  The HIP runtime is pre-loaded first so that this module and PyTorch-ROCm share
one runtime in a process (see _hip_runtime.py).
"""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_parent = os.path.dirname(_here)
if _parent not in sys.path:
    sys.path.insert(0, _parent)

import _hip_runtime  # noqa: E402

_hip_runtime.preload()

from sparsesolvers.binding import *  # noqa: E402,F401,F403
from sparsesolvers.binding import Homotopy, HomotopyReport, Irls, IrlsReport, Omp, OmpReport, version  # noqa: E402,F401
