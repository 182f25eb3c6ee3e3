"""One HIP runtime per process.

PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so under
torch/lib and request them by FILE name, while libss_hip.so needs the SONAME
libamdhip64.so.7.  If libss_hip.so is loaded first it binds /opt/rocm's runtime and a
later `import torch` brings in a second HSA runtime, which then finds no GPU ("No HIP
GPUs are available").  Pre-loading torch's copy (when torch is installed) makes both
resolve to the same runtime regardless of import order.  Without torch the system ROCm
runtime is used.  Set SS_HIP_RUNTIME=system to skip the pre-load.
"""
import ctypes
import importlib.util
import os

_done = False


def preload():
    global _done
    if _done:
        return
    _done = True
    if os.environ.get("SS_HIP_RUNTIME", "").lower() == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    path = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass
