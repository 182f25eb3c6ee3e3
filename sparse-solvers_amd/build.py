#!/usr/bin/env python3
"""Builds the native parts of the MI355X Homotopy path, in-tree:

  lib/libss_hip.so          HIP kernels + C-ABI (include/ss_hip.h), hipcc --offload-arch=gfx950
  lib/libsparsesolvers.so   C++14 host library mirroring the reference's ss:: API (include/ss/*.h)
  python/sparsesolvers/binding*.so   pybind11 module `sparsesolvers.binding`

hipcc cross-compiles for gfx950 without a GPU.  Objects are cached under build/ and only
rebuilt when a source or header is newer.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, "include")
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(HERE, "src")
LIB = os.path.join(HERE, "lib")
BUILD = os.path.join(HERE, "build")
PYPKG = os.path.join(HERE, "python", "sparsesolvers")

ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXX = os.environ.get("CXX") or shutil.which("g++") or "g++"

HIP_UNITS = [
    # (source, extra flags)
    ("sweep.hip", []),
    # scalar bookkeeping mirrors the reference's separately-rounded products and sums
    ("activeset.hip", ["-ffp-contract=off"]),
    ("persist.hip", ["-ffp-contract=off"]),
    ("solo.hip", ["-ffp-contract=off"]),
    ("irls.hip", ["-ffp-contract=off"]),
    ("gemm.hip", []),
    ("homotopy.hip", []),
    ("utils.hip", ["-ffp-contract=off"]),
    ("subgram.hip", []),
    # reference-order engine: separately rounded products and sums in a fixed order
    ("reforder.hip", ["-ffp-contract=off"]),
    # column-sharded single-signal solve: replicated active-set arithmetic (separately rounded like activeset.hip)
    ("colshard.hip", ["-ffp-contract=off"]),
    ("subbatch.hip", ["-ffp-contract=off", "-fno-slp-vectorize"]),
    # screened form of one signal: fp16 copy of A, the subset's Gram matrix, the screening pass (bounds, not reported values)
    ("screen.hip", []),
    # the resident subset solve (one workgroup, Gram values in registers): scalar bookkeeping rounds like the reference's
    ("resident.hip", ["-ffp-contract=off"]),
]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _headers():
    hs = []
    for d in (INCLUDE, os.path.join(INCLUDE, "ss"), CSRC, SRC):
        if os.path.isdir(d):
            hs += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hpp", ".inc"))]
    return hs


def build_hip(verbose=False):
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(BUILD, exist_ok=True)
    hdrs = _headers()
    objs = []

    def compile_one(unit):
        src, extra = unit
        srcp = os.path.join(CSRC, src)
        obj = os.path.join(BUILD, src.replace(".hip", ".o"))
        if _newer(obj, [srcp] + hdrs):
            cmd = [HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-comment",
                   "-I", INCLUDE, "-I", CSRC] + extra + ["-c", srcp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            _run(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(HIP_UNITS)) as ex:
        objs = list(ex.map(compile_one, HIP_UNITS))
    so = os.path.join(LIB, "libss_hip.so")
    if _newer(so, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", so] + objs)
    return so


def build_host(verbose=False):
    """C++14 host library (ss::solver & co.) on top of the C-ABI: src/lib.cpp + src/solvers/*-hip.cpp
    (the op<compute_mode::HIP, T> specialisations behind the compute-mode seam)."""
    os.makedirs(LIB, exist_ok=True)
    src = os.path.join(SRC, "lib.cpp")
    if not os.path.exists(src):
        return None
    srcs = [src]
    sdir = os.path.join(SRC, "solvers")
    if os.path.isdir(sdir):
        srcs += sorted(os.path.join(sdir, f) for f in os.listdir(sdir) if f.endswith(".cpp"))
    so = os.path.join(LIB, "libsparsesolvers.so")
    extra_hdrs = [os.path.join(sdir, f) for f in os.listdir(sdir) if f.endswith(".h")] if os.path.isdir(sdir) else []
    extra_hdrs += [os.path.join(INCLUDE, "kernelpp", f) for f in os.listdir(os.path.join(INCLUDE, "kernelpp"))]
    if _newer(so, srcs + _headers() + extra_hdrs):
        cmd = [CXX, "-std=c++14", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-I", INCLUDE, "-I", SRC] + srcs + [
               "-o", so, "-L", LIB, "-lss_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        _run(cmd)
    return so


def build_binding(verbose=False):
    """pybind11 module sparsesolvers.binding (same names as the reference's binding.cpp)."""
    src = os.path.join(PYPKG, "binding.cpp")
    if not os.path.exists(src):
        return None
    import pybind11
    ext = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    so = os.path.join(PYPKG, "binding" + ext)
    if _newer(so, [src] + _headers() + [os.path.join(LIB, "libsparsesolvers.so")]):
        cmd = [CXX, "-std=c++14", "-O2", "-fPIC", "-shared", "-fvisibility=hidden",
               "-I", INCLUDE, "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"],
               src, "-o", so, "-L", LIB, "-lsparsesolvers", "-lss_hip",
               "-Wl,-rpath,$ORIGIN/../../lib"]
        if verbose:
            print(" ".join(cmd))
        _run(cmd)
    return so


def build_all(verbose=False):
    out = [build_hip(verbose), build_host(verbose), build_binding(verbose)]
    return [o for o in out if o]


if __name__ == "__main__":
    for p in build_all(verbose="-v" in sys.argv):
        print("built", os.path.relpath(p, ROOT))
