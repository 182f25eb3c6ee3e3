"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/ss_hip.h declares, the C++14 API compiles against a reference-style user
program and reports errors as values, and the pybind11 module has the reference's surface
(bindings/python/sparsesolvers/binding.cpp:114-148).  No compute calls (no GPU here).
"""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sparse-solvers_amd")
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    return True


def test_c_abi_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "ss_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ss_hip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 19
    import sship
    assert declared == set(sship.SYMBOLS)
    lib = ctypes.CDLL(sship.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert sship.version().count(".") == 2
    assert sship.device_count() >= 0


def test_no_device_is_an_error_not_a_fallback(built):
    import sship
    if sship.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(sship.SsHipError) as e:
        sship.Homotopy(np.eye(4, dtype=np.float32))
    assert "no HIP device" in str(e.value)


def test_product_never_touches_the_oracle():
    """the product path must not import, link or call anything under oracle/"""
    for base, _, files in os.walk(PKG):
        if os.path.basename(base) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "ss_oracle" not in txt and "import oracle" not in txt, os.path.join(base, f)
    for f in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", f)
        if os.path.isfile(p):
            assert "ss_oracle" not in open(p).read()


def test_cpp_api_compiles_and_reports_errors_as_values(built, tmp_path):
    exe = str(tmp_path / "test_ss_api")
    lib = os.path.join(PKG, "lib")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "test_ss_api.cpp"), "-o", exe,
           "-L", lib, "-lsparsesolvers", "-lss_hip", "-Wl,-rpath," + lib]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    import sship
    if sship.device_count() > 0:
        pytest.skip("a GPU is visible here: the device run is in test_gpu_parity.py")
    r = subprocess.run([exe, "--no-device"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "no HIP device" in r.stdout


def test_python_surface(built):
    sys.path.insert(0, os.path.join(PKG, "python"))
    import sparsesolvers as ss
    v = ss.version()
    assert isinstance(v, list) and len(v) == 3 and all(isinstance(i, int) for i in v)
    rep = ss.HomotopyReport()
    rep.iter = 7
    rep.solution_error = 0.25
    assert rep.iter == 7 and rep.solution_error == 0.25
    with pytest.raises(RuntimeError, match="Unexpected number of dimensions. Expected 2 but got 1"):
        ss.Homotopy(np.ones(4))
    import sship
    if sship.device_count() == 0:
        # construction succeeds like the reference's (errors are values at solve time)
        h = ss.Homotopy(np.eye(3))
        with pytest.raises(RuntimeError, match="no HIP device"):
            h.solve(np.ones(3))
        with pytest.raises(RuntimeError):
            h.solve(np.ones(3, dtype=np.float32))   # dtype of b != dtype of A
        with pytest.raises(TypeError):
            h.solve([1.0, 1.0, 1.0])                # b is noconvert


def test_no_null_stream_memset_in_the_device_sources():
    """Regression guard for the round-1 'iter = 1, err = 0, gamma = FLT_MIN' flake (faults): a
    blocking-API hipMemset runs on the NULL stream, which is not ordered against the context's
    hipStreamNonBlocking stream — a late memset zeroed y after its upload.  Every memset in the C sources
    must be hipMemsetAsync on the context's stream."""
    import re
    csrc = os.path.join(ROOT, "sparse-solvers_amd", "csrc")
    bad = []
    for name in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, name)).read()
        for m in re.finditer(r"\bhipMemset(2D|3D)?\s*\(", text):
            bad.append("%s:%d" % (name, text.count("\n", 0, m.start()) + 1))
    assert not bad, "blocking hipMemset on the null stream: %s" % ", ".join(bad)


def test_every_option_key_is_documented_in_the_header():
    """ss_hip_set_option accepts a key iff include/ss_hip.h says what it does: the keys of the dispatcher in csrc/homotopy.hip
    (strcmp(key, "...")) must all appear, quoted, in the header's option list."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "sparse-solvers_amd", "csrc", "homotopy.hip")).read()
    body = src[src.index("int ss_hip_set_option"):src.index("int ss_hip_get_option")]
    keys = sorted(set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', body)))
    hdr = open(os.path.join(root, "include", "ss_hip.h")).read()
    assert len(keys) >= 40
    missing = [k for k in keys if ('"%s"' % k) not in hdr]
    assert not missing, "options without a line in include/ss_hip.h: %s" % missing


def test_python_mirror_of_the_statistics_struct_matches_the_header():
    """sship.Stats (ctypes) mirrors struct ss_hip_stats field by field, in order and in type: a field added to the header only would make
    ss_hip_get_stats write past the Python object."""
    import ctypes
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "sparse-solvers_amd", "python"))
    import sship
    hdr = open(os.path.join(root, "include", "ss_hip.h")).read()
    body = hdr[hdr.index("typedef struct ss_hip_stats"):hdr.index("} ss_hip_stats;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(uint64_t|double|uint32_t|float)\s+([a-z0-9_]+)\s*;", body)
    ctype = {"uint64_t": ctypes.c_uint64, "double": ctypes.c_double, "uint32_t": ctypes.c_uint32, "float": ctypes.c_float}
    assert len(fields) >= 40
    mirror = [(n, t) for n, t in sship.Stats._fields_]
    assert [n for _, n in fields] == [n for n, _ in mirror]
    assert [ctype[t] for t, _ in fields] == [t for _, t in mirror]
