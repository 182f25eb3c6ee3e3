"""ctypes front-end of the CPU oracle — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module (see ``oracle/ss_oracle.h``).  It restates
``/root/reference/src/solvers/homotopy-cpu.cpp:186-275`` on the CPU and is the
checker the HIP path is compared against; it is never the thing shipped or
measured as the product.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libss_oracle.so")

SPARSE_NOTRANS = 1
STRICT_SIGN = 2
ZERO_ON_REMOVAL = 4   # opt-in, not the reference (restates the HIP option zero_on_removal = 1)
TIE_GUARD = 8         # opt-in, not the reference (restates the HIP option tie_guard = 1)
CBLAS = 16            # timing leg only: the GEMVs through a dlopen'd CBLAS (load_cblas)


class _Report(ctypes.Structure):
    _fields_ = [("iter", ctypes.c_uint32), ("solution_error", ctypes.c_double)]


class _Trace(ctypes.Structure):
    _fields_ = [
        ("capacity", ctypes.c_uint32),
        ("count", ctypes.c_uint32),
        ("idx", ctypes.POINTER(ctypes.c_uint32)),
        ("added", ctypes.POINTER(ctypes.c_uint8)),
        ("gamma", ctypes.POINTER(ctypes.c_double)),
        ("c_inf", ctypes.POINTER(ctypes.c_double)),
    ]


def build(force=False):
    """Compile libss_oracle.so with the committed Makefile (gcc, a few seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("ss_oracle.c", "ss_oracle_impl.inc", "ss_oracle_irls.inc", "ss_oracle.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "libss_oracle.so"], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return _LIB_PATH


_lib = None


def usable_cpus():
    """CPUs this process may actually use: affinity mask and cgroup quota, capped at 16
    (the GPU box advertises 128 logical CPUs but grants a share of 16 per GPU; running
    128 OpenMP threads there oversubscribes ~8x).  SS_ORACLE_THREADS overrides."""
    if os.environ.get("SS_ORACLE_THREADS"):
        return max(1, int(os.environ["SS_ORACLE_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 16))


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, sz, pd = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_ssize_t
        for suf, ct in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
            f = getattr(L, "ss_oracle_homotopy_" + suf)
            f.restype = ctypes.c_int
            f.argtypes = [vp, sz, sz, pd, pd, vp, pd, ct, ctypes.c_uint32, vp, pd,
                          ctypes.POINTER(_Report), ctypes.POINTER(_Trace), ctypes.c_uint]
            f = getattr(L, "ss_oracle_omp_" + suf)
            f.restype = ctypes.c_int
            f.argtypes = [vp, sz, sz, pd, pd, vp, pd, ct, ctypes.c_uint32, vp, pd,
                          ctypes.POINTER(_Report), ctypes.POINTER(_Trace)]
            for name in ("ss_oracle_gemv_t_", "ss_oracle_gemv_n_"):
                g = getattr(L, name + suf)
                g.restype = None
                g.argtypes = [vp, sz, sz, pd, pd, vp, vp]
            g = getattr(L, "ss_oracle_square_permute_" + suf)
            g.restype = None
            g.argtypes = [vp, sz, sz, sz]
            for name, args, res in (
                    ("ss_oracle_qr_", [vp, sz, sz, vp, vp], None),
                    ("ss_oracle_qr_q_", [vp, sz, sz, vp], None),
                    ("ss_oracle_qr_r_", [vp, vp, sz, vp], None),
                    ("ss_oracle_qr_solve_", [vp, vp, sz, sz, vp, vp], None),
                    ("ss_oracle_cholesky_", [vp, sz, vp], ctypes.c_int),
                    ("ss_oracle_cholesky_solve_", [vp, sz, vp, vp], None),
                    ("ss_oracle_irls_", [vp, sz, sz, vp, ct, ctypes.c_uint32, vp,
                                         ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_int)], ctypes.c_int)):
                g = getattr(L, name + suf)
                g.argtypes = args
                g.restype = res
        L.ss_oracle_erase_last_rowcol_f32.restype = None
        L.ss_oracle_erase_last_rowcol_f32.argtypes = [vp, sz, sz]
        L.ss_oracle_insert_last_rowcol_f32.restype = None
        L.ss_oracle_insert_last_rowcol_f32.argtypes = [vp, sz, sz, ctypes.c_float]
        L.ss_oracle_inverse_create.restype = vp
        L.ss_oracle_inverse_create.argtypes = [sz, ctypes.c_int]
        L.ss_oracle_inverse_destroy.argtypes = [vp]
        L.ss_oracle_inverse_insert.argtypes = [vp, sz, vp]
        L.ss_oracle_inverse_remove.argtypes = [vp, sz]
        L.ss_oracle_inverse_size.restype = sz
        L.ss_oracle_inverse_size.argtypes = [vp]
        L.ss_oracle_inverse_get.argtypes = [vp, vp]
        L.ss_oracle_rank_index_create.restype = vp
        L.ss_oracle_rank_index_destroy.argtypes = [vp]
        L.ss_oracle_rank_index_insert.argtypes = [vp, ctypes.c_uint32]
        L.ss_oracle_rank_index_erase.argtypes = [vp, ctypes.c_uint32]
        L.ss_oracle_rank_index_rank_of.argtypes = [vp, ctypes.c_uint32]
        L.ss_oracle_rank_index_rank_at.restype = ctypes.c_uint32
        L.ss_oracle_rank_index_rank_at.argtypes = [vp, sz]
        L.ss_oracle_rank_index_size.restype = sz
        L.ss_oracle_rank_index_size.argtypes = [vp]
        L.ss_oracle_num_threads.restype = ctypes.c_int
        L.ss_oracle_set_num_threads.argtypes = [ctypes.c_int]
        L.ss_oracle_set_num_threads(usable_cpus())
        _lib = L
    return _lib


def load_cblas(threads=None):
    """dlopen a CBLAS for the CBLAS flag, like the reference's loader (blas_wrapper.cpp:33-66): scipy's bundled
    OpenBLAS (LP64, symbols prefixed scipy_), else a system libopenblas / libblas.  Returns its description
    or None if none is usable."""
    import glob
    L = lib()
    L.ss_oracle_load_cblas.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    L.ss_oracle_load_cblas.restype = ctypes.c_int
    L.ss_oracle_cblas_config.restype = ctypes.c_char_p
    threads = usable_cpus() if threads is None else int(threads)
    cands = []
    try:
        import scipy
        d = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
        cands += [(p, b"scipy_") for p in sorted(glob.glob(os.path.join(d, "libscipy_openblas-*.so")))]
    except Exception:
        pass
    cands += [(p, b"") for p in ("libopenblas.so.0", "libopenblas.so", "libcblas.so.3", "libblas.so.3")]
    for path, prefix in cands:
        if L.ss_oracle_load_cblas(path.encode(), prefix, threads) == 0:
            cfg = L.ss_oracle_cblas_config()
            return "%s (%s)" % (os.path.basename(path), cfg.decode().strip() if cfg else "cblas")
    return None


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError("oracle supports float32/float64, got %s" % dtype)


def num_threads():
    return lib().ss_oracle_num_threads()


def homotopy(A, y, tolerance, max_iterations, flags=SPARSE_NOTRANS, trace=False):
    """run_solver<T> (homotopy-cpu.cpp:186-275) on the CPU.

    A may have any 2-D strides (row-major, padded row-major, column-major views).
    Returns (x, iter, solution_error[, trace dict]).
    """
    A = np.asarray(A)
    suf = _suffix(A.dtype)
    y = np.asarray(y)
    if y.dtype != A.dtype:
        raise TypeError("dtype of y must match A")
    if A.ndim != 2 or y.ndim != 1:
        raise ValueError("A must be 2-D and y 1-D")
    m, n = A.shape
    if y.shape[0] != m:
        raise ValueError("len(y) != rows of A")
    item = A.dtype.itemsize
    x = np.zeros(n, dtype=A.dtype)
    rep = _Report()
    tr = _Trace()
    keep = None
    if trace:
        cap = int(max_iterations) + 1
        keep = (np.zeros(cap, np.uint32), np.zeros(cap, np.uint8),
                np.zeros(cap, np.float64), np.zeros(cap, np.float64))
        tr.capacity = cap
        tr.idx = keep[0].ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        tr.added = keep[1].ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        tr.gamma = keep[2].ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        tr.c_inf = keep[3].ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    fn = getattr(lib(), "ss_oracle_homotopy_" + suf)
    rc = fn(A.ctypes.data, m, n, A.strides[0] // item, A.strides[1] // item,
            y.ctypes.data, y.strides[0] // item, tolerance, int(max_iterations),
            x.ctypes.data, 1, ctypes.byref(rep), ctypes.byref(tr) if trace else None,
            int(flags))
    if rc != 0:
        raise RuntimeError("oracle homotopy failed with code %d" % rc)
    if trace:
        k = tr.count
        return x, rep.iter, rep.solution_error, {
            "idx": keep[0][:k].copy(), "added": keep[1][:k].copy(),
            "gamma": keep[2][:k].copy(), "c_inf": keep[3][:k].copy()}
    return x, rep.iter, rep.solution_error


def omp(A, y, tolerance, max_iterations):
    """Orthogonal matching pursuit (no reference implementation exists: unpinned).
    Returns (x, iter, ||A^T r||_inf, selected columns in pick order)."""
    A = np.asarray(A)
    suf = _suffix(A.dtype)
    y = np.asarray(y)
    if y.dtype != A.dtype:
        raise TypeError("dtype of y must match A")
    m, n = A.shape
    item = A.dtype.itemsize
    x = np.zeros(n, dtype=A.dtype)
    rep = _Report()
    cap = int(max_iterations) + 1
    idx = np.zeros(cap, np.uint32)
    tr = _Trace()
    tr.capacity = cap
    tr.idx = idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    rc = getattr(lib(), "ss_oracle_omp_" + suf)(
        A.ctypes.data, m, n, A.strides[0] // item, A.strides[1] // item, y.ctypes.data,
        y.strides[0] // item, tolerance, int(max_iterations), x.ctypes.data, 1,
        ctypes.byref(rep), ctypes.byref(tr))
    if rc != 0:
        raise RuntimeError("oracle omp failed with code %d" % rc)
    return x, rep.iter, rep.solution_error, idx[:tr.count].copy()


def gemv_t(A, v):
    """c = A^T v with the oracle's summation order."""
    A = np.asarray(A)
    suf = _suffix(A.dtype)
    v = np.ascontiguousarray(v, dtype=A.dtype)
    m, n = A.shape
    item = A.dtype.itemsize
    c = np.zeros(n, dtype=A.dtype)
    getattr(lib(), "ss_oracle_gemv_t_" + suf)(
        A.ctypes.data, m, n, A.strides[0] // item, A.strides[1] // item, v.ctypes.data, c.ctypes.data)
    return c


def gemv_n(A, x):
    """y = A x with the oracle's summation order (reconstruct_signal, lib.cpp:78-104)."""
    A = np.asarray(A)
    suf = _suffix(A.dtype)
    x = np.ascontiguousarray(x, dtype=A.dtype)
    m, n = A.shape
    item = A.dtype.itemsize
    y = np.zeros(m, dtype=A.dtype)
    getattr(lib(), "ss_oracle_gemv_n_" + suf)(
        A.ctypes.data, m, n, A.strides[0] // item, A.strides[1] // item, x.ctypes.data, y.ctypes.data)
    return y


def square_permute(A, src, dest):
    A = np.ascontiguousarray(A)
    suf = _suffix(A.dtype)
    out = A.copy()
    getattr(lib(), "ss_oracle_square_permute_" + suf)(out.ctypes.data, A.shape[0], src, dest)
    return out


def erase_last_rowcol(A):
    A = np.ascontiguousarray(A, dtype=np.float32)
    M, N = A.shape
    buf = A.copy().ravel()
    lib().ss_oracle_erase_last_rowcol_f32(buf.ctypes.data, M, N)
    return buf[:(M - 1) * (N - 1)].reshape(M - 1, N - 1).copy()


def insert_last_rowcol(A, val=0.0):
    A = np.ascontiguousarray(A, dtype=np.float32)
    M, N = A.shape
    buf = np.empty((M + 1) * (N + 1), dtype=np.float32)
    buf[:M * N] = A.ravel()
    lib().ss_oracle_insert_last_rowcol_f32(buf.ctypes.data, M, N, val)
    return buf.reshape(M + 1, N + 1)


class OnlineColumnInverse:
    """online_column_inverse<T> (online_inverse.h:35-63)."""

    def __init__(self, m, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.m = m
        self._h = lib().ss_oracle_inverse_create(m, 1 if self.dtype == np.float64 else 0)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ss_oracle_inverse_destroy(self._h)
            self._h = None

    def insert(self, rank, col):
        col = np.ascontiguousarray(col, dtype=self.dtype)
        assert col.shape == (self.m,)
        if lib().ss_oracle_inverse_insert(self._h, rank, col.ctypes.data) != 0:
            raise RuntimeError("insert failed")

    def remove(self, rank):
        if lib().ss_oracle_inverse_remove(self._h, rank) != 0:
            raise RuntimeError("remove failed")

    def N(self):
        return lib().ss_oracle_inverse_size(self._h)

    def inverse(self):
        k = self.N()
        out = np.zeros((k, k), dtype=self.dtype)
        if k:
            lib().ss_oracle_inverse_get(self._h, out.ctypes.data)
        return out


class RankIndex:
    """rank_index<uint32_t> (rank_index.h:26-98)."""

    def __init__(self):
        self._h = lib().ss_oracle_rank_index_create()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ss_oracle_rank_index_destroy(self._h)
            self._h = None

    def insert(self, item):
        return lib().ss_oracle_rank_index_insert(self._h, item)

    def erase(self, item):
        return bool(lib().ss_oracle_rank_index_erase(self._h, item))

    def rank_of(self, item):
        return lib().ss_oracle_rank_index_rank_of(self._h, item)

    def rank_at(self, rank):
        return lib().ss_oracle_rank_index_rank_at(self._h, rank)

    def size(self):
        return lib().ss_oracle_rank_index_size(self._h)


# ---- IRLS and the factorisations under it (ss_oracle_irls.inc) ---------------------------------

def _rowmajor(A):
    A = np.ascontiguousarray(A)
    _suffix(A.dtype)
    return A


class QR:
    """qr_decomposition<T> (qr_decomposition.h:93-227): Householder QR of an M x N matrix, M >= N."""

    def __init__(self, A):
        A = _rowmajor(A)
        self.M, self.N = A.shape
        assert self.M >= self.N
        self.suf = _suffix(A.dtype)
        self.qr = np.empty_like(A)
        self.rdiag = np.empty(self.N, A.dtype)
        getattr(lib(), "ss_oracle_qr_" + self.suf)(A.ctypes.data, self.M, self.N, self.qr.ctypes.data,
                                                   self.rdiag.ctypes.data)

    def q(self):
        q = np.empty_like(self.qr)
        getattr(lib(), "ss_oracle_qr_q_" + self.suf)(self.qr.ctypes.data, self.M, self.N, q.ctypes.data)
        return q

    def r(self):
        r = np.empty((self.N, self.N), self.qr.dtype)
        getattr(lib(), "ss_oracle_qr_r_" + self.suf)(self.qr.ctypes.data, self.rdiag.ctypes.data, self.N,
                                                     r.ctypes.data)
        return r

    def solve(self, b):
        b = np.ascontiguousarray(b, dtype=self.qr.dtype)
        x = np.empty(self.N, self.qr.dtype)
        getattr(lib(), "ss_oracle_qr_solve_" + self.suf)(self.qr.ctypes.data, self.rdiag.ctypes.data, self.M,
                                                         self.N, b.ctypes.data, x.ctypes.data)
        return x


def cholesky(A):
    """cholesky_decomposition<T> (cholesky_decomposition.h:56-83) -> (L lower, isspd)"""
    A = _rowmajor(A)
    N = A.shape[0]
    L = np.empty_like(A)
    ok = getattr(lib(), "ss_oracle_cholesky_" + _suffix(A.dtype))(A.ctypes.data, N, L.ctypes.data)
    return L, bool(ok)


def cholesky_solve(L, b):
    L = _rowmajor(L)
    b = np.ascontiguousarray(b, dtype=L.dtype)
    x = np.empty_like(b)
    getattr(lib(), "ss_oracle_cholesky_solve_" + _suffix(L.dtype))(L.ctypes.data, L.shape[0], b.ctypes.data,
                                                                   x.ctypes.data)
    return x


def irls(A, y, tolerance, max_iterations):
    """run_solver<T> of irls-cpu.cpp:63-124 on the CPU -> (x, iter, solution_error (eps), spd_failure)"""
    A = _rowmajor(A)
    y = np.ascontiguousarray(y)
    if y.dtype != A.dtype:
        raise TypeError("dtype of y must match A")
    M, N = A.shape
    x = np.zeros(N, A.dtype)
    it = ctypes.c_uint32(0)
    eps = ctypes.c_double(0.0)
    spd = ctypes.c_int(0)
    rc = getattr(lib(), "ss_oracle_irls_" + _suffix(A.dtype))(
        A.ctypes.data, M, N, y.ctypes.data, tolerance, int(max_iterations), x.ctypes.data,
        ctypes.byref(it), ctypes.byref(eps), ctypes.byref(spd))
    if rc != 0:
        raise RuntimeError("oracle irls failed with code %d" % rc)
    return x, int(it.value), float(eps.value), bool(spd.value)
