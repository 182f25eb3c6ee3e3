"""ctypes binding of the C-ABI in include/ss_hip.h (libss_hip.so).

Used by bench.py and the GPU parity tests so that they exercise exactly the symbols a
maintainer of the reference would bind.  There is no CPU fallback
here: if the library or a GPU is missing the calls raise.

Arrays may be numpy arrays (host) or anything exposing ``data_ptr()`` / ``__cuda_array_interface__``
(device memory, e.g. torch tensors on ``cuda``) — the library asks the HIP runtime where
a pointer lives.
"""
import ctypes
import os

import numpy as np

import _hip_runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libss_hip.so")

# every symbol include/ss_hip.h declares
SYMBOLS = [
    "ss_hip_device_count", "ss_hip_version",
    "ss_hip_homotopy_create_f32", "ss_hip_homotopy_create_f64", "ss_hip_homotopy_destroy",
    "ss_hip_homotopy_solve_f32", "ss_hip_homotopy_solve_f64",
    "ss_hip_omp_solve_f32", "ss_hip_omp_solve_f64",
    "ss_hip_homotopy_solve_batch_f32", "ss_hip_homotopy_solve_batch_f64",
    "ss_hip_record_bytes", "ss_hip_homotopy_solve_batch_compact_f32", "ss_hip_homotopy_solve_batch_compact_f64",
    "ss_hip_gemv_t_f32", "ss_hip_gemv_t_f64", "ss_hip_gemm_t_f32", "ss_hip_gram_cols_f32", "ss_hip_gram_cols_f64",
    "ss_hip_subset_gram_f32",
    "ss_hip_reconstruct_f32", "ss_hip_reconstruct_f64", "ss_hip_norm_l1_f32", "ss_hip_norm_l1_f64",
    "ss_hip_set_profiling", "ss_hip_get_stats", "ss_hip_reset_stats",
    "ss_hip_set_option", "ss_hip_get_option", "ss_hip_get_trace", "ss_hip_ctx_info",
    "ss_hip_irls_create_f32", "ss_hip_irls_create_f64", "ss_hip_irls_solve_f32", "ss_hip_irls_solve_f64",
    "ss_hip_irls_destroy",
    "ss_hip_comm_unique_id", "ss_hip_homotopy_colshard_create_f32", "ss_hip_homotopy_colshard_solve_f32",
    "ss_hip_homotopy_colshard_create_f64", "ss_hip_homotopy_colshard_solve_f64",
]


COMM_ID_BYTES = 128
_CB_U64 = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t)
_CB_F32 = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t)
_CB_F64 = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_size_t)


class Collectives(ctypes.Structure):
    """struct ss_hip_collectives (include/ss_hip.h): host-side in-place all-reduces"""
    _fields_ = [("user", ctypes.c_void_p), ("allreduce_max_u64", _CB_U64), ("allreduce_min_u64", _CB_U64),
                ("allreduce_sum_f32", _CB_F32)]


class Collectives64(ctypes.Structure):
    """struct ss_hip_collectives_f64: fp64 contexts gather their (value, index) reductions by MAX and sum doubles"""
    _fields_ = [("user", ctypes.c_void_p), ("allreduce_max_u64", _CB_U64), ("allreduce_sum_f64", _CB_F64)]


class Stats(ctypes.Structure):
    _fields_ = [
        ("solves", ctypes.c_uint64),
        ("iterations", ctypes.c_uint64),
        ("sweep_launches", ctypes.c_uint64),
        ("sweep_ms", ctypes.c_double),
        ("sweep_bytes", ctypes.c_uint64),
        ("sweep1_launches", ctypes.c_uint64),
        ("sweep1_ms", ctypes.c_double),
        ("sweep1_bytes", ctypes.c_uint64),
        ("solve_ms", ctypes.c_double),
        ("batch_rounds", ctypes.c_uint64),
        ("lookahead_sweeps", ctypes.c_uint64),
        ("sweep32_launches", ctypes.c_uint64),
        ("sweep32_ms", ctypes.c_double),
        ("sweep32_bytes", ctypes.c_uint64),
        ("gram_fallbacks", ctypes.c_uint64),
        ("persist_fallbacks", ctypes.c_uint64),
        ("gram_full_builds", ctypes.c_uint64),
        ("solo_solves", ctypes.c_uint64),
        ("solo_retries", ctypes.c_uint64),
        ("gram_build_ms", ctypes.c_double),
        ("gram_alloc_ms", ctypes.c_double),
        ("cq_launches", ctypes.c_uint64),
        ("cq_ms", ctypes.c_double),
        ("cq_bytes", ctypes.c_uint64),
        ("sweep64_launches", ctypes.c_uint64),
        ("sweep64_ms", ctypes.c_double),
        ("sweep64_flops", ctypes.c_uint64),
        ("sweep64_bytes", ctypes.c_uint64),
        ("batch_col_rounds", ctypes.c_uint64),
        ("sweep32_timed_cols", ctypes.c_uint64),
        ("sweep32_bytes_timed", ctypes.c_uint64),
        ("tie_reruns", ctypes.c_uint64),
        ("ro_resweeps", ctypes.c_uint64),
        ("subset_signals", ctypes.c_uint64),
        ("subset_redone", ctypes.c_uint64),
        ("sub_solve_ms", ctypes.c_double),
        ("sub_verify_ms", ctypes.c_double),
        ("c0_gemm_ms", ctypes.c_double),
        ("c0_gemm_flops", ctypes.c_double),
        ("screen_signals", ctypes.c_uint64),
        ("screen_redone", ctypes.c_uint64),
        ("screen_launches", ctypes.c_uint64),
        ("screen_ms", ctypes.c_double),
        ("screen_bytes", ctypes.c_uint64),
        ("screen_headroom", ctypes.c_double),
        ("first16_launches", ctypes.c_uint64),
        ("first16_ms", ctypes.c_double),
        ("first16_bytes", ctypes.c_uint64),
        ("screen_resident", ctypes.c_uint64),
        ("screen_tier2", ctypes.c_uint64),
        ("why_removal", ctypes.c_uint64),
        ("why_positions", ctypes.c_uint64),
        ("why_breakpoints", ctypes.c_uint64),
        ("why_guard", ctypes.c_uint64),
        ("why_no_candidate", ctypes.c_uint64),
        ("why_first_state", ctypes.c_uint64),
        ("why_irregular", ctypes.c_uint64),
        ("why_overflow", ctypes.c_uint64),
        ("why_column", ctypes.c_uint64),
        ("why_tie", ctypes.c_uint64),
        ("screen_recheck", ctypes.c_uint64),
        ("res_solve_launches", ctypes.c_uint64),
        ("res_solve_ms", ctypes.c_double),
        ("screen_rescued", ctypes.c_uint64),
        ("screen_rescue_tried", ctypes.c_uint64),
    ]


_lib = None


def lib():
    """Loads libss_hip.so (raises OSError if it was not built — no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libss_hip.so not built: run `python sparse-solvers_amd/build.py` "
                      "(expected at %s)" % LIB_PATH)
    _hip_runtime.preload()
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, pd, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_ssize_t, ctypes.c_uint32
    cp = ctypes.c_char_p
    L.ss_hip_device_count.restype = ctypes.c_int
    L.ss_hip_version.restype = ctypes.c_char_p
    for suf, ct in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
        f = getattr(L, "ss_hip_homotopy_create_" + suf)
        f.restype = vp
        f.argtypes = [vp, sz, sz, pd, pd, ctypes.c_int, cp, sz]
        f = getattr(L, "ss_hip_homotopy_solve_" + suf)
        f.restype = ctypes.c_int
        f.argtypes = [vp, vp, pd, ct, u32, vp, pd, ctypes.POINTER(u32),
                      ctypes.POINTER(ctypes.c_double), cp, sz]
        f = getattr(L, "ss_hip_omp_solve_" + suf)
        f.restype = ctypes.c_int
        f.argtypes = [vp, vp, pd, ct, u32, vp, pd, ctypes.POINTER(u32),
                      ctypes.POINTER(ctypes.c_double), cp, sz]
        f = getattr(L, "ss_hip_homotopy_solve_batch_" + suf)
        f.restype = ctypes.c_int
        f.argtypes = [vp, vp, sz, pd, pd, ct, u32, vp, pd, pd, vp, vp, cp, sz]
        f = getattr(L, "ss_hip_homotopy_solve_batch_compact_" + suf)
        f.restype = ctypes.c_int
        f.argtypes = [vp, vp, sz, pd, pd, ct, u32, u32, vp, cp, sz]
        f = getattr(L, "ss_hip_gemv_t_" + suf)
        f.restype = ctypes.c_int
        f.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float), cp, sz]
        f = getattr(L, "ss_hip_reconstruct_" + suf)
        f.restype = ctypes.c_int
        f.argtypes = [vp, vp, vp, cp, sz]
        f = getattr(L, "ss_hip_norm_l1_" + suf)
        f.restype = ctypes.c_int
        f.argtypes = [vp, sz, sz, pd, pd, ctypes.c_int, cp, sz]
        f = getattr(L, "ss_hip_irls_create_" + suf)
        f.restype = vp
        f.argtypes = [vp, sz, sz, pd, pd, ctypes.c_int, cp, sz]
        f = getattr(L, "ss_hip_irls_solve_" + suf)
        f.restype = ctypes.c_int
        f.argtypes = [vp, vp, pd, ct, u32, vp, pd, ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_double),
                      ctypes.POINTER(ctypes.c_int), cp, sz]
    L.ss_hip_subset_gram_f32.restype = ctypes.c_int
    L.ss_hip_subset_gram_f32.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float), cp, sz]
    L.ss_hip_record_bytes.restype = sz
    L.ss_hip_record_bytes.argtypes = [u32, ctypes.c_int]
    L.ss_hip_gemm_t_f32.restype = ctypes.c_int
    L.ss_hip_gemm_t_f32.argtypes = [vp, vp, sz, pd, vp, pd, ctypes.c_int, ctypes.POINTER(ctypes.c_float), cp, sz]
    for nme in ("ss_hip_gram_cols_f32", "ss_hip_gram_cols_f64"):
        getattr(L, nme).restype = ctypes.c_int
        getattr(L, nme).argtypes = [vp, vp, sz, vp, pd, ctypes.c_int, ctypes.POINTER(ctypes.c_float), cp, sz]
    L.ss_hip_homotopy_destroy.restype = None
    L.ss_hip_homotopy_destroy.argtypes = [vp]
    L.ss_hip_irls_destroy.restype = None
    L.ss_hip_irls_destroy.argtypes = [vp]
    L.ss_hip_set_profiling.argtypes = [vp, ctypes.c_int]
    L.ss_hip_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    L.ss_hip_reset_stats.argtypes = [vp]
    L.ss_hip_set_option.argtypes = [vp, cp, ctypes.c_long]
    L.ss_hip_get_option.argtypes = [vp, cp, ctypes.POINTER(ctypes.c_long)]
    L.ss_hip_get_trace.argtypes = [vp, u32, vp, vp, vp, vp, ctypes.POINTER(u32)]
    L.ss_hip_ctx_info.argtypes = [vp, ctypes.POINTER(sz), ctypes.POINTER(sz),
                                  ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.ss_hip_comm_unique_id.restype = ctypes.c_int
    L.ss_hip_comm_unique_id.argtypes = [vp, cp, sz]
    L.ss_hip_homotopy_colshard_create_f32.restype = vp
    L.ss_hip_homotopy_colshard_create_f32.argtypes = [vp, sz, sz, pd, pd, sz, sz, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int,
                                                      ctypes.POINTER(Collectives), cp, sz]
    L.ss_hip_homotopy_colshard_solve_f32.restype = ctypes.c_int
    L.ss_hip_homotopy_colshard_solve_f32.argtypes = [vp, vp, pd, ctypes.c_float, u32, vp, pd, ctypes.POINTER(u32),
                                                     ctypes.POINTER(ctypes.c_double), cp, sz]
    L.ss_hip_homotopy_colshard_create_f64.restype = vp
    L.ss_hip_homotopy_colshard_create_f64.argtypes = [vp, sz, sz, pd, pd, sz, sz, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int,
                                                      ctypes.POINTER(Collectives64), cp, sz]
    L.ss_hip_homotopy_colshard_solve_f64.restype = ctypes.c_int
    L.ss_hip_homotopy_colshard_solve_f64.argtypes = [vp, vp, pd, ctypes.c_double, u32, vp, pd, ctypes.POINTER(u32),
                                                     ctypes.POINTER(ctypes.c_double), cp, sz]
    _lib = L
    return L


def device_count():
    return lib().ss_hip_device_count()


def version():
    return lib().ss_hip_version().decode()


def norm_l1(A, device=0):
    """ss::norm_l1 on the device, in place: every column of A (numpy array or torch tensor, host or
    device, any 2-D strides) divided by its l1 norm (src/linalg/norms.h:22-27)."""
    ptr, shape, strides, dt, keep = _describe(A)
    if len(shape) != 2:
        raise ValueError("A must be 2-D")
    suffix, _ = _suffix(dt)
    _sync_producers(A)
    err = ctypes.create_string_buffer(512)
    rc = getattr(lib(), "ss_hip_norm_l1_" + suffix)(ptr, int(shape[0]), int(shape[1]), strides[0], strides[1], device,
                                                     err, len(err))
    if rc != 0:
        raise SsHipError(rc, err.value.decode())
    return A


class SsHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ss_hip error %d: %s" % (code, msg))
        self.code = code


def _describe(a):
    """-> (pointer, shape, strides_in_elements, np.dtype, keepalive)"""
    if isinstance(a, np.ndarray):
        item = a.dtype.itemsize
        return a.ctypes.data, a.shape, tuple(s // item for s in a.strides), a.dtype, a
    if hasattr(a, "data_ptr"):      # torch tensor (host or device)
        import torch
        dt = {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64)}[a.dtype]
        return a.data_ptr(), tuple(a.shape), tuple(a.stride()), dt, a
    raise TypeError("expected a numpy array or a torch tensor")


def _sync_producers(*arrays):
    """The library consumes device pointers on the context's own (non-blocking) stream, which is not
    ordered against the stream that produced them: wait for the caller's current torch stream first
    ('streams')."""
    for a in arrays:
        if a is not None and hasattr(a, "data_ptr") and getattr(a, "is_cuda", False):
            import torch
            torch.cuda.current_stream(a.device).synchronize()
            return


def _suffix(dt):
    if dt == np.float32:
        return "f32", ctypes.c_float
    if dt == np.float64:
        return "f64", ctypes.c_double
    raise TypeError("only float32 / float64 are supported, got %s" % dt)


class Homotopy:
    """A device-resident copy of the sensing matrix + the solver loop (one HIP stream)."""

    def __init__(self, A, device=0):
        ptr, shape, strides, dt, keep = _describe(A)
        if len(shape) != 2:
            raise ValueError("A must be 2-D")
        _sync_producers(A)
        self.suffix, self.ctype = _suffix(dt)
        self.dtype = dt
        self.m, self.n = int(shape[0]), int(shape[1])
        err = ctypes.create_string_buffer(512)
        fn = getattr(lib(), "ss_hip_homotopy_create_" + self.suffix)
        self._h = fn(ptr, self.m, self.n, strides[0], strides[1], device, err, len(err))
        if not self._h:
            raise SsHipError(-1, err.value.decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().ss_hip_homotopy_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, err):
        if rc != 0:
            raise SsHipError(rc, err.value.decode())

    def solve_omp(self, y, tolerance=None, max_iterations=100, out=None):
        """orthogonal matching pursuit on the same device copy -> (x, iter, ||A^T r||_inf)"""
        return self.solve(y, tolerance, max_iterations, out, _entry="ss_hip_omp_solve_")

    def solve(self, y, tolerance=None, max_iterations=100, out=None, _entry="ss_hip_homotopy_solve_"):
        """-> (x, iter, solution_error); defaults mirror the reference binding
        (tolerance = eps(T)*10, max_iterations = 100: binding.cpp:94-95)."""
        yp, yshape, ystr, ydt, keep = _describe(y)
        if ydt != self.dtype:
            raise TypeError("dtype of y (%s) does not match the matrix (%s)" % (ydt, self.dtype))
        if len(yshape) != 1 or yshape[0] != self.m:
            raise ValueError("y must have length m = %d" % self.m)
        if tolerance is None:
            tolerance = float(np.finfo(self.dtype).eps) * 10
        if out is None:
            out = np.empty(self.n, dtype=self.dtype)
        xp, xshape, xstr, xdt, keepx = _describe(out)
        if xdt != self.dtype or len(xshape) != 1 or xshape[0] != self.n:
            raise ValueError("out must be a length-n vector of the matrix dtype")
        it = ctypes.c_uint32(0)
        e = ctypes.c_double(0.0)
        err = ctypes.create_string_buffer(512)
        _sync_producers(y, out)
        fn = getattr(lib(), _entry + self.suffix)
        rc = fn(self._h, yp, ystr[0], self.ctype(tolerance), int(max_iterations), xp, xstr[0],
                ctypes.byref(it), ctypes.byref(e), err, len(err))
        self._check(rc, err)
        return out, int(it.value), float(e.value)

    def solve_batch(self, Y, tolerance=None, max_iterations=100, out=None):
        """Y: (B, m) -> X (B, n), iters (B,), errors (B,); Y / out may live on the device"""
        Yp, shape, strides, dt, keep = _describe(Y)
        if dt != self.dtype or len(shape) != 2 or shape[1] != self.m:
            raise ValueError("Y must be (B, m) of the matrix dtype")
        B = int(shape[0])
        if tolerance is None:
            tolerance = float(np.finfo(self.dtype).eps) * 10
        X = np.empty((B, self.n), dtype=self.dtype) if out is None else out
        Xp, xshape, xstr, xdt, keepx = _describe(X)
        if xdt != self.dtype or tuple(xshape) != (B, self.n):
            raise ValueError("out must be (B, n) of the matrix dtype")
        iters = np.zeros(B, dtype=np.uint32)
        errs = np.zeros(B, dtype=np.float64)
        err = ctypes.create_string_buffer(512)
        _sync_producers(Y, X)
        fn = getattr(lib(), "ss_hip_homotopy_solve_batch_" + self.suffix)
        rc = fn(self._h, Yp, B, strides[0], strides[1], self.ctype(tolerance), int(max_iterations),
                Xp, xstr[0], xstr[1], iters.ctypes.data, errs.ctypes.data, err, len(err))
        self._check(rc, err)
        return X, iters, errs

    def record_bytes(self, kmax):
        return int(lib().ss_hip_record_bytes(int(kmax), 1 if self.dtype == np.float64 else 0))

    def solve_batch_compact(self, Y, tolerance=None, max_iterations=100, kmax=96, out=None):
        """Y: (B, m) -> records (B, record_bytes) uint8: {u32 K, u32 iter, f64 err, u32 idx[kmax], T val[kmax]}
        per signal (include/ss_hip.h), packed on the device.  `out`: a uint8 numpy array or torch tensor
        (host or device) of that shape; default a numpy array.  Decode with sharding.unpack_records."""
        Yp, shape, strides, dt, keep = _describe(Y)
        if dt != self.dtype or len(shape) != 2 or shape[1] != self.m:
            raise ValueError("Y must be (B, m) of the matrix dtype")
        B = int(shape[0])
        if tolerance is None:
            tolerance = float(np.finfo(self.dtype).eps) * 10
        rb = self.record_bytes(kmax)
        if out is None:
            out = np.empty((B, rb), dtype=np.uint8)
        if isinstance(out, np.ndarray):
            ok = out.dtype == np.uint8 and out.shape == (B, rb) and out.flags.c_contiguous
            rp = out.ctypes.data
        else:
            import torch
            ok = out.dtype == torch.uint8 and tuple(out.shape) == (B, rb) and out.is_contiguous()
            rp = out.data_ptr()
        if not ok:
            raise ValueError("out must be a contiguous (B, %d) uint8 array" % rb)
        err = ctypes.create_string_buffer(512)
        _sync_producers(Y, out)
        fn = getattr(lib(), "ss_hip_homotopy_solve_batch_compact_" + self.suffix)
        rc = fn(self._h, Yp, B, strides[0], strides[1], self.ctype(tolerance), int(max_iterations), int(kmax),
                rp, err, len(err))
        self._check(rc, err)
        return out

    def gemv_t(self, r, repeats=1, out=None):
        """c = A^T r on the device copy -> (c, mean kernel ms); `out` (host array or device tensor, length n) receives c"""
        rp, shape, strides, dt, keep = _describe(r)
        if dt != self.dtype or len(shape) != 1 or shape[0] != self.m or strides[0] != 1:
            raise ValueError("r must be a contiguous length-m vector of the matrix dtype")
        c = np.empty(self.n, dtype=self.dtype) if out is None else out
        cp, cshape, cstrides, cdt, keepc = _describe(c)
        if cdt != self.dtype or len(cshape) != 1 or cshape[0] != self.n or cstrides[0] != 1:
            raise ValueError("out must be a contiguous length-n vector of the matrix dtype")
        ms = ctypes.c_float(0.0)
        err = ctypes.create_string_buffer(512)
        _sync_producers(r, c)
        fn = getattr(lib(), "ss_hip_gemv_t_" + self.suffix)
        self._check(fn(self._h, rp, cp, int(repeats), ctypes.byref(ms), err, len(err)), err)
        return c, float(ms.value)

    def gemm_t(self, R, repeats=1, out=None):
        """C[b] = A^T R[b] for the rows of R (B, m) on the MFMA units -> (C (B, n), mean ms)"""
        Rp, shape, strides, dt, keep = _describe(R)
        if dt != np.float32 or self.dtype != np.float32 or len(shape) != 2 or shape[1] != self.m or strides[1] != 1:
            raise ValueError("R must be a (B, m) float32 array with contiguous rows")
        B = int(shape[0])
        if out is None:
            out = np.empty((B, self.n), dtype=np.float32)
        Cp, cshape, cstr, cdt, keepc = _describe(out)
        if cdt != np.float32 or tuple(cshape) != (B, self.n) or cstr[1] != 1:
            raise ValueError("out must be (B, n) float32 with contiguous rows")
        ms = ctypes.c_float(0.0)
        err = ctypes.create_string_buffer(512)
        self._check(lib().ss_hip_gemm_t_f32(self._h, Rp, B, strides[0], Cp, cstr[0], int(repeats),
                                            ctypes.byref(ms), err, len(err)), err)
        return out, float(ms.value)

    def gram_cols(self, cols, repeats=1):
        """G[s] = A^T a_{cols[s]} for up to 32 columns in one pass -> (G (S, n), mean ms)"""
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        G = np.empty((len(cols), self.n), dtype=self.dtype)
        ms = ctypes.c_float(0.0)
        err = ctypes.create_string_buffer(512)
        self._check(getattr(lib(), "ss_hip_gram_cols_" + self.suffix)(self._h, cols.ctypes.data, len(cols), G.ctypes.data, self.n,
                                               int(repeats), ctypes.byref(ms), err, len(err)), err)
        return G, float(ms.value)

    def subset_gram(self, cols, repeats=1):
        """Gs = A_S^T A_S for exactly 256 columns (csrc/subgram.hip) -> (Gs (256, 256) float32, mean kernel ms)"""
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        if cols.shape != (256,):
            raise ValueError("cols must hold 256 column indices")
        G = np.empty((256, 256), dtype=np.float32)
        ms = ctypes.c_float(0.0)
        err = ctypes.create_string_buffer(512)
        self._check(lib().ss_hip_subset_gram_f32(self._h, cols.ctypes.data, G.ctypes.data, int(repeats), ctypes.byref(ms), err, len(err)), err)
        return G, float(ms.value)

    def reconstruct(self, x):
        """y = A x on the device copy (ss::reconstruct_signal)."""
        xp, shape, strides, dt, keep = _describe(x)
        if dt != self.dtype or len(shape) != 1 or shape[0] != self.n or strides[0] != 1:
            raise ValueError("x must be a contiguous length-n vector of the matrix dtype")
        y = np.empty(self.m, dtype=self.dtype)
        err = ctypes.create_string_buffer(512)
        fn = getattr(lib(), "ss_hip_reconstruct_" + self.suffix)
        self._check(fn(self._h, xp, y.ctypes.data, err, len(err)), err)
        return y

    def set_profiling(self, on):
        lib().ss_hip_set_profiling(self._h, 1 if on else 0)

    def reset_stats(self):
        lib().ss_hip_reset_stats(self._h)

    def stats(self):
        s = Stats()
        lib().ss_hip_get_stats(self._h, ctypes.byref(s))
        return {f[0]: getattr(s, f[0]) for f in Stats._fields_}

    def trace(self):
        """path of the last solve (option "trace" must be on): dict of arrays"""
        cnt = ctypes.c_uint32(0)
        lib().ss_hip_get_trace(self._h, 0, None, None, None, None, ctypes.byref(cnt))
        k = int(cnt.value)
        idx = np.zeros(k, np.uint32)
        added = np.zeros(k, np.uint8)
        gamma = np.zeros(k, np.float64)
        c_inf = np.zeros(k, np.float64)
        if k:
            lib().ss_hip_get_trace(self._h, k, idx.ctypes.data, added.ctypes.data, gamma.ctypes.data,
                                   c_inf.ctypes.data, ctypes.byref(cnt))
        return {"idx": idx, "added": added, "gamma": gamma, "c_inf": c_inf}

    def set_option(self, key, value):
        rc = lib().ss_hip_set_option(self._h, key.encode(), int(value))
        if rc != 0:
            raise SsHipError(rc, "unknown option %r" % key)

    def get_option(self, key):
        v = ctypes.c_long(0)
        rc = lib().ss_hip_get_option(self._h, key.encode(), ctypes.byref(v))
        if rc != 0:
            raise SsHipError(rc, "unknown option %r" % key)
        return int(v.value)


def comm_unique_id():
    """ncclGetUniqueId through the library (rank 0 calls it and distributes the 128 bytes) -> bytes"""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    err = ctypes.create_string_buffer(512)
    rc = lib().ss_hip_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p), err, len(err))
    if rc != 0:
        raise SsHipError(rc, err.value.decode())
    return buf.raw


class ColumnSharded(Homotopy):
    """ONE signal over a column-sharded dictionary (ss_hip_homotopy_colshard_*_f32 / _f64): this rank owns the columns
    [col_lo, col_lo + A_local.shape[1]) of the m x n_total matrix.  Transport: `comm_id` (128 bytes from
    comm_unique_id(), the same on every rank: RCCL) or `allreduce` = a callable (numpy array, op) -> None that
    all-reduces the array IN PLACE over the ranks, op in {"max", "min", "sum"} (host collectives: tests, other
    transports); world == 1 needs neither."""

    def __init__(self, A_local, col_lo, n_total, rank=0, world=1, comm_id=None, allreduce=None, device=0):
        ptr, shape, strides, dt, keep = _describe(A_local)
        if len(shape) != 2 or dt not in (np.float32, np.float64):
            raise ValueError("A_local must be a 2-D float32 or float64 matrix")
        _sync_producers(A_local)
        self.suffix, self.ctype = _suffix(dt)
        self.dtype = dt
        self.m, self.n = int(shape[0]), int(shape[1])
        self.col_lo, self.n_total = int(col_lo), int(n_total)
        self._coll = None
        coll_p = None
        if comm_id is None and allreduce is not None:
            def wrap(op, ctype_np):
                def cb(user, buf, count):
                    try:
                        allreduce(np.ctypeslib.as_array(buf, shape=(int(count),)), op)
                        return 0
                    except Exception:                     # an exception must not cross the C boundary
                        import traceback
                        traceback.print_exc()
                        return 1
                return cb
            if dt == np.float64:
                self._cbs = (_CB_U64(wrap("max", np.uint64)), _CB_F64(wrap("sum", np.float64)))
                self._coll = Collectives64(None, *self._cbs)
            else:
                self._cbs = (_CB_U64(wrap("max", np.uint64)), _CB_U64(wrap("min", np.uint64)), _CB_F32(wrap("sum", np.float32)))
                self._coll = Collectives(None, *self._cbs)
            coll_p = ctypes.byref(self._coll)
        idbuf = None
        if comm_id is not None:
            if len(comm_id) != COMM_ID_BYTES:
                raise ValueError("comm_id must be %d bytes" % COMM_ID_BYTES)
            idbuf = ctypes.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        err = ctypes.create_string_buffer(512)
        # (an empty shard has no data pointer worth passing)
        self._h = getattr(lib(), "ss_hip_homotopy_colshard_create_" + self.suffix)(
            ptr if self.n else None, self.m, self.n, strides[0], strides[1], self.col_lo, self.n_total, device,
            ctypes.cast(idbuf, ctypes.c_void_p) if idbuf is not None else None, int(rank), int(world), coll_p, err, len(err))
        if not self._h:
            raise SsHipError(-1, err.value.decode())

    def solve(self, y, tolerance=None, max_iterations=100, out=None):
        """-> (x_local, iter, solution_error): the shard's coefficients"""
        yp, yshape, ystr, ydt, keep = _describe(y)
        if ydt != self.dtype or len(yshape) != 1 or yshape[0] != self.m:
            raise ValueError("y must be a %s vector of length m = %d" % (np.dtype(self.dtype).name, self.m))
        if tolerance is None:
            tolerance = float(np.finfo(self.dtype).eps) * 10
        if out is None:
            out = np.empty(self.n, dtype=self.dtype)
        xp, xshape, xstr, xdt, keepx = _describe(out)
        if xdt != self.dtype or len(xshape) != 1 or xshape[0] != self.n:
            raise ValueError("out must be a %s vector of the shard's width" % np.dtype(self.dtype).name)
        it = ctypes.c_uint32(0)
        e = ctypes.c_double(0.0)
        err = ctypes.create_string_buffer(512)
        _sync_producers(y, out)
        rc = getattr(lib(), "ss_hip_homotopy_colshard_solve_" + self.suffix)(self._h, yp, ystr[0], self.ctype(tolerance), int(max_iterations),
                                                      xp if self.n else None, xstr[0] if self.n else 1, ctypes.byref(it), ctypes.byref(e),
                                                      err, len(err))
        self._check(rc, err)
        return out, int(it.value), float(e.value)


class Irls:
    """IRLS on the device (the reference's ss::irls<T>): Householder QR of A at construction
    (rows >= columns), the reweighting loop in one launch per solve."""

    def __init__(self, A, device=0):
        ptr, shape, strides, dt, keep = _describe(A)
        if len(shape) != 2:
            raise ValueError("A must be 2-D")
        self.suffix, self.ctype = _suffix(dt)
        self.dtype = dt
        self.m, self.n = int(shape[0]), int(shape[1])
        err = ctypes.create_string_buffer(512)
        fn = getattr(lib(), "ss_hip_irls_create_" + self.suffix)
        self._h = fn(ptr, self.m, self.n, strides[0], strides[1], device, err, len(err))
        if not self._h:
            raise SsHipError(-1, err.value.decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().ss_hip_irls_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def solve(self, y, tolerance=None, max_iterations=100, out=None):
        """-> (x, iter, solution_error, spd_failure); defaults mirror the reference binding (binding.cpp:94-95)"""
        yp, yshape, ystr, ydt, keep = _describe(y)
        if ydt != self.dtype:
            raise TypeError("dtype of y (%s) does not match the matrix (%s)" % (ydt, self.dtype))
        if len(yshape) != 1 or yshape[0] != self.m:
            raise ValueError("y must have length m = %d" % self.m)
        if tolerance is None:
            tolerance = float(np.finfo(self.dtype).eps) * 10
        if out is None:
            out = np.empty(self.n, dtype=self.dtype)
        xp, xshape, xstr, xdt, keepx = _describe(out)
        if xdt != self.dtype or len(xshape) != 1 or xshape[0] != self.n:
            raise ValueError("out must be a length-n vector of the matrix dtype")
        it = ctypes.c_uint32(0)
        e = ctypes.c_double(0.0)
        spd = ctypes.c_int(0)
        err = ctypes.create_string_buffer(512)
        fn = getattr(lib(), "ss_hip_irls_solve_" + self.suffix)
        rc = fn(self._h, yp, ystr[0], self.ctype(tolerance), int(max_iterations), xp, xstr[0],
                ctypes.byref(it), ctypes.byref(e), ctypes.byref(spd), err, len(err))
        if rc != 0:
            raise SsHipError(rc, err.value.decode())
        return out, int(it.value), float(e.value), bool(spd.value)
