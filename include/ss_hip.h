/*
 * ss_hip.h — C-ABI of the MI355X (gfx950) Homotopy l1 hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types,
 * no exceptions.  Every entry point returns 0 on success or a non-zero
 * ss_hip_status and writes a NUL-terminated message into `err` (if non-NULL).
 * The C++14 host layer (include/ss/*.h) and the pybind11 module call nothing
 * else; a maintainer of the reference would bind exactly these symbols from a
 * new `op<compute_mode::HIP, T>` specialisation.
 *
 * Citations are file:line under /root/reference.
 *
 * Data pointers (A, y, x, Y, X, r, c) may be HOST or DEVICE pointers; the library
 * asks the HIP runtime which.  The sensing matrix is copied to the device (and
 * re-laid-out column-contiguous) once, at create time — the natural upload point
 * because ss::solver captures A at construction (include/ss/ss.h:98-105).  Later
 * mutation of the caller's A is not observed.
 */
#ifndef SS_HIP_H
#define SS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_HIP_ABI_VERSION 6     /* (5: the per-reason counters of round 4; 6: screen_rescued / screen_rescue_tried, the colshard _f64 entry points) */

typedef struct ss_hip_ctx ss_hip_ctx;

typedef enum ss_hip_status {
    SS_HIP_OK            = 0,
    SS_HIP_EINVAL        = 1,  /* precondition violated (homotopy-cpu.cpp:193-199 asserts) */
    SS_HIP_ENODEVICE     = 2,  /* no usable HIP device                                     */
    SS_HIP_ERUNTIME      = 3,  /* a HIP runtime call failed (message carries hipGetErrorString) */
    SS_HIP_ENOMEM        = 4,
    SS_HIP_ECAPACITY     = 5,  /* active set outgrew the workspace capacity                */
    SS_HIP_ETYPE         = 6   /* f32 entry point called on an f64 context or vice versa   */
} ss_hip_status;

/* Number of HIP devices visible to this process (0 if none / runtime unusable). */
int ss_hip_device_count(void);

/* "major.minor.patch" of this library; ABI version above. */
const char* ss_hip_version(void);

/*
 * Replaces the state construction of ss::solver<T, homotopy_policy>
 * (include/ss/ss.h:98-105, include/ss/policies.h:42): captures the m x n sensing
 * matrix, element (i,j) at A[i*stride_row + j*stride_col] (strides in ELEMENTS;
 * row-major, padded row-major and column-major views all accepted, the layout
 * rules of src/linalg/blas_wrapper.h:63-94 generalised).
 * Returns NULL on failure (message in err).
 */
ss_hip_ctx* ss_hip_homotopy_create_f32(const float* A, size_t m, size_t n,
                                       ptrdiff_t stride_row, ptrdiff_t stride_col,
                                       int device, char* err, size_t errlen);
ss_hip_ctx* ss_hip_homotopy_create_f64(const double* A, size_t m, size_t n,
                                       ptrdiff_t stride_row, ptrdiff_t stride_col,
                                       int device, char* err, size_t errlen);

void ss_hip_homotopy_destroy(ss_hip_ctx* ctx);

/*
 * Replaces solve_homotopy::op<compute_mode, T> (src/solvers/homotopy.h:27-38,
 * run_solver src/solvers/homotopy-cpu.cpp:186-275):
 *   min ||x||_1  s.t.  A x = y
 *   y   : m elements, increment incy (elements)
 *   x   : n elements, increment incx; fully overwritten
 *   tol : eps(T) <= tol < 1;   max_iter > 0
 *   iter_out / err_out : ss::homotopy_report{iter, solution_error} (policies.h:25-32)
 */
int ss_hip_homotopy_solve_f32(ss_hip_ctx* ctx, const float* y, ptrdiff_t incy,
                              float tol, uint32_t max_iter,
                              float* x, ptrdiff_t incx,
                              uint32_t* iter_out, double* err_out,
                              char* err, size_t errlen);
int ss_hip_homotopy_solve_f64(ss_hip_ctx* ctx, const double* y, ptrdiff_t incy,
                              double tol, uint32_t max_iter,
                              double* x, ptrdiff_t incx,
                              uint32_t* iter_out, double* err_out,
                              char* err, size_t errlen);

/*
 * Orthogonal matching pursuit on the same context (NOT in the reference, which ships only
 * homotopy and irls: include/ss/ss.h:60-64; BASELINE.json names an `ss::omp` solver).  Greedy:
 *   r = y;  while (iter < max_iter && ||A^T r||_inf > tol) {
 *       idx = argmax |A^T r|;  S += idx;  x_S = argmin ||y - A_S x_S||_2;  r = y - A_S x_S;  }
 * Reuses the correlation sweep (one right-hand side), the arg-max epilogue and the bordered
 * (A_S^T A_S)^-1 of the Homotopy path.  Same argument meaning as ss_hip_homotopy_solve_*;
 * the report is {iterations, ||A^T r||_inf at exit}.
 */
int ss_hip_omp_solve_f32(ss_hip_ctx* ctx, const float* y, ptrdiff_t incy,
                         float tol, uint32_t max_iter, float* x, ptrdiff_t incx,
                         uint32_t* iter_out, double* err_out, char* err, size_t errlen);
int ss_hip_omp_solve_f64(ss_hip_ctx* ctx, const double* y, ptrdiff_t incy,
                         double tol, uint32_t max_iter, double* x, ptrdiff_t incx,
                         uint32_t* iter_out, double* err_out, char* err, size_t errlen);

/*
 * Batch of B signals sharing the context's sensing matrix: signal b is
 * Y[b*y_stride + i*incy], its solution X[b*x_stride + j*incx].
 * iter_out[B], err_out[B] receive the per-signal reports.
 * fp32 batches advance in lock-step on the MFMA units — from "batch_cols_min" (default 24) signals in the column
 * form (per round one pass over the matrix forms the Gram columns of the entering columns), from "batch_gram_min"
 * (default 512) in the Gram form on G = A^T A; where neither applies, from "batch_min" (default 192) the 2B
 * correlation GEMVs of a round become two GEMMs over the shared matrix.  Smaller batches and fp64 run one signal at a
 * time (the single-signal engine, see option "engine").  The lock-step forms agree with the single-signal engine
 * to rounding (same supports and iteration counts), not bit for bit.
 */
int ss_hip_homotopy_solve_batch_f32(ss_hip_ctx* ctx, const float* Y, size_t B,
                                    ptrdiff_t y_stride, ptrdiff_t incy,
                                    float tol, uint32_t max_iter,
                                    float* X, ptrdiff_t x_stride, ptrdiff_t incx,
                                    uint32_t* iter_out, double* err_out,
                                    char* err, size_t errlen);
int ss_hip_homotopy_solve_batch_f64(ss_hip_ctx* ctx, const double* Y, size_t B,
                                    ptrdiff_t y_stride, ptrdiff_t incy,
                                    double tol, uint32_t max_iter,
                                    double* X, ptrdiff_t x_stride, ptrdiff_t incx,
                                    uint32_t* iter_out, double* err_out,
                                    char* err, size_t errlen);

/*
 * The same batch with COMPACT output: one fixed-size record per signal instead of a dense row of n
 * coefficients (4096 signals x 65536 columns are 1 GiB dense, 3 MiB compact) — what a batched caller such
 * as the reference's benchmark driver (src/solvers/homotopy_bench.cpp:39-48) keeps of a solve, and the unit
 * the multi-GPU path gathers (one RCCL all_gather of records, no dense X anywhere).  Record b starts at
 * records + b * ss_hip_record_bytes(kmax, is_f64) and holds, packed on the device from the solver's own
 * support lists (no scan of x):
 *     uint32 K          number of non-zero coefficients of x_b (may exceed kmax: then only the first kmax,
 *                       by column index, are stored)
 *     uint32 iter       homotopy_report::iter        (policies.h:25-32)
 *     double err        homotopy_report::solution_error
 *     uint32 idx[kmax]  their column indices, ascending; unused entries 0
 *     T      val[kmax]  their values; unused entries 0
 * `records` may be a host or a device pointer (B * record_bytes bytes, 8-byte aligned).
 */
size_t ss_hip_record_bytes(uint32_t kmax, int is_f64);
int ss_hip_homotopy_solve_batch_compact_f32(ss_hip_ctx* ctx, const float* Y, size_t B,
                                            ptrdiff_t y_stride, ptrdiff_t incy,
                                            float tol, uint32_t max_iter, uint32_t kmax,
                                            void* records, char* err, size_t errlen);
int ss_hip_homotopy_solve_batch_compact_f64(ss_hip_ctx* ctx, const double* Y, size_t B,
                                            ptrdiff_t y_stride, ptrdiff_t incy,
                                            double tol, uint32_t max_iter, uint32_t kmax,
                                            void* records, char* err, size_t errlen);

/*
 * The correlation sweep on its own, c = A^T r — the blas::xgemv(CblasTrans, ...)
 * of residual_vector (homotopy-cpu.cpp:97).  Runs `repeats` launches (>= 1) and
 * reports the mean kernel time in milliseconds measured with HIP events on the
 * context's stream (ms_out may be NULL).  r: m elements, c: n elements.
 */
int ss_hip_gemv_t_f32(ss_hip_ctx* ctx, const float* r, float* c, int repeats, float* ms_out,
                      char* err, size_t errlen);
int ss_hip_gemv_t_f64(ss_hip_ctx* ctx, const double* r, double* c, int repeats, float* ms_out,
                      char* err, size_t errlen);

/*
 * Batched correlations C[b][:] = A^T R[b][:] for B right-hand sides sharing the context's
 * matrix — the MFMA (fp32 matrix-core) form the batched solver uses once B signals run in
 * lock-step.  R: B rows of m elements (row stride ldR), C: B rows of n elements (ldC).
 * fp32 contexts only.  ms_out: mean GEMM time over `repeats` launches (HIP events).
 */
int ss_hip_gemm_t_f32(ss_hip_ctx* ctx, const float* R, size_t B, ptrdiff_t ldR, float* C, ptrdiff_t ldC,
                      int repeats, float* ms_out, char* err, size_t errlen);

/*
 * Gram columns G[s][:] = A^T a_{cols[s]} for up to 32 dictionary columns (host index list) in
 * ONE pass over the matrix: the "lookahead sweep" of the single-signal solver, which fetches
 * the correlations of the 32 most likely next entrants at once so that most Homotopy
 * iterations need no sweep at all.  32 right-hand sides are 16 flop per byte of A — still
 * HBM-bound on MI355X — computed on the fp32 MFMA units.  G: S rows of n elements (ldG).
 */
int ss_hip_gram_cols_f32(ss_hip_ctx* ctx, const uint32_t* cols, size_t S, float* G, ptrdiff_t ldG,
                         int repeats, float* ms_out, char* err, size_t errlen);
/* the same pass in double precision (v_mfma_f64_16x16x4_f64) */
int ss_hip_gram_cols_f64(ss_hip_ctx* ctx, const uint32_t* cols, size_t S, double* G, ptrdiff_t ldG,
                         int repeats, float* ms_out, char* err, size_t errlen);

/*
 * Gs[i][j] = a_{cols[i]} . a_{cols[j]} for a subset of exactly 256 dictionary columns (host index list; entries
 * >= n give zero rows / columns): the 256 x 256 Gram matrix the early speculative iterations of the single-signal
 * engine run on while the full Gram columns are still being swept (csrc/subgram.hip).  Every entry is, bit for
 * bit, the value ss_hip_gram_cols_f32 returns for the same pair of columns.  Gs: 256 * 256 floats, row-major.
 */
int ss_hip_subset_gram_f32(ss_hip_ctx* ctx, const uint32_t* cols, float* Gs, int repeats, float* ms_out,
                           char* err, size_t errlen);

/*
 * y = A x on the device copy — ss::reconstruct_signal (src/lib.cpp:78-104).
 * x: n elements, y: m elements.
 */
int ss_hip_reconstruct_f32(ss_hip_ctx* ctx, const float* x, float* y, char* err, size_t errlen);
int ss_hip_reconstruct_f64(ss_hip_ctx* ctx, const double* x, double* y, char* err, size_t errlen);

/*
 * ss::norm_l1 (src/linalg/norms.h:22-27, src/lib.cpp:106-112): every column of the m x n matrix is divided,
 * IN PLACE, by its l1 norm sum_i |A(i, j)| (a zero column becomes NaN, like the reference's 0 / 0).  A may be
 * a device buffer (normalised where it lives: a column reduction and a scale kernel) or a host matrix (streamed
 * through the device in row panels); element (i, j) at A[i*stride_row + j*stride_col].  No context is needed:
 * in the reference this runs before the solver is constructed (src/solvers/test_util.h:167).
 */
int ss_hip_norm_l1_f32(float* A, size_t m, size_t n, ptrdiff_t stride_row, ptrdiff_t stride_col,
                       int device, char* err, size_t errlen);
int ss_hip_norm_l1_f64(double* A, size_t m, size_t n, ptrdiff_t stride_row, ptrdiff_t stride_col,
                       int device, char* err, size_t errlen);

/* ---- measurement ---------------------------------------------------------- */

typedef struct ss_hip_stats {
    uint64_t solves;               /* solve calls since the last reset                         */
    uint64_t iterations;           /* homotopy iterations over those solves                    */
    uint64_t sweep_launches;       /* fused 2-RHS sweep launches [c,q] = A^T [r,p] (timed ones) */
    double   sweep_ms;             /* sum of their HIP-event durations (profiling on)          */
    uint64_t sweep_bytes;          /* algorithmic bytes of ONE fused sweep: m*n*s + 2*m*s + 2*n*s */
    uint64_t sweep1_launches;      /* 1-RHS sweep launches (initial c = A^T y)                 */
    double   sweep1_ms;
    uint64_t sweep1_bytes;         /* m*n*s + m*s + n*s                                         */
    double   solve_ms;             /* HIP-event time of whole solves (upload of y .. x ready)  */
    uint64_t batch_rounds;         /* lock-step rounds run by the batched (MFMA) path          */
    uint64_t lookahead_sweeps;     /* 32-RHS lookahead sweeps run by the fp32 single-signal engine */
    uint64_t sweep32_launches;     /* ... of which timed with HIP events (profiling on)         */
    double   sweep32_ms;           /* sum of their durations                                    */
    uint64_t sweep32_bytes;        /* algorithmic bytes of ONE lookahead sweep: m*n*s + 32*m*s + 32*n*s */
    uint64_t gram_fallbacks;       /* solves re-run in residual form: tolerance too tight for Gram-form correlations */
    uint64_t persist_fallbacks;    /* solves re-run without the resident kernel (its grid was not resident)         */
    uint64_t gram_full_builds;     /* times the full Gram matrix A^T A was formed for the batched Gram form        */
    uint64_t solo_solves;          /* solves that ran in the speculative form (one workgroup + verification)        */
    uint64_t solo_retries;         /* speculative launches whose verification failed (the solve went on in the resident form) */
    /* ABI version 2 */
    double   gram_build_ms;        /* HIP-event time of the MFMA GEMM(s) that formed G = A^T A (2 m n^2 flops each; n padded to 256) */
    double   gram_alloc_ms;        /* host wall time of allocating G (hipMalloc of n_pad^2 fp32)                                    */
    uint64_t cq_launches;          /* batched Gram form, profiling on: timed launches of k_la_cq (c, q of every live signal)        */
    double   cq_ms;                /* sum of their HIP-event durations                                                             */
    uint64_t cq_bytes;             /* algorithmic bytes of those launches: per live signal and round (K + 3) n s — K rows of G read,
                                      c0 read, c and q written                                                                     */
    uint64_t sweep64_launches;     /* first lookahead sweeps of fp32 solves that fetched 64 Gram columns in one pass (timed ones)   */
    double   sweep64_ms;           /* sum of their HIP-event durations (profiling on)                                              */
    uint64_t sweep64_flops;        /* algorithmic flops of ONE such pass: 2 * 64 * m * n (MFMA-bound: 16x the flops per byte of A
                                      of a GEMV)                                                                                  */
    uint64_t sweep64_bytes;        /* its algorithmic bytes: m*n*s + 64*m*s + 64*n*s                                               */
    uint64_t batch_col_rounds;     /* rounds of mid-size batches run in the column form (Gram columns of the entering columns
                                      formed per round, 64 signals per pass over A)                                                */
    uint64_t sweep32_timed_cols;   /* dictionary columns covered by ONE timed lookahead launch: n, or — early form with the pass
                                      dealt out by shader engine (option early_se) — the main launch's share of them (57344 of
                                      65536: the rest runs beside it on another stream); its algorithmic bytes are
                                      m*cols*s + 32*m*s + 32*cols*s                                                                 */
    /* ABI version 3 */
    uint64_t sweep32_bytes_timed;  /* algorithmic bytes of the timed lookahead launches, summed launch by launch (a plain pass covers all
                                      n columns, the early form's main launch its share): sweep32_bytes_timed / sweep32_ms is the rate  */
    uint64_t tie_reruns;           /* signals solved again in the reference-order engine because a step-length scan met an exact tie
                                      (option "tie_rerun"): an off-support column attained max|c|, the reference's strict t > 0
                                      (homotopy-cpu.cpp:143-153) skips it for good, and which rounding hits that is luck                 */
    uint64_t ro_resweeps;          /* reference-order engine (engine 3): iterations whose direction had to be rebuilt from the signs of
                                      the re-computed correlations, i.e. that took a second pass over A (otherwise one per iteration)   */
    uint64_t subset_signals;       /* batched Gram form, subset form (csrc/subbatch.hip): signals solved by one workgroup on the 448 columns
                                      with the largest |A^T y| and confirmed against all columns                                        */
    uint64_t subset_redone;        /* ... signals that form declined (left its common path) or whose check failed: solved again in the
                                      lock-step Gram form                                                                              */
    double   sub_solve_ms;         /* profiling on: HIP-event time of the form's selection + per-signal solves                            */
    double   sub_verify_ms;        /* ... and of its check over all columns                                                              */
    double   c0_gemm_ms;           /* ... and of the batch GEMM C0 = Y A (k_gemm_tn_f32: c0 = A^T y of every signal of a chunk)            */
    double   c0_gemm_flops;        /* its algorithmic flops: 2 * rows * ldm * n_pad per chunk (rows = signals padded to 128)               */
    /* ABI version 4 */
    uint64_t screen_signals;       /* single fp32 signals solved in the screened form (csrc/screen.hip): the subset solve on the subset's own
                                      Gram matrix, every state of the path certified against all columns by one pass over the fp16 copy
                                      of A (rigorous error bound; nothing reported comes from that pass)                                  */
    uint64_t screen_redone;        /* ... signals that form declined or could not certify: solved again in the default engine              */
    uint64_t screen_launches;      /* timed launches of the screening pass k_scr_gemm (profiling on)                                      */
    double   screen_ms;            /* sum of their HIP-event durations                                                                    */
    uint64_t screen_bytes;         /* their algorithmic bytes: ldm * n_pad * 2 (fp16 copy of A) + 96 * ldm * 2 + n_pad * 4 each             */
    double   screen_headroom;      /* largest (|c~| + eps) / bound over the columns outside the subset and the states of the LAST
                                      screened solve (< 1: certified; refreshed by ss_hip_get_stats)                                       */
    uint64_t first16_launches;     /* timed launches of the screened form's first pass over the fp16 copy, k_scr_first (option
                                      "screen_first16"; profiling on)                                                                     */
    double   first16_ms;           /* sum of their HIP-event durations                                                                    */
    uint64_t first16_bytes;        /* their algorithmic bytes: ldm * n_pad * 2 (fp16 copy of A) + ldm * 4 (y) + n_pad * 4 (c~0) each        */
    /* ABI version 5 */
    uint64_t screen_resident;      /* screened signals (fp32 and fp64) whose path ran in the resident kernel (csrc/resident.hip: one workgroup,
                                      Gram values in registers) and was certified                                                         */
    uint64_t screen_tier2;         /* fp64: signals the resident tier (256 columns) did not report and the sub-dictionary tier (2048 columns,
                                      launch-per-iteration engine) took next                                                              */
    /* why a signal of the subset / screened forms was NOT reported by the tier that tried it (one signal may count in several;
       a signal that ends up in the default engine is counted once in screen_redone / subset_redone as before)                          */
    uint64_t why_removal;          /* a column would leave the support (only regular paths are certified)                                  */
    uint64_t why_positions;        /* more support columns than the subset kernel holds (72 fp32 / 144 fp64)                               */
    uint64_t why_breakpoints;      /* more states than its log holds (80 / 160)                                                           */
    uint64_t why_guard;            /* tolerance below the Gram-form guard                                                                 */
    uint64_t why_no_candidate;     /* no positive step-length candidate on the subset                                                     */
    uint64_t why_first_state;      /* state 0 not certified: the columns left out of the subset may reach lambda_0 (crowded first state)   */
    uint64_t why_irregular;        /* lambda went up along the path (e.g. derailed by the reference's first-step sign quirk)               */
    uint64_t why_overflow;         /* a residual overflowed the half-precision range                                                      */
    uint64_t why_column;           /* the screening pass (or the subset form's check) could not certify a (column, state)                  */
    uint64_t why_tie;              /* the subset's scan met an exact tie (its view: the default engine decides)                            */
    uint64_t screen_recheck;       /* screened signals whose uncertified (column, state) pairs were re-derived exactly in fp32 and passed   */
    uint64_t res_solve_launches;   /* timed launches (profiling on) of the path kernel of a screened single signal — k_res_solve (or, option
                                      screen_resident = 0, k_sub_solve): ONE workgroup, all iterations of the solve                        */
    double   res_solve_ms;         /* sum of their HIP-event durations                                                                    */
    /* ABI version 6 */
    uint64_t screen_rescued;       /* screened signals (fp32 and fp64 resident tier) certified by the RESCUE: the first attempt declined — a planted column was ranked out
                                      of the subset — its log named the missing columns, the second attempt held them (option "screen_rescue")  */
    uint64_t screen_rescue_tried;  /* rescues attempted                                                                                  */
} ss_hip_stats;

/* ---- IRLS: the reference's second solver (src/solvers/irls-cpu.cpp:63-124) ----------------------
 *
 * Replaces construction of solver<T, irls_policy>'s state — irls_state(A) = qr_decomposition<T>(A)
 * (include/ss/policies.h:77-85, src/lib.cpp:51-57, src/linalg/qr_decomposition.h:93-139) — and
 * solve_irls::op<mode, T> (src/solvers/irls.h:27-38).  Requires m >= n (the reference asserts it).
 * create: uploads A like ss_hip_homotopy_create_*, factorises it on the device (Householder QR,
 * thin Q, R, Q^T Q).  solve: outputs are irls_report{iter, solution_error, spd_failure}
 * (policies.h:58-72); x is normalised to sum 1 like the reference's (irls-cpu.cpp:121).
 * An IRLS context is not a Homotopy context: each family of entry points rejects the other's. */
ss_hip_ctx* ss_hip_irls_create_f32(const float* A, size_t m, size_t n, ptrdiff_t stride_row, ptrdiff_t stride_col,
                                   int device, char* err, size_t errlen);
ss_hip_ctx* ss_hip_irls_create_f64(const double* A, size_t m, size_t n, ptrdiff_t stride_row, ptrdiff_t stride_col,
                                   int device, char* err, size_t errlen);
int ss_hip_irls_solve_f32(ss_hip_ctx* ctx, const float* y, ptrdiff_t incy, float tolerance, uint32_t max_iterations,
                          float* x, ptrdiff_t incx, uint32_t* iter_out, double* solution_error_out,
                          int* spd_failure_out, char* err, size_t errlen);
int ss_hip_irls_solve_f64(ss_hip_ctx* ctx, const double* y, ptrdiff_t incy, double tolerance, uint32_t max_iterations,
                          double* x, ptrdiff_t incx, uint32_t* iter_out, double* solution_error_out,
                          int* spd_failure_out, char* err, size_t errlen);
void ss_hip_irls_destroy(ss_hip_ctx* ctx);

/* ---- one signal over a COLUMN-SHARDED dictionary (csrc/colshard.hip) -------------------------------
 *
 * For dictionaries beyond one GPU's memory (or n >> 10^6): every rank — one process per GPU — owns the columns
 * [col_lo, col_lo + n_local) of the m x n_total sensing matrix; the reference's iteration (homotopy-cpu.cpp:236-272)
 * runs with the O(m n) work local to the shard (the correlation sweep of the shard's own context) and its reductions
 * over all columns as device-side collectives: one (value, index) all-reduce for lambda = ||A^T r||_inf and the
 * left-most arg-max (homotopy-cpu.cpp:32-37), one for the smallest step length and its left-most column (:122-163),
 * and one all-reduce of m + kcap floats that carries the entering column from its owner to everyone.  The active
 * set (support, (A_S^T A_S)^-1, x_S, the active columns) is replicated: every rank performs the same update.
 * Results equal the single-GPU residual form (option "engine" = 0) to rounding; sharded and unsharded runs of THIS
 * entry point agree bit for bit.
 *
 * Transport: RCCL (librccl.so is opened at run time) — rank 0 calls ss_hip_comm_unique_id, distributes the 128 bytes
 * out of band (MPI, torch.distributed, a file) and every rank passes them to create, which calls ncclCommInitRank —
 * or, with comm_id == NULL, a table of HOST collectives (in-place all-reduces on host memory, called by every rank in
 * the same order; tests and other transports).  world == 1 needs neither.
 * x_local receives the shard's n_local coefficients; iter_out / err_out as ss_hip_homotopy_solve_f32.  Options
 * "strict_sign", "tie_guard", "zero_on_removal", "trace" apply; destroy with ss_hip_homotopy_destroy. */
#define SS_HIP_COMM_ID_BYTES 128
typedef struct ss_hip_collectives {
    void* user;
    int (*allreduce_max_u64)(void* user, uint64_t* buf, size_t count);   /* 0 on success */
    int (*allreduce_min_u64)(void* user, uint64_t* buf, size_t count);
    int (*allreduce_sum_f32)(void* user, float* buf, size_t count);
} ss_hip_collectives;
int ss_hip_comm_unique_id(unsigned char* id /* SS_HIP_COMM_ID_BYTES */, char* err, size_t errlen);
ss_hip_ctx* ss_hip_homotopy_colshard_create_f32(const float* A_local, size_t m, size_t n_local,
                                                ptrdiff_t stride_row, ptrdiff_t stride_col,
                                                size_t col_lo, size_t n_total, int device,
                                                const unsigned char* comm_id, int rank, int world,
                                                const ss_hip_collectives* host_collectives,
                                                char* err, size_t errlen);
int ss_hip_homotopy_colshard_solve_f32(ss_hip_ctx* ctx, const float* y, ptrdiff_t incy, float tol, uint32_t max_iter,
                                       float* x_local, ptrdiff_t incx, uint32_t* iter_out, double* err_out,
                                       char* err, size_t errlen);
/* The same in fp64 (round 4).  A double and its column index do not share one 64-bit word, so the two (value, index)
 * reductions of an iteration travel as a [world][2] table of 64-bit words — every rank fills its own pair, zeros elsewhere —
 * gathered by ONE exact MAX all-reduce each and reduced by every rank in the same order (largest value, then smallest global
 * index): still three collectives per iteration (2 x 16 * world bytes, (m + kcap) * 8 bytes), at most 64 ranks.  The host
 * table therefore needs two entries only.  Sharded and unsharded runs agree bit for bit, like in fp32. */
typedef struct ss_hip_collectives_f64 {
    void* user;
    int (*allreduce_max_u64)(void* user, uint64_t* buf, size_t count);   /* 0 on success */
    int (*allreduce_sum_f64)(void* user, double* buf, size_t count);
} ss_hip_collectives_f64;
ss_hip_ctx* ss_hip_homotopy_colshard_create_f64(const double* A_local, size_t m, size_t n_local,
                                                ptrdiff_t stride_row, ptrdiff_t stride_col,
                                                size_t col_lo, size_t n_total, int device,
                                                const unsigned char* comm_id, int rank, int world,
                                                const ss_hip_collectives_f64* host_collectives,
                                                char* err, size_t errlen);
int ss_hip_homotopy_colshard_solve_f64(ss_hip_ctx* ctx, const double* y, ptrdiff_t incy, double tol, uint32_t max_iter,
                                       double* x_local, ptrdiff_t incx, uint32_t* iter_out, double* err_out,
                                       char* err, size_t errlen);

/* profiling != 0: bracket every sweep launch with HIP events on the context's stream. */
int ss_hip_set_profiling(ss_hip_ctx* ctx, int profiling);
int ss_hip_get_stats(ss_hip_ctx* ctx, ss_hip_stats* out);
int ss_hip_reset_stats(ss_hip_ctx* ctx);

/*
 * Tuning knobs (integers), for benchmarks only; unknown keys return SS_HIP_EINVAL.
 *   "sweep_variant"  kernel variant of the sweep (see csrc/sweep.hip)
 *   "lookahead"      iterations the host enqueues ahead of the device's done flag
 *   "strict_sign"    1 = seed the first direction with sign(c[idx]) instead of the
 *                    reference's sign(|c[idx]|) (homotopy-cpu.cpp:223-227); default 0
 *   "trace"          1 = record the homotopy path of each solve (ss_hip_get_trace)
 *   "engine"         single-signal Homotopy: 1 (default) = lookahead engine — Gram
 *                    columns A^T a_j of active columns are cached and A is swept (32 right-hand
 *                    sides per pass) only when an uncached column enters — unless the tolerance
 *                    is below 2^-14 (fp32) / 2^-42 (fp64) * ||A^T y||_inf, too tight for Gram-form
 *                    correlations: such a solve runs as 0; 2 = lookahead engine unconditionally;
 *                    0 = one fused 2-RHS sweep per iteration (residual form);
 *                    3 = reference-order engine (csrc/reforder.hip): the reference's values statement for statement
 *                    — c = A^T(y - A x) re-computed, the direction formed from the signs of THAT c —
 *                    with every reduction in ONE documented order (8 partial sums, term r to partial r & 7, combined
 *                    ((0+1)+(2+3))+((4+5)+(6+7)), products and sums separately rounded): the path is reproducible bit
 *                    for bit by any implementation that states the same order.  One pass over A per iteration (2 GiB
 *                    at 8192 x 65536, 0.84 of the HBM peak); batches run up to 4 signals per pass; the arbiter of
 *                    "tie_rerun".
 *   "batch_screen"   1 (default) = fp32 batches of at least 4 signals on a context without G = A^T A (see "batch_gram_min"), on dictionaries
 *                    the screened form applies to ("screen_single"), run in that form chunk by chunk (64 signals): c0 of the chunk
 *                    by the batch GEMM, every signal solved by one workgroup on its subset's own Gram matrix, one screening launch
 *                    over the fp16 copy of A for the whole chunk (four signals per workgroup); a signal it hands back is solved by
 *                    the default single-signal engine.  Results agree with solve() to rounding, not bit for bit; 0 = the column
 *                    form ("batch_cols_min") or one solve per signal as before
 *   "batch_subset"   1 (default) = batches that run in Gram form on the full G = A^T A use the SUBSET form
 *                    (csrc/subbatch.hip): every signal is solved by one workgroup on the 448 columns with the largest
 *                    |A^T y| and every breakpoint is then checked against all columns by the same chain of fmas; a signal
 *                    the form declines (left its common path) or whose check fails is solved again in the lock-step
 *                    form (ss_hip_stats::subset_signals / subset_redone); 0 = the lock-step form for all
 *   "screen_single"  1 (default) = single fp32 signals on dictionaries of >= 16 Mi entries and >= 8192 columns take the
 *                    SCREENED form (csrc/screen.hip): c0 = A^T y in fp32, the whole path by one workgroup on the 448 columns
 *                    with the largest |c0| (their Gram matrix formed from A), then ONE pass over an fp16 copy of A (kept by
 *                    the context: half of A's bytes again) that certifies every state of the path against all columns —
 *                    |c~| + eps <= 7/8 lambda with eps = 2^-9 ||a_i|| ||r_k|| + flush terms, a rigorous bound.  Nothing
 *                    reported comes from that pass; a signal it cannot certify, or whose path leaves the subset form's
 *                    common path, is solved again by the default engine (ss_hip_stats::screen_signals / screen_redone).
 *                    fp64 contexts (>= 32768 columns, >= 64 Mi entries): the same certificate around the fp64 engine, which
 *                    solves the path on a sub-dictionary of the 2048 columns with the largest |c0| (a context of its own).
 *                    2 = on every shape the form can run on (tests); 0 = never.  Initial value: environment variable
 *                    SS_HIP_SCREEN_SINGLE when set.  Stands in for the default speculative engine only ("la_fused" = 3,
 *                    "early_solo" = 1); with G = A^T A in HBM the subset form on G ("gram_single") is used instead only where the
 *                    half-precision first pass ("screen_first16") does not apply — with it the screened form is the faster one.
 *   "screen_first16" 1 (default) = the screened form of one fp32 signal reads the fp16 copy for its FIRST pass too (row counts padded to
 *                    a multiple of 512, at most 15872): c~0 = A16^T y ranks the columns, the exact fp32 c0 of the 448 chosen ones
 *                    is formed beside their Gram matrix (lambda_0, the first pick and the path come from those), and state 0 is
 *                    certified like every other state: T + eps_0 <= 7/8 lambda_0 with T above every |c~0| left out and
 *                    eps_0 = 2^-9 max ||a_i|| ||y|| + the flush term.  No fp32 pass over A is left in a certified solve
 *                    (ss_hip_stats::first16_*); 0 = c0 = A^T y by the fp32 sweep
 *   "screen_first8"  1 (default) = where the padded row count is a multiple of 1024 that first pass reads an FP8 (OCP e4m3) copy of A instead of the
 *                    fp16 copy (a quarter of A's bytes again, made on the first such solve): the pass only RANKS the columns and bounds what it
 *                    left out — eps_0 = 2^-4 x 1.02 max ||a_i|| ||y|| + its flush term, sixteen times the fp16 pass's, written by the pass
 *                    itself for the state-0 certificate; nothing it computes is reported.  fp32 and fp64 contexts; 0 = the fp16 copy
 *                    (no fp8 copy is made); an allocation that does not fit does the same by itself
 *   "screen_resident" 1 (default) = the path of the screened form runs in the one-workgroup resident kernel (csrc/resident.hip; fp32: 448 columns,
 *                    72 positions; fp64: 256 columns, 136 positions — the fp64 resident tier); 0 = fp32: k_sub_solve, fp64: the sub-dictionary
 *                    tier only.  OMP takes the screened form only with 1
 *   "screen_rescue"  1 (default) = a single fp32 signal the screened form declined because its path ran out of positions or an outside column beat
 *                    a state is scanned for the columns the ranking missed (the certificate pass over the declined solve's early states) and
 *                    solved once more in the same form with those columns in the subset (ss_hip_stats::screen_rescued); 0 = it goes back
 *   "screen_recheck" 1 (default) = columns the half-precision certificate cannot clear are decided exactly from A in the solve's precision
 *                    (k_scr_recheck) and a last step an outside column stops early is repaired (k_scr_repair); 0 = such signals go back
 *   "gram_reserve"   1 (default) = the memory of G = A^T A is reserved on a helper thread when the first batch of >= 4 signals arrives, so that
 *                    the batch that forms G does not wait for the allocation; 0 = allocated on first use
 *   "temporal_cols"  leading dictionary columns the sweeps read with cache-allocating loads (the rest: non-temporal); 0 (default) = none
 *   "sweep_f64_variant" tiling of the 32-column fp64 pass: 0 (default) = 256 columns / 512 threads, one workgroup per CU; 1, 2 = 128 / 256, two or three
 *   "solo_full_gram" tests: 1 = the speculative form may run with G = A^T A as its cache too (default 0: it does not — gathers of scattered entries)
 *   "early_probe"    developer aid: 1 = the early form without overlap (the passes first, then the speculative launch)
 *   "pass_dbg_ptr"   developer aid: a device buffer (1 + 4 x 4096 u64) that receives a per-workgroup trace of the early form's passes (tools/probe_pass_trace.py)
 *   "colshard_fail_prepare" test aid: 1 = the next column-sharded solve fails on this rank while it prepares (the ranks must all leave)
 *   "ro_slots"       1..8 (default 8; fp64 contexts use at most 4): signals the reference-order engine runs in lock-step per pass over A (batches in
 *                    engine 3, a batch's tie re-runs); every signal's words are those of its own solve
 *   "ro_staged"      1 (default) = its sweep stages the dictionary through LDS (coalesced loads); 0 = direct 16-byte
 *                    loads (one signal per pass; the same bits at 0.35 of the HBM peak): A/B aid
 *   "ro_force_resweep" developer aid: 1 = every iteration of engine 3 takes its second sweep (the check of the
 *                    speculated signs is treated as failed): a schedule, not arithmetic — the same bits
 *   "la_fused"       form of the lookahead engine's iterations: 3 (default, fp32) = speculative resident form:
 *                    one workgroup iterates on a 256-column subset and every breakpoint is re-derived over
 *                    all columns, bit for bit, before anything is committed; 2 = one resident launch on all
 *                    CUs (k_la_persist, fp32; same results as 3; fp64 runs as 1), 1 = one launch per
 *                    iteration, 0 = separate kernels
 *   "solo_subset"    columns beyond the support a speculative launch may hold (default 256; tests use small
 *                    values to provoke failed verifications)
 *   "sweep32_variant" tiling of the 32-RHS lookahead sweep (0 default; 1-7 measured alternatives, same results)
 *   "first_sweep_cols" 32 (default) / 64: 64 = the first lookahead pass of a fp32 solve (plain speculative form)
 *                    fetches the entering column and the 63 largest |A^T y| in ONE pass over A (2*64*m*n flops:
 *                    MFMA-bound, 0.61 ms at 8192 x 65536) instead of 32; measured no faster end to end
 *   "early_solo"     1 (default) = early form of the speculative engine (fp32, n > 16384): the first speculative
 *                    launch iterates on the 256 x 256 Gram matrix of its column subset (csrc/subgram.hip, the
 *                    pass's own arithmetic) while two 32-column passes over A run beside it on a second stream;
 *                    the passes' columns are what the verification then needs.  Same results, bit for bit, as 0
 *                    (= the passes first, then the launch)
 *   "early_pass"     tiling of the early form's two passes: 2 (default) = 128-column LDS-staged tiles, three 256-thread
 *                    workgroups per CU (all 512 tiles of 8192 x 65536 resident on the 255 CUs the speculative launch
 *                    leaves: 0.49 ms per pass beside it); 0 = one 32-column tile per single-wave workgroup (0.52 ms).
 *                    Same results bit for bit
 *   "early_se"       (3 = the same dealing-out for ANY tile count: 7u tiles per shader engine in the main launch, the
 *                    workgroups dealt out by SE come back until their SE has had u more — measured no faster than one launch per
 *                    pass beyond 65536 columns, kept as an option)
 *                    1 (default) = the early form's two passes are dealt out around the speculative workgroup: the
 *                    hardware gives every shader engine the same number of workgroups of a grid, and the SE of that
 *                    workgroup has 7 CUs for its share — so the main launch takes 14 tiles per SE, a second launch puts
 *                    two more workgroups on every SE, which pick their tile by where they run and leave at once on that
 *                    SE, and the last two tiles are formed by a VALU kernel: every other CU carries exactly two tiles
 *                    (0.37-0.38 ms per pass instead of 0.47-0.49; 256-CU parts, 57345..65536 columns); 0 = one launch
 *                    per pass.  Same results bit for bit
 *   "early_adapt"    1 (default) = the early form's second pass takes its 32 columns from the speculative launch's
 *                    progress when the first pass has finished (columns that have entered its support, then those
 *                    closest to entering); 0 = from the |c0| ranking.  Decides which Gram columns are fetched when,
 *                    never a result
 *   "sweep_cols_f64" 64 (default) / 32: right-hand sides of one fp64 lookahead pass (k_gemm32_tn_f64<RH>): fp64
 *                    passes are MFMA-bound, so 64 columns cost 1.6 x the time of 32 and a solve needs fewer passes
 *                    and fewer round trips through the host; same results as 32
 *   "sweep_cols_f64_late" 32 (default) / 64: the same for the third and later passes of a solve — misses that late on
 *                    the path are sparse (16384 x 131072, k = 128: the third pass serves the last ~8 iterations), a
 *                    narrow pass (3.4 ms against 5.4) covers them; same results
 *   "cache_mib"      memory budget of the lookahead engine's Gram-column cache (default 2048)
 *   "batch_min"      smallest fp32 batch that takes the lock-step MFMA path (default 192: below
 *                    that, one lookahead solve per signal is faster)
 *   "batch_chunk"    signals processed together by the batched path (default 4096)
 *   "batch_gram_min" (where the screened batch form applies — "batch_screen" — G pays later and is formed for a batch of at
 *                    least max(batch_gram_min, 1536) signals, or once the context has received 3072 signals in batches)
 *                    smallest lock-step batch that forms G = A^T A (n^2 fp32, 2 m n^2 flops once) and then
 *                    takes every signal's correlations from rows of G instead of two GEMMs per round
 *                    (default 512; once G exists every lock-step batch uses it; 0 = never)
 *   "batch_cols_min" / "batch_cols_max"  fp32 batches of at least batch_cols_min signals (default 24; max 0 = no upper
 *                    limit) with no G at hand — below batch_gram_min, or G switched off or too large — run in
 *                    lock-step in the column form: per round ONE pass over A per 64 signals forms the Gram columns
 *                    of the columns that enter (a single solve spends three passes on one signal), and correlations
 *                    come from those cached columns as in the Gram form; chunks of at most 448 signals; smaller
 *                    batches run one solve per signal; batch_cols_min = 0: never (round-1 behaviour: one solve per
 *                    signal below batch_min, two GEMMs per round from there on)
 *   "scan_blocks"    workgroups per signal of the step-length scan in the lock-step Gram forms with 64 signals or more
 *                    (default 8: a launch of 4096 signals x 64 workgroups spends its time on reductions and tickets,
 *                    not on its 9 bytes per column; same results); 0 = one per 1024 columns
 *   "batch_fused_scan" 1 (default) = the lock-step Gram forms scan inside the Gram-form pass (k_la_cqs: c and q of a
 *                    workgroup's columns stay in registers while the signal's workgroups meet for lambda = ||c||_inf);
 *                    0 = two kernels (k_la_cq writes c and q, k_scansel reads them back).  Same results bit for bit
 *   "cq_cols" / "cq_rows"  columns per thread (4, 8, 16 (default), 32) and rows of G in flight per thread (1, 2 (default),
 *                    3, 4, 8) of that pass: a workgroup of 256 threads reads runs of 256 * cq_cols columns of each of its
 *                    signal's K rows of G — 16 KiB runs stream at 82 % of the HBM peak where 4 KiB runs reached 64 %.
 *                    "cq_vec4" 1 = four consecutive columns per thread with 16-byte loads (cq_cols = 4 only; no faster).
 *                    Same results bit for bit
 *   "gram_full_gib"  largest G — and largest column cache of the column form — that may be allocated
 *                    (default 64 GiB; 0 = never form G)
 *   "gram_full_after" opt-in (default 0 = never): single-signal solves (fp32) after which the context forms G
 *                    for them as well: with G in HBM every Gram column is at hand and a solve needs no pass over
 *                    A beyond A^T y — at the price of n^2 fp32 of HBM (17 GiB at 8192 x 65536) and one GEMM;
 *                    1 = from the first solve
 *   "gram_single"    1 (default) = once G exists (a large batch formed it, or gram_full_after) single-signal
 *                    solves use it as their Gram-column cache; 0 = they keep their own 32-column sweeps
 *   "gram_symmetric" 1 (default) = G is formed from the GEMM tiles on and above the diagonal, each stored to both
 *                    sides (half the flops, G exactly symmetric); 0 = the full product
 *   "profile_every"  with profiling on, bracket only every k-th fused sweep with events
 *   "profile_solve_every" with profiling on, only every k-th solve carries events at all (each event costs stream time)
 *   "tie_guard"      0 (default) = the reference's strict `t > 0` (homotopy-cpu.cpp:135,145,151): an
 *                    off-support column that attains max|c| exactly (it tied with an inserted column
 *                    within an ulp) is skipped for good and such a solve runs to max_iterations, as the
 *                    reference's does; 1 = opt-in fix: that column enters by a zero-length step
 *   "tie_rerun"      1 (default) = a solve (or a signal of a batch) whose step-length scan met an EXACT tie — an
 *                    off-support column that attains max|c|, candidate t == 0 — is solved again in the
 *                    reference-order engine (engine 3) and that result is returned: after such a tie the reference's
 *                    strict t > 0 decides by rounding whether the path derails, so the fast engines (other summation
 *                    orders) do not guess; ss_hip_stats::tie_reruns counts them.  0 = keep the fast engine's path
 *                    (with "tie_guard" = 1 the tie is resolved by the guard and nothing is re-run)
 *   "zero_on_removal" 0 (default) = a coefficient whose column leaves the support keeps the reference's
 *                    x + gamma*d rounding residue (homotopy-cpu.cpp:246-252; 0 or an ulp, and a
 *                    re-inserted column may bounce out again); 1 = opt-in fix: it is set to exactly 0.
 *                    Both fixes are restated in the CPU oracle (SS_ORACLE_TIE_GUARD,
 *                    SS_ORACLE_ZERO_ON_REMOVAL) so that the opt-in modes are checked too.
 */
int ss_hip_set_option(ss_hip_ctx* ctx, const char* key, long value);
int ss_hip_get_option(ss_hip_ctx* ctx, const char* key, long* value);

/*
 * The homotopy path of the LAST solve when option "trace" is on: entry 0 is the initial
 * pick, entry t the column toggled by iteration t (added = 1 insert / 0 remove), the step
 * length gamma taken and lambda = ||c||_inf at the start of that iteration.  Writes up to
 * `capacity` entries into each non-NULL array; *count receives the number available.
 */
int ss_hip_get_trace(ss_hip_ctx* ctx, uint32_t capacity, uint32_t* idx, uint8_t* added,
                     double* gamma, double* c_inf, uint32_t* count);

/* Shape / placement queries. */
int ss_hip_ctx_info(const ss_hip_ctx* ctx, size_t* m, size_t* n, int* is_f64, int* device);

#ifdef __cplusplus
}
#endif
#endif /* SS_HIP_H */
