/*
 * ss_oracle.c — TEST INFRASTRUCTURE ONLY (see ss_oracle.h for scope and pinning).
 *
 * CPU restatement of the reference's Homotopy l1 hot path:
 *   src/solvers/homotopy-cpu.cpp:32-275, src/linalg/online_inverse.h:76-300,
 *   src/linalg/rank_index.h:53-98  (paths relative to /root/reference).
 */
#include "ss_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif


/* ---- optional CBLAS for the four GEMV call sites (timing leg only) ----------------------------
 * The reference does its GEMVs in an OpenBLAS it dlopen()s at first use (src/linalg/blas_wrapper.cpp:
 * 33-66).  ss_oracle_load_cblas() does the same with whatever CBLAS the host has (bench.py passes
 * scipy's bundled libscipy_openblas, symbols prefixed "scipy_"); with SS_ORACLE_CBLAS the homotopy
 * driver then calls cblas_sgemv / cblas_dgemv at homotopy-cpu.cpp:96,97,116,120 instead of the
 * fixed-order loops.  Parity tests never set the flag: a BLAS's summation order is its own. */
#include <dlfcn.h>
#include <stdio.h>
typedef void (*ss_sgemv_fn)(int, int, int, int, float, const float*, int, const float*, int, float, float*, int);
typedef void (*ss_dgemv_fn)(int, int, int, int, double, const double*, int, const double*, int, double, double*, int);
static void* g_blas_handle = NULL;
static ss_sgemv_fn g_sgemv = NULL;
static ss_dgemv_fn g_dgemv = NULL;
static char g_blas_config[256] = "";

int ss_oracle_load_cblas(const char* path, const char* prefix, int threads)
{
    char name[128];
    if (g_blas_handle) { dlclose(g_blas_handle); g_blas_handle = NULL; g_sgemv = NULL; g_dgemv = NULL; }
    g_blas_handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!g_blas_handle) return -1;
    if (!prefix) prefix = "";
    snprintf(name, sizeof(name), "%scblas_sgemv", prefix);
    g_sgemv = (ss_sgemv_fn)dlsym(g_blas_handle, name);
    snprintf(name, sizeof(name), "%scblas_dgemv", prefix);
    g_dgemv = (ss_dgemv_fn)dlsym(g_blas_handle, name);
    if (!g_sgemv || !g_dgemv) { dlclose(g_blas_handle); g_blas_handle = NULL; g_sgemv = NULL; g_dgemv = NULL; return -2; }
    snprintf(name, sizeof(name), "%sopenblas_set_num_threads", prefix);
    void (*set_threads)(int) = (void (*)(int))dlsym(g_blas_handle, name);
    if (set_threads && threads > 0) set_threads(threads);
    snprintf(name, sizeof(name), "%sopenblas_get_config", prefix);
    char* (*get_config)(void) = (char* (*)(void))dlsym(g_blas_handle, name);
    snprintf(g_blas_config, sizeof(g_blas_config), "%s", get_config ? get_config() : "cblas");
    return 0;
}

const char* ss_oracle_cblas_config(void) { return g_blas_handle ? g_blas_config : NULL; }

#define T      float
#define SUF    f32
#define T_MAX  FLT_MAX
#define T_EPS  FLT_EPSILON
#define T_TINY FLT_MIN
#define T_GEMV g_sgemv
#define T_HYPOT hypotf
#include "ss_oracle_impl.inc"
#include "ss_oracle_irls.inc"
#undef T
#undef SUF
#undef T_MAX
#undef T_EPS
#undef T_TINY
#undef T_GEMV
#undef T_HYPOT

#define T      double
#define SUF    f64
#define T_MAX  DBL_MAX
#define T_EPS  DBL_EPSILON
#define T_TINY DBL_MIN
#define T_GEMV g_dgemv
#define T_HYPOT hypot
#include "ss_oracle_impl.inc"
#include "ss_oracle_irls.inc"
#undef T
#undef SUF
#undef T_MAX
#undef T_EPS
#undef T_TINY
#undef T_GEMV
#undef T_HYPOT

int ss_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void ss_oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

void ss_oracle_erase_last_rowcol_f32(float* A, size_t M, size_t N)
{
    erase_last_rowcol_f32(A, M, N);
}

void ss_oracle_insert_last_rowcol_f32(float* A, size_t M, size_t N, float val)
{
    insert_last_rowcol_f32(A, M, N, val);
}

/* ---- online_column_inverse handle ------------------------------------------ */
struct ss_oracle_inverse {
    int is_f64;
    inverse_t_f32 f;
    inverse_t_f64 d;
};

ss_oracle_inverse* ss_oracle_inverse_create(size_t m, int is_f64)
{
    ss_oracle_inverse* h = (ss_oracle_inverse*)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->is_f64 = is_f64;
    h->f.m = m;
    h->d.m = m;
    return h;
}

void ss_oracle_inverse_destroy(ss_oracle_inverse* h)
{
    if (!h) return;
    inverse_free_f32(&h->f);
    inverse_free_f64(&h->d);
    free(h);
}

int ss_oracle_inverse_insert(ss_oracle_inverse* h, size_t rank, const void* col)
{
    return h->is_f64 ? inverse_insert_f64(&h->d, rank, (const double*)col)
                     : inverse_insert_f32(&h->f, rank, (const float*)col);
}

int ss_oracle_inverse_remove(ss_oracle_inverse* h, size_t rank)
{
    return h->is_f64 ? inverse_remove_f64(&h->d, rank) : inverse_remove_f32(&h->f, rank);
}

size_t ss_oracle_inverse_size(const ss_oracle_inverse* h)
{
    return h->is_f64 ? h->d.n : h->f.n;
}

void ss_oracle_inverse_get(const ss_oracle_inverse* h, void* out)
{
    if (h->is_f64) memcpy(out, h->d.inv, h->d.n * h->d.n * sizeof(double));
    else           memcpy(out, h->f.inv, h->f.n * h->f.n * sizeof(float));
}

/* ---- rank_index<uint32_t>, rank_index.h:53-98 ------------------------------- */
struct ss_oracle_rank_index {
    uint32_t* v;
    size_t    n, cap;
};

ss_oracle_rank_index* ss_oracle_rank_index_create(void)
{
    return (ss_oracle_rank_index*)calloc(1, sizeof(ss_oracle_rank_index));
}

void ss_oracle_rank_index_destroy(ss_oracle_rank_index* r)
{
    if (!r) return;
    free(r->v);
    free(r);
}

static size_t ri_lower_bound(const ss_oracle_rank_index* r, uint32_t item)
{
    size_t lo = 0, hi = r->n;
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        if (r->v[mid] < item) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* rank_index.h:65-75: a duplicate insert is a no-op that returns the existing rank */
int ss_oracle_rank_index_insert(ss_oracle_rank_index* r, uint32_t item)
{
    const size_t b = ri_lower_bound(r, item);
    if (b == r->n || r->v[b] != item) {
        if (r->n == r->cap) {
            const size_t cap = r->cap ? r->cap * 2 : 8;
            uint32_t* v = (uint32_t*)realloc(r->v, cap * sizeof(uint32_t));
            if (!v) return -1;
            r->v = v;
            r->cap = cap;
        }
        memmove(r->v + b + 1, r->v + b, (r->n - b) * sizeof(uint32_t));
        r->v[b] = item;
        r->n++;
    }
    return (int)b;
}

/* rank_index.h:89-98: erases *lower_bound(item) WITHOUT checking equality;
 * false only when the bound is end() */
int ss_oracle_rank_index_erase(ss_oracle_rank_index* r, uint32_t item)
{
    const size_t b = ri_lower_bound(r, item);
    if (b == r->n) return 0;
    memmove(r->v + b, r->v + b + 1, (r->n - b - 1) * sizeof(uint32_t));
    r->n--;
    return 1;
}

/* rank_index.h:77-83 */
int ss_oracle_rank_index_rank_of(const ss_oracle_rank_index* r, uint32_t item)
{
    const size_t b = ri_lower_bound(r, item);
    return (b == r->n || r->v[b] != item) ? -1 : (int)b;
}

uint32_t ss_oracle_rank_index_rank_at(const ss_oracle_rank_index* r, size_t rank)
{
    return r->v[rank];
}

size_t ss_oracle_rank_index_size(const ss_oracle_rank_index* r)
{
    return r->n;
}
