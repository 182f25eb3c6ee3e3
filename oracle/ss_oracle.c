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

#define T      float
#define SUF    f32
#define T_MAX  FLT_MAX
#define T_EPS  FLT_EPSILON
#define T_HYPOT hypotf
#include "ss_oracle_impl.inc"
#include "ss_oracle_irls.inc"
#undef T
#undef SUF
#undef T_MAX
#undef T_EPS
#undef T_HYPOT

#define T      double
#define SUF    f64
#define T_MAX  DBL_MAX
#define T_EPS  DBL_EPSILON
#define T_HYPOT hypot
#include "ss_oracle_impl.inc"
#include "ss_oracle_irls.inc"
#undef T
#undef SUF
#undef T_MAX
#undef T_EPS
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
