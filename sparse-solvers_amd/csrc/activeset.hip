// activeset.hip — everything of a Homotopy iteration that is not the sweep:
// the step-length scan, the support toggle, the online (A_S^T A_S)^-1 update, the new
// direction and the two m-vectors r = y - A x, p = A d that feed the next sweep.
//
// Reference (paths under /root/reference):
//   find_max_gamma scan        src/solvers/homotopy-cpu.cpp:122-163   -> k_scansel
//   inverse_add_or_remove      src/solvers/homotopy-cpu.cpp:166-183   -> k_scansel (rank_index part)
//   online_column_inverse      src/linalg/online_inverse.h:183-293    -> k_gramupd
//   direction update           src/solvers/homotopy-cpu.cpp:257-267   -> k_gramupd
//   x += gamma * d             src/solvers/homotopy-cpu.cpp:252       -> k_scansel
//   residual_vector (A x part) src/solvers/homotopy-cpu.cpp:94-96     -> k_rp
//   p = A d                    src/solvers/homotopy-cpu.cpp:114-116   -> k_rp
//   loop control / inf_norm    src/solvers/homotopy-cpu.cpp:32-37,235-274 -> k_scansel
//
// All of it is O(n + K*m + K^2) per iteration against the sweep's O(m*n); these kernels
// are latency-, not bandwidth-bound, and are kept simple.  The (A_S^T A_S)^-1 matrix is
// kept as an explicit inverse updated by bordering/deflation exactly like the reference
// (NOT a Cholesky factor): every entry of the new inverse is an independent expression
// of the old one, which is what a 1024-thread workgroup wants, whereas a Cholesky
// append/downdate is a chain of K dependent steps.  The inverse is rebuilt out of place
// (ping-pong buffers) directly in sorted-support order, so the reference's rotate /
// memmove churn (square_permute, insert_last_rowcol) disappears.
//
// This file is compiled with -ffp-contract=off: products and sums round separately,
// like the reference's scalar code.
#include "ss_hip_internal.h"
#include "ss_hip_device.h"

#include <algorithm>
#include <cfloat>

namespace sship {

// ---- k_init: first pick, homotopy-cpu.cpp:217-229 ------------------------------------
template <typename T>
__global__ __launch_bounds__(kUpdThreads)
void k_init(const T* __restrict__ At, SlotDims L, const T* __restrict__ c,
            const T* __restrict__ pmax_val, const uint32_t* __restrict__ pmax_idx, uint32_t nb,
            T* __restrict__ d, uint8_t* __restrict__ insup, uint32_t* __restrict__ gam,
            uint32_t* __restrict__ touched, T* __restrict__ inv0, T tol, int strict_sign,
            DevState* st, TraceEntry* trace)
{
    {   // slot = blockIdx.y
        const size_t s = blockIdx.y;
        c += s * L.n_pad; d += s * L.n_pad; insup += s * L.n_pad;
        pmax_val += s * L.pmax_stride; pmax_idx += s * L.pmax_stride;
        gam += s * 2 * L.kcap; touched += s * 2 * L.kcap;
        inv0 += s * 2 * (size_t)L.kcap * L.kcap;
        st += s;
        if (s != 0) trace = nullptr;
    }
    const uint32_t ldm = L.ldm;
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    T c_inf;
    uint32_t idx;
    reduce_sweep_partials(pmax_val, pmax_idx, nb, c_inf, idx, sv, si);
    const T* col = At + (size_t)idx * ldm;
    const T dot = block_dot(col, col, ldm, sv);
    if (threadIdx.x == 0) {
        // online_inverse.h:193-201: inv = [1 / ||col||^2] through xnrm2
        const T nrm = sqrt(dot);
        const T inv00 = T(1) / (nrm * nrm);
        // first-step quirk: the sign is taken of c_inf = |c[idx]| >= 0
        const T seed = strict_sign ? c[idx] : c_inf;
        d[idx] = sign_tol(seed, tol) * inv00;
        insup[idx] = 1;
        gam[0] = idx;
        touched[0] = idx;
        inv0[0] = inv00;
        st->done = 0;
        st->status = 0;
        st->iter = 0;
        st->K = 1;
        st->ntouched = 1;
        st->idx = idx;
        st->rank = 0;
        st->added = 1;
        st->cur = 0;
        st->done_round = 0;
        st->c_inf = (double)c_inf;
        st->gamma = 0.0;
        st->lambda0 = (float)c_inf;
        st->dot = (double)dot;
        if (trace != nullptr) {
            trace[0].idx = idx;
            trace[0].added = 1;
            trace[0].gamma = 0.0;
            trace[0].c_inf = (double)c_inf;
        }
    }
}

// ---- k_rp: r = y - A x ; p = A d over the touched columns -----------------------------
// 256 threads = 4 waves; a workgroup owns kRpRows rows, wave w takes the columns
// touched[w], touched[w+4], ... (independent 8/16-byte loads, unrolled), the four partial
// sums are combined in wave order through LDS.
constexpr int kRpThreads = 256;
constexpr int kRpRowsPerLane = 2;
constexpr int kRpRows = 64 * kRpRowsPerLane;

template <typename T>
__global__ __launch_bounds__(kRpThreads)
void k_rp(const T* __restrict__ At, SlotDims L, const T* __restrict__ y,
          const T* __restrict__ x, const T* __restrict__ d,
          const uint32_t* __restrict__ touched2 /* [2][kcap] */,
          T* __restrict__ rhs, const DevState* st)
{
    const uint32_t ldm = L.ldm, kcap = L.kcap;
    T* rhs_p;
    {   // slot = blockIdx.y
        const size_t s = blockIdx.y;
        y += s * ldm; x += s * L.n_pad; d += s * L.n_pad;
        touched2 += s * 2 * kcap;
        rhs_p = rhs + ((size_t)L.b_pad + s) * ldm;
        rhs += s * ldm;
        st += s;
    }
    if (st->done) return;
    typedef T V2 __attribute__((ext_vector_type(2)));
    __shared__ T s_r[4][kRpRows];
    __shared__ T s_p[4][kRpRows];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t row = blockIdx.x * kRpRows + lane * kRpRowsPerLane;     // ldm % 256 == 0
    const uint32_t nt = st->ntouched;
    const uint32_t* touched = touched2 + (size_t)st->cur * kcap;
    T ar0 = T(0), ar1 = T(0), ap0 = T(0), ap1 = T(0);
    uint32_t j = wave;
    for (; j + 12 < nt; j += 16) {
        uint32_t cj[4];
        V2 a[4];
        T xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            cj[u] = touched[j + 4 * u];
            a[u] = *reinterpret_cast<const V2*>(At + (size_t)cj[u] * ldm + row);
            xv[u] = x[cj[u]];
            dv[u] = d[cj[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ar0 += xv[u] * a[u][0]; ar1 += xv[u] * a[u][1];
            ap0 += dv[u] * a[u][0]; ap1 += dv[u] * a[u][1];
        }
    }
    for (; j < nt; j += 4) {
        const uint32_t c0 = touched[j];
        const V2 a0 = *reinterpret_cast<const V2*>(At + (size_t)c0 * ldm + row);
        const T xv = x[c0], dv = d[c0];
        ar0 += xv * a0[0]; ar1 += xv * a0[1];
        ap0 += dv * a0[0]; ap1 += dv * a0[1];
    }
    s_r[wave][lane * 2] = ar0; s_r[wave][lane * 2 + 1] = ar1;
    s_p[wave][lane * 2] = ap0; s_p[wave][lane * 2 + 1] = ap1;
    __syncthreads();
    if (threadIdx.x < kRpRows) {
        const uint32_t t = threadIdx.x, i = blockIdx.x * kRpRows + t;
        const T sr = ((s_r[0][t] + s_r[1][t]) + s_r[2][t]) + s_r[3][t];
        const T sp = ((s_p[0][t] + s_p[1][t]) + s_p[2][t]) + s_p[3][t];
        // padding rows stay exactly zero even if x or d went non-finite (0 * inf = NaN)
        rhs[i] = (i < L.m) ? (y[i] - sr) : T(0);
        rhs_p[i] = (i < L.m) ? sp : T(0);
    }
}

// ---- select_toggle: the serial end of find_max_gamma + inverse_add_or_remove's index part, run
// ---- by ONE workgroup once every workgroup's (min gamma, idx) partial is visible: final pick,
// ---- support toggle, x update.  Returns false when the solve terminated here (done raised).
template <typename T>
__device__ __forceinline__ bool select_toggle(uint32_t round, T c_inf, uint32_t ns,
                                              const T* pmin_val, const uint32_t* pmin_idx,
                                              T* __restrict__ x, const T* __restrict__ d, uint8_t* __restrict__ insup,
                                              uint32_t* __restrict__ gam2, uint32_t* __restrict__ touched2, uint32_t kcap,
                                              DevState* st, uint32_t* hflags, bool post_round,
                                              TraceEntry* trace, uint32_t trace_cap, int zero_on_removal,
                                              uint32_t* ndone, uint32_t nslots, const int32_t* __restrict__ slot_of,
                                              T* sv, uint32_t* si, uint32_t* s_cnt,
                                              uint32_t* o_idx = nullptr, uint32_t* o_rank = nullptr,
                                              uint32_t* o_added = nullptr, uint32_t* o_knew = nullptr, T* o_gamma = nullptr)
{
    // ---- last workgroup: final (gamma, idx) of find_max_gamma: smallest positive
    // candidate, left-most index; (T_MAX, 0) when there is none (homotopy-cpu.cpp:123-124)
        T g = Lim<T>::max();
    uint32_t idx = 0xffffffffu;
    for (uint32_t b = threadIdx.x; b < ns; b += blockDim.x) {
        const T ov = __hip_atomic_load(&pmin_val[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t oi = __hip_atomic_load(&pmin_idx[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (better_min(ov, oi, g, idx)) { g = ov; idx = oi; }
    }
    block_reduce_pair<T, false>(g, idx, sv, si);
    if (!(g < Lim<T>::max())) idx = 0;

    const uint32_t cur = st->cur;
    const uint32_t K = st->K;
    const uint32_t nt = st->ntouched;
    const uint32_t* gam = gam2 + (size_t)cur * kcap;
    uint32_t* gam_new = gam2 + (size_t)(cur ^ 1u) * kcap;
    const uint32_t* tch = touched2 + (size_t)cur * kcap;
    uint32_t* tch_new = touched2 + (size_t)(cur ^ 1u) * kcap;
    const bool added = insup[idx] == 0;

    // rank of idx in the sorted support / touched list (rank_index.h:65-83)
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t lr = 0, lt = 0;
    for (uint32_t j = threadIdx.x; j < K; j += blockDim.x) lr += (gam[j] < idx) ? 1u : 0u;
    for (uint32_t j = threadIdx.x; j < nt; j += blockDim.x) lt += (tch[j] < idx) ? 1u : 0u;
    if (lr) atomicAdd(&s_cnt[0], lr);
    if (lt) atomicAdd(&s_cnt[1], lt);
    __syncthreads();
    const uint32_t rank = s_cnt[0];
    const uint32_t trank = s_cnt[1];
    const uint32_t K_new = added ? K + 1 : K - 1;
    // touched list: sorted union of every support so far (the lists hold kcap entries)
    const bool seen = (trank < nt) && (tch[trank] == idx);
    const uint32_t nt_new = (added && !seen) ? nt + 1 : nt;

    if (trace != nullptr && threadIdx.x == 0 && round < trace_cap) {
        trace[round].idx = idx;
        trace[round].added = added ? 1u : 0u;
        trace[round].gamma = (double)g;
        trace[round].c_inf = (double)c_inf;
    }

    if (K_new == 0 || K_new > kcap || nt_new > kcap) {
        // K_new == 0: homotopy-cpu.cpp:248-249, the support became empty -> break before x
        // is updated; the report carries the c_inf of the previous iteration's end.
        // K_new > kcap (or more distinct columns touched than the lists hold): workspace exhausted.
        if (threadIdx.x == 0) {
            if (K_new == 0) {
                insup[idx] = 0;
                st->K = 0;
                st->idx = idx;
                st->rank = rank;
                st->added = 0;
                st->gamma = (double)g;
                st->iter = round;
            } else {
                st->status = SS_HIP_ECAPACITY;
                st->iter = round - 1;
            }
            st->c_inf = (double)c_inf;
            st->done_round = round;
            st->need_sweep = 0;
            st->done = 1;
            signal_done(hflags, ndone, nslots, round);
        }
        return false;
    }

    // x += gamma * direction over the OLD support (homotopy-cpu.cpp:252; d is zero elsewhere).
    // The column that leaves the support lands on x + (-x/d)*d, i.e. 0 up to an ulp; the
    // reference keeps that residue, and a residue of the wrong sign makes a later re-insertion
    // of the column bounce straight out again (gamma ~ 1e-18 steps).  By default the entry is
    // set to exactly 0 (option "zero_on_removal" = 0 restores the reference's residue).
    for (uint32_t j = threadIdx.x; j < K; j += blockDim.x) {
        const uint32_t col = gam[j];
        const T xn = x[col] + g * d[col];
        x[col] = (!added && zero_on_removal && col == idx) ? T(0) : xn;
    }

    // new sorted support, written out of place
    if (added) {
        for (uint32_t j = threadIdx.x; j < K_new; j += blockDim.x)
            gam_new[j] = (j < rank) ? gam[j] : (j == rank ? idx : gam[j - 1]);
    } else {
        for (uint32_t j = threadIdx.x; j < K_new; j += blockDim.x)
            gam_new[j] = gam[j + (j >= rank ? 1u : 0u)];
    }
    if (added && !seen) {
        for (uint32_t j = threadIdx.x; j < nt_new; j += blockDim.x)
            tch_new[j] = (j < trank) ? tch[j] : (j == trank ? idx : tch[j - 1]);
    } else {
        for (uint32_t j = threadIdx.x; j < nt_new; j += blockDim.x) tch_new[j] = tch[j];
    }

    if (threadIdx.x == 0) {
        insup[idx] = added ? 1 : 0;
        st->K = K_new;
        st->ntouched = nt_new;
        st->idx = idx;
        st->rank = rank;
        st->added = added ? 1u : 0u;
        st->gamma = (double)g;
        st->c_inf = (double)c_inf;
        st->iter = round;
        // lookahead engine: the inserted column needs its Gram column; sweep only if not cached
        st->need_sweep = (slot_of != nullptr && added && slot_of[idx] < 0) ? 1u : 0u;
        if (hflags && post_round) __hip_atomic_store(&hflags[0], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // the same values in registers, for a caller that goes on in this launch (DevState fields
    // are read through the scalar cache, which does not see this launch's vector stores)
    if (o_idx) { *o_idx = idx; *o_rank = rank; *o_added = added ? 1u : 0u; *o_knew = K_new; *o_gamma = g; }
    return true;
}

// ---- k_scansel: find_max_gamma's scan (homotopy-cpu.cpp:122-163) in every workgroup,
// ---- then loop control, pick, support toggle and x update in the last one to arrive ----
constexpr int kScanPerThread = 4;

template <typename T>
__global__ __launch_bounds__(kSmallThreads)
void k_scansel(uint32_t round, T tol, uint32_t max_iter, uint32_t n,
               const T* __restrict__ c, const T* __restrict__ q, T* __restrict__ x,
               const T* __restrict__ d, uint8_t* __restrict__ insup,
               const T* __restrict__ pmax_val, const uint32_t* __restrict__ pmax_idx, uint32_t nb,
               T* pmin_val, uint32_t* pmin_idx,
               uint32_t* __restrict__ gam2, uint32_t* __restrict__ touched2, SlotDims L,
               DevState* st, uint32_t* hflags, TraceEntry* trace, uint32_t trace_cap,
               int zero_on_removal, int tie_guard, uint32_t* ndone, uint32_t nslots,
               T* __restrict__ tcand, const int32_t* __restrict__ slot_of, int tie_exit)
{
    const uint32_t kcap = L.kcap;
    {   // slot = blockIdx.y
        const size_t s = blockIdx.y;
        c += s * L.n_pad; q += s * L.n_pad; x += s * L.n_pad; d += s * L.n_pad; insup += s * L.n_pad;
        pmax_val += s * L.pmax_stride; pmax_idx += s * L.pmax_stride;
        pmin_val += s * L.pmin_stride; pmin_idx += s * L.pmin_stride;
        gam2 += s * 2 * kcap; touched2 += s * 2 * kcap;
        st += s;
        if (s != 0) trace = nullptr;
    }
    if (st->done) return;
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_cnt[2];
    __shared__ uint32_t s_flag;

    // inf_norm of the correlations the sweep just produced (homotopy-cpu.cpp:270 / :219)
    T c_inf;
    uint32_t imax;
    reduce_sweep_partials(pmax_val, pmax_idx, nb, c_inf, imax, sv, si);

    // do { ... } while (iter < max_iter && c_inf > tolerance)   (homotopy-cpu.cpp:236,272)
    // round t starts iteration t, so the while-test of iteration t-1 is evaluated here;
    // every workgroup takes the same branch (same inputs, exact max).
    if ((round > 1 && !(c_inf > tol)) || round > max_iter) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->c_inf = (double)c_inf;
            st->iter = round - 1;
            st->done_round = round;
            st->need_sweep = 0;
            st->done = 1;
            signal_done(hflags, ndone, nslots, round);
        }
        return;
    }

    T best = Lim<T>::max();
    uint32_t best_i = 0xffffffffu;
    bool tie = false;      // an off-support candidate that is exactly 0 (see DevState::tie_stall)
    // (the column that LEFT the support in the previous iteration still sits on the boundary, |c| = lambda: its t = 0
    // is the rule, not a tie — the strict t > 0 exists to keep it out)
    const uint32_t just_removed = (round > 1 && st->added == 0u) ? st->idx : 0xffffffffu;
    const bool in_band = tie_band<T>(c_inf, (T)st->c_inf, (T)st->gamma, (T)st->lambda0);
    for (uint32_t base = blockIdx.x * (kSmallThreads * kScanPerThread); base < n;
         base += gridDim.x * (kSmallThreads * kScanPerThread))
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        const uint32_t i = base + k * kSmallThreads + threadIdx.x;
        if (i < n) {
            T m = Lim<T>::max();
            if (insup[i]) {
                const T t = -x[i] / d[i];
                if (t > T(0) && t < m) m = t;
            } else {
                const T qi = q[i], ci = c[i];
                const T dl = T(1) - qi, dr = T(1) + qi;
                // tie guard: an off-support column that ATTAINS the maximum (|c_i| == c_inf, so
                // t == 0) has overtaken the support by rounding — two columns reached the
                // boundary within an ulp and the other one was inserted first.  The reference's
                // strict `t > 0` then skips it for good and the path never reaches the solution;
                // with the guard it is inserted by a zero-length step (option "tie_guard").
                if (dl != T(0)) {
                    T t = (c_inf - ci) / dl;
                    if (tie_guard && t == T(0) && dl > T(0)) t = Lim<T>::tiny();
                    if (t == T(0) && i != just_removed && in_band) tie = true;
                    if (t > T(0) && t < m) m = t;
                }
                if (dr != T(0)) {
                    T t = (c_inf + ci) / dr;
                    if (tie_guard && t == T(0) && dr > T(0)) t = Lim<T>::tiny();
                    if (t == T(0) && i != just_removed && in_band) tie = true;
                    if (t > T(0) && t < m) m = t;
                }
            }
            // lookahead ranking: only columns that are neither active nor cached compete
            if (tcand != nullptr) tcand[i] = (insup[i] || slot_of[i] >= 0) ? Lim<T>::max() : m;
            if (better_min(m, i, best, best_i)) { best = m; best_i = i; }
        }
    }
    if (tie) __hip_atomic_store(&st->tie_stall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    block_reduce_pair<T, false>(best, best_i, sv, si);
    if (threadIdx.x == 0) {
        // the partials are the only data that crosses workgroups in this launch (x, d, insup and the
        // lists are inputs): moved with L2-bypassing stores / loads, so the ticket needs no cache fences
        __hip_atomic_store(&pmin_val[blockIdx.x], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pmin_idx[blockIdx.x], best_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!arrive_last_relaxed(&st->ticket_scan, gridDim.x, &s_flag)) return;

    // tie stall (the flag moved with L2-bypassing stores before the tickets): the host re-runs this signal in the
    // reference-order engine — no point in following a path the reference's own rounding may not take
    if (tie_exit && __hip_atomic_load(&st->tie_stall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        if (threadIdx.x == 0) {
            st->status = kStatusTieRerun;
            st->c_inf = (double)c_inf;
            st->iter = round - 1;
            st->done_round = round;
            st->need_sweep = 0;
            st->done = 1;
            signal_done(hflags, ndone, nslots, round);
        }
        return;
    }
    select_toggle<T>(round, c_inf, gridDim.x, pmin_val, pmin_idx, x, d, insup, gam2, touched2, kcap, st, hflags,
                     true, trace, trace_cap, zero_on_removal, ndone, nslots, slot_of, sv, si, s_cnt);
}

// =========================================================================================
// Lookahead engine (fp32 single signal).  Instead of sweeping A every iteration, the Gram
// columns g_j = A^T a_j of the active columns are cached: with them
//     c = A^T y - sum_j x_j g_j ,   q = sum_j d_j g_j ,   u1 = g_idx[Gamma] ,  a_idx.a_idx = g_idx[idx]
// are O(K n) gathers, and A is only swept when a column enters whose g is not cached — and that
// sweep fetches, in the same pass over A, the g of the 31 columns most likely to enter next
// (smallest step-length candidates of find_max_gamma), so most iterations need no sweep at all.
// Same algorithm, same decisions; only the order of floating-point operations in c and q
// changes (Gram form instead of residual form).
// =========================================================================================

// first pick after c0 = A^T y: idx = argmax |c0| (homotopy-cpu.cpp:217-221)
template <typename T>
__global__ __launch_bounds__(kSmallThreads)
void k_la_init_pick(const T* __restrict__ pmax_val, const uint32_t* __restrict__ pmax_idx, uint32_t nb,
                    uint8_t* __restrict__ insup, uint32_t* __restrict__ gam, uint32_t* __restrict__ touched,
                    DevState* st, TraceEntry* trace, T tol, T gram_guard, uint32_t* hflags, uint32_t full_rows)
{
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    T c_inf;
    uint32_t idx;
    reduce_sweep_partials(pmax_val, pmax_idx, nb, c_inf, idx, sv, si);
    // Gram form computes c = c0 - sum_j x_j g_j: its absolute error is ~eps * ||c0||_inf * sqrt(K), which
    // must stay far below the lambda the path is asked to reach.  A tolerance below gram_guard * ||c0||_inf
    // is answered with "retry in residual form" (the host re-runs the solve with one sweep per iteration).
    if (gram_guard > T(0) && tol < gram_guard * c_inf) {
        if (threadIdx.x == 0) {
            st->status = kStatusRetryResidual;
            st->need_sweep = 0;
            st->done = 1;
            signal_done(hflags, nullptr, 1u, 0u);
        }
        return;
    }
    if (threadIdx.x == 0) {
        insup[idx] = 1;
        gam[0] = idx;
        touched[0] = idx;
        st->done = 0; st->status = 0; st->iter = 0;
        st->K = 1; st->ntouched = 1; st->idx = idx; st->rank = 0; st->added = 1;
        st->cur = 1;    // the update kernel reads the new lists from buffer cur^1 = 0 and flips
        st->done_round = 0;
        st->c_inf = (double)c_inf;
        st->gamma = 0.0;
        st->lambda0 = (float)c_inf;
        // (full_rows != 0: the full Gram matrix is the cache — every column is there already)
        st->need_sweep = full_rows ? 0u : 1u; st->cache_used = full_rows; st->nsweeps = 0;
        if (trace != nullptr) { trace[0].idx = idx; trace[0].added = 1; trace[0].gamma = 0.0; trace[0].c_inf = (double)c_inf; }
    }
}

// choose the columns of the next lookahead sweep: the entering column first, then not-yet-
// cached columns with small step-length candidates (init: large |c0|).  The ranking is a
// prefetch heuristic, not part of the algorithm's decisions, so it is approximate on purpose:
// every thread offers the best of its n/1024 strided columns and the 32 best offers win.
constexpr int kTopS = 32;                   // columns of a sweep (the first sweep of a fp32 solve may take 64: nsel)
constexpr int kSwStride = 64;               // sw_list layout: rcols[kSwStride] then drows[kSwStride]

template <typename T>
__global__ __launch_bounds__(kUpdThreads)
void k_la_top(const T* __restrict__ tcand, const T* __restrict__ c, uint32_t n, int init_mode,
              const uint8_t* __restrict__ insup, int32_t* __restrict__ slot_of, uint32_t gcap,
              uint32_t* __restrict__ sw_list, DevState* st, uint32_t* hflags, uint32_t* __restrict__ slot_col,
              uint32_t nsel)
{
    if (st->done || !st->need_sweep) return;
    const uint32_t idx = st->idx;
    const uint32_t tid = threadIdx.x;

    // this thread's two best offers among its columns (MAX = nothing to offer).  In the scan
    // mode tcand already carries MAX for active and cached columns; in init mode nothing is
    // cached yet and only idx is active.
    T v1 = Lim<T>::max(), v2 = Lim<T>::max();
    uint32_t i1 = 0xffffffffu, i2 = 0xffffffffu;
    const T* __restrict__ src = init_mode ? c : tcand;
    for (uint32_t i0 = tid; i0 < n; i0 += 8 * kUpdThreads) {
        T raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                               // eight independent loads in flight
            const uint32_t i = i0 + (uint32_t)u * kUpdThreads;
            raw[u] = i < n ? src[i] : Lim<T>::max();
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = i0 + (uint32_t)u * kUpdThreads;
            if (i >= n) break;
            T v = raw[u];
            if (init_mode) v = v < T(0) ? v : -v;                   // -|c|: large correlations first
            if (init_mode == 2 && (insup[i] || slot_of[i] >= 0)) v = Lim<T>::max();   // OMP: active / cached columns are out
            if (i == idx) v = -Lim<T>::max();                       // the entering column always wins
            if (better_min(v, i, v1, i1)) { v2 = v1; i2 = i1; v1 = v; i1 = i; }
            else if (better_min(v, i, v2, i2)) { v2 = v; i2 = i; }
        }
    }
    // stage 1: every wave extracts the 8 best of its 128 offers (wave-level reductions only)
    constexpr int kPerWave = 8;
    constexpr int kNW = kUpdThreads / 64;
    __shared__ T s_cv[kNW * kPerWave];
    __shared__ uint32_t s_ci[kNW * kPerWave];
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = 0; r < kPerWave; ++r) {
        T bv = v1;
        uint32_t bi = i1;
        wave_reduce_pair<T, false>(bv, bi);
        if (bi == i1 && bi != 0xffffffffu) { v1 = v2; i1 = i2; v2 = Lim<T>::max(); i2 = 0xffffffffu; }   // offer taken
        if (lane == 0) { s_cv[wave * kPerWave + r] = bv; s_ci[wave * kPerWave + r] = bi; }
    }
    __syncthreads();
    if (wave != 0) return;
    // stage 2: one wave ranks the 128 finalists and hands out cache slots, best first
    constexpr int kFinal = kNW * kPerWave;
    static_assert(kFinal <= 128, "two finalists per lane");
    T f1 = s_cv[lane];
    uint32_t j1 = s_ci[lane];
    T f2 = Lim<T>::max();
    uint32_t j2 = 0xffffffffu;
    if (lane + 64 < kFinal) { f2 = s_cv[lane + 64]; j2 = s_ci[lane + 64]; }
    if (better_min(f2, j2, f1, j1)) { const T tv = f1; f1 = f2; f2 = tv; const uint32_t ti = j1; j1 = j2; j2 = ti; }
    uint32_t used = st->cache_used;
    uint32_t count = 0;
    for (uint32_t sidx = 0; sidx < nsel; ++sidx) {
        T bv = f1;
        uint32_t bi = j1;
        wave_reduce_pair<T, false>(bv, bi);
        if (!(bv < Lim<T>::max()) || bi == 0xffffffffu || used >= gcap) break;   // uniform
        if (bi == j1) { f1 = f2; j1 = j2; f2 = Lim<T>::max(); j2 = 0xffffffffu; }
        if (lane == 0) {
            sw_list[count] = bi;               // rcols: right-hand side = column bi of A
            sw_list[kSwStride + count] = used; // drows: output row = cache slot
            slot_of[bi] = (int32_t)used;
            if (slot_col != nullptr) slot_col[used] = bi;      // (the speculative form lists the cached columns)
        }
        ++used;
        ++count;
    }
    if (lane == 0) {
        for (uint32_t s2 = count; s2 < (uint32_t)kSwStride; ++s2) { sw_list[s2] = 0xffffffffu; sw_list[kSwStride + s2] = 0xffffffffu; }
        st->cache_used = used;
        st->nsweeps += 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (slot_of[idx] < 0) {                 // cache budget exhausted: the host re-runs the solve in residual form
            st->status = kStatusRetryResidual;
            st->need_sweep = 0;
            st->done = 1;
            signal_done(hflags, nullptr, 1u, st->iter);
        }
    }
}

// four consecutive elements with 16-byte loads (rows of the Gram cache, c0): a wave reads 1 KiB (fp32) per instruction
// where four strided 4-byte loads per lane moved 256 bytes each
__device__ __forceinline__ void load4(const float* __restrict__ p, float (&v)[4])
{
    const v4f t = *reinterpret_cast<const v4f*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
__device__ __forceinline__ void load4(const double* __restrict__ p, double (&v)[4])
{
    const v2d a = *reinterpret_cast<const v2d*>(p), b = *reinterpret_cast<const v2d*>(p + 2);
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
}

// the four columns of a thread: VEC — consecutive (base + 4 tid .. + 3, one 16-byte load per row); else strided by the
// workgroup (base + tid + 256 k, four 4-byte loads per row: each wave-instruction reads 256 contiguous bytes)
// (lim: padded columns left from the workgroup's base — a multiple of 256, so the guards are uniform per workgroup and
// per k: rows of c0 hold n_pad columns, a multiple of 256 but not of the workgroup's run)
template <bool VEC, typename T>
__device__ __forceinline__ void load_cols(const T* __restrict__ p, T (&v)[4], uint32_t lim)
{
    if (VEC) {
        if (4u * threadIdx.x < lim) load4(p, v);
        else { v[0] = v[1] = v[2] = v[3] = T(0); }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ((uint32_t)k * kSmallThreads < lim) ? p[k * kSmallThreads] : T(0);
    }
}
// the same with N columns per thread (strided form; VEC only with N == 4)
// (lim: padded columns left from the workgroup's base, a multiple of 256: a wide workgroup at the end of a row reads
// only the 256-column groups that exist — uniform per workgroup)
template <bool VEC, int N, typename T>
__device__ __forceinline__ void load_colsN(const T* __restrict__ p, T (&v)[N], uint32_t lim)
{
    static_assert(!VEC || N == 4, "16-byte form: four columns per thread");
    if (VEC) {
        T t[4] = { T(0), T(0), T(0), T(0) };
        if (4u * threadIdx.x < lim) load4(p, t);
        for (int k = 0; k < 4 && k < N; ++k) v[k] = t[k];
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = ((uint32_t)k * kSmallThreads < lim) ? p[k * kSmallThreads] : T(0);
    }
}
template <bool VEC>
__device__ __forceinline__ uint32_t col_of(uint32_t base, uint32_t tid, int k)
{
    return VEC ? base + 4u * tid + (uint32_t)k : base + (uint32_t)k * kSmallThreads + tid;
}

// c = c0 - sum_j x_j g_j ; q = sum_j d_j g_j over the touched columns; partial max |c|
constexpr uint32_t kCqChunk = 1024;
constexpr uint32_t kCqTile = 256;        // touched columns staged in LDS per pass

template <typename T, bool VEC>
__global__ __launch_bounds__(kSmallThreads)
void k_la_cq(const T* __restrict__ gcache, const int32_t* __restrict__ slot_of, const T* __restrict__ c0,
             const T* __restrict__ x, const T* __restrict__ d, const uint32_t* __restrict__ touched2,
             uint32_t n, uint32_t gpitch, SlotDims L, T* __restrict__ c, T* __restrict__ q,
             T* __restrict__ pmax_val, uint32_t* __restrict__ pmax_idx, const DevState* st)
{
    {   // slot = blockIdx.y (batched Gram form: one signal per slot; gcache = the full A^T A and slot_of = null,
        // or gcache = the batch's column cache and slot_of = one row table per slot)
        const size_t s = blockIdx.y;
        c0 += s * L.n_pad; x += s * L.n_pad; d += s * L.n_pad; c += s * L.n_pad; q += s * L.n_pad;
        if (slot_of != nullptr) slot_of += s * L.n_pad;
        touched2 += s * 2 * L.kcap;
        pmax_val += s * L.pmax_stride; pmax_idx += s * L.pmax_stride;
        st += s;
    }
    if (st->done) return;
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_slot[kCqTile];
    __shared__ T s_x[kCqTile];
    __shared__ T s_d[kCqTile];
    const uint32_t nt = st->ntouched;
    const uint32_t* touched = touched2 + (size_t)st->cur * L.kcap;
    const uint32_t base = blockIdx.x * kCqChunk;
    const uint32_t lim = L.n_pad - base;                       // (padded columns left from here)
    const T* gbase = gcache + base + (VEC ? 4u : 1u) * threadIdx.x;               // gpitch % 1024 == 0: rows never run out
    T ax[4] = { T(0), T(0), T(0), T(0) }, ad[4] = { T(0), T(0), T(0), T(0) };
    for (uint32_t j0 = 0; j0 < nt; j0 += kCqTile) {
        const uint32_t cnt = (nt - j0 < kCqTile) ? (nt - j0) : kCqTile;
        __syncthreads();
        if (threadIdx.x < cnt) {
            const uint32_t col = touched[j0 + threadIdx.x];
            s_slot[threadIdx.x] = slot_of != nullptr ? (uint32_t)slot_of[col] : col;
            s_x[threadIdx.x] = x[col];
            s_d[threadIdx.x] = d[col];
        }
        __syncthreads();
        uint32_t j = 0;
        for (; j + 4 <= cnt; j += 4) {
            T gv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) load_cols<VEC>(gbase + (size_t)s_slot[j + u] * gpitch, gv[u], lim);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T xj = s_x[j + u], dj = s_d[j + u];
#pragma unroll
                for (int k = 0; k < 4; ++k) { ax[k] += xj * gv[u][k]; ad[k] += dj * gv[u][k]; }
            }
        }
        for (; j < cnt; ++j) {
            T gv1[4];
            load_cols<VEC>(gbase + (size_t)s_slot[j] * gpitch, gv1, lim);
            const T xj = s_x[j], dj = s_d[j];
#pragma unroll
            for (int k = 0; k < 4; ++k) { ax[k] += xj * gv1[k]; ad[k] += dj * gv1[k]; }
        }
    }
    T bv = T(-1);
    uint32_t bi = 0xffffffffu;
    T c0v[4];
    load_cols<VEC>(c0 + base + (VEC ? 4u : 1u) * threadIdx.x, c0v, lim);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = col_of<VEC>(base, threadIdx.x, k);
        if (i < n) {
            const T cv = c0v[k] - ax[k];
            c[i] = cv;
            q[i] = ad[k];
            const T a = cv < T(0) ? -cv : cv;
            if (better_max(a, i, bv, bi)) { bv = a; bi = i; }
        }
    }
    block_reduce_pair<T, true>(bv, bi, sv, si);
    if (threadIdx.x == 0) { pmax_val[blockIdx.x] = bv; pmax_idx[blockIdx.x] = bi; }
}

// ---- k_la_cqs: k_la_cq and k_scansel in ONE launch (batched Gram forms, option batch_fused_scan) ----------
// k_scansel re-reads, per signal and round, the c and q that k_la_cq wrote microseconds earlier (9 bytes per
// column: 2.4 GB per round for 4096 signals of 65536 columns — at 2 TB/s the HBM kernel furthest below its
// roof).  Here a workgroup keeps c and q of its 1024 columns in REGISTERS across the one thing the scan has to
// wait for — lambda = ||c||_inf over all columns of the signal: the workgroups of a signal meet at a per-signal
// counter (DevState::bar_count, monotonic over the rounds of the batch; target = round * gridDim.x), read each
// other's partial maxima (L2-bypassing), and go on to find_max_gamma's scan (homotopy-cpu.cpp:122-163) with the
// same expressions and predicates as k_scansel.  The last one to arrive picks, toggles and updates x
// (select_toggle) as before.  c and q are written only where someone still reads them: on the support and for
// each workgroup's best candidate (the entering column is one of those; k_gramupd takes its sign from there).
// Same bits as the two-kernel form (tested).
// Residency: the workgroups of a signal are 64 consecutive block ids, dispatched in order, so the oldest unfinished
// signal always has all of its workgroups resident; the wait is bounded all the same (kCqsSpinLimit): on expiry the
// slot ends with SS_HIP_ERUNTIME instead of hanging the queue.
constexpr uint32_t kCqsSpinLimit = 1u << 22;

template <typename T, bool VEC, int CPT, int RIF>
__global__ __launch_bounds__(kSmallThreads)
void k_la_cqs(const T* __restrict__ gcache, const int32_t* __restrict__ slot_of, const T* __restrict__ c0,
              T* __restrict__ x, const T* __restrict__ d, uint32_t* __restrict__ touched2, uint32_t* __restrict__ gam2,
              uint32_t n, uint32_t gpitch, SlotDims L, T* __restrict__ c, T* __restrict__ q,
              T* pmax_val, uint32_t* pmax_idx, T* pmin_val, uint32_t* pmin_idx, uint8_t* __restrict__ insup,
              DevState* st, uint32_t round, T tol, uint32_t max_iter, uint32_t* hflags, TraceEntry* trace, uint32_t trace_cap,
              int zero_on_removal, int tie_guard, uint32_t* ndone, uint32_t nslots, int tie_exit)
{
    const uint32_t kcap = L.kcap;
    const int32_t* slot_tab = slot_of;
    {
        const size_t s = blockIdx.y;
        c0 += s * L.n_pad; x += s * L.n_pad; d += s * L.n_pad; c += s * L.n_pad; q += s * L.n_pad; insup += s * L.n_pad;
        if (slot_tab != nullptr) slot_tab += s * L.n_pad;
        touched2 += s * 2 * kcap; gam2 += s * 2 * kcap;
        pmax_val += s * L.pmax_stride; pmax_idx += s * L.pmax_stride;
        pmin_val += s * L.pmin_stride; pmin_idx += s * L.pmin_stride;
        st += s;
        if (s != 0) trace = nullptr;
    }
    if (st->done) return;
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_cnt[2];
    __shared__ uint32_t s_flag;
    __shared__ uint32_t s_slot[kCqTile];
    __shared__ T s_x[kCqTile];
    __shared__ T s_d[kCqTile];
    const uint32_t nt = st->ntouched;
    const uint32_t* touched = touched2 + (size_t)st->cur * kcap;
    const uint32_t base = blockIdx.x * (uint32_t)(kSmallThreads * CPT);      // CPT columns per thread
    const uint32_t lim = L.n_pad - base;                                      // (n_pad, a multiple of 256, is what rows hold at least)
    const T* gbase = gcache + base + (VEC ? 4u : 1u) * threadIdx.x;
    T ax[CPT], ad[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) { ax[k] = T(0); ad[k] = T(0); }
    for (uint32_t j0 = 0; j0 < nt; j0 += kCqTile) {                  // (the loop of k_la_cq, statement for statement)
        const uint32_t cnt = (nt - j0 < kCqTile) ? (nt - j0) : kCqTile;
        __syncthreads();
        if (threadIdx.x < cnt) {
            const uint32_t col = touched[j0 + threadIdx.x];
            s_slot[threadIdx.x] = slot_tab != nullptr ? (uint32_t)slot_tab[col] : col;
            s_x[threadIdx.x] = x[col];
            s_d[threadIdx.x] = d[col];
        }
        __syncthreads();
        uint32_t j = 0;
        for (; j + RIF <= cnt; j += RIF) {                       // RIF rows of G in flight per thread
            T gv[RIF][CPT];
#pragma unroll
            for (int u = 0; u < RIF; ++u) load_colsN<VEC, CPT>(gbase + (size_t)s_slot[j + u] * gpitch, gv[u], lim);
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                const T xj = s_x[j + u], dj = s_d[j + u];
#pragma unroll
                for (int k = 0; k < CPT; ++k) { ax[k] += xj * gv[u][k]; ad[k] += dj * gv[u][k]; }
            }
        }
        for (; j < cnt; ++j) {
            T gv1[CPT];
            load_colsN<VEC, CPT>(gbase + (size_t)s_slot[j] * gpitch, gv1, lim);
            const T xj = s_x[j], dj = s_d[j];
#pragma unroll
            for (int k = 0; k < CPT; ++k) { ax[k] += xj * gv1[k]; ad[k] += dj * gv1[k]; }
        }
    }
    T cv[CPT], qv[CPT];
    uint8_t act[CPT];
    T bv = T(-1);
    uint32_t bi = 0xffffffffu;
    T c0v[CPT];
    load_colsN<VEC, CPT>(c0 + base + (VEC ? 4u : 1u) * threadIdx.x, c0v, lim);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const uint32_t i = col_of<VEC>(base, threadIdx.x, k);
        cv[k] = T(0); qv[k] = T(0); act[k] = 0;
        if (i < n) {
            cv[k] = c0v[k] - ax[k];
            qv[k] = ad[k];
            act[k] = insup[i];
            const T a = cv[k] < T(0) ? -cv[k] : cv[k];
            if (better_max(a, i, bv, bi)) { bv = a; bi = i; }
        }
    }
    block_reduce_pair<T, true>(bv, bi, sv, si);
    // ---- the meeting: every workgroup of the signal posts its maximum, then reads all of them ------------------
    if (threadIdx.x == 0) {
        __hip_atomic_store(&pmax_val[blockIdx.x], bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pmax_idx[blockIdx.x], bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&st->bar_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t target = round * gridDim.x;
        uint32_t ok = 0;
        for (uint32_t spin = 0; spin < kCqsSpinLimit; ++spin) {
            if (__hip_atomic_load(&st->bar_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        s_flag = ok;
    }
    __syncthreads();
    if (s_flag == 0u) {
        // the signal's workgroups were not resident together: give up on this slot rather than hang
        if (threadIdx.x == 0) {
            __hip_atomic_store(&st->status, (uint32_t)SS_HIP_ERUNTIME, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) { st->done_round = round; st->need_sweep = 0; st->done = 1; signal_done(hflags, ndone, nslots, round); }
        }
        return;
    }
    T c_inf;
    uint32_t imax;
    reduce_partials_agent(pmax_val, pmax_idx, gridDim.x, c_inf, imax, sv, si);
    // do { ... } while (iter < max_iter && c_inf > tolerance)   (homotopy-cpu.cpp:236,272; as in k_scansel)
    if ((round > 1 && !(c_inf > tol)) || round > max_iter) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->c_inf = (double)c_inf;
            st->iter = round - 1;
            st->done_round = round;
            st->need_sweep = 0;
            st->done = 1;
            signal_done(hflags, ndone, nslots, round);
        }
        return;
    }
    // ---- find_max_gamma's scan over the columns in registers (k_scansel's expressions) ---------------------------
    const uint32_t just_removed = (round > 1 && st->added == 0u) ? st->idx : 0xffffffffu;
    const bool in_band = tie_band<T>(c_inf, (T)st->c_inf, (T)st->gamma, (T)st->lambda0);
    T best = Lim<T>::max();
    uint32_t best_i = 0xffffffffu;
    T best_c = T(0), best_q = T(0);
    bool tie = false;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const uint32_t i = col_of<VEC>(base, threadIdx.x, k);
        if (i < n) {
            T m = Lim<T>::max();
            if (act[k]) {
                const T t = -x[i] / d[i];
                if (t > T(0) && t < m) m = t;
                c[i] = cv[k];                        // the support's correlations: k_gramupd forms sign(c - gamma q) from them
                q[i] = qv[k];
            } else {
                const T qi = qv[k], ci = cv[k];
                const T dl = T(1) - qi, dr = T(1) + qi;
                if (dl != T(0)) {
                    T t = (c_inf - ci) / dl;
                    if (tie_guard && t == T(0) && dl > T(0)) t = Lim<T>::tiny();
                    if (t == T(0) && i != just_removed && in_band) tie = true;
                    if (t > T(0) && t < m) m = t;
                }
                if (dr != T(0)) {
                    T t = (c_inf + ci) / dr;
                    if (tie_guard && t == T(0) && dr > T(0)) t = Lim<T>::tiny();
                    if (t == T(0) && i != just_removed && in_band) tie = true;
                    if (t > T(0) && t < m) m = t;
                }
            }
            if (better_min(m, i, best, best_i)) { best = m; best_i = i; best_c = cv[k]; best_q = qv[k]; }
        }
    }
    if (tie) __hip_atomic_store(&st->tie_stall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        // the workgroup's best candidate, with its c and q: the thread that owns it stores them (the entering column of
        // the signal is the best of some workgroup)
        const T mine = best;
        const uint32_t mine_i = best_i;
        block_reduce_pair<T, false>(best, best_i, sv, si);
        if (mine_i == best_i && mine == best && best_i != 0xffffffffu) { c[best_i] = best_c; q[best_i] = best_q; }
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&pmin_val[blockIdx.x], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pmin_idx[blockIdx.x], best_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!arrive_last_relaxed(&st->ticket_scan, gridDim.x, &s_flag)) return;
    if (tie_exit && __hip_atomic_load(&st->tie_stall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        if (threadIdx.x == 0) {
            st->status = kStatusTieRerun;
            st->c_inf = (double)c_inf;
            st->iter = round - 1;
            st->done_round = round;
            st->need_sweep = 0;
            st->done = 1;
            signal_done(hflags, ndone, nslots, round);
        }
        return;
    }
    select_toggle<T>(round, c_inf, gridDim.x, pmin_val, pmin_idx, x, d, insup, gam2, touched2, kcap, st, hflags,
                     true, trace, trace_cap, zero_on_removal, ndone, nslots, (const int32_t*)nullptr, sv, si, s_cnt);
}

// ---- k_omp_select: orthogonal matching pursuit's pick (one workgroup per slot) -------------
// c = A^T r has just been swept: idx = argmax |c| (first index), loop control
//   while (iter < max_iter && c_inf > tol) { insert idx; x_S = lstsq; r = y - A_S x_S; }
// There is no OMP in the reference; the report mirrors homotopy_report {iter, ||A^T r||_inf}.
template <typename T>
__global__ __launch_bounds__(kSmallThreads)
void k_omp_select(uint32_t round, T tol, uint32_t max_iter,
                  const T* __restrict__ pmax_val, const uint32_t* __restrict__ pmax_idx, uint32_t nb,
                  uint8_t* __restrict__ insup, uint32_t* __restrict__ gam2, uint32_t* __restrict__ touched2,
                  SlotDims L, DevState* st, uint32_t* hflags, TraceEntry* trace, uint32_t trace_cap,
                  uint32_t* ndone, uint32_t nslots)
{
    const uint32_t kcap = L.kcap;
    {
        const size_t s = blockIdx.y;
        pmax_val += s * L.pmax_stride; pmax_idx += s * L.pmax_stride;
        insup += s * L.n_pad;
        gam2 += s * 2 * kcap; touched2 += s * 2 * kcap;
        st += s;
        if (s != 0) trace = nullptr;
    }
    if (st->done) return;
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_cnt;
    T c_inf;
    uint32_t idx;
    reduce_sweep_partials(pmax_val, pmax_idx, nb, c_inf, idx, sv, si);

    const uint32_t cur = st->cur;
    const uint32_t K = st->K;
    // stop: tolerance reached, iteration budget spent, the support is full, or the best
    // column is already active (its correlation should be ~0: numerical stall)
    const bool stall = insup[idx] != 0;
    if (!(c_inf > tol) || round > max_iter || K >= kcap || stall) {
        if (threadIdx.x == 0) {
            st->c_inf = (double)c_inf;
            st->iter = round - 1;
            st->done_round = round;
            if (K >= kcap && c_inf > tol && round <= max_iter && !stall) st->status = SS_HIP_ECAPACITY;
            st->done = 1;
            signal_done(hflags, ndone, nslots, round);
        }
        return;
    }
    const uint32_t* gam = gam2 + (size_t)cur * kcap;
    uint32_t* gam_new = gam2 + (size_t)(cur ^ 1u) * kcap;
    uint32_t* tch_new = touched2 + (size_t)(cur ^ 1u) * kcap;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t lr = 0;
    for (uint32_t j = threadIdx.x; j < K; j += blockDim.x) lr += (gam[j] < idx) ? 1u : 0u;
    if (lr) atomicAdd(&s_cnt, lr);
    __syncthreads();
    const uint32_t rank = s_cnt;
    const uint32_t K_new = K + 1;
    for (uint32_t j = threadIdx.x; j < K_new; j += blockDim.x) {
        const uint32_t v = (j < rank) ? gam[j] : (j == rank ? idx : gam[j - 1]);
        gam_new[j] = v;
        tch_new[j] = v;          // the residual kernel walks the `touched` list == support
    }
    if (threadIdx.x == 0) {
        insup[idx] = 1;
        st->K = K_new;
        st->ntouched = K_new;
        st->idx = idx;
        st->rank = rank;
        st->added = 1;
        st->gamma = 0.0;
        st->c_inf = (double)c_inf;
        st->iter = round;
        if (trace != nullptr && round < trace_cap) {
            trace[round].idx = idx;
            trace[round].added = 1;
            trace[round].gamma = 0.0;
            trace[round].c_inf = (double)c_inf;
        }
        if (hflags) __hip_atomic_store(&hflags[0], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- update_direction: online_column_inverse's bordering / deflation (online_inverse.h:224-248,
// ---- 275-290) and the new direction (homotopy-cpu.cpp:257-267), run by ONE workgroup once u1 and
// ---- st->dot are visible.
constexpr int kMvRows = 8, kMvSlices = 2;   // rows x 64-element slices of a matrix-vector product a wave has in flight
constexpr int kInvBatch = 8;      // elements of the new inverse a thread has in flight (global memory: latency-bound)

template <typename T>
__device__ __forceinline__ void update_direction(uint32_t cur, uint32_t K_new, uint32_t rank, bool added,
                                                 const uint32_t* __restrict__ gam_old, const uint32_t* __restrict__ gam_new,
                                                 T* inv0, T* inv1, T* u1, T* u2, T* sgn,
                                                 const T* __restrict__ c, const T* __restrict__ q, T* __restrict__ d,
                                                 T tol, DevState* st, int omp, T* __restrict__ x, int first, int strict_sign,
                                                 uint32_t kcap, T* sv, T* s_dp, T g, bool cq_agent = false)
{
    T& s_d = *s_dp;
    // ---- last workgroup --------------------------------------------------------------
    const T* Iold = cur ? inv1 : inv0;
    T* Inew = cur ? inv0 : inv1;
    const uint32_t K_old = added ? K_new - 1 : K_new + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NW = (int)(blockDim.x >> 6);
    const size_t P = kcap;

    if (added) {
        const uint32_t nn = K_old;
        // u2 = inv * u1 (online_inverse.h:224-225), one wave per row; four rows at a time so that their
        // loads overlap (the inverse lives in global memory here: a row at a time is one memory round
        // trip per row — 40 us of an fp64 iteration at K = 128).  Per row the sum is formed as before.
        for (uint32_t i0 = wave; i0 < nn; i0 += kMvRows * NW) {
            T acc[kMvRows];
#pragma unroll
            for (int r = 0; r < kMvRows; ++r) acc[r] = T(0);
            for (uint32_t j0 = lane; j0 < nn; j0 += 64 * kMvSlices) {
                T uj[kMvSlices], v[kMvRows][kMvSlices];
#pragma unroll
                for (int t = 0; t < kMvSlices; ++t) { const uint32_t j = j0 + 64u * (uint32_t)t; uj[t] = j < nn ? u1[j] : T(0); }
#pragma unroll
                for (int r = 0; r < kMvRows; ++r)
#pragma unroll
                    for (int t = 0; t < kMvSlices; ++t) {
                        const uint32_t i = i0 + (uint32_t)r * NW, j = j0 + 64u * (uint32_t)t;
                        v[r][t] = (i < nn && j < nn) ? Iold[i * P + j] : T(0);
                    }
#pragma unroll
                for (int t = 0; t < kMvSlices; ++t)
#pragma unroll
                    for (int r = 0; r < kMvRows; ++r)
                        if (j0 + 64u * (uint32_t)t < nn) acc[r] += v[r][t] * uj[t];      // per row: j ascending, as before
            }
#pragma unroll
            for (int r = 0; r < kMvRows; ++r) {
                const uint32_t i = i0 + (uint32_t)r * NW;
                const T a = wave_sum(acc[r]);
                if (lane == 0 && i < nn) u2[i] = a;
            }
        }
        __syncthreads();
        // d = 1 / (dot - u1.u2) (online_inverse.h:228)
        T part = T(0);
        for (uint32_t j = threadIdx.x; j < nn; j += blockDim.x) part += u1[j] * u2[j];
        const T s = block_sum(part, sv);
        if (threadIdx.x == 0) {
            const T dotv = (T)load_handoff(&st->dot);
            if (nn == 0) {
                // first column: inv = [1 / ||col||^2] through the norm (online_inverse.h:193-201)
                const T nrm = sqrt(dotv);
                s_d = T(1) / (nrm * nrm);
            } else {
                s_d = T(1) / (dotv - s);
            }
        }
        __syncthreads();
        const T dv = s_d;
        // new inverse in sorted order: [inv + d u2 u2^T, -d u2; -d u2^T, d] with the new
        // row/column at position `rank` (online_inverse.h:229-248)
        const uint32_t tot = K_new * K_new;
        for (uint32_t e0 = threadIdx.x; e0 < tot; e0 += kInvBatch * blockDim.x) {
            T v[kInvBatch];
#pragma unroll
            for (int r = 0; r < kInvBatch; ++r) {                     // independent elements per thread and step
                const uint32_t e = e0 + (uint32_t)r * blockDim.x;
                v[r] = T(0);
                if (e >= tot) continue;
                const uint32_t a = e / K_new, b = e - a * K_new;
                if (a == rank && b == rank) {
                    v[r] = dv;
                } else if (a == rank) {
                    v[r] = -dv * u2[b - (b > rank ? 1u : 0u)];
                } else if (b == rank) {
                    v[r] = -dv * u2[a - (a > rank ? 1u : 0u)];
                } else {
                    const uint32_t oa = a - (a > rank ? 1u : 0u), ob = b - (b > rank ? 1u : 0u);
                    v[r] = Iold[oa * P + ob] + (dv * u2[oa]) * u2[ob];
                }
            }
#pragma unroll
            for (int r = 0; r < kInvBatch; ++r) {
                const uint32_t e = e0 + (uint32_t)r * blockDim.x;
                if (e < tot) { const uint32_t a = e / K_new, b = e - a * K_new; Inew[a * P + b] = v[r]; }
            }
        }
    } else {
        // remove row/column `rank` (online_inverse.h:275-290)
        const uint32_t nn = K_old;
        const T dd = Iold[rank * P + rank];
        const T sc = -(T(1) / dd);
        for (uint32_t i = threadIdx.x; i < nn; i += blockDim.x) u2[i] = Iold[i * P + rank] * sc;
        __syncthreads();
        const uint32_t tot = K_new * K_new;
        for (uint32_t e0 = threadIdx.x; e0 < tot; e0 += kInvBatch * blockDim.x) {
            T v[kInvBatch];
#pragma unroll
            for (int r = 0; r < kInvBatch; ++r) {
                const uint32_t e = e0 + (uint32_t)r * blockDim.x;
                v[r] = T(0);
                if (e >= tot) continue;
                const uint32_t a = e / K_new, b = e - a * K_new;
                const uint32_t oa = a + (a >= rank ? 1u : 0u), ob = b + (b >= rank ? 1u : 0u);
                v[r] = Iold[oa * P + ob] + (-dd * u2[oa]) * u2[ob];
            }
#pragma unroll
            for (int r = 0; r < kInvBatch; ++r) {
                const uint32_t e = e0 + (uint32_t)r * blockDim.x;
                if (e < tot) { const uint32_t a = e / K_new, b = e - a * K_new; Inew[a * P + b] = v[r]; }
            }
        }
    }

    if (omp) {
        // orthogonal matching pursuit: x_S = (A_S^T A_S)^-1 A_S^T y, written to its columns
        __syncthreads();
        for (uint32_t a = wave; a < K_new; a += NW) {
            T acc = T(0);
            for (uint32_t b = lane; b < K_new; b += 64) acc += Inew[a * P + b] * sgn[b];
            acc = wave_sum(acc);
            if (lane == 0) x[gam_new[a]] = acc;
        }
        __syncthreads();
        if (threadIdx.x == 0) st->cur = cur ^ 1u;
        return;
    }

    // sign(c[Gamma]) with dead zone tol (homotopy-cpu.cpp:259-260).  The correlations after
    // the step are c - gamma*q (c_new = A^T(y - A(x + gamma d)) = c - gamma A^T A d); only
    // their sign is used here, the next sweep recomputes c itself from r.
    for (uint32_t a = threadIdx.x; a < K_new; a += blockDim.x) {
        const uint32_t col = gam_new[a];
        // (cq_agent: c, q were stored by other workgroups of this launch with agent-scope atomic stores)
        const T cc = cq_agent ? __hip_atomic_load(&c[col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : c[col];
        const T qq = cq_agent ? __hip_atomic_load(&q[col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : q[col];
        T cn = cc - g * qq;
        // first-step quirk (homotopy-cpu.cpp:223-227): the seed is sign(|c[idx]|) = +1
        if (first) cn = strict_sign ? cc : (cc < T(0) ? -cc : cc);
        sgn[a] = sign_tol(cn, tol);
    }
    // clear the old direction
    for (uint32_t j = threadIdx.x; j < K_old; j += blockDim.x) d[gam_old[j]] = T(0);
    __syncthreads();
    // direction = inv * sign (homotopy-cpu.cpp:263), scattered to its columns (:266); four rows at a time
    for (uint32_t a0 = wave; a0 < K_new; a0 += kMvRows * NW) {
        T acc[kMvRows];
#pragma unroll
        for (int r = 0; r < kMvRows; ++r) acc[r] = T(0);
        for (uint32_t b0 = lane; b0 < K_new; b0 += 64 * kMvSlices) {
            T sb[kMvSlices], v[kMvRows][kMvSlices];
#pragma unroll
            for (int t = 0; t < kMvSlices; ++t) { const uint32_t b = b0 + 64u * (uint32_t)t; sb[t] = b < K_new ? sgn[b] : T(0); }
#pragma unroll
            for (int r = 0; r < kMvRows; ++r)
#pragma unroll
                for (int t = 0; t < kMvSlices; ++t) {
                    const uint32_t a = a0 + (uint32_t)r * NW, b = b0 + 64u * (uint32_t)t;
                    v[r][t] = (a < K_new && b < K_new) ? Inew[a * P + b] : T(0);
                }
#pragma unroll
            for (int t = 0; t < kMvSlices; ++t)
#pragma unroll
                for (int r = 0; r < kMvRows; ++r)
                    if (b0 + 64u * (uint32_t)t < K_new) acc[r] += v[r][t] * sb[t];
        }
#pragma unroll
        for (int r = 0; r < kMvRows; ++r) {
            const uint32_t a = a0 + (uint32_t)r * NW;
            const T sa = wave_sum(acc[r]);
            if (lane == 0 && a < K_new) d[gam_new[a]] = sa;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) st->cur = cur ^ 1u;
}

// ---- k_gramupd: u1 = A_S^T a_idx and a_idx . a_idx (online_inverse.h:209-218), one
// ---- workgroup per active column; the last to arrive borders / deflates
// ---- (A_S^T A_S)^-1 (online_inverse.h:224-248, 275-290) and forms the new direction
// ---- (homotopy-cpu.cpp:257-267) ---------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kUpdThreads)
void k_gramupd(const T* __restrict__ At, SlotDims L, const uint32_t* __restrict__ gam2,
               T* inv0, T* inv1, T* u1, T* u2, T* sgn,
               const T* __restrict__ c, const T* __restrict__ q, T* __restrict__ d, T tol,
               DevState* st, int omp, const T* __restrict__ y, T* __restrict__ x,
               const T* __restrict__ gcache, const int32_t* __restrict__ slot_of, uint32_t gpitch,
               int first, int strict_sign)
{
    const uint32_t ldm = L.ldm, kcap = L.kcap;
    {   // slot = blockIdx.y
        const size_t s = blockIdx.y;
        if (omp && gcache == nullptr) y += s * ldm;
        if (omp) x += s * L.n_pad;
        gam2 += s * 2 * kcap;
        inv0 += s * 2 * (size_t)kcap * kcap; inv1 += s * 2 * (size_t)kcap * kcap;
        u1 += s * kcap; u2 += s * kcap; sgn += s * kcap;
        c += s * L.n_pad; q += s * L.n_pad; d += s * L.n_pad;
        if (slot_of != nullptr) slot_of += s * L.n_pad;
        st += s;
    }
    if (st->done) return;
    typedef T V4 __attribute__((ext_vector_type(16 / sizeof(T))));
    constexpr int VN = 16 / sizeof(T);
    __shared__ T sv[16];
    __shared__ T s_d;
    __shared__ uint32_t s_flag;

    const uint32_t cur = st->cur;
    const uint32_t K_new = st->K;
    const uint32_t rank = st->rank;
    const bool added = st->added != 0;
    const uint32_t* gam_old = gam2 + (size_t)cur * kcap;
    const uint32_t* gam_new = gam2 + (size_t)(cur ^ 1u) * kcap;

    if (gcache != nullptr) {
        // lookahead engine: one workgroup; u1 and a_idx.a_idx are gathered from the cached
        // Gram column of the entering column
        if (st->status != 0) return;
        if (added) {
            const T* gi = gcache + (size_t)(slot_of != nullptr ? (uint32_t)slot_of[st->idx] : st->idx) * gpitch;
            for (uint32_t b = threadIdx.x; b < K_new; b += blockDim.x) {
                const T v = gi[gam_new[b]];
                if (b == rank) st->dot = (double)v;
                else u1[b - (b > rank ? 1u : 0u)] = v;
                if (omp) sgn[b] = y[gam_new[b]];          // b_S = A_S^T y = c0[S] (y carries c0 here)
            }
        }
        __syncthreads();
    } else if (added && blockIdx.x < K_new) {
        const uint32_t b = blockIdx.x;
        const V4* col = reinterpret_cast<const V4*>(At + (size_t)gam_new[b] * ldm);
        const V4* cnew = reinterpret_cast<const V4*>(At + (size_t)st->idx * ldm);
        const uint32_t nv = ldm / VN;                       // multiple of 64
        T acc = T(0), accy = T(0);
        if (omp) {
            // OMP also needs b_S = A_S^T y for the least-squares solve x_S = inv * b_S
            const V4* yv = reinterpret_cast<const V4*>(y);
#pragma unroll 4
            for (uint32_t i = threadIdx.x; i < nv; i += kUpdThreads) {
                const V4 a = col[i], bnew = cnew[i], yy = yv[i];
#pragma unroll
                for (int e = 0; e < VN; ++e) { acc += a[e] * bnew[e]; accy += a[e] * yy[e]; }
            }
        } else {
#pragma unroll 4
            for (uint32_t i = threadIdx.x; i < nv; i += kUpdThreads) {
                const V4 a = col[i], bnew = cnew[i];
#pragma unroll
                for (int e = 0; e < VN; ++e) acc += a[e] * bnew[e];
            }
        }
        const T v = block_sum(acc, sv);
        T vy = T(0);
        if (omp) vy = block_sum(accy, sv);
        if (threadIdx.x == 0) {
            if (b == rank) st->dot = (double)v;
            else u1[b - (b > rank ? 1u : 0u)] = v;
            if (omp) sgn[b] = vy;                      // b_S in the new sorted order
        }
    }
    if (gcache == nullptr && !arrive_last(&st->ticket_gram, gridDim.x, &s_flag)) return;

    update_direction<T>(cur, K_new, rank, added, gam_old, gam_new, inv0, inv1, u1, u2, sgn, c, q, d, tol, st, omp, x,
                        first, strict_sign, kcap, sv, &s_d, (T)st->gamma);
    // lookahead engine: the entering column's Gram column has arrived and been used
    if (gcache != nullptr && threadIdx.x == 0) st->need_sweep = 0;
}

// ---- k_la_iter: one whole iteration of the lookahead engine in ONE launch --------------------
//   phase 1 (every workgroup)  c = c0 - sum_j x_j g_j, q = sum_j d_j g_j over its columns, max |c|
//   grid barrier               lambda = ||c||_inf needs every workgroup's maximum
//   phase 2 (every workgroup)  find_max_gamma's scan over its columns        (homotopy-cpu.cpp:122-163)
//   last workgroup to arrive   pick, support toggle, x update; then, if the entering column's Gram
//                              column is cached, the inverse update and the new direction.  If it is
//                              not, need_sweep is raised (mirrored to the host, which enqueues
//                              k_la_top + the lookahead sweep + k_gramupd) and later k_la_iter
//                              launches are no-ops until that update has run.
// Every launch, working or not, bumps DevState::seq and mirrors it to hflags[0] so that the host
// can keep a fixed number of launches queued ahead of the device.
constexpr int kItThreads = 256;
constexpr int kItCols = 2;                               // columns per thread and pass
constexpr uint32_t kItChunk = kItThreads * kItCols;
constexpr uint32_t kItMaxBlocks = 256;                   // all resident at once on any gfx950 part

// Grid barrier.  Safe because every workgroup of the launch is resident (<= 256 workgroups of 256
// threads, tiny LDS); arrivals are counted monotonically over the solve (target = barrier rounds so far * grid).
// The spin is bounded: on expiry the caller abandons the solve instead of hanging the queue.
__device__ __forceinline__ bool grid_barrier(uint32_t* counter, uint32_t target, uint32_t* s_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t ok = 0;
        for (uint32_t spin = 0; spin < (1u << 21); ++spin) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0u;
}

// The same barrier without cache fences, for launches whose cross-workgroup data moves with agent-scope
// atomic stores and loads (a release fence writes the whole L2 back, an acquire invalidates it).
__device__ __forceinline__ bool grid_barrier_relaxed(uint32_t* counter, uint32_t target, uint32_t* s_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t ok = 0;
        for (uint32_t spin = 0; spin < (1u << 21); ++spin) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0u;
}

// THREADS x COLS columns per workgroup and pass: 256 x 2 in fp32; 1024 x 1 in fp64, where the serial part of the
// last workgroup (inverse, direction: all in global memory) is a chain of memory round trips and four times
// the threads keep four times the loads in flight
template <typename T, int THREADS, int COLS>
__global__ __launch_bounds__(THREADS)
void k_la_iter(T tol, uint32_t max_iter, uint32_t n,
               const T* __restrict__ gcache, const int32_t* __restrict__ slot_of, const T* __restrict__ c0,
               uint32_t gpitch, T* c, T* q, T* x, T* d, uint8_t* insup,
               T* pmax_val, uint32_t* pmax_idx, T* pmin_val, uint32_t* pmin_idx,
               uint32_t* gam2, uint32_t* touched2, T* inv0, T* inv1, T* u1, T* u2, T* sgn,
               T* __restrict__ tcand, SlotDims L, DevState* st, uint32_t* hflags,
               TraceEntry* trace, uint32_t trace_cap, int zero_on_removal, int tie_guard, uint64_t* dbg,
               unsigned char* slog, uint32_t slog_cap, uint32_t slog_kmax)
{
    // slog (optional, screen.hip's fp64 form): the state every launch STARTS from — lambda and the non-zero coefficients —
    // logged by workgroup 0: cnt[cap] u32, lambda[cap] f64, cols[cap][kmax] u32, vals[cap][kmax] T
    uint64_t ts[8];
    ts[0] = wall_clock64();
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_cnt[2];
    __shared__ uint32_t s_flag;
    __shared__ T s_dd;
    __shared__ uint32_t s_slot[kCqTile];
    __shared__ T s_x[kCqTile];
    __shared__ T s_dv[kCqTile];
    const uint32_t kcap = L.kcap;
    const uint32_t tid = threadIdx.x;

    if (st->done || st->need_sweep) {            // finished, or waiting for a lookahead sweep
        if (blockIdx.x == 0 && tid == 0) bump_seq(st, hflags);
        return;
    }
    const uint32_t round = st->iter + 1u;
    const uint32_t cur0 = st->cur;
    // columns with a non-zero x or d: the support when leaving columns are zeroed exactly
    // (their terms would add exact zeros), every column ever touched otherwise
    const uint32_t nt = zero_on_removal ? st->K : st->ntouched;
    const uint32_t* touched = (zero_on_removal ? gam2 : touched2) + (size_t)cur0 * kcap;

    // ---- phase 1: Gram-form correlations --------------------------------------------------
    T bv = T(-1);
    uint32_t bi = 0xffffffffu;
    for (uint32_t base = blockIdx.x * (uint32_t)(THREADS * COLS); base < n; base += gridDim.x * (uint32_t)(THREADS * COLS)) {
        const T* gbase = gcache + base + tid;            // gpitch % 1024 == 0: rows never run out
        T ax[COLS], ad[COLS];
#pragma unroll
        for (int k = 0; k < COLS; ++k) { ax[k] = T(0); ad[k] = T(0); }
        for (uint32_t j0 = 0; j0 < nt; j0 += kCqTile) {
            const uint32_t cnt = (nt - j0 < kCqTile) ? (nt - j0) : kCqTile;
            __syncthreads();
            if (tid < cnt) {
                const uint32_t col = touched[j0 + tid];
                s_slot[tid] = (uint32_t)slot_of[col];
                s_x[tid] = x[col];
                s_dv[tid] = d[col];
            }
            __syncthreads();
            // whole groups of UNR rows with all their loads in flight; a short last group is padded with
            // zero coefficients on a valid row (exact zeros) — a row at a time would be a memory round trip each
            // (more rows in flight were tried for the narrow dictionary of the fp64 screened form's sub-context, where this phase is two
            // workgroups walking K rows: 16 change nothing — 14 us, two CUs' bandwidth —, 32 spill at 1024 threads: 60 us)
            constexpr int UNR = 8;
            for (uint32_t j = 0; j < cnt; j += UNR) {
                T gv[UNR][COLS];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const T* g = gbase + (size_t)s_slot[j + u < cnt ? j + u : j] * gpitch;
#pragma unroll
                    for (int k = 0; k < COLS; ++k) gv[u][k] = g[k * THREADS];
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const bool live = j + u < cnt;
                    const T xj = live ? s_x[j + u] : T(0), dj = live ? s_dv[j + u] : T(0);
#pragma unroll
                    for (int k = 0; k < COLS; ++k) { ax[k] += xj * gv[u][k]; ad[k] += dj * gv[u][k]; }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < COLS; ++k) {
            const uint32_t i = base + k * THREADS + tid;
            if (i < n) {
                const T cv = c0[i] - ax[k];
                // (everything that crosses workgroups in this launch moves with L2-bypassing stores and
                // loads: no cache fences at the barrier or at the ticket)
                __hip_atomic_store(&c[i], cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&q[i], ad[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const T a = cv < T(0) ? -cv : cv;
                if (better_max(a, i, bv, bi)) { bv = a; bi = i; }
            }
        }
    }
    block_reduce_pair<T, true>(bv, bi, sv, si);
    if (tid == 0) {
        __hip_atomic_store(&pmax_val[blockIdx.x], bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pmax_idx[blockIdx.x], bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ts[1] = wall_clock64();

    const uint32_t bar_round = st->bar_rounds + 1u;
    if (!grid_barrier_relaxed(&st->bar_count, bar_round * gridDim.x, &s_flag)) {
        if (blockIdx.x == 0 && tid == 0) {               // cannot happen with a resident grid
            st->status = SS_HIP_ERUNTIME;
            st->done = 1;
            signal_done(hflags, nullptr, 1u, round);
        }
        return;
    }

    ts[2] = wall_clock64();
    // lambda = ||c||_inf (homotopy-cpu.cpp:270 / :219); same exact value in every workgroup
    T c_inf;
    uint32_t imax;
    reduce_partials_agent(pmax_val, pmax_idx, gridDim.x, c_inf, imax, sv, si);

    if (slog != nullptr && blockIdx.x == 0) {
        const uint32_t t = round - 1u;
        if (t < slog_cap) {
            uint32_t* l_cnt = reinterpret_cast<uint32_t*>(slog);
            double* l_lam = reinterpret_cast<double*>(slog + (((size_t)slog_cap * 4 + 7) & ~(size_t)7));
            double* l_exp = l_lam + slog_cap;                  // where the previous step left lambda: lambda_prev - gamma_prev
            uint32_t* l_cols = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(l_exp) + (size_t)slog_cap * 8);
            T* l_vals = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(l_cols) + ((((size_t)slog_cap * slog_kmax * 4) + 7) & ~(size_t)7));
            const bool fits = nt <= slog_kmax;
            if (fits)
                for (uint32_t j = tid; j < nt; j += blockDim.x) {
                    const uint32_t col = touched[j];
                    l_cols[(size_t)t * slog_kmax + j] = col;
                    l_vals[(size_t)t * slog_kmax + j] = x[col];
                }
            if (tid == 0) { l_cnt[t] = fits ? nt : 0xffffffffu; l_lam[t] = (double)c_inf; l_exp[t] = st->c_inf - st->gamma; }
        }
    }

    // do { ... } while (iter < max_iter && c_inf > tolerance)   (homotopy-cpu.cpp:236,272)
    if ((round > 1 && !(c_inf > tol)) || round > max_iter) {
        if (blockIdx.x == 0 && tid == 0) {
            st->c_inf = (double)c_inf;
            st->iter = round - 1;
            st->done_round = round;
            st->need_sweep = 0;
            st->done = 1;
            st->bar_rounds = bar_round;
            signal_done(hflags, nullptr, 1u, round);
            bump_seq(st, hflags);
        }
        return;
    }

    // ---- phase 2: step-length scan (same expressions as k_scansel) ---------------------------
    T best = Lim<T>::max();
    uint32_t best_i = 0xffffffffu;
    const uint32_t just_removed = (st->iter >= 1u && st->added == 0u) ? st->idx : 0xffffffffu;   // (see k_scansel)
    const bool in_band = tie_band<T>(c_inf, (T)st->c_inf, (T)st->gamma, (T)st->lambda0);
    for (uint32_t base = blockIdx.x * (uint32_t)(THREADS * COLS); base < n; base += gridDim.x * (uint32_t)(THREADS * COLS))
#pragma unroll
    for (int k = 0; k < COLS; ++k) {
        const uint32_t i = base + k * THREADS + tid;
        if (i < n) {
            T m = Lim<T>::max();
            const bool act = insup[i] != 0;
            if (act) {
                const T t = -x[i] / d[i];
                if (t > T(0) && t < m) m = t;
            } else {
                const T qi = __hip_atomic_load(&q[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const T ci = __hip_atomic_load(&c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const T dl = T(1) - qi, dr = T(1) + qi;
                if (dl != T(0)) {
                    T t = (c_inf - ci) / dl;
                    if (tie_guard && t == T(0) && dl > T(0)) t = Lim<T>::tiny();
                    if (t == T(0) && i != just_removed && in_band) __hip_atomic_store(&st->tie_stall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (see DevState::tie_stall)
                    if (t > T(0) && t < m) m = t;
                }
                if (dr != T(0)) {
                    T t = (c_inf + ci) / dr;
                    if (tie_guard && t == T(0) && dr > T(0)) t = Lim<T>::tiny();
                    if (t == T(0) && i != just_removed && in_band) __hip_atomic_store(&st->tie_stall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t > T(0) && t < m) m = t;
                }
            }
            tcand[i] = (act || slot_of[i] >= 0) ? Lim<T>::max() : m;
            if (better_min(m, i, best, best_i)) { best = m; best_i = i; }
        }
    }
    block_reduce_pair<T, false>(best, best_i, sv, si);
    if (tid == 0) {
        __hip_atomic_store(&pmin_val[blockIdx.x], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pmin_idx[blockIdx.x], best_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ts[3] = wall_clock64();
    if (!arrive_last_relaxed(&st->ticket_scan, gridDim.x, &s_flag)) return;
    ts[4] = wall_clock64();

    // ---- last workgroup -----------------------------------------------------------------------
    if (tid == 0) st->bar_rounds = bar_round;
    uint32_t idx = 0, rank = 0, added = 0, K_new = 0;
    T g = T(0);
    const bool go = select_toggle<T>(round, c_inf, gridDim.x, pmin_val, pmin_idx, x, d, insup, gam2, touched2, kcap,
                                     st, hflags, false, trace, trace_cap, zero_on_removal, nullptr, 1u, slot_of,
                                     sv, si, s_cnt, &idx, &rank, &added, &K_new, &g);
    if (!go) {
        if (tid == 0) bump_seq(st, hflags);
        return;
    }
    ts[5] = wall_clock64();
    const uint32_t* gam_old = gam2 + (size_t)cur0 * kcap;
    const uint32_t* gam_new = gam2 + (size_t)(cur0 ^ 1u) * kcap;
    if (added) {
        const int32_t slot = slot_of[idx];
        if (slot < 0) {
            // not cached: select_toggle raised need_sweep; tell the host and wait for the sweep
            if (tid == 0) {
                const uint32_t nm = st->nmiss + 1u;
                st->nmiss = nm;
                __hip_atomic_store(&hflags[2], nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                bump_seq(st, hflags);
            }
            return;
        }
        // u1 = A_S^T a_idx and a_idx.a_idx gathered from the cached Gram column (online_inverse.h:209-218)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                  // gam_new written by select_toggle
        const T* gi = gcache + (size_t)slot * gpitch;
        for (uint32_t b = tid; b < K_new; b += blockDim.x) {
            const T v = gi[gam_new[b]];
            if (b == rank) st->dot = (double)v;
            else u1[b - (b > rank ? 1u : 0u)] = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    update_direction<T>(cur0, K_new, rank, added != 0, gam_old, gam_new, inv0, inv1, u1, u2, sgn, c, q, d, tol, st, 0,
                        x, 0, 0, kcap, sv, &s_dd, g, true);
    if (tid == 0) bump_seq(st, hflags);
    if (dbg != nullptr && tid == 0 && round < 1024u) {     // stage timestamps of the last workgroup (100 MHz)
        ts[6] = wall_clock64();
        ts[7] = blockIdx.x;
        for (int k2 = 0; k2 < 8; ++k2) dbg[(size_t)round * 8 + k2] = ts[k2];
    }
}

// ---- k_la_omp: one iteration of orthogonal matching pursuit in Gram form ------------------------------
// c = A^T(y - A_S x_S) = c0 - sum_j x_j g_j from the cached Gram columns (every workgroup, its columns),
// grid barrier for ||c||_inf and its first index, then workgroup 0 alone: loop control and pick (as
// k_omp_select), and — if the Gram column of the pick is cached — the bordered inverse and the
// least-squares coefficients x_S = (A_S^T A_S)^-1 A_S^T y with A_S^T y = c0[S].  A pick without cached
// Gram column raises need_sweep: the host runs k_la_top (ranking by |c|: the next picks are the
// largest correlations) + the 32-column sweep + k_gramupd and launches again.
template <typename T>
__global__ __launch_bounds__(kItThreads)
void k_la_omp(T tol, uint32_t max_iter, uint32_t n, T gram_guard,
              const T* __restrict__ gcache, const int32_t* __restrict__ slot_of, const T* __restrict__ c0,
              uint32_t gpitch, T* c, T* x, uint8_t* insup, T* pmax_val, uint32_t* pmax_idx,
              uint32_t* gam2, uint32_t* touched2, T* inv0, T* inv1, T* u1, T* u2, T* sgn, T* q_unused, T* d_unused,
              SlotDims L, DevState* st, uint32_t* hflags, TraceEntry* trace, uint32_t trace_cap,
              unsigned char* slog, uint32_t slog_cap, uint32_t slog_kmax)
{
    // (slog: the state every launch starts from, as in k_la_iter — the fp64 screened form's sub-context)
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_flag;
    __shared__ uint32_t s_cnt;
    __shared__ T s_dd;
    __shared__ uint32_t s_slot[kCqTile];
    __shared__ T s_x[kCqTile];
    const uint32_t kcap = L.kcap;
    const uint32_t tid = threadIdx.x;

    if (st->done || st->need_sweep) {
        if (blockIdx.x == 0 && tid == 0) bump_seq(st, hflags);
        return;
    }
    const uint32_t round = st->iter + 1u;
    const uint32_t cur0 = st->cur;
    const uint32_t K = st->K;
    const uint32_t* gam = gam2 + (size_t)cur0 * kcap;

    // ---- c in Gram form over this workgroup's columns ---------------------------------------------------
    T bv = T(-1);
    uint32_t bi = 0xffffffffu;
    for (uint32_t base = blockIdx.x * kItChunk; base < n; base += gridDim.x * kItChunk) {
        const T* gbase = gcache + base + tid;
        T ax[kItCols];
#pragma unroll
        for (int k = 0; k < kItCols; ++k) ax[k] = T(0);
        for (uint32_t j0 = 0; j0 < K; j0 += kCqTile) {
            const uint32_t cnt = (K - j0 < kCqTile) ? (K - j0) : kCqTile;
            __syncthreads();
            if (tid < cnt) {
                const uint32_t cl = gam[j0 + tid];
                s_slot[tid] = (uint32_t)slot_of[cl];
                s_x[tid] = x[cl];
            }
            __syncthreads();
            for (uint32_t j = 0; j < cnt; ++j) {
                const T* g = gbase + (size_t)s_slot[j] * gpitch;
                const T xj = s_x[j];
#pragma unroll
                for (int k = 0; k < kItCols; ++k) ax[k] += xj * g[k * kItThreads];
            }
        }
#pragma unroll
        for (int k = 0; k < kItCols; ++k) {
            const uint32_t i = base + k * kItThreads + tid;
            if (i < n) {
                const T cv = c0[i] - ax[k];
                c[i] = cv;
                const T a = cv < T(0) ? -cv : cv;
                if (better_max(a, i, bv, bi)) { bv = a; bi = i; }
            }
        }
    }
    block_reduce_pair<T, true>(bv, bi, sv, si);
    // (the partial maxima are all that crosses workgroups here: L2-bypassing stores, no fences)
    if (tid == 0) {
        __hip_atomic_store(&pmax_val[blockIdx.x], bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pmax_idx[blockIdx.x], bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t bar_round = st->bar_rounds + 1u;
    if (!grid_barrier_relaxed(&st->bar_count, bar_round * gridDim.x, &s_flag)) {
        if (blockIdx.x == 0 && tid == 0) {
            st->status = SS_HIP_ERUNTIME;
            st->done = 1;
            signal_done(hflags, nullptr, 1u, round);
        }
        return;
    }
    if (blockIdx.x != 0) return;                       // the serial part belongs to workgroup 0
    if (tid == 0) st->bar_rounds = bar_round;

    T c_inf;
    uint32_t idx;
    reduce_partials_agent(pmax_val, pmax_idx, gridDim.x, c_inf, idx, sv, si);
    if (slog != nullptr) {
        const uint32_t t = round - 1u;
        if (t < slog_cap) {
            uint32_t* l_cnt = reinterpret_cast<uint32_t*>(slog);
            double* l_lam = reinterpret_cast<double*>(slog + (((size_t)slog_cap * 4 + 7) & ~(size_t)7));
            double* l_exp = l_lam + slog_cap;                  // (OMP takes no step lengths: the pick's |c| itself is what counts)
            uint32_t* l_cols = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(l_exp) + (size_t)slog_cap * 8);
            T* l_vals = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(l_cols) + ((((size_t)slog_cap * slog_kmax * 4) + 7) & ~(size_t)7));
            const bool fits = K <= slog_kmax;
            if (fits)
                for (uint32_t j = tid; j < K; j += blockDim.x) {
                    const uint32_t col = gam[j];
                    l_cols[(size_t)t * slog_kmax + j] = col;
                    l_vals[(size_t)t * slog_kmax + j] = x[col];
                }
            if (tid == 0) { l_cnt[t] = fits ? K : 0xffffffffu; l_lam[t] = (double)c_inf; l_exp[t] = (double)c_inf; }
        }
    }
    if (round == 1u && gram_guard > T(0) && tol < gram_guard * c_inf) {
        // tolerance too tight for Gram-form correlations: the host re-runs in residual form
        if (tid == 0) {
            st->status = kStatusRetryResidual;
            st->done = 1;
            signal_done(hflags, nullptr, 1u, 0u);
            bump_seq(st, hflags);
        }
        return;
    }
    // stop: tolerance reached, iteration budget spent, the support is full, or the best column is
    // already active (numerical stall) — k_omp_select's rules
    const bool stall = insup[idx] != 0;
    if (!(c_inf > tol) || round > max_iter || K >= kcap || stall) {
        if (tid == 0) {
            st->c_inf = (double)c_inf;
            st->iter = round - 1;
            st->done_round = round;
            if (K >= kcap && c_inf > tol && round <= max_iter && !stall) st->status = SS_HIP_ECAPACITY;
            st->done = 1;
            signal_done(hflags, nullptr, 1u, round);
            bump_seq(st, hflags);
        }
        return;
    }
    uint32_t* gam_new = gam2 + (size_t)(cur0 ^ 1u) * kcap;
    uint32_t* tch_new = touched2 + (size_t)(cur0 ^ 1u) * kcap;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    uint32_t lr = 0;
    for (uint32_t j = tid; j < K; j += blockDim.x) lr += (gam[j] < idx) ? 1u : 0u;
    if (lr) atomicAdd(&s_cnt, lr);
    __syncthreads();
    const uint32_t rank = s_cnt;
    const uint32_t K_new = K + 1;
    for (uint32_t j = tid; j < K_new; j += blockDim.x) {
        const uint32_t v = (j < rank) ? gam[j] : (j == rank ? idx : gam[j - 1]);
        gam_new[j] = v;
        tch_new[j] = v;
    }
    const int32_t slot = slot_of[idx];
    if (tid == 0) {
        insup[idx] = 1;
        st->K = K_new;
        st->ntouched = K_new;
        st->idx = idx;
        st->rank = rank;
        st->added = 1;
        st->gamma = 0.0;
        st->c_inf = (double)c_inf;
        st->iter = round;
        if (trace != nullptr && round < trace_cap) {
            trace[round].idx = idx;
            trace[round].added = 1;
            trace[round].gamma = 0.0;
            trace[round].c_inf = (double)c_inf;
        }
        if (slot < 0) {
            st->need_sweep = 1;
            const uint32_t nm = st->nmiss + 1u;
            st->nmiss = nm;
            __hip_atomic_store(&hflags[2], nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            bump_seq(st, hflags);
        }
    }
    if (slot < 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const T* gi = gcache + (size_t)slot * gpitch;
    for (uint32_t b = tid; b < K_new; b += blockDim.x) {
        const T v = gi[gam_new[b]];
        if (b == rank) st->dot = (double)v;
        else u1[b - (b > rank ? 1u : 0u)] = v;
        sgn[b] = c0[gam_new[b]];                       // b_S = A_S^T y
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    update_direction<T>(cur0, K_new, rank, true, gam, gam_new, inv0, inv1, u1, u2, sgn, c, q_unused, d_unused, tol, st, 1,
                        x, 0, 0, kcap, sv, &s_dd, T(0));
    if (tid == 0) bump_seq(st, hflags);
}

// ---- y = A x (reconstruct_signal, lib.cpp:78-104) -------------------------------------
template <typename T>
__global__ __launch_bounds__(kSmallThreads)
void k_gemv_n(const T* __restrict__ At, uint32_t ldm, uint32_t m, uint32_t n,
              const T* __restrict__ x, T* __restrict__ y)
{
    // one thread per row; columns with x[j] == 0 are skipped (exact: they add 0)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    T acc = T(0);
    for (uint32_t j = 0; j < n; ++j) {
        const T xj = x[j];
        if (xj != T(0)) acc += At[(size_t)j * ldm + i] * xj;
    }
    y[i] = acc;
}

// ---- k_absmax: per-slot partial (max |c|, first index) of the GEMM's correlation rows -------
constexpr uint32_t kAbsmaxChunk = 1024;

template <typename T>
__global__ __launch_bounds__(kSmallThreads)
void k_absmax(const T* __restrict__ c, uint32_t n, SlotDims L, T* __restrict__ pmax_val,
              uint32_t* __restrict__ pmax_idx, const DevState* st)
{
    const size_t s = blockIdx.y;
    if (st[s].done) return;
    c += s * L.n_pad;
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    T v = T(-1);
    uint32_t ix = 0xffffffffu;
    const uint32_t base = blockIdx.x * kAbsmaxChunk;
#pragma unroll
    for (uint32_t k = 0; k < kAbsmaxChunk / kSmallThreads; ++k) {
        const uint32_t i = base + k * kSmallThreads + threadIdx.x;
        if (i < n) {
            const T a = c[i] < T(0) ? -c[i] : c[i];
            if (better_max(a, i, v, ix)) { v = a; ix = i; }
        }
    }
    block_reduce_pair<T, true>(v, ix, sv, si);
    if (threadIdx.x == 0) {
        pmax_val[s * L.pmax_stride + blockIdx.x] = v;
        pmax_idx[s * L.pmax_stride + blockIdx.x] = ix;
    }
}

// ---- k_tile_list: compact list of the 128-row GEMM tiles that still hold a running signal --
// list[0..count) = active tile indices, list[mtiles] = count.  The GEMM maps its leading
// count*ntiles workgroups onto these tiles, so the live work is contiguous in blockIdx and
// spreads over all XCDs / CUs (a strided subset of blockIdx lands on a few CUs only).
__global__ __launch_bounds__(128)
void k_tile_list(const DevState* __restrict__ st, uint32_t nslots, uint32_t mtiles, uint32_t* __restrict__ list)
{
    __shared__ uint32_t s_count;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    for (uint32_t t = 0; t < mtiles; ++t) {
        const uint32_t slot = t * 128u + threadIdx.x;
        const int running = (slot < nslots) && (st[slot].done == 0);
        const int any = __syncthreads_or(running);
        if (threadIdx.x == 0 && any) list[s_count++] = t;
    }
    if (threadIdx.x == 0) list[mtiles] = s_count;
}

hipError_t launch_tile_list(const ss_hip_ctx* ctx, const DevState* st, uint32_t nslots, uint32_t rows,
                            uint32_t* list)
{
    hipLaunchKernelGGL(k_tile_list, dim3(1), dim3(128), 0, ctx->stream, st, nslots, rows / 128u, list);
    return hipGetLastError();
}

// ---- launchers ------------------------------------------------------------------------

template <typename T>
hipError_t launch_init(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t nparts, T tol)
{
    hipLaunchKernelGGL((k_init<T>), dim3(1, nslots), dim3(kUpdThreads), 0, ctx->stream,
                       static_cast<const T*>(ctx->At), ws.dims, ws.c, ws.pmax_val, ws.pmax_idx,
                       nparts, ws.d, ws.insup, ws.gam, ws.touched, ws.inv[0], tol,
                       ctx->strict_sign, ws.st, ws.trace);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_rp(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots)
{
    const uint32_t blocks = ctx->ldm / kRpRows;
    hipLaunchKernelGGL((k_rp<T>), dim3(blocks, nslots), dim3(kRpThreads), 0, ctx->stream,
                       static_cast<const T*>(ctx->At), ws.dims, ws.y, ws.x, ws.d, ws.touched,
                       ws.rhs, ws.st);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_iteration_tail(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round,
                                 uint32_t nparts, T tol, uint32_t max_iter)
{
    const uint32_t n = (uint32_t)ctx->n;
    const uint32_t per_block = kSmallThreads * kScanPerThread;
    uint32_t ns = (n + per_block - 1) / per_block;
    if (ns > ws.dims.pmin_stride) ns = ws.dims.pmin_stride;   // k_scansel grid-strides
    hipLaunchKernelGGL((k_scansel<T>), dim3(ns, nslots), dim3(kSmallThreads), 0, ctx->stream, round, tol,
                       max_iter, n, ws.c, ws.q, ws.x, ws.d, ws.insup, ws.pmax_val, ws.pmax_idx,
                       nparts, ws.pmin_val, ws.pmin_idx, ws.gam, ws.touched, ws.dims, ws.st,
                       ctx->dev_flags, ws.trace, ws.trace_cap, ctx->zero_on_removal, ctx->tie_guard, ws.ndone, nslots,
                       (T*)nullptr, (const int32_t*)nullptr, (ctx->tie_rerun && !ctx->tie_guard) ? 1 : 0);
    uint32_t gb = round + 1;
    if (gb > ws.kcap) gb = ws.kcap;
    hipLaunchKernelGGL((k_gramupd<T>), dim3(gb, nslots), dim3(kUpdThreads), 0, ctx->stream,
                       static_cast<const T*>(ctx->At), ws.dims, ws.gam, ws.inv[0], ws.inv[1],
                       ws.u1, ws.u2, ws.sgn, ws.c, ws.q, ws.d, tol, ws.st, 0, (const T*)nullptr, (T*)nullptr,
                       (const T*)nullptr, (const int32_t*)nullptr, 0u, 0, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_rp(ctx, ws, nslots);
}

// one OMP round after the sweep c = A^T r: pick, bordered inverse + least squares, residual
template <typename T>
hipError_t launch_omp_tail(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round,
                           uint32_t nparts, T tol, uint32_t max_iter)
{
    hipLaunchKernelGGL((k_omp_select<T>), dim3(1, nslots), dim3(kSmallThreads), 0, ctx->stream, round, tol,
                       max_iter, ws.pmax_val, ws.pmax_idx, nparts, ws.insup, ws.gam, ws.touched, ws.dims,
                       ws.st, ctx->dev_flags, ws.trace, ws.trace_cap, ws.ndone, nslots);
    uint32_t gb = round;                 // support size after this round's insert is <= round
    if (gb > ws.kcap) gb = ws.kcap;
    hipLaunchKernelGGL((k_gramupd<T>), dim3(gb, nslots), dim3(kUpdThreads), 0, ctx->stream,
                       static_cast<const T*>(ctx->At), ws.dims, ws.gam, ws.inv[0], ws.inv[1],
                       ws.u1, ws.u2, ws.sgn, ws.c, ws.q, ws.d, tol, ws.st, 1, (const T*)ws.y, ws.x,
                       (const T*)nullptr, (const int32_t*)nullptr, 0u, 0, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_rp(ctx, ws, nslots);
}

// ---- lookahead engine launchers -------------------------------------------------------------
template <typename T>
hipError_t launch_la_init_pick(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nparts, T tol, bool full_gram)
{
    // option "engine" = 1: guard on (tolerance vs ||A^T y||_inf), 2: lookahead engine unconditionally
    const T guard = ctx->engine == 1 ? (T)(sizeof(T) == 4 ? kGramGuard : kGramGuard64) : T(0);
    hipLaunchKernelGGL((k_la_init_pick<T>), dim3(1), dim3(kSmallThreads), 0, ctx->stream, ws.pmax_val,
                       ws.pmax_idx, nparts, ws.insup, ws.gam, ws.touched, ws.st, ws.trace, tol, guard, ctx->dev_flags,
                       full_gram ? ctx->n_pad : 0u);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_la_top(const ss_hip_ctx* ctx, Workspace<T>& ws, int init_mode, uint32_t nsel)
{
    // init_mode 1: rank by |c0| (first batch of a Homotopy solve); 2: by the current |c| (OMP)
    hipLaunchKernelGGL((k_la_top<T>), dim3(1), dim3(kUpdThreads), 0, ctx->stream, ws.tcand, init_mode == 2 ? ws.c : ws.c0,
                       (uint32_t)ctx->n, init_mode, ws.insup, ws.slot_of, ws.gcap, ws.sw_list, ws.st, ctx->dev_flags,
                       ws.gram_is_full ? (uint32_t*)nullptr : ws.slot_col, nsel > (uint32_t)kSwStride ? (uint32_t)kSwStride : nsel);
    return hipGetLastError();
}

// ---- k_la_reset: everything a lookahead solve clears before its first sweep, in ONE launch (ten small
// ---- memsets / copies on the stream cost ~4 us each): x, d, membership flags, the slot map (-1), the
// ---- exchange area of the resident kernel, DevState, and r = y for the A^T y sweep
template <typename T>
__global__ __launch_bounds__(kSmallThreads)
void k_la_reset(T* __restrict__ x, T* __restrict__ d, uint8_t* __restrict__ insup, int32_t* __restrict__ slot_of,
                uint32_t n_pad, uint32_t* __restrict__ la_sync, uint32_t sync_head_words, uint32_t sync_words,
                uint32_t* __restrict__ st_words, uint32_t st_nwords, uint32_t* __restrict__ ndone,
                T* __restrict__ y, T* __restrict__ rhs, uint32_t ldm, const T* __restrict__ y_user, long long incy, uint32_t m)
{
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (uint32_t i = gtid; i < n_pad; i += gsz) {
        x[i] = T(0); d[i] = T(0); insup[i] = 0;
        if (slot_of != nullptr) slot_of[i] = -1;
    }
    if (y_user != nullptr) {
        // the caller's signal is on the device: taken from there (no copy command in front of this launch)
        for (uint32_t i = gtid; i < ldm; i += gsz) { const T v = i < m ? y_user[(long long)i * incy] : T(0); y[i] = v; rhs[i] = v; }
    } else {
        for (uint32_t i = gtid; i < ldm; i += gsz) rhs[i] = y[i];
    }
    for (uint32_t i = gtid; i < sync_words; i += gsz) la_sync[i] = i < sync_head_words ? 0u : 0xffffffffu;
    for (uint32_t i = gtid; i < st_nwords; i += gsz) st_words[i] = 0u;
    if (gtid == 0) *ndone = 0u;
}

template <typename T>
hipError_t launch_la_reset(const ss_hip_ctx* ctx, Workspace<T>& ws, bool clear_slots, const T* y_user, ptrdiff_t incy)
{
    const uint32_t grid = std::min<uint32_t>((ctx->n_pad + kSmallThreads - 1) / kSmallThreads, 256u);
    hipLaunchKernelGGL((k_la_reset<T>), dim3(grid), dim3(kSmallThreads), 0, ctx->stream, ws.x, ws.d, ws.insup,
                       clear_slots ? ws.slot_of : (int32_t*)nullptr, ctx->n_pad, reinterpret_cast<uint32_t*>(ws.la_sync),
                       (uint32_t)(sizeof(LaSync) / 4), (uint32_t)(kLaSyncBytes / 4), reinterpret_cast<uint32_t*>(ws.st),
                       (uint32_t)(sizeof(DevState) / 4), ws.ndone, ws.y, ws.rhs, ctx->ldm, y_user, (long long)incy, (uint32_t)ctx->m);
    return hipGetLastError();
}
template hipError_t launch_la_reset<float>(const ss_hip_ctx*, Workspace<float>&, bool, const float*, ptrdiff_t);
template hipError_t launch_la_reset<double>(const ss_hip_ctx*, Workspace<double>&, bool, const double*, ptrdiff_t);

template <typename T>
hipError_t launch_la_update(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t round, T tol)
{
    // (round 0: the first sign is taken from c0 = A^T y itself)
    hipLaunchKernelGGL((k_gramupd<T>), dim3(1, 1), dim3(kUpdThreads), 0, ctx->stream,
                       static_cast<const T*>(ctx->At), ws.dims, ws.gam, ws.inv[0], ws.inv[1],
                       ws.u1, ws.u2, ws.sgn, round == 0 ? (const T*)ws.c0 : (const T*)ws.c, ws.q, ws.d, tol, ws.st, 0, (const T*)nullptr, (T*)nullptr,
                       (const T*)ws.gcache, (const int32_t*)ws.slot_of, ws.gpitch, round == 0 ? 1 : 0, ctx->strict_sign);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_la_cq(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t* nparts_out)
{
    const uint32_t n = (uint32_t)ctx->n;
    const uint32_t nb = (n + kCqChunk - 1) / kCqChunk;
    if (nb > ws.dims.pmax_stride) return hipErrorInvalidValue;
    if (nparts_out) *nparts_out = nb;
    hipLaunchKernelGGL((k_la_cq<T, false>), dim3(nb), dim3(kSmallThreads), 0, ctx->stream, ws.gcache, ws.slot_of,
                       ws.c0, ws.x, ws.d, ws.touched, n, ws.gpitch, ws.dims, ws.c, ws.q, ws.pmax_val, ws.pmax_idx, ws.st);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_la_iter(const ss_hip_ctx* ctx, Workspace<T>& ws, T tol, uint32_t max_iter)
{
    const uint32_t n = (uint32_t)ctx->n;
    constexpr int TH = sizeof(T) == 8 ? 1024 : kItThreads, CL = sizeof(T) == 8 ? 1 : kItCols;
    constexpr uint32_t chunk = (uint32_t)(TH * CL);
    uint32_t nb = (n + chunk - 1) / chunk;
    const uint32_t cap = std::min<uint32_t>(std::min<uint32_t>(kItMaxBlocks, (uint32_t)ctx->num_cus),
                                            std::min<uint32_t>(ws.dims.pmax_stride, ws.dims.pmin_stride));
    if (nb > cap) nb = cap;                              // the kernel grid-strides; the grid must be resident
    if (nb == 0) nb = 1;
    hipLaunchKernelGGL((k_la_iter<T, TH, CL>), dim3(nb), dim3(TH), 0, ctx->stream, tol, max_iter, n,
                       (const T*)ws.gcache, (const int32_t*)ws.slot_of, (const T*)ws.c0, ws.gpitch,
                       ws.c, ws.q, ws.x, ws.d, ws.insup, ws.pmax_val, ws.pmax_idx, ws.pmin_val, ws.pmin_idx,
                       ws.gam, ws.touched, ws.inv[0], ws.inv[1], ws.u1, ws.u2, ws.sgn, ws.tcand, ws.dims, ws.st,
                       ctx->dev_flags, ws.trace, ws.trace_cap, ctx->zero_on_removal, ctx->tie_guard, ws.la_dbg,
                       static_cast<unsigned char*>(ctx->slog), ctx->slog_cap, ctx->slog_kmax);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_la_omp(const ss_hip_ctx* ctx, Workspace<T>& ws, T tol, uint32_t max_iter)
{
    const uint32_t n = (uint32_t)ctx->n;
    uint32_t nb = (n + kItChunk - 1) / kItChunk;
    const uint32_t cap = std::min<uint32_t>(std::min<uint32_t>(kItMaxBlocks, (uint32_t)ctx->num_cus), ws.dims.pmax_stride);
    if (nb > cap) nb = cap;
    if (nb == 0) nb = 1;
    const T guard = ctx->engine == 1 ? (T)(sizeof(T) == 4 ? kGramGuard : kGramGuard64) : T(0);
    hipLaunchKernelGGL((k_la_omp<T>), dim3(nb), dim3(kItThreads), 0, ctx->stream, tol, max_iter, n, guard,
                       (const T*)ws.gcache, (const int32_t*)ws.slot_of, (const T*)ws.c0, ws.gpitch,
                       ws.c, ws.x, ws.insup, ws.pmax_val, ws.pmax_idx, ws.gam, ws.touched, ws.inv[0], ws.inv[1],
                       ws.u1, ws.u2, ws.sgn, ws.q, ws.d, ws.dims, ws.st, ctx->dev_flags, ws.trace, ws.trace_cap,
                       static_cast<unsigned char*>(ctx->slog), ctx->slog_cap, ctx->slog_kmax);
    return hipGetLastError();
}

// the inverse update k_la_omp left pending (its pick was not cached): k_gramupd, cached + OMP mode
template <typename T>
hipError_t launch_la_omp_update(const ss_hip_ctx* ctx, Workspace<T>& ws, T tol)
{
    hipLaunchKernelGGL((k_gramupd<T>), dim3(1, 1), dim3(kUpdThreads), 0, ctx->stream,
                       static_cast<const T*>(ctx->At), ws.dims, ws.gam, ws.inv[0], ws.inv[1],
                       ws.u1, ws.u2, ws.sgn, ws.c, ws.q, ws.d, tol, ws.st, 1, (const T*)ws.c0, ws.x,
                       (const T*)ws.gcache, (const int32_t*)ws.slot_of, ws.gpitch, 0, 0);
    return hipGetLastError();
}

// batched Gram form: is the tolerance too tight for Gram-form correlations in any slot? (k_la_init_pick's
// rule, against the ||A^T y||_inf k_init left in DevState::c_inf) -> hflags[4] = 1
template <typename T>
__global__ __launch_bounds__(kSmallThreads)
void k_gram_guard_batched(const DevState* __restrict__ st, uint32_t nslots, T tol, T guard, uint32_t* hflags)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nslots && tol < guard * (T)st[s].c_inf)
        __hip_atomic_store(&hflags[4], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T>
hipError_t launch_gram_guard_batched(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, T tol)
{
    const T guard = (T)(sizeof(T) == 4 ? kGramGuard : kGramGuard64);
    hipLaunchKernelGGL((k_gram_guard_batched<T>), dim3((nslots + kSmallThreads - 1) / kSmallThreads), dim3(kSmallThreads), 0,
                       ctx->stream, (const DevState*)ws.st, nslots, tol, guard, ctx->dev_flags);
    return hipGetLastError();
}

// ---- batched Gram form: every signal slot against the full Gram matrix G = A^T A ------------------------
// ---- column form of a mid-size batch: the entering columns of one round, compacted into the pass's lists ---------
// One workgroup.  Slot b that is still running and has just INSERTED column idx gets a row of the batch's column
// cache: row = row_base + b; rcols / drows are filled from entry 0 (the pass kernels skip a list whose first entry
// is 0xffffffff, so groups of 64 beyond the live picks cost microseconds), and the slot's row table learns the row.
constexpr uint32_t kBatchColsThreads = 512;
__global__ __launch_bounds__(kBatchColsThreads)
void k_batch_cols(const DevState* __restrict__ st, uint32_t nslots, uint32_t row_base, int first, uint32_t n_pad,
                  int32_t* __restrict__ bslot, uint32_t* __restrict__ rcols, uint32_t* __restrict__ drows, uint32_t cap)
{
    __shared__ uint32_t s_wave[kBatchColsThreads / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    bool live = false;
    uint32_t idx = 0;
    if (tid < nslots) {
        const DevState& s = st[tid];
        live = s.done == 0 && s.status == 0 && (first || s.added != 0);
        idx = s.idx;
    }
    const uint64_t bal = __ballot(live);
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t off = 0, total = 0;
    for (uint32_t w = 0; w < kBatchColsThreads / 64; ++w) { if (w < wave) off += s_wave[w]; total += s_wave[w]; }
    if (live) {
        const uint32_t pos = off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        rcols[pos] = idx;
        drows[pos] = row_base + tid;
        bslot[(size_t)tid * n_pad + idx] = (int32_t)(row_base + tid);
    }
    for (uint32_t i = total + tid; i < cap; i += kBatchColsThreads) { rcols[i] = 0xffffffffu; drows[i] = 0xffffffffu; }
}

hipError_t launch_batch_cols(const ss_hip_ctx* ctx, const DevState* st, uint32_t nslots, uint32_t row_base, bool first,
                             int32_t* bslot, uint32_t* rcols, uint32_t* drows, uint32_t cap)
{
    if (nslots > kBatchColsThreads || cap < nslots) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_batch_cols, dim3(1), dim3(kBatchColsThreads), 0, ctx->stream, st, nslots, row_base, first ? 1 : 0,
                       (uint32_t)ctx->n_pad, bslot, rcols, drows, cap);
    return hipGetLastError();
}

// fused form (round != 0): k_la_cqs — the Gram-form pass, lambda, the scan and the pick of `round` in one launch
// (launch_tail_gram_batched is then told to skip its k_scansel)
bool cqs_usable(const ss_hip_ctx* ctx, uint32_t pmin_stride)
{
    const uint32_t nb = ((uint32_t)ctx->n + kCqChunk - 1) / kCqChunk;
    return ctx->batch_fused_scan != 0 && nb <= pmin_stride && nb <= 2048u;
}

template <typename T>
hipError_t launch_cq_gram_batched(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, const T* G, uint32_t gpitch,
                                  const T* c0b, uint32_t* nparts_out, const int32_t* bslot, uint32_t round, T tol, uint32_t max_iter)
{
    const uint32_t n = (uint32_t)ctx->n;
    const uint32_t nb = (n + kCqChunk - 1) / kCqChunk;
    if (nb > ws.dims.pmax_stride) return hipErrorInvalidValue;
    if (nparts_out) *nparts_out = nb;
    if (round != 0) {
        if (nb > ws.dims.pmin_stride) return hipErrorInvalidValue;
#define SS_CQS_LAUNCH(VEC)                                                                                                   \
        hipLaunchKernelGGL((k_la_cqs<T, VEC, 4, 4>), dim3(nb, nslots), dim3(kSmallThreads), 0, ctx->stream, G, bslot,              \
                           c0b, ws.x, (const T*)ws.d, ws.touched, ws.gam, n, gpitch, ws.dims, ws.c, ws.q,                           \
                           ws.pmax_val, ws.pmax_idx, ws.pmin_val, ws.pmin_idx, ws.insup, ws.st, round, tol, max_iter,               \
                           ctx->dev_flags, ws.trace, ws.trace_cap, ctx->zero_on_removal, ctx->tie_guard, ws.ndone, nslots,          \
                           (ctx->tie_rerun && !ctx->tie_guard) ? 1 : 0)
#define SS_CQS_WIDE(CPT, RIF)                                                                                               \
        {                                                                                                                    \
            const uint32_t nbw = (n + (uint32_t)(kSmallThreads * CPT) - 1u) / (uint32_t)(kSmallThreads * CPT);                     \
            hipLaunchKernelGGL((k_la_cqs<T, false, CPT, RIF>), dim3(nbw, nslots), dim3(kSmallThreads), 0, ctx->stream, G, bslot,   \
                               c0b, ws.x, (const T*)ws.d, ws.touched, ws.gam, n, gpitch, ws.dims, ws.c, ws.q,                      \
                               ws.pmax_val, ws.pmax_idx, ws.pmin_val, ws.pmin_idx, ws.insup, ws.st, round, tol, max_iter,          \
                               ctx->dev_flags, ws.trace, ws.trace_cap, ctx->zero_on_removal, ctx->tie_guard, ws.ndone, nslots,     \
                               (ctx->tie_rerun && !ctx->tie_guard) ? 1 : 0);                                                       \
            if (nparts_out) *nparts_out = nbw;                                                                               \
        }
        // columns per thread x Gram rows in flight (options cq_cols, cq_rows): wider workgroups read longer runs of a row
        const int cc = ctx->cq_cols, rr = ctx->cq_rows;
        if (cc == 8 && rr == 2) SS_CQS_WIDE(8, 2)
        else if (cc == 8 && rr == 4) SS_CQS_WIDE(8, 4)
        else if (cc == 8 && rr == 8) SS_CQS_WIDE(8, 8)
        else if (cc == 16 && rr == 2) SS_CQS_WIDE(16, 2)
        else if (cc == 16 && rr == 4) SS_CQS_WIDE(16, 4)
        else if (cc == 4 && rr == 8) SS_CQS_WIDE(4, 8)
        else if (cc == 32 && rr == 2) SS_CQS_WIDE(32, 2)
        else if (cc == 8 && rr == 1) SS_CQS_WIDE(8, 1)
        else if (cc == 16 && rr == 1) SS_CQS_WIDE(16, 1)
        else if (cc == 32 && rr == 1) SS_CQS_WIDE(32, 1)
        else if (cc == 16 && rr == 3) SS_CQS_WIDE(16, 3)
        else if (ctx->cq_vec4) SS_CQS_LAUNCH(true); else SS_CQS_LAUNCH(false);
#undef SS_CQS_WIDE
#undef SS_CQS_LAUNCH
        return hipGetLastError();
    }
#define SS_CQ_LAUNCH(VEC)                                                                                                    \
    hipLaunchKernelGGL((k_la_cq<T, VEC>), dim3(nb, nslots), dim3(kSmallThreads), 0, ctx->stream, G, bslot,                         \
                       c0b, (const T*)ws.x, (const T*)ws.d, (const uint32_t*)ws.touched, n, gpitch, ws.dims, ws.c, ws.q,            \
                       ws.pmax_val, ws.pmax_idx, (const DevState*)ws.st)
    if (ctx->cq_vec4) SS_CQ_LAUNCH(true); else SS_CQ_LAUNCH(false);
#undef SS_CQ_LAUNCH
    return hipGetLastError();
}

// scan + select, then the inverse update with u1 gathered from row idx of G, for all slots.  Column form
// (cols != nullptr): between the two, the Gram columns of this round's entering columns are formed — one pass over
// A per 64 live picks (launch_gemm64_tn_f32) into rows row_base + slot of the batch's column cache G.
template <typename T>
hipError_t launch_tail_gram_batched(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round, uint32_t nparts,
                                    T tol, uint32_t max_iter, const T* G, uint32_t gpitch, const BatchCols* cols, bool scan_done)
{
    const uint32_t n = (uint32_t)ctx->n;
    const uint32_t per_block = kSmallThreads * kScanPerThread;
    uint32_t ns = (n + per_block - 1) / per_block;
    if (ns > ws.dims.pmin_stride) ns = ws.dims.pmin_stride;
    // (many slots: fewer, longer workgroups per slot — the kernel grid-strides; with 64 per slot a launch of 4096 slots is
    // 262 144 workgroups whose reductions and tickets, not their 9 bytes per column, set its time; option scan_blocks)
    if (nslots >= 64u && ctx->scan_blocks > 0 && ns > (uint32_t)ctx->scan_blocks) ns = (uint32_t)ctx->scan_blocks;
    if (!scan_done)          // (fused form: k_la_cqs has scanned and picked already)
    hipLaunchKernelGGL((k_scansel<T>), dim3(ns, nslots), dim3(kSmallThreads), 0, ctx->stream, round, tol,
                       max_iter, n, ws.c, ws.q, ws.x, ws.d, ws.insup, ws.pmax_val, ws.pmax_idx,
                       nparts, ws.pmin_val, ws.pmin_idx, ws.gam, ws.touched, ws.dims, ws.st,
                       ctx->dev_flags, ws.trace, ws.trace_cap, ctx->zero_on_removal, ctx->tie_guard, ws.ndone, nslots,
                       (T*)nullptr, (const int32_t*)nullptr, (ctx->tie_rerun && !ctx->tie_guard) ? 1 : 0);
    if (cols != nullptr) {
        hipError_t e = launch_batch_cols(ctx, ws.st, nslots, cols->row_base, false, cols->bslot, cols->rcols, cols->drows, cols->cap);
        if (e != hipSuccess) return e;
        e = launch_batch_passes(ctx, cols, nslots);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_gramupd<T>), dim3(1, nslots), dim3(kUpdThreads), 0, ctx->stream,
                       static_cast<const T*>(ctx->At), ws.dims, ws.gam, ws.inv[0], ws.inv[1],
                       ws.u1, ws.u2, ws.sgn, ws.c, ws.q, ws.d, tol, ws.st, 0, (const T*)nullptr, (T*)nullptr,
                       G, cols != nullptr ? (const int32_t*)cols->bslot : (const int32_t*)nullptr, gpitch, 0, ctx->strict_sign);
    return hipGetLastError();
}

// the passes of one round of the column form: 64 picks per pass; a pass whose list is empty returns at once
hipError_t launch_batch_passes(const ss_hip_ctx* ctx, const BatchCols* cols, uint32_t nslots)
{
    for (uint32_t g = 0; g * 64u < nslots; ++g) {
        const hipError_t e = launch_gemm64_tn_f32(ctx, cols->rcols + g * 64u, cols->drows + g * 64u, cols->cache,
                                                  cols->pitch, nullptr);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <typename T>
hipError_t launch_la_scansel(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t round, uint32_t nparts, T tol,
                             uint32_t max_iter)
{
    const uint32_t n = (uint32_t)ctx->n;
    const uint32_t per_block = kSmallThreads * kScanPerThread;
    uint32_t ns = (n + per_block - 1) / per_block;
    if (ns > ws.dims.pmin_stride) ns = ws.dims.pmin_stride;
    hipLaunchKernelGGL((k_scansel<T>), dim3(ns, 1), dim3(kSmallThreads), 0, ctx->stream, round, tol,
                       max_iter, n, ws.c, ws.q, ws.x, ws.d, ws.insup, ws.pmax_val, ws.pmax_idx,
                       nparts, ws.pmin_val, ws.pmin_idx, ws.gam, ws.touched, ws.dims, ws.st,
                       ctx->dev_flags, ws.trace, ws.trace_cap, ctx->zero_on_removal, ctx->tie_guard, ws.ndone, 1u,
                       ws.tcand, (const int32_t*)ws.slot_of, (ctx->tie_rerun && !ctx->tie_guard) ? 1 : 0);
    return hipGetLastError();
}

// the same launch without the lookahead ranking and without the early exit: the reference-order engine
// follows the path wherever its own rounding takes it
template <typename T>
hipError_t launch_scansel_plain(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round, uint32_t nparts, T tol, uint32_t max_iter)
{
    const uint32_t n = (uint32_t)ctx->n;
    const uint32_t per_block = kSmallThreads * kScanPerThread;
    uint32_t ns = (n + per_block - 1) / per_block;
    if (ns > ws.dims.pmin_stride) ns = ws.dims.pmin_stride;
    hipLaunchKernelGGL((k_scansel<T>), dim3(ns, nslots), dim3(kSmallThreads), 0, ctx->stream, round, tol,
                       max_iter, n, ws.c, ws.q, ws.x, ws.d, ws.insup, ws.pmax_val, ws.pmax_idx,
                       nparts, ws.pmin_val, ws.pmin_idx, ws.gam, ws.touched, ws.dims, ws.st,
                       ctx->dev_flags, ws.trace, ws.trace_cap, ctx->zero_on_removal, ctx->tie_guard, ws.ndone, nslots,
                       (T*)nullptr, (const int32_t*)nullptr, 0);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_absmax(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t* nparts_out)
{
    const uint32_t n = (uint32_t)ctx->n;
    const uint32_t nb = (n + kAbsmaxChunk - 1) / kAbsmaxChunk;
    if (nb > ws.dims.pmax_stride) return hipErrorInvalidValue;
    if (nparts_out) *nparts_out = nb;
    hipLaunchKernelGGL((k_absmax<T>), dim3(nb, nslots), dim3(kSmallThreads), 0, ctx->stream, ws.c, n,
                       ws.dims, ws.pmax_val, ws.pmax_idx, ws.st);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_gemv_n(const ss_hip_ctx* ctx, const T* x_dev, T* y_dev)
{
    const uint32_t m = (uint32_t)ctx->m;
    hipLaunchKernelGGL((k_gemv_n<T>), dim3((m + kSmallThreads - 1) / kSmallThreads),
                       dim3(kSmallThreads), 0, ctx->stream, static_cast<const T*>(ctx->At), ctx->ldm,
                       m, (uint32_t)ctx->n, x_dev, y_dev);
    return hipGetLastError();
}

template hipError_t launch_init<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t, uint32_t, float);
template hipError_t launch_init<double>(const ss_hip_ctx*, Workspace<double>&, uint32_t, uint32_t, double);
template hipError_t launch_rp<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t);
template hipError_t launch_rp<double>(const ss_hip_ctx*, Workspace<double>&, uint32_t);
template hipError_t launch_iteration_tail<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t, uint32_t,
                                                 uint32_t, float, uint32_t);
template hipError_t launch_iteration_tail<double>(const ss_hip_ctx*, Workspace<double>&, uint32_t, uint32_t,
                                                  uint32_t, double, uint32_t);
template hipError_t launch_omp_tail<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t, uint32_t,
                                           uint32_t, float, uint32_t);
template hipError_t launch_omp_tail<double>(const ss_hip_ctx*, Workspace<double>&, uint32_t, uint32_t,
                                            uint32_t, double, uint32_t);
template hipError_t launch_la_init_pick<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t, float, bool);
template hipError_t launch_la_init_pick<double>(const ss_hip_ctx*, Workspace<double>&, uint32_t, double, bool);
template hipError_t launch_la_top<float>(const ss_hip_ctx*, Workspace<float>&, int, uint32_t);
template hipError_t launch_la_top<double>(const ss_hip_ctx*, Workspace<double>&, int, uint32_t);
template hipError_t launch_la_update<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t, float);
template hipError_t launch_la_update<double>(const ss_hip_ctx*, Workspace<double>&, uint32_t, double);
template hipError_t launch_la_cq<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t*);
template hipError_t launch_gram_guard_batched<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t, float);
template hipError_t launch_cq_gram_batched<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t, const float*, uint32_t, const float*, uint32_t*, const int32_t*, uint32_t, float, uint32_t);
template hipError_t launch_tail_gram_batched<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t, uint32_t, uint32_t, float, uint32_t, const float*, uint32_t, const BatchCols*, bool);
template hipError_t launch_la_cq<double>(const ss_hip_ctx*, Workspace<double>&, uint32_t*);
template hipError_t launch_la_iter<float>(const ss_hip_ctx*, Workspace<float>&, float, uint32_t);
template hipError_t launch_la_omp<float>(const ss_hip_ctx*, Workspace<float>&, float, uint32_t);
template hipError_t launch_la_omp<double>(const ss_hip_ctx*, Workspace<double>&, double, uint32_t);
template hipError_t launch_la_omp_update<float>(const ss_hip_ctx*, Workspace<float>&, float);
template hipError_t launch_la_omp_update<double>(const ss_hip_ctx*, Workspace<double>&, double);
template hipError_t launch_la_iter<double>(const ss_hip_ctx*, Workspace<double>&, double, uint32_t);
template hipError_t launch_scansel_plain<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t, uint32_t, uint32_t, float, uint32_t);
template hipError_t launch_scansel_plain<double>(const ss_hip_ctx*, Workspace<double>&, uint32_t, uint32_t, uint32_t, double, uint32_t);
template hipError_t launch_la_scansel<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t, uint32_t, float, uint32_t);
template hipError_t launch_la_scansel<double>(const ss_hip_ctx*, Workspace<double>&, uint32_t, uint32_t, double, uint32_t);
template hipError_t launch_absmax<float>(const ss_hip_ctx*, Workspace<float>&, uint32_t, uint32_t*);
template hipError_t launch_absmax<double>(const ss_hip_ctx*, Workspace<double>&, uint32_t, uint32_t*);
template hipError_t launch_gemv_n<float>(const ss_hip_ctx*, const float*, float*);
template hipError_t launch_gemv_n<double>(const ss_hip_ctx*, const double*, double*);

}  // namespace sship
