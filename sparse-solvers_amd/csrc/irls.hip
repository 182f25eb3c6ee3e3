// irls.hip — the reference's second solver, IRLS (iteratively reweighted least squares), on the device.
//
// Reference (paths under /root/reference):
//   qr_decomposition<T>        src/linalg/qr_decomposition.h:93-190   -> k_qr_panel / k_qr_panel_reg, k_qr_apply_panel(_reg), k_qr_formq_all / k_qr_formq_reg, k_irls_setup(_r), k_irls_gram_tiled
//   cholesky_decomposition<T>  src/linalg/cholesky_decomposition.h:56-100 -> k_irls_solve (in place)
//   irls_newton / run_solver   src/solvers/irls-cpu.cpp:39-124        -> k_irls_solve
//
// IRLS needs M >= N (a tall sensing matrix) and is dense O(M N^2) once + O(N^3) per iteration: it is
// off the Homotopy hot path and off the benchmark; it is here so that the reference's public API
// (ss::irls<T>, sparsesolvers.Irls) is complete on the device.  What the device layout buys:
//
//   * the factorisation runs once, at construction, on the column-contiguous copy of A every
//     other kernel of this library uses: one launch per Householder column, one workgroup per
//     trailing column (each recomputes the reflector from the read-only pivot column, so the
//     launch has no internal hand-off);
//   * irls_newton's N x N x M product Q^T (Q W) — recomputed by the reference in every iteration —
//     is (Q^T Q) with its columns scaled by w: Q^T Q is formed once (k_irls_setup) and the
//     iteration only scales it (same values up to the rounding of q*w before or after the sum);
//   * the whole iteration loop — scale, Cholesky, two triangular solves, t = Q s, x = Q^T t,
//     R x = x, threshold, second largest, reweighting, loop control — is one launch of one
//     workgroup: no host round trips; the O(M N) products are coalesced over the columns of Q^T.
//
// Summation orders differ from the reference's BLAS (unpinned there); parity is by tolerance.
#include "ss_hip_internal.h"
#include "ss_hip_device.h"

#include <algorithm>
#include <cstdlib>

namespace sship {

template <typename T>
struct IrlsState {
    T* Vt = nullptr;      // [n][ldm] Householder vectors, column k in row k (zero above the diagonal)
    T* Qt = nullptr;      // [n][ldm] Q^T: row j = column j of the thin Q
    T* R = nullptr;       // [n][n] upper triangular, row-major
    T* G0 = nullptr;      // [n][n] lower triangle of Q^T Q, row-major
    T* L = nullptr;       // [n][n] scratch: scaled matrix, then its Cholesky factor
    T* rdiag = nullptr;   // [n]
    T* vec = nullptr;     // qTb, s, xnext, w, x: 5 x [n]; then t, y: 2 x [ldm]
    IrlsResult* res = nullptr;   // device copy of the report
    void* ctl = nullptr;         // IrlsCtl<T>: loop state of the blocked form
    uint32_t* done_host = nullptr;   // pinned: the while-test of the blocked form, read once per Newton iteration
};

constexpr int kQrThreads = 256;
constexpr int kIrlsThreads = 1024;

template <typename T> __device__ __forceinline__ T t_abs(T v) { return v < T(0) ? -v : v; }

// maximum over the workgroup, the same value in every thread (sv: LDS scratch of >= 16 entries)
template <typename T>
__device__ __forceinline__ T block_max(T v, T* sv)
{
    v = max(v, dpp_mov<kDppXor1>(v));
    v = max(v, dpp_mov<kDppXor2>(v));
    v = max(v, dpp_mov<kDppHalfMirror>(v));
    v = max(v, dpp_mov<kDppMirror>(v));
    v = max(max(lane_value(v, 0), lane_value(v, 16)), max(lane_value(v, 32), lane_value(v, 48)));
    const int nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = sv[0];
    for (int w = 1; w < nw; ++w) r = max(r, sv[w]);
    __syncthreads();
    return r;
}

// The reflectors k0 .. k0 + nb - 1 of a factored panel applied, in order, to one trailing column per workgroup: the same
// statements as the reference's Householder step (qr_decomposition.h:111-134; the reflector read from Vt, where the pivot's workgroup left exactly the values the other
// workgroups form for themselves), so a column receives the same arithmetic in the same order — one launch per panel
// instead of one per reflector.
template <typename T>
__global__ __launch_bounds__(kQrThreads)
void k_qr_apply_panel(T* __restrict__ At, const T* __restrict__ Vt, uint32_t ldm, uint32_t m, uint32_t k0, uint32_t nb)
{
    __shared__ T sv[16];
    T* a = At + (size_t)(k0 + nb + blockIdx.x) * ldm;
    for (uint32_t k = k0; k < k0 + nb; ++k) {
        const T* v = Vt + (size_t)k * ldm;
        const T vk = v[k];
        if (vk == T(0)) continue;                               // (a zero pivot column: the others are left alone)
        T part = T(0);
        for (uint32_t i = k + threadIdx.x; i < m; i += kQrThreads) part += v[i] * a[i];
        const T s = block_sum(part, sv) / -vk;
        for (uint32_t i = k + threadIdx.x; i < m; i += kQrThreads) a[i] += v[i] * s;
        __syncthreads();
    }
}

// A whole panel of nb <= 32 columns factored in ONE launch: workgroup j owns column k0 + j.  It applies the reflectors
// k0 .. k0 + j - 1 to its column as each becomes available (k_qr_apply_panel's statements), then forms its own reflector
// (the Householder step's statements, qr_decomposition.h:111-134, its column being the pivot) and publishes it: Vt row, rdiag, then a flag — release / acquire at
// agent scope, the pattern of arrive_last (ss_hip_device.h): the reader's leader spins on the flag, acquires, and the
// workgroup reads the row with vector loads behind a barrier.  A workgroup only ever waits for lower-numbered ones, all
// nb <= 32 workgroups are resident: no deadlock.  Same arithmetic per column in the same order as one launch per
// reflector — the same bits — without the 32 launches (15.6 us each) of a panel.
template <typename T>
__global__ __launch_bounds__(kQrThreads)
void k_qr_panel(T* __restrict__ At, T* __restrict__ Vt, T* __restrict__ rdiag, uint32_t ldm, uint32_t m, uint32_t k0,
                uint32_t* __restrict__ ready)
{
    __shared__ T sv[16];
    __shared__ uint32_t s_gave_up;
    const uint32_t kown = k0 + blockIdx.x;
    T* a = At + (size_t)kown * ldm;
    if (threadIdx.x == 0) s_gave_up = 0u;
    __syncthreads();
    for (uint32_t k = k0; k < kown; ++k) {
        if (threadIdx.x == 0) {
            // (bounded: a producer that never shows up — it cannot happen with <= 32 resident workgroups — must not hang the device:
            // the column then publishes a NaN diagonal, which every later stage carries into the result)
            uint32_t spins = 0;
            while (__hip_atomic_load(&ready[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && spins < (1u << 26)) { __builtin_amdgcn_s_sleep(2); ++spins; }
            if (spins >= (1u << 26)) s_gave_up = 1u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (s_gave_up != 0u) break;
        const T* v = Vt + (size_t)k * ldm;
        const T vk = v[k];
        if (vk == T(0)) continue;                               // (a zero pivot column: the others are left alone)
        T part = T(0);
        for (uint32_t i = k + threadIdx.x; i < m; i += kQrThreads) part += v[i] * a[i];
        const T s = block_sum(part, sv) / -vk;
        for (uint32_t i = k + threadIdx.x; i < m; i += kQrThreads) a[i] += v[i] * s;
        __syncthreads();
    }
    // my column is the pivot of step kown
    const uint32_t k = kown;
    const T* piv = a;
    if (s_gave_up != 0u) {
        for (uint32_t i = threadIdx.x; i < ldm; i += kQrThreads) Vt[(size_t)k * ldm + i] = T(0);
        if (threadIdx.x == 0) rdiag[k] = T(0) / T(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(&ready[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    T amax = T(0);
    for (uint32_t i = k + threadIdx.x; i < m; i += kQrThreads) amax = max(amax, t_abs(piv[i]));
    amax = block_max(amax, sv);
    T nrm2 = T(0);
    if (amax > T(0)) {
        T part = T(0);
        for (uint32_t i = k + threadIdx.x; i < m; i += kQrThreads) { const T z = piv[i] / amax; part += z * z; }
        nrm2 = amax * sqrt(block_sum(part, sv));
        __syncthreads();
    }
    if (nrm2 == T(0)) {
        for (uint32_t i = threadIdx.x; i < ldm; i += kQrThreads) Vt[(size_t)k * ldm + i] = T(0);
        if (threadIdx.x == 0) rdiag[k] = -nrm2;
    } else {
        if (piv[k] < T(0)) nrm2 = -nrm2;
        const T vk = piv[k] / nrm2 + T(1);
        for (uint32_t i = threadIdx.x; i < ldm; i += kQrThreads)
            Vt[(size_t)k * ldm + i] = (i < k || i >= m) ? T(0) : (i == k ? vk : piv[i] / nrm2);
        if (threadIdx.x == 0) rdiag[k] = -nrm2;
    }
    // publish: every wave's stores drained, then the leader releases and raises the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(&ready[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Back-accumulation of the thin Q, all steps of one column in one workgroup: column j of Q starts as e_j at step j and then
// receives the reflectors j, j - 1, ... 0 — independent of every other column (the reference's back-accumulation, qr_decomposition.h:141-172, in its order).
template <typename T>
__global__ __launch_bounds__(kQrThreads)
void k_qr_formq_all(const T* __restrict__ Vt, T* __restrict__ Qt, uint32_t ldm, uint32_t m)
{
    __shared__ T sv[16];
    const uint32_t j = blockIdx.x;
    T* qc = Qt + (size_t)j * ldm;
    for (uint32_t i = threadIdx.x; i < ldm; i += kQrThreads) qc[i] = (i == j) ? T(1) : T(0);
    __syncthreads();
    for (uint32_t kk = j + 1; kk-- > 0;) {
        const T* v = Vt + (size_t)kk * ldm;
        const T vk = v[kk];
        if (vk == T(0)) continue;
        T part = T(0);
        for (uint32_t i = kk + threadIdx.x; i < m; i += kQrThreads) part += v[i] * qc[i];
        const T s = -block_sum(part, sv) / vk;
        for (uint32_t i = kk + threadIdx.x; i < m; i += kQrThreads) qc[i] += s * v[i];
        __syncthreads();
    }
}

// ==== register-resident forms of the factorisation kernels (round 4) ==========================================================
// The kernels above walk a column in global memory for every reflector (read for the dot product, read + write for the update:
// three passes of m elements per reflector, through L2).  ldm is a multiple of 256, so a workgroup of 256 threads can HOLD a column:
// thread t keeps the rows t + 256 e, e < RPT = ldm / 256, in registers from the first reflector to the last; a reflector then costs
// its own read (shared by the NC columns a workgroup carries), RPT fmas per column, ONE barrier (the partial sums go through
// alternating LDS slots) and RPT fmas again.  The statements are the reference's (qr_decomposition.h:111-172); the rows of a partial
// sum are the same residue class mod 256 as before, the classes are combined in another order (parity by tolerance, as for every
// sum of this file).  Reflector rows above the diagonal and beyond m are stored as zeros in Vt, so no row range is tested: those
// products are exact zeros.  Used for ldm <= 8192 (RPT <= 32); taller matrices keep the kernels above.

// sum of NC values over the workgroup (256 threads), every thread receives them; pp alternates the LDS slot: one barrier per call
template <typename T, int NC>
__device__ __forceinline__ void block_sum_pp(T (&v)[NC], T (*sv)[NC][4], int& pp)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const T w = wave_sum(v[c]);
        if (lane == 0) sv[pp][c][wave] = w;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = ((sv[pp][c][0] + sv[pp][c][1]) + sv[pp][c][2]) + sv[pp][c][3];
    pp ^= 1;
}

// (RPT is the next of 4, 8, 16, 32 at or above ldm / 256: the slots past ldm hold zeros and are never stored)
template <typename T, int RPT>
__device__ __forceinline__ void col_load(T (&a)[RPT], const T* __restrict__ col, uint32_t ldm)
{
#pragma unroll
    for (int e = 0; e < RPT; ++e) { const uint32_t i = threadIdx.x + 256u * (uint32_t)e; a[e] = i < ldm ? col[i] : T(0); }
}
template <typename T, int RPT>
__device__ __forceinline__ void col_store(const T (&a)[RPT], T* __restrict__ col, uint32_t ldm)
{
#pragma unroll
    for (int e = 0; e < RPT; ++e) { const uint32_t i = threadIdx.x + 256u * (uint32_t)e; if (i < ldm) col[i] = a[e]; }
}

// the reflectors k0 .. k0 + nb - 1 applied, in order, to NC trailing columns per workgroup (k_qr_apply_panel's statements)
template <typename T, int RPT, int NC>
__global__ __launch_bounds__(kQrThreads)
void k_qr_apply_panel_reg(T* __restrict__ At, const T* __restrict__ Vt, uint32_t ldm, uint32_t n, uint32_t k0, uint32_t nb)
{
    __shared__ T sv[2][NC][4];
    T a[NC][RPT], v[RPT], vn[RPT];
    const uint32_t c0 = k0 + nb + NC * blockIdx.x;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint32_t col = c0 + (uint32_t)c < n ? c0 + (uint32_t)c : n - 1u;      // (a column past the end: a copy, never stored)
        col_load<T, RPT>(a[c], At + (size_t)col * ldm, ldm);
    }
    col_load<T, RPT>(v, Vt + (size_t)k0 * ldm, ldm);
    int pp = 0;
    for (uint32_t k = k0; k < k0 + nb; ++k) {
        const T vk = Vt[(size_t)k * ldm + k];
        if (k + 1u < k0 + nb) col_load<T, RPT>(vn, Vt + (size_t)(k + 1u) * ldm, ldm);
        if (vk != T(0)) {                                       // (a zero pivot column: the others are left alone)
            T part[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                part[c] = T(0);
#pragma unroll
                for (int e = 0; e < RPT; ++e) part[c] += v[e] * a[c][e];
            }
            block_sum_pp<T, NC>(part, sv, pp);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const T sc = part[c] / -vk;
#pragma unroll
                for (int e = 0; e < RPT; ++e) a[c][e] += v[e] * sc;
            }
        }
#pragma unroll
        for (int e = 0; e < RPT; ++e) v[e] = vn[e];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (c0 + (uint32_t)c < n) col_store<T, RPT>(a[c], At + (size_t)(c0 + (uint32_t)c) * ldm, ldm);
}

// a panel of nb <= 32 columns factored in one launch (k_qr_panel's protocol and statements), the workgroup's column in registers
template <typename T, int RPT>
__global__ __launch_bounds__(kQrThreads)
void k_qr_panel_reg(T* __restrict__ At, T* __restrict__ Vt, T* __restrict__ rdiag, uint32_t ldm, uint32_t m, uint32_t k0,
                    uint32_t* __restrict__ ready)
{
    __shared__ T sv[2][1][4];
    __shared__ T sm[16];
    __shared__ T s_piv;
    __shared__ uint32_t s_gave_up;
    const uint32_t kown = k0 + blockIdx.x, tid = threadIdx.x;
    T a[RPT], v[RPT];
    col_load<T, RPT>(a, At + (size_t)kown * ldm, ldm);
    if (tid == 0) s_gave_up = 0u;
    __syncthreads();
    int pp = 0;
    for (uint32_t k = k0; k < kown; ++k) {
        if (tid == 0) {
            // (bounded, like k_qr_panel: a producer that never shows up must not hang the device)
            uint32_t spins = 0;
            while (__hip_atomic_load(&ready[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && spins < (1u << 26)) { __builtin_amdgcn_s_sleep(1); ++spins; }
            if (spins >= (1u << 26)) s_gave_up = 1u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (s_gave_up != 0u) break;
        const T* vr = Vt + (size_t)k * ldm;
        const T vk = vr[k];
        if (vk == T(0)) continue;
        col_load<T, RPT>(v, vr, ldm);
        T part[1] = { T(0) };
#pragma unroll
        for (int e = 0; e < RPT; ++e) part[0] += v[e] * a[e];
        block_sum_pp<T, 1>(part, sv, pp);
        const T sc = part[0] / -vk;
#pragma unroll
        for (int e = 0; e < RPT; ++e) a[e] += v[e] * sc;
    }
    // the column as the panel's reflectors have left it (its rows above the diagonal are R's)
    col_store<T, RPT>(a, At + (size_t)kown * ldm, ldm);
    const uint32_t k = kown;
    T* vout = Vt + (size_t)k * ldm;
    if (s_gave_up != 0u) {
#pragma unroll
        for (int e = 0; e < RPT; ++e) if (tid + 256u * (uint32_t)e < ldm) vout[tid + 256u * (uint32_t)e] = T(0);
        if (tid == 0) rdiag[k] = T(0) / T(0);
    } else {
        // my column is the pivot of step kown (qr_decomposition.h:111-134): 2-norm of rows k .. m - 1 without overflow
        T amax = T(0), mine = T(0);
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
            const uint32_t i = tid + 256u * (uint32_t)e;
            if (i >= k && i < m) amax = max(amax, t_abs(a[e]));
            if (i == k) mine = a[e];
        }
        if (tid == (k & 255u)) s_piv = mine;
        amax = block_max(amax, sm);                              // (its barriers publish s_piv too)
        const T pivk = s_piv;
        T nrm2 = T(0);
        if (amax > T(0)) {
            T part[1] = { T(0) };
#pragma unroll
            for (int e = 0; e < RPT; ++e) {
                const uint32_t i = tid + 256u * (uint32_t)e;
                if (i >= k && i < m) { const T z = a[e] / amax; part[0] += z * z; }
            }
            block_sum_pp<T, 1>(part, sv, pp);
            nrm2 = amax * sqrt(part[0]);
        }
        if (nrm2 == T(0)) {
#pragma unroll
            for (int e = 0; e < RPT; ++e) if (tid + 256u * (uint32_t)e < ldm) vout[tid + 256u * (uint32_t)e] = T(0);
            if (tid == 0) rdiag[k] = -nrm2;
        } else {
            if (pivk < T(0)) nrm2 = -nrm2;
            const T vk = pivk / nrm2 + T(1);
#pragma unroll
            for (int e = 0; e < RPT; ++e) {
                const uint32_t i = tid + 256u * (uint32_t)e;
                if (i < ldm) vout[i] = (i < k || i >= m) ? T(0) : (i == k ? vk : a[e] / nrm2);
            }
            if (tid == 0) rdiag[k] = -nrm2;
        }
    }
    // publish: every wave's stores drained, then the leader releases and raises the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(&ready[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// back-accumulation of the thin Q, NC neighbouring columns per workgroup: column j starts as e_j and receives the reflectors j,
// j - 1, ... 0 (k_qr_formq_all's statements).  A reflector kk > j meets e_j with v_kk[j] = 0: an exact no-op, so the NC columns share
// one walk from the last of them down.
template <typename T, int RPT, int NC>
__global__ __launch_bounds__(kQrThreads)
void k_qr_formq_reg(const T* __restrict__ Vt, T* __restrict__ Qt, uint32_t ldm, uint32_t n)
{
    __shared__ T sv[2][NC][4];
    T q[NC][RPT], v[RPT], vn[RPT];
    const uint32_t j0 = NC * blockIdx.x;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < RPT; ++e) q[c][e] = (threadIdx.x + 256u * (uint32_t)e == j0 + (uint32_t)c) ? T(1) : T(0);
    const uint32_t top = (j0 + NC - 1u < n ? j0 + NC - 1u : n - 1u);
    col_load<T, RPT>(v, Vt + (size_t)top * ldm, ldm);
    int pp = 0;
    for (uint32_t kk = top + 1u; kk-- > 0;) {
        const T vk = Vt[(size_t)kk * ldm + kk];
        if (kk > 0u) col_load<T, RPT>(vn, Vt + (size_t)(kk - 1u) * ldm, ldm);
        if (vk != T(0)) {
            T part[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                part[c] = T(0);
#pragma unroll
                for (int e = 0; e < RPT; ++e) part[c] += v[e] * q[c][e];
            }
            block_sum_pp<T, NC>(part, sv, pp);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const T sc = -part[c] / vk;
#pragma unroll
                for (int e = 0; e < RPT; ++e) q[c][e] += sc * v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < RPT; ++e) v[e] = vn[e];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (j0 + (uint32_t)c < n) col_store<T, RPT>(q[c], Qt + (size_t)(j0 + (uint32_t)c) * ldm, ldm);
}

// R (qr_decomposition.h:174-190); workgroup i does row i
template <typename T>
__global__ __launch_bounds__(kQrThreads)
void k_irls_setup_r(const T* __restrict__ At, const T* __restrict__ rdiag, T* __restrict__ R, uint32_t ldm, uint32_t n)
{
    const uint32_t i = blockIdx.x;
    for (uint32_t j = threadIdx.x; j < n; j += kQrThreads)
        R[(size_t)i * n + j] = j > i ? At[(size_t)j * ldm + i] : (j == i ? rdiag[i] : T(0));
}

// the lower triangle of Q^T Q by 32 x 32 tiles: workgroup b = tile (bi >= bj); 64 rows of the two blocks of columns staged in LDS
// per step, a 2 x 2 patch of the tile per thread (k_irls_setup walks a whole column of Q per entry: n^2 / 2 columns through L2)
template <typename T>
__global__ __launch_bounds__(256)
void k_irls_gram_tiled(const T* __restrict__ Qt, T* __restrict__ G0, uint32_t ldm, uint32_t m, uint32_t n)
{
    constexpr uint32_t TB = 32, RC = 64;
    __shared__ T sI[TB][RC + 1];
    __shared__ T sJ[TB][RC + 1];
    const uint32_t b = blockIdx.x;
    uint32_t bi = (uint32_t)((sqrtf(8.f * (float)b + 1.f) - 1.f) * 0.5f);
    while (bi * (bi + 1u) / 2u > b) --bi;
    while ((bi + 1u) * (bi + 2u) / 2u <= b) ++bi;
    const uint32_t bj = b - bi * (bi + 1u) / 2u;                 // bj <= bi
    const uint32_t tid = threadIdx.x, ti = tid >> 4, tj = tid & 15u;
    T acc[2][2] = { { T(0), T(0) }, { T(0), T(0) } };
    const uint32_t lc = tid >> 3, lr = (tid & 7u) * 8u;           // staging: column lc of the block, rows lr .. lr + 7 of the step
    const uint32_t ci = bi * TB + lc, cj = bj * TB + lc;
    for (uint32_t r0 = 0; r0 < m; r0 += RC) {
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) {
            const uint32_t r = r0 + lr + q;
            sI[lc][lr + q] = (ci < n && r < m) ? Qt[(size_t)ci * ldm + r] : T(0);
            sJ[lc][lr + q] = (cj < n && r < m) ? Qt[(size_t)cj * ldm + r] : T(0);
        }
        __syncthreads();
#pragma unroll 8
        for (uint32_t r = 0; r < RC; ++r) {
            const T a0 = sI[2u * ti][r], a1 = sI[2u * ti + 1u][r], b0 = sJ[2u * tj][r], b1 = sJ[2u * tj + 1u][r];
            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
        }
        __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const uint32_t i = bi * TB + 2u * ti + (uint32_t)x, j = bj * TB + 2u * tj + (uint32_t)y;
            if (i < n && j < n) {
                if (j <= i) G0[(size_t)i * n + j] = acc[x][y];
                else if (bi == bj) G0[(size_t)i * n + j] = T(0);          // (above the diagonal inside a diagonal tile)
            }
        }
}

// R (qr_decomposition.h:174-190) and the lower triangle of Q^T Q; workgroup i does row i
template <typename T>
__global__ __launch_bounds__(kQrThreads)
void k_irls_setup(const T* __restrict__ At, const T* __restrict__ Qt, const T* __restrict__ rdiag,
                  T* __restrict__ R, T* __restrict__ G0, uint32_t ldm, uint32_t m, uint32_t n)
{
    const uint32_t i = blockIdx.x;
    for (uint32_t j = threadIdx.x; j < n; j += kQrThreads)
        R[(size_t)i * n + j] = j > i ? At[(size_t)j * ldm + i] : (j == i ? rdiag[i] : T(0));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* qi = Qt + (size_t)i * ldm;
    for (uint32_t j = wave; j < n; j += kQrThreads / 64) {
        T acc = T(0);
        if (j <= i) {
            const T* qj = Qt + (size_t)j * ldm;
            for (uint32_t r = lane; r < m; r += 64) acc += qi[r] * qj[r];
        }
        acc = wave_sum(acc);
        if (lane == 0) G0[(size_t)i * n + j] = j <= i ? acc : T(0);
    }
}

// block-wide (value, index) maximum by value, first index on ties; excl: index to leave out
template <typename T>
__device__ __forceinline__ void block_max_excl(const T* v, uint32_t n, uint32_t excl, T& mv, uint32_t& mi, T* sv, uint32_t* si)
{
    mv = -Lim<T>::max();
    mi = 0xffffffffu;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
        if (i != excl && better_max(v[i], i, mv, mi)) { mv = v[i]; mi = i; }
    block_reduce_pair<T, true>(mv, mi, sv, si);
    __syncthreads();
}

// The whole solve: irls-cpu.cpp:63-124 with irls_newton :39-61 inlined.  One workgroup.
template <typename T>
__global__ __launch_bounds__(kIrlsThreads)
void k_irls_solve(const T* __restrict__ Qt, const T* __restrict__ R, const T* __restrict__ G0, T* __restrict__ L,
                  T* __restrict__ vec, uint32_t ldm, uint32_t m, uint32_t n, T tol, uint32_t max_iter,
                  IrlsResult* res)
{
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    T* qTb = vec;
    T* s = vec + n;
    T* xnext = vec + 2 * (size_t)n;
    T* w = vec + 3 * (size_t)n;
    T* x = vec + 4 * (size_t)n;
    T* t = vec + 5 * (size_t)n;
    const T* y = t + ldm;
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NW = kIrlsThreads / 64;
    const T p = T(0.9);

    for (uint32_t i = tid; i < n; i += kIrlsThreads) { x[i] = T(0); w[i] = T(1); xnext[i] = T(1); }
    // qTb = Q^T y (:54) does not change over the iterations
    for (uint32_t j = wave; j < n; j += NW) {
        T acc = T(0);
        const T* qj = Qt + (size_t)j * ldm;
        for (uint32_t r = lane; r < m; r += 64) acc += qj[r] * y[r];
        acc = wave_sum(acc);
        if (lane == 0) qTb[j] = acc;
    }
    __syncthreads();

    uint32_t iter = 0;
    int spd_error = 0;
    T abstol = T(1), eps = T(1), second = T(0);
    do {
        // ---- irls_newton ---------------------------------------------------------------------
        // tril(Q^T (Q w)) = tril(Q^T Q) with column j scaled by w_j (:48-49, cholesky_decomposition.h:67)
        for (size_t e = tid; e < (size_t)n * n; e += kIrlsThreads) {
            const uint32_t i = (uint32_t)(e / n), j = (uint32_t)(e - (size_t)i * n);
            L[e] = j <= i ? G0[e] * w[j] : T(0);
        }
        __syncthreads();
        // Cholesky, column by column (cholesky_decomposition.h:69-82)
        bool isspd = true;
        for (uint32_t j = 0; j < n; ++j) {
            if (j > 0) {
                const T* lj = L + (size_t)j * n;
                for (uint32_t i = j + wave; i < n; i += NW) {
                    const T* li = L + (size_t)i * n;
                    T acc = T(0);
                    for (uint32_t k2 = lane; k2 < j; k2 += 64) acc += li[k2] * lj[k2];
                    acc = wave_sum(acc);
                    if (lane == 0) s[i] = acc;                   // column update, applied below
                }
                __syncthreads();
                for (uint32_t i = j + tid; i < n; i += kIrlsThreads) L[(size_t)i * n + j] -= s[i];
                __syncthreads();
            }
            const T ajj = sqrt(L[(size_t)j * n + j]);
            if (ajj <= Lim<T>::eps()) isspd = false;          // :78 (a NaN pivot passes, as in the reference)
            const T inv = T(1) / ajj;
            __syncthreads();
            for (uint32_t i = j + tid; i < n; i += kIrlsThreads) L[(size_t)i * n + j] *= inv;
            __syncthreads();
        }
        if (!isspd) { spd_error = 1; break; }                   // uniform: every thread saw the same pivots
        // s = (L L^T)^-1 qTb (:55, cholesky_decomposition.h:90-100)
        for (uint32_t i = tid; i < n; i += kIrlsThreads) s[i] = qTb[i];
        __syncthreads();
        for (uint32_t i = 0; i < n; ++i) {                       // L z = b
            T part = T(0);
            for (uint32_t k2 = tid; k2 < i; k2 += kIrlsThreads) part += L[(size_t)i * n + k2] * s[k2];
            const T acc = block_sum(part, sv);
            __syncthreads();
            if (tid == 0) s[i] = (s[i] - acc) / L[(size_t)i * n + i];
            __syncthreads();
        }
        for (uint32_t ii = n; ii-- > 0;) {                       // L^T x = z
            T part = T(0);
            for (uint32_t k2 = ii + 1 + tid; k2 < n; k2 += kIrlsThreads) part += L[(size_t)k2 * n + ii] * s[k2];
            const T acc = block_sum(part, sv);
            __syncthreads();
            if (tid == 0) s[ii] = (s[ii] - acc) / L[(size_t)ii * n + ii];
            __syncthreads();
        }
        // t = Q s (:56): thread per row, coalesced over the rows of Q^T
        for (uint32_t r = tid; r < m; r += kIrlsThreads) {
            T acc = T(0);
            for (uint32_t j = 0; j < n; ++j) acc += Qt[(size_t)j * ldm + r] * s[j];
            t[r] = acc;
        }
        __syncthreads();
        // xnext = Q^T t (:58)
        for (uint32_t j = wave; j < n; j += NW) {
            T acc = T(0);
            const T* qj = Qt + (size_t)j * ldm;
            for (uint32_t r = lane; r < m; r += 64) acc += qj[r] * t[r];
            acc = wave_sum(acc);
            if (lane == 0) xnext[j] = acc;
        }
        __syncthreads();
        // R x = x, upper triangular (:59)
        for (uint32_t ii = n; ii-- > 0;) {
            T part = T(0);
            for (uint32_t k2 = ii + 1 + tid; k2 < n; k2 += kIrlsThreads) part += R[(size_t)ii * n + k2] * xnext[k2];
            const T acc = block_sum(part, sv);
            __syncthreads();
            if (tid == 0) xnext[ii] = (xnext[ii] - acc) / R[(size_t)ii * n + ii];
            __syncthreads();
        }
        // ---- run_solver --------------------------------------------------------------------------
        T mx;
        uint32_t mi;
        block_max_excl(xnext, n, 0xffffffffu, mx, mi, sv, si);
        abstol = mx * tol;                                        // :100
        for (uint32_t i = tid; i < n; i += kIrlsThreads) {        // :103-104
            const T v = xnext[i] < abstol ? T(0) : xnext[i];
            xnext[i] = v;
            x[i] = v;
        }
        __syncthreads();
        // second largest value (:107): the largest once the (first) largest is left out
        block_max_excl(xnext, n, 0xffffffffu, mx, mi, sv, si);
        if (n >= 2) {
            T m2;
            uint32_t i2;
            block_max_excl(xnext, n, mi, m2, i2, sv, si);
            second = m2;
        } else {
            second = mx;
        }
        {
            const T cand = second / T(n);                         // :110
            if (cand < eps) eps = cand;
        }
        T part = T(0);
        for (uint32_t i = tid; i < n; i += kIrlsThreads) {        // :113
            const T v = (T)pow((double)(x[i] * x[i] + eps), (double)p / 2.0 - 1.0);
            w[i] = v;
            part += v;
        }
        const T sum = block_sum(part, sv);
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kIrlsThreads) w[i] /= sum;   // :114
        __syncthreads();
        ++iter;
    } while (iter < max_iter && second > abstol);

    // finally, normalise x (:121)
    T part = T(0);
    for (uint32_t i = tid; i < n; i += kIrlsThreads) part += x[i];
    const T sum = block_sum(part, sv);
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kIrlsThreads) x[i] /= sum;
    if (tid == 0) {
        res->iter = iter;
        res->spd_failure = (uint32_t)spd_error;
        res->solution_error = (double)eps;
    }
}


// ==== the Newton loop as a chain of launches (n >= kIrlsBlockedMin): blocked Cholesky, blocked triangular solves, ==========
// ==== the products with Q on all CUs ==========================================================================================
// k_irls_solve above keeps the whole loop in ONE workgroup: at n = 1024 its column-by-column Cholesky is 1024 steps of two
// barriers with every inner product a round trip to L2 (30 ms per iteration), its three triangular solves 2048 barriers each,
// and the two products with Q (16 MB each) pass through one CU.  Here: the Cholesky factor by panels of 32 columns — diagonal
// block in LDS by one wave, the panel below it one row per thread, the trailing update by 32 x 32 tiles on all CUs (3 launches
// per panel) —, the triangular solves by blocks of 32 (one barrier pair per block instead of per unknown), t = Q s and Q^T t
// spread over the chip.  The host reads one word per Newton iteration (the while-test).  Same formulas as the reference
// (irls-cpu.cpp:39-124, cholesky_decomposition.h:56-100); sums are formed in different orders (parity by tolerance, as before).
constexpr uint32_t kIrlsBlockedMin = 96;
constexpr uint32_t kChB = 32;                     // panel width / solve block

template <typename T>
struct IrlsCtl {                                  // device-resident loop state
    T eps, abstol, second;
    uint32_t iter, done, spd_bad, pad_;
};

template <typename T>
__global__ __launch_bounds__(256)
void k_irls_init(T* __restrict__ vec, uint32_t n, IrlsCtl<T>* __restrict__ ctl)
{
    T* xnext = vec + 2 * (size_t)n;
    T* w = vec + 3 * (size_t)n;
    T* x = vec + 4 * (size_t)n;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) { x[i] = T(0); w[i] = T(1); xnext[i] = T(1); }
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->eps = T(1); ctl->abstol = T(1); ctl->second = T(0); ctl->iter = 0; ctl->done = 0; ctl->spd_bad = 0; }
}

// out[j] = Qt[j] . v   (one wave per column of Q; Q^T y and Q^T t)
template <typename T>
__global__ __launch_bounds__(256)
void k_irls_qt_vec(const T* __restrict__ Qt, uint32_t ldm, uint32_t m, uint32_t n, const T* __restrict__ v, T* __restrict__ out,
                   const IrlsCtl<T>* __restrict__ ctl)
{
    if (ctl != nullptr && ctl->done) return;
    const uint32_t lane = threadIdx.x & 63u, j = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (j >= n) return;
    const T* q = Qt + (size_t)j * ldm;
    T acc = T(0);
    for (uint32_t r = lane; r < m; r += 64u) acc += q[r] * v[r];
    acc = wave_sum(acc);
    if (lane == 0) out[j] = acc;
}

// t[r] = sum_j Qt[j][r] s[j]   (a thread per row, coalesced over the rows of Q^T)
template <typename T>
__global__ __launch_bounds__(256)
void k_irls_q_vec(const T* __restrict__ Qt, uint32_t ldm, uint32_t m, uint32_t n, const T* __restrict__ s, T* __restrict__ t,
                  const IrlsCtl<T>* __restrict__ ctl)
{
    if (ctl->done) return;
    __shared__ T ss[256];
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    T acc = T(0);
    for (uint32_t j0 = 0; j0 < n; j0 += 256u) {
        __syncthreads();
        if (j0 + threadIdx.x < n) ss[threadIdx.x] = s[j0 + threadIdx.x];
        __syncthreads();
        const uint32_t cnt = n - j0 < 256u ? n - j0 : 256u;
        if (r < m)
            for (uint32_t j = 0; j < cnt; ++j) acc += Qt[(size_t)(j0 + j) * ldm + r] * ss[j];
    }
    if (r < m) t[r] = acc;
}

// L = tril(Q^T Q) with column j scaled by w_j (irls-cpu.cpp:48-49)
template <typename T>
__global__ __launch_bounds__(256)
void k_irls_scale(const T* __restrict__ G0, const T* __restrict__ w, T* __restrict__ L, uint32_t n, const IrlsCtl<T>* __restrict__ ctl)
{
    if (ctl->done) return;
    const size_t e = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (e >= (size_t)n * n) return;
    const uint32_t i = (uint32_t)(e / n), j = (uint32_t)(e - (size_t)i * n);
    L[e] = j <= i ? G0[e] * w[j] : T(0);
}

// Cholesky, panel k0: the diagonal block (one wave, in LDS), then — same launch, the other workgroups wait for nothing: they
// are a second kernel — see k_chol_below.  Pivot rule of the reference: ajj = sqrt(a_jj); ajj <= eps marks the matrix not SPD.
template <typename T>
__global__ __launch_bounds__(64)
void k_chol_diag(T* __restrict__ L, uint32_t n, uint32_t k0, IrlsCtl<T>* __restrict__ ctl)
{
    // Thread l keeps ROW l of the block in registers; a step publishes the pivot and the scaled column through LDS (two
    // barriers) and updates the rows with register arithmetic — the same statements per element in the same order as the
    // all-in-LDS form (45 us per block: every a -= b c there was two dependent LDS round trips), 32 x faster steps.
    if (ctl->done) return;
    __shared__ T colbuf[kChB];
    __shared__ T s_diag;
    const uint32_t l = threadIdx.x;
    const uint32_t nb = n - k0 < kChB ? n - k0 : kChB;
    T row[kChB];
#pragma unroll
    for (uint32_t c = 0; c < kChB; ++c) row[c] = (l < nb && c <= l) ? L[(size_t)(k0 + l) * n + k0 + c] : T(0);
    bool bad = false;
#pragma unroll
    for (uint32_t c = 0; c < kChB; ++c) {
        if (c < nb) {                                             // (uniform)
            if (l == c) s_diag = row[c];
            __syncthreads();
            const T ajj = sqrt(s_diag);
            if (ajj <= Lim<T>::eps()) bad = true;
            const T inv = T(1) / ajj;
            if (l >= c && l < nb) row[c] *= inv;
            if (l < kChB) colbuf[l] = row[c];
            __syncthreads();
            // right-looking inside the block: a_lc' -= l_lc l_c'c for c' > c, l >= c'
#pragma unroll
            for (uint32_t c2 = c + 1; c2 < kChB; ++c2)
                if (c2 < nb && l >= c2 && l < nb) row[c2] -= row[c] * colbuf[c2];
        }
    }
    if (l < nb) {
#pragma unroll
        for (uint32_t c = 0; c < kChB; ++c)
            if (c <= l) L[(size_t)(k0 + l) * n + k0 + c] = row[c];
    }
    if (l == 0 && bad) ctl->spd_bad = 1;
}

// ... the panel below the diagonal block: row i solves x L11^T = a_i (32 unknowns), one row per thread
template <typename T>
__global__ __launch_bounds__(256)
void k_chol_below(T* __restrict__ L, uint32_t n, uint32_t k0, const IrlsCtl<T>* __restrict__ ctl)
{
    if (ctl->done) return;
    __shared__ T D[kChB][kChB + 1];
    const uint32_t nb = kChB;                                   // (called only for full diagonal blocks with rows below)
    for (uint32_t e = threadIdx.x; e < kChB * kChB; e += 256u) { const uint32_t a = e / kChB, b = e % kChB; D[a][b] = L[(size_t)(k0 + a) * n + k0 + b]; }
    __syncthreads();
    const uint32_t i = k0 + kChB + blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    T* row = L + (size_t)i * n + k0;
    T xv[kChB];
#pragma unroll
    for (uint32_t c = 0; c < nb; ++c) {
        T a = row[c];
#pragma unroll
        for (uint32_t c2 = 0; c2 < c; ++c2) a -= xv[c2] * D[c][c2];
        xv[c] = a / D[c][c];
    }
#pragma unroll
    for (uint32_t c = 0; c < nb; ++c) row[c] = xv[c];
}

// ... the trailing update, lower triangle only: A22[i][j] -= sum_c P[i][c] P[j][c], 32 x 32 tiles
template <typename T>
__global__ __launch_bounds__(256)
void k_chol_trail(T* __restrict__ L, uint32_t n, uint32_t k0, const IrlsCtl<T>* __restrict__ ctl)
{
    if (ctl->done) return;
    __shared__ T Pi[kChB][kChB + 1], Pj[kChB][kChB + 1];
    const uint32_t b = blockIdx.x;
    uint32_t t = (uint32_t)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while (t * (t + 1u) / 2u > b) --t;
    while ((t + 1u) * (t + 2u) / 2u <= b) ++t;
    const uint32_t ti = t, tj = b - t * (t + 1u) / 2u;          // ti >= tj
    const uint32_t base = k0 + kChB;
    const uint32_t i0 = base + ti * kChB, j0 = base + tj * kChB;
    for (uint32_t e = threadIdx.x; e < kChB * kChB; e += 256u) {
        const uint32_t a = e / kChB, c = e % kChB;
        Pi[a][c] = i0 + a < n ? L[(size_t)(i0 + a) * n + k0 + c] : T(0);
        Pj[a][c] = j0 + a < n ? L[(size_t)(j0 + a) * n + k0 + c] : T(0);
    }
    __syncthreads();
    const uint32_t jj = threadIdx.x & 31u, ib = threadIdx.x >> 5;           // 8 row groups of 4
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t ii = ib * 4u + q;
        const uint32_t i = i0 + ii, j = j0 + jj;
        if (i < n && j < n && j <= i) {
            T acc = T(0);
#pragma unroll
            for (uint32_t c = 0; c < kChB; ++c) acc += Pi[ii][c] * Pj[jj][c];
            L[(size_t)i * n + j] -= acc;
        }
    }
}

// s = (L L^T)^-1 qTb by blocks of 32 (cholesky_decomposition.h:90-100).  One workgroup; s in LDS.
template <typename T>
__global__ __launch_bounds__(kIrlsThreads)
void k_irls_chol_solve(const T* __restrict__ L, uint32_t n, T* __restrict__ vec, IrlsCtl<T>* __restrict__ ctl)
{
    if (ctl->done) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cs[];
    T* ss = reinterpret_cast<T*>(smem_cs);                       // [n]
    __shared__ T D[kChB][kChB + 1];
    __shared__ T z[kChB];
    const uint32_t tid = threadIdx.x;
    if (ctl->spd_bad) {                                          // irls-cpu.cpp: a failed factorisation ends the loop
        if (tid == 0) ctl->done = 1;
        return;
    }
    const T* qTb = vec;
    T* s = vec + n;
    for (uint32_t i = tid; i < n; i += kIrlsThreads) ss[i] = qTb[i];
    __syncthreads();
    // L z = b, top to bottom
    for (uint32_t b0 = 0; b0 < n; b0 += kChB) {
        const uint32_t nb = n - b0 < kChB ? n - b0 : kChB;
        for (uint32_t e = tid; e < kChB * kChB; e += kIrlsThreads) { const uint32_t a = e / kChB, c = e % kChB; D[a][c] = (a < nb && c <= a) ? L[(size_t)(b0 + a) * n + b0 + c] : T(0); }
        __syncthreads();
        if (tid < 64) {
            T mine = tid < nb ? ss[b0 + tid] : T(0);
            for (uint32_t c = 0; c < nb; ++c) {
                const T zc = lane_value(mine, (int)c) / D[c][c];
                if (tid == c) mine = zc;
                else if (tid > c && tid < nb) mine -= D[tid][c] * zc;
            }
            if (tid < nb) { ss[b0 + tid] = mine; z[tid] = mine; }
        }
        __syncthreads();
        for (uint32_t i = b0 + kChB + tid; i < n; i += kIrlsThreads) {
            const T* row = L + (size_t)i * n + b0;
            T acc = T(0);
#pragma unroll 8
            for (uint32_t c = 0; c < kChB; ++c) acc += row[c] * z[c];
            ss[i] -= acc;
        }
        __syncthreads();
    }
    // L^T x = z, bottom to top
    const uint32_t nblk = (n + kChB - 1) / kChB;
    for (uint32_t bb = nblk; bb-- > 0;) {
        const uint32_t b0 = bb * kChB;
        const uint32_t nb = n - b0 < kChB ? n - b0 : kChB;
        for (uint32_t e = tid; e < kChB * kChB; e += kIrlsThreads) { const uint32_t a = e / kChB, c = e % kChB; D[a][c] = (a < nb && c <= a) ? L[(size_t)(b0 + a) * n + b0 + c] : T(0); }
        __syncthreads();
        if (tid < 64) {
            T mine = tid < nb ? ss[b0 + tid] : T(0);
            for (uint32_t cc = nb; cc-- > 0;) {
                const T xc = lane_value(mine, (int)cc) / D[cc][cc];
                if (tid == cc) mine = xc;
                else if (tid < cc) mine -= D[cc][tid] * xc;          // (L^T)[tid][cc] = L[cc][tid]
            }
            if (tid < nb) { ss[b0 + tid] = mine; z[tid] = mine; }
        }
        __syncthreads();
        for (uint32_t i = tid; i < b0; i += kIrlsThreads) {
            T acc = T(0);
            for (uint32_t c = 0; c < nb; ++c) acc += L[(size_t)(b0 + c) * n + i] * z[c];
            ss[i] -= acc;
        }
        __syncthreads();
    }
    for (uint32_t i = tid; i < n; i += kIrlsThreads) s[i] = ss[i];
}

// R x = x (upper triangular, by blocks of 32), then run_solver's part of the iteration (irls-cpu.cpp:100-116) and the while-test.
template <typename T>
__global__ __launch_bounds__(kIrlsThreads)
void k_irls_tail(const T* __restrict__ R, uint32_t n, T* __restrict__ vec, T tol, uint32_t max_iter, IrlsCtl<T>* __restrict__ ctl)
{
    if (ctl->done) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tl[];
    T* xs = reinterpret_cast<T*>(smem_tl);                       // [n]
    __shared__ T D[kChB][kChB + 1];
    __shared__ T z[kChB];
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    const uint32_t tid = threadIdx.x;
    T* xnext = vec + 2 * (size_t)n;
    T* w = vec + 3 * (size_t)n;
    T* x = vec + 4 * (size_t)n;
    const T p = T(0.9);
    for (uint32_t i = tid; i < n; i += kIrlsThreads) xs[i] = xnext[i];
    __syncthreads();
    const uint32_t nblk = (n + kChB - 1) / kChB;
    for (uint32_t bb = nblk; bb-- > 0;) {
        const uint32_t b0 = bb * kChB;
        const uint32_t nb = n - b0 < kChB ? n - b0 : kChB;
        for (uint32_t e = tid; e < kChB * kChB; e += kIrlsThreads) { const uint32_t a = e / kChB, c = e % kChB; D[a][c] = (a < nb && c < nb && c >= a) ? R[(size_t)(b0 + a) * n + b0 + c] : T(0); }
        __syncthreads();
        if (tid < 64) {
            T mine = tid < nb ? xs[b0 + tid] : T(0);
            for (uint32_t cc = nb; cc-- > 0;) {
                const T xc = lane_value(mine, (int)cc) / D[cc][cc];
                if (tid == cc) mine = xc;
                else if (tid < cc) mine -= D[tid][cc] * xc;
            }
            if (tid < nb) { xs[b0 + tid] = mine; z[tid] = mine; }
        }
        __syncthreads();
        for (uint32_t i = tid; i < b0; i += kIrlsThreads) {
            const T* row = R + (size_t)i * n + b0;
            T acc = T(0);
            for (uint32_t c = 0; c < nb; ++c) acc += row[c] * z[c];
            xs[i] -= acc;
        }
        __syncthreads();
    }
    for (uint32_t i = tid; i < n; i += kIrlsThreads) xnext[i] = xs[i];
    __syncthreads();
    // ---- run_solver ------------------------------------------------------------------------------
    T mx;
    uint32_t mi;
    block_max_excl(xnext, n, 0xffffffffu, mx, mi, sv, si);
    const T abstol = mx * tol;                                    // :100
    for (uint32_t i = tid; i < n; i += kIrlsThreads) {            // :103-104
        const T v = xnext[i] < abstol ? T(0) : xnext[i];
        xnext[i] = v;
        x[i] = v;
    }
    __syncthreads();
    block_max_excl(xnext, n, 0xffffffffu, mx, mi, sv, si);
    T second;
    if (n >= 2) {
        T m2;
        uint32_t i2;
        block_max_excl(xnext, n, mi, m2, i2, sv, si);
        second = m2;
    } else {
        second = mx;
    }
    T eps = ctl->eps;
    {
        const T cand = second / T(n);                             // :110
        if (cand < eps) eps = cand;
    }
    T part = T(0);
    for (uint32_t i = tid; i < n; i += kIrlsThreads) {            // :113
        const T v = (T)pow((double)(x[i] * x[i] + eps), (double)p / 2.0 - 1.0);
        w[i] = v;
        part += v;
    }
    const T sum = block_sum(part, sv);
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kIrlsThreads) w[i] /= sum;   // :114
    if (tid == 0) {
        const uint32_t it = ctl->iter + 1u;
        ctl->iter = it;
        ctl->eps = eps;
        ctl->abstol = abstol;
        ctl->second = second;
        if (!(it < max_iter && second > abstol)) ctl->done = 1;     // the do-while test (:117)
    }
}

// finally, normalise x (:121) and publish the report
template <typename T>
__global__ __launch_bounds__(kIrlsThreads)
void k_irls_finish(T* __restrict__ vec, uint32_t n, const IrlsCtl<T>* __restrict__ ctl, IrlsResult* __restrict__ res)
{
    __shared__ T sv[16];
    T* x = vec + 4 * (size_t)n;
    const uint32_t tid = threadIdx.x;
    T part = T(0);
    for (uint32_t i = tid; i < n; i += kIrlsThreads) part += x[i];
    const T sum = block_sum(part, sv);
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kIrlsThreads) x[i] /= sum;
    if (tid == 0) {
        res->iter = ctl->iter;
        res->spd_failure = ctl->spd_bad;
        res->solution_error = (double)ctl->eps;
    }
}

// ---- host side ---------------------------------------------------------------------------------------
template <typename T>
static IrlsState<T>* state_of(ss_hip_ctx* ctx) { return static_cast<IrlsState<T>*>(ctx->irls); }

template <typename T>
hipError_t irls_factor(ss_hip_ctx* ctx)
{
    const uint32_t m = (uint32_t)ctx->m, n = (uint32_t)ctx->n, ldm = ctx->ldm;
    auto* S = new IrlsState<T>();
    ctx->irls = S;
    hipError_t e;
#define IRLS_TRY(call) do { e = (call); if (e != hipSuccess) return e; } while (0)
    IRLS_TRY(hipMalloc(&S->Vt, (size_t)n * ldm * sizeof(T)));
    IRLS_TRY(hipMalloc(&S->Qt, (size_t)n * ldm * sizeof(T)));
    IRLS_TRY(hipMalloc(&S->R, (size_t)n * n * sizeof(T)));
    IRLS_TRY(hipMalloc(&S->G0, (size_t)n * n * sizeof(T)));
    IRLS_TRY(hipMalloc(&S->L, (size_t)n * n * sizeof(T)));
    IRLS_TRY(hipMalloc(&S->rdiag, (size_t)n * sizeof(T)));
    // (behind the vectors: one flag per column for the one-launch panel kernel of the factorisation, zeroed with them)
    const size_t vec_bytes = (5 * (size_t)n + 2 * (size_t)ldm) * sizeof(T);
    IRLS_TRY(hipMalloc(&S->vec, vec_bytes + (size_t)n * sizeof(uint32_t)));
    IRLS_TRY(hipMalloc(&S->res, sizeof(IrlsResult)));
    IRLS_TRY(hipMalloc(&S->ctl, sizeof(IrlsCtl<T>)));
    IRLS_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->done_host), 64, hipHostMallocDefault));
    IRLS_TRY(hipMemsetAsync(S->Qt, 0, (size_t)n * ldm * sizeof(T), ctx->stream));
    IRLS_TRY(hipMemsetAsync(S->G0, 0, (size_t)n * n * sizeof(T), ctx->stream));
    IRLS_TRY(hipMemsetAsync(S->vec, 0, vec_bytes + (size_t)n * sizeof(uint32_t), ctx->stream));
    T* At = static_cast<T*>(ctx->At);
    uint32_t* qr_ready = nullptr;                                 // (flags of the one-launch panel kernel: behind S->vec, zeroed above)
    if (ldm <= 8192u && !std::getenv("SS_HIP_IRLS_QR_GLOBAL")) {
        // register-resident form (round 4): a workgroup holds its column(s) for a whole panel / the whole back-accumulation
        // (SS_HIP_IRLS_QR_GLOBAL keeps the kernels that walk global memory: A/B aid)
        constexpr uint32_t NBQ = 32;
        uint32_t* const ready = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(S->vec) + vec_bytes);
        qr_ready = ready;
        const uint32_t rpt = ldm / 256u;
#define IRLS_QR_REG(RPT, NCA, NCQ)                                                                                                         \
        for (uint32_t k0 = 0; k0 < n; k0 += NBQ) {                                                                                         \
            const uint32_t nb = std::min<uint32_t>(NBQ, n - k0);                                                                           \
            hipLaunchKernelGGL((k_qr_panel_reg<T, RPT>), dim3(nb), dim3(kQrThreads), 0, ctx->stream, At, S->Vt, S->rdiag, ldm, m, k0, ready); \
            IRLS_TRY(hipGetLastError());                                                                                                   \
            if (k0 + nb < n) {                                                                                                             \
                const uint32_t nt = n - (k0 + nb);                                                                                         \
                hipLaunchKernelGGL((k_qr_apply_panel_reg<T, RPT, NCA>), dim3((nt + NCA - 1) / NCA), dim3(kQrThreads), 0, ctx->stream, At,     \
                                   (const T*)S->Vt, ldm, n, k0, nb);                                                                       \
                IRLS_TRY(hipGetLastError());                                                                                               \
            }                                                                                                                              \
        }                                                                                                                                  \
        hipLaunchKernelGGL((k_qr_formq_reg<T, RPT, NCQ>), dim3((n + NCQ - 1) / NCQ), dim3(kQrThreads), 0, ctx->stream, (const T*)S->Vt, S->Qt, ldm, n); \
        IRLS_TRY(hipGetLastError());
        if (rpt <= 4u) { IRLS_QR_REG(4, 2, 4) }
        else if (rpt <= 8u) { IRLS_QR_REG(8, 2, 4) }
        else if (rpt <= 16u) { IRLS_QR_REG(16, 2, 4) }
        else { IRLS_QR_REG(32, 1, 2) }
#undef IRLS_QR_REG
    } else {
        // By panels of 32 columns: the panel's own columns step by step (launches of <= 32 workgroups), then its 32 reflectors
        // applied to every trailing column in ONE launch; Q's columns are independent of each other: one launch for all of them.
        // Same arithmetic per column, in the same order: the same bits as the round-2 form.
        constexpr uint32_t NBQ = 32;
        // (one flag per column for the one-launch panel kernel)
        qr_ready = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(S->vec) + vec_bytes);
        uint32_t* const ready = qr_ready;
        for (uint32_t k0 = 0; k0 < n; k0 += NBQ) {
            const uint32_t nb = std::min<uint32_t>(NBQ, n - k0);
            hipLaunchKernelGGL((k_qr_panel<T>), dim3(nb), dim3(kQrThreads), 0, ctx->stream, At, S->Vt, S->rdiag, ldm, m, k0, ready);
            IRLS_TRY(hipGetLastError());
            if (k0 + nb < n) {
                hipLaunchKernelGGL((k_qr_apply_panel<T>), dim3(n - (k0 + nb)), dim3(kQrThreads), 0, ctx->stream, At, (const T*)S->Vt, ldm, m, k0, nb);
                IRLS_TRY(hipGetLastError());
            }
        }
        hipLaunchKernelGGL((k_qr_formq_all<T>), dim3(n), dim3(kQrThreads), 0, ctx->stream, (const T*)S->Vt, S->Qt, ldm, m);
        IRLS_TRY(hipGetLastError());
    }
    if (std::getenv("SS_HIP_IRLS_FUSED") || std::getenv("SS_HIP_IRLS_QR_GLOBAL")) {
        hipLaunchKernelGGL((k_irls_setup<T>), dim3(n), dim3(kQrThreads), 0, ctx->stream, (const T*)At, (const T*)S->Qt,
                           (const T*)S->rdiag, S->R, S->G0, ldm, m, n);
    } else {
        // R row by row; Q^T Q by 32 x 32 tiles on and below the diagonal (G0 was zeroed above: the tiles above it stay zero)
        hipLaunchKernelGGL((k_irls_setup_r<T>), dim3(n), dim3(kQrThreads), 0, ctx->stream, (const T*)At, (const T*)S->rdiag, S->R, ldm, n);
        const uint32_t tb = (n + 31u) / 32u;
        hipLaunchKernelGGL((k_irls_gram_tiled<T>), dim3(tb * (tb + 1u) / 2u), dim3(256), 0, ctx->stream, (const T*)S->Qt, S->G0, ldm, m, n);
    }
    IRLS_TRY(hipGetLastError());
    IRLS_TRY(hipStreamSynchronize(ctx->stream));
    return hipSuccess;
}

// y_dev -> the state's y buffer is the caller's job (irls_y_buffer); x is left in irls_x_buffer
template <typename T>
hipError_t irls_solve(ss_hip_ctx* ctx, T tol, uint32_t max_iter, IrlsResult* res_host)
{
    IrlsState<T>* S = state_of<T>(ctx);
    hipError_t e;
    const uint32_t n = (uint32_t)ctx->n, m = (uint32_t)ctx->m, ldm = ctx->ldm;
    hipStream_t st = ctx->stream;
    // (the solve kernels keep one vector of n elements in LDS)
    const size_t vec_lds = (size_t)n * sizeof(T);
    if (n < kIrlsBlockedMin || vec_lds > 96 * 1024 || std::getenv("SS_HIP_IRLS_FUSED")) {
        hipLaunchKernelGGL((k_irls_solve<T>), dim3(1), dim3(kIrlsThreads), 0, st, (const T*)S->Qt, (const T*)S->R,
                           (const T*)S->G0, S->L, S->vec, ldm, m, n, tol, max_iter, S->res);
        IRLS_TRY(hipGetLastError());
        IRLS_TRY(hipMemcpyAsync(res_host, S->res, sizeof(IrlsResult), hipMemcpyDeviceToHost, st));
        return hipSuccess;
    }
    static const bool attr_ok = [] {
        const bool a = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_irls_chol_solve<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
        const bool b = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_irls_tail<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
        if (!(a && b)) (void)hipGetLastError();
        return a && b;
    }();
    if (!attr_ok) return hipErrorInvalidConfiguration;
    IrlsCtl<T>* ctl = static_cast<IrlsCtl<T>*>(S->ctl);
    T* vec = S->vec;
    T* qTb = vec;
    T* s = vec + n;
    T* xnext = vec + 2 * (size_t)n;
    T* w = vec + 3 * (size_t)n;
    T* t = vec + 5 * (size_t)n;
    const T* y = t + ldm;
    hipLaunchKernelGGL((k_irls_init<T>), dim3((n + 255) / 256), dim3(256), 0, st, vec, n, ctl);
    hipLaunchKernelGGL((k_irls_qt_vec<T>), dim3((n + 3) / 4), dim3(256), 0, st, (const T*)S->Qt, ldm, m, n, y, qTb, (const IrlsCtl<T>*)nullptr);
    IRLS_TRY(hipGetLastError());
    for (uint32_t it = 0; it < max_iter; ++it) {
        hipLaunchKernelGGL((k_irls_scale<T>), dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, st, (const T*)S->G0, (const T*)w, S->L, n, (const IrlsCtl<T>*)ctl);
        for (uint32_t k0 = 0; k0 < n; k0 += kChB) {
            hipLaunchKernelGGL((k_chol_diag<T>), dim3(1), dim3(64), 0, st, S->L, n, k0, ctl);
            if (k0 + kChB < n) {
                const uint32_t below = n - (k0 + kChB);
                hipLaunchKernelGGL((k_chol_below<T>), dim3((below + 255) / 256), dim3(256), 0, st, S->L, n, k0, (const IrlsCtl<T>*)ctl);
                const uint32_t T_ = (below + kChB - 1) / kChB;
                hipLaunchKernelGGL((k_chol_trail<T>), dim3(T_ * (T_ + 1) / 2), dim3(256), 0, st, S->L, n, k0, (const IrlsCtl<T>*)ctl);
            }
        }
        hipLaunchKernelGGL((k_irls_chol_solve<T>), dim3(1), dim3(kIrlsThreads), vec_lds, st, (const T*)S->L, n, vec, ctl);
        hipLaunchKernelGGL((k_irls_q_vec<T>), dim3((m + 255) / 256), dim3(256), 0, st, (const T*)S->Qt, ldm, m, n, (const T*)s, t, (const IrlsCtl<T>*)ctl);
        hipLaunchKernelGGL((k_irls_qt_vec<T>), dim3((n + 3) / 4), dim3(256), 0, st, (const T*)S->Qt, ldm, m, n, (const T*)t, xnext, (const IrlsCtl<T>*)ctl);
        hipLaunchKernelGGL((k_irls_tail<T>), dim3(1), dim3(kIrlsThreads), vec_lds, st, (const T*)S->R, n, vec, tol, max_iter, ctl);
        IRLS_TRY(hipGetLastError());
        IRLS_TRY(hipMemcpyAsync(S->done_host, &ctl->done, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        IRLS_TRY(hipStreamSynchronize(st));
        if (*S->done_host != 0u) break;
    }
    hipLaunchKernelGGL((k_irls_finish<T>), dim3(1), dim3(kIrlsThreads), 0, st, vec, n, (const IrlsCtl<T>*)ctl, S->res);
    IRLS_TRY(hipGetLastError());
    IRLS_TRY(hipMemcpyAsync(res_host, S->res, sizeof(IrlsResult), hipMemcpyDeviceToHost, st));
    return hipSuccess;
#undef IRLS_TRY
}

template <typename T> T* irls_y_buffer(ss_hip_ctx* ctx) { return state_of<T>(ctx)->vec + 5 * ctx->n + ctx->ldm; }
template <typename T> T* irls_x_buffer(ss_hip_ctx* ctx) { return state_of<T>(ctx)->vec + 4 * ctx->n; }

template <typename T>
static void free_state(IrlsState<T>* S)
{
    if (!S) return;
    void* ptrs[] = { S->Vt, S->Qt, S->R, S->G0, S->L, S->rdiag, S->vec, S->res, S->ctl };
    if (S->done_host) (void)hipHostFree(S->done_host);
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete S;
}

void irls_free(ss_hip_ctx* ctx)
{
    if (!ctx->irls) return;
    if (ctx->is_f64) free_state(static_cast<IrlsState<double>*>(ctx->irls));
    else free_state(static_cast<IrlsState<float>*>(ctx->irls));
    ctx->irls = nullptr;
}

template hipError_t irls_factor<float>(ss_hip_ctx*);
template hipError_t irls_factor<double>(ss_hip_ctx*);
template hipError_t irls_solve<float>(ss_hip_ctx*, float, uint32_t, IrlsResult*);
template hipError_t irls_solve<double>(ss_hip_ctx*, double, uint32_t, IrlsResult*);
template float* irls_y_buffer<float>(ss_hip_ctx*);
template double* irls_y_buffer<double>(ss_hip_ctx*);
template float* irls_x_buffer<float>(ss_hip_ctx*);
template double* irls_x_buffer<double>(ss_hip_ctx*);

}  // namespace sship
