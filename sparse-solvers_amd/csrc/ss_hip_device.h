// Device-side helpers shared by the active-set kernels (activeset.hip, persist.hip):
// ordered (value, index) reductions, sums, and the in-launch hand-off primitives.
#pragma once

#include <cfloat>

#include "ss_hip_internal.h"

namespace sship {

template <typename T> struct Lim;
template <> struct Lim<float>  { static constexpr float  max() { return FLT_MAX; } static constexpr float  tiny() { return FLT_MIN; } static constexpr float  eps() { return FLT_EPSILON; } };
template <> struct Lim<double> { static constexpr double max() { return DBL_MAX; } static constexpr double tiny() { return DBL_MIN; } static constexpr double eps() { return DBL_EPSILON; } };

typedef float  v4f __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kSmallThreads = 256;
constexpr int kUpdThreads = 1024;

// order-preserving map float -> u32 (smaller float, smaller key)
__device__ __forceinline__ uint32_t ordered_key(float v)
{
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ---- block reductions --------------------------------------------------------------

// "better" for the arg-max of |c| (ixamax): larger value, ties -> smaller index
template <typename T>
__device__ __forceinline__ bool better_max(T v, uint32_t i, T bv, uint32_t bi)
{
    return v > bv || (v == bv && i < bi);
}
// "better" for the step length: smaller value, ties -> smaller (left-most) index
template <typename T>
__device__ __forceinline__ bool better_min(T v, uint32_t i, T bv, uint32_t bi)
{
    return v < bv || (v == bv && i < bi);
}

// ---- wave-level data movement on the DPP path (one VALU op each; __shfl_xor goes through the
// ---- LDS crossbar, ~100 cycles per step) ---------------------------------------------------------
// Within a row of 16 lanes: xor 1, xor 2 (quad permutes), then mirror within 8 and within 16 —
// applied to values that are already uniform per quad / per 8 these complete the butterfly.
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // row_half_mirror
constexpr int kDppMirror = 0x140;       // row_mirror

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) { return __uint_as_float(dpp_mov<CTRL>(__float_as_uint(v))); }
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = dpp_mov<CTRL>((uint32_t)b), hi = dpp_mov<CTRL>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float lane_value(float v, int l) { return __uint_as_float(lane_value(__float_as_uint(v), l)); }
__device__ __forceinline__ double lane_value(double v, int l)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = lane_value((uint32_t)b, l), hi = lane_value((uint32_t)(b >> 32), l);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// every lane receives the best (value, index) pair of the wave
template <typename T, bool MAX>
__device__ __forceinline__ void wave_reduce_pair(T& v, uint32_t& i)
{
#define SSHIP_PAIR_STEP(CTRL)                                                                   \
    {                                                                                           \
        const T ov = dpp_mov<CTRL>(v);                                                          \
        const uint32_t oi = dpp_mov<CTRL>(i);                                                   \
        const bool take = MAX ? better_max(ov, oi, v, i) : better_min(ov, oi, v, i);            \
        if (take) { v = ov; i = oi; }                                                           \
    }
    SSHIP_PAIR_STEP(kDppXor1)
    SSHIP_PAIR_STEP(kDppXor2)
    SSHIP_PAIR_STEP(kDppHalfMirror)
    SSHIP_PAIR_STEP(kDppMirror)
#undef SSHIP_PAIR_STEP
    // the four rows (the order of a max / min with its index tie-break does not matter)
    T bv = lane_value(v, 0);
    uint32_t bi = lane_value(i, 0);
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const T ov = lane_value(v, 16 * r);
        const uint32_t oi = lane_value(i, 16 * r);
        const bool take = MAX ? better_max(ov, oi, bv, bi) : better_min(ov, oi, bv, bi);
        if (take) { bv = ov; bi = oi; }
    }
    v = bv;
    i = bi;
}

// all threads of the block receive the reduced pair; sv/si: LDS scratch of >= 16 entries
template <typename T, bool MAX>
__device__ __forceinline__ void block_reduce_pair(T& v, uint32_t& i, T* sv, uint32_t* si)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    wave_reduce_pair<T, MAX>(v, i);
    __syncthreads();
    if (lane == 0) { sv[wave] = v; si[wave] = i; }
    __syncthreads();
    T bv = sv[0];
    uint32_t bi = si[0];
    for (int w = 1; w < nw; ++w) {
        const T ov = sv[w];
        const uint32_t oi = si[w];
        const bool take = MAX ? better_max(ov, oi, bv, bi) : better_min(ov, oi, bv, bi);
        if (take) { bv = ov; bi = oi; }
    }
    v = bv;
    i = bi;
}

// Sum over the wave, the same value in every lane.  Fixed association: butterfly inside each row
// of 16 lanes (1, 2, 4, 8 apart), then ((row0 + row1) + row2) + row3.
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppHalfMirror>(v);
    v += dpp_mov<kDppMirror>(v);
    return ((lane_value(v, 0) + lane_value(v, 16)) + lane_value(v, 32)) + lane_value(v, 48);
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* sv)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sv[wave] = v;
    __syncthreads();
    T s = sv[0];
    for (int w = 1; w < nw; ++w) s += sv[w];
    return s;
}

// max |c| and its first index from the per-workgroup partials of the sweep
template <typename T>
__device__ __forceinline__ void reduce_sweep_partials(const T* pv, const uint32_t* pi, uint32_t nb,
                                                      T& val, uint32_t& idx, T* sv, uint32_t* si)
{
    T v = T(-1);
    uint32_t ix = 0xffffffffu;
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
        const T ov = pv[b];
        const uint32_t oi = pi[b];
        if (better_max(ov, oi, v, ix)) { v = ov; ix = oi; }
    }
    block_reduce_pair<T, true>(v, ix, sv, si);
    if (ix == 0xffffffffu) ix = 0;   // all-NaN correlations: keep every later index in range
    val = v;
    idx = ix;
}

// the same over partials that other workgroups of THIS launch stored with agent-scope atomic stores
template <typename T>
__device__ __forceinline__ void reduce_partials_agent(const T* pv, const uint32_t* pi, uint32_t nb,
                                                      T& val, uint32_t& idx, T* sv, uint32_t* si)
{
    T v = T(-1);
    uint32_t ix = 0xffffffffu;
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
        const T ov = __hip_atomic_load(&pv[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t oi = __hip_atomic_load(&pi[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (better_max(ov, oi, v, ix)) { v = ov; ix = oi; }
    }
    block_reduce_pair<T, true>(v, ix, sv, si);
    if (ix == 0xffffffffu) ix = 0;
    val = v;
    idx = ix;
}

template <typename T>
__device__ __forceinline__ T sign_tol(T v, T tol)   // homotopy-cpu.cpp:59-67
{
    if (v > tol) return T(1);
    if (v < -tol) return T(-1);
    return T(0);
}

// Is lambda = ||c||_inf where the previous step left it?  In exact arithmetic lambda_now = lambda_prev - gamma_prev; a
// candidate that is exactly 0 while that holds within rounding (256 eps of the solve's first lambda: the absolute
// error of a Gram-form correlation is ~eps * ||A^T y||_inf * sqrt(K)) is a TIE decided by rounding.  When lambda jumped
// instead, an off-support column dominates by a margin — every summation order sees that, nothing to arbitrate.
template <typename T>
__device__ __forceinline__ bool tie_band(T c_inf_now, T lambda_prev, T gamma_prev, T lambda0)
{
    const T expect = lambda_prev - gamma_prev;
    const T diff = c_inf_now > expect ? c_inf_now - expect : expect - c_inf_now;
    return diff <= T(256) * Lim<T>::eps() * lambda0;
}

// dot product of two contiguous device rows of length len (multiple of 256) by one block
template <typename T>
__device__ __forceinline__ T block_dot(const T* a, const T* b, uint32_t len, T* sv)
{
    T acc = T(0);
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) acc += a[i] * b[i];
    return block_sum(acc, sv);
}

// ---- in-launch hand-off: "last workgroup to arrive finishes the job" --------------------
// Placement-independent release/acquire at agent scope (cdna_hip_programming.md §6
// Guideline 16, counter form): every wave drains its stores, the workgroup's leader
// releases and takes a ticket; the workgroup that draws the last ticket acquires and may
// then read, with VECTOR loads, what the others stored.  Returns true in that workgroup.
// The counter is reset by the last arriver (every other workgroup has already arrived).
__device__ __forceinline__ bool arrive_last(uint32_t* counter, uint32_t total, uint32_t* s_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t last = (t == total - 1u) ? 1u : 0u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0u;
}

// The same ticket without cache fences, for hand-offs whose payload is itself moved with agent-scope
// atomic (L2-bypassing) stores and loads: every wave drains its stores, the leader takes a ticket.
// (A release fence writes back the whole L2 and an acquire invalidates it: with thousands of
// workgroups per launch — the batched kernels — that alone costs milliseconds.)
__device__ __forceinline__ bool arrive_last_relaxed(uint32_t* counter, uint32_t total, uint32_t* s_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t last = (t == total - 1u) ? 1u : 0u;
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0u;
}

// a scalar another workgroup stored in this launch: read it on the vector path, L1 bypassed
__device__ __forceinline__ double load_handoff(const double* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a slot finished: count it; the solve is over when every slot has (host sees hflags[1])
__device__ __forceinline__ void signal_done(uint32_t* hflags, uint32_t* ndone, uint32_t nslots, uint32_t round)
{
    uint32_t prev = nslots - 1u;
    if (ndone != nullptr) prev = __hip_atomic_fetch_add(ndone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (hflags != nullptr) {
        if (prev + 1u >= nslots) __hip_atomic_store(&hflags[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&hflags[0], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// one more iteration launch of the fused lookahead engine has executed (host pump, hflags[0])
__device__ __forceinline__ void bump_seq(DevState* st, uint32_t* hflags)
{
    const uint32_t sq = st->seq + 1u;
    st->seq = sq;
    __hip_atomic_store(&hflags[0], sq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace sship
