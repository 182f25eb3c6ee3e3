// Device-side helpers shared by the active-set kernels (activeset.hip, persist.hip):
// ordered (value, index) reductions, sums, and the in-launch hand-off primitives.
#pragma once

#include <cfloat>

#include "ss_hip_internal.h"

namespace sship {

template <typename T> struct Lim;
template <> struct Lim<float>  { static constexpr float  max() { return FLT_MAX; } static constexpr float  tiny() { return FLT_MIN; } };
template <> struct Lim<double> { static constexpr double max() { return DBL_MAX; } static constexpr double tiny() { return DBL_MIN; } };

typedef float  v4f __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kSmallThreads = 256;
constexpr int kUpdThreads = 1024;

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

template <typename T, bool MAX>
__device__ __forceinline__ void wave_reduce_pair(T& v, uint32_t& i)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const T ov = __shfl_xor(v, off, 64);
        const uint32_t oi = __shfl_xor(i, off, 64);
        const bool take = MAX ? better_max(ov, oi, v, i) : better_min(ov, oi, v, i);
        if (take) { v = ov; i = oi; }
    }
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

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
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

template <typename T>
__device__ __forceinline__ T sign_tol(T v, T tol)   // homotopy-cpu.cpp:59-67
{
    if (v > tol) return T(1);
    if (v < -tol) return T(-1);
    return T(0);
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
