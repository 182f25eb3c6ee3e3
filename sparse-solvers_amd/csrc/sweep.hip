// sweep.hip — the correlation sweep  [c, q] = A^T [r, p]  for gfx950 (MI355X).
//
// Replaces the two CblasTrans GEMVs of every Homotopy iteration
// (/root/reference/src/solvers/homotopy-cpu.cpp:97 `c = A^T (y - A x)` and :120
// `q = A^T (A d)`) with ONE pass over the device copy of A: both right-hand sides
// are applied to each 16-byte piece of A while it is in registers, so A is read from
// HBM once per iteration instead of the reference's four times.
//
// Layout: At[n_pad][ldm], dictionary column j contiguous (ldm a multiple of 256
// elements, zero padded).  A wave owns CPW columns and streams them with 16-byte
// loads (64 lanes x 16 B = 1 KiB per wave-instruction, fully coalesced); the
// right-hand sides live in LDS (2 x m x sizeof(T), 64 KiB at m = 8192 fp32) and are
// re-read from there once per CPW columns.  Per lane the partial sums run over the
// rows i = lane*V + 64*V*t in ascending t; the 64 lane sums are combined by a
// butterfly (xor 32,16,8,4,2,1).  That fixed order makes the result independent of
// grid size and kernel variant.
//
// Roofline: HBM.  Algorithmic bytes per launch = m*n*s (A) + nrhs*m*s (rhs) +
// nrhs*n*s (out); 2 flop per element per right-hand side => 0.5-1 flop/B, far below
// the ~20 flop/B ridge, so no MFMA here (MI355X_MICROARCH.md §HBM).
//
// The epilogue also produces, per workgroup, max |c| and its first index: the
// ixamax of inf_norm (homotopy-cpu.cpp:32-37), finished by the next kernel.
#include "ss_hip_internal.h"

namespace sship {

typedef float  v4f __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <typename T> struct VecOf;
template <> struct VecOf<float>  { using type = v4f; static constexpr int N = 4; };
template <> struct VecOf<double> { using type = v2d; static constexpr int N = 2; };

__device__ __forceinline__ float  fma_t(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Streams one chunk of CPW columns against the right-hand sides in LDS.
// DEPTH-stage register ring: the loads of step t+DEPTH-1 are issued before the FMAs of step
// t, so each wave keeps (DEPTH-1)*CPW .. DEPTH*CPW 16-byte loads (1 KiB each per wave) in
// flight.  The step counter is wave-uniform (rows is a multiple of 64*VN), so guards are
// scalar branches and the steady-state loop has none.
template <typename T, int NRHS, int CPW, int DEPTH, bool NTL>
__device__ __forceinline__ void stream_columns(const char* const (&cb)[CPW], const char* lds_b, uint32_t mc,
                                               uint32_t nsteps, uint32_t lane_b, T (&acc)[CPW][NRHS])
{
    using V = typename VecOf<T>::type;
    constexpr int VN = VecOf<T>::N;
    constexpr uint32_t step_b = 64 * 16;                 // bytes per wave-step per column
    V a[DEPTH][CPW];

#define SS_LOAD(STAGE, TSTEP)                                                                 \
    _Pragma("unroll") for (int c = 0; c < CPW; ++c) {                                         \
        const V* p_ = reinterpret_cast<const V*>(cb[c] + (lane_b + (TSTEP) * step_b));        \
        a[STAGE][c] = NTL ? __builtin_nontemporal_load(p_) : *p_;                             \
    }
#define SS_COMPUTE(STAGE, TSTEP)                                                              \
    {                                                                                         \
        V rv_[NRHS];                                                                          \
        _Pragma("unroll") for (int k = 0; k < NRHS; ++k) rv_[k] = *reinterpret_cast<const V*>( \
            lds_b + ((uint32_t)k * mc * (uint32_t)sizeof(T) + lane_b + (TSTEP) * step_b));    \
        _Pragma("unroll") for (int c = 0; c < CPW; ++c)                                       \
        _Pragma("unroll") for (int k = 0; k < NRHS; ++k)                                      \
        _Pragma("unroll") for (int e = 0; e < VN; ++e)                                        \
            acc[c][k] = fma_t(a[STAGE][c][e], rv_[k][e], acc[c][k]);                          \
    }

#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s)
        if ((uint32_t)s < nsteps) { SS_LOAD(s, (uint32_t)s) }

    uint32_t t = 0;
    for (; t + (2 * DEPTH - 1) <= nsteps; t += DEPTH) {      // steady state: no guards
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            SS_LOAD((s + DEPTH - 1) % DEPTH, t + (uint32_t)(s + DEPTH - 1))
            SS_COMPUTE(s, t + (uint32_t)s)
        }
    }
    for (; t < nsteps; t += DEPTH) {                          // drain
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            if (t + (uint32_t)(s + DEPTH - 1) < nsteps) {
                SS_LOAD((s + DEPTH - 1) % DEPTH, t + (uint32_t)(s + DEPTH - 1))
            }
            if (t + (uint32_t)s < nsteps) { SS_COMPUTE(s, t + (uint32_t)s) }
        }
    }
#undef SS_LOAD
#undef SS_COMPUTE
}

template <typename T, int NRHS, int CPW, int WAVES, bool NT, int DEPTH, int BPC>
__global__ __launch_bounds__(WAVES * 64, (BPC * WAVES) / 4)
void k_sweep(const T* __restrict__ At, uint32_t ldm, uint32_t n, uint32_t ngroups, uint32_t mc,
             const T* __restrict__ rhs, size_t rhs_stride, T* __restrict__ out0, T* __restrict__ out1,
             T* __restrict__ pmax_val, uint32_t* __restrict__ pmax_idx, const DevState* st,
             uint32_t temporal_groups)
{
    using V = typename VecOf<T>::type;
    constexpr int VN = VecOf<T>::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* lds = reinterpret_cast<T*>(smem);                      // [NRHS][mc]

    if (st != nullptr && st->done != 0) return;               // uniform: solve already finished

    const uint32_t lane = threadIdx.x & 63u;
    // wave index as a scalar so that column base addresses live in SGPRs
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nchunks = (ldm + mc - 1) / mc;

    T best = T(-1);
    uint32_t best_idx = 0xffffffffu;
    bool lds_valid = false;

    for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const uint32_t col0 = g * (WAVES * CPW) + wave * CPW;
        T acc[CPW][NRHS];
#pragma unroll
        for (int c = 0; c < CPW; ++c)
#pragma unroll
            for (int k = 0; k < NRHS; ++k) acc[c][k] = T(0);

        for (uint32_t ch = 0; ch < nchunks; ++ch) {
            const uint32_t r0 = ch * mc;
            const uint32_t rows = (ldm - r0 < mc) ? (ldm - r0) : mc;
            if (nchunks > 1 || !lds_valid) {
                if (lds_valid) __syncthreads();                // previous chunk fully consumed
#pragma unroll
                for (int k = 0; k < NRHS; ++k)
                    for (uint32_t i = threadIdx.x * VN; i < rows; i += WAVES * 64 * VN)
                        *reinterpret_cast<V*>(&lds[k * mc + i]) =
                            *reinterpret_cast<const V*>(&rhs[(size_t)k * rhs_stride + r0 + i]);
                __syncthreads();
                lds_valid = true;
            }

            // Scalar (SGPR) base address per column + ONE per-lane 32-bit byte offset, so
            // every load is `global_load_dwordx4 v, v_off, s[base] offset:imm`.
            const char* cb[CPW];
#pragma unroll
            for (int c = 0; c < CPW; ++c)
                cb[c] = reinterpret_cast<const char*>(At + (size_t)(col0 + c) * ldm + r0);
            const char* lds_b = reinterpret_cast<const char*>(lds);

            const uint32_t nsteps = rows / (64 * VN);
            const uint32_t lane_b = lane * 16;
            // leading `temporal_groups` column groups are read with ordinary (cache-allocating)
            // loads, the rest with the non-temporal hint (wave-uniform choice)
            if (NT && g >= temporal_groups)
                stream_columns<T, NRHS, CPW, DEPTH, true>(cb, lds_b, mc, nsteps, lane_b, acc);
            else
                stream_columns<T, NRHS, CPW, DEPTH, false>(cb, lds_b, mc, nsteps, lane_b, acc);
        }

#pragma unroll
        for (int c = 0; c < CPW; ++c)
#pragma unroll
            for (int k = 0; k < NRHS; ++k) acc[c][k] = wave_sum(acc[c][k]);

#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const uint32_t col = col0 + c;
            if (col < n) {
                if (lane == 0) {
                    out0[col] = acc[c][0];
                    if (NRHS > 1) out1[col] = acc[c][NRHS - 1];
                }
                const T a = acc[c][0] < T(0) ? -acc[c][0] : acc[c][0];
                if (a > best) { best = a; best_idx = col; }   // ascending cols: first max kept
            }
        }
    }

    if (pmax_val == nullptr) return;
    // workgroup reduction of (max |c|, first index)
    __syncthreads();
    T* sval = reinterpret_cast<T*>(smem);
    uint32_t* sidx = reinterpret_cast<uint32_t*>(smem + WAVES * sizeof(T));
    if (lane == 0) { sval[wave] = best; sidx[wave] = best_idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        T bv = sval[0];
        uint32_t bi = sidx[0];
        for (int w = 1; w < WAVES; ++w) {
            const T v = sval[w];
            const uint32_t ix = sidx[w];
            if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
        }
        pmax_val[blockIdx.x] = bv;
        pmax_idx[blockIdx.x] = bi;
    }
}

struct Variant { int waves, cpw, nt, depth, blocks_per_cu; };
// variant table (ss_hip_set_option "sweep_variant")
static const Variant kVariants[] = {
    // waves, cols/wave, nt, depth, blocks/CU
    { 8, 4, 1, 2, 2 },   // 0 default
    { 8, 4, 0, 2, 2 },   // 1 plain (temporal) loads
    { 8, 4, 1, 3, 2 },   // 2
    { 8, 4, 1, 4, 1 },   // 3
    { 4, 4, 1, 3, 2 },   // 4
    { 16, 4, 1, 2, 1 },  // 5
    { 16, 2, 1, 3, 1 },  // 6
    { 8, 2, 1, 4, 2 },   // 7
    { 4, 2, 1, 4, 2 },   // 8
    { 8, 4, 1, 2, 1 },   // 9
    { 4, 4, 1, 2, 2 },   // 10
    { 4, 4, 1, 2, 4 },   // 11 (needs LDS <= 40 KiB per block to reach 4 blocks/CU)
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
constexpr size_t kLdsBudget = 65536;

size_t sweep_max_lds_bytes() { return kLdsBudget; }

template <typename T, int NRHS, int CPW, int WAVES, bool NT, int DEPTH, int BPC>
static hipError_t launch_one(const ss_hip_ctx* ctx, const Variant& v, const T* rhs, size_t rhs_stride, T* out0, T* out1,
                             T* pmax_val, uint32_t* pmax_idx, uint32_t* nblocks_out,
                             const DevState* st)
{
    const uint32_t ldm = ctx->ldm;
    const uint32_t ngroups = ctx->n_pad / (WAVES * CPW);
    uint32_t mc = (uint32_t)(kLdsBudget / (NRHS * sizeof(T)));
    mc -= mc % kRowPad;
    if (mc > ldm) mc = ldm;
    const size_t lds_bytes = (size_t)NRHS * mc * sizeof(T);
    uint32_t grid = (uint32_t)ctx->num_cus * (uint32_t)v.blocks_per_cu;
    if (grid > ngroups) grid = ngroups;
    if (grid > kMaxSweepBlocks) grid = kMaxSweepBlocks;
    if (nblocks_out) *nblocks_out = grid;
    hipLaunchKernelGGL((k_sweep<T, NRHS, CPW, WAVES, NT, DEPTH, BPC>), dim3(grid), dim3(WAVES * 64), lds_bytes,
                       ctx->stream, static_cast<const T*>(ctx->At), ldm, (uint32_t)ctx->n, ngroups, mc,
                       rhs, rhs_stride, out0, out1, pmax_val, pmax_idx, st,
                       (uint32_t)(ctx->temporal_cols / (WAVES * CPW)));
    return hipGetLastError();
}

template <typename T, int NRHS>
static hipError_t dispatch(const ss_hip_ctx* ctx, const T* rhs, size_t rhs_stride, T* out0, T* out1, T* pmax_val,
                           uint32_t* pmax_idx, uint32_t* nblocks_out, const DevState* st)
{
    int vi = ctx->sweep_variant;
    if (vi < 0 || vi >= kNumVariants) vi = 0;
    const Variant& v = kVariants[vi];
#define SS_CASE(W, C, N, D, B)                                                               \
    if (v.waves == W && v.cpw == C && v.nt == N && v.depth == D && v.blocks_per_cu == B)     \
        return launch_one<T, NRHS, C, W, (N != 0), D, B>(ctx, v, rhs, rhs_stride, out0, out1, pmax_val,  \
                                                         pmax_idx, nblocks_out, st);
    SS_CASE(8, 4, 1, 2, 2)
    SS_CASE(8, 4, 0, 2, 2)
    SS_CASE(8, 4, 1, 3, 2)
    SS_CASE(8, 4, 1, 4, 1)
    SS_CASE(4, 4, 1, 3, 2)
    SS_CASE(16, 4, 1, 2, 1)
    SS_CASE(16, 2, 1, 3, 1)
    SS_CASE(8, 2, 1, 4, 2)
    SS_CASE(4, 2, 1, 4, 2)
    SS_CASE(8, 4, 1, 2, 1)
    SS_CASE(4, 4, 1, 2, 2)
    SS_CASE(4, 4, 1, 2, 4)
#undef SS_CASE
    return hipErrorInvalidValue;
}

template <typename T>
hipError_t launch_sweep(const ss_hip_ctx* ctx, const T* rhs, size_t rhs_stride, int nrhs, T* out0, T* out1,
                        T* pmax_val, uint32_t* pmax_idx, uint32_t* nblocks_out, const DevState* st)
{
    if (nrhs == 2) return dispatch<T, 2>(ctx, rhs, rhs_stride, out0, out1, pmax_val, pmax_idx, nblocks_out, st);
    return dispatch<T, 1>(ctx, rhs, rhs_stride, out0, nullptr, pmax_val, pmax_idx, nblocks_out, st);
}

template hipError_t launch_sweep<float>(const ss_hip_ctx*, const float*, size_t, int, float*, float*, float*,
                                        uint32_t*, uint32_t*, const DevState*);
template hipError_t launch_sweep<double>(const ss_hip_ctx*, const double*, size_t, int, double*, double*,
                                         double*, uint32_t*, uint32_t*, const DevState*);

}  // namespace sship
