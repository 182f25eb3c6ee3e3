// subgram.hip — the Gram matrix of a 256-column subset, Gs = A_S^T A_S, bit for bit what the lookahead sweep
// (k_gemm32_tn_f32, gemm.hip) would put into the Gram-column cache for those columns.
//
// Why it exists: the speculative form of the lookahead engine (k_la_persist<true>, persist.hip) iterates on a
// subset of 256 columns and only ever needs Gram values BETWEEN subset columns: the support's rows restricted
// to the subset (c and q of the subset), u1 = G[idx][support] and G[idx][idx] (the inverse update,
// /root/reference/src/linalg/online_inverse.h:209-228).  Those 65 536 dot products are 8 MiB of A and 1 GFLOP —
// tens of microseconds — whereas the full Gram columns (all n entries of each: what the all-column verification of
// every breakpoint needs) are two passes over A.  With Gs at hand the iterations start right after c0 = A^T y and
// the passes over A run beside them on the other CUs instead of in front of them (homotopy.hip, early form).
//
// Bit for bit: the verification compares lambda, the step length and the pick of every breakpoint with the
// solo launch's BITWISE, and it forms c and q from the sweep's rows — so Gs must be the sweep's arithmetic
// exactly.  The sweep accumulates every output in one accumulator of v_mfma_f32_32x32x2_f32, which is a
// k-ordered chain of fp32 fmas (cdna_hip_programming.md §3, FP32-input MFMA), in this order of k:
//     for K-step kt (32 rows), for g in 0..3, for t in 0..3:  k = 32 kt + 8 g + t   then   k = 32 kt + 8 g + 4 + t
// (the instruction takes k-quad 2g from its lower and 2g + 1 from its upper half-wave).  The same chain is
// formed here with v_fma_f32, one output per lane; rows m .. ldm-1 are zero padding in both.
#include "ss_hip_internal.h"
#include "ss_hip_device.h"

namespace sship {

constexpr uint32_t kSgTile = 16;             // outputs per workgroup: 16 x 16, one per lane of 256
constexpr uint32_t kSgRows = 256;            // rows of A staged per step (a multiple of the sweep's K-step of 32)
constexpr uint32_t kSgPitch = kSgRows + 4;   // LDS row pitch in floats (16-byte aligned, conflict-free b128 reads)

typedef float sg_v4f __attribute__((ext_vector_type(4)));

// Gs[i][j] = a_{cols[i]} . a_{cols[j]} for i, j < 256 (cols entries >= n: zero row / column).  One workgroup per
// 16 x 16 tile on or above the diagonal, mirrored store (the chain of (i, j) and of (j, i) is the same chain of
// the same commutative products).  seed: if non-null, Gs[0][0] is also stored there (the first inverse update
// reads a_idx . a_idx from the entering column's cache row before the pass that fills that row has run).
__global__ __launch_bounds__(256)
void k_subset_gram(const float* __restrict__ At, uint32_t ldm, uint32_t n, const uint32_t* __restrict__ cols,
                   float* __restrict__ Gs, const DevState* __restrict__ st, float* __restrict__ seed_base,
                   const int32_t* __restrict__ slot_of, uint32_t gpitch)
{
    if (st != nullptr && st->done) return;
    __shared__ __attribute__((aligned(16))) float sA[kSgTile][kSgPitch];
    __shared__ __attribute__((aligned(16))) float sB[kSgTile][kSgPitch];
    // tile (bi <= bj) from the linear block index
    const uint32_t b = blockIdx.x;
    uint32_t t = (uint32_t)((__fsqrt_rn(8.f * (float)b + 1.f) - 1.f) * 0.5f);
    while (t * (t + 1u) / 2u > b) --t;
    while ((t + 1u) * (t + 2u) / 2u <= b) ++t;
    const uint32_t bj = t, bi = b - t * (t + 1u) / 2u;
    const uint32_t tid = threadIdx.x;
    const uint32_t ii = tid >> 4, jj = tid & 15u;
    // staging: thread -> (column c of the 16, float4 q of the 64 per 256-row step); 4 passes cover both tiles
    const uint32_t sc = tid >> 4, sq = tid & 15u;
    const uint32_t ca = cols[bi * kSgTile + sc], cb = cols[bj * kSgTile + sc];
    const float* ga = At + (size_t)(ca < n ? ca : 0u) * ldm;
    const float* gb = At + (size_t)(cb < n ? cb : 0u) * ldm;
    const sg_v4f zero4 = { 0.f, 0.f, 0.f, 0.f };
    float acc = 0.f;
    // the loads of step s + 1 are in flight under the fma chain of step s (each step is 2 x 16 KiB per workgroup)
    sg_v4f va[4], vb[4];
#define SG_LOAD(R0)                                                                             \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                             \
        const uint32_t q_ = sq + 16u * (uint32_t)p;          /* float4 index within the step: 0..63 */ \
        va[p] = ca < n ? *reinterpret_cast<const sg_v4f*>(ga + (R0) + 4u * q_) : zero4;         \
        vb[p] = cb < n ? *reinterpret_cast<const sg_v4f*>(gb + (R0) + 4u * q_) : zero4;         \
    }
    SG_LOAD(0u)
    for (uint32_t r0 = 0; r0 < ldm; r0 += kSgRows) {
        __syncthreads();                                                     // previous step fully consumed
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t q = sq + 16u * (uint32_t)p;
            *reinterpret_cast<sg_v4f*>(&sA[sc][4u * q]) = va[p];
            *reinterpret_cast<sg_v4f*>(&sB[sc][4u * q]) = vb[p];
        }
        __syncthreads();
        if (r0 + kSgRows < ldm) SG_LOAD(r0 + kSgRows)
        // the sweep's chain: groups of 8 consecutive rows in the order 0 4 1 5 2 6 3 7
#pragma unroll 4
        for (uint32_t k8 = 0; k8 < kSgRows; k8 += 8) {
            const sg_v4f a0 = *reinterpret_cast<const sg_v4f*>(&sA[ii][k8]), a1 = *reinterpret_cast<const sg_v4f*>(&sA[ii][k8 + 4]);
            const sg_v4f b0 = *reinterpret_cast<const sg_v4f*>(&sB[jj][k8]), b1 = *reinterpret_cast<const sg_v4f*>(&sB[jj][k8 + 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc = __builtin_fmaf(a0[e], b0[e], acc);
                acc = __builtin_fmaf(a1[e], b1[e], acc);
            }
        }
    }
#undef SG_LOAD
    const uint32_t gi = bi * kSgTile + ii, gj = bj * kSgTile + jj;
    Gs[(size_t)gi * kSoloWidth + gj] = acc;
    if (bi != bj) Gs[(size_t)gj * kSoloWidth + gi] = acc;
    if (seed_base != nullptr && gi == 0u && gj == 0u) {
        // a_idx . a_idx where the first inverse update (k_gramupd, round 0) looks for it: row slot_of[idx], column idx
        const uint32_t idx = cols[0];
        const int32_t slot = idx < n ? slot_of[idx] : -1;
        if (slot >= 0) seed_base[(size_t)slot * gpitch + idx] = acc;
    }
}

// The pass's outputs for a RANGE of dictionary columns, by the same VALU chain: D[drows[s]][c] = a_c . a_{rcols[s]}
// for c0 <= c < c0 + 16 * (gridDim.x / 2), s < 32 (entries 0xffffffff of the lists are skipped; an empty list —
// first entry 0xffffffff — returns at once, like the pass).  One workgroup per 16 x 16 outputs, 8.7 KB of LDS.  Used by
// the early form for the last two tiles of each pass (homotopy.hip, early_prologue): with one CU held by the solo
// workgroup 510 tiles of 128 columns are what the other 255 can carry two each.
constexpr uint32_t kCgRows = 64;             // rows of A staged per step
constexpr uint32_t kCgPitch = kCgRows + 4;
__global__ __launch_bounds__(256)
void k_cols_gram(const float* __restrict__ At, uint32_t ldm, uint32_t c0, const uint32_t* __restrict__ rcols,
                 const uint32_t* __restrict__ drows, float* __restrict__ D, uint32_t ldd)
{
    if (rcols[0] == 0xffffffffu) return;
    __shared__ __attribute__((aligned(16))) float sA[kSgTile][kCgPitch];
    __shared__ __attribute__((aligned(16))) float sB[kSgTile][kCgPitch];
    const uint32_t bi = blockIdx.x & 1u, bj = blockIdx.x >> 1;
    const uint32_t tid = threadIdx.x;
    const uint32_t ii = tid >> 4, jj = tid & 15u;
    const uint32_t sc = tid >> 4, sq = tid & 15u;            // staging: column sc of the 16, float4 sq of the 16 per step
    const uint32_t ca = rcols[bi * kSgTile + sc];
    const bool va_ok = ca != 0xffffffffu;
    const float* ga = At + (size_t)(va_ok ? ca : 0u) * ldm + 4u * sq;
    const float* gb = At + (size_t)(c0 + bj * kSgTile + sc) * ldm + 4u * sq;
    const sg_v4f zero4 = { 0.f, 0.f, 0.f, 0.f };
    float acc = 0.f;
    // register ring three steps deep (a step is 4 KiB per operand and 64 fmas of chain: the loads need the lead)
    sg_v4f va[3], vb[3];
#define CG_LOAD(SET, R0)                                                                        \
    {                                                                                           \
        va[SET] = va_ok ? *reinterpret_cast<const sg_v4f*>(ga + (R0)) : zero4;                  \
        vb[SET] = __builtin_nontemporal_load(reinterpret_cast<const sg_v4f*>(gb + (R0)));       \
    }
#define CG_STEP(SET, R0)                                                                        \
    {                                                                                           \
        __syncthreads();                                                                        \
        *reinterpret_cast<sg_v4f*>(&sA[sc][4u * sq]) = va[SET];                                 \
        *reinterpret_cast<sg_v4f*>(&sB[sc][4u * sq]) = vb[SET];                                 \
        __syncthreads();                                                                        \
        if ((R0) + 3u * kCgRows < ldm) CG_LOAD(SET, (R0) + 3u * kCgRows)                         \
        /* the pass's chain (a = right-hand side s, b = dictionary column c, as in the MFMA) */ \
        _Pragma("unroll") for (uint32_t k8 = 0; k8 < kCgRows; k8 += 8) {                        \
            const sg_v4f a0 = *reinterpret_cast<const sg_v4f*>(&sA[ii][k8]), a1 = *reinterpret_cast<const sg_v4f*>(&sA[ii][k8 + 4]); \
            const sg_v4f b0 = *reinterpret_cast<const sg_v4f*>(&sB[jj][k8]), b1 = *reinterpret_cast<const sg_v4f*>(&sB[jj][k8 + 4]); \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                     \
                acc = __builtin_fmaf(a0[e], b0[e], acc);                                        \
                acc = __builtin_fmaf(a1[e], b1[e], acc);                                        \
            }                                                                                   \
        }                                                                                       \
    }
    CG_LOAD(0, 0u)
    if (kCgRows < ldm) CG_LOAD(1, kCgRows)
    if (2u * kCgRows < ldm) CG_LOAD(2, 2u * kCgRows)
    uint32_t r0 = 0;
    for (; r0 + 3u * kCgRows <= ldm; r0 += 3u * kCgRows) {
        CG_STEP(0, r0)
        CG_STEP(1, r0 + kCgRows)
        CG_STEP(2, r0 + 2u * kCgRows)
    }
    if (r0 < ldm) { CG_STEP(0, r0) r0 += kCgRows; }
    if (r0 < ldm) { CG_STEP(1, r0) r0 += kCgRows; }
#undef CG_LOAD
#undef CG_STEP
    const uint32_t dr = drows[bi * kSgTile + ii];
    if (dr != 0xffffffffu) D[(size_t)dr * ldd + c0 + bj * kSgTile + jj] = acc;
}

// columns c0 .. c0 + ncols - 1 (both multiples of 16) of the 32 Gram columns of a pass, on the given stream
hipError_t launch_cols_gram_on(const ss_hip_ctx* ctx, hipStream_t on, uint32_t c0, uint32_t ncols, const uint32_t* rcols,
                               const uint32_t* drows, float* D, uint32_t ldd)
{
    if (ctx->ldm % kCgRows != 0 || c0 % kSgTile != 0 || ncols % kSgTile != 0 || c0 + ncols > ctx->n_pad) return hipErrorInvalidValue;
    if (ncols == 0) return hipSuccess;
    hipLaunchKernelGGL(k_cols_gram, dim3(2u * (ncols / kSgTile)), dim3(256), 0, on, static_cast<const float*>(ctx->At), ctx->ldm, c0,
                       rcols, drows, D, ldd);
    return hipGetLastError();
}

hipError_t launch_subset_gram_f32(const ss_hip_ctx* ctx, const uint32_t* cols_dev, float* Gs, const DevState* st,
                                  float* seed_base, const int32_t* slot_of, uint32_t gpitch)
{
    if (ctx->ldm % kSgRows != 0) return hipErrorInvalidValue;
    constexpr uint32_t T = kSoloWidth / kSgTile;                // 16 tiles per side
    hipLaunchKernelGGL(k_subset_gram, dim3(T * (T + 1) / 2), dim3(256), 0, ctx->stream, static_cast<const float*>(ctx->At),
                       ctx->ldm, (uint32_t)ctx->n, cols_dev, Gs, st, seed_base, slot_of, gpitch);
    return hipGetLastError();
}

}  // namespace sship
