// screen.hip — the SCREENED form of one fp32 signal: the subset form of subbatch.hip without G = A^T A.
//
// The default single-signal engine reads A three times per solve: c0 = A^T y, then two passes that form the Gram columns
// of the support over ALL n columns — and those two passes exist only so that every breakpoint the speculative
// iterations took on a column subset can be checked against the other ~65 000 columns (find_max_gamma and inf_norm run
// over all columns: /root/reference/src/solvers/homotopy-cpu.cpp:100-164, :236).  That check asks one question per
// (column i outside the subset, state k of the path): is |c_i| = |a_i . r_k| below lambda_k — by how much is irrelevant.
// Here it is answered by ONE pass over a half-precision copy of A with a rigorous error bound:
//
//   k_sub_select   the 448 columns with the largest |c0| (subbatch.hip)
//   k_sgram_part / k_sgram_sum    Gs = A_S^T A_S of those columns from the fp32 A (v_mfma_f32_32x32x2_f32, 8 row chunks
//                  summed in a fixed order: deterministic)
//   k_sub_solve    the whole path on the subset, all arithmetic in fp32 on Gs (subbatch.hip, gsub = 1): this is what is
//                  REPORTED — support, coefficients, iterations, lambda
//   k_scr_residuals   r_k = y - A_S x_S(k) for every logged state k >= 1, in fp32, then scaled by a power of two and
//                  rounded to fp16; ||r_k||^2
//   k_scr_gemm     C~ = A16^T [r16_1 .. r16_K] on v_mfma_f32_32x32x16_f16, 1.07 GB instead of two passes of 2.15 GB, and in
//                  its epilogue, for every column outside the subset and every state:
//                      |c~_ik| + eps_ik <= bound_k        eps_ik = 2^-9 ||a_i|| ||r_k|| (+ the flush terms)
//
// Why that is enough (exact arithmetic first).  Let lambda_k be the logged max |c| of state k.  If every column outside the
// subset has |c_i(k)| < lambda_k and |c_i(k+1)| < lambda_{k+1} then (a) lambda_k = max |c| over ALL columns is the
// subset's, and (b) the column's step-length candidates (homotopy-cpu.cpp:130-161) are t = (lambda_k -/+ c_i)/(1 -/+ q_i) with
// q_i = (c_i(k) - c_i(k+1)) / gamma_k: positive numerators; a non-positive denominator gives t <= 0 or no candidate, which
// the reference skips; a positive one gives t > gamma_k  <=>  +/- c_i(k+1) < lambda_{k+1}.  So the reference's scan over
// all n columns picks what the subset's scan picked, state after state.  The bound keeps a MARGIN (1/8 of lambda_k, and
// 1e-5 lambda_0 absolute) between "certified" and "equal", far above any fp32 rounding of the reference's own correlations
// (the columns that matter sit at ~0.4 lambda on a Gaussian dictionary; the subset holds the ones near lambda).  The state
// a path ENDS in (lambda <= tolerance) is certified against the tolerance itself: no column outside the subset keeps the
// path going.  The last step before it is the rounding-level tie of every column that every form of this library treats
// the same way (whichever column the reference inserts there enters with x = 0).
//
// The error bound.  a16 = fl16(sA a), r16 = fl16(s_k r) with powers of two sA, s_k; round-to-nearest gives relative
// errors <= 2^-11 each in the normal range, the MFMA accumulates in fp32 (<= ldm 2^-24 relative to sum |a||r| at worst),
// so |c~ - c| <= (2^-10 + 2^-22 + ldm 2^-24) sum |a_ir||r_kr| <= 2^-9 ||a_i|| ||r_k|| for ldm <= 16384 (checked).  Entries
// below the fp16 normal range (and a possible flush of subnormal inputs by the matrix unit) add at most 2^-14 per entry in
// scaled units: 2^-14 sqrt(ldm) (||r_k|| / sA + ||a_i|| / s_k).  An fp16 overflow of a residual raises the failure flag.
//
// Nothing reported comes from the half-precision pass: a column it cannot certify makes the signal kStatusSubsetFail and
// the host solves it again in the default engine (exact fp32 passes over A) — like every declined signal of the subset form.
#include "ss_hip_internal.h"
#include "ss_hip_device.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace sship {

typedef _Float16 scr_h8 __attribute__((ext_vector_type(8)));
typedef float scr_v16f __attribute__((ext_vector_type(16)));
typedef float scr_v4f __attribute__((ext_vector_type(4)));
typedef uint32_t scr_u4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kScrRhs = 96;                 // right-hand sides of the screening pass: states 1 .. nlog - 1 (<= kSbLog - 1 = 79)
constexpr uint32_t kScrCols = 128;               // dictionary columns per workgroup of the pass
constexpr uint32_t kScrKc = 128;                 // rows per stage
constexpr uint32_t kScrPitchB = kScrKc * 2 + 16; // bytes per LDS row: 272 (16-byte reads of 8 consecutive rows: conflict-free)
constexpr uint32_t kScrTab = 4;                  // floats per state in the table: 1 / (sA s_k), bound_k, 1 / s_k, spare
constexpr uint32_t kSgSplit = 8;                 // row chunks of the subset Gram matrix (partials summed in order)
constexpr uint32_t kSgT = 64;                    // its tile: 64 x 64 outputs per workgroup, a 32 x 32 quadrant per wave
constexpr uint32_t kSgStep = 64;                 // rows staged per step
constexpr uint32_t kSgPitchF = kSgStep + 4;      // floats per LDS row
static_assert(kSbS % kSgT == 0, "subset Gram tiles");
static_assert(kSbLog - 1 <= kScrRhs, "screening pass: right-hand sides");

struct ScreenState {
    __half* a16 = nullptr;       // [n_pad][ldm] fl16(sA * A), column-contiguous like A
    float* anorm = nullptr;      // [n_pad] ||a_i||_2, rounded up
    float* meta = nullptr;       // [0] sA  [1] 1 / sA  [2] bits(max |A|)  [3] headroom of the last solve (bits, as uint)
    __half* r16 = nullptr;       // [kScrRhs][ldm] fl16(s_k * r_k)
    float* rn2p = nullptr;       // [ldm / 64][kScrRhs] partial sums of ||r_k||^2, one per workgroup of k_scr_residuals
    float* tab = nullptr;        // [kScrRhs][kScrTab]
    float* gs_part = nullptr;    // [kSgSplit][kSbS][kSbS]
    float* gs = nullptr;         // [kSbS][kSbS]
    int gemm_attr = -1;
};

// ---- one-time preparation ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_a16_stats(const float* __restrict__ At, uint32_t ldm, float* __restrict__ anorm, float* __restrict__ meta)
{
    __shared__ float sv[16];
    const float* a = At + (size_t)blockIdx.x * ldm;
    float ss = 0.f, mx = 0.f;
    for (uint32_t r = threadIdx.x * 4u; r < ldm; r += 1024u) {
        const scr_v4f v = *reinterpret_cast<const scr_v4f*>(a + r);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ss = __builtin_fmaf(v[e], v[e], ss); mx = fmaxf(mx, fabsf(v[e])); }
    }
    ss = block_sum(ss, sv);
    __syncthreads();
    // (max over the workgroup through the same scratch)
    mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
    mx = fmaxf(mx, __shfl_xor(mx, 8)); mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    if ((threadIdx.x & 63u) == 0u) sv[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        anorm[blockIdx.x] = sqrtf(ss) * 1.0001f;                       // (rounded up: it scales an upper bound)
        atomicMax(reinterpret_cast<uint32_t*>(meta) + 2, __float_as_uint(mx));
    }
}

__global__ void k_a16_scale(float* __restrict__ meta)
{
    const float amax = __uint_as_float(reinterpret_cast<const uint32_t*>(meta)[2]);
    int e = 0;
    if (amax > 0.f && amax < 3.0e38f) e = (int)floorf(log2f(16384.f / amax));
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    meta[0] = ldexpf(1.f, e);
    meta[1] = ldexpf(1.f, -e);
}

__global__ __launch_bounds__(256)
void k_a16_convert(const float* __restrict__ At, size_t total8, const float* __restrict__ meta, __half* __restrict__ a16)
{
    const float sA = meta[0];
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256u) {
        const scr_v4f v0 = *reinterpret_cast<const scr_v4f*>(At + 8u * i), v1 = *reinterpret_cast<const scr_v4f*>(At + 8u * i + 4u);
        scr_h8 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (_Float16)(v0[e] * sA); h[4 + e] = (_Float16)(v1[e] * sA); }
        *reinterpret_cast<scr_h8*>(a16 + 8u * i) = h;
    }
}

// ---- Gs = A_S^T A_S of the 448 subset columns, from the fp32 A ------------------------------------------------------
// One workgroup per (64 x 64 tile on or above the diagonal, row chunk); a 32 x 32 quadrant per wave on v_mfma_f32_32x32x2_f32;
// the columns' rows staged through LDS 64 at a time (coalesced 256-byte runs per column), the next stage's loads in flight
// under the MFMAs.  Partials per row chunk; mirrored store.
__global__ __launch_bounds__(256)
void k_sgram_part(const float* __restrict__ At, uint32_t ldm, uint32_t n, const uint32_t* __restrict__ sub, uint32_t rows_per,
                  float* __restrict__ part)
{
    __shared__ __attribute__((aligned(16))) float sI[kSgT][kSgPitchF];
    __shared__ __attribute__((aligned(16))) float sJ[kSgT][kSgPitchF];
    constexpr uint32_t NT = kSbS / kSgT;                        // 7 tiles per side
    const uint32_t b = blockIdx.x;
    uint32_t t = (uint32_t)((__fsqrt_rn(8.f * (float)b + 1.f) - 1.f) * 0.5f);
    while (t * (t + 1u) / 2u > b) --t;
    while ((t + 1u) * (t + 2u) / 2u <= b) ++t;
    const uint32_t bj = t, bi = b - t * (t + 1u) / 2u;
    (void)NT;
    const uint32_t chunk = blockIdx.y;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t wi = w >> 1, wj = w & 1u;
    const uint32_t sc = tid >> 4, sq = tid & 15u;               // staging: column sc (+16 p) of the 64, float4 sq of the 16 per step
    const float* gi[4];
    const float* gj[4];
    bool oki[4], okj[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t ci = sub[bi * kSgT + sc + 16u * (uint32_t)p], cj = sub[bj * kSgT + sc + 16u * (uint32_t)p];
        oki[p] = ci < n; okj[p] = cj < n;
        gi[p] = At + (size_t)(oki[p] ? ci : 0u) * ldm + (size_t)chunk * rows_per + 4u * sq;
        gj[p] = At + (size_t)(okj[p] ? cj : 0u) * ldm + (size_t)chunk * rows_per + 4u * sq;
    }
    const scr_v4f zero4 = { 0.f, 0.f, 0.f, 0.f };
    scr_v4f vi[4], vj[4];
#define SGM_LOAD(R0)                                                                      \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                       \
        vi[p] = oki[p] ? *reinterpret_cast<const scr_v4f*>(gi[p] + (R0)) : zero4;         \
        vj[p] = okj[p] ? *reinterpret_cast<const scr_v4f*>(gj[p] + (R0)) : zero4;         \
    }
    SGM_LOAD(0u)
    scr_v16f acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const uint32_t r = lane & 31u, h = lane >> 5;
    for (uint32_t r0 = 0; r0 < rows_per; r0 += kSgStep) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<scr_v4f*>(&sI[sc + 16u * (uint32_t)p][4u * sq]) = vi[p];
            *reinterpret_cast<scr_v4f*>(&sJ[sc + 16u * (uint32_t)p][4u * sq]) = vj[p];
        }
        __syncthreads();
        if (r0 + kSgStep < rows_per) SGM_LOAD(r0 + kSgStep)
#pragma unroll
        for (uint32_t k8 = 0; k8 < kSgStep; k8 += 8) {
            const scr_v4f a = *reinterpret_cast<const scr_v4f*>(&sI[32u * wi + r][k8 + 4u * h]);
            const scr_v4f bb = *reinterpret_cast<const scr_v4f*>(&sJ[32u * wj + r][k8 + 4u * h]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bb[e], acc, 0, 0, 0);
        }
    }
#undef SGM_LOAD
    float* P = part + (size_t)chunk * kSbS * kSbS;
    const uint32_t gjj = bj * kSgT + 32u * wj + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const uint32_t gii = bi * kSgT + 32u * wi + (uint32_t)(e & 3) + 8u * (uint32_t)(e >> 2) + 4u * h;
        P[(size_t)gii * kSbS + gjj] = acc[e];
        if (bi != bj) P[(size_t)gjj * kSbS + gii] = acc[e];
    }
}

__global__ __launch_bounds__(256)
void k_sgram_sum(const float* __restrict__ part, uint32_t nsplit, float* __restrict__ gs)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= kSbS * kSbS) return;
    float s = part[i];
    for (uint32_t c = 1; c < nsplit; ++c) s += part[(size_t)c * kSbS * kSbS + i];
    gs[i] = s;
}

// ---- r_k = y - A_S x_S(k) of every logged state k >= 1, scaled and rounded to fp16 -----------------------------------
// One workgroup per 64 rows: the support's columns' 64 rows and the coefficient table (transposed: [position][state]) in
// LDS; a thread owns 4 rows x 4 states (two 16-byte LDS reads per 16 fmas), states beyond 64 in a second round.
// ||r_k||^2 leaves as one partial per (workgroup, state) — summed in a fixed order by the screening pass.  Workgroup 0
// also writes the per-state table.  (Positions >= P_k carry x = 0 in the log: every state runs over all positions.)
__device__ __forceinline__ float scr_state_scale(float lam_eff)
{
    int e = 12;
    if (lam_eff > 0.f && lam_eff < 3.0e38f) e = (int)floorf(log2f(4096.f / lam_eff));
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return ldexpf(1.f, e);
}

__global__ __launch_bounds__(256)
void k_scr_residuals(const float* __restrict__ At, uint32_t ldm, uint32_t n, const float* __restrict__ y,
                     const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ pcol,
                     const float* __restrict__ LX, float tol, const float* __restrict__ meta, __half* __restrict__ r16,
                     float* __restrict__ rn2p, float* __restrict__ tab, uint32_t* __restrict__ headroom, DevState* __restrict__ st)
{
    __shared__ __attribute__((aligned(16))) float sAc[kSbRows][64];
    __shared__ __attribute__((aligned(16))) float sXt[kSbRows][kSbLog];
    __shared__ float sS[kSbLog];
    if (st->status != 0u) return;
    const uint32_t nlog = st->solo_nlog;
    if (nlog < 2u) return;
    const uint32_t nst = nlog - 1u;
    const uint32_t tid = threadIdx.x;
    const uint32_t r0 = blockIdx.x * 64u;
    const uint32_t Pfin = hdr[(nlog - 1u) * 8u];
    for (uint32_t e = tid; e < kSbLog * kSbRows; e += 256u) {
        const uint32_t kk = e / kSbRows, p = e - kk * kSbRows;
        sXt[p][kk] = kk < nst ? LX[(size_t)kSbRows + e] : 0.f;                 // (state k = kk + 1)
    }
    if (tid < kSbLog) {
        float sc = 1.f;
        if (tid < nst) {
            const uint32_t* hh = hdr + (size_t)(tid + 1u) * 8u;
            const float lam = __uint_as_float(hh[4]);
            const bool final_state = !(hh[1] & 1u);
            sc = scr_state_scale(final_state ? fmaxf(lam, tol) : lam);
        }
        sS[tid] = sc;
    }
    for (uint32_t p = tid >> 4; p < Pfin; p += 16u) {
        const uint32_t col = pcol[p];
        const scr_v4f v = col < n ? *reinterpret_cast<const scr_v4f*>(At + (size_t)col * ldm + r0 + 4u * (tid & 15u))
                                  : scr_v4f{ 0.f, 0.f, 0.f, 0.f };
        *reinterpret_cast<scr_v4f*>(&sAc[p][4u * (tid & 15u)]) = v;
    }
    __syncthreads();
    const uint32_t rg = tid & 15u, sgp = tid >> 4;
    const scr_v4f yv = *reinterpret_cast<const scr_v4f*>(y + r0 + 4u * rg);         // (rows m .. ldm - 1 of y and of A are zero)
    bool ovf = false;
    for (uint32_t s0 = 4u * sgp; s0 < nst; s0 += 64u) {
        scr_v4f acc[4] = { yv, yv, yv, yv };
        for (uint32_t p = 0; p < Pfin; ++p) {
            const scr_v4f a4 = *reinterpret_cast<const scr_v4f*>(&sAc[p][4u * rg]);
            const scr_v4f x4 = *reinterpret_cast<const scr_v4f*>(&sXt[p][s0]);
#pragma unroll
            for (int si = 0; si < 4; ++si)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) acc[si][ri] = __builtin_fmaf(-x4[si], a4[ri], acc[si][ri]);
        }
#pragma unroll
        for (int si = 0; si < 4; ++si) {
            const uint32_t kk = s0 + (uint32_t)si;                    // (uniform over the 16 lanes of a state group)
            float ss = 0.f;
            const float sc = sS[kk < kSbLog ? kk : 0u];
            __half hv[4];
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const float r = acc[si][ri];
                ss = __builtin_fmaf(r, r, ss);
                const float v = r * sc;
                if (kk < nst && !(fabsf(v) < 60000.f)) ovf = true;
                hv[ri] = __float2half_rn(v);
            }
            ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4); ss += __shfl_xor(ss, 8);
            if (kk < nst) {
                *reinterpret_cast<uint2*>(r16 + (size_t)kk * ldm + r0 + 4u * rg) = *reinterpret_cast<const uint2*>(hv);
                if (rg == 0u) rn2p[(size_t)blockIdx.x * kScrRhs + kk] = ss;
            }
        }
    }
    if (ovf) __hip_atomic_store(&st->need_sweep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0u) {
        const float lam0 = st->lambda0;
        if (tid == 0u) *headroom = 0u;
        if (tid < nst) {
            const uint32_t* hh = hdr + (size_t)(tid + 1u) * 8u;
            const float lam = __uint_as_float(hh[4]);
            const bool final_state = !(hh[1] & 1u);
            const float slack = 1e-5f * lam0;
            float bound;
            if (final_state && !(lam > tol)) bound = tol * 0.9375f - slack;     // the path ended by tolerance: nothing out there keeps it going
            else bound = lam * 0.875f - slack;
            const float inv_sk = 1.f / sS[tid];
            tab[tid * kScrTab + 0] = meta[1] * inv_sk;
            tab[tid * kScrTab + 1] = bound;
            tab[tid * kScrTab + 2] = inv_sk;
            tab[tid * kScrTab + 3] = lam;
        }
    }
}

// ---- the screening pass: C~ = A16^T R16 and the test of every (column outside the subset, state) -------------------
// Workgroup = 128 dictionary columns x up to 96 right-hand sides; wave = 32 columns x 2 or 3 MFMA tiles of 32 states (the
// third only when the path logged more than 64 states).  A stage is 128 rows: 32 KB of A16 and 16 / 24 KB of R16 (from
// L2) land in LDS by 16-byte stores of coalesced 256-byte runs; TWO stages of loads are in flight (two register sets)
// under the MFMAs of a third.  HBM-bound: 1.07 GB at 8192 x 65536.
__global__ __launch_bounds__(256, 2)
void k_scr_gemm(const __half* __restrict__ a16, uint32_t ldm, uint32_t n, const __half* __restrict__ r16,
                const float* __restrict__ anorm, const float* __restrict__ rn2p, const float* __restrict__ tab,
                const uint32_t* __restrict__ sub, const float* __restrict__ meta, DevState* __restrict__ st, uint32_t* __restrict__ headroom)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (st->status != 0u) return;
    const uint32_t nlog = st->solo_nlog;
    if (nlog < 2u) return;
    const uint32_t nst = nlog - 1u;
    const bool t3 = nst > 64u;                                   // (uniform) the third tile of states is in use
    unsigned char* sA = smem;                                   // [128][272]
    unsigned char* sR = smem + (size_t)kScrCols * kScrPitchB;   // [96][272]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t col0 = blockIdx.x * kScrCols;
    const uint32_t lc = tid >> 4, piece = tid & 15u;
    const __half* ga = a16 + (size_t)(col0 + lc) * ldm + 8u * piece;
    const __half* gr = r16 + (size_t)lc * ldm + 8u * piece;
    scr_u4 pa0[8], pr0[6], pa1[8], pr1[6];
#define SCR_LOAD(PA, PR, R0)                                                                                      \
    {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                             \
            PA[i] = __builtin_nontemporal_load(reinterpret_cast<const scr_u4*>(ga + (size_t)(16 * i) * ldm + (R0)));  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
            PR[i] = *reinterpret_cast<const scr_u4*>(gr + (size_t)(16 * i) * ldm + (R0));                         \
        if (t3) {                                                                                                 \
            _Pragma("unroll") for (int i = 4; i < 6; ++i)                                                         \
                PR[i] = *reinterpret_cast<const scr_u4*>(gr + (size_t)(16 * i) * ldm + (R0));                     \
        }                                                                                                         \
    }
    scr_v16f acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const uint32_t r = lane & 31u, h = lane >> 5;
    const unsigned char* rdA = sA + (size_t)(32u * w + r) * kScrPitchB + 16u * h;
    const unsigned char* rdR = sR + (size_t)r * kScrPitchB + 16u * h;
#define SCR_STAGE(PA, PR, MORE, RNEXT)                                                                                \
    {                                                                                                             \
        __syncthreads();                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                             \
            *reinterpret_cast<scr_u4*>(sA + (size_t)(lc + 16u * (uint32_t)i) * kScrPitchB + 16u * piece) = PA[i]; \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
            *reinterpret_cast<scr_u4*>(sR + (size_t)(lc + 16u * (uint32_t)i) * kScrPitchB + 16u * piece) = PR[i]; \
        if (t3) {                                                                                                 \
            _Pragma("unroll") for (int i = 4; i < 6; ++i)                                                         \
                *reinterpret_cast<scr_u4*>(sR + (size_t)(lc + 16u * (uint32_t)i) * kScrPitchB + 16u * piece) = PR[i]; \
        }                                                                                                         \
        __syncthreads();                                                                                          \
        if (MORE) SCR_LOAD(PA, PR, (RNEXT))                                                                       \
        _Pragma("unroll") for (uint32_t ks = 0; ks < kScrKc / 16u; ++ks) {                                        \
            const scr_h8 bq = *reinterpret_cast<const scr_h8*>(rdA + 32u * ks);                                   \
            const scr_h8 aq0 = *reinterpret_cast<const scr_h8*>(rdR + 32u * ks);                                  \
            const scr_h8 aq1 = *reinterpret_cast<const scr_h8*>(rdR + (size_t)32 * kScrPitchB + 32u * ks);        \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq0, bq, acc[0], 0, 0, 0);                            \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq1, bq, acc[1], 0, 0, 0);                            \
            if (t3) {                                                                                             \
                const scr_h8 aq2 = *reinterpret_cast<const scr_h8*>(rdR + (size_t)64 * kScrPitchB + 32u * ks);    \
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq2, bq, acc[2], 0, 0, 0);                        \
            }                                                                                                     \
        }                                                                                                         \
    }
    // (ldm is a multiple of 256: an even number of stages.)  Every workgroup starts at a row offset of its own and wraps
    // around — the sum's order is free here — so that the 512 workgroups do not walk the same 256-byte phase of their
    // 16-KiB-strided columns together (the HBM channels are selected by those address bits)
    const uint32_t nstage = ldm / kScrKc;
    const uint32_t sbase = (blockIdx.x * 29u) % nstage;
#define SCR_ROW(S) ((sbase + (S) >= nstage ? sbase + (S) - nstage : sbase + (S)) * kScrKc)
    SCR_LOAD(pa0, pr0, SCR_ROW(0u))
    SCR_LOAD(pa1, pr1, SCR_ROW(1u))
    for (uint32_t sidx = 0; sidx < nstage; sidx += 2u) {
        SCR_STAGE(pa0, pr0, sidx + 2u < nstage, SCR_ROW(sidx + 2u))
        SCR_STAGE(pa1, pr1, sidx + 3u < nstage, SCR_ROW(sidx + 3u))
    }
#undef SCR_ROW
#undef SCR_STAGE
#undef SCR_LOAD
    // ---- epilogue: the per-state table and the subset's columns into LDS, then every (column, state) of this wave ------
    __syncthreads();
    float* sT = reinterpret_cast<float*>(smem);                 // [96][4]: 1/(sA s_k), bound, eps factor 1 (x ||a||), eps term 2
    uint32_t* sSub = reinterpret_cast<uint32_t*>(smem) + kScrRhs * 4u;   // [kSbS] the subset's columns, ascending (0xffffffff: none)
    float* sPart = reinterpret_cast<float*>(smem) + kScrRhs * 4u + kSbS; // [ldm / 64][96] the partial sums of ||r_k||^2
    const float inv_sA = meta[1];
    const float sq_ldm = sqrtf((float)ldm);
    const uint32_t nblk = ldm / 64u;
    // (128 workgroups' partials in flight at once, summed per state in the order of the workgroups that wrote them: deterministic)
    float s2 = 0.f;
    for (uint32_t b0 = 0; b0 < nblk; b0 += 128u) {
        const uint32_t nb = nblk - b0 < 128u ? nblk - b0 : 128u;
        for (uint32_t e = tid; e < nb * kScrRhs; e += 256u) sPart[e] = rn2p[(size_t)b0 * kScrRhs + e];
        __syncthreads();
        if (tid < nst)
            for (uint32_t b = 0; b < nb; ++b) s2 += sPart[(size_t)b * kScrRhs + tid];
        __syncthreads();
    }
    if (tid < kScrRhs) {
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;
        if (tid < nst) {
            const float rn = sqrtf(s2) * 1.001f;
            const float inv_sk = tab[tid * kScrTab + 2];
            f0 = tab[tid * kScrTab + 0];
            f1 = tab[tid * kScrTab + 1];
            f2 = 0.001953125f * rn + 6.103515625e-05f * sq_ldm * inv_sk;        // x ||a_i||:  2^-9 ||r_k|| + 2^-14 sqrt(ldm) / s_k
            f3 = 6.103515625e-05f * sq_ldm * rn * inv_sA;                        // 2^-14 sqrt(ldm) ||r_k|| / sA
        }
        sT[tid * 4 + 0] = f0; sT[tid * 4 + 1] = f1; sT[tid * 4 + 2] = f2; sT[tid * 4 + 3] = f3;
    }
    for (uint32_t e = tid; e < kSbS; e += 256u) sSub[e] = sub[e];
    __syncthreads();
    const uint32_t col = col0 + 32u * w + r;
    // is the column in the subset (k_sub_solve dealt with those)?  lower bound in the ascending list
    uint32_t lo = 0, hi = kSbS;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sSub[mid] < col) lo = mid + 1u; else hi = mid; }
    const bool mine = col < n && !(lo < kSbS && sSub[lo] == col);
    const float an = anorm[col < n ? col : 0u];
    bool flag = false;
    float worst = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t kk = 32u * (uint32_t)t + (uint32_t)(e & 3) + 8u * (uint32_t)(e >> 2) + 4u * h;
            if (kk < nst) {
                const scr_v4f T4 = *reinterpret_cast<const scr_v4f*>(&sT[kk * 4u]);
                const float v = fabsf(acc[t][e]) * T4[0] + (an * T4[2] + T4[3]);
                if (!(v <= T4[1])) flag = true;
                const float ratio = T4[1] > 0.f ? v / T4[1] : 3.0e38f;
                worst = fmaxf(worst, ratio == ratio ? ratio : 3.0e38f);
            }
        }
    }
    if (mine && flag) __hip_atomic_store(&st->need_sweep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read by k_sub_finish)
    if (!mine) worst = 0.f;
    worst = fmaxf(worst, __shfl_xor(worst, 1)); worst = fmaxf(worst, __shfl_xor(worst, 2)); worst = fmaxf(worst, __shfl_xor(worst, 4));
    worst = fmaxf(worst, __shfl_xor(worst, 8)); worst = fmaxf(worst, __shfl_xor(worst, 16)); worst = fmaxf(worst, __shfl_xor(worst, 32));
    if (lane == 0u && worst > 0.f) atomicMax(headroom, __float_as_uint(worst));
}

// ---- host side ---------------------------------------------------------------------------------------------------
static ScreenState* scr_of(ss_hip_ctx* ctx) { return static_cast<ScreenState*>(ctx->screen); }

void screen_free(ss_hip_ctx* ctx)
{
    ScreenState* S = scr_of(ctx);
    if (!S) return;
    void* ptrs[] = { S->a16, S->anorm, S->meta, S->r16, S->rn2p, S->tab, S->gs_part, S->gs };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete S;
    ctx->screen = nullptr;
}

// (the main loop's two tiles; the epilogue's tables — 96 x 4 + 448 + 128 x 96 floats — fit inside)
static size_t scr_gemm_lds(uint32_t) { return std::max<size_t>((size_t)(kScrCols + kScrRhs) * kScrPitchB, ((size_t)kScrRhs * 4 + kSbS + (size_t)128 * kScrRhs) * 4); }

// Shape / option test, and — the first time it says yes — the preparation: the fp16 copy of A (half of A's bytes again),
// the column norms.  A failed allocation switches the form off for this context (the default engine goes on as before).
bool screen_form_usable(ss_hip_ctx* ctx)
{
    if (ctx->screen_single == 0 || ctx->is_f64 || ctx->kind != 0 || ctx->screen_failed_alloc) return false;
    if (ctx->colshard != nullptr) return false;
    const uint32_t ldm = ctx->ldm, np = ctx->n_pad;
    // the bound's accumulation term assumes ldm <= 16384; tiles: 128 columns, 128 rows; subset Gram: 4 or 8 chunks of 64-row steps
    if (ldm % kScrKc != 0 || np % kScrCols != 0 || ldm > 16384u || ctx->n < kSbS) return false;
    // where it pays: the pass over the fp16 copy replaces two fp32 passes — dictionaries of at least 16 Mi entries (option 2: any shape)
    if (ctx->screen_single < 2 && ((size_t)ctx->m * ctx->n < ((size_t)16 << 20) || ctx->n < 8192u)) return false;
    if (!sub_form_usable(ctx)) return false;
    if (ctx->screen != nullptr) return true;
    ScreenState* S = new ScreenState();
    ctx->screen = S;
    bool ok = true;
    auto alloc = [&](void** p, size_t bytes) { if (ok && hipMalloc(p, bytes) != hipSuccess) { (void)hipGetLastError(); ok = false; } };
    alloc(reinterpret_cast<void**>(&S->a16), (size_t)np * ldm * sizeof(__half));
    alloc(reinterpret_cast<void**>(&S->anorm), (size_t)np * sizeof(float));
    alloc(reinterpret_cast<void**>(&S->meta), 4 * sizeof(float));
    alloc(reinterpret_cast<void**>(&S->r16), (size_t)kScrRhs * ldm * sizeof(__half));
    alloc(reinterpret_cast<void**>(&S->rn2p), (size_t)(ldm / 64u) * kScrRhs * sizeof(float));
    alloc(reinterpret_cast<void**>(&S->tab), (size_t)kScrRhs * kScrTab * sizeof(float));
    alloc(reinterpret_cast<void**>(&S->gs_part), (size_t)kSgSplit * kSbS * kSbS * sizeof(float));
    alloc(reinterpret_cast<void**>(&S->gs), (size_t)kSbS * kSbS * sizeof(float));
    if (ok) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scr_gemm), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)scr_gemm_lds(ldm));
        if (e != hipSuccess) { (void)hipGetLastError(); ok = false; }
    }
    if (!ok) { screen_free(ctx); ctx->screen_failed_alloc = 1; return false; }
    hipStream_t s = ctx->stream;
    const float* At = static_cast<const float*>(ctx->At);
    (void)hipMemsetAsync(S->meta, 0, 4 * sizeof(float), s);
    (void)hipMemsetAsync(S->r16, 0, (size_t)kScrRhs * ldm * sizeof(__half), s);
    hipLaunchKernelGGL(k_a16_stats, dim3(np), dim3(256), 0, s, At, ldm, S->anorm, S->meta);
    hipLaunchKernelGGL(k_a16_scale, dim3(1), dim3(1), 0, s, S->meta);
    const size_t total8 = (size_t)np * ldm / 8;
    hipLaunchKernelGGL(k_a16_convert, dim3((unsigned)std::min<size_t>((total8 + 255) / 256, 65536)), dim3(256), 0, s, At, total8,
                       (const float*)S->meta, S->a16);
    if (hipGetLastError() != hipSuccess) { screen_free(ctx); ctx->screen_failed_alloc = 1; return false; }
    return true;
}

// c0 = A^T y is in ws.c0, r = y in ws.rhs (block 0); everything on the context's stream.  e0..e3 (profiling): before the
// selection, before the residuals, before and after the screening pass.
hipError_t launch_screen_form(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, uint32_t max_iter, hipEvent_t e0, hipEvent_t e1,
                              hipEvent_t e2, hipEvent_t e3)
{
    ScreenState* S = scr_of(ctx);
    if (S == nullptr || ctx->sub_buf == nullptr) return hipErrorInvalidConfiguration;
    const SubBufs B = sub_bufs(ctx, 1);
    hipStream_t s = ctx->stream;
    const uint32_t ldm = ctx->ldm, n = (uint32_t)ctx->n, np = ctx->n_pad;
    const float* At = static_cast<const float*>(ctx->At);
    if (e0) (void)hipEventRecord(e0, s);
    (void)launch_sub_select(ctx, B, 1, ws.c0);
    const uint32_t nsplit = (ldm % (kSgSplit * kSgStep) == 0) ? kSgSplit : 4u;        // (ldm is a multiple of 256)
    constexpr uint32_t NT = kSbS / kSgT;
    hipLaunchKernelGGL(k_sgram_part, dim3(NT * (NT + 1) / 2, nsplit), dim3(256), 0, s, At, ldm, n, (const uint32_t*)B.sub, ldm / nsplit, S->gs_part);
    hipLaunchKernelGGL(k_sgram_sum, dim3((kSbS * kSbS + 255) / 256), dim3(256), 0, s, (const float*)S->gs_part, nsplit, S->gs);
    (void)launch_sub_solve(ctx, ws, B, 1, (const float*)S->gs, kSbS, 1, ws.c0, tol, max_iter);
    if (e1) (void)hipEventRecord(e1, s);
    hipLaunchKernelGGL(k_scr_residuals, dim3(ldm / 64u), dim3(256), 0, s, At, ldm, n, (const float*)ws.rhs,
                       (const uint32_t*)B.hdr, (const uint32_t*)B.pcol, (const float*)B.LX, tol,
                       (const float*)S->meta, S->r16, S->rn2p, S->tab, reinterpret_cast<uint32_t*>(S->meta) + 3, ws.st);
    if (e2) (void)hipEventRecord(e2, s);
    hipLaunchKernelGGL(k_scr_gemm, dim3(np / kScrCols), dim3(256), scr_gemm_lds(ldm), s, (const __half*)S->a16, ldm, n, (const __half*)S->r16,
                       (const float*)S->anorm, (const float*)S->rn2p, (const float*)S->tab, (const uint32_t*)B.sub, (const float*)S->meta,
                       ws.st, reinterpret_cast<uint32_t*>(S->meta) + 3);
    if (e3) (void)hipEventRecord(e3, s);
    (void)launch_sub_finish(ctx, ws, 1);
    return hipGetLastError();
}

double screen_read_headroom(ss_hip_ctx* ctx)
{
    ScreenState* S = scr_of(ctx);
    if (S == nullptr) return 0.0;
    uint32_t bits = 0;
    if (hipMemcpy(&bits, reinterpret_cast<uint32_t*>(S->meta) + 3, sizeof(bits), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return 0.0; }
    float f;
    std::memcpy(&f, &bits, sizeof(f));
    return (double)f;
}

}  // namespace sship
