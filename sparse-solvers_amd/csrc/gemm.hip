// gemm.hip — batched correlations  D = R · Atᵀ  on the MFMA units (fp32 in, fp32 accumulate).
//
// When B signals share one sensing matrix, the 2B transposed GEMVs of a lock-step Homotopy
// round (c_b = Aᵀ r_b, q_b = Aᵀ p_b; /root/reference/src/solvers/homotopy-cpu.cpp:97,:120)
// become one GEMM with arithmetic intensity ≈ B flop/B — compute-bound from B ≈ 16, which
// is where the matrix cores pay (a single signal is memory-bound: sweep.hip).
//
//   R  : [Mg][ldr]   right-hand sides, one per row (r_0..r_{B-1}, p_0..p_{B-1}), K-contiguous
//   At : [Ng][ldq]   dictionary columns, K-contiguous (the context's device copy)
//   D  : [Mg][ldd]   D[b][j] = sum_k R[b][k] * At[j][k]
//
// Both operands are K-contiguous, so each lane fetches 4 consecutive k of one row with a
// single ds_read_b128 and feeds them to 4 `v_mfma_f32_32x32x2_f32`.  The instruction takes
// k = lane>>5 from each half-wave; the lower half supplies k-quad 2g, the upper half quad
// 2g+1 of an 8-wide k-group — a permutation of k that is the same for both operands, so every
// product a[i][k]·b[j][k] appears exactly once.  f32 MFMA is an exact fp32 fma chain
// (cdna_hip_programming.md §3), i.e. the numerics of the GEMV path, in a different order.
//
// Tile: 128 x 128 x 32 per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4 MFMA
// accumulators), register-staged double buffering (global loads of step t+1 in flight under
// the 64 MFMAs of step t), one barrier per K-step, 2 workgroups per CU (72 KiB LDS each).
// Roofline: MFMA fp32, 157.3 TFLOP/s dense peak (MI355X_MICROARCH.md).
#include "ss_hip_internal.h"

namespace sship {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int GM = 128, GN = 128, GK = 32, GPAD = 4;
constexpr int GLD = GK + GPAD;                 // LDS row pitch in floats (144 B: conflict-free b128 reads)

__global__ __launch_bounds__(256, 2)
void k_gemm_tn_f32(const float* __restrict__ R, const float* __restrict__ Q, float* __restrict__ D,
                   uint32_t mtiles, uint32_t K, uint32_t ldr, uint32_t ldq, uint32_t ldd,
                   const uint32_t* __restrict__ row_tile_skip)
{
    __shared__ __attribute__((aligned(16))) float sR[2][GM][GLD];
    __shared__ __attribute__((aligned(16))) float sQ[2][GN][GLD];

    // row tiles fastest: concurrently resident workgroups share the same panel of At.  With a
    // tile list only its `nact` tiles are computed, by the leading nact*ntiles workgroups.
    uint32_t bm, bn;
    if (row_tile_skip != nullptr) {
        const uint32_t nact = row_tile_skip[mtiles];
        if (nact == 0 || blockIdx.x / nact >= gridDim.x / mtiles) return;
        bm = row_tile_skip[blockIdx.x % nact];
        bn = blockIdx.x / nact;
    } else {
        bm = blockIdx.x % mtiles;
        bn = blockIdx.x / mtiles;
    }

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t wm = wave & 1u, wn = wave >> 1;
    const uint32_t h = lane >> 5, l31 = lane & 31u;

    // staging map: thread -> (row r0 + 32*j, k-quad)
    const uint32_t srow = tid >> 3, squad = tid & 7u;
    const float* gR = R + (size_t)(bm * GM + srow) * ldr + squad * 4;
    const float* gQ = Q + (size_t)(bn * GN + srow) * ldq + squad * 4;

    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    v4f stR[4], stQ[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        stR[j] = *reinterpret_cast<const v4f*>(gR + (size_t)(32 * j) * ldr);
        stQ[j] = *reinterpret_cast<const v4f*>(gQ + (size_t)(32 * j) * ldq);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<v4f*>(&sR[0][srow + 32 * j][squad * 4]) = stR[j];
        *reinterpret_cast<v4f*>(&sQ[0][srow + 32 * j][squad * 4]) = stQ[j];
    }
    __syncthreads();

    const uint32_t nk = K / GK;
    uint32_t cur = 0;
    for (uint32_t kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) {
            const uint32_t koff = (kt + 1) * GK;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                stR[j] = *reinterpret_cast<const v4f*>(gR + (size_t)(32 * j) * ldr + koff);
                stQ[j] = *reinterpret_cast<const v4f*>(gQ + (size_t)(32 * j) * ldq + koff);
            }
        }
#pragma unroll
        for (int g = 0; g < GK / 8; ++g) {
            const uint32_t kq = (2u * g + h) * 4u;
            v4f a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const v4f*>(&sR[cur][wm * 64 + i * 32 + l31][kq]);
                b[i] = *reinterpret_cast<const v4f*>(&sQ[cur][wn * 64 + i * 32 + l31][kq]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<v4f*>(&sR[cur ^ 1u][srow + 32 * j][squad * 4]) = stR[j];
                *reinterpret_cast<v4f*>(&sQ[cur ^ 1u][srow + 32 * j][squad * 4]) = stQ[j];
            }
        }
        __syncthreads();
        cur ^= 1u;
    }

    // C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t col = bn * GN + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t row = bm * GM + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                D[(size_t)row * ldd + col] = acc[i][j][e];
            }
        }
}

// D[Mg][ldd] = R[Mg][ldr] * At^T ; Mg % 128 == 0, ctx->n_pad % 128 == 0, ldm % 32 == 0
hipError_t launch_gemm_tn_f32(const ss_hip_ctx* ctx, const float* R, uint32_t Mg, uint32_t ldr,
                              float* D, uint32_t ldd, const uint32_t* row_tile_skip)
{
    if (Mg % GM != 0 || ctx->n_pad % GN != 0 || ctx->ldm % GK != 0) return hipErrorInvalidValue;
    const uint32_t mtiles = Mg / GM, ntiles = ctx->n_pad / GN;
    hipLaunchKernelGGL(k_gemm_tn_f32, dim3(mtiles * ntiles), dim3(256), 0, ctx->stream, R,
                       static_cast<const float*>(ctx->At), D, mtiles, ctx->ldm, ldr, ctx->ldm, ldd,
                       row_tile_skip);
    return hipGetLastError();
}

}  // namespace sship
