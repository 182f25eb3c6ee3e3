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

// SYM: the symmetric product G = At · At^T (R == Q, square): only the tiles on and above the diagonal are
// computed (blockIdx enumerates the pairs bm <= bn, column panel by column panel) and every off-diagonal tile
// is stored to both sides.  G[i][j] and G[j][i] are the same k-ordered fma chain of the same (commutative)
// products, so the mirrored copy is bit for bit what the full product would have put there.
// BLK (the correlations of a batch, c = A^T y and the two GEMMs of a GEMM-form round): an output that is ONE fma
// chain over all m rows carries a rounding error that grows like sqrt(m) * eps * |partial sums| — five times what the
// single-signal sweep's 64 per-lane sums + tree leave in c0 = A^T y (measured on the step lengths of small steps:
// tools/dbg_colform.py) — and that error stays in every c = c0 - sum_j x_j g_j of the path.  With BLK
// every K-step (32 rows) runs its own chain from zero and the 32-row sums are added up in a second accumulator:
// chains of 32 + m / 32 instead of m.  Not for SYM: G's entries have to be the mirrored-tile chains.
template <bool SYM, bool BLK = false>
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
    if (SYM) {
        // blockIdx.x = bn (bn + 1) / 2 + bm with bm <= bn
        const uint32_t b = blockIdx.x;
        uint32_t t = (uint32_t)((__fsqrt_rn(8.f * (float)b + 1.f) - 1.f) * 0.5f);
        while ((uint64_t)t * (t + 1u) / 2u > b) --t;
        while ((uint64_t)(t + 1u) * (t + 2u) / 2u <= b) ++t;
        bn = t;
        bm = b - (uint32_t)((uint64_t)t * (t + 1u) / 2u);
    } else if (row_tile_skip != nullptr) {
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
    v16f tot[BLK ? 2 : 1][BLK ? 2 : 1];                     // BLK: the sum of the 32-row chains so far
    v16f zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[i][j] = zero16;
            if (BLK) tot[i][j] = zero16;
        }

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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], (BLK && g == 0 && t == 0) ? zero16 : acc[i][j], 0, 0, 0);
        }
        if (BLK) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) tot[i][j][e] += acc[i][j][e];
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
                D[(size_t)row * ldd + col] = BLK ? tot[i][j][e] : acc[i][j][e];
            }
            if (SYM && bm != bn) {
                // the mirrored tile: registers 4g .. 4g+3 are four consecutive rows, i.e. 16 contiguous bytes of
                // row `col` of G; the two half-waves and the four groups fill one 128-byte line per column
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t row0 = bm * GM + wm * 64 + i * 32 + 8 * g + 4 * h;
                    v4f v = { acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3] };
                    *reinterpret_cast<v4f*>(&D[(size_t)col * ldd + row0]) = v;
                }
            }
        }
}

// ---- 32-row variant: the "lookahead sweep" ---------------------------------------------------
// D[s][:] = At · r_s for up to 32 right-hand sides in ONE pass over At.  With 32 rows the MFMA
// work per byte of At is 16 flop/B — still left of the fp32 ridge (157 TFLOP/s / 6.6 TB/s =
// 24 flop/B) — so this kernel is HBM-bound like the 2-RHS sweep and costs about the same time,
// but yields 32 correlation vectors.  The single-signal solver uses it to fetch the Gram columns
// A^T a_j of the 32 most likely next entrants at once (homotopy.hip, engine "lookahead").
//
// Right-hand sides are rows of At itself, named by rcols[s] (0xffffffff = unused row -> zeros);
// output row s goes to D + drows[s]*ldd (scattered into the Gram-column cache).
// Tile 32 x 256 x 32 per 256-thread workgroup: wave w owns dictionary columns [64w, 64w+64) as
// two 32x32 MFMA tiles; same K-contiguous ds_read_b128 feeding as above.
constexpr int HM = 32;

// One workgroup per CU (83 KiB LDS), 8 waves, each wave one 32x32 accumulator (32 dictionary
// columns).  To keep enough HBM requests in flight from a single workgroup the global loads
// run THREE K-steps ahead through a ring of register sets (3 x 36 KiB per workgroup), LDS is
// double buffered, one barrier per K-step.
// RH = 32 or 64 right-hand sides per pass.  With 64 the pass is no longer HBM-bound: 2 * 64 * m * n flops
// (68.7 GFLOP at C2) take ~0.45 ms on the fp32 MFMA units against 0.27 ms for the bytes — but it replaces TWO
// 32-column passes (2 x 0.37 ms) and one round trip through the host, which is why the first lookahead sweep
// of a solve (the entering column + the 63 largest |c0|) uses it.
// developer aid (option pass_dbg_ptr): where and when every workgroup of the lookahead passes ran — entry e of the
// buffer: {start, end (100 MHz ticks), XCC_ID << 16 | HW_ID, blockIdx.x | tiles << 32}; word 0 counts the entries
__device__ uint64_t* g_pass_dbg = nullptr;

// BYSE (early form, the tiles the main launch leaves: homotopy.hip): the workgroup's tile follows from WHERE it runs.
// The hardware deals a grid out statically — every shader engine of every XCD gets the same number of workgroups,
// whatever its CUs hold — so a launch of 2 * kSeCount workgroups puts two on every SE; a workgroup reads its (XCC, SE)
// from the hardware registers, leaves at once if that is the solo workgroup's SE (st->solo_where: its 7 other CUs have
// their two tiles from the main launch), and otherwise — if it is the first or second of this launch on its SE
// (se_count) — takes the next tile from `first` on.  Every CU but the solo workgroup's then carries two tiles.
// FIXUP: a third launch in the same stream covers whatever tiles that left undone (none, as long as the dealing-out is
// what was measured; the RESULT must not depend on it).
template <int HN, int HT, int BPC, int KS = GK, bool DRY = false, int RH = HM, bool BYSE = false, bool FIXUP = false>
__global__ __launch_bounds__(HT, BPC)
void k_gemm32_tn_f32(const float* __restrict__ At, const uint32_t* __restrict__ rcols,
                     const uint32_t* __restrict__ drows, float* __restrict__ D,
                     uint32_t K, uint32_t ldq, uint32_t ldd, uint32_t ntiles,
                     const DevState* __restrict__ st, uint32_t first, uint32_t* se_count, uint32_t quota)
{
    if (!BYSE && st != nullptr && (st->done != 0 || st->need_sweep != 1)) return;   // no sweep needed this round
    // (BYSE: arrivals, for the main launch of the same pass — it is held back until all of these hold their CUs: a
    // workgroup of THIS launch that finds its SE full would hold up every workgroup behind it in the grid)
    if (BYSE && !FIXUP && threadIdx.x == 0) __hip_atomic_fetch_add(&se_count[kSeCount], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (rcols[0] == 0xffffffffu) return;                        // an empty list (lists are filled from entry 0)
    uint32_t my_tile = 0;
    __shared__ uint32_t s_tile;
    // BYSE: claims the workgroup's next tile (0xffffffff: none left for it).  `quota` tiles per shader engine in all —
    // two at 8192 x 65536 (one per workgroup of the launch); wider dictionaries give every CU more tiles, and the
    // workgroups of this launch then come back for more until their SE has had its share (round 3: any tile count)
    auto claim = [&]() {
        if (threadIdx.x == 0) {
            uint32_t hw_, xcc_;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
            const uint32_t se = ((xcc_ & 7u) << 2) | ((hw_ >> 13) & 3u);
            const uint32_t where = __hip_atomic_load(&st->solo_where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t solo_se = 0xffffffffu;
            if (where != 0u) solo_se = ((((where - 1u) >> 16) & 7u) << 2) | (((where - 1u) >> 13) & 3u);
            uint32_t t = 0xffffffffu;
            if (se != solo_se) {
                const uint32_t slot = __hip_atomic_fetch_add(&se_count[se], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (slot < quota) t = first + __hip_atomic_fetch_add(&se_count[kSeCount + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_tile = t;
        }
        __syncthreads();
        const uint32_t t = s_tile;
        __syncthreads();
        return t;
    };
    if (BYSE) {
        if (FIXUP) {
            // third launch of the pass (same stream, afterwards): whatever the second left undone — nothing, unless
            // the hardware dealt its workgroups out differently than assumed; coverage must not depend on that
            my_tile = first + __hip_atomic_load(&se_count[kSeCount + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + blockIdx.x;
        } else {
            my_tile = claim();
        }
        if (my_tile >= ntiles) return;
    }
    uint64_t* const pdbg = g_pass_dbg;
    const uint64_t t_dbg0 = pdbg != nullptr ? wall_clock64() : 0ull;
    constexpr int LD = KS + GPAD;                               // LDS row pitch in floats
    constexpr int RB = RH / 32;                                 // 32-row blocks of right-hand sides
    __shared__ __attribute__((aligned(16))) float sR[2][RH][LD];
    __shared__ __attribute__((aligned(16))) float sQ[2][HN][LD];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t h = lane >> 5, l31 = lane & 31u;
    constexpr int TPR = KS / 4;                                 // threads per staged row (one k-quad each)
    constexpr int RPP = HT / TPR;                               // rows staged per pass
    constexpr int NJ = HN / RPP;                                // passes per column tile
    static_assert(RH * TPR <= HT, "the R tile is staged in one pass");
    const uint32_t srow = tid / TPR, squad = tid % TPR;        // staging: rows srow + RPP*j, k-quad squad
    const bool has_r = tid < RH * TPR;                          // R tile: RH rows x TPR quads

    const uint32_t rc = rcols[srow % (uint32_t)RH];
    const bool rvalid = has_r && rc != 0xffffffffu;
    const float* gR = At + (size_t)(rvalid ? rc : 0u) * ldq + squad * 4;
    const v4f zero4 = { 0.f, 0.f, 0.f, 0.f };
    const uint32_t nk = K / KS;

    // (plain launches: tiles first + blockIdx.x, + gridDim.x, ...; BYSE: one claimed tile after the other)
    for (uint32_t bn = BYSE ? my_tile : first + blockIdx.x; bn < ntiles; bn = BYSE ? ((FIXUP || quota <= 2u) ? ntiles : claim()) : bn + gridDim.x) {
        const float* gQ = At + (size_t)(bn * HN + srow) * ldq + squad * 4;
        v16f acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

        v4f rR[3], rQ[3][NJ];
#define G32_LOAD(SET, KT)                                                                      \
    {                                                                                          \
        const uint32_t koff_ = (KT) * KS;                                                      \
        rR[SET] = rvalid ? *reinterpret_cast<const v4f*>(gR + koff_) : zero4;                  \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                         \
            rQ[SET][j] = __builtin_nontemporal_load(                                           \
                reinterpret_cast<const v4f*>(gQ + (size_t)(RPP * j) * ldq + koff_));           \
    }
#define G32_STORE(SET, BUF)                                                                    \
    {                                                                                          \
        if (has_r) *reinterpret_cast<v4f*>(&sR[BUF][srow][squad * 4]) = rR[SET];               \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                         \
            *reinterpret_cast<v4f*>(&sQ[BUF][srow + RPP * j][squad * 4]) = rQ[SET][j];         \
    }
#define G32_COMPUTE(BUF)                                                                       \
    _Pragma("unroll") for (int g = 0; g < KS / 8; ++g) {                                       \
        const uint32_t kq_ = (2u * g + h) * 4u;                                                \
        v4f a_[RB];                                                                            \
        _Pragma("unroll") for (int r = 0; r < RB; ++r)                                         \
            a_[r] = *reinterpret_cast<const v4f*>(&sR[BUF][r * 32 + l31][kq_]);                \
        const v4f b_ = *reinterpret_cast<const v4f*>(&sQ[BUF][wave * 32 + l31][kq_]);          \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                          \
            _Pragma("unroll") for (int r = 0; r < RB; ++r) {                                   \
                if (DRY) acc[r][t] += a_[r][t] + b_[t];    /* measurement aid: data movement without MFMA */ \
                else acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[r][t], b_[t], acc[r], 0, 0, 0); \
            }                                                                                  \
    }

        // prologue: tiles 0,1,2 in flight; tile 0 -> LDS[0]; tile 3 re-uses set 0
        G32_LOAD(0, 0u)
        if (nk > 1) G32_LOAD(1, 1u)
        if (nk > 2) G32_LOAD(2, 2u)
        __syncthreads();                                        // previous column tile fully consumed
        G32_STORE(0, 0)
        if (nk > 3) G32_LOAD(0, 3u)
        __syncthreads();

        // steady state, unrolled by 3 so that the register sets have fixed names:
        // iteration kt computes tile kt from LDS[kt&1], stores tile kt+1 (set (kt+1)%3) into the
        // other buffer and refills that set with tile kt+4
        uint32_t kt = 0;
        for (; kt + 3 <= nk; kt += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const uint32_t k = kt + (uint32_t)u;
                const int buf = (int)(k & 1u);
                if (buf == 0) { G32_COMPUTE(0) } else { G32_COMPUTE(1) }
                if (k + 1 < nk) {
                    const int set = (u + 1) % 3;
                    if (buf == 0) { G32_STORE(set, 1) } else { G32_STORE(set, 0) }
                    if (k + 4 < nk) G32_LOAD(set, k + 4)
                }
                __syncthreads();
            }
        }
        for (; kt < nk; ++kt) {                                 // nk % 3 leftovers (not hit when K % 96 == 0)
            const int buf = (int)(kt & 1u);
            if (buf == 0) { G32_COMPUTE(0) } else { G32_COMPUTE(1) }
            if (kt + 1 < nk) {
                const uint32_t set = (kt + 1) % 3;
                if (set == 0) { if (buf == 0) { G32_STORE(0, 1) } else { G32_STORE(0, 0) } if (kt + 4 < nk) G32_LOAD(0, kt + 4) }
                else if (set == 1) { if (buf == 0) { G32_STORE(1, 1) } else { G32_STORE(1, 0) } if (kt + 4 < nk) G32_LOAD(1, kt + 4) }
                else { if (buf == 0) { G32_STORE(2, 1) } else { G32_STORE(2, 0) } if (kt + 4 < nk) G32_LOAD(2, kt + 4) }
            }
            __syncthreads();
        }
#undef G32_LOAD
#undef G32_STORE
#undef G32_COMPUTE

        const uint32_t col = bn * HN + wave * 32 + l31;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t row = (uint32_t)r * 32u + (e & 3) + 8 * (e >> 2) + 4 * h;
                const uint32_t dr = drows[row];
                if (dr != 0xffffffffu) D[(size_t)dr * ldd + col] = acc[r][e];
            }
    }
    if (pdbg != nullptr && tid == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const uint64_t e = atomicAdd(reinterpret_cast<unsigned long long*>(pdbg), 1ull);
        if (e < 4096ull) {
            uint64_t* o = pdbg + 1 + 4 * e;
            o[0] = t_dbg0; o[1] = wall_clock64(); o[2] = ((uint64_t)(xcc & 0xfu) << 16) | (hw & 0xffffu);
            o[3] = (uint64_t)blockIdx.x | ((uint64_t)ntiles << 32);
        }
    }
}

// ---- deep-ring form: the same tile and arithmetic, RING K-steps of global loads in flight -----------
// A CU has to keep ~8 TB/s x latency / 256 CUs ≈ 64-96 KiB of HBM requests outstanding; three
// 36-KiB sets that advance in barrier lock-step are at the edge of that.  Here the ring is RING sets
// deep (RING even and nk % RING == 0, so LDS buffer and register set are compile-time per unrolled
// step; ldm is a multiple of 256, hence nk of 8).  Same summation order as k_gemm32_tn_f32: bitwise
// identical output.
template <int HN, int HT, int RING>
__global__ __launch_bounds__(HT, 1)
void k_gemm32r_tn_f32(const float* __restrict__ At, const uint32_t* __restrict__ rcols,
                      const uint32_t* __restrict__ drows, float* __restrict__ D,
                      uint32_t K, uint32_t ldq, uint32_t ldd, uint32_t ntiles,
                      const DevState* __restrict__ st)
{
    static_assert(RING % 2 == 0, "ring depth must be even");
    if (st != nullptr && (st->done != 0 || st->need_sweep != 1)) return;   // no sweep needed this round
    __shared__ __attribute__((aligned(16))) float sR[2][HM][GLD];
    __shared__ __attribute__((aligned(16))) float sQ[2][HN][GLD];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t h = lane >> 5, l31 = lane & 31u;
    constexpr int RPP = HT / 8;                                 // rows staged per pass
    constexpr int NJ = HN / RPP;                                // passes per column tile
    const uint32_t srow = tid >> 3, squad = tid & 7u;
    const bool has_r = tid < 256;

    const uint32_t rc = rcols[srow & 31u];
    const bool rvalid = has_r && rc != 0xffffffffu;
    const float* gR = At + (size_t)(rvalid ? rc : 0u) * ldq + squad * 4;
    const v4f zero4 = { 0.f, 0.f, 0.f, 0.f };
    const uint32_t nk = K / GK;

    for (uint32_t bn = blockIdx.x; bn < ntiles; bn += gridDim.x) {
        const float* gQ = At + (size_t)(bn * HN + srow) * ldq + squad * 4;
        v16f acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;

        v4f rR[RING], rQ[RING][NJ];
#define R32_LOAD(SET, KT)                                                                      \
    {                                                                                          \
        const uint32_t koff_ = (KT) * GK;                                                      \
        rR[SET] = rvalid ? *reinterpret_cast<const v4f*>(gR + koff_) : zero4;                  \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                         \
            rQ[SET][j] = __builtin_nontemporal_load(                                           \
                reinterpret_cast<const v4f*>(gQ + (size_t)(RPP * j) * ldq + koff_));           \
    }
#define R32_STORE(SET, BUF)                                                                    \
    {                                                                                          \
        if (has_r) *reinterpret_cast<v4f*>(&sR[BUF][srow][squad * 4]) = rR[SET];               \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                         \
            *reinterpret_cast<v4f*>(&sQ[BUF][srow + RPP * j][squad * 4]) = rQ[SET][j];         \
    }
#pragma unroll
        for (int s = 0; s < RING; ++s) R32_LOAD(s, (uint32_t)s)
        __syncthreads();                                        // previous column tile fully consumed
        R32_STORE(0, 0)
        if ((uint32_t)RING < nk) R32_LOAD(0, (uint32_t)RING)
        __syncthreads();

        for (uint32_t kt = 0; kt < nk; kt += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                const uint32_t k = kt + (uint32_t)u;
                const int buf = u & 1;
#pragma unroll
                for (int g = 0; g < GK / 8; ++g) {
                    const uint32_t kq = (2u * g + h) * 4u;
                    const v4f a = *reinterpret_cast<const v4f*>(&sR[buf][l31][kq]);
                    const v4f b = *reinterpret_cast<const v4f*>(&sQ[buf][wave * 32 + l31][kq]);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
                }
                if (k + 1 < nk) {
                    const int set = (u + 1) % RING;
                    R32_STORE(set, buf ^ 1)
                    if (k + 1 + RING < nk) R32_LOAD(set, k + 1 + RING)
                }
                __syncthreads();
            }
        }
#undef R32_LOAD
#undef R32_STORE

        const uint32_t col = bn * HN + wave * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t row = (e & 3) + 8 * (e >> 2) + 4 * h;
            const uint32_t dr = drows[row];
            if (dr != 0xffffffffu) D[(size_t)dr * ldd + col] = acc[e];
        }
    }
}

// ---- barrier-free form of the fp32 pass -------------------------------------------------------------
// Every wave is on its own: it streams ITS 32 columns of A (coalesced: 8 lanes per 128-B row
// segment, 8 rows per instruction) and its own copy of the 32 right-hand-side rows (L2 hits) three
// K-steps ahead in registers, turns each K-step into MFMA layout through a wave-private LDS tile
// (write, wait, read back as k-quads) and issues its 16 MFMAs.  No workgroup barrier anywhere, so a
// wave waiting for memory never stalls the other waves of its SIMD.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2)
void k_gemm32w_tn_f32(const float* __restrict__ At, const uint32_t* __restrict__ rcols,
                      const uint32_t* __restrict__ drows, float* __restrict__ D,
                      uint32_t K, uint32_t ldq, uint32_t ldd, uint32_t ntiles,
                      const DevState* __restrict__ st)
{
    if (st != nullptr && (st->done != 0 || st->need_sweep != 1)) return;   // no sweep needed this round
    if (rcols[0] == 0xffffffffu) return;                                    // an empty list (lists are filled from entry 0)
    __shared__ __attribute__((aligned(16))) float sT[WAVES][2][32][GLD];    // per wave: R tile, Q tile

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t h = lane >> 5, l31 = lane & 31u;
    const uint32_t r8 = lane >> 3, quad = lane & 7u;            // staging: rows r8 + 8j, k-quad `quad`
    float (*sR)[GLD] = sT[wave][0];
    float (*sQ)[GLD] = sT[wave][1];
    const v4f zero4 = { 0.f, 0.f, 0.f, 0.f };
    const uint32_t nk = K / GK;

    const float* gR[4];
    bool rvalid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t rc = rcols[r8 + 8 * j];
        rvalid[j] = rc != 0xffffffffu;
        gR[j] = At + (size_t)(rvalid[j] ? rc : 0u) * ldq + quad * 4;
    }

    for (uint32_t tile = blockIdx.x * WAVES + wave; tile < ntiles; tile += gridDim.x * WAVES) {
        const float* gQ = At + (size_t)(tile * 32 + r8) * ldq + quad * 4;
        v16f acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        v4f rR[3][4], rQ[3][4];
#define W32_LOAD(SET, KT)                                                                      \
    {                                                                                          \
        const uint32_t koff_ = (KT) * GK;                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                        \
            rR[SET][j] = rvalid[j] ? *reinterpret_cast<const v4f*>(gR[j] + koff_) : zero4;     \
            rQ[SET][j] = __builtin_nontemporal_load(                                           \
                reinterpret_cast<const v4f*>(gQ + (size_t)(8 * j) * ldq + koff_));             \
        }                                                                                      \
    }
#define W32_STEP(SET)                                                                          \
    {                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                        \
            *reinterpret_cast<v4f*>(&sR[r8 + 8 * j][quad * 4]) = rR[SET][j];                   \
            *reinterpret_cast<v4f*>(&sQ[r8 + 8 * j][quad * 4]) = rQ[SET][j];                   \
        }                                                                                      \
        __builtin_amdgcn_wave_barrier();                                                       \
        v4f a_[GK / 8], b_[GK / 8];                                                            \
        _Pragma("unroll") for (int g = 0; g < GK / 8; ++g) {                                   \
            const uint32_t kq_ = (2u * g + h) * 4u;                                            \
            a_[g] = *reinterpret_cast<const v4f*>(&sR[l31][kq_]);                              \
            b_[g] = *reinterpret_cast<const v4f*>(&sQ[l31][kq_]);                              \
        }                                                                                      \
        __builtin_amdgcn_wave_barrier();                                                       \
        _Pragma("unroll") for (int g = 0; g < GK / 8; ++g)                                     \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                      \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[g][t], b_[g][t], acc, 0, 0, 0);  \
    }

        W32_LOAD(0, 0u)
        if (nk > 1) W32_LOAD(1, 1u)
        if (nk > 2) W32_LOAD(2, 2u)
        uint32_t kt = 0;
        for (; kt + 3 <= nk; kt += 3) {
            W32_STEP(0)
            if (kt + 3 < nk) W32_LOAD(0, kt + 3)
            W32_STEP(1)
            if (kt + 4 < nk) W32_LOAD(1, kt + 4)
            W32_STEP(2)
            if (kt + 5 < nk) W32_LOAD(2, kt + 5)
        }
        if (kt < nk) { W32_STEP(0) ++kt; }
        if (kt < nk) { W32_STEP(1) ++kt; }
#undef W32_LOAD
#undef W32_STEP

        const uint32_t col = tile * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t row = (e & 3) + 8 * (e >> 2) + 4 * h;
            const uint32_t dr = drows[row];
            if (dr != 0xffffffffu) D[(size_t)dr * ldd + col] = acc[e];
        }
    }
}

// ---- the early form's pass: barrier-free, one 32-column tile per single-wave workgroup, THREE waves per SIMD ----
// Runs beside a solo launch that holds one CU (homotopy.hip, early form).  k_gemm32w_tn_f32 needs 210 VGPRs: two
// waves per SIMD, i.e. exactly the 2048 wave slots of an empty chip for the 2048 tiles of C2 — with one CU taken
// four tiles found no slot and ran alone afterwards, one latency-bound wave each (0.64 ms per pass instead of 0.42).
// Here the register ring is two K-steps deep and the k-groups of a step are consumed in two halves: <= 168 VGPRs,
// three waves per SIMD (12 per CU carry as many bytes in flight as 8 with the deeper ring), every tile resident
// from the start.  Same k-order of the accumulation as every other tiling of the pass: bitwise identical output.
__global__ __launch_bounds__(64, 3)
void k_gemm32e_tn_f32(const float* __restrict__ At, const uint32_t* __restrict__ rcols,
                      const uint32_t* __restrict__ drows, float* __restrict__ D,
                      uint32_t K, uint32_t ldq, uint32_t ldd, uint32_t ntiles,
                      const DevState* __restrict__ st)
{
    if (st != nullptr && (st->done != 0 || st->need_sweep != 1)) return;   // no sweep needed this round
    if (rcols[0] == 0xffffffffu) return;                                    // an empty list (lists are filled from entry 0)
    __shared__ __attribute__((aligned(16))) float sT[2][32][GLD];           // R tile, Q tile

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t h = lane >> 5, l31 = lane & 31u;
    const uint32_t r8 = lane >> 3, quad = lane & 7u;            // staging: rows r8 + 8j, k-quad `quad`
    float (*sR)[GLD] = sT[0];
    float (*sQ)[GLD] = sT[1];
    const v4f zero4 = { 0.f, 0.f, 0.f, 0.f };
    const uint32_t nk = K / GK;

    const float* gR[4];
    bool rvalid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t rc = rcols[r8 + 8 * j];
        rvalid[j] = rc != 0xffffffffu;
        gR[j] = At + (size_t)(rvalid[j] ? rc : 0u) * ldq + quad * 4;
    }

    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const float* gQ = At + (size_t)(tile * 32 + r8) * ldq + quad * 4;
        v16f acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        v4f rR[2][4], rQ[2][4];
#define E32_LOAD(SET, KT)                                                                      \
    {                                                                                          \
        const uint32_t koff_ = (KT) * GK;                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                        \
            rR[SET][j] = rvalid[j] ? *reinterpret_cast<const v4f*>(gR[j] + koff_) : zero4;     \
            rQ[SET][j] = __builtin_nontemporal_load(                                           \
                reinterpret_cast<const v4f*>(gQ + (size_t)(8 * j) * ldq + koff_));             \
        }                                                                                      \
    }
#define E32_STEP(SET)                                                                          \
    {                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                        \
            *reinterpret_cast<v4f*>(&sR[r8 + 8 * j][quad * 4]) = rR[SET][j];                   \
            *reinterpret_cast<v4f*>(&sQ[r8 + 8 * j][quad * 4]) = rQ[SET][j];                   \
        }                                                                                      \
        __builtin_amdgcn_wave_barrier();                                                       \
        _Pragma("unroll") for (int gh = 0; gh < GK / 8; gh += 2) {                             \
            v4f a_[2], b_[2];                                                                  \
            _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                    \
                const uint32_t kq_ = (2u * (gh + g) + h) * 4u;                                 \
                a_[g] = *reinterpret_cast<const v4f*>(&sR[l31][kq_]);                          \
                b_[g] = *reinterpret_cast<const v4f*>(&sQ[l31][kq_]);                          \
            }                                                                                  \
            _Pragma("unroll") for (int g = 0; g < 2; ++g)                                      \
                _Pragma("unroll") for (int t = 0; t < 4; ++t)                                  \
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[g][t], b_[g][t], acc, 0, 0, 0); \
        }                                                                                      \
        __builtin_amdgcn_wave_barrier();                                                       \
    }
        E32_LOAD(0, 0u)
        if (nk > 1) E32_LOAD(1, 1u)
        uint32_t kt = 0;
        for (; kt + 2 <= nk; kt += 2) {
            E32_STEP(0)
            if (kt + 2 < nk) E32_LOAD(0, kt + 2)
            E32_STEP(1)
            if (kt + 3 < nk) E32_LOAD(1, kt + 3)
        }
        if (kt < nk) { E32_STEP(0) ++kt; }
#undef E32_LOAD
#undef E32_STEP

        const uint32_t col = tile * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t row = (e & 3) + 8 * (e >> 2) + 4 * h;
            const uint32_t dr = drows[row];
            if (dr != 0xffffffffu) D[(size_t)dr * ldd + col] = acc[e];
        }
    }
}

// ---- the same pass in fp64 (engine 1 for double): v_mfma_f64_16x16x4_f64 -------------------------
// 32 right-hand sides x 256 columns per 512-thread workgroup, K-step 16 doubles (128 B per row, the
// same bytes per step as the fp32 kernel): wave w owns 32 columns as 2 x 2 tiles of 16 x 16.  Per
// K-step a wave issues 16 MFMAs (4 k-groups of 4 x 4 tiles).  At C5 (16384 x 131072) the pass moves
// 16 GiB (2.15 ms at 8 TB/s) and needs 137 GFLOP (1.75 ms at the 78.6 TFLOP/s fp64 peak): HBM-bound
// on paper, close to both roofs in practice — like its fp32 sibling.
// MFMA operand layout (CDNA3 ISA, 16x16x4 f64): A: lane -> (row lane%16, k lane/16); B: lane ->
// (k lane/16, col lane%16); D (4 doubles per lane): element j -> (row 4*j + lane/16, col lane%16)
// (checked against numpy: test_gram_cols_vs_numpy[float64]).
typedef double v2d_t __attribute__((ext_vector_type(2)));
typedef double v4d_t __attribute__((ext_vector_type(4)));
constexpr int DK = 16;                       // doubles per K-step
constexpr int DLD = DK + 2;                  // LDS row pitch in doubles (144 B, as in the fp32 kernel)

// RH = 32 or 64 right-hand sides.  In double precision the 32-column pass is already close to both roofs (137
// GFLOP against 78.6 TFLOP/s = 1.75 ms of matrix-core time inside a 3.4 ms pass); 64 columns double the flops
// (MFMA-bound: ~3.5 ms at the peak) but not the bytes, so one wide pass costs little more than one narrow pass and
// replaces two of them.
// HN / HT / BPC: 256 columns per 512-thread workgroup, one per CU (default), or 128 columns per 256-thread workgroup,
// two or three per CU — one workgroup's barrier waits are then another's MFMAs (SQ counters, DESIGN.md §3.9: with one
// workgroup per CU a third of the wave cycles of the 32-column pass are parked at the barrier).
template <int RH, int HN = 256, int HT = 512, int BPC = 1>
__global__ __launch_bounds__(HT, BPC)
void k_gemm32_tn_f64(const double* __restrict__ At, const uint32_t* __restrict__ rcols,
                     const uint32_t* __restrict__ drows, double* __restrict__ D,
                     uint32_t K, uint32_t ldq, uint32_t ldd, uint32_t ntiles,
                     const DevState* __restrict__ st, uint32_t part_stride = 0)
{
    if (st != nullptr && (st->done != 0 || st->need_sweep != 1)) return;   // no sweep needed this round
    static_assert(HN / 32 == HT / 64 && RH * 8 <= HT, "one 32-column block per wave; the R tile is staged in one pass");
    // part_stride != 0 (narrow dictionaries: too few column tiles to fill the chip): the rows are split over gridDim.y
    // workgroups — this one takes rows blockIdx.y * K .. + K and stores its partial sums, indexed by right-hand side, at
    // D + blockIdx.y * part_stride; k_gemm_f64_sum adds them up in order
    if (part_stride != 0u) { At += (size_t)blockIdx.y * K; D += (size_t)blockIdx.y * part_stride; }
    constexpr int RB = RH / 16;                                 // 16-row blocks of right-hand sides (2 or 4)
    __shared__ __attribute__((aligned(16))) double sR[2][RH][DLD];
    __shared__ __attribute__((aligned(16))) double sQ[2][HN][DLD];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t l15 = lane & 15u, kq = lane >> 4;           // MFMA: row / column within the tile, k within the group
    constexpr int RPP = HT / 8;                                 // rows staged per pass (64)
    constexpr int NJ = HN / RPP;                                // passes per column tile (4)
    const uint32_t srow = tid >> 3, spair = tid & 7u;          // staging: rows srow + RPP*j, doubles 2*spair, 2*spair+1
    const bool has_r = tid < RH * 8;                            // R tile: RH rows x 8 pairs

    const uint32_t rc = rcols[srow % (uint32_t)RH];
    const bool rvalid = has_r && rc != 0xffffffffu;
    const double* gR = At + (size_t)(rvalid ? rc : 0u) * ldq + spair * 2;
    const v2d_t zero2 = { 0.0, 0.0 };
    const uint32_t nk = K / DK;

    for (uint32_t bn = blockIdx.x; bn < ntiles; bn += gridDim.x) {
        const double* gQ = At + (size_t)(bn * HN + srow) * ldq + spair * 2;
        v4d_t acc[RB][2];
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = v4d_t{ 0.0, 0.0, 0.0, 0.0 };

        v2d_t rR[RH > 32 ? 2 : 3], rQ[RH > 32 ? 2 : 3][NJ];
#define D32_LOAD(SET, KT)                                                                      \
    {                                                                                          \
        const uint32_t koff_ = (KT) * DK;                                                      \
        rR[SET] = rvalid ? *reinterpret_cast<const v2d_t*>(gR + koff_) : zero2;                \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                         \
            rQ[SET][j] = __builtin_nontemporal_load(                                           \
                reinterpret_cast<const v2d_t*>(gQ + (size_t)(RPP * j) * ldq + koff_));         \
    }
#define D32_STORE(SET, BUF)                                                                    \
    {                                                                                          \
        if (has_r) *reinterpret_cast<v2d_t*>(&sR[BUF][srow][spair * 2]) = rR[SET];             \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                         \
            *reinterpret_cast<v2d_t*>(&sQ[BUF][srow + RPP * j][spair * 2]) = rQ[SET][j];       \
    }
#define D32_COMPUTE(BUF)                                                                       \
    _Pragma("unroll") for (int g = 0; g < DK / 4; ++g) {                                       \
        const uint32_t k_ = 4u * g + kq;                                                       \
        const double b0_ = sQ[BUF][wave * 32 + l15][k_], b1_ = sQ[BUF][wave * 32 + 16 + l15][k_]; \
        _Pragma("unroll") for (int i = 0; i < RB; ++i) {                                       \
            const double a_ = sR[BUF][16 * i + l15][k_];                                       \
            acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b0_, acc[i][0], 0, 0, 0);     \
            acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b1_, acc[i][1], 0, 0, 0);     \
        }                                                                                      \
    }

        if (RH > 32) {
            // 64 right-hand sides: eight accumulators of four doubles per lane leave room for a ring of TWO K-steps
            // (three spilled 84 registers to scratch: 7.9 ms per pass)
            D32_LOAD(0, 0u)
            if (nk > 1) D32_LOAD(1, 1u)
            __syncthreads();                                    // previous column tile fully consumed
            D32_STORE(0, 0)
            if (nk > 2) D32_LOAD(0, 2u)
            __syncthreads();
            uint32_t kt = 0;
            for (; kt + 2 <= nk; kt += 2) {
                D32_COMPUTE(0)
                D32_STORE(1, 1)                                 // (kt + 1 < nk holds here)
                if (kt + 3 < nk) D32_LOAD(1, kt + 3)
                __syncthreads();
                D32_COMPUTE(1)
                if (kt + 2 < nk) {
                    D32_STORE(0, 0)
                    if (kt + 4 < nk) D32_LOAD(0, kt + 4)
                }
                __syncthreads();
            }
            if (kt < nk) {                                      // odd number of K-steps (not hit: ldm is a multiple of 256)
                D32_COMPUTE(0)
                __syncthreads();
            }
        } else {
        D32_LOAD(0, 0u)
        if (nk > 1) D32_LOAD(1, 1u)
        if (nk > 2) D32_LOAD(2, 2u)
        __syncthreads();                                        // previous column tile fully consumed
        D32_STORE(0, 0)
        if (nk > 3) D32_LOAD(0, 3u)
        __syncthreads();

        uint32_t kt = 0;
        for (; kt + 3 <= nk; kt += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const uint32_t k = kt + (uint32_t)u;
                const int buf = (int)(k & 1u);
                if (buf == 0) { D32_COMPUTE(0) } else { D32_COMPUTE(1) }
                if (k + 1 < nk) {
                    const int set = (u + 1) % 3;
                    if (buf == 0) { D32_STORE(set, 1) } else { D32_STORE(set, 0) }
                    if (k + 4 < nk) D32_LOAD(set, k + 4)
                }
                __syncthreads();
            }
        }
        for (; kt < nk; ++kt) {                                 // nk % 3 leftovers
            const int buf = (int)(kt & 1u);
            if (buf == 0) { D32_COMPUTE(0) } else { D32_COMPUTE(1) }
            if (kt + 1 < nk) {
                const uint32_t set = (kt + 1) % 3;
                if (set == 0) { if (buf == 0) { D32_STORE(0, 1) } else { D32_STORE(0, 0) } if (kt + 4 < nk) D32_LOAD(0, kt + 4) }
                else if (set == 1) { if (buf == 0) { D32_STORE(1, 1) } else { D32_STORE(1, 0) } if (kt + 4 < nk) D32_LOAD(1, kt + 4) }
                else { if (buf == 0) { D32_STORE(2, 1) } else { D32_STORE(2, 0) } if (kt + 4 < nk) D32_LOAD(2, kt + 4) }
            }
            __syncthreads();
        }
        }
#undef D32_LOAD
#undef D32_STORE
#undef D32_COMPUTE

#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t col = bn * HN + wave * 32 + 16 * j + l15;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t row = 16 * i + 4 * e + kq;
                    const uint32_t dr = drows[row];
                    if (dr != 0xffffffffu) D[(size_t)(part_stride != 0u ? row : dr) * ldd + col] = acc[i][j][e];
                }
            }
    }
}

// D[drows[s]][col] = sum over the row chunks, in order, of the partial sums of a split pass
__global__ __launch_bounds__(256)
void k_gemm_f64_sum(const double* __restrict__ part, uint32_t nsplit, uint32_t part_stride, uint32_t ncols, uint32_t nrhs,
                    const uint32_t* __restrict__ drows, double* __restrict__ D, uint32_t ldd, const DevState* __restrict__ st)
{
    if (st != nullptr && (st->done != 0 || st->need_sweep != 1)) return;
    const uint32_t col = blockIdx.x * 256u + threadIdx.x, s = blockIdx.y;
    if (col >= ncols || s >= nrhs) return;
    const uint32_t dr = drows[s];
    if (dr == 0xffffffffu) return;
    double acc = part[(size_t)s * ncols + col];
    for (uint32_t c = 1; c < nsplit; ++c) acc += part[(size_t)c * part_stride + (size_t)s * ncols + col];
    D[(size_t)dr * ldd + col] = acc;
}

// narrow dictionaries (option pass_ksplit > 1: the sub-context of the fp64 screened form): RH right-hand sides, rows split
template <int RH>
static hipError_t launch_gemm_split_f64(const ss_hip_ctx* ctx, const uint32_t* rcols, const uint32_t* drows, double* D, uint32_t ldd,
                                        const DevState* st)
{
    const uint32_t ns = (uint32_t)ctx->pass_ksplit, np = ctx->n_pad;
    const uint32_t ntiles = np / 256, Kc = ctx->ldm / ns;
    const uint32_t stride = 64u * np;                            // doubles per row chunk (64 right-hand sides at most)
    double* part = static_cast<double*>(ctx->pass_part);
    hipLaunchKernelGGL((k_gemm32_tn_f64<RH>), dim3(ntiles, ns), dim3(512), 0, ctx->stream, static_cast<const double*>(ctx->At),
                       rcols, drows, part, Kc, ctx->ldm, np, ntiles, st, stride);
    hipLaunchKernelGGL(k_gemm_f64_sum, dim3((np + 255) / 256, RH), dim3(256), 0, ctx->stream, (const double*)part, ns, stride, np, (uint32_t)RH,
                       drows, D, ldd, st);
    return hipGetLastError();
}

hipError_t launch_gemm32_tn_f64(const ss_hip_ctx* ctx, const uint32_t* rcols, const uint32_t* drows,
                                double* D, uint32_t ldd, const DevState* st)
{
    if (ctx->n_pad % 256 != 0 || ctx->ldm % DK != 0) return hipErrorInvalidValue;
    if (ctx->pass_ksplit > 1 && ctx->pass_part != nullptr) return launch_gemm_split_f64<32>(ctx, rcols, drows, D, ldd, st);
    if (ctx->sweep_f64_variant == 1 || ctx->sweep_f64_variant == 2) {
        // 128-column tiles, 256 threads, two or three workgroups per CU
        const uint32_t nt = ctx->n_pad / 128;
        const uint32_t cap = (ctx->sweep_f64_variant == 1 ? 2u : 3u) * (uint32_t)ctx->num_cus;
        const uint32_t g = nt < cap ? nt : cap;
        if (ctx->sweep_f64_variant == 1)
            hipLaunchKernelGGL((k_gemm32_tn_f64<32, 128, 256, 2>), dim3(g), dim3(256), 0, ctx->stream, static_cast<const double*>(ctx->At),
                               rcols, drows, D, ctx->ldm, ctx->ldm, ldd, nt, st);
        else
            hipLaunchKernelGGL((k_gemm32_tn_f64<32, 128, 256, 3>), dim3(g), dim3(256), 0, ctx->stream, static_cast<const double*>(ctx->At),
                               rcols, drows, D, ctx->ldm, ctx->ldm, ldd, nt, st);
        return hipGetLastError();
    }
    const uint32_t ntiles = ctx->n_pad / 256;
    const uint32_t grid = ntiles < (uint32_t)ctx->num_cus ? ntiles : (uint32_t)ctx->num_cus;
    hipLaunchKernelGGL(k_gemm32_tn_f64<32>, dim3(grid), dim3(512), 0, ctx->stream, static_cast<const double*>(ctx->At),
                       rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st);
    return hipGetLastError();
}

// the 64-column pass in double precision (rcols / drows hold 64 entries each)
hipError_t launch_gemm64_tn_f64(const ss_hip_ctx* ctx, const uint32_t* rcols, const uint32_t* drows,
                                double* D, uint32_t ldd, const DevState* st)
{
    if (ctx->n_pad % 256 != 0 || ctx->ldm % DK != 0) return hipErrorInvalidValue;
    if (ctx->pass_ksplit > 1 && ctx->pass_part != nullptr) return launch_gemm_split_f64<64>(ctx, rcols, drows, D, ldd, st);
    const uint32_t ntiles = ctx->n_pad / 256;
    const uint32_t grid = ntiles < (uint32_t)ctx->num_cus ? ntiles : (uint32_t)ctx->num_cus;
    hipLaunchKernelGGL(k_gemm32_tn_f64<64>, dim3(grid), dim3(512), 0, ctx->stream, static_cast<const double*>(ctx->At),
                       rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st);
    return hipGetLastError();
}

hipError_t set_pass_debug(uint64_t* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_pass_dbg), &buf, sizeof(buf)); }

// D[drows[s]][:] = At · At[rcols[s]][:] for s < 32 (entries 0xffffffff are skipped)
hipError_t launch_gemm32_tn_f32(const ss_hip_ctx* ctx, const uint32_t* rcols, const uint32_t* drows,
                                float* D, uint32_t ldd, const DevState* st)
{
    if (ctx->n_pad % 256 != 0 || ctx->ldm % GK != 0) return hipErrorInvalidValue;
    const float* At = static_cast<const float*>(ctx->At);
    if (ctx->sweep32_variant == 6 || ctx->sweep32_variant == 7) {
        // deep register ring (4 or 8 K-steps of loads in flight); nk = ldm / 32 is a multiple of 8
        const uint32_t ntiles = ctx->n_pad / 256;
        const uint32_t grid = ntiles < (uint32_t)ctx->num_cus ? ntiles : (uint32_t)ctx->num_cus;
        if (ctx->ldm % 256 != 0) return hipErrorInvalidValue;
        if (ctx->sweep32_variant == 6)
            hipLaunchKernelGGL((k_gemm32r_tn_f32<256, 512, 4>), dim3(grid), dim3(512), 0, ctx->stream,
                               At, rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st);
        else
            hipLaunchKernelGGL((k_gemm32r_tn_f32<256, 512, 8>), dim3(grid), dim3(512), 0, ctx->stream,
                               At, rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st);
    } else if (ctx->sweep32_variant == 5) {
        // measurement aid: the same data movement (global -> registers -> LDS -> registers) without MFMA
        const uint32_t ntiles = ctx->n_pad / 256;
        const uint32_t grid = ntiles < (uint32_t)ctx->num_cus ? ntiles : (uint32_t)ctx->num_cus;
        hipLaunchKernelGGL((k_gemm32_tn_f32<256, 512, 1, GK, true>), dim3(grid), dim3(512), 0, ctx->stream,
                           At, rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st, 0u, nullptr, 2u);
    } else if (ctx->sweep32_variant == 4) {
        // K-step 64: 256 contiguous bytes per row and step (fewer, larger DRAM bursts per stream)
        const uint32_t ntiles = ctx->n_pad / 256;
        const uint32_t grid = ntiles < (uint32_t)ctx->num_cus ? ntiles : (uint32_t)ctx->num_cus;
        if (ctx->ldm % 64 != 0) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_gemm32_tn_f32<256, 512, 1, 64>), dim3(grid), dim3(512), 0, ctx->stream,
                           At, rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st, 0u, nullptr, 2u);
    } else if (ctx->sweep32_variant == 8 || ctx->sweep32_variant == 9) {
        // measurement aid: the early form's pass (one 32-column tile per single-wave workgroup), or two-wave workgroups
        const uint32_t ntiles = ctx->n_pad / 32;
        if (ctx->sweep32_variant == 8)
            hipLaunchKernelGGL((k_gemm32e_tn_f32), dim3(ntiles), dim3(64), 0, ctx->stream, At, rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st);
        else
            hipLaunchKernelGGL((k_gemm32w_tn_f32<2>), dim3((ntiles + 1) / 2), dim3(128), 0, ctx->stream, At, rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st);
    } else if (ctx->sweep32_variant == 3) {
        // barrier-free: 32-column tiles, one per wave, 4 waves per workgroup, 2 workgroups per CU
        const uint32_t ntiles = ctx->n_pad / 32;
        const uint32_t cap = 2u * (uint32_t)ctx->num_cus;
        const uint32_t want = (ntiles + 3u) / 4u;
        hipLaunchKernelGGL((k_gemm32w_tn_f32<4>), dim3(want < cap ? want : cap), dim3(256), 0, ctx->stream,
                           At, rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st);
    } else if (ctx->sweep32_variant == 0) {
        // one 256-column tile per workgroup of 512 threads, one workgroup per CU
        const uint32_t ntiles = ctx->n_pad / 256;
        const uint32_t grid = ntiles < (uint32_t)ctx->num_cus ? ntiles : (uint32_t)ctx->num_cus;
        hipLaunchKernelGGL((k_gemm32_tn_f32<256, 512, 1>), dim3(grid), dim3(512), 0, ctx->stream,
                           At, rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st, 0u, nullptr, 2u);
    } else {
        // 128-column tiles, 256 threads, several workgroups per CU: one's barrier waits are
        // another's compute
        const uint32_t ntiles = ctx->n_pad / 128;
        const uint32_t per_cu = ctx->sweep32_variant == 1 ? 2u : 3u;
        const uint32_t cap = per_cu * (uint32_t)ctx->num_cus;
        const uint32_t grid = ntiles < cap ? ntiles : cap;
        if (ctx->sweep32_variant == 1)
            hipLaunchKernelGGL((k_gemm32_tn_f32<128, 256, 2>), dim3(grid), dim3(256), 0, ctx->stream,
                               At, rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st, 0u, nullptr, 2u);
        else
            hipLaunchKernelGGL((k_gemm32_tn_f32<128, 256, 3>), dim3(grid), dim3(256), 0, ctx->stream,
                               At, rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st, 0u, nullptr, 2u);
    }
    return hipGetLastError();
}

// the barrier-free 32-column pass with one 32-column tile per single-wave workgroup, on a given stream, ungated
// (early form of the speculative engine: it runs beside a solo launch that occupies one CU — 2048 small workgroups
// spread evenly over whatever CUs are free, where 256 one-per-CU workgroups would leave one waiting for a CU)
// (10 KB of unused dynamic LDS on top of the kernel's 46 KB: at most TWO of these workgroups fit a CU, which is what
// makes the placement below the hardware's only choice)
constexpr uint32_t kSePad = 10240;

hipError_t launch_gemm32se_on(const ss_hip_ctx* ctx, hipStream_t on, const uint32_t* rcols, const uint32_t* drows, float* D, uint32_t ldd,
                              uint32_t first, uint32_t ntiles, const DevState* st, uint32_t* se_count, uint32_t quota)
{
    if (ctx->n_pad % 128 != 0 || ctx->ldm % GK != 0 || st == nullptr || se_count == nullptr) return hipErrorInvalidValue;
    // (option early_se = 2, tests: ONE workgroup per SE, so that half the tiles are left to the fix-up launch)
    // quota: tiles per shader engine in all (2: one per workgroup of this launch; more: they come back for the next)
    hipLaunchKernelGGL((k_gemm32_tn_f32<128, 256, 3, GK, false, HM, true>), dim3(early_se_wgs(ctx)), dim3(256), kSePad, on,
                       static_cast<const float*>(ctx->At), rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st, first, se_count, quota);
    // ... and whatever that left undone (nothing if the workgroups were dealt out as assumed: these then leave at once)
    if (ntiles > first)
        hipLaunchKernelGGL((k_gemm32_tn_f32<128, 256, 3, GK, false, HM, true, true>), dim3(ntiles - first), dim3(256), 0, on,
                           static_cast<const float*>(ctx->At), rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st, first, se_count, quota);
    return hipGetLastError();
}

// tiles [first, last) of a pass, one workgroup each, at most two per CU (the partial round a wide dictionary leaves)
hipError_t launch_gemm32range_on(const ss_hip_ctx* ctx, hipStream_t on, const uint32_t* rcols, const uint32_t* drows, float* D, uint32_t ldd,
                                 uint32_t first, uint32_t last)
{
    if (ctx->n_pad % 128 != 0 || ctx->ldm % GK != 0 || last <= first) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_gemm32_tn_f32<128, 256, 3>), dim3(last - first), dim3(256), kSePad, on, static_cast<const float*>(ctx->At),
                       rcols, drows, D, ctx->ldm, ctx->ldm, ldd, last, (const DevState*)nullptr, first, nullptr, 2u);
    return hipGetLastError();
}

hipError_t launch_gemm32w_on(const ss_hip_ctx* ctx, hipStream_t on, const uint32_t* rcols, const uint32_t* drows, float* D, uint32_t ldd,
                             uint32_t tiles128)
{
    if (ctx->n_pad % 32 != 0 || ctx->ldm % GK != 0) return hipErrorInvalidValue;
    if (tiles128 != 0) {
        // the first tiles128 tiles only, at most two workgroups per CU (the caller covers the other columns: early_prologue)
        hipLaunchKernelGGL((k_gemm32_tn_f32<128, 256, 3>), dim3(tiles128), dim3(256), kSePad, on, static_cast<const float*>(ctx->At),
                           rcols, drows, D, ctx->ldm, ctx->ldm, ldd, tiles128, (const DevState*)nullptr, 0u, nullptr, 2u);
        return hipGetLastError();
    }
    // (also with fewer tiles than CUs: a single-wave workgroup of the tiling below streams its 32 columns at one wave's
    // pace, 0.43 ms whatever the width of the dictionary — measured 1.39 against 1.59 ms per solve at n = 24 000)
    if (ctx->early_pass == 2 && ctx->n_pad % 128 == 0) {
        // LDS-staged tiling with 128-column tiles, three 256-thread workgroups per CU (46 KB LDS each): 765 slots on the
        // 255 CUs the solo launch leaves, every one of the 512 tiles of C2 resident from the start
        const uint32_t nt = ctx->n_pad / 128;
        const uint32_t cap = 3u * (uint32_t)ctx->num_cus;
        hipLaunchKernelGGL((k_gemm32_tn_f32<128, 256, 3>), dim3(nt < cap ? nt : cap), dim3(256), 0, on, static_cast<const float*>(ctx->At),
                           rcols, drows, D, ctx->ldm, ctx->ldm, ldd, nt, (const DevState*)nullptr, 0u, nullptr, 2u);
        return hipGetLastError();
    }
    const uint32_t ntiles = ctx->n_pad / 32;
    hipLaunchKernelGGL((k_gemm32e_tn_f32), dim3(ntiles), dim3(64), 0, on, static_cast<const float*>(ctx->At),
                       rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, (const DevState*)nullptr);
    return hipGetLastError();
}

// the 64-column pass (first lookahead sweep of a solve): rcols / drows hold 64 entries each
hipError_t launch_gemm64_tn_f32(const ss_hip_ctx* ctx, const uint32_t* rcols, const uint32_t* drows,
                                float* D, uint32_t ldd, const DevState* st)
{
    if (ctx->n_pad % 256 != 0 || ctx->ldm % GK != 0) return hipErrorInvalidValue;
    const uint32_t ntiles = ctx->n_pad / 256;
    const uint32_t grid = ntiles < (uint32_t)ctx->num_cus ? ntiles : (uint32_t)ctx->num_cus;
    hipLaunchKernelGGL((k_gemm32_tn_f32<256, 512, 1, GK, false, 64>), dim3(grid), dim3(512), 0, ctx->stream,
                       static_cast<const float*>(ctx->At), rcols, drows, D, ctx->ldm, ctx->ldm, ldd, ntiles, st, 0u, nullptr, 2u);
    return hipGetLastError();
}

// D[Mg][ldd] = R[Mg][ldr] * At^T ; Mg % 128 == 0, ctx->n_pad % 128 == 0, ldm % 32 == 0
// blocked: 32-row chains summed in a second accumulator (the correlations of a batch); false: one chain per output
// (the full product G = A^T A of option gram_symmetric = 0: the symmetric build's chains)
hipError_t launch_gemm_tn_f32(const ss_hip_ctx* ctx, const float* R, uint32_t Mg, uint32_t ldr,
                              float* D, uint32_t ldd, const uint32_t* row_tile_skip, bool blocked)
{
    if (Mg % GM != 0 || ctx->n_pad % GN != 0 || ctx->ldm % GK != 0) return hipErrorInvalidValue;
    const uint32_t mtiles = Mg / GM, ntiles = ctx->n_pad / GN;
    if (blocked)
        hipLaunchKernelGGL((k_gemm_tn_f32<false, true>), dim3(mtiles * ntiles), dim3(256), 0, ctx->stream, R,
                           static_cast<const float*>(ctx->At), D, mtiles, ctx->ldm, ldr, ctx->ldm, ldd, row_tile_skip);
    else
        hipLaunchKernelGGL((k_gemm_tn_f32<false, false>), dim3(mtiles * ntiles), dim3(256), 0, ctx->stream, R,
                           static_cast<const float*>(ctx->At), D, mtiles, ctx->ldm, ldr, ctx->ldm, ldd, row_tile_skip);
    return hipGetLastError();
}

// G[n_pad][ldd] = At · At^T from the tiles on and above the diagonal, each stored to both sides
hipError_t launch_gemm_sym_f32(const ss_hip_ctx* ctx, float* G, uint32_t ldd)
{
    if (ctx->n_pad % GN != 0 || ctx->ldm % GK != 0 || ldd % 4 != 0) return hipErrorInvalidValue;
    const uint64_t t = ctx->n_pad / GN;
    const uint64_t blocks = t * (t + 1) / 2;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const float* At = static_cast<const float*>(ctx->At);
    hipLaunchKernelGGL((k_gemm_tn_f32<true, false>), dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, At, At, G, (uint32_t)t,
                       ctx->ldm, ctx->ldm, ctx->ldm, ldd, (const uint32_t*)nullptr);
    return hipGetLastError();
}

}  // namespace sship
