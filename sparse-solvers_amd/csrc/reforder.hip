// reforder.hip — the "reference-order" engine (option "engine" = 3): the reference's iteration statement for
// statement (two transposed sweeps per iteration, the direction formed from the RE-COMPUTED correlations),
// with every floating-point reduction carried out in one documented, layout-independent order:
//
//   * a dot product / GEMV output keeps EIGHT partial sums; term number r (the row index of a transposed sweep,
//     the column index of  A x  and  A d, the position in the support of the K x K products) goes to partial
//     r & 7 in ascending r; the partials are combined ((0+1)+(2+3))+((4+5)+(6+7));
//   * products and sums are separately rounded (this file is compiled with -ffp-contract=off).
//
// That is the arithmetic a plain (non-FMA, 8-way unrolled) CPU GEMV performs, and it does not depend on grid
// size, tiling or thread count: the whole homotopy path — every pick, every step length, every coefficient —
// is reproducible bit for bit on any implementation that states the same order.  The fast engines (one fused
// sweep per iteration; Gram form) sum in other orders and agree with it to rounding; where rounding DECIDES
// (two columns reach the boundary within an ulp: homotopy-cpu.cpp:143-153 then skips the later one for good)
// they hand the signal to this engine (DevState::tie_stall, homotopy.hip).
//
// Reference (paths under /root/reference):
//   residual_vector            src/solvers/homotopy-cpu.cpp:87-98     -> k_ro_mv (A x), k_ro_sweep (A^T r)
//   find_max_gamma  p, q       src/solvers/homotopy-cpu.cpp:114-120   -> k_ro_mv (A d), k_ro_sweep (A^T p)
//   find_max_gamma  scan       src/solvers/homotopy-cpu.cpp:122-163   -> k_scansel (activeset.hip: element-wise)
//   online_column_inverse      src/linalg/online_inverse.h:183-293    -> k_ro_update
//   sign / direction / loop    src/solvers/homotopy-cpu.cpp:257-272   -> k_ro_dir
//   first pick                 src/solvers/homotopy-cpu.cpp:215-229   -> k_ro_init
//
// Schedule: ONE pass over A per iteration.  The reference forms the direction from sign(c_Gamma) of the correlations it
// has just re-computed, which would take a second pass (q = A^T A d needs the direction); here the signs are taken from
// c - gamma q (equal in exact arithmetic), the fused sweep [c, q] = A^T [r, p] runs, and k_ro_check compares the signs of
// the re-computed c with the ones the direction was built from — equal: every word is the reference order's; different
// (lambda within rounding of the tolerance): the direction is rebuilt from the true signs and q alone is swept again.
// Which sweeps run is a schedule; the values are the same either way.
//
// Roofline: HBM (m n s bytes per iteration).  The order pins one lane to a (column, 4 classes) pair, so the sweep stages
// the dictionary through LDS (k_ro_sweep_t: coalesced loads, every lane then reads its rows of its column from LDS):
// 0.84 of the HBM peak on configs[1]; the direct form (k_ro_sweep, 16 bytes of every 32 per lane) reaches 0.35.
#include "ss_hip_internal.h"
#include "ss_hip_device.h"

#include <algorithm>

namespace sship {

template <typename T> struct RoVec;
template <> struct RoVec<float>  { using V = v4f; static constexpr int VN = 4; static constexpr int LPC = 2; };
template <> struct RoVec<double> { using V = v2d; static constexpr int VN = 2; static constexpr int LPC = 4; };

constexpr int kRoThreads = 256;
constexpr int kRoUnroll = 8;            // 8-row groups a lane has in flight (8 x 16 B per lane)
constexpr uint32_t kRoChunk = 4096;     // rows of the two columns a chain dot product stages in LDS at a time

template <typename T>
__device__ __forceinline__ T combine8(const T (&s)[8])
{
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// the partial sums of one column are spread over LPC neighbouring lanes (VN partials each, classes h*VN ..):
// every lane of the group receives the combined sum
template <typename T>
__device__ __forceinline__ T combine_lanes(const T (&acc)[RoVec<T>::VN])
{
    if (RoVec<T>::VN == 4) {
        T s = (acc[0] + acc[1]) + (acc[2] + acc[RoVec<T>::VN - 1]);
        s = s + __shfl_xor(s, 1, 64);                 // (0..3) + (4..7)
        return s;
    } else {
        T s = acc[0] + acc[1];                        // 2h, 2h + 1
        s = s + __shfl_xor(s, 1, 64);                 // (0+1)+(2+3) | (4+5)+(6+7)
        s = s + __shfl_xor(s, 2, 64);
        return s;
    }
}

// ---- k_ro_sweep: [out0, out1] = A^T [v0, v1], 8 partial sums per column and right-hand side by row & 7 ----------------
// LPC lanes per column: lane h of a column reads rows 8 t + h*VN .. + VN - 1 (one 16-byte load) for ascending
// t and keeps VN partial sums per right-hand side; the right-hand sides sit in LDS.  Per workgroup: max |out0| and its
// first index (inf_norm).  gate != 0: the launch only runs when DevState::ro_redo is raised (the q-only sweep after a
// direction that had to be rebuilt).
template <typename T, int NRHS>
__global__ __launch_bounds__(kRoThreads)
void k_ro_sweep(const T* __restrict__ At, uint32_t ldm, uint32_t n, uint32_t ngroups, uint32_t mc,
                const T* __restrict__ v, size_t v_stride, T* __restrict__ out0, T* __restrict__ out1,
                T* __restrict__ pmax_val, uint32_t* __restrict__ pmax_idx, const DevState* st, int gate)
{
    using V = typename RoVec<T>::V;
    constexpr int VN = RoVec<T>::VN, LPC = RoVec<T>::LPC, CPB = kRoThreads / LPC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* lds = reinterpret_cast<T*>(smem);                              // [NRHS][mc]
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    if (st != nullptr && (st->done != 0 || (gate && st->ro_redo == 0u))) return;
    const uint32_t h = threadIdx.x % LPC, cl = threadIdx.x / LPC;
    const uint32_t nchunks = (ldm + mc - 1) / mc;
    T best = T(-1);
    uint32_t best_idx = 0xffffffffu;
    bool lds_valid = false;
    for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const uint32_t col = g * CPB + cl;
        T acc[NRHS][VN];
#pragma unroll
        for (int k = 0; k < NRHS; ++k)
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[k][e] = T(0);
        for (uint32_t ch = 0; ch < nchunks; ++ch) {
            const uint32_t r0 = ch * mc;
            const uint32_t rows = (ldm - r0 < mc) ? (ldm - r0) : mc;
            if (nchunks > 1 || !lds_valid) {
                if (lds_valid) __syncthreads();
#pragma unroll
                for (int k = 0; k < NRHS; ++k)
                    for (uint32_t i = threadIdx.x * VN; i < rows; i += kRoThreads * VN)
                        *reinterpret_cast<V*>(&lds[(size_t)k * mc + i]) = *reinterpret_cast<const V*>(&v[(size_t)k * v_stride + r0 + i]);
                __syncthreads();
                lds_valid = true;
            }
            const V* cp = reinterpret_cast<const V*>(At + (size_t)col * ldm + r0) + h;
            const V* vp = reinterpret_cast<const V*>(lds) + h;
            const uint32_t nsteps = rows / 8u;                        // rows is a multiple of 256
            for (uint32_t t = 0; t < nsteps; t += kRoUnroll) {
                V a[kRoUnroll];
#pragma unroll
                for (int u = 0; u < kRoUnroll; ++u) a[u] = __builtin_nontemporal_load(cp + (size_t)(t + u) * LPC);
#pragma unroll
                for (int u = 0; u < kRoUnroll; ++u) {
#pragma unroll
                    for (int k = 0; k < NRHS; ++k) {
                        const V b = vp[(size_t)k * (mc / VN) + (t + u) * LPC];
#pragma unroll
                        for (int e = 0; e < VN; ++e) acc[k][e] = acc[k][e] + a[u][e] * b[e];
                    }
                }
            }
        }
        const T s0 = combine_lanes<T>(acc[0]);
        T s1 = T(0);
        if (NRHS > 1) s1 = combine_lanes<T>(acc[NRHS - 1]);
        if (h == 0 && col < n) {
            out0[col] = s0;
            if (NRHS > 1) out1[col] = s1;
            const T a = s0 < T(0) ? -s0 : s0;
            if (a > best) { best = a; best_idx = col; }               // ascending columns: first maximum kept
        }
    }
    if (pmax_val == nullptr) return;
    block_reduce_pair<T, true>(best, best_idx, sv, si);
    if (threadIdx.x == 0) { pmax_val[blockIdx.x] = best; pmax_idx[blockIdx.x] = best_idx; }
}

// ---- k_ro_sweep_t: the same sums, the dictionary staged through LDS ------------------------------------------------
// The order above pins ONE lane to all rows of a (column, 4 or 2 classes) pair, so a direct sweep reads 16 bytes out of
// every 32 per lane and a wave-instruction touches 32 columns: a quarter of every cache line per request (measured:
// 0.35-0.45 of the HBM peak).  Here the workgroup loads a tile of CPB columns x 64 rows the coalesced way (16 lanes per
// 256 B of a column; f64: 32 lanes per 512 B), parks it in LDS (column pitch 64 rows + LPC vectors: conflict-free for
// the loaders and for the readers) and every lane then reads ITS rows of ITS column from there, in the same ascending
// order: bit for bit the sums of k_ro_sweep.  Two register sets of loads in flight (2 x 32 KiB per workgroup), two LDS
// buffers, one barrier per stage; the 64 rows of the right-hand sides travel with the tile.
constexpr uint32_t kRoStageRows = 64;

template <typename T>
constexpr uint32_t ro_stage_lds_vecs(int nrhs)      // 16-byte vectors of ONE buffer: the tile + the right-hand sides' rows
{
    return (uint32_t)(kRoThreads / RoVec<T>::LPC) * (kRoStageRows * sizeof(T) / 16 + RoVec<T>::LPC) + (uint32_t)nrhs * (kRoStageRows * sizeof(T) / 16);
}

// Right-hand side k = b * NS + s is block b (0: r, 1: p) of slot s: v + b * blk_stride + s * ldm; its output goes to
// out + b * out_blk + s * n_pad.  NS > 1: several signals share ONE pass over the dictionary (a batch's tie re-runs,
// batches in engine 3); a slot that is done (or, gate != 0, whose direction stood the check) keeps its outputs.
template <typename T, int NB, int NS>
__global__ __launch_bounds__(kRoThreads, 2)
void k_ro_sweep_t(const T* __restrict__ At, uint32_t ldm, uint32_t n, uint32_t ngroups,
                  const T* __restrict__ v, size_t blk_stride, T* __restrict__ out, size_t out_blk, uint32_t n_pad,
                  T* __restrict__ pmax_val, uint32_t* __restrict__ pmax_idx, uint32_t pmax_stride, const DevState* st, int gate)
{
    using V = typename RoVec<T>::V;
    constexpr int VN = RoVec<T>::VN, LPC = RoVec<T>::LPC, CPB = kRoThreads / LPC, NRHS = NB * NS;
    constexpr uint32_t RS = kRoStageRows;
    constexpr uint32_t VPC = RS * sizeof(T) / 16;                     // vectors of one column in a stage (16 / 32)
    constexpr uint32_t PV = VPC + LPC;                                // column pitch in LDS, in vectors
    constexpr uint32_t NLD = CPB * VPC / kRoThreads;                  // tile loads per lane and stage (8)
    constexpr uint32_t RV = RS / VN;                                  // vectors of one right-hand side's rows in a stage
    constexpr uint32_t BUFV = CPB * PV + NRHS * RV;
    static_assert(CPB * VPC % kRoThreads == 0 && NRHS * RV <= kRoThreads, "stage shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    V* lds = reinterpret_cast<V*>(smem);                              // [2][BUFV]
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    uint32_t live = 0;                                                // slots this launch works for
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
        if (st == nullptr || (st[sl].done == 0 && (!gate || st[sl].ro_redo != 0u))) live |= 1u << sl;
    if (live == 0u) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t h = tid % LPC, cl = tid / LPC;
    const uint32_t nstages = ldm / RS;                                // ldm is a multiple of 256: a multiple of 4 stages
    const bool has_rhs = tid < NRHS * RV;
    const uint32_t rk = tid / RV;                                     // (only meaningful when has_rhs)
    const T* vsrc = v + (size_t)(rk / NS) * blk_stride + (size_t)(rk % NS) * ldm + (size_t)(tid % RV) * VN;
    T best[NS];
    uint32_t best_idx[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) { best[sl] = T(-1); best_idx[sl] = 0xffffffffu; }
    for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        // loader: vector j of this lane is part (j*256 + tid) % VPC of column (j*256 + tid) / VPC
        const T* src[NLD];
        uint32_t dst[NLD];
#pragma unroll
        for (uint32_t j = 0; j < NLD; ++j) {
            const uint32_t vi = j * kRoThreads + tid;
            const uint32_t c = vi / VPC, part = vi % VPC;
            src[j] = At + (size_t)(g * CPB + c) * ldm + (size_t)part * VN;
            dst[j] = c * PV + part;
        }
        V ra[2][NLD], rb[2];
        rb[0] = rb[1] = V{};
        auto issue = [&](int set, uint32_t sg) {
            const size_t r0 = (size_t)sg * RS;
#pragma unroll
            for (uint32_t j = 0; j < NLD; ++j) ra[set][j] = __builtin_nontemporal_load(reinterpret_cast<const V*>(src[j] + r0));
            if (has_rhs) rb[set] = *reinterpret_cast<const V*>(vsrc + r0);
        };
        T acc[NRHS][VN];
#pragma unroll
        for (int k = 0; k < NRHS; ++k)
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[k][e] = T(0);
        auto stage = [&](int set, uint32_t sg) {
            V* buf = lds + (size_t)(sg & 1u) * BUFV;
#pragma unroll
            for (uint32_t j = 0; j < NLD; ++j) buf[dst[j]] = ra[set][j];
            if (has_rhs) buf[CPB * PV + tid] = rb[set];
            if (sg + 2 < nstages) issue(set, sg + 2);
            __syncthreads();
            const V* ap = buf + cl * PV + h;
            const V* bp = buf + CPB * PV + h;
            constexpr int TU = NRHS >= 8 ? 1 : (NRHS >= 4 ? 2 : (int)(RS / 8));   // (many right-hand sides: registers for the sums, not for look-ahead)
#pragma unroll TU
            for (uint32_t t = 0; t < RS / 8; ++t) {
                const V a = ap[t * LPC];
#pragma unroll
                for (int k = 0; k < NRHS; ++k) {
                    const V b = bp[(uint32_t)k * RV + t * LPC];
#pragma unroll
                    for (int e = 0; e < VN; ++e) acc[k][e] = acc[k][e] + a[e] * b[e];
                }
            }
        };
        __syncthreads();                                              // (the previous group's last stage is consumed)
        issue(0, 0);
        issue(1, 1);
        for (uint32_t sg = 0; sg < nstages; sg += 2) {
            stage(0, sg);
            stage(1, sg + 1);
        }
        const uint32_t col = g * CPB + cl;
#pragma unroll
        for (int k = 0; k < NRHS; ++k) {
            const T sum = combine_lanes<T>(acc[k]);
            const int bl = k / NS, sl = k % NS;
            if (h == 0 && col < n && ((live >> sl) & 1u)) {
                out[(size_t)bl * out_blk + (size_t)sl * n_pad + col] = sum;
                if (bl == 0) {
                    const T a = sum < T(0) ? -sum : sum;
                    if (a > best[sl]) { best[sl] = a; best_idx[sl] = col; }   // ascending columns: first maximum kept
                }
            }
        }
    }
    if (pmax_val == nullptr) return;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        T bv = best[sl];
        uint32_t bi = best_idx[sl];
        block_reduce_pair<T, true>(bv, bi, sv, si);
        if (threadIdx.x == 0 && ((live >> sl) & 1u)) {
            pmax_val[(size_t)sl * pmax_stride + blockIdx.x] = bv;
            pmax_idx[(size_t)sl * pmax_stride + blockIdx.x] = bi;
        }
        __syncthreads();
    }
}

// ---- k_ro_mv: r = y - A x over the touched columns (mode 0), p = A d over the support (mode 1) ----------
// One thread per row; term `col` goes to partial col & 7, columns ascending (the lists are sorted).
template <typename T>
__global__ __launch_bounds__(kRoThreads)
void k_ro_mv(const T* __restrict__ At, SlotDims L, const T* __restrict__ y, const T* __restrict__ coef,
             const uint32_t* __restrict__ list2, int mode, T* __restrict__ out, const DevState* __restrict__ st, int gate)
{
    {   // slot = blockIdx.y
        const size_t sl = blockIdx.y;
        y += sl * L.ldm; coef += sl * L.n_pad; list2 += sl * 2 * L.kcap; out += sl * L.ldm; st += sl;
    }
    if (st->done || (gate && st->ro_redo == 0u)) return;
    const uint32_t i = blockIdx.x * kRoThreads + threadIdx.x;
    if (i >= L.ldm) return;
    const uint32_t cnt = mode == 0 ? st->ntouched : st->K;
    const uint32_t* list = list2 + (size_t)st->cur * L.kcap;
    T s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = T(0);
    for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t col = list[j];
        const T prod = At[(size_t)col * L.ldm + i] * coef[col];
        switch (col & 7u) {                                           // uniform over the launch
            case 0: s[0] = s[0] + prod; break;
            case 1: s[1] = s[1] + prod; break;
            case 2: s[2] = s[2] + prod; break;
            case 3: s[3] = s[3] + prod; break;
            case 4: s[4] = s[4] + prod; break;
            case 5: s[5] = s[5] + prod; break;
            case 6: s[6] = s[6] + prod; break;
            default: s[7] = s[7] + prod; break;
        }
    }
    const T sum = combine8<T>(s);
    // (padding rows stay exactly zero even if x or d went non-finite)
    out[i] = (i < L.m) ? (mode == 0 ? (y[i] - sum) : sum) : T(0);
}

// ---- chain dot product of two device columns by one workgroup: both are staged through LDS in chunks, the
// ---- first LPC lanes walk the 8 partial sums in ascending row order.  The result is valid in thread 0.
template <typename T>
__device__ __forceinline__ T ro_chain_dot(const T* __restrict__ a, const T* __restrict__ b, uint32_t ldm, T* lds)
{
    using V = typename RoVec<T>::V;
    constexpr int VN = RoVec<T>::VN, LPC = RoVec<T>::LPC;
    T acc[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[e] = T(0);
    for (uint32_t r0 = 0; r0 < ldm; r0 += kRoChunk) {
        const uint32_t rows = (ldm - r0 < kRoChunk) ? (ldm - r0) : kRoChunk;
        __syncthreads();                                              // the previous chunk is consumed
        for (uint32_t i = threadIdx.x * VN; i < rows; i += blockDim.x * VN) {
            *reinterpret_cast<V*>(&lds[i]) = *reinterpret_cast<const V*>(&a[r0 + i]);
            *reinterpret_cast<V*>(&lds[kRoChunk + i]) = *reinterpret_cast<const V*>(&b[r0 + i]);
        }
        __syncthreads();
        if (threadIdx.x < (uint32_t)LPC) {
            const V* ap = reinterpret_cast<const V*>(lds) + threadIdx.x;
            const V* bp = reinterpret_cast<const V*>(lds + kRoChunk) + threadIdx.x;
            const uint32_t nsteps = rows / 8u;
#pragma unroll 8
            for (uint32_t t = 0; t < nsteps; ++t) {
                const V x = ap[t * LPC], w = bp[t * LPC];
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[e] = acc[e] + x[e] * w[e];
            }
        }
    }
    return combine_lanes<T>(acc);                                     // (lanes >= LPC carry zeros; thread 0 reads lanes 0..LPC-1)
}

// ---- k_ro_init: first pick (homotopy-cpu.cpp:217-229), inv = [1 / ||col||^2] through the norm --------------
template <typename T>
__global__ __launch_bounds__(kRoThreads)
void k_ro_init(const T* __restrict__ At, SlotDims L, const T* __restrict__ c,
               const T* __restrict__ pmax_val, const uint32_t* __restrict__ pmax_idx, uint32_t nb,
               T* __restrict__ d, uint8_t* __restrict__ insup, uint32_t* __restrict__ gam,
               uint32_t* __restrict__ touched, T* __restrict__ inv0, T tol, int strict_sign,
               DevState* st, TraceEntry* trace)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* lds = reinterpret_cast<T*>(smem);
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    {   // slot = blockIdx.y
        const size_t sl = blockIdx.y;
        c += sl * L.n_pad; d += sl * L.n_pad; insup += sl * L.n_pad;
        pmax_val += sl * L.pmax_stride; pmax_idx += sl * L.pmax_stride;
        gam += sl * 2 * L.kcap; touched += sl * 2 * L.kcap; inv0 += sl * 2 * (size_t)L.kcap * L.kcap;
        st += sl;
        if (sl != 0) trace = nullptr;
    }
    T c_inf;
    uint32_t idx;
    reduce_sweep_partials(pmax_val, pmax_idx, nb, c_inf, idx, sv, si);
    const T* col = At + (size_t)idx * L.ldm;
    const T dot = ro_chain_dot<T>(col, col, L.ldm, lds);
    if (threadIdx.x == 0) {
        const T nrm = sqrt(dot);                                      // online_inverse.h:193-201 (xnrm2)
        const T inv00 = T(1) / (nrm * nrm);
        const T seed = strict_sign ? c[idx] : c_inf;                  // first-step quirk (homotopy-cpu.cpp:223-227)
        d[idx] = sign_tol(seed, tol) * inv00;
        insup[idx] = 1;
        gam[0] = idx;
        touched[0] = idx;
        inv0[0] = inv00;
        st->done = 0; st->status = 0; st->iter = 0;
        st->K = 1; st->ntouched = 1; st->idx = idx; st->rank = 0; st->added = 1; st->cur = 0;
        st->done_round = 0;
        st->nsweeps = 0; st->ro_redo = 0;
        st->c_inf = (double)c_inf;
        st->gamma = 0.0;
        st->lambda0 = (float)c_inf;
        st->dot = (double)dot;
        if (trace != nullptr) { trace[0].idx = idx; trace[0].added = 1; trace[0].gamma = 0.0; trace[0].c_inf = (double)c_inf; }
    }
}

// ---- k_ro_update: online_column_inverse::insert / remove (online_inverse.h:183-293) ------------------------
// insert: workgroup b < K_new forms u1[.] = a_{Gamma_b} . a_idx (b == rank: a_idx . a_idx) as a chain dot
// product; the last to arrive forms u2 = inv u1 (8 partials by position & 7), d = 1 / (dot - u1 . u2) and the
// bordered inverse, element by element, directly in sorted-support order.  remove: the deflated inverse.
template <typename T>
__global__ __launch_bounds__(kRoThreads)
void k_ro_update(const T* __restrict__ At, SlotDims L, const uint32_t* __restrict__ gam2,
                 T* inv0, T* inv1, T* u1, T* u2, T* sgn, const T* c, const T* q, T* d, T tol,
                 DevState* st)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* lds = reinterpret_cast<T*>(smem);
    __shared__ T s_d;
    __shared__ uint32_t s_flag;
    {   // slot = blockIdx.y
        const size_t sl = blockIdx.y;
        gam2 += sl * 2 * L.kcap;
        inv0 += sl * 2 * (size_t)L.kcap * L.kcap; inv1 += sl * 2 * (size_t)L.kcap * L.kcap;
        u1 += sl * L.kcap; u2 += sl * L.kcap; sgn += sl * L.kcap;
        c += sl * L.n_pad; q += sl * L.n_pad; d += sl * L.n_pad;
        st += sl;
    }
    if (st->done) return;
    const uint32_t ldm = L.ldm, kcap = L.kcap;
    const uint32_t cur = st->cur;
    const uint32_t K_new = st->K;
    const uint32_t rank = st->rank;
    const bool added = st->added != 0;
    const uint32_t* gam_new = gam2 + (size_t)(cur ^ 1u) * kcap;
    if (added && blockIdx.x < K_new) {
        const uint32_t b = blockIdx.x;
        const T v = ro_chain_dot<T>(At + (size_t)gam_new[b] * ldm, At + (size_t)st->idx * ldm, ldm, lds);
        if (threadIdx.x == 0) {
            if (b == rank) st->dot = (double)v;
            else u1[b - (b > rank ? 1u : 0u)] = v;
        }
    }
    if (!arrive_last(&st->ticket_gram, gridDim.x, &s_flag)) return;

    const T* Iold = cur ? inv1 : inv0;
    T* Inew = cur ? inv0 : inv1;
    const size_t P = kcap;
    const uint32_t K_old = added ? K_new - 1 : K_new + 1;
    if (added) {
        const uint32_t nn = K_old;
        for (uint32_t i = threadIdx.x; i < nn; i += blockDim.x) {     // u2 = inv * u1 (online_inverse.h:224-225)
            T s[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] = T(0);
            for (uint32_t j0 = 0; j0 < nn; j0 += 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (j0 + (uint32_t)k < nn) s[k] = s[k] + Iold[i * P + j0 + k] * u1[j0 + k];
            }
            u2[i] = combine8<T>(s);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            T s[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] = T(0);
            for (uint32_t j0 = 0; j0 < nn; j0 += 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (j0 + (uint32_t)k < nn) s[k] = s[k] + u1[j0 + k] * u2[j0 + k];
            }
            const T dotv = (T)load_handoff(&st->dot);
            s_d = T(1) / (dotv - combine8<T>(s));                     // online_inverse.h:228
        }
        __syncthreads();
        const T dv = s_d;
        const uint32_t tot = K_new * K_new;
        for (uint32_t e = threadIdx.x; e < tot; e += blockDim.x) {    // online_inverse.h:229-248
            const uint32_t a = e / K_new, b = e - a * K_new;
            T v;
            if (a == rank && b == rank) v = dv;
            else if (a == rank) v = -dv * u2[b - (b > rank ? 1u : 0u)];
            else if (b == rank) v = -dv * u2[a - (a > rank ? 1u : 0u)];
            else {
                const uint32_t oa = a - (a > rank ? 1u : 0u), ob = b - (b > rank ? 1u : 0u);
                v = Iold[oa * P + ob] + (dv * u2[oa]) * u2[ob];
            }
            Inew[a * P + b] = v;
        }
    } else {
        const uint32_t nn = K_old;                                    // online_inverse.h:275-290
        const T dd = Iold[rank * P + rank];
        const T sc = -(T(1) / dd);
        for (uint32_t i = threadIdx.x; i < nn; i += blockDim.x) u2[i] = Iold[i * P + rank] * sc;
        __syncthreads();
        const uint32_t tot = K_new * K_new;
        for (uint32_t e = threadIdx.x; e < tot; e += blockDim.x) {
            const uint32_t a = e / K_new, b = e - a * K_new;
            const uint32_t oa = a + (a >= rank ? 1u : 0u), ob = b + (b >= rank ? 1u : 0u);
            Inew[a * P + b] = Iold[oa * P + ob] + (-dd * u2[oa]) * u2[ob];
        }
    }
    __syncthreads();
    // the signs of the correlations after the step, taken from c - gamma q (homotopy-cpu.cpp:259-260 reads them off the
    // re-computed c: k_ro_check compares once that exists), and direction = inv * sign (:263), 8 partials by position
    {
        const T g = (T)st->gamma;
        const uint32_t* gam_old = gam2 + (size_t)cur * kcap;
        for (uint32_t a = threadIdx.x; a < K_new; a += blockDim.x) {
            const uint32_t col = gam_new[a];
            sgn[a] = sign_tol(c[col] - g * q[col], tol);
        }
        for (uint32_t j = threadIdx.x; j < K_old; j += blockDim.x) d[gam_old[j]] = T(0);
        __syncthreads();
        for (uint32_t a = threadIdx.x; a < K_new; a += blockDim.x) {
            T s8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) s8[k] = T(0);
            for (uint32_t b0 = 0; b0 < K_new; b0 += 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (b0 + (uint32_t)k < K_new) s8[k] = s8[k] + Inew[a * P + b0 + k] * sgn[b0 + k];
            }
            d[gam_new[a]] = combine8<T>(s8);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { st->cur = cur ^ 1u; st->ro_redo = 0u; }
}

// ---- k_ro_check: lambda = ||c||_inf of the correlations just re-computed and the loop's while-test
// ---- (homotopy-cpu.cpp:270-272); then the signs the direction was built from (k_ro_update: c - gamma q) against
// ---- sign(c_Gamma) of the re-computed c (:259-260).  Equal: nothing to do.  Different (lambda within rounding of the
// ---- dead zone): the direction is rebuilt from the true signs (:263, 8 partials by position) and ro_redo is raised —
// ---- p = A d and q = A^T p are then formed again before the scan.
template <typename T>
__global__ __launch_bounds__(kUpdThreads)
void k_ro_check(const T* c, const T* pmax_val, const uint32_t* pmax_idx, uint32_t nb,
                T* d, const uint32_t* gam2, const T* inv0, const T* inv1, T* sgn, SlotDims L,
                T tol, uint32_t max_iter, DevState* st, uint32_t* hflags, uint32_t* ndone, uint32_t nslots, int force)
{
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    {   // slot = blockIdx.y
        const size_t sl = blockIdx.y;
        c += sl * L.n_pad; d += sl * L.n_pad;
        pmax_val += sl * L.pmax_stride; pmax_idx += sl * L.pmax_stride;
        gam2 += sl * 2 * L.kcap; sgn += sl * L.kcap;
        inv0 += sl * 2 * (size_t)L.kcap * L.kcap; inv1 += sl * 2 * (size_t)L.kcap * L.kcap;
        st += sl;
    }
    if (st->done) return;
    T c_inf;
    uint32_t imax;
    reduce_sweep_partials(pmax_val, pmax_idx, nb, c_inf, imax, sv, si);
    const uint32_t iter = st->iter;
    if (iter == 0u) return;                                           // (the first direction carries the seed's sign: :223-227)
    if (!(iter < max_iter && c_inf > tol)) {
        if (threadIdx.x == 0) {
            st->c_inf = (double)c_inf;
            st->done_round = iter + 1u;
            st->done = 1;
            signal_done(hflags, ndone, nslots, iter + 1u);
        }
        return;
    }
    const uint32_t kcap = L.kcap;
    const uint32_t cur = st->cur;
    const uint32_t K = st->K;
    const uint32_t* gam = gam2 + (size_t)cur * kcap;
    const T* I = cur ? inv1 : inv0;
    const size_t P = kcap;
    int differs = force;
    for (uint32_t a = threadIdx.x; a < K; a += blockDim.x) {
        const T s_true = sign_tol(c[gam[a]], tol);
        if (s_true != sgn[a]) { differs = 1; }
    }
    if (!__syncthreads_or(differs)) return;
    for (uint32_t a = threadIdx.x; a < K; a += blockDim.x) sgn[a] = sign_tol(c[gam[a]], tol);
    __syncthreads();
    for (uint32_t a = threadIdx.x; a < K; a += blockDim.x) {
        T s8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s8[k] = T(0);
        for (uint32_t b0 = 0; b0 < K; b0 += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (b0 + (uint32_t)k < K) s8[k] = s8[k] + I[a * P + b0 + k] * sgn[b0 + k];
        }
        d[gam[a]] = combine8<T>(s8);
    }
    if (threadIdx.x == 0) { st->ro_redo = 1u; st->nsweeps = st->nsweeps + 1u; }
}

// ---- launchers -----------------------------------------------------------------------------------------------
// most slots one pass carries: 8 right-hand sides (with 16 the sweep's LDS reads and sums cost more than a second pass)
// (fp32: 8 slots = 16 right-hand sides per pass — the stage's right-hand-side loaders are the 256 threads —, fp64: 4)
template <typename T> constexpr uint32_t ro_max_slots() { return sizeof(T) == 4 ? 8u : 4u; }
uint32_t ro_slots_max(const ss_hip_ctx* ctx, bool f64) { return ctx->ro_staged ? (f64 ? ro_max_slots<double>() : ro_max_slots<float>()) : 1u; }

template <typename T, int NB, int NS>
static hipError_t ro_sweep_go(const ss_hip_ctx* ctx, uint32_t grid, uint32_t ngroups, const T* v, size_t blk_stride, T* out, size_t out_blk,
                              uint32_t n_pad, T* pmax_val, uint32_t* pmax_idx, uint32_t pmax_stride, const DevState* st, bool gate)
{
    static const bool attr_ok = [] {
        const int lim = (int)(2 * ro_stage_lds_vecs<T>(NB * NS) * 16);
        const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ro_sweep_t<T, NB, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        return ok;
    }();
    if (!attr_ok) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_ro_sweep_t<T, NB, NS>), dim3(grid), dim3(kRoThreads), (size_t)2 * ro_stage_lds_vecs<T>(NB * NS) * 16, ctx->stream,
                       static_cast<const T*>(ctx->At), ctx->ldm, (uint32_t)ctx->n, ngroups, v, blk_stride, out, out_blk, n_pad,
                       pmax_val, pmax_idx, pmax_stride, st, gate ? 1 : 0);
    return hipGetLastError();
}

// [c, q](slot s) = A^T [r, p](slot s) for nslots slots of the workspace layout (nblk = 1: the first block only, from v to out)
template <typename T>
hipError_t launch_ro_sweep(const ss_hip_ctx* ctx, const T* v, size_t blk_stride, T* out, size_t out_blk, uint32_t n_pad, int nblk,
                           uint32_t nslots, T* pmax_val, uint32_t* pmax_idx, uint32_t pmax_stride, uint32_t* nblocks_out,
                           const DevState* st, bool gate)
{
    constexpr uint32_t CPB = kRoThreads / RoVec<T>::LPC;
    const uint32_t ngroups = ctx->n_pad / CPB;                        // n_pad is a multiple of 256
    uint32_t grid = std::min<uint32_t>(ngroups, kMaxSweepBlocks);
    if (nblocks_out) *nblocks_out = grid;
    if (ctx->ro_staged || nslots > 1) {
        hipError_t e = hipErrorInvalidValue;
#define SS_RO_CASE(NBV, NSV) e = ro_sweep_go<T, NBV, NSV>(ctx, grid, ngroups, v, blk_stride, out, out_blk, n_pad, pmax_val, pmax_idx, pmax_stride, st, gate)
        if (nblk == 1) {
            if (nslots == 1) SS_RO_CASE(1, 1); else if (nslots == 2) SS_RO_CASE(1, 2); else if (nslots <= 4) SS_RO_CASE(1, 4);
            else if constexpr (sizeof(T) == 4) { if (nslots <= 8) SS_RO_CASE(1, 8); }
        } else {
            if (nslots == 1) SS_RO_CASE(2, 1); else if (nslots == 2) SS_RO_CASE(2, 2); else if (nslots <= 4) SS_RO_CASE(2, 4);
            else if constexpr (sizeof(T) == 4) { if (nslots <= 8) SS_RO_CASE(2, 8); }
        }
#undef SS_RO_CASE
        if (e == hipSuccess || nslots > 1) return e;
        (void)hipGetLastError();                                      // (the LDS attribute was refused: the direct form)
    }
    // direct form (option ro_staged = 0, one slot): the right-hand sides whole in LDS, up to 32 KiB
    uint32_t mc = (uint32_t)(32768u / (sizeof(T) * nblk));
    mc -= mc % kRowPad;
    if (mc > ctx->ldm) mc = ctx->ldm;
    if (nblk == 2)
        hipLaunchKernelGGL((k_ro_sweep<T, 2>), dim3(grid), dim3(kRoThreads), (size_t)2 * mc * sizeof(T), ctx->stream,
                           static_cast<const T*>(ctx->At), ctx->ldm, (uint32_t)ctx->n, ngroups, mc, v, blk_stride, out, out + out_blk,
                           pmax_val, pmax_idx, st, gate ? 1 : 0);
    else
        hipLaunchKernelGGL((k_ro_sweep<T, 1>), dim3(grid), dim3(kRoThreads), (size_t)mc * sizeof(T), ctx->stream,
                           static_cast<const T*>(ctx->At), ctx->ldm, (uint32_t)ctx->n, ngroups, mc, v, blk_stride, out, (T*)nullptr,
                           pmax_val, pmax_idx, st, gate ? 1 : 0);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_ro_mv(const ss_hip_ctx* ctx, Workspace<T>& ws, int mode, uint32_t nslots, bool gate)
{
    const uint32_t blocks = (ctx->ldm + kRoThreads - 1) / kRoThreads;
    T* out = mode == 0 ? ws.rhs : ws.rhs + (size_t)ws.dims.b_pad * ctx->ldm;
    hipLaunchKernelGGL((k_ro_mv<T>), dim3(blocks, nslots), dim3(kRoThreads), 0, ctx->stream, static_cast<const T*>(ctx->At), ws.dims,
                       (const T*)ws.y, mode == 0 ? (const T*)ws.x : (const T*)ws.d,
                       mode == 0 ? (const uint32_t*)ws.touched : (const uint32_t*)ws.gam, mode, out, (const DevState*)ws.st, gate ? 1 : 0);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_ro_init(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t nparts, T tol)
{
    hipLaunchKernelGGL((k_ro_init<T>), dim3(1, nslots), dim3(kRoThreads), 2 * (size_t)kRoChunk * sizeof(T), ctx->stream,
                       static_cast<const T*>(ctx->At), ws.dims, (const T*)ws.c, (const T*)ws.pmax_val,
                       (const uint32_t*)ws.pmax_idx, nparts, ws.d, ws.insup, ws.gam, ws.touched, ws.inv[0], tol,
                       ctx->strict_sign, ws.st, ws.trace);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_ro_update(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round, T tol)
{
    uint32_t gb = round + 1;                                          // support size after this round is <= round + 1
    if (gb > ws.kcap) gb = ws.kcap;
    hipLaunchKernelGGL((k_ro_update<T>), dim3(gb, nslots), dim3(kRoThreads), 2 * (size_t)kRoChunk * sizeof(T), ctx->stream,
                       static_cast<const T*>(ctx->At), ws.dims, (const uint32_t*)ws.gam, ws.inv[0], ws.inv[1], ws.u1, ws.u2,
                       ws.sgn, (const T*)ws.c, (const T*)ws.q, ws.d, tol, ws.st);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_ro_check(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t nparts, T tol, uint32_t max_iter)
{
    hipLaunchKernelGGL((k_ro_check<T>), dim3(1, nslots), dim3(kUpdThreads), 0, ctx->stream, (const T*)ws.c, (const T*)ws.pmax_val,
                       (const uint32_t*)ws.pmax_idx, nparts, ws.d, (const uint32_t*)ws.gam, (const T*)ws.inv[0], (const T*)ws.inv[1],
                       ws.sgn, ws.dims, tol, max_iter, ws.st, ctx->dev_flags, ws.ndone, nslots, ctx->ro_force_resweep);
    return hipGetLastError();
}

// one round of the engine for the first nslots slots of the workspace (homotopy-cpu.cpp:236-272, one pass over A)
template <typename T>
hipError_t launch_ro_round(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round, uint32_t nparts, T tol, uint32_t max_iter)
{
    const size_t rblk = (size_t)ws.dims.b_pad * ctx->ldm, cblk = (size_t)ws.dims.b_pad * ws.dims.n_pad;
    hipError_t e;
    if ((e = launch_ro_mv<T>(ctx, ws, 0, nslots, false)) != hipSuccess) return e;
    if ((e = launch_ro_mv<T>(ctx, ws, 1, nslots, false)) != hipSuccess) return e;
    if ((e = launch_ro_sweep<T>(ctx, ws.rhs, rblk, ws.c, cblk, ws.dims.n_pad, 2, nslots, ws.pmax_val, ws.pmax_idx, ws.dims.pmax_stride,
                                nullptr, ws.st, false)) != hipSuccess) return e;
    if ((e = launch_ro_check<T>(ctx, ws, nslots, nparts, tol, max_iter)) != hipSuccess) return e;
    if ((e = launch_ro_mv<T>(ctx, ws, 1, nslots, true)) != hipSuccess) return e;
    if ((e = launch_ro_sweep<T>(ctx, ws.rhs + rblk, rblk, ws.q, cblk, ws.dims.n_pad, 1, nslots, (T*)nullptr, (uint32_t*)nullptr, 0u,
                                nullptr, ws.st, true)) != hipSuccess) return e;
    if ((e = launch_scansel_plain<T>(ctx, ws, nslots, round, nparts, tol, max_iter)) != hipSuccess) return e;
    return launch_ro_update<T>(ctx, ws, nslots, round, tol);
}

#define SS_RO_INST(T)                                                                                                          \
    template hipError_t launch_ro_sweep<T>(const ss_hip_ctx*, const T*, size_t, T*, size_t, uint32_t, int, uint32_t, T*, uint32_t*, uint32_t, \
                                           uint32_t*, const DevState*, bool);                                                   \
    template hipError_t launch_ro_mv<T>(const ss_hip_ctx*, Workspace<T>&, int, uint32_t, bool);                               \
    template hipError_t launch_ro_init<T>(const ss_hip_ctx*, Workspace<T>&, uint32_t, uint32_t, T);                           \
    template hipError_t launch_ro_update<T>(const ss_hip_ctx*, Workspace<T>&, uint32_t, uint32_t, T);                         \
    template hipError_t launch_ro_check<T>(const ss_hip_ctx*, Workspace<T>&, uint32_t, uint32_t, T, uint32_t);                \
    template hipError_t launch_ro_round<T>(const ss_hip_ctx*, Workspace<T>&, uint32_t, uint32_t, uint32_t, T, uint32_t);
SS_RO_INST(float)
SS_RO_INST(double)
#undef SS_RO_INST

}  // namespace sship
