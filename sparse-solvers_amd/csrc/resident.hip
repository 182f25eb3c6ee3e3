// resident.hip — the RESIDENT subset solve: the whole homotopy path of one signal in ONE workgroup, with the subset's Gram values
// in REGISTERS and the active-set inverse as a packed triangle in LDS.  fp64 and fp32.
//
// Why.  The screened forms (screen.hip) solve the path on a column subset S and certify every state of it against all columns
// with one pass over the fp16 copy of A.  In fp32 the path ran in k_sub_solve (subbatch.hip: Gram rows of the subset in 129 KB of
// LDS, nine barriers per iteration, 5.9 us each); in fp64 nothing fit one workgroup's LDS (128 positions x 448 columns x 8 bytes)
// and the path ran in the launch-per-iteration engine on a 2048-column sub-dictionary: 138 launches of 37 us — 5.2 of the 8.0 ms
// of a configs[4] solve (A 16384 x 131072).  What an iteration needs is small:
//
//   q_j = sum_p d_p G[p][j]  over the subset's columns j and the support's positions p   (homotopy-cpu.cpp:114-120, Gram form)
//   c_j <- c_j - gamma q_j                                                                (:97, Gram form; see "Arithmetic")
//   the scan for the smallest positive step (:122-163), the toggle, x += gamma d          (:236-252)
//   the bordered inverse (online_inverse.h:183-248) and the direction inv . sign(c_S)     (homotopy-cpu.cpp:257-267)
//
// — P x |S| Gram values, P^2 entries of the inverse, and all of it fits ONE compute unit if the Gram values live in its 512 KB of
// vector registers instead of its 160 KB of LDS:
//
//   * a thread holds the Gram values of CT columns x PT positions (fp64: 4 x 18 = 72 doubles = 144 VGPRs of the 256 a thread has
//     at two waves per SIMD; fp32: 8 x 9 floats); the eight lanes of a column group own the positions p = 8 e + g, g = 0..7, so
//     the group is busy from the 8th column on; one 8-byte LDS read of d_p feeds CT fmas; three DPP steps add the eight partial
//     sums — every lane of the group then holds q of its CT columns;
//   * the inverse is kept as the packed upper triangle I[b (b + 1) / 2 + a], a <= b (fp64, 144 positions: 83.5 KB; the square
//     would take 167 KB).  It is symmetric up to the rounding of (dv u2_a) u2_b against (dv u2_b) u2_a: one ulp, where the fp64
//     tolerance is 1e-10 and the fp32 one 1e-5;
//   * ONE pass over the inverse per iteration forms both u2 = I u1 (the bordering vector) and w = I s (the old block's share of
//     the new direction): with I' = [[I + dv u2 u2^T, -dv u2], [-dv u2^T, dv]] and s' = [s; sigma],
//         d' = I' s' = [w - beta u2; beta],   beta = dv (sigma - u2 . s)
//     — exact algebra, fresh signs and a fresh product every iteration (nothing drifts); the in-place update of the triangle then
//     runs off the critical path (nobody reads it before the next iteration's pass);
//   * four workgroup barriers per iteration (pick / signs + u1 / u2 + w / direction), no hand-shake: the arg-min carries the
//     subset index (ascending columns: the left-most rule is the smaller index) and the support's positions sit in an LDS table.
//
// Only REGULAR paths are certified by the forms that use this kernel (every step inserts a column: k_scr_residuals, k_s64_dense
// refuse the others), so a step that would REMOVE a column ends the kernel with kStatusSubsetDecline and the caller's next engine
// solves the signal; there is no removal code here.  Everything the reference's loop decides is decided with the reference's
// predicates (strict t > 0, left-most minimum, sign dead zone, first-step sign quirk: homotopy-cpu.cpp:122-163, :59-67, :223-227).
//
// Arithmetic.  c is carried as c <- fma(-gamma, q, c) from c0 (exact fp32 / fp64 dot products of the subset's columns with y):
// P rounding errors of size eps |gamma q| — the same count and size as the chain c0 - sum_p x_p g_p has.  Sums run in a fixed
// order (per lane ascending positions, then the DPP tree): deterministic.  Results agree with the oracle within the parity
// tolerances (tests/test_gpu_screen.py), not bit for bit; the bit-for-bit engine is reforder.hip.
#include "ss_hip_internal.h"
#include "ss_hip_device.h"
#include "resident.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace sship {

template <typename T> struct ResVec;
template <> struct ResVec<double> { typedef double v2 __attribute__((ext_vector_type(2))); };
template <> struct ResVec<float>  { typedef float v2 __attribute__((ext_vector_type(2))); };

template <typename T>
struct ResLds {
    T* I;              // [PCAP][PCAP + 1] the inverse over the positions (symmetric: every element is computed once and stored on both sides; zero beyond P)
    T* dvec;           // [PCAP] direction by position (zero beyond P)
    T* xvec;           // [PCAP] x by position
    T* u1sg;           // [PCAP][2] {u1, sign}
    T* u2w;            // [PCAP][2] {u2, w}
    int32_t* posof;    // [S] position of a subset column in the support, -1 = not in it
    uint32_t* subc;    // [S] the subset's columns, ascending (0xffffffff: none)
    uint32_t* pcol;    // [PCAP] column of a position
    uint32_t* psub;    // [PCAP] its subset index
};

template <typename T>
__host__ __device__ inline size_t res_lds_bytes()
{
    typedef ResCfg<T> C;
    return ((size_t)C::PCAP * (C::PCAP + 1) + 6 * (size_t)C::PCAP) * sizeof(T) + ((size_t)2 * C::S + 2 * (size_t)C::PCAP) * 4;
}

template <typename T> __device__ __forceinline__ T res_sqrt(T v);
template <> __device__ __forceinline__ float res_sqrt<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double res_sqrt<double>(double v) { return sqrt(v); }
template <typename T> __device__ __forceinline__ T res_fma(T a, T b, T c);
template <> __device__ __forceinline__ float res_fma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ double res_fma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> __device__ __forceinline__ T res_abs(T v) { return v < T(0) ? -v : v; }

// A value every lane holds alike, moved to scalar registers: branches on it are scalar branches.  (Left in vector registers, every
// `break` of the solve loop is a divergent branch to the compiler, and its structurizer carries all loop state — the CT x PT Gram
// registers included — through a copy per exit: twice the registers, spilled in fp64.)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __uint_as_float(uni(__float_as_uint(v))); }
__device__ __forceinline__ double uni(double v)
{
    const uint32_t lo = uni((uint32_t)__double2loint(v)), hi = uni((uint32_t)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}

// Workgroup barrier for data that crosses it in LDS only: waits for this wave's LDS operations, not for its global loads and stores
// (__syncthreads() carries a release fence — s_waitcnt vmcnt(0) — so every barrier of a round would wait for the round's log
// stores and for the entering column's Gram row, ~1000 cycles each, where nothing behind the barrier needs them yet)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// dst = (lane's bit of msk) ? src : dst, written over dst
__device__ __forceinline__ void cmov_inplace(float& dst, float src, unsigned long long msk)
{
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(dst) : "v"(src), "s"(msk));
}
__device__ __forceinline__ void cmov_inplace(double& dst, double src, unsigned long long msk)
{
    uint32_t lo = (uint32_t)__double2loint(dst), hi = (uint32_t)__double2hiint(dst);
    const uint32_t slo = (uint32_t)__double2loint(src), shi = (uint32_t)__double2hiint(src);
    asm volatile("v_cndmask_b32_e64 %0, %0, %2, %4\n\tv_cndmask_b32_e64 %1, %1, %3, %4" : "+v"(lo), "+v"(hi) : "v"(slo), "v"(shi), "s"(msk));
    dst = __hiloint2double((int)hi, (int)lo);
}

// sum over the eight lanes of a column group (lanes 8 k .. 8 k + 7 of a DPP row): every lane receives it
template <typename T>
__device__ __forceinline__ T group8_sum(T v)
{
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppHalfMirror>(v);
    return v;
}

// max over the wave, every lane receives it (values >= 0 or -1)
template <typename T>
__device__ __forceinline__ T wave_max_val(T v)
{
    { const T o = dpp_mov<kDppXor1>(v); v = o > v ? o : v; }
    { const T o = dpp_mov<kDppXor2>(v); v = o > v ? o : v; }
    { const T o = dpp_mov<kDppHalfMirror>(v); v = o > v ? o : v; }
    { const T o = dpp_mov<kDppMirror>(v); v = o > v ? o : v; }
    T b = lane_value(v, 0);
#pragma unroll
    for (int r = 1; r < 4; ++r) { const T o = lane_value(v, 16 * r); b = o > b ? o : b; }
    return b;
}

// ---- k_res_solve ------------------------------------------------------------------------------------------------------
// One workgroup per slot.  Gs: the subset's Gram matrix [S][gpitch] (rows and columns by subset index); c0: dense over the
// dictionary's columns, exact at the subset's (c0[sub[j]]); sub: the subset, ascending.  Logs every state (header, x by position),
// leaves dense x, the sorted support / touched lists and the slot's state like k_sub_solve (subbatch.hip).
// OMP: orthogonal matching pursuit on the same data (no reference implementation: DESIGN.md §3.7) — the pick is the left-most
// largest |c|, x_S = inv . c0_S by the same bordered update (x' = [x - beta u2; beta], beta = dv (c0_new - u2 . c0_S)), c by the chain.
//
// Lanes: a column group is eight consecutive lanes (g = 0..7) and CT columns; lane g holds the group's Gram values of the positions
// p = 8 e + g.  The partial sums of a chain leave by a reduce-scatter (recursive halving on the DPP path): lane g ends with the sum of
// column g & (CT - 1) of its group — the column it OWNS (lanes g < CT): its c, q, scan and bookkeeping are that lane's scalars.
template <typename T, bool OMP, bool STAMPS = false>
__global__ __launch_bounds__(ResCfg<T>::THREADS)
void k_res_solve(const T* __restrict__ Gs_all, uint32_t gpitch, size_t g_slot_stride, const T* __restrict__ c0_all, uint32_t c0_stride,
                 const uint32_t* __restrict__ sub_all, uint32_t n, T tol, uint32_t max_iter, int strict_sign, int tie_guard, int tie_exit,
                 uint32_t kcap, uint32_t* __restrict__ log_hdr, T* __restrict__ log_H, uint32_t* __restrict__ log_pcol, T* __restrict__ log_X, T* __restrict__ log_D,
                 T* __restrict__ x_all, uint32_t x_stride, uint32_t* __restrict__ gam2_all, uint32_t* __restrict__ touched2_all,
                 DevState* __restrict__ st_all, TraceEntry* trace, uint32_t trace_cap, unsigned long long* dbg = nullptr)
{
    // dbg (developer aid, SS_HIP_RES_STAMPS; Homotopy only): cycles of slot 0 by phase, summed over the rounds — [0] scan + arg-min,
    // [1] log, the entering row, x and c updates, [2] the pass over the inverse, [3] dots, direction, bordered inverse, [4] chain, [5] rounds
    unsigned long long tph[6] = { 0, 0, 0, 0, 0, 0 };
    unsigned long long tlast = 0;
#define RES_STAMP(PH) if constexpr (STAMPS) { const unsigned long long now_ = __builtin_readcyclecounter(); tph[PH] += now_ - tlast; tlast = now_; }
    typedef ResCfg<T> C;
    typedef typename ResVec<T>::v2 V2;
    constexpr uint32_t S = C::S, THREADS = C::THREADS, PCAP = C::PCAP, LOGCAP = C::LOGCAP;
    constexpr int CT = C::CT, PT = C::PCAP / 8;
    constexpr uint32_t NW = (THREADS + 63) / 64;
    constexpr uint32_t IP = PCAP + 1u;                  // row pitch of the inverse in LDS (odd: the rows of a wave start in different banks)
    static_assert(THREADS / 8 * CT == S, "eight lanes per column group");
    static_assert(PCAP % 8 == 0 && (CT == 4 || CT == 8), "positions dealt out over the eight lanes of a group");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ T s_max[16];
    __shared__ T s_minv[16];
    __shared__ uint32_t s_mini[16];
    __shared__ T s_f[4];            // [0] Gram value of the entering column with itself  [1] its sign (OMP: its c0)
    __shared__ uint32_t s_tie;
    ResLds<T> L;
    {
        T* p = reinterpret_cast<T*>(smem);
        L.I = p; p += (size_t)PCAP * IP;
        L.dvec = p; p += PCAP; L.xvec = p; p += PCAP; L.u1sg = p; p += 2 * PCAP; L.u2w = p; p += 2 * PCAP;
        L.posof = reinterpret_cast<int32_t*>(p);
        L.subc = reinterpret_cast<uint32_t*>(L.posof + S);
        L.pcol = L.subc + S;
        L.psub = L.pcol + PCAP;
    }
    const uint32_t slot = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t grp = t >> 3, g = t & 7u;
    const T* Gs = Gs_all + (size_t)slot * g_slot_stride;
    const T* c0 = c0_all + (size_t)slot * c0_stride;
    const uint32_t* sub = sub_all + (size_t)slot * S;
    T* x_out = x_all + (size_t)slot * x_stride;
    uint32_t* gam_out = gam2_all + (size_t)slot * 2 * kcap;
    uint32_t* tch_out = touched2_all + (size_t)slot * 2 * kcap;
    DevState* st = st_all + slot;
    uint32_t* hdr = log_hdr + (size_t)slot * LOGCAP * 8;
    T* LH = log_H != nullptr ? log_H + (size_t)slot * LOGCAP * 2 : nullptr;
    T* LX = log_X + (size_t)slot * LOGCAP * PCAP;
    T* LD = log_D != nullptr ? log_D + (size_t)slot * LOGCAP * PCAP : nullptr;
    if (slot != 0) trace = nullptr;

    // ---- the subset; my owned column ---------------------------------------------------------------------------------
    const uint32_t jc0 = grp * (uint32_t)CT;            // my group's first column (subset index)
    const bool owner = g < (uint32_t)CT;                // (CT = 4: lanes 4..7 end the reduce-scatter with copies of lanes 0..3's sums)
    const uint32_t myj = jc0 + (g & (uint32_t)(CT - 1));   // subset index of the column this lane's sums belong to
    const uint32_t mycol = sub[myj];
    const bool myvalid = owner && mycol < n;
    const T my_c0 = mycol < n ? c0[mycol] : T(0);
    T my_c = my_c0, my_q = T(0);
    T gr[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < PT; ++e) gr[c][e] = T(0);
    for (uint32_t j = t; j < S; j += THREADS) { L.subc[j] = sub[j]; L.posof[j] = -1; }
    for (uint32_t p = t; p < PCAP; p += THREADS) {
        L.dvec[p] = T(0); L.xvec[p] = T(0); L.pcol[p] = 0xffffffffu; L.psub[p] = 0u;
        L.u1sg[2 * p] = T(0); L.u1sg[2 * p + 1] = T(0); L.u2w[2 * p] = T(0); L.u2w[2 * p + 1] = T(0);
    }
    for (uint32_t e = t; e < PCAP * IP; e += THREADS) L.I[e] = T(0);     // (entries beyond the support are read as zeros by the pass)
    if (t == 0) s_tie = 0u;
    // The subset's Gram matrix was written by other compute units' launches: a first touch of every 128-byte line brings it into THIS
    // XCD's L2, so that the entering column's row — the round's only trip to memory — is an L2 hit.  (The loads complete under the
    // prologue; their sum is consumed at the very end.)
    T warm = T(0);
    {
        const size_t gbytes = (size_t)S * gpitch * sizeof(T);
        const unsigned char* gb = reinterpret_cast<const unsigned char*>(Gs);
#pragma unroll 4
        for (size_t off = (size_t)t * 128u; off < gbytes; off += (size_t)THREADS * 128u) warm += *reinterpret_cast<const T*>(gb + off);
    }

    // block-wide (value, index) reductions.  Within a wave the owning lanes hold ascending subset indices, and so do the waves: the
    // left-most minimum (maximum) is the first lane of the first wave that holds the extreme VALUE — a value-only reduction on the DPP
    // path, a ballot, one readlane; the waves' results meet in LDS.
    auto wave_first = [&](T v, T ext, uint32_t idx) __attribute__((always_inline)) -> uint32_t {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(v == ext);
        const int src = bal != 0ull ? (int)__builtin_ctzll(bal) : 0;
        return (uint32_t)__builtin_amdgcn_readlane((int)idx, src);
    };
    auto block_min = [&](T& v, uint32_t& i) __attribute__((always_inline)) {     // one barrier; s_minv / s_mini are rewritten a whole round later
        const T wv = -wave_max_val<T>(-v);
        const uint32_t wi = wave_first(v, wv, i);
        if (lane == 0) { s_minv[wave] = wv; s_mini[wave] = wi; }
        lds_barrier();
        T bv = s_minv[0];
        uint32_t bi = s_mini[0];
#pragma unroll
        for (uint32_t w = 1; w < NW; ++w) {
            const T ov = s_minv[w];
            const uint32_t oi = s_mini[w];
            if (ov < bv) { bv = ov; bi = oi; }
        }
        v = uni(bv); i = uni(bi);
    };
    auto block_argmax = [&](T& v, uint32_t& i) __attribute__((always_inline)) {  // (prologue and OMP picks: its own barriers)
        const T wv = wave_max_val<T>(v);
        const uint32_t wi = wave_first(v, wv, i);
        lds_barrier();
        if (lane == 0) { s_minv[wave] = wv; s_mini[wave] = wi; }
        lds_barrier();
        T bv = s_minv[0];
        uint32_t bi = s_mini[0];
#pragma unroll
        for (uint32_t w = 1; w < NW; ++w) {
            const T ov = s_minv[w];
            const uint32_t oi = s_mini[w];
            if (ov > bv) { bv = ov; bi = oi; }
        }
        v = uni(bv); i = uni(bi);
        lds_barrier();
    };
    // the bordered inverse in place (online_inverse.h:232-248): every element on or above the diagonal is computed once and stored on
    // both sides.  Row a is paired with row Pn - 1 - a (Pn = P + 1 rows): together Pn + 1 elements — a thread walks the elements
    // k = vg, vg + NG, ... of its pair; no division or square root per element, every step useful.  (P = 0xffffffff: a round abandoned.)
    auto border_inverse = [&](uint32_t P, T dvv) __attribute__((always_inline)) {
        constexpr uint32_t RW = PCAP / 2u + 1u;                     // row pairs a thread group spans
        constexpr uint32_t NG = THREADS / RW;                       // thread groups: the stride over a pair's elements
        const uint32_t Pn = P + 1u;
        const uint32_t pa = t % RW, vg = t / RW;
        const uint32_t a0 = pa, a1 = Pn - 1u - pa;                   // (a0 <= a1 while pa < (Pn + 1) / 2; a0 == a1: the middle row of an odd count)
        if (P != 0xffffffffu && vg < NG && 2u * pa < Pn + 1u - 0u && a0 <= a1) {
            const uint32_t n0 = Pn - a0;                            // elements (a0, a0 .. Pn - 1)
            const uint32_t ntot = a0 == a1 ? n0 : n0 + (Pn - a1);   // ... then (a1, a1 .. Pn - 1)
            const T u0 = a0 < P ? L.u2w[2 * a0] : T(0), u1v = a1 < P ? L.u2w[2 * a1] : T(0);
#pragma nounroll
            for (uint32_t k = vg; k < ntot; k += NG) {
                const bool first = k < n0;
                const uint32_t a = first ? a0 : a1;
                const uint32_t b = first ? a0 + k : a1 + (k - n0);
                const T ua = first ? u0 : u1v;
                T val;
                if (b == P) val = a == P ? dvv : -dvv * ua;
                else val = L.I[a * IP + b] + (dvv * ua) * L.u2w[2 * b];
                L.I[a * IP + b] = val;
                L.I[b * IP + a] = val;
            }
        }
    };

    // The entering column (subset index spi) at position Pn: its Gram values with my group's CT columns into my registers (the
    // lanes with g == Pn % 8 keep them); returns its Gram value with my OWNED column.
    // pendP / pend_dv: the bordered-inverse update the PREVIOUS round left undone runs here, between the row's loads and their first
    // use — under the round's only trip to memory (nobody reads the inverse before the next pass, two barriers on)
    auto enter_row = [&](uint32_t spi, uint32_t Pn, uint32_t pendP, T pend_dv) __attribute__((always_inline)) -> T {
        const T* row = Gs + (size_t)spi * gpitch;
        const T mine_g = row[myj];
        T gv[CT];
        if constexpr (sizeof(T) == 8) {
            const v2d a = *reinterpret_cast<const v2d*>(row + jc0), b = *reinterpret_cast<const v2d*>(row + jc0 + 2);
            gv[0] = a[0]; gv[1] = a[1]; gv[2] = b[0]; gv[3] = b[1];
        } else {
            const v4f a = *reinterpret_cast<const v4f*>(row + jc0), b = *reinterpret_cast<const v4f*>(row + jc0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gv[e] = a[e]; gv[4 + e] = b[e]; }
        }
        border_inverse(pendP, pend_dv);
        // (Pn is the same in every lane: the register block e = Pn / 8 is reached by scalar branches, and within it the lanes with
        // g == Pn % 8 take the values by a conditional move IN PLACE — as plain selects over all CT x PT registers this was 72 / 288
        // vector instructions per round, and the register allocator kept old and new values side by side)
        const uint32_t e_new = uni(Pn >> 3);
        const unsigned long long msk = __builtin_amdgcn_ballot_w64(g == (Pn & 7u));
#pragma unroll
        for (int e = 0; e < PT; ++e) {
            if (e_new == (uint32_t)e) {
#pragma unroll
                for (int c = 0; c < CT; ++c) cmov_inplace(gr[c][e], gv[c], msk);
            }
        }
        return mine_g;
    };
    // sum_p vec[p] G[p][my column] over the positions < P (vec is zero beyond P): CT partial sums per lane, then the reduce-scatter
    // over the group's eight lanes — mirror (g <-> 7 - g), xor 2, xor 1; a lane keeps the half its own index bit names
    auto chain = [&](const T* vec, uint32_t P) __attribute__((always_inline)) -> T {
        T acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = T(0);
#pragma unroll
        for (int e = 0; e < PT; ++e) {
            if ((uint32_t)(8 * e) < P) {                  // (uniform)
                const T dval = vec[8 * e + (int)g];
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = res_fma<T>(dval, gr[c][e], acc[c]);
            }
        }
        T h4[4];
        if constexpr (CT == 8) {
            const bool up = (g & 4u) != 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const T keep = up ? acc[4 + i] : acc[i], send = up ? acc[i] : acc[4 + i];
                h4[i] = keep + dpp_mov<kDppHalfMirror>(send);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) h4[i] = acc[i] + dpp_mov<kDppHalfMirror>(acc[i]);
        }
        const bool b1 = (g & 2u) != 0u, b0 = (g & 1u) != 0u;
        const T k0 = (b1 ? h4[2] : h4[0]) + dpp_mov<kDppXor2>(b1 ? h4[0] : h4[2]);
        const T k1 = (b1 ? h4[3] : h4[1]) + dpp_mov<kDppXor2>(b1 ? h4[1] : h4[3]);
        return (b0 ? k1 : k0) + dpp_mov<kDppXor1>(b0 ? k0 : k1);
    };
    // ONE pass over the packed inverse: u2 = I u1 and w = I s (s = the second entries of u1sg), eight lanes per row
    auto inverse_pass = [&](uint32_t P) __attribute__((always_inline)) {
        // (a lane's terms b = g, g + 8, ... are requested CH at a time before their fmas: few LDS round trips per row, addresses are
        // a base plus immediates; entries and operands beyond P are zeros.  fp64 takes the row in chunks — all at once would need
        // ~100 registers beside the Gram values)
        constexpr int CH = sizeof(T) == 8 ? 6 : PT;
        for (uint32_t a = grp; a < P; a += THREADS / 8u) {
            const T* row = L.I + (size_t)a * IP + g;
            const T* us = L.u1sg + 2u * g;
            T s1 = T(0), s2 = T(0);
#pragma nounroll
            for (int e0 = 0; e0 < PT; e0 += CH) {
                if ((uint32_t)(8 * e0) < P) {                     // (uniform)
                    T iv[CH];
                    V2 pr[CH];
#pragma unroll
                    for (int e = 0; e < CH; ++e) {
                        if (e0 + e < PT) {
                            iv[e] = row[8 * (e0 + e)];
                            pr[e] = *reinterpret_cast<const V2*>(&us[16 * (e0 + e)]);
                        } else { iv[e] = T(0); pr[e] = V2{ T(0), T(0) }; }
                    }
#pragma unroll
                    for (int e = 0; e < CH; ++e) {
                        s1 = res_fma<T>(iv[e], pr[e][0], s1);
                        s2 = res_fma<T>(iv[e], pr[e][1], s2);
                    }
                }
            }
            s1 = group8_sum<T>(s1); s2 = group8_sum<T>(s2);
            if (g == 0) { L.u2w[2 * a] = s1; L.u2w[2 * a + 1] = s2; }
        }
    };
    // u1 . u2 and u2 . s: every wave walks the same sums (nothing to publish)
    auto two_dots = [&](uint32_t P, T& d1, T& d2) __attribute__((always_inline)) {
        T a1 = T(0), a2 = T(0);
#pragma nounroll
        for (uint32_t b = lane; b < P; b += 64u) {
            const V2 pr = *reinterpret_cast<const V2*>(&L.u1sg[2 * b]);
            const T u2b = L.u2w[2 * b];
            a1 = res_fma<T>(pr[0], u2b, a1);
            a2 = res_fma<T>(u2b, pr[1], a2);
        }
        d1 = wave_sum<T>(a1); d2 = wave_sum<T>(a2);
    };
    uint32_t P = 0, status = 0, iter = 0, nlog = 0, reason = 0;
    T c_inf = T(0), lambda_prev = T(0), gamma_prev = T(0), lambda0 = T(0);
    int32_t mypos = -1;
    bool tie_any = false;
    lds_barrier();

    // ---- first pick (homotopy-cpu.cpp:217-229): the left-most largest |c0| of the subset; inv = [1 / ||a||^2] through the norm
    uint32_t sp0;
    {
        T bv = myvalid ? res_abs<T>(my_c) : T(-1);
        uint32_t bi = myvalid ? myj : 0xffffffffu;
        block_argmax(bv, bi);
        sp0 = bi;
        lambda0 = bv;
    }
    if (sp0 >= S) { status = kStatusSubsetDecline; reason |= kReasonNoCand; }
    else {
        const T g00 = enter_row(sp0, 0u, 0xffffffffu, T(0));
        if (owner && myj == sp0) {
            const T nrm = res_sqrt<T>(g00);
            const T inv00 = T(1) / (nrm * nrm);
            const T seed = strict_sign ? my_c : res_abs<T>(my_c);
            const T sg0 = sign_tol<T>(seed, tol);
            L.I[0] = inv00;
            // Homotopy: the first direction; OMP: the first least-squares coefficient (x = inv . c0_S) — d stays 0
            if (OMP) L.xvec[0] = inv00 * my_c; else L.dvec[0] = inv00 * sg0;
            L.pcol[0] = mycol; L.psub[0] = sp0; L.posof[sp0] = 0;
            mypos = 0;
            if (!OMP && trace != nullptr) { trace[0].idx = mycol; trace[0].added = 1; trace[0].gamma = 0.0; trace[0].c_inf = (double)res_abs<T>(my_c); }
        }
        P = 1;
        c_inf = lambda0;
        lambda_prev = lambda0;
        // (the tolerance guard of the Gram forms, k_la_init_pick's: the caller's usual engine decides what to do below it)
        if (!((double)tol >= (sizeof(T) == 8 ? kGramGuard64 : kGramGuard) * (double)lambda0)) { status = kStatusSubsetDecline; reason |= kReasonGuard; }
        lds_barrier();
    }

    // Both loops below have ONE exit, at the top of a round; whatever a round finds out after that (a removal, no candidate, the
    // positions spent) lowers the uniform flag `ok`, the rest of the round runs with its stores switched off and the next round's
    // top leaves.  (Every further `break` is, to the compiler, an exit its structurizer carries all loop state through — the
    // CT x PT Gram registers included — with a copy per exit: twice the registers, spilled in fp64.)
    if constexpr (OMP) {
        // ======== orthogonal matching pursuit (no reference counterpart; the statement of activeset.hip: k_la_omp) ===========
        // state k: support of k columns, x_S = inv . c0_S; c = c0 - sum_p x_p g_p; stop when ||c||_inf <= tol or the budget is spent;
        // otherwise the left-most largest |c| enters.  State 0 is the empty support (c = c0): its pick is the prologue's.
        if (status == 0u && !(lambda0 > tol)) { status = kStatusSubsetDecline; reason |= kReasonNoCand; }      // (nothing to pick: the usual engine reports it)
        if (status == 0u) {
            if (t == 0) {
                uint32_t* h = hdr;
                h[0] = 0u; h[1] = 1u; h[2] = L.subc[sp0]; h[3] = 1u; h[4] = __float_as_uint((float)lambda0); h[5] = 0u; h[6] = 0xffffffffu; h[7] = 0u;
                if (LH != nullptr) { LH[0] = lambda0; LH[1] = T(0); }
                if (trace != nullptr && 1u < trace_cap) { trace[1].idx = L.subc[sp0]; trace[1].added = 1u; trace[1].gamma = 0.0; trace[1].c_inf = (double)lambda0; }
            }
            if (t < PCAP) LX[t] = T(0);
            nlog = 1;
            iter = 1;
        }
        for (uint32_t round = 2; status == 0u; ++round) {
            my_c = my_c0 - chain(L.xvec, P);                        // c of this state
            T bv = T(-1);
            uint32_t bi = 0xffffffffu;
            if (myvalid) { bv = res_abs<T>(my_c); bi = myj; }
            block_argmax(bv, bi);                                   // ||c||_inf over the subset and its left-most column
            c_inf = bv;
            const bool stop = !(c_inf > tol) || round > max_iter;
            const bool logfull = nlog >= LOGCAP;
            const uint32_t spi = bi < S ? bi : 0u;
            const bool stall = !stop && (bi >= S || uni(L.posof[spi]) >= 0);      // (the largest |c| sits ON the support: a numerical stall)
            const bool full = !stop && (P >= PCAP || P + 1u > kcap);
            const bool ok = !stop && !logfull && !stall && !full;
            if (logfull) { status = kStatusSubsetDecline; reason |= kReasonLog; }
            else if (stall) { status = kStatusSubsetDecline; reason |= kReasonNoCand; }
            else if (full) { status = kStatusSubsetDecline; reason |= kReasonPositions; }
            else if (stop) { status = 0xffffffffu; iter = round - 1; }          // (the regular end: status set back to 0 below the loop)
            const uint32_t idx = uni(L.subc[spi]);
            if (!logfull) {
                if (t == 0) {
                    uint32_t* h = hdr + nlog * 8;
                    h[0] = P; h[1] = stop ? 0u : 1u; h[2] = ok ? idx : 0xffffffffu; h[3] = ok ? 1u : 0u; h[4] = __float_as_uint((float)c_inf); h[5] = 0u;
                    h[6] = 0xffffffffu; h[7] = 0u;
                    if (LH != nullptr) { LH[nlog * 2] = c_inf; LH[nlog * 2 + 1] = T(0); }
                    if (ok && trace != nullptr && round < trace_cap) { trace[round].idx = idx; trace[round].added = 1u; trace[round].gamma = 0.0; trace[round].c_inf = (double)c_inf; }
                }
                if (t < PCAP) LX[(size_t)nlog * PCAP + t] = t < P ? L.xvec[t] : T(0);
                ++nlog;
            }
            const T my_g = enter_row(spi, ok ? P : 0xffffffffu, 0xffffffffu, T(0));
            // u1 = G[idx][support], b_S = c0_S beside it (the pass forms u2 = I u1 and I b = x of the old support at once)
            if (owner && mypos >= 0) { L.u1sg[2 * mypos] = my_g; L.u1sg[2 * mypos + 1] = my_c0; }
            if (ok && owner && myj == spi) { s_f[0] = my_g; s_f[1] = my_c0; L.posof[spi] = (int32_t)P; L.pcol[P] = idx; L.psub[P] = spi; mypos = (int32_t)P; }
            lds_barrier();
            inverse_pass(P);
            lds_barrier();
            T d1, d2;
            two_dots(P, d1, d2);
            const T dvv = T(1) / (s_f[0] - d1);
            const T beta = dvv * (s_f[1] - d2);
            if (ok && t < P) L.xvec[t] = L.u2w[2 * t + 1] - beta * L.u2w[2 * t];
            if (ok && t == P) L.xvec[P] = beta;
            border_inverse(ok ? P : 0xffffffffu, dvv);
            lds_barrier();
            if (ok) { P += 1u; iter = round; }
        }
        if (status == 0xffffffffu) status = 0u;
    } else {
        // ======== Homotopy ================================================================================================
        if (status == 0u) my_q = chain(L.dvec, P);
        uint32_t pendP = 0xffffffffu;
        T pend_dv = T(0);
        for (uint32_t round = 1; status == 0u; ++round) {
            if constexpr (STAMPS) tlast = __builtin_readcyclecounter();
            // ---- loop control (homotopy-cpu.cpp:236, 272) and the log of this state
            const bool stop = (round > 1 && !(c_inf > tol)) || round > max_iter;
            const bool logfull = nlog >= LOGCAP;
            const bool in_band = tie_band<T>(c_inf, lambda_prev, gamma_prev, lambda0);
            // ---- find_max_gamma's scan (homotopy-cpu.cpp:122-163) over the subset, one column per owning lane
            T m = Lim<T>::max();
            bool tie = false;
            if (myvalid) {
                if (mypos >= 0) {
                    const T tt = -L.xvec[mypos] / L.dvec[mypos];
                    if (tt > T(0) && tt < m) m = tt;
                } else {
                    const T dl = T(1) - my_q, dr = T(1) + my_q;
                    if (dl != T(0)) {
                        T tt = (c_inf - my_c) / dl;
                        if (tie_guard && tt == T(0) && dl > T(0)) tt = Lim<T>::tiny();
                        if (tt == T(0) && in_band) tie = true;
                        if (tt > T(0) && tt < m) m = tt;
                    }
                    if (dr != T(0)) {
                        T tt = (c_inf + my_c) / dr;
                        if (tie_guard && tt == T(0) && dr > T(0)) tt = Lim<T>::tiny();
                        if (tt == T(0) && in_band) tie = true;
                        if (tt > T(0) && tt < m) m = tt;
                    }
                }
            }
            if (tie) s_tie = 1u;                                   // (read behind the reduction's barrier; cleared behind the next one)
            T gmm = m;
            uint32_t bi = myvalid ? myj : 0xffffffffu;             // (ascending subset: the left-most column is the smaller index)
            block_min(gmm, bi);                                    // ---- barrier 1
            RES_STAMP(0)
            if (!stop) tie_any = uni(s_tie) != 0u || tie_any;      // (the last state takes no step: its scan does not count)
            const uint32_t spi = bi < S ? bi : 0u;
            const bool tied = !stop && tie_any && tie_exit != 0;
            const bool nocand = !stop && !tied && (bi >= S || !(gmm < Lim<T>::max()));       // (no positive candidate: the reference toggles column 0)
            const bool leaves = !stop && !tied && !nocand && uni(L.posof[spi]) >= 0;           // (a column leaves: not this form's path)
            const bool full = !stop && !tied && !nocand && !leaves && (P >= PCAP || P + 1u > kcap);
            const bool ok = !stop && !logfull && !tied && !nocand && !leaves && !full;
            if (logfull) { status = kStatusSubsetDecline; reason |= kReasonLog; }
            else if (stop) { status = 0xffffffffu; iter = round - 1; }          // (the regular end: status set back to 0 below the loop)
            else if (tied) { status = kStatusTieRerun; reason |= kReasonTie; iter = round - 1; }
            else if (nocand) { status = kStatusSubsetDecline; reason |= kReasonNoCand; }
            else if (leaves) { status = kStatusSubsetDecline; reason |= kReasonRemoval; }
            else if (full) { status = kStatusSubsetDecline; reason |= kReasonPositions; }
            const uint32_t idx = uni(L.subc[spi]);
            if (!logfull) {
                if (t == 0) {
                    uint32_t* h = hdr + nlog * 8;
                    h[0] = P; h[1] = stop ? 0u : 1u; h[2] = ok ? idx : 0xffffffffu; h[3] = ok ? 1u : 0u; h[4] = __float_as_uint((float)c_inf);
                    h[5] = ok ? __float_as_uint((float)gmm) : 0u; h[6] = 0xffffffffu; h[7] = in_band ? 1u : 0u;
                    if (LH != nullptr) { LH[nlog * 2] = c_inf; LH[nlog * 2 + 1] = ok ? gmm : T(0); }
                    if (ok && trace != nullptr && round < trace_cap) { trace[round].idx = idx; trace[round].added = 1u; trace[round].gamma = (double)gmm; trace[round].c_inf = (double)c_inf; }
                }
                if (t < PCAP) {
                    LX[(size_t)nlog * PCAP + t] = t < P ? L.xvec[t] : T(0);
                    if (LD != nullptr) LD[(size_t)nlog * PCAP + t] = t < P ? L.dvec[t] : T(0);
                }
                if (stop || tied || ok) ++nlog;                    // (a declined round's state is not part of the path)
            }
            // (the entering column's Gram values of my columns: the round's only trip to memory, under the x and c updates)
            const T my_g = enter_row(spi, ok ? P : 0xffffffffu, pendP, pend_dv);
            pendP = 0xffffffffu;
            // ---- x += gamma d over the old support (homotopy-cpu.cpp:252); c of the next state and its largest magnitude
            if (ok && t < P) L.xvec[t] = L.xvec[t] + gmm * L.dvec[t];
            if (ok) my_c = res_fma<T>(-gmm, my_q, my_c);
            {
                const T mx = wave_max_val<T>(myvalid ? res_abs<T>(my_c) : T(-1));
                if (lane == 0) s_max[wave] = mx;
            }
            // ---- the column enters at position P: u1 = G[idx][support] and sign(c_S) of the correlations after the step, dead zone tol
            // (homotopy-cpu.cpp:257-267), side by side for the pass over the inverse
            if (owner && mypos >= 0) { L.u1sg[2 * mypos] = my_g; L.u1sg[2 * mypos + 1] = sign_tol<T>(my_c, tol); }
            if (ok && owner && myj == spi) {
                s_f[0] = my_g; s_f[1] = sign_tol<T>(my_c, tol);
                L.posof[spi] = (int32_t)P; L.pcol[P] = idx; L.psub[P] = spi; mypos = (int32_t)P;
            }
            lds_barrier();                                      // ---- barrier 2
            RES_STAMP(1)
            T c_next = s_max[0];
#pragma unroll
            for (uint32_t w = 1; w < NW; ++w) { const T o = s_max[w]; c_next = o > c_next ? o : c_next; }
            c_next = uni(c_next);
            if (t == 0) s_tie = 0u;
            inverse_pass(P);                                      // u2 = I u1 (online_inverse.h:218-225), w = I s
            lds_barrier();                                      // ---- barrier 3
            RES_STAMP(2)
            T d1, d2;
            two_dots(P, d1, d2);
            const T dvv = T(1) / (s_f[0] - d1);                    // online_inverse.h:228-231
            const T beta = dvv * (s_f[1] - d2);
            // the new direction: [w - beta u2; beta]
            if (ok && t < P) L.dvec[t] = L.u2w[2 * t + 1] - beta * L.u2w[2 * t];
            if (ok && t == P) L.dvec[P] = beta;
            pendP = ok ? P : 0xffffffffu;                          // (the bordered inverse: under the next round's memory trip — enter_row)
            pend_dv = dvv;
            lds_barrier();                                      // ---- barrier 4
            RES_STAMP(3)
            if (ok) {
                P += 1u;
                lambda_prev = c_inf;
                gamma_prev = gmm;
                c_inf = c_next;
                iter = round;
            }
            my_q = chain(L.dvec, P);
            RES_STAMP(4)
            if constexpr (STAMPS) tph[5] += 1;
        }
        if (status == 0xffffffffu) status = 0u;
    }
    if constexpr (STAMPS) {
        if (dbg != nullptr && slot == 0u && t == 0u)
            for (int q = 0; q < 6; ++q) dbg[q] = tph[q];
    }
#undef RES_STAMP
    lds_barrier();
    // ---- hand-over: the positions' columns, dense x, sorted support / touched lists, the slot's state ------------------------
    if (t < PCAP) log_pcol[(size_t)slot * PCAP + t] = t < P ? L.pcol[t] : 0xffffffffu;
    if (status == 0u || status == kStatusTieRerun) {
        if (t < P) {
            const uint32_t col = L.pcol[t];
            uint32_t rk = 0;
#pragma nounroll
            for (uint32_t b = 0; b < P; ++b) rk += L.pcol[b] < col ? 1u : 0u;
            gam_out[rk] = col;
            tch_out[rk] = col;
            x_out[col] = L.xvec[t];
        }
    }
    if (t == 0) {
        st->done = 1;
        st->status = status;
        st->iter = iter;
        st->K = P;
        st->ntouched = P;
        st->cur = 0;
        st->c_inf = (double)c_inf;
        st->gamma = (double)gamma_prev;
        st->tie_stall = tie_any ? 1u : 0u;
        st->lambda0 = (float)lambda0;
        st->solo_nlog = nlog;
        st->need_sweep = 0;
        st->done_round = iter + 1u;
        st->sub_reason = reason;
    }
    if (warm == T(1.2345e-30)) st->pad4_[0] = 1u;            // (keeps the warm-up loads alive; never true for a sum of Gram values)
}

// ======== fp64: the subset's Gram matrix Gs = A_S^T A_S (256 x 256) and the exact c0 of its columns ===========================
// One workgroup per (64 x 64 tile on or above the diagonal, row chunk); a 32 x 32 quadrant per wave on v_mfma_f64_16x16x4_f64
// (2 x 2 tiles); the columns' rows staged through LDS 32 at a time (256-byte runs per column), the next stage's loads in flight
// under the MFMAs.  Partials per row chunk, summed in order by k_sgram64_sum: deterministic.
constexpr uint32_t kG64T = 64, kG64Step = 32, kG64Pitch = kG64Step + 1;
typedef double res_v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256)
void k_sgram64_part(const double* __restrict__ At, uint32_t ldm, uint32_t n, const uint32_t* __restrict__ sub, uint32_t rows_per,
                    double* __restrict__ part)
{
    constexpr uint32_t S = ResCfg<double>::S;
    __shared__ double sI[kG64T][kG64Pitch];
    __shared__ double sJ[kG64T][kG64Pitch];
    sub += (size_t)blockIdx.z * S;                                // (blockIdx.z = slot of a batch: its subset, its partials)
    part += (size_t)blockIdx.z * gridDim.y * S * S;
    const uint32_t b = blockIdx.x;
    uint32_t tt = (uint32_t)((__fsqrt_rn(8.f * (float)b + 1.f) - 1.f) * 0.5f);
    while (tt * (tt + 1u) / 2u > b) --tt;
    while ((tt + 1u) * (tt + 2u) / 2u <= b) ++tt;
    const uint32_t bj = tt, bi = b - tt * (tt + 1u) / 2u;
    const uint32_t chunk = blockIdx.y;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t wi = w >> 1, wj = w & 1u;
    const uint32_t sc = tid >> 2, sq = tid & 3u;               // staging: column sc of the 64, rows 8 sq .. 8 sq + 7 of the 32
    const uint32_t ci = sub[bi * kG64T + sc], cj = sub[bj * kG64T + sc];
    const bool oki = ci < n, okj = cj < n;
    const double* gi = At + (size_t)(oki ? ci : 0u) * ldm + (size_t)chunk * rows_per + 8u * sq;
    const double* gj = At + (size_t)(okj ? cj : 0u) * ldm + (size_t)chunk * rows_per + 8u * sq;
    const v2d zero2 = { 0.0, 0.0 };
    v2d vi[4], vj[4];
#define SG64_LOAD(R0)                                                                         \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                           \
        vi[p] = oki ? *reinterpret_cast<const v2d*>(gi + (R0) + 2 * p) : zero2;               \
        vj[p] = okj ? *reinterpret_cast<const v2d*>(gj + (R0) + 2 * p) : zero2;               \
    }
    SG64_LOAD(0u)
    res_v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = res_v4d{ 0.0, 0.0, 0.0, 0.0 };
    const uint32_t l15 = lane & 15u, kq = lane >> 4;
    for (uint32_t r0 = 0; r0 < rows_per; r0 += kG64Step) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            sI[sc][8u * sq + 2u * (uint32_t)p] = vi[p][0]; sI[sc][8u * sq + 2u * (uint32_t)p + 1u] = vi[p][1];
            sJ[sc][8u * sq + 2u * (uint32_t)p] = vj[p][0]; sJ[sc][8u * sq + 2u * (uint32_t)p + 1u] = vj[p][1];
        }
        __syncthreads();
        if (r0 + kG64Step < rows_per) SG64_LOAD(r0 + kG64Step)
#pragma unroll
        for (uint32_t k4 = 0; k4 < kG64Step; k4 += 4) {
            const double a0 = sI[32u * wi + l15][k4 + kq], a1 = sI[32u * wi + 16u + l15][k4 + kq];
            const double b0 = sJ[32u * wj + l15][k4 + kq], b1 = sJ[32u * wj + 16u + l15][k4 + kq];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#undef SG64_LOAD
    double* Pp = part + (size_t)chunk * S * S;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t row = bi * kG64T + 32u * wi + 16u * (uint32_t)i + 4u * (uint32_t)e + kq;     // (A side: the output's row)
                const uint32_t col = bj * kG64T + 32u * wj + 16u * (uint32_t)j + l15;
                Pp[(size_t)row * S + col] = acc[i][j][e];
                if (bi != bj) Pp[(size_t)col * S + row] = acc[i][j][e];
            }
}

// gs = the chunks' partials added in order; the workgroups beyond that form the EXACT fp64 c0 = a_j . y of one subset column each
// (fixed order: a thread's strided terms ascending, then block_sum), written into the dense c0 at the column
__global__ __launch_bounds__(256)
void k_sgram64_sum(const double* __restrict__ part, uint32_t nsplit, double* __restrict__ gs, const double* __restrict__ At, uint32_t ldm,
                   const double* __restrict__ y, const uint32_t* __restrict__ sub, uint32_t n, double* __restrict__ c0, uint32_t c0_stride)
{
    constexpr uint32_t S = ResCfg<double>::S;
    constexpr uint32_t NB = S * S / 256u;
    {   // (blockIdx.y = slot of a batch)
        const size_t slot = blockIdx.y;
        part += slot * nsplit * S * S; gs += slot * S * S; y += slot * ldm; sub += slot * S; c0 += slot * c0_stride;
    }
    if (blockIdx.x >= NB) {
        __shared__ double sv[16];
        const uint32_t col = sub[blockIdx.x - NB];
        const double* a = At + (size_t)(col < n ? col : 0u) * ldm;
        double acc = 0.0;
        for (uint32_t r0 = 2u * threadIdx.x; r0 < ldm; r0 += 8u * 512u) {
            v2d av[8], yv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t r = r0 + 512u * (uint32_t)q;
                av[q] = r < ldm ? *reinterpret_cast<const v2d*>(a + r) : v2d{ 0.0, 0.0 };
                yv[q] = r < ldm ? *reinterpret_cast<const v2d*>(y + r) : v2d{ 0.0, 0.0 };
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { acc = __builtin_fma(av[q][0], yv[q][0], acc); acc = __builtin_fma(av[q][1], yv[q][1], acc); }
        }
        acc = block_sum(acc, sv);
        if (threadIdx.x == 0u && col < n) c0[col] = acc;
        return;
    }
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    double s = part[i];
    for (uint32_t c = 1; c < nsplit; ++c) s += part[(size_t)c * S * S + i];
    gs[i] = s;
}

// ---- r_k = y - A_S x_S(k) of every logged state k >= 1 in fp64, scaled by a power of two and rounded to fp16; ||r_k||^2 as
// per-workgroup partials; workgroup 0: the per-state table of the screening pass (bounds) and the certificate of state 0.
// The fp64 twin of k_scr_residuals (screen.hip) over k_res_solve<double>'s log; the support's columns come from the dictionary
// itself.  One workgroup per 64 rows; a thread owns 4 rows x 4 states x 3 rounds of 64 states; positions 16 at a time.
constexpr uint32_t kR64Rhs = 192;                 // states the residual block holds (= screen.hip's kS64Rhs)
__device__ __forceinline__ float res_state_scale(float lam_eff)
{
    int e = 12;
    if (lam_eff > 0.f && lam_eff < 3.0e38f) e = (int)floorf(log2f(4096.f / lam_eff));
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return ldexpf(1.f, e);
}

__global__ __launch_bounds__(256)
void k_res_residuals64(const double* __restrict__ At, uint32_t ldm, uint32_t n, const double* __restrict__ y, const uint32_t* __restrict__ hdr,
                       const double* __restrict__ LH, const uint32_t* __restrict__ pcol, const double* __restrict__ LX, double tol,
                       const float* __restrict__ meta, __half* __restrict__ r16, float* __restrict__ rn2p, float* __restrict__ tab,
                       uint32_t* __restrict__ headroom, DevState* __restrict__ st, int first16, int omp, const float* __restrict__ slotmeta,
                       uint32_t* __restrict__ fl, int scan)
{
    // scan: the rescue's look at a DECLINED solve's log (screen.hip: launch_screen64_rescue_scan; see k_scr_residuals) — the residuals of its
    // states as usual, irregular states and the one the path ended in switched off, nothing of the slot's state written
    // fl (one signal, may be null): the list of the exact re-check (screen.hip: k_scr_recheck), cleared here; a state 0 the first pass's
    // threshold cannot certify is left to it ([1024 + 1 .. + 3]: flag, bits(bound_0), bits(eps_0))
    constexpr uint32_t kFlCap = kScrFlCap;
    // slotmeta (a batch, may be null): per slot {||y||^2, what the columns left out of the subset stay below, 1 / s_y} of a first pass
    // that rounded the signals to fp16 as well (k_scr_gemm in its writing mode) — one signal: meta[5], meta[6]
    constexpr uint32_t PCAP = ResCfg<double>::PCAP;
    constexpr uint32_t LOGCAP = ResCfg<double>::LOGCAP;
    constexpr uint32_t XP = kR64Rhs + 8u;
    {
        const size_t slot = blockIdx.y;
        y += slot * ldm; hdr += slot * LOGCAP * 8; LH += slot * LOGCAP * 2; pcol += slot * PCAP; LX += slot * LOGCAP * PCAP;
        r16 += slot * kR64Rhs * ldm; rn2p += slot * (ldm / 64u) * kR64Rhs; tab += slot * kR64Rhs * 4u; st += slot;
        if (slotmeta != nullptr) slotmeta += slot * 4u;
    }
    __shared__ __attribute__((aligned(16))) double sAc[16][64];
    __shared__ __attribute__((aligned(16))) double sXt[16][XP];
    __shared__ float sS[kR64Rhs];
    if (st->status != 0u && !scan) return;
    const uint32_t tid = threadIdx.x, r0 = blockIdx.x * 64u;
    const uint32_t nlog = st->solo_nlog;
    float ratio0 = 0.f;
    if (fl != nullptr && blockIdx.x == 0u && blockIdx.y == 0u && tid == 0u) {
        fl[0] = 0u; fl[kFlCap + 1u] = 0u; fl[kFlCap + 4u] = 0u; fl[kFlCap + 12u] = 0u; fl[kFlCap + 13u] = 0u;
        if (!scan) fl[kFlCap + 14u] = 0u;                            // (the re-check's list of failed columns: a scan leaves the first attempt's alone)
    }
    if (first16 && !scan && blockIdx.x == 0u && tid == 0u && nlog >= 1u) {
        // state 0 after a first pass in half precision (k_scr_first): every column left out of the subset has |c~0| < T, so
        // |c0| < T + eps_0 — certified against lambda_0 (the subset's exact max |c0|) with the margin of every other state
        const float lam0 = (float)LH[0] * 0.9999999f;
        const float yn = sqrtf(slotmeta != nullptr ? slotmeta[0] : meta[5]) * 1.001f;
        // (one signal: the error model of the first pass that ran — k_scr_first / k_scr_first8; a batch: its ranking GEMM over the fp16 copy)
        float eps0 = slotmeta != nullptr ? 0.0019726562f * yn * meta[4] + 6.103515625e-05f * sqrtf((float)ldm) * yn * meta[1]
                                         : meta[8] * yn * meta[4] + meta[9] * sqrtf((float)ldm) * yn;
        if (slotmeta != nullptr) eps0 += 6.103515625e-05f * sqrtf((float)ldm) * meta[4] * slotmeta[2];     // (the signal was rounded too: its flush term)
        const float bound0 = lam0 * 0.875f - 1e-12f * lam0;
        const float v0 = (slotmeta != nullptr ? slotmeta[1] : meta[6]) + eps0;
        if (!(v0 <= bound0)) {
            if (fl != nullptr && bound0 > 0.f) { fl[kFlCap + 1u] = 1u; fl[kFlCap + 2u] = __float_as_uint(bound0); fl[kFlCap + 3u] = __float_as_uint(eps0); }
            else { __hip_atomic_store(&st->need_sweep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicOr(&st->sub_reason, kReasonFirstState); }
        }
        ratio0 = bound0 > 0.f && v0 == v0 ? v0 / bound0 : 3.0e38f;
    }
    if (nlog < 2u) {
        if (blockIdx.x == 0u && tid == 0u && blockIdx.y == 0u) *headroom = __float_as_uint(ratio0);
        return;
    }
    const uint32_t nst = nlog - 1u;                                // states 1 .. nst
    if (nst > kR64Rhs) {
        if (blockIdx.x == 0u && tid == 0u && !scan) { __hip_atomic_store(&st->need_sweep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicOr(&st->sub_reason, kReasonLog); }
        return;
    }
    const uint32_t Pfin = hdr[(size_t)(nlog - 1u) * 8u];
    if (tid < kR64Rhs) {
        float sc = 1.f;
        if (tid < nst) {
            const double lam = LH[(size_t)(tid + 1u) * 2u];
            const bool final_state = !(hdr[(size_t)(tid + 1u) * 8u + 1u] & 1u);
            sc = res_state_scale((float)(final_state ? fmax(lam, tol) : lam));
        }
        sS[tid] = sc;
    }
    const uint32_t rg = tid & 15u, sgp = tid >> 4;
    double acc[3][4][4];
    {
        const v2d y0 = *reinterpret_cast<const v2d*>(y + r0 + 4u * rg), y1 = *reinterpret_cast<const v2d*>(y + r0 + 4u * rg + 2u);
#pragma unroll
        for (int rd = 0; rd < 3; ++rd)
#pragma unroll
            for (int si = 0; si < 4; ++si) { acc[rd][si][0] = y0[0]; acc[rd][si][1] = y0[1]; acc[rd][si][2] = y1[0]; acc[rd][si][3] = y1[1]; }
    }
    for (uint32_t u0 = 0; u0 < Pfin; u0 += 16u) {
        __syncthreads();
        {   // 16 positions x 64 rows of their columns, 16 x nst coefficients (positions beyond a state's support carry x = 0 in the log)
            const uint32_t p = tid >> 4, q4 = tid & 15u;
            const uint32_t u = u0 + p;
            const uint32_t col = u < Pfin ? pcol[u] : 0xffffffffu;
            v2d a0 = { 0.0, 0.0 }, a1 = { 0.0, 0.0 };
            if (col < n) {
                a0 = *reinterpret_cast<const v2d*>(At + (size_t)col * ldm + r0 + 4u * q4);
                a1 = *reinterpret_cast<const v2d*>(At + (size_t)col * ldm + r0 + 4u * q4 + 2u);
            }
            *reinterpret_cast<v2d*>(&sAc[p][4u * q4]) = a0;
            *reinterpret_cast<v2d*>(&sAc[p][4u * q4 + 2u]) = a1;
            for (uint32_t e = tid; e < 16u * XP; e += 256u) {
                const uint32_t k = e / 16u, pp = e - k * 16u;          // (16 consecutive positions of a state: one 128-byte run of the log)
                sXt[pp][k] = (u0 + pp < Pfin && k < nst) ? LX[(size_t)(k + 1u) * PCAP + u0 + pp] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int rd = 0; rd < 3; ++rd) {
            const uint32_t s0 = 64u * (uint32_t)rd + 4u * sgp;
            if (s0 < nst) {                                           // (uniform over the 16 lanes of a state group)
#pragma unroll 4
                for (uint32_t p = 0; p < 16u; ++p) {
                    const v2d a0 = *reinterpret_cast<const v2d*>(&sAc[p][4u * rg]), a1 = *reinterpret_cast<const v2d*>(&sAc[p][4u * rg + 2u]);
                    const v2d x0 = *reinterpret_cast<const v2d*>(&sXt[p][s0]), x1 = *reinterpret_cast<const v2d*>(&sXt[p][s0 + 2u]);
                    const double av[4] = { a0[0], a0[1], a1[0], a1[1] }, xv[4] = { x0[0], x0[1], x1[0], x1[1] };
#pragma unroll
                    for (int si = 0; si < 4; ++si)
#pragma unroll
                        for (int ri = 0; ri < 4; ++ri) acc[rd][si][ri] = __builtin_fma(-xv[si], av[ri], acc[rd][si][ri]);
                }
            }
        }
    }
    bool ovf = false;
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) {
#pragma unroll
        for (int si = 0; si < 4; ++si) {
            const uint32_t kk = 64u * (uint32_t)rd + 4u * sgp + (uint32_t)si;
            const float sc = sS[kk < kR64Rhs ? kk : 0u];
            float ss = 0.f;
            __half hv[4];
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const float r = (float)acc[rd][si][ri];
                ss = __builtin_fmaf(r, r, ss);
                const float v = r * sc;
                if (kk < nst && !(fabsf(v) < 60000.f)) ovf = true;
                hv[ri] = __float2half_rn(v);
            }
            ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4); ss += __shfl_xor(ss, 8);
            if (kk < nst) {
                *reinterpret_cast<uint2*>(r16 + (size_t)kk * ldm + r0 + 4u * rg) = *reinterpret_cast<const uint2*>(hv);
                if (rg == 0u) rn2p[(size_t)blockIdx.x * kR64Rhs + kk] = ss * 1.0001f;     // (the cast of r to float: inside)
            }
        }
    }
    if (ovf && !scan) { __hip_atomic_store(&st->need_sweep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicOr(&st->sub_reason, kReasonOverflow); }
    if (blockIdx.x == 0u) {
        if (tid == 0u && blockIdx.y == 0u) *headroom = __float_as_uint(ratio0);
        if (tid < nst) {
            const uint32_t* hh = hdr + (size_t)(tid + 1u) * 8u;
            const uint32_t* hp = hdr + (size_t)tid * 8u;
            const double lam_d = LH[(size_t)(tid + 1u) * 2u];
            const double lam_pd = LH[(size_t)tid * 2u], gam_pd = LH[(size_t)tid * 2u + 1u];
            const float lam = (float)lam_d;
            const bool final_state = !(hh[1] & 1u);
            const float slack = 1e-12f * (float)LH[0];              // (fp64: the reference's own rounding is 1e-16)
            float bound;
            if (omp) {
                // OMP: state k's pick is the largest |c| — nothing outside the subset may reach it; the final state (||c|| <= tol: no
                // pick) against the tolerance
                bound = (final_state && !(lam_d > tol) ? (float)tol * 0.9375f : lam * 0.875f) - slack;
            } else {
                // (where the step INTO this state left lambda: lambda_{k-1} - gamma_{k-1}; the smaller of that and max |c| counts:
                // k_scr_residuals)
                const double lam_exp = lam_pd - gam_pd;
                const bool ls_jump = final_state && !(lam_d > tol) && !(lam_exp > 1e-13 * LH[0]);
                if (ls_jump) bound = (float)tol * 0.9375f - slack;
                else bound = fminf(lam, (float)lam_exp) * 0.875f - slack;
                // only REGULAR paths are certified: every step inserts a column, lambda goes down
                const bool irregular = hp[3] == 0u || lam_d > lam_pd * (1.0 + 1e-12);
                if (irregular && !scan) {
                    bound = -1.f;
                    atomicOr(&st->sub_reason, kReasonIrregular);
                    __hip_atomic_store(&st->need_sweep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (no screening pass for it)
                }
                if (scan && (irregular || final_state)) bound = 3.0e38f;
            }
            const float inv_sk = 1.f / sS[tid];
            tab[tid * 4u + 0] = meta[1] * inv_sk;
            tab[tid * 4u + 1] = bound;
            tab[tid * 4u + 2] = inv_sk;
            tab[tid * 4u + 3] = lam;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
template <typename T>
bool res_solve_usable()
{
    static const bool ok = [] {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_res_solve<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)res_lds_bytes<T>());
        const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_res_solve<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)res_lds_bytes<T>());
        if (e != hipSuccess || e2 != hipSuccess) (void)hipGetLastError();
        return e == hipSuccess && e2 == hipSuccess;
    }();
    return ok;
}
template bool res_solve_usable<float>();
template bool res_solve_usable<double>();

template <typename T>
hipError_t launch_res_solve(ss_hip_ctx* ctx, uint32_t nslots, const T* Gs, uint32_t gpitch, size_t g_slot_stride, const T* c0, uint32_t c0_stride,
                            const uint32_t* sub, T tol, uint32_t max_iter, uint32_t kcap, const ResLog<T>& log, T* x, uint32_t x_stride,
                            uint32_t* gam2, uint32_t* touched2, DevState* st, TraceEntry* trace, uint32_t trace_cap, bool omp)
{
    if (!res_solve_usable<T>()) return hipErrorInvalidConfiguration;
    static unsigned long long* dbg = nullptr;
    static const bool stamps = std::getenv("SS_HIP_RES_STAMPS") != nullptr;
    if (stamps && !omp) {
        static const bool ok = [] {
            const bool a = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_res_solve<T, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)res_lds_bytes<T>()) == hipSuccess;
            return a && hipMalloc(&dbg, 8 * sizeof(unsigned long long)) == hipSuccess;
        }();
        if (ok) {
            hipLaunchKernelGGL((k_res_solve<T, false, true>), dim3(nslots), dim3(ResCfg<T>::THREADS), res_lds_bytes<T>(), ctx->stream, Gs, gpitch, g_slot_stride, c0, c0_stride,
                               sub, (uint32_t)ctx->n, tol, max_iter, ctx->strict_sign, ctx->tie_guard, (ctx->tie_rerun && !ctx->tie_guard) ? 1 : 0, kcap,
                               log.hdr, log.H, log.pcol, log.X, log.D, x, x_stride, gam2, touched2, st, trace, trace_cap, dbg);
            unsigned long long tp[6];
            if (hipMemcpyAsync(tp, dbg, sizeof(tp), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess) {
                const double r = tp[5] ? (double)tp[5] : 1.0;
                std::fprintf(stderr, "[k_res_solve<%s>, cycles per round over %llu rounds] scan+argmin %.0f  row+x+c %.0f  inverse pass %.0f  dots+dir+border %.0f  chain %.0f\n",
                             sizeof(T) == 8 ? "double" : "float", tp[5], tp[0] / r, tp[1] / r, tp[2] / r, tp[3] / r, tp[4] / r);
            }
            return hipGetLastError();
        }
    }
    if (omp)
        hipLaunchKernelGGL((k_res_solve<T, true>), dim3(nslots), dim3(ResCfg<T>::THREADS), res_lds_bytes<T>(), ctx->stream, Gs, gpitch, g_slot_stride, c0, c0_stride,
                           sub, (uint32_t)ctx->n, tol, max_iter, ctx->strict_sign, ctx->tie_guard, (ctx->tie_rerun && !ctx->tie_guard) ? 1 : 0, kcap,
                           log.hdr, log.H, log.pcol, log.X, log.D, x, x_stride, gam2, touched2, st, trace, trace_cap);
    else
        hipLaunchKernelGGL((k_res_solve<T, false>), dim3(nslots), dim3(ResCfg<T>::THREADS), res_lds_bytes<T>(), ctx->stream, Gs, gpitch, g_slot_stride, c0, c0_stride,
                           sub, (uint32_t)ctx->n, tol, max_iter, ctx->strict_sign, ctx->tie_guard, (ctx->tie_rerun && !ctx->tie_guard) ? 1 : 0, kcap,
                           log.hdr, log.H, log.pcol, log.X, log.D, x, x_stride, gam2, touched2, st, trace, trace_cap);
    return hipGetLastError();
}
template hipError_t launch_res_solve<float>(ss_hip_ctx*, uint32_t, const float*, uint32_t, size_t, const float*, uint32_t, const uint32_t*, float, uint32_t,
                                            uint32_t, const ResLog<float>&, float*, uint32_t, uint32_t*, uint32_t*, DevState*, TraceEntry*, uint32_t, bool);
template hipError_t launch_res_solve<double>(ss_hip_ctx*, uint32_t, const double*, uint32_t, size_t, const double*, uint32_t, const uint32_t*, double, uint32_t,
                                             uint32_t, const ResLog<double>&, double*, uint32_t, uint32_t*, uint32_t*, DevState*, TraceEntry*, uint32_t, bool);

hipError_t launch_sgram64(ss_hip_ctx* ctx, const uint32_t* sub, const double* y, double* part, double* gs, double* c0, uint32_t nslots, uint32_t c0_stride)
{
    constexpr uint32_t S = ResCfg<double>::S, NT = S / kG64T;
    const uint32_t ldm = ctx->ldm;
    uint32_t nsplit = ldm / 512u;                                // (ldm is a multiple of 256)
    if (nsplit == 0u || ldm % (nsplit * kG64Step) != 0u) nsplit = ldm / 256u;
    if (nsplit > kSg64MaxSplit) nsplit = kSg64MaxSplit;
    while (nsplit > 1u && ldm % (nsplit * kG64Step) != 0u) --nsplit;
    const double* At = static_cast<const double*>(ctx->At);
    // (a slot's partials take nsplit * S * S doubles: the caller's `part` holds kSg64MaxSplit * S * S per slot — slots are packed by nsplit)
    hipLaunchKernelGGL(k_sgram64_part, dim3(NT * (NT + 1) / 2, nsplit, nslots), dim3(256), 0, ctx->stream, At, ldm, (uint32_t)ctx->n, sub, ldm / nsplit, part);
    hipLaunchKernelGGL(k_sgram64_sum, dim3(S * S / 256u + S, nslots), dim3(256), 0, ctx->stream, (const double*)part, nsplit, gs, At, ldm, y, sub, (uint32_t)ctx->n, c0,
                       c0_stride);
    return hipGetLastError();
}

hipError_t launch_res_residuals64(ss_hip_ctx* ctx, const double* y, const ResLog<double>& log, double tol, const float* meta, void* r16, float* rn2p,
                                  float* tab, uint32_t* headroom, DevState* st, bool first16, bool omp, uint32_t nslots, const float* slotmeta, uint32_t* fl, bool scan)
{
    hipLaunchKernelGGL(k_res_residuals64, dim3(ctx->ldm / 64u, nslots), dim3(256), 0, ctx->stream, static_cast<const double*>(ctx->At), ctx->ldm, (uint32_t)ctx->n, y,
                       (const uint32_t*)log.hdr, (const double*)log.H, (const uint32_t*)log.pcol, (const double*)log.X, tol, meta,
                       static_cast<__half*>(r16), rn2p, tab, headroom, st, first16 ? 1 : 0, omp ? 1 : 0, slotmeta, fl, scan ? 1 : 0);
    return hipGetLastError();
}

}  // namespace sship
