// subbatch.hip — the SUBSET form of the batched Gram form (fp32, G = A^T A resident): every signal of a batch is solved
// by ONE workgroup on a subset of 448 columns, then checked against all n columns.
//
// The lock-step Gram form (activeset.hip: k_la_cqs) re-derives c = c0 - G_S^T x_S and q = G_S^T d_S of ALL n columns
// in every round: K rows of G per signal and round, 545 MB per signal at 8192 x 65536 with k = 64 — 2.2 TB per 4096
// signals, 0.43 s at 0.82 of the HBM peak.  But a round only needs those values to (a) take max |c| and (b) find the
// smallest step-length candidate (find_max_gamma, /root/reference/src/solvers/homotopy-cpu.cpp:100-164), and both are
// decided among a handful of columns.  So, per signal:
//
//   k_sub_select   the 448 columns with the largest |c0| (two-level radix select, ties by index: deterministic),
//                  sorted by column; the first pick (ixamax of |c0|, homotopy-cpu.cpp:217-221) is among them;
//   k_sub_solve    ONE workgroup runs the whole path on that subset: Gram rows restricted to the subset (gathered
//                  from G as columns enter, 1.75 KiB each), the explicit inverse (online_inverse.h:183-293) and all
//                  vectors in LDS; every breakpoint is logged (positions' coefficients x, d; lambda; step; pick);
//   k_sub_verify   for every column OUTSIDE the subset and every logged breakpoint: c and q by the same chain of fmas
//                  from the same rows of G, then the reference's predicates — |c| must not exceed lambda, no candidate
//                  may beat the logged step (or tie it from the left).  One pass over the K rows of G serves all
//                  breakpoints: 16.8 MB per signal instead of 545.
//
// (The last step of a path that ends by tolerance is the one place where the check allows a tie: there every column's
// candidate equals the step within rounding, whichever column is inserted enters with x = 0 and the path ends in the
// next round — k_sub_verify.)
// A signal whose check fails, or that leaves the common path (a column leaves the support with a rounding residue in
// reference mode, more than 72 columns ever entered, more than 80 breakpoints, no positive candidate, an emptied
// support), is DECLINED: nothing of it is reported, the host solves it again in the lock-step form (homotopy.hip).
// An exact tie (DevState::tie_stall) goes to the reference-order engine as in every other form.
//
// Arithmetic of this form (both kernels, bit for bit the same): positions 0, 1, 2, ... are given out in order of
// entry and never re-used (a column that left keeps its position with coefficients 0; re-entering takes a new one);
//   c_j = c0_j, then  c_j = fma(-x_p, G[col_p][j], c_j)  for p = 0 .. P-1;   q_j = 0, then  q_j = fma(d_p, G[col_p][j], q_j).
// Everything else follows the lock-step kernels (k_scansel's predicates, select_toggle's update, sign dead zone).
#include "ss_hip_internal.h"
#include "ss_hip_device.h"
#include "resident.h"
#include "subcheck.h"

#include <algorithm>
#include <cstdlib>

namespace sship {

// (kSbS columns per subset = threads of k_sub_solve; kSbRows positions; kSbLog breakpoints: ss_hip_internal.h)
constexpr uint32_t kSbInvPitch = kSbRows + 1;
constexpr uint32_t kSelThreads = 512;
constexpr uint32_t kSelBins = 2048;

__device__ __forceinline__ uint32_t mag_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// ---- k_sub_select: the kSbS largest |c0| of every slot, sorted by column ------------------------------------------
// Two levels of 11 bits of the magnitude give the threshold key (22 bits); the columns above it and, left-most first, as
// many of the threshold key as are still needed are written out in column order.  Every pass reads c0 coalesced: wave w
// owns the columns [w * seg, (w + 1) * seg) and walks them 64 at a time (the first version gave every thread a contiguous
// chunk: 64 cache lines per load instruction, 20 ms per 4096 signals against 8 for their solves).
__global__ __launch_bounds__(kSelThreads)
void k_sub_select(const float* __restrict__ c0, uint32_t n, uint32_t n_pad, uint32_t* __restrict__ sub,
                  uint32_t* __restrict__ first_pick, float* __restrict__ first_val, uint32_t nsel)
{
    // nsel: columns to select per slot (kSbS for the subset form; the fp64 screened form takes more: screen.hip)
    constexpr uint32_t NW = kSelThreads / 64u;
    __shared__ uint32_t hist[kSelBins];
    __shared__ float sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_bin, s_above;
    __shared__ uint32_t w_sel[NW], w_eq[NW];
    const uint32_t slot = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    c0 += (size_t)slot * n_pad;
    sub += (size_t)slot * nsel;
    const uint32_t seg = ((n + NW - 1) / NW + 63u) & ~63u;
    const uint32_t lo = wave * seg < n ? wave * seg : n, hi = lo + seg < n ? lo + seg : n;
    const uint32_t want = n < nsel ? n : nsel;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    uint32_t prefix_key = 0, above = 0;
    for (int level = 0; level < 2; ++level) {
        for (uint32_t b = t; b < kSelBins; b += kSelThreads) hist[b] = 0u;
        __syncthreads();
        float bv = -1.f;
        uint32_t bi = 0xffffffffu;
        for (uint32_t i0 = lo; i0 < hi; i0 += 256u) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + 64u * (uint32_t)u + lane; v[u] = i < hi ? c0[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + 64u * (uint32_t)u + lane;
                if (i >= hi) continue;
                const uint32_t m = mag_bits(v[u]);
                if (level == 0) {
                    atomicAdd(&hist[m >> 20], 1u);
                    const float a = fabsf(v[u]);
                    if (better_max(a, i, bv, bi)) { bv = a; bi = i; }
                } else if ((m >> 20) == prefix_key) atomicAdd(&hist[(m >> 9) & 0x7ffu], 1u);
            }
        }
        if (level == 0) {
            // ixamax of |c0| (left-most): the first pick
            block_reduce_pair<float, true>(bv, bi, sv, si);
            if (t == 0) { first_pick[slot] = bi == 0xffffffffu ? 0u : bi; first_val[slot] = bv; }
        }
        __syncthreads();
        // thread t owns bins (from the top) 4t .. 4t+3; the crossing of `want` by the running count from the top
        uint32_t mine = 0;
        for (uint32_t u = 0; u < 4; ++u) mine += hist[kSelBins - 1u - (4u * t + u)];
        // exclusive prefix over the threads: within the wave by shuffles, across waves through LDS
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += up; }
        if (lane == 63) w_sel[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (uint32_t w = 0; w < wave; ++w) before += w_sel[w];
        if (above + before < want && above + before + mine >= want) {
            uint32_t acc = above + before;
            for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t b = kSelBins - 1u - (4u * t + u);
                if (acc + hist[b] >= want) { s_bin = b; s_above = acc; break; }
                acc += hist[b];
            }
        }
        __syncthreads();
        if (level == 0) prefix_key = s_bin; else prefix_key = (prefix_key << 11) | s_bin;
        above = s_above;
        __syncthreads();
    }
    const uint32_t T22 = prefix_key;                 // key = magnitude >> 9
    const uint32_t need_eq = want - above;           // columns taken from the threshold key, left-most first

    // per wave: columns above the key, columns of the key
    uint32_t n_sel = 0, n_eq = 0;
    for (uint32_t i0 = lo; i0 < hi; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const uint32_t k = i < hi ? mag_bits(c0[i]) >> 9 : 0u;
        n_sel += (uint32_t)__popcll(__ballot(i < hi && k > T22));
        n_eq += (uint32_t)__popcll(__ballot(i < hi && k == T22));
    }
    if (lane == 0) { w_sel[wave] = n_sel; w_eq[wave] = n_eq; }
    __syncthreads();
    uint32_t eq_before = 0, out = 0;
    for (uint32_t w = 0; w < wave; ++w) {
        const uint32_t tk = eq_before < need_eq ? (need_eq - eq_before < w_eq[w] ? need_eq - eq_before : w_eq[w]) : 0u;
        out += w_sel[w] + tk;
        eq_before += w_eq[w];
    }
    const uint32_t take = eq_before < need_eq ? (need_eq - eq_before < n_eq ? need_eq - eq_before : n_eq) : 0u;
    uint32_t eq_run = 0;
    for (uint32_t i0 = lo; i0 < hi; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const uint32_t k = i < hi ? mag_bits(c0[i]) >> 9 : 0u;
        const bool is_eq = i < hi && k == T22;
        const uint64_t beq = __ballot(is_eq);
        const uint32_t rank_eq = eq_run + (uint32_t)__popcll(beq & lt_mask);
        const bool chosen = (i < hi && k > T22) || (is_eq && rank_eq < take);
        const uint64_t bch = __ballot(chosen);
        const uint32_t pos = out + (uint32_t)__popcll(bch & lt_mask);
        if (chosen && pos < nsel) sub[pos] = i;
        out += (uint32_t)__popcll(bch);
        eq_run += (uint32_t)__popcll(beq);
    }
    // (fewer than kSbS columns in all: the rest of the list is "no column")
    for (uint32_t p = want + t; p < nsel; p += kSelThreads) sub[p] = 0xffffffffu;
}

// ---- k_sub_select1: the same selection for ONE slot, for latency ----------------------------------------------------
// k_sub_select walks c0 four times with four loads in flight per thread: fine with 4096 slots in the grid, 118 us for one.
// Here one workgroup of 1024 threads walks the (<= 65536) values with 16 coalesced 16-byte loads per thread all in flight
// together (one L2 round trip per walk; holding the 64 values of a thread in registers across the levels spills at 1024
// threads); the chosen columns go through two LDS bit masks (above the threshold key / of the key) and are written out in
// column order from there.
// Same rule, same result: the kSbS largest by the 22-bit key, left-most first within the threshold key.
constexpr uint32_t kSel1Threads = 1024, kSel1J = 16;
__global__ __launch_bounds__(kSel1Threads)
void k_sub_select1(const float* __restrict__ c0, uint32_t n, uint32_t n_pad, uint32_t* __restrict__ sub,
                   uint32_t* __restrict__ first_pick, float* __restrict__ first_val, float* __restrict__ thr_out,
                   const float* __restrict__ wmax, uint32_t nwmax)
{
    // wmax != nullptr (screen.hip's half-precision first pass): the values are a RANKING aid, not the path's c0, and the pass
    // left the maxima of its waves' columns in wmax[nwmax] — elements, so the floor comes from those without a walk — and the
    // selection stops at the 11-bit key (exponent + 3 mantissa bits): the 448 largest by that key, left-most first within the
    // threshold key; thr_out = the end of that key's range.  No first pick (k_sub_solve takes it from the subset's exact c0).
    const bool coarse = wmax != nullptr;
    const uint32_t kshift = coarse ? 20u : 9u;
    constexpr uint32_t NW = kSel1Threads / 64u;
    __shared__ uint32_t hist[kSelBins];
    __shared__ uint32_t m_sel[kSel1Threads * 2u], m_eq[kSel1Threads * 2u];
    __shared__ float sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_bin, s_above;
    __shared__ uint32_t w_tot[NW];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t want = n < kSbS ? n : kSbS;
    m_sel[2u * t] = 0u; m_sel[2u * t + 1u] = 0u; m_eq[2u * t] = 0u; m_eq[2u * t + 1u] = 0u;
    // exclusive prefix of a per-thread count over the workgroup (in thread order) and the total
    auto block_excl = [&](uint32_t mine, uint32_t& total) -> uint32_t {
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += up; }
        __syncthreads();
        if (lane == 63u) w_tot[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine, tot = 0u;
        for (uint32_t w = 0; w < NW; ++w) { if (w < wave) before += w_tot[w]; tot += w_tot[w]; }
        total = tot;
        return before;
    };
    // one walk over the values: 16 coalesced 16-byte loads per thread, all in flight together (c0 sits in L2: the sweep wrote it)
#define SEL1_WALK(BODY)                                                                        \
    {                                                                                          \
        v4f v_[kSel1J];                                                                        \
        _Pragma("unroll") for (uint32_t j_ = 0; j_ < kSel1J; ++j_) {                           \
            const uint32_t base_ = 4u * (j_ * kSel1Threads + t);                               \
            v_[j_] = base_ < n_pad ? *reinterpret_cast<const v4f*>(c0 + base_) : v4f{ 0.f, 0.f, 0.f, 0.f }; \
        }                                                                                      \
        _Pragma("unroll") for (uint32_t j_ = 0; j_ < kSel1J; ++j_) {                           \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                 \
                const uint32_t i = 4u * (j_ * kSel1Threads + t) + (uint32_t)e_;                \
                if (i < n) { const float val = v_[j_][e_]; const uint32_t m = mag_bits(val); BODY } \
            }                                                                                  \
        }                                                                                      \
    }

    // Level -1: a floor for the threshold from the THREADS' maxima — every one of them is an element, so the bin in which
    // their count from the top reaches `want` is at or below the threshold's bin, and only elements from that bin up
    // (a few hundred to a few thousand of 65536) enter the histograms.  (All 65536 in one histogram: their magnitudes share a
    // handful of bins and the LDS atomics on those serialise.)
    uint32_t prefix_key = 0, above = 0, floor_bin = 0;
    for (int level = -1; level < (coarse ? 1 : 2); ++level) {
        hist[2u * t] = 0u; hist[2u * t + 1u] = 0u;
        __syncthreads();
        if (level == -1 && coarse) {
            for (uint32_t i = t; i < nwmax; i += kSel1Threads) atomicAdd(&hist[mag_bits(wmax[i]) >> 20], 1u);
            if (nwmax < want) floor_bin = 0u;
        } else if (level == -1) {
            float bv = -1.f;
            uint32_t bi = 0xffffffffu, mmax = 0u;
            bool any = false;
            SEL1_WALK({ mmax = m > mmax ? m : mmax; any = true; const float a = fabsf(val); if (better_max(a, i, bv, bi)) { bv = a; bi = i; } })
            if (any) atomicAdd(&hist[mmax >> 20], 1u);
            // ixamax of |c0| (left-most): the first pick
            block_reduce_pair<float, true>(bv, bi, sv, si);
            if (t == 0) { first_pick[0] = bi == 0xffffffffu ? 0u : bi; first_val[0] = bv; }
        } else if (level == 0) {
            SEL1_WALK({ (void)val; if ((m >> 20) >= floor_bin) atomicAdd(&hist[m >> 20], 1u); })
        } else {
            SEL1_WALK({ (void)val; if ((m >> 20) == prefix_key) atomicAdd(&hist[(m >> 9) & 0x7ffu], 1u); })
        }
        __syncthreads();
        // thread t owns bins (from the top) 2t, 2t + 1: the crossing of `want` by the running count from the top
        const uint32_t h0 = hist[kSelBins - 1u - 2u * t], h1 = hist[kSelBins - 2u - 2u * t];
        uint32_t total = 0;
        const uint32_t base_above = level == 1 ? above : 0u;
        const uint32_t before = block_excl(h0 + h1, total);
        if (t == 0) { s_bin = 0u; s_above = 0u; }               // (level -1 with fewer than `want` thread maxima: no floor)
        __syncthreads();
        if (base_above + before < want && base_above + before + h0 + h1 >= want) {
            const uint32_t acc = base_above + before;
            if (acc + h0 >= want) { s_bin = kSelBins - 1u - 2u * t; s_above = acc; }
            else { s_bin = kSelBins - 2u - 2u * t; s_above = acc + h0; }
        }
        __syncthreads();
        if (level == -1) floor_bin = s_bin;
        else if (level == 0) { prefix_key = s_bin; above = s_above; }
        else { prefix_key = (prefix_key << 11) | s_bin; above = s_above; }
        __syncthreads();
    }
    const uint32_t T22 = prefix_key;                              // (coarse: the 11-bit key)
    const uint32_t need_eq = want - above;
    // (what every column left out stays below: the end of the threshold key's range — screen.hip's half-precision first pass)
    if (thr_out != nullptr && t == 0) thr_out[0] = __uint_as_float(T22 + 1u >= (0x7f800000u >> kshift) ? 0x7f800000u : (T22 + 1u) << kshift);
    SEL1_WALK({ (void)val; const uint32_t k = m >> kshift;
                if (k > T22) atomicOr(&m_sel[i >> 5], 1u << (i & 31u));
                else if (k == T22) atomicOr(&m_eq[i >> 5], 1u << (i & 31u)); })
#undef SEL1_WALK
    __syncthreads();
    // thread t owns the columns 64 t .. 64 t + 63
    const uint64_t sel64 = (uint64_t)m_sel[2u * t] | ((uint64_t)m_sel[2u * t + 1u] << 32);
    uint64_t eq64 = (uint64_t)m_eq[2u * t] | ((uint64_t)m_eq[2u * t + 1u] << 32);
    uint32_t tot = 0;
    const uint32_t eq_before = block_excl((uint32_t)__popcll(eq64), tot);
    uint32_t take = eq_before < need_eq ? need_eq - eq_before : 0u;
    uint64_t eq_take = 0ull;
    while (take != 0u && eq64 != 0ull) { const uint64_t low = eq64 & (0ull - eq64); eq_take |= low; eq64 ^= low; --take; }
    uint64_t chosen = sel64 | eq_take;
    uint32_t out = block_excl((uint32_t)__popcll(chosen), tot);
    while (chosen != 0ull) {
        const uint32_t b = (uint32_t)__builtin_ctzll(chosen);
        chosen &= chosen - 1ull;
        if (out < kSbS) sub[out] = 64u * t + b;
        ++out;
    }
    for (uint32_t p = want + t; p < kSbS; p += kSel1Threads) sub[p] = 0xffffffffu;
}

// ---- k_sub_select1w: ONE slot, dictionaries wider than 65536 columns -------------------------------------------------
// k_sub_select1's walks in chunks of 65536 columns; the chosen columns go through two short LDS lists instead of bit masks
// (which would need n bits): the < 448 above the threshold key and the columns OF the key, of which the left-most are
// taken — by counting ranks, the lists are short.  Should the key's list overflow (thousands of equal magnitudes) the
// columns are written out by an ordered walk instead (two block-wide prefix sums per 4096 columns: slow, and only then).
constexpr uint32_t kSel1ECap = 2048, kSel1SCap = 2048;          // (the 11-bit key's threshold bin holds up to ~1000 columns)
__global__ __launch_bounds__(kSel1Threads)
void k_sub_select1w(const float* __restrict__ c0, uint32_t n, uint32_t n_pad, uint32_t* __restrict__ sub,
                    uint32_t* __restrict__ first_pick, float* __restrict__ first_val, uint32_t nsel, float* __restrict__ thr_out,
                    const float* __restrict__ wmax, uint32_t nwmax)
{
    // (wmax != nullptr: the ranking after screen.hip's half-precision first pass — floor from the pass's wave maxima, 11-bit key,
    // no first pick: see k_sub_select1)
    const bool coarse = wmax != nullptr;
    const uint32_t kshift = coarse ? 20u : 9u;
    // nsel <= kSel1SCap columns are selected (kSbS for the subset form; the fp64 screened form's sub-dictionary takes 2048)
    constexpr uint32_t NW = kSel1Threads / 64u;
    constexpr uint32_t CH = 4u * kSel1J * kSel1Threads;          // columns per chunk of a walk
    __shared__ uint32_t hist[kSelBins];
    __shared__ uint32_t s_list[kSel1SCap], e_list[kSel1ECap];
    __shared__ float sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_bin, s_above, s_ns, s_ne;
    __shared__ uint32_t w_tot[NW];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t want = n < nsel ? n : nsel;
    auto block_excl = [&](uint32_t mine, uint32_t& total) -> uint32_t {
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += up; }
        __syncthreads();
        if (lane == 63u) w_tot[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine, tot = 0u;
        for (uint32_t w = 0; w < NW; ++w) { if (w < wave) before += w_tot[w]; tot += w_tot[w]; }
        total = tot;
        return before;
    };
#define SEL1W_WALK(BODY)                                                                       \
    for (uint32_t cb_ = 0; cb_ < n_pad; cb_ += CH) {                                           \
        v4f v_[kSel1J];                                                                        \
        _Pragma("unroll") for (uint32_t j_ = 0; j_ < kSel1J; ++j_) {                           \
            const uint32_t base_ = cb_ + 4u * (j_ * kSel1Threads + t);                         \
            v_[j_] = base_ < n_pad ? *reinterpret_cast<const v4f*>(c0 + base_) : v4f{ 0.f, 0.f, 0.f, 0.f }; \
        }                                                                                      \
        _Pragma("unroll") for (uint32_t j_ = 0; j_ < kSel1J; ++j_) {                           \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                 \
                const uint32_t i = cb_ + 4u * (j_ * kSel1Threads + t) + (uint32_t)e_;          \
                if (i < n) { const float val = v_[j_][e_]; const uint32_t m = mag_bits(val); BODY } \
            }                                                                                  \
        }                                                                                      \
    }
    uint32_t prefix_key = 0, above = 0, floor_bin = 0;
    for (int level = -1; level < (coarse ? 1 : 2); ++level) {
        hist[2u * t] = 0u; hist[2u * t + 1u] = 0u;
        __syncthreads();
        if (level == -1 && coarse) {
            for (uint32_t i = t; i < nwmax; i += kSel1Threads) atomicAdd(&hist[mag_bits(wmax[i]) >> 20], 1u);
        } else if (level == -1) {
            float bv = -1.f;
            uint32_t bi = 0xffffffffu, mmax = 0u;
            bool any = false;
            SEL1W_WALK({ mmax = m > mmax ? m : mmax; any = true; const float a = fabsf(val); if (better_max(a, i, bv, bi)) { bv = a; bi = i; } })
            if (any) atomicAdd(&hist[mmax >> 20], 1u);
            block_reduce_pair<float, true>(bv, bi, sv, si);
            if (t == 0) { first_pick[0] = bi == 0xffffffffu ? 0u : bi; first_val[0] = bv; }
        } else if (level == 0) {
            SEL1W_WALK({ (void)val; if ((m >> 20) >= floor_bin) atomicAdd(&hist[m >> 20], 1u); })
        } else {
            SEL1W_WALK({ (void)val; if ((m >> 20) == prefix_key) atomicAdd(&hist[(m >> 9) & 0x7ffu], 1u); })
        }
        __syncthreads();
        const uint32_t h0 = hist[kSelBins - 1u - 2u * t], h1 = hist[kSelBins - 2u - 2u * t];
        uint32_t total = 0;
        const uint32_t base_above = level == 1 ? above : 0u;
        const uint32_t before = block_excl(h0 + h1, total);
        if (t == 0) { s_bin = 0u; s_above = 0u; s_ns = 0u; s_ne = 0u; }
        __syncthreads();
        if (base_above + before < want && base_above + before + h0 + h1 >= want) {
            const uint32_t acc = base_above + before;
            if (acc + h0 >= want) { s_bin = kSelBins - 1u - 2u * t; s_above = acc; }
            else { s_bin = kSelBins - 2u - 2u * t; s_above = acc + h0; }
        }
        __syncthreads();
        if (level == -1) floor_bin = s_bin;
        else if (level == 0) { prefix_key = s_bin; above = s_above; }
        else { prefix_key = (prefix_key << 11) | s_bin; above = s_above; }
        __syncthreads();
    }
    const uint32_t T22 = prefix_key;
    const uint32_t need_eq = want - above;
    if (thr_out != nullptr && t == 0) thr_out[0] = __uint_as_float(T22 + 1u >= (0x7f800000u >> kshift) ? 0x7f800000u : (T22 + 1u) << kshift);
    SEL1W_WALK({ (void)val; const uint32_t k = m >> kshift;
                 if (k > T22) { const uint32_t p_ = atomicAdd(&s_ns, 1u); if (p_ < nsel) s_list[p_] = i; }
                 else if (k == T22) { const uint32_t p_ = atomicAdd(&s_ne, 1u); if (p_ < kSel1ECap) e_list[p_] = i; } })
    __syncthreads();
    const uint32_t ns = s_ns, ne = s_ne;                       // (ns = `above` < want)
    if (ne <= kSel1ECap) {
        // the need_eq left-most columns of the key join the list; then every entry goes to its rank
        for (uint32_t q = t; q < ne; q += kSel1Threads) {
            const uint32_t me = e_list[q];
            uint32_t rank = 0;
            for (uint32_t e = 0; e < ne; ++e) rank += e_list[e] < me ? 1u : 0u;
            if (rank < need_eq && ns + rank < nsel) s_list[ns + rank] = me;
        }
        __syncthreads();
        const uint32_t tot = ns + need_eq < nsel ? ns + need_eq : nsel;      // = want
        for (uint32_t q = t; q < tot; q += kSel1Threads) {
            const uint32_t me = s_list[q];
            uint32_t rank = 0;
            for (uint32_t e = 0; e < tot; ++e) rank += s_list[e] < me ? 1u : 0u;
            sub[rank] = me;
        }
    } else {
        // ordered walk: 4096 consecutive columns per step, thread t its four
        uint32_t out = 0, eq_run = 0;
        for (uint32_t b0 = 0; b0 < n_pad; b0 += 4u * kSel1Threads) {
            const uint32_t base = b0 + 4u * t;
            const v4f v = base < n_pad ? *reinterpret_cast<const v4f*>(c0 + base) : v4f{ 0.f, 0.f, 0.f, 0.f };
            uint32_t isel = 0, ieq = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t i = base + (uint32_t)e;
                if (i >= n) continue;
                const uint32_t k = mag_bits(v[e]) >> kshift;
                if (k > T22) isel |= 1u << e; else if (k == T22) ieq |= 1u << e;
            }
            uint32_t tot = 0;
            const uint32_t eq_before = eq_run + block_excl((uint32_t)__popc(ieq), tot);
            eq_run += tot;
            uint32_t take = eq_before < need_eq ? need_eq - eq_before : 0u, chosen = isel;
#pragma unroll
            for (int e = 0; e < 4; ++e) if ((ieq >> e) & 1u) { if (take != 0u) { chosen |= 1u << e; --take; } }
            uint32_t pos = out + block_excl((uint32_t)__popc(chosen), tot);
            out += tot;
#pragma unroll
            for (int e = 0; e < 4; ++e) if ((chosen >> e) & 1u) { if (pos < nsel) sub[pos] = base + (uint32_t)e; ++pos; }
        }
    }
#undef SEL1W_WALK
    for (uint32_t p = want + t; p < nsel; p += kSel1Threads) sub[p] = 0xffffffffu;
}

// block_reduce_pair with ONE barrier: the caller gives every use in a round a scratch pair of its own, so no barrier has to
// protect the previous use's readers (the next write to the same pair is a whole round of barriers away).  Same comparisons in
// the same order: the same result.
template <typename T, bool MAX>
__device__ __forceinline__ void block_reduce_pair_1b(T& v, uint32_t& i, T* sv, uint32_t* si)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    wave_reduce_pair<T, MAX>(v, i);
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

// ---- k_sub_solve: the whole path of one signal on its subset -----------------------------------------------------
struct SubLds {
    float* Gc;         // [kSbRows][kSbS]  Gram rows of the positions, restricted to the subset
    float* I;          // [kSbRows][kSbInvPitch] explicit inverse over the positions (dead rows / columns are zero)
    float* xs;         // [kSbRows] x of the positions
    float* ds;         // [kSbRows] direction
    float* u1;         // [kSbRows]
    float* u2;         // [kSbRows]
    float* sg;         // [kSbRows]
    uint32_t* pcol;    // [kSbRows] column of a position
    uint32_t* psub;    // [kSbRows] its index in the subset
    uint32_t* alive;   // [kSbRows] 1 while the position is in the support
    uint32_t* sub;     // [kSbS]
    float* cs;         // [kSbS] c of the subset columns (this round)
    float* qs;         // [kSbS]
};
__host__ __device__ inline size_t sub_lds_bytes()
{
    return ((size_t)kSbRows * kSbS + (size_t)kSbRows * kSbInvPitch + 8 * (size_t)kSbRows + 3 * (size_t)kSbS) * 4;
}


template <bool STAMPS>
__global__ __launch_bounds__(kSbS)
void k_sub_solve(const float* __restrict__ G, uint32_t gpitch, const float* __restrict__ c0_all, uint32_t n, uint32_t n_pad,
                 const uint32_t* __restrict__ sub_all, const uint32_t* __restrict__ first_pick,
                 float tol, uint32_t max_iter, int strict_sign, int zero_on_removal, int tie_guard, int tie_exit,
                 uint32_t* __restrict__ log_hdr, uint32_t* __restrict__ log_pcol, float* __restrict__ log_X, float* __restrict__ log_D,
                 float* __restrict__ x_all, uint32_t* __restrict__ gam2_all, uint32_t* __restrict__ touched2_all, uint32_t kcap,
                 DevState* __restrict__ st_all, TraceEntry* trace, uint32_t trace_cap, int gsub, uint32_t g_slot_stride, unsigned long long* dbg)
{
    // dbg (developer aid, SS_HIP_SUB_STAMPS): cycles of slot 0 by phase, summed over the rounds: [0] chain, [1] max |c|, [2] log + scan,
    // [3] arg-min, [4] hand-shake + x update, [5] u1 / u2, [6] inverse update + signs, [7] direction, [8] rounds
    unsigned long long tph[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long tlast = 0;
#define SUB_STAMP(P) if constexpr (STAMPS) { const unsigned long long now_ = __builtin_readcyclecounter(); tph[P] += now_ - tlast; tlast = now_; }
    // gsub != 0 (screened form of one signal, screen.hip): G is the subset's own Gram matrix Gs[kSbS][gpitch], rows and
    // columns by subset index, instead of the full G = A^T A
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float sv[16];
    __shared__ uint32_t si[16];
    __shared__ float sv2[16];
    __shared__ uint32_t si2[16];
    __shared__ uint32_t s_u[4];
    __shared__ float s_f[2];
    SubLds L;
    {
        float* p = reinterpret_cast<float*>(smem);
        L.Gc = p; p += (size_t)kSbRows * kSbS;
        L.I = p; p += (size_t)kSbRows * kSbInvPitch;
        L.xs = p; p += kSbRows; L.ds = p; p += kSbRows; L.u1 = p; p += kSbRows; L.u2 = p; p += kSbRows; L.sg = p; p += kSbRows;
        L.pcol = reinterpret_cast<uint32_t*>(p); p += kSbRows;
        L.psub = reinterpret_cast<uint32_t*>(p); p += kSbRows;
        L.alive = reinterpret_cast<uint32_t*>(p); p += kSbRows;
        L.sub = reinterpret_cast<uint32_t*>(p); p += kSbS;
        L.cs = p; p += kSbS; L.qs = p;
    }
    const uint32_t slot = blockIdx.x, j = threadIdx.x;
    if (gsub) G += (size_t)slot * g_slot_stride;               // (a batch in the screened form: every slot its own subset Gram matrix)
    const float* c0 = c0_all + (size_t)slot * n_pad;
    float* x_out = x_all + (size_t)slot * n_pad;
    uint32_t* gam_out = gam2_all + (size_t)slot * 2 * kcap;
    uint32_t* tch_out = touched2_all + (size_t)slot * 2 * kcap;
    DevState* st = st_all + slot;
    uint32_t* hdr = log_hdr + (size_t)slot * kSbLog * 8;
    float* LX = log_X + (size_t)slot * kSbLog * kSbRows;
    float* LD = log_D + (size_t)slot * kSbLog * kSbRows;
    if (slot != 0) trace = nullptr;

    const uint32_t mycol = sub_all[(size_t)slot * kSbS + j];
    const bool valid = mycol < n;
    L.sub[j] = mycol;
    const float c0v = valid ? c0[mycol] : 0.f;
    for (uint32_t e = j; e < kSbRows * kSbInvPitch; e += kSbS) L.I[e] = 0.f;
    for (uint32_t p = 0; p < kSbRows; ++p) L.Gc[(size_t)p * kSbS + j] = 0.f;      // (rows not yet given out are read with zero coefficients)
    if (j < kSbRows) { L.xs[j] = 0.f; L.ds[j] = 0.f; L.u1[j] = 0.f; L.u2[j] = 0.f; L.sg[j] = 0.f; L.pcol[j] = 0xffffffffu; L.psub[j] = 0u; L.alive[j] = 0u; }
    uint32_t idx0 = first_pick[slot];
    if (gsub == 2) {
        // (screen.hip's half-precision first pass ranked the columns by an APPROXIMATE A^T y; c0 of the subset's columns is exact:
        // the first pick is the left-most largest of those — the screening pass vouches for every column outside)
        float bv = valid ? fabsf(c0v) : -1.f;
        uint32_t bi = valid ? mycol : 0xffffffffu;
        block_reduce_pair<float, true>(bv, bi, sv, si);
        idx0 = bi;
    }
    if (j == 0) s_u[0] = 0xffffffffu;
    __syncthreads();
    if (valid && mycol == idx0) s_u[0] = j;
    __syncthreads();
    int32_t mypos = -1;                       // position of my column while it is in the support
    uint32_t P = 0, K = 0;                    // positions given out, support size
    uint32_t status = 0, iter = 0, nlog = 0, reason = 0;
    float c_inf = 0.f, lambda_prev = 0.f, gamma_prev = 0.f, lambda0 = 0.f;
    uint32_t just_removed = 0xffffffffu;
    bool tie_any = false;

    auto gather_row = [&](uint32_t p, uint32_t col, uint32_t sidx) {       // Gc[p][.] = G[col][sub[.]]  (sidx: col's index in the subset)
        L.Gc[(size_t)p * kSbS + j] = valid ? (gsub ? G[(size_t)sidx * gpitch + j] : G[(size_t)col * gpitch + mycol]) : 0.f;
    };
    // acc = sum_b M[b] * v[b] over the positions, the chain in position order; eight operands are read ahead of their fmas
    // (a plain loop waits an LDS round trip per term).  Entries of I, sg, u1, u2 at positions >= P are zero: whole groups of 8.
    auto row_dot = [&](const float* Mrow, const float* vec) -> float {
        float acc = 0.f;
        for (uint32_t b0 = 0; b0 < P; b0 += 8u) {
            float mv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) mv[e] = Mrow[b0 + (uint32_t)e];
            const v4f v0 = *reinterpret_cast<const v4f*>(vec + b0), v1 = *reinterpret_cast<const v4f*>(vec + b0 + 4u);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_fmaf(mv[e], v0[e], acc);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_fmaf(mv[4 + e], v1[e], acc);
        }
        return acc;
    };
    auto direction = [&]() {                                 // ds = I * sg over the positions, one thread per row
        if (j < P) {
            const float acc = row_dot(&L.I[j * kSbInvPitch], L.sg);
            L.ds[j] = L.alive[j] ? acc : 0.f;
        }
    };

    if (s_u[0] == 0xffffffffu) {
        status = kStatusSubsetDecline; reason |= kReasonNoCand;        // (cannot happen: the largest |c0| is in the subset)
    } else {
        // first pick (homotopy-cpu.cpp:217-229): inv = [1 / ||a||^2] through the norm, direction = inv * sign
        const uint32_t sp0 = s_u[0];
        gather_row(0, idx0, sp0);
        __syncthreads();
        if (j == 0) {
            const float dot = L.Gc[sp0];
            const float nrm = sqrtf(dot);
            const float inv00 = 1.f / (nrm * nrm);
            const float cc = c0[idx0];
            const float seed = strict_sign ? cc : fabsf(cc);
            L.I[0] = inv00;
            L.sg[0] = sign_tol(seed, tol);
            L.pcol[0] = idx0; L.psub[0] = sp0; L.alive[0] = 1u; L.xs[0] = 0.f;
            if (trace != nullptr) { trace[0].idx = idx0; trace[0].added = 1; trace[0].gamma = 0.0; trace[0].c_inf = (double)fabsf(cc); }
        }
        P = 1; K = 1;
        if (j == sp0) mypos = 0;
        __syncthreads();
        direction();
        __syncthreads();
        lambda0 = fabsf(c0[idx0]);
        lambda_prev = lambda0;                 // (k_init leaves c_inf = lambda0, gamma = 0: the first round's lambda is "where the last step left it")
        // (screened form: the tolerance guard of the Gram forms, k_la_init_pick's — the caller's usual engine decides what to do)
        if (gsub && !((double)tol >= kGramGuard * (double)lambda0)) { status = kStatusSubsetDecline; reason |= kReasonGuard; }

        for (uint32_t round = 1; status == 0u; ++round) {
            // (the round's hand-shake words — whose column was picked, a tie seen — are cleared here: the reductions' barriers lie
            // between this and their writers, the previous round's readers are many barriers back)
            if (j == 0) { s_u[1] = 0xffffffffu; s_u[2] = 0u; s_u[3] = 0u; }
            if constexpr (STAMPS) tlast = __builtin_readcyclecounter();
            // ---- c, q of my column: the chain over the positions -----------------------------------------------
            float cv = c0v, qv = 0.f;
            for (uint32_t p = 0; p < P; p += 4) {                     // (whole groups of 4: positions >= P carry x = d = 0 and zero rows)
                const v4f x4 = *reinterpret_cast<const v4f*>(&L.xs[p]), d4 = *reinterpret_cast<const v4f*>(&L.ds[p]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = L.Gc[(size_t)(p + e) * kSbS + j];
                    cv = __builtin_fmaf(-x4[e], g, cv);
                    qv = __builtin_fmaf(d4[e], g, qv);
                }
            }
            L.cs[j] = cv; L.qs[j] = qv;
            SUB_STAMP(0)
            float mx = valid ? fabsf(cv) : -1.f;
            uint32_t mxi = mycol;
            block_reduce_pair_1b<float, true>(mx, mxi, sv, si);
            c_inf = mx;
            SUB_STAMP(1)
            // ---- loop control (homotopy-cpu.cpp:236, 272) ---------------------------------------------------------
            const bool stop = (round > 1 && !(c_inf > tol)) || round > max_iter;
            if (nlog >= kSbLog) { status = kStatusSubsetDecline; reason |= kReasonLog; break; }
            if (j == 0) {
                uint32_t* h = hdr + nlog * 8;
                h[0] = P; h[1] = stop ? 0u : 1u; h[2] = 0xffffffffu; h[3] = 0u; h[4] = __float_as_uint(c_inf); h[5] = 0u;
                h[6] = just_removed; h[7] = tie_band<float>(c_inf, lambda_prev, gamma_prev, lambda0) ? 1u : 0u;
            }
            if (j < kSbRows) { LX[(size_t)nlog * kSbRows + j] = j < P ? L.xs[j] : 0.f; LD[(size_t)nlog * kSbRows + j] = j < P ? L.ds[j] : 0.f; }
            if (stop) { iter = round - 1; ++nlog; break; }
            // ---- find_max_gamma's scan (homotopy-cpu.cpp:122-163) over the subset ---------------------------------------
            const bool in_band = tie_band<float>(c_inf, lambda_prev, gamma_prev, lambda0);
            float m = Lim<float>::max();
            bool tie = false;
            if (valid) {
                if (mypos >= 0) {
                    const float t = -L.xs[mypos] / L.ds[mypos];
                    if (t > 0.f && t < m) m = t;
                } else {
                    const float dl = 1.f - qv, dr = 1.f + qv;
                    if (dl != 0.f) {
                        float t = (c_inf - cv) / dl;
                        if (tie_guard && t == 0.f && dl > 0.f) t = Lim<float>::tiny();
                        if (t == 0.f && mycol != just_removed && in_band) tie = true;
                        if (t > 0.f && t < m) m = t;
                    }
                    if (dr != 0.f) {
                        float t = (c_inf + cv) / dr;
                        if (tie_guard && t == 0.f && dr > 0.f) t = Lim<float>::tiny();
                        if (t == 0.f && mycol != just_removed && in_band) tie = true;
                        if (t > 0.f && t < m) m = t;
                    }
                }
            }
            SUB_STAMP(2)
            if (tie) s_u[3] = 1u;                                    // (read behind the reduction's barrier)
            float g = m;
            uint32_t idx = valid ? mycol : 0xffffffffu;
            block_reduce_pair_1b<float, false>(g, idx, sv2, si2);
            tie_any = s_u[3] != 0u || tie_any;
            SUB_STAMP(3)
            if (tie_any && tie_exit) { status = kStatusTieRerun; reason |= kReasonTie; iter = round - 1; ++nlog; break; }
            if (!(g < Lim<float>::max())) { status = kStatusSubsetDecline; reason |= kReasonNoCand; break; }     // (no positive candidate: the reference toggles column 0)
            // whose column is it, and is it in the support?
            if (valid && mycol == idx) { s_u[1] = j; s_u[2] = mypos >= 0 ? 1u + (uint32_t)mypos : 0u; }
            __syncthreads();
            const uint32_t spi = s_u[1];
            const bool added = s_u[2] == 0u;
            const uint32_t rpos = added ? P : s_u[2] - 1u;
            // (the entering column's Gram value of my column: the round's only trip to memory, on its way under the log, the x update and
            // their barriers; it lands in the position's row further down)
            float gpre = 0.f;
            if (added && P < kSbRows && valid) gpre = gsub ? G[(size_t)spi * gpitch + j] : G[(size_t)idx * gpitch + mycol];
            if (j == 0) { hdr[nlog * 8 + 2] = idx; hdr[nlog * 8 + 3] = added ? 1u : 0u; hdr[nlog * 8 + 5] = __float_as_uint(g); }
            ++nlog;
            if (trace != nullptr && j == 0 && round < trace_cap) {
                trace[round].idx = idx; trace[round].added = added ? 1u : 0u; trace[round].gamma = (double)g; trace[round].c_inf = (double)c_inf;
            }
            const uint32_t K_new = added ? K + 1u : K - 1u;
            if (K_new == 0u || K_new > kcap || (added && P >= kSbRows)) { status = kStatusSubsetDecline; reason |= K_new == 0u ? kReasonRemoval : kReasonPositions; break; }
            // ---- x += gamma d over the old support (homotopy-cpu.cpp:252) -----------------------------------------------
            if (j < P && L.alive[j]) {
                const float xn = L.xs[j] + g * L.ds[j];
                L.xs[j] = xn;
            }
            if (added) {
                // the column enters at position P: its Gram row and its bookkeeping share the x update's barrier
                L.Gc[(size_t)P * kSbS + j] = gpre;
                if (j == 0) { L.pcol[P] = idx; L.psub[P] = spi; L.xs[P] = 0.f; }
            }
            __syncthreads();
            SUB_STAMP(4)
            if (!added) {
                // the column leaves: in reference mode it keeps its rounding residue (and stays in c through it): not this form's path
                const float res = L.xs[rpos];
                if (res != 0.f && !zero_on_removal) { status = kStatusSubsetDecline; reason |= kReasonRemoval; break; }
                __syncthreads();
                // deflate (online_inverse.h:275-290): u3 = -I[.][r] / I[r][r];  I' = I + (-I[r][r] u3) u3^T; row / column r zero
                const float dd = L.I[rpos * kSbInvPitch + rpos];
                if (j < P) L.u2[j] = L.I[j * kSbInvPitch + rpos] * (-(1.f / dd));
                if (j == 0) { L.xs[rpos] = 0.f; L.alive[rpos] = 0u; }
                __syncthreads();
                const uint32_t qa = kSbS / P, rb = kSbS - qa * P;
                uint32_t a = j / P, b = j - a * P;
                for (uint32_t e = j; e < P * P; e += kSbS, a += qa, b += rb) {
                    if (b >= P) { b -= P; ++a; }
                    const float v = (a == rpos || b == rpos) ? 0.f : L.I[a * kSbInvPitch + b] + (-dd * L.u2[a]) * L.u2[b];
                    L.I[a * kSbInvPitch + b] = v;
                }
                if ((int32_t)rpos == mypos) mypos = -1;
                just_removed = idx;
                K = K_new;
                // sign(c_Gamma) after the step (below): nothing reads sg before the barrier that follows
                if (j < P) {
                    const uint32_t sp = L.psub[j];
                    const float cn = L.cs[sp] - g * L.qs[sp];
                    L.sg[j] = (L.alive[j] && j != rpos) ? sign_tol(cn, tol) : 0.f;
                }
            } else {
                // u1 = G[idx][support], the bordered inverse (online_inverse.h:209-248)
                if (j < P) L.u1[j] = L.alive[j] ? L.Gc[(size_t)P * kSbS + L.psub[j]] : 0.f;
                __syncthreads();
                if (j < P) L.u2[j] = row_dot(&L.I[j * kSbInvPitch], L.u1);
                __syncthreads();
                SUB_STAMP(5)
                // (every thread walks the same chain u1 . u2 — the value thread 0 alone used to publish through another barrier)
                const float dv = 1.f / (L.Gc[(size_t)P * kSbS + spi] - row_dot(L.u1, L.u2));
                const uint32_t Pn = P + 1u;
                // (element e = a * Pn + b for e = j, j + 448, ...: the row / column advance without a division per element)
                const uint32_t qa = kSbS / Pn, rb = kSbS - qa * Pn;
                uint32_t a = j / Pn, b = j - a * Pn;
                for (uint32_t e = j; e < Pn * Pn; e += kSbS, a += qa, b += rb) {
                    if (b >= Pn) { b -= Pn; ++a; }
                    float v;
                    if (a == P && b == P) v = dv;
                    else if (a == P) v = -dv * L.u2[b];
                    else if (b == P) v = -dv * L.u2[a];
                    else v = L.I[a * kSbInvPitch + b] + (dv * L.u2[a]) * L.u2[b];
                    L.I[a * kSbInvPitch + b] = v;
                }
                if (j == 0) L.alive[P] = 1u;
                if (j == spi) mypos = (int32_t)P;
                // ---- sign(c_Gamma) of the correlations after the step (c - gamma q), dead zone tol (homotopy-cpu.cpp:257-267); the new
                // position is alive (its flag is being written by thread 0 right now)
                if (j < Pn) {
                    const uint32_t sp = L.psub[j];
                    const float cn = L.cs[sp] - g * L.qs[sp];
                    L.sg[j] = (j == P || L.alive[j]) ? sign_tol(cn, tol) : 0.f;
                }
                P = Pn;
                K = K_new;
                just_removed = 0xffffffffu;
            }
            __syncthreads();
            SUB_STAMP(6)
            // ---- the direction from those signs
            direction();
            __syncthreads();
            SUB_STAMP(7)
            if constexpr (STAMPS) tph[8] += 1;
            iter = round;
            lambda_prev = c_inf;
            gamma_prev = g;
        }
    }
    __syncthreads();
    // ---- hand-over: dense x, sorted support / touched lists, the state -----------------------------------------------
    if (j < kSbRows) log_pcol[(size_t)slot * kSbRows + j] = j < P ? L.pcol[j] : 0xffffffffu;
    if (status == 0u || status == kStatusTieRerun) {
        // (a column that re-entered holds two positions: the later one counts, the earlier is dead with x = 0)
        if (j < P) {
            bool last = true;
            for (uint32_t e = j + 1; e < P; ++e) last = last && L.pcol[e] != L.pcol[j];
            L.u1[j] = last ? 1.f : 0.f;
        }
        __syncthreads();
        if (j < P && L.u1[j] != 0.f) {
            const uint32_t col = L.pcol[j];
            uint32_t rs = 0, rt = 0;
            for (uint32_t b = 0; b < P; ++b) {
                if (L.u1[b] != 0.f && L.pcol[b] < col) { ++rt; if (L.alive[b]) ++rs; }
            }
            tch_out[rt] = col;
            if (L.alive[j]) gam_out[rs] = col;
            x_out[col] = L.xs[j];
        }
        if (j == 0) {
            uint32_t nt = 0;
            for (uint32_t a = 0; a < P; ++a) nt += L.u1[a] != 0.f ? 1u : 0u;
            st->ntouched = nt;
        }
    }
    if (j == 0) {
        st->done = 1;
        st->status = status;
        st->iter = iter;
        st->K = K;
        st->cur = 0;
        st->c_inf = (double)c_inf;
        st->gamma = (double)gamma_prev;
        st->tie_stall = tie_any ? 1u : 0u;
        st->lambda0 = lambda0;
        st->solo_nlog = nlog;                  // breakpoints logged (k_sub_verify)
        st->need_sweep = 0;                    // (k_sub_verify raises it when a breakpoint does not hold)
        st->done_round = iter + 1u;
        st->sub_reason = reason;
    }
    if constexpr (STAMPS) {
        if (dbg != nullptr && slot == 0u && j == 0u)
            for (int q = 0; q < 9; ++q) dbg[q] = tph[q];
    }
#undef SUB_STAMP
}

// ---- k_sub_verify: every column outside the subset against every logged breakpoint -----------------------------
// One workgroup = 512 columns of one slot, a thread TWO of them (j and j + 256): their Gram values g[p] = G[col_p][j] sit in
// registers (2 x 72), the coefficient tables of the log are staged in LDS once per workgroup and every 16-byte read of
// them feeds 16 fmas (4 positions x {c, q} x 2 columns).  (Measured on the way: coefficients through the scalar cache —
// the 37-KiB tables of a slot thrash its 16 KiB — 150-160 ms per 4096 signals at 8192 x 65536; one column per thread
// from LDS is bound by the LDS pipe, every read serving 8 fmas.)  The chain per (column, breakpoint) is k_sub_solve's,
// term for term.  A candidate is only evaluated exactly (IEEE division, the reference's predicates) when a conservative
// bound cannot rule it out.
constexpr int kVsThreads = 256;
constexpr uint32_t kVsCols = 2 * kVsThreads;     // columns a workgroup checks
constexpr uint32_t kVsPitch = kSbRows;           // floats per breakpoint row in LDS
__host__ __device__ inline size_t sub_verify_lds_bytes() { return (2 * (size_t)kSbLog * kVsPitch + (size_t)kSbLog * 8 + kSbRows) * 4; }

// (sub_check — the reference's predicates for one (column, breakpoint) — lives in subcheck.h: screen.hip's exact re-check uses it too)
__global__ __launch_bounds__(kVsThreads, 2)
void k_sub_verify(const float* __restrict__ G, uint32_t gpitch, const float* __restrict__ c0_all, uint32_t n, uint32_t n_pad,
                  const uint32_t* __restrict__ sub_all, const uint32_t* __restrict__ log_hdr, const uint32_t* __restrict__ log_pcol,
                  const float* __restrict__ log_X, const float* __restrict__ log_D, int tie_guard, float tol, DevState* __restrict__ st_all)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_in[kVsCols / 32u];
    float* sX = reinterpret_cast<float*>(smem);                       // [kSbLog][kVsPitch]
    float* sD = sX + (size_t)kSbLog * kVsPitch;
    uint32_t* sH = reinterpret_cast<uint32_t*>(sD + (size_t)kSbLog * kVsPitch);   // [kSbLog][8]
    uint32_t* sP = sH + (size_t)kSbLog * 8;                          // [kSbRows] columns of the positions
    const uint32_t slot = blockIdx.y;
    DevState* st = st_all + slot;
    if (st->status != 0u) return;
    const uint32_t nlog = st->solo_nlog;
    if (nlog == 0u) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * kVsCols;
    {
        const uint32_t* hdr = log_hdr + (size_t)slot * kSbLog * 8;
        const v4f* LX4 = reinterpret_cast<const v4f*>(log_X + (size_t)slot * kSbLog * kSbRows);
        const v4f* LD4 = reinterpret_cast<const v4f*>(log_D + (size_t)slot * kSbLog * kSbRows);
        const uint32_t nv = nlog * kSbRows / 4u;                      // (16-byte pieces: rows are 288 bytes)
        for (uint32_t e = tid; e < nv; e += kVsThreads) { reinterpret_cast<v4f*>(sX)[e] = LX4[e]; reinterpret_cast<v4f*>(sD)[e] = LD4[e]; }
        for (uint32_t e = tid; e < nlog * 8; e += kVsThreads) sH[e] = hdr[e];
        if (tid < kSbRows) sP[tid] = log_pcol[(size_t)slot * kSbRows + tid];
        if (tid < kVsCols / 32u) s_in[tid] = 0u;
    }
    __syncthreads();
    // which of this workgroup's columns are in the subset (k_sub_solve dealt with those)
    {
        const uint32_t* sub = sub_all + (size_t)slot * kSbS;
        for (uint32_t e = tid; e < kSbS; e += kVsThreads) {
            const uint32_t c = sub[e];
            if (c >= base && c < base + kVsCols) atomicOr(&s_in[(c - base) >> 5], 1u << ((c - base) & 31u));
        }
    }
    __syncthreads();
    const uint32_t Pfin = sH[(nlog - 1u) * 8];
    const uint32_t j0 = base + tid, j1 = j0 + kVsThreads;
    const bool mine0 = j0 < n && !((s_in[tid >> 5] >> (tid & 31u)) & 1u);
    const bool mine1 = j1 < n && !((s_in[(tid + kVsThreads) >> 5] >> (tid & 31u)) & 1u);
    const size_t jc0 = j0 < n ? j0 : 0u, jc1 = j1 < n ? j1 : 0u;
    float g0[kSbRows], g1[kSbRows];
#pragma unroll
    for (uint32_t p = 0; p < kSbRows; ++p) {
        const float* row = G + (size_t)sP[p < Pfin ? p : 0u] * gpitch;
        g0[p] = p < Pfin ? row[jc0] : 0.f;
        g1[p] = p < Pfin ? row[jc1] : 0.f;
    }
    const float* c0 = c0_all + (size_t)slot * n_pad;
    const float c00 = mine0 ? c0[j0] : 0.f, c01 = mine1 ? c0[j1] : 0.f;
    bool fail = false, tie = false;
    for (uint32_t k = 0; k < nlog; ++k) {
        const uint32_t Pk = sH[k * 8];
        const float* xk = sX + (size_t)k * kVsPitch;
        const float* dk = sD + (size_t)k * kVsPitch;
        float cv0 = c00, qv0 = 0.f, cv1 = c01, qv1 = 0.f;
#pragma unroll
        for (uint32_t p4 = 0; p4 < kSbRows; p4 += 4) {
            if (p4 < Pk) {                                           // (uniform; positions >= Pk carry zero coefficients)
                const v4f x4 = *reinterpret_cast<const v4f*>(xk + p4), d4 = *reinterpret_cast<const v4f*>(dk + p4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cv0 = __builtin_fmaf(-x4[e], g0[p4 + e], cv0);
                    qv0 = __builtin_fmaf(d4[e], g0[p4 + e], qv0);
                    cv1 = __builtin_fmaf(-x4[e], g1[p4 + e], cv1);
                    qv1 = __builtin_fmaf(d4[e], g1[p4 + e], qv1);
                }
            }
        }
        if (mine0) sub_check(cv0, qv0, j0, k, nlog, sH, tol, tie_guard, st, fail, tie);
        if (mine1) sub_check(cv1, qv1, j1, k, nlog, sH, tol, tie_guard, st, fail, tie);
    }
    if (fail) {
        __hip_atomic_store(&st->need_sweep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read by k_sub_finish)
        if (!(__hip_atomic_load(&st->sub_reason, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kReasonColumn)) atomicOr(&st->sub_reason, kReasonColumn);
    }
    if (tie) __hip_atomic_store(&st->tie_stall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// verdicts of k_sub_verify into the status word (its workgroups read `status` while they run)
__global__ void k_sub_finish(DevState* __restrict__ st_all, uint32_t nslots)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    DevState* st = st_all + s;
    if (st->status == 0u && st->need_sweep != 0u) st->status = kStatusSubsetFail;
    st->need_sweep = 0u;
}

// ---- host side ---------------------------------------------------------------------------------------------------
bool sub_form_usable(ss_hip_ctx* ctx)
{
    if (ctx->sub_attr_set < 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sub_solve<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)sub_lds_bytes());
        if (e != hipSuccess) (void)hipGetLastError();
        ctx->sub_attr_set = e == hipSuccess ? 1 : 0;
    }
    return ctx->sub_attr_set == 1;
}

size_t sub_buffer_bytes(uint32_t nslots)
{
    // sub, first pick + value, log header, position columns, X, D
    return (size_t)nslots * ((size_t)kSbS * 4 + 8 + (size_t)kSbLog * 8 * 4 + (size_t)kSbRows * 4 + 2 * (size_t)kSbLog * kSbRows * 4);
}

SubBufs sub_bufs(ss_hip_ctx* ctx, uint32_t nslots)
{
    SubBufs B;
    unsigned char* b = static_cast<unsigned char*>(ctx->sub_buf);
    B.sub = reinterpret_cast<uint32_t*>(b); b += (size_t)nslots * kSbS * 4;
    B.fpick = reinterpret_cast<uint32_t*>(b); b += (size_t)nslots * 4;
    B.fval = reinterpret_cast<float*>(b); b += (size_t)nslots * 4;
    B.hdr = reinterpret_cast<uint32_t*>(b); b += (size_t)nslots * kSbLog * 8 * 4;
    B.pcol = reinterpret_cast<uint32_t*>(b); b += (size_t)nslots * kSbRows * 4;
    B.LX = reinterpret_cast<float*>(b); b += (size_t)nslots * kSbLog * kSbRows * 4;
    B.LD = reinterpret_cast<float*>(b);
    return B;
}

hipError_t launch_sub_select(ss_hip_ctx* ctx, const SubBufs& B, uint32_t nslots, const float* c0, float* thr_out, const float* wmax, uint32_t nwmax)
{
    // (thr_out, one slot only: a value every column left out stays below; wmax: see k_sub_select1)
    if (nslots == 1 && ctx->n_pad <= 4u * kSel1J * kSel1Threads)        // (one slot: the walking form, for latency)
        hipLaunchKernelGGL(k_sub_select1, dim3(1), dim3(kSel1Threads), 0, ctx->stream, c0, (uint32_t)ctx->n, ctx->n_pad, B.sub, B.fpick, B.fval, thr_out,
                           wmax, nwmax);
    else if (nslots == 1)                                               // (... in chunks of 65536 columns)
        hipLaunchKernelGGL(k_sub_select1w, dim3(1), dim3(kSel1Threads), 0, ctx->stream, c0, (uint32_t)ctx->n, ctx->n_pad, B.sub, B.fpick, B.fval, kSbS, thr_out,
                           wmax, nwmax);
    else
        hipLaunchKernelGGL(k_sub_select, dim3(nslots), dim3(kSelThreads), 0, ctx->stream, c0, (uint32_t)ctx->n, ctx->n_pad, B.sub, B.fpick, B.fval, kSbS);
    return hipGetLastError();
}

// the nsel columns with the largest |v| of ONE vector, ascending (screen.hip's fp64 form: v = float(|A^T y|))
hipError_t launch_select_top(ss_hip_ctx* ctx, const float* v, uint32_t n, uint32_t n_pad, uint32_t nsel, uint32_t* sub, uint32_t* fpick, float* fval,
                             float* thr_out, const float* wmax, uint32_t nwmax)
{
    // (thr_out: a value every entry left out stays below; only the one-workgroup kernel reports it — nsel <= kSel1SCap; wmax: see k_sub_select1)
    if (nsel <= kSel1SCap) hipLaunchKernelGGL(k_sub_select1w, dim3(1), dim3(kSel1Threads), 0, ctx->stream, v, n, n_pad, sub, fpick, fval, nsel, thr_out,
                                              wmax, nwmax);
    else if (thr_out != nullptr) return hipErrorInvalidConfiguration;
    else hipLaunchKernelGGL(k_sub_select, dim3(1), dim3(kSelThreads), 0, ctx->stream, v, n, n_pad, sub, fpick, fval, nsel);
    return hipGetLastError();
}

// G, gpitch: the full Gram matrix (gsub = 0) or the subset's own (gsub = 1, one slot: screen.hip)
hipError_t launch_sub_solve(ss_hip_ctx* ctx, Workspace<float>& ws, const SubBufs& B, uint32_t nslots, const float* G, uint32_t gpitch, int gsub,
                            const float* c0, float tol, uint32_t max_iter, uint32_t g_slot_stride)
{
    if (ctx->sub_dbg != nullptr) {
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sub_solve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sub_lds_bytes()) == hipSuccess;
        if (ok)
            hipLaunchKernelGGL(k_sub_solve<true>, dim3(nslots), dim3(kSbS), sub_lds_bytes(), ctx->stream, G, gpitch, c0, (uint32_t)ctx->n,
                               ctx->n_pad, (const uint32_t*)B.sub, (const uint32_t*)B.fpick, tol, max_iter, ctx->strict_sign, ctx->zero_on_removal,
                               ctx->tie_guard, (ctx->tie_rerun && !ctx->tie_guard) ? 1 : 0, B.hdr, B.pcol, B.LX, B.LD, ws.x, ws.gam, ws.touched,
                               ws.dims.kcap, ws.st, ws.trace, ws.trace_cap, gsub, g_slot_stride, static_cast<unsigned long long*>(ctx->sub_dbg));
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_sub_solve<false>, dim3(nslots), dim3(kSbS), sub_lds_bytes(), ctx->stream, G, gpitch, c0, (uint32_t)ctx->n,
                       ctx->n_pad, (const uint32_t*)B.sub, (const uint32_t*)B.fpick, tol, max_iter, ctx->strict_sign, ctx->zero_on_removal,
                       ctx->tie_guard, (ctx->tie_rerun && !ctx->tie_guard) ? 1 : 0, B.hdr, B.pcol, B.LX, B.LD, ws.x, ws.gam, ws.touched,
                       ws.dims.kcap, ws.st, ws.trace, ws.trace_cap, gsub, g_slot_stride, static_cast<unsigned long long*>(ctx->sub_dbg));
    return hipGetLastError();
}

hipError_t launch_sub_finish(ss_hip_ctx* ctx, Workspace<float>& ws, uint32_t nslots)
{
    hipLaunchKernelGGL(k_sub_finish, dim3((nslots + 255) / 256), dim3(256), 0, ctx->stream, ws.st, nslots);
    return hipGetLastError();
}

hipError_t launch_sub_finish_st(ss_hip_ctx* ctx, DevState* st, uint32_t nslots)
{
    hipLaunchKernelGGL(k_sub_finish, dim3((nslots + 255) / 256), dim3(256), 0, ctx->stream, st, nslots);
    return hipGetLastError();
}

hipError_t launch_sub_form(ss_hip_ctx* ctx, Workspace<float>& ws, uint32_t nslots, const float* c0, float tol, uint32_t max_iter,
                           hipEvent_t e0, hipEvent_t e1, hipEvent_t e2)
{
    if (!sub_form_usable(ctx) || ctx->sub_buf == nullptr || ctx->gram_full == nullptr) return hipErrorInvalidConfiguration;
    const SubBufs B = sub_bufs(ctx, nslots);
    const uint32_t n = (uint32_t)ctx->n;
    hipStream_t s = ctx->stream;
    if (e0) (void)hipEventRecord(e0, s);
    (void)launch_sub_select(ctx, B, nslots, c0);
    if (e0 && ctx->ev_sub_sel) (void)hipEventRecord(ctx->ev_sub_sel, s);
    (void)launch_sub_solve(ctx, ws, B, nslots, (const float*)ctx->gram_full, ctx->gram_pitch, 0, c0, tol, max_iter);
    if (e1) (void)hipEventRecord(e1, s);
    hipLaunchKernelGGL(k_sub_verify, dim3((n + kVsCols - 1) / kVsCols, nslots), dim3(kVsThreads), sub_verify_lds_bytes(), s,
                       (const float*)ctx->gram_full, ctx->gram_pitch, c0, n, ctx->n_pad, (const uint32_t*)B.sub, (const uint32_t*)B.hdr,
                       (const uint32_t*)B.pcol, (const float*)B.LX, (const float*)B.LD, ctx->tie_guard, tol, ws.st);
    (void)launch_sub_finish(ctx, ws, nslots);
    if (e2) (void)hipEventRecord(e2, s);
    return hipGetLastError();
}

}  // namespace sship
