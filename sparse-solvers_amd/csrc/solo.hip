// solo.hip — verification of the speculative ("solo") form of the lookahead engine.
//
// A solo launch (k_la_persist<true>, persist.hip) runs Homotopy iterations in ONE workgroup on a subset
// of 256 columns — the support, the cached columns and the best-ranked entrant candidates — with no
// exchange between workgroups: lambda = ||c||_inf and the step length (find_max_gamma,
// /root/reference/src/solvers/homotopy-cpu.cpp:100-164) are taken over the subset alone.  That is the
// reference's iteration only if no column outside the subset would have changed either of them.  The
// launch therefore logs every breakpoint (support lists with x_S and d_S, lambda, the pick), and
//
//   k_la_verify    recomputes, for ALL columns and all logged breakpoints at once, c = c0 - sum_j x_j g_j
//                  and q = sum_j d_j g_j from the same Gram rows in the same order of operations (sorted
//                  support, separately rounded products and sums), then max |c| and the step-length scan
//                  with the reference's predicates; one pass over the Gram rows serves up to 32
//                  breakpoints (64 accumulators per column);
//   k_la_vpublish  reduces the per-workgroup partials and compares them, bit for bit, with what the solo
//                  launch used.  What a solo launch leaves behind is only STAGED (persist.hip): if every
//                  breakpoint agrees, the staged state becomes the state of the solve and the outcome
//                  (finished / Gram column missing / tier outgrown) is released to the host.  If breakpoint
//                  k* disagrees nothing is committed; the next solo launch repeats exactly the verified
//                  iterations before k* (same subset, same arithmetic, hence the same state), commits, and
//                  the resident form goes on from there — no decision taken from a subset reaches the
//                  caller unchecked.
//
// The solo kernel itself stops before the last step of a path (lambda - gamma <= tolerance): there every
// column's candidate ties within rounding and no subset can know the winner.
//
// k_la_verify also leaves what the next steps rank by: tcand (k_la_top's input) and the two best entrant
// candidates of each workgroup's 256 columns (the next subset); k_la_cand_init seeds the latter from |c0|.
// Compiled with -ffp-contract=off like persist.hip (the arithmetic has to match it bit for bit).
#include "ss_hip_internal.h"
#include "ss_hip_device.h"

#include <algorithm>

namespace sship {

constexpr int kVfThreads = 256;
constexpr uint32_t kVfUnion = 144;           // support columns a chunk of breakpoints can involve (96 + 40, rounded up)
constexpr uint32_t kVfRedPitch = kSoloWidth + 16;    // row pitch of the reduction buffer (16 readers per row: conflict-free)

// the kCandPerBlock best (smallest key, then smallest column) of the block's offers -> out[0 .. kCandPerBlock)
// (two per 256-column block were too few: with 64 support columns spread over 256 blocks four solves in ten
// have a block that holds three of them, and the third was then never ranked)
__device__ __forceinline__ void block_topn(float key, uint32_t colv, uint64_t* out, float* sv, uint32_t* si)
{
#pragma unroll
    for (uint32_t r = 0; r < kCandPerBlock; ++r) {
        float k1 = key;
        uint32_t c1 = colv;
        block_reduce_pair<float, false>(k1, c1, sv, si);
        if (threadIdx.x == 0) out[r] = c1 != 0xffffffffu ? (((uint64_t)ordered_key(k1) << 32) | c1) : ~0ull;
        if (colv == c1) { key = Lim<float>::max(); colv = 0xffffffffu; }     // taken
        __syncthreads();
    }
}

// entrant candidates before the first scan: the largest |c0| of every 256-column block
__global__ __launch_bounds__(kVfThreads)
void k_la_cand_init(const float* __restrict__ c0, uint32_t n, const DevState* __restrict__ st, uint64_t* __restrict__ cand_top)
{
    __shared__ float sv[16];
    __shared__ uint32_t si[16];
    if (st->done) return;
    const uint32_t i = blockIdx.x * kVfThreads + threadIdx.x;
    float key = Lim<float>::max();
    uint32_t cl = 0xffffffffu;
    if (i < n) {
        const float v = c0[i];
        key = v < 0.f ? v : -v;
        cl = i;
    }
    block_topn(key, cl, cand_top + kCandPerBlock * (size_t)blockIdx.x, sv, si);
}

// Two threads per column: thread (half h, column t) carries the breakpoints 20h .. 20h+19 of a chunk
// (40 accumulators: two waves per SIMD can be resident and hide each other's LDS latency).
constexpr int kVfyThreads = 2 * (int)kSoloWidth;
constexpr uint32_t kVfHalf = kSoloChunk / 2;

__global__ __launch_bounds__(kVfyThreads)
void k_la_verify(uint32_t n, int full_g, const float* __restrict__ gcache, uint32_t gpitch,
                 const int32_t* __restrict__ slot_of, const float* __restrict__ c0,
                 const uint32_t* __restrict__ log, const uint8_t* __restrict__ sub_pos,
                 const DevState* __restrict__ st, uint32_t* __restrict__ v_max, uint64_t* __restrict__ v_min,
                 uint32_t nvwg, float* __restrict__ tcand, uint64_t* __restrict__ cand_top, int tie_guard, uint32_t nrows,
                 uint64_t* dbg, uint32_t* tie_flag)
{
    uint64_t tsv[8];
    tsv[0] = wall_clock64();
    __shared__ uint32_t s_tieok[kSoloChunk];            // breakpoint k: lambda is where the previous step left it (tie_band) ...
    __shared__ uint32_t s_tiejr[kSoloChunk];            // ... and the column that left the support in the step before (or none)
    __shared__ float sX[kVfUnion][kSoloChunk];          // x_S of union column r at breakpoint k (0 when absent)
    __shared__ float sD[kVfUnion][kSoloChunk];
    __shared__ uint64_t sRed[kSoloChunk * kVfRedPitch]; // reductions: one row per breakpoint
    __shared__ __attribute__((aligned(16))) uint32_t s_sub[kSoloWidth];   // header: columns of the subset
    __shared__ uint32_t s_row[kSoloWidth];              // header: their Gram rows
    __shared__ uint32_t s_pres[kSoloWidth];             // bit k: position is in the support at breakpoint k (k < 32) ...
    __shared__ uint32_t s_preh[kSoloWidth];             // ... and at breakpoint 32 + k
    __shared__ __attribute__((aligned(16))) uint32_t s_part[kSoloWidth];  // != 0: position is in the support at some breakpoint of the chunk
    __shared__ uint32_t s_rank[kSoloWidth];             // position -> index in the union (sorted by column)
    __shared__ uint32_t s_urow[kVfUnion];               // union index -> Gram row
    __shared__ uint32_t s_hdr[kSoloChunk][8];
    __shared__ uint32_t s_U;
    __shared__ float sv[16];
    __shared__ uint32_t si[16];

    if (st->solo_pending != 1u) return;                  // (2 = a replay: its iterations were verified before)
    const uint32_t nlog = st->solo_nlog;
    if (nlog == 0u) return;
    const uint32_t tid = threadIdx.x, wg = blockIdx.x;
    const uint32_t tcol = tid & (kSoloWidth - 1u), half = tid / kSoloWidth;
    const uint32_t i = wg * kSoloWidth + tcol;
    const bool valid = i < n;
    const float c0v = valid ? c0[i] : 0.f;
    const bool cached = valid && slot_of[i] >= 0;
    if (tid < kSoloWidth) {
        // (the Gram row of a subset column is looked up NOW, not taken from the header: in the early form the
        // solo launch ran before its columns had rows, which the passes over A have filled in since)
        const uint32_t cl = log[tid];
        s_sub[tid] = cl;
        s_row[tid] = cl < n ? (full_g ? cl : (uint32_t)slot_of[cl]) : 0xffffffffu;
    }
    if (tid == 0) s_U = 0u;                              // (free until the first chunk: 1 + index of the last scan entry)
    __syncthreads();
    // this thread's column in the subset?  (sub_pos is only trusted if the header agrees)
    uint32_t mypos = 0xffffffffu;
    if (valid) {
        const uint32_t p = sub_pos[i];
        if (s_sub[p] == i) mypos = p;
    }
    // the last entry that carries a scan: its candidates feed tcand and the next subset
    // (one parallel load of the flag words; s_U is free until the first chunk)
    static_assert(kSoloLogCap <= (uint32_t)kVfyThreads && kSoloLogCap % 64u == 0u, "whole waves look at the flag words");
    if (tid < kSoloLogCap) {
        const uint32_t fl = tid < nlog ? (log[kSoloHeaderWords + (size_t)tid * kSoloEntryWords + 1] & 1u) : 0u;
        const uint64_t b = __ballot(fl != 0u);
        if ((tid & 63u) == 0u && b) atomicMax(&s_U, (tid & ~63u) + 64u - (uint32_t)__builtin_clzll(b));
    }
    __syncthreads();
    const uint32_t last_scan = s_U ? s_U - 1u : 0xffffffffu;

    const uint32_t* entries = log + kSoloHeaderWords;
    tsv[1] = wall_clock64();
    for (uint32_t k0 = 0; k0 < nlog; k0 += kSoloChunk) {
        const uint32_t J = nlog - k0 < kSoloChunk ? nlog - k0 : kSoloChunk;
        __syncthreads();
        // ---- the chunk's union of support columns, sorted by column; coefficient tables ---------------
        if (tid < kSoloWidth) { s_pres[tid] = 0u; s_preh[tid] = 0u; }
        // the chunk's entries, one bulk copy into LDS (the reduction buffer is free here): everything below
        // reads them from there — loads from the log issued entry by entry cost a memory round trip each
        uint32_t* const sE = reinterpret_cast<uint32_t*>(sRed);
        {
            const uint32_t* src = entries + (size_t)k0 * kSoloEntryWords;
            const uint32_t tot = J * kSoloEntryWords;
            for (uint32_t e0 = 0; e0 < tot; e0 += 8 * kVfyThreads) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const uint32_t e = e0 + (uint32_t)u * kVfyThreads + tid; v[u] = e < tot ? src[e] : 0u; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const uint32_t e = e0 + (uint32_t)u * kVfyThreads + tid; if (e < tot) sE[e] = v[u]; }
            }
        }
        if (tid == 0) s_U = 0u;
        __syncthreads();
        if (k0 == 0) tsv[2] = wall_clock64();
        if (tid < J * 8u) s_hdr[tid >> 3][tid & 7u] = sE[(tid >> 3) * kSoloEntryWords + (tid & 7u)];
        if (tid < J) {
            // tie stalls (DevState::tie_stall): an exact-zero candidate counts only while lambda is where the previous
            // step left it, and never for the column that has just left the support — the rules of k_scansel, from the
            // log: the previous entry (the state the launch started from for the first one) has lambda, gamma and the pick
            const uint32_t g = k0 + tid;
            const uint32_t* cur = sE + tid * kSoloEntryWords;
            float lam_prev, gam_prev;
            uint32_t jr = 0xffffffffu;
            if (g == 0u) {
                lam_prev = (float)st->c_inf; gam_prev = (float)st->gamma;
                if (st->iter >= 1u && st->added == 0u) jr = st->idx;
            } else {
                const uint32_t* pe = entries + (size_t)(g - 1u) * kSoloEntryWords;
                lam_prev = __uint_as_float(pe[4]); gam_prev = __uint_as_float(pe[5]);
                if (cur[0] + 1u == pe[0]) jr = pe[3];                     // the support shrank: that pick was a removal
            }
            s_tieok[tid] = tie_band<float>(__uint_as_float(cur[4]), lam_prev, gam_prev, st->lambda0) ? 1u : 0u;
            s_tiejr[tid] = jr;
        }
        for (uint32_t pr = tid; pr < J * kSoloListPitch; pr += kVfyThreads) {
            const uint32_t k = pr / kSoloListPitch, j = pr - k * kSoloListPitch;
            const uint32_t* e = sE + k * kSoloEntryWords;
            if (j < e[0]) {
                const uint32_t pos = e[8 + kSoloListPitch + j] & (kSoloWidth - 1u);
                if (k < 32u) atomicOr(&s_pres[pos], 1u << k); else atomicOr(&s_preh[pos], 1u << (k - 32u));
            }
        }
        __syncthreads();
        if (tid < kSoloWidth) s_part[tid] = s_pres[tid] | s_preh[tid];
        __syncthreads();
        if (tid < kSoloWidth) {
            const bool part = s_part[tid] != 0u;
            const uint32_t mycol = s_sub[tid];
            uint32_t r = 0;
            if (part) {
                const uint4* p4 = reinterpret_cast<const uint4*>(s_part);
                const uint4* c4 = reinterpret_cast<const uint4*>(s_sub);
#pragma unroll 8
                for (uint32_t u = 0; u < kSoloWidth / 4; ++u) {
                    const uint4 pp = p4[u], cc = c4[u];
                    r += (pp.x != 0u && cc.x < mycol) ? 1u : 0u;
                    r += (pp.y != 0u && cc.y < mycol) ? 1u : 0u;
                    r += (pp.z != 0u && cc.z < mycol) ? 1u : 0u;
                    r += (pp.w != 0u && cc.w < mycol) ? 1u : 0u;
                }
            }
            s_rank[tid] = part ? r : 0xffffffffu;
            if (part) {
                atomicAdd(&s_U, 1u);
                if (r < kVfUnion) s_urow[r] = s_row[tid] < nrows ? s_row[tid] : 0u;   // (a support column always has its Gram row)
            }
        }
        __syncthreads();
        const uint32_t U = s_U < kVfUnion ? s_U : kVfUnion;      // (U <= 96 + 32 by construction)
        for (uint32_t e = tid; e < U * kSoloChunk; e += kVfyThreads) {       // absent columns contribute exact zeros
            (&sX[0][0])[e] = 0.f;
            (&sD[0][0])[e] = 0.f;
        }
        __syncthreads();
        for (uint32_t pr = tid; pr < J * kSoloListPitch; pr += kVfyThreads) {
            const uint32_t k = pr / kSoloListPitch, j = pr - k * kSoloListPitch;
            const uint32_t* e = sE + k * kSoloEntryWords;
            if (j < e[0]) {
                const uint32_t r = s_rank[e[8 + kSoloListPitch + j] & (kSoloWidth - 1u)];
                if (r < kVfUnion) {
                    sX[r][k] = __uint_as_float(e[8 + 2 * kSoloListPitch + j]);
                    sD[r][k] = __uint_as_float(e[8 + 3 * kSoloListPitch + j]);
                }
            }
        }
        __syncthreads();

        if (k0 == 0) tsv[3] = wall_clock64();
        // ---- one pass over the union's Gram rows: 2 x 16 sums per thread, sorted-support order ---------
        const uint32_t kb = half * kVfHalf;                // this thread's breakpoints: kb .. kb + 15 of the chunk
        float ax[kVfHalf], ad[kVfHalf];
#pragma unroll
        for (int k = 0; k < (int)kVfHalf; ++k) { ax[k] = 0.f; ad[k] = 0.f; }
        const size_t ci = valid ? i : 0u;
        constexpr int RB = 8;                              // Gram rows in flight per thread (each a trip to L2 / HBM)
        if (kb < J) {
            for (uint32_t r0 = 0; r0 < U; r0 += RB) {
                float gv[RB];
#pragma unroll
                for (int t = 0; t < RB; ++t) {
                    const uint32_t r = r0 + (uint32_t)t < U ? r0 + (uint32_t)t : r0;
                    gv[t] = gcache[(size_t)s_urow[r] * gpitch + ci];
                }
#pragma unroll
                for (int t = 0; t < RB; ++t) {
                    if (r0 + (uint32_t)t >= U) break;
                    const float g = gv[t];
                    const float* xr = sX[r0 + t] + kb;
                    const float* dr = sD[r0 + t] + kb;
#pragma unroll
                    for (int kg = 0; kg < (int)kVfHalf; kg += 4) {
                        if (kb + (uint32_t)kg >= J) break;             // (uniform per wave: whole groups of 4 breakpoints)
#pragma unroll
                        for (int k = kg; k < kg + 4; ++k) {
                            ax[k] += xr[k] * g;
                            ad[k] += dr[k] * g;
                        }
                    }
                }
            }
        }

        if (k0 == 0) tsv[4] = wall_clock64();
        // ---- max |c| per breakpoint ---------------------------------------------------------------------
        uint32_t* red32 = reinterpret_cast<uint32_t*>(sRed);
#pragma unroll
        for (int k = 0; k < (int)kVfHalf; ++k) {
            const float cv = c0v - ax[k];
            const float a = cv < 0.f ? -cv : cv;
            red32[(kb + k) * kVfRedPitch + tcol] = (valid && kb + (uint32_t)k < J) ? (__float_as_uint(a) & 0x7fffffffu) : 0u;
        }
        __syncthreads();
        for (uint32_t k = tid >> 4; k < kSoloChunk; k += kVfyThreads / 16) {
            const uint32_t sub = tid & 15u;
            uint32_t m = 0u;
            for (uint32_t j = 0; j < kSoloWidth / 16; ++j) {
                const uint32_t v = red32[k * kVfRedPitch + j * 16u + sub];
                m = v > m ? v : m;                      // (|c| >= 0: the bit patterns order like the values; NaN wins)
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
            if (sub == 0 && k < J) v_max[(size_t)(k0 + k) * nvwg + wg] = m;
        }
        __syncthreads();

        if (k0 == 0) tsv[5] = wall_clock64();
        // ---- step-length candidates per breakpoint (find_max_gamma's off-support part, :137-159) --------
        const uint32_t pres = mypos != 0xffffffffu ? s_pres[mypos] : 0u;
        const uint32_t preh = mypos != 0xffffffffu ? s_preh[mypos] : 0u;
        float m_last = Lim<float>::max();
        bool act_last = false, own_last = false;
#pragma unroll
        for (int k = 0; k < (int)kVfHalf; ++k) {
            const uint32_t kk = kb + (uint32_t)k;
            uint64_t pk = ~0ull;
            if (valid && kk < J && (s_hdr[kk][1] & 1u)) {
                const bool act = kk < 32u ? ((pres >> kk) & 1u) : ((preh >> (kk - 32u)) & 1u);
                float m = Lim<float>::max();
                if (!act) {
                    const float c_inf = __uint_as_float(s_hdr[kk][4]);
                    const float ci2 = c0v - ax[k], qi = ad[k];
                    const float dl = 1.f - qi, dr = 1.f + qi;
                    if (dl != 0.f) {
                        float t = (c_inf - ci2) / dl;
                        if (tie_guard && t == 0.f && dl > 0.f) t = Lim<float>::tiny();
                        if (t == 0.f && s_tieok[kk] && i != s_tiejr[kk]) __hip_atomic_store(tie_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (t > 0.f && t < m) m = t;
                    }
                    if (dr != 0.f) {
                        float t = (c_inf + ci2) / dr;
                        if (tie_guard && t == 0.f && dr > 0.f) t = Lim<float>::tiny();
                        if (t == 0.f && s_tieok[kk] && i != s_tiejr[kk]) __hip_atomic_store(tie_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (t > 0.f && t < m) m = t;
                    }
                }
                if (m < Lim<float>::max()) pk = ((uint64_t)__float_as_uint(m) << 32) | i;
                if (k0 + kk == last_scan) { m_last = m; act_last = act; own_last = true; }
            }
            sRed[kk * kVfRedPitch + tcol] = pk;
        }
        __syncthreads();
        for (uint32_t k = tid >> 4; k < kSoloChunk; k += kVfyThreads / 16) {
            const uint32_t sub = tid & 15u;
            uint64_t m = ~0ull;
            for (uint32_t j = 0; j < kSoloWidth / 16; ++j) {
                const uint64_t v = sRed[k * kVfRedPitch + j * 16u + sub];
                m = v < m ? v : m;                      // positive floats order like their bits; ties -> left-most column
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)m, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(m >> 32), o);
                const uint64_t v = ((uint64_t)hi << 32) | lo;
                m = v < m ? v : m;
            }
            if (sub == 0 && k < J) v_min[(size_t)(k0 + k) * nvwg + wg] = m;
        }
        if (k0 == 0) tsv[6] = wall_clock64();
        // ---- what the next steps rank by (from the last scan of the log; the thread that carries it) -------
        if (last_scan >= k0 && last_scan < k0 + J) {
            __syncthreads();
            // (full-G mode: every column is "cached" and k_la_top never runs; tcand then only ranks subsets)
            if (own_last) tcand[i] = (act_last || (cached && !full_g)) ? Lim<float>::max() : m_last;
            // next subset: cached columns come in through their slots (cache mode), the support through its lists
            const bool offer = own_last && !act_last && (full_g || !cached) && m_last < Lim<float>::max();
            block_topn(offer ? m_last : Lim<float>::max(), offer ? i : 0xffffffffu, cand_top + kCandPerBlock * (size_t)wg, sv, si);
        }
    }
    if (dbg != nullptr && wg == 0 && tid == 0) {       // developer aid: stage timestamps of workgroup 0 (row 1600 + entries)
        tsv[7] = wall_clock64();
        for (int q2 = 0; q2 < 8; ++q2) dbg[(size_t)(1600 + nlog) * 8 + q2] = tsv[q2];
    }
}

// Compares the log with the truth, then commits or withholds what the solo launch staged; always counts
// the launch group for the host pump.
//   every entry verified   the staged state becomes the state of the solve (exactly what k_la_persist's own
//                          hand-over writes) and its outcome is released to the host flags;
//   entry k* failed        nothing is committed.  With g > 0 verified iterations before it the next solo
//                          launch repeats exactly those (same subset, same arithmetic: same state) and
//                          hands over to the resident form; with g = 0 the resident form goes on at once.
struct CommitArgs {
    const uint32_t* stage;
    float* x; float* d; uint8_t* insup; uint32_t* gam2; float* inv0; float* inv1; uint32_t kcap; LaSync* sy;
    uint32_t* touched2; int zero_on_removal;
};

constexpr int kPubThreads = 1024;            // 8 threads per log entry; the commit copies with all of them

__global__ __launch_bounds__(kPubThreads)
void k_la_vpublish(const uint32_t* __restrict__ log, DevState* st, const uint32_t* __restrict__ v_max,
                   const uint64_t* __restrict__ v_min, uint32_t nvwg, uint32_t* hflags, CommitArgs ca)
{
    __shared__ uint32_t s_first, s_good, s_scan;
    const uint32_t tid = threadIdx.x;
    const uint32_t pending = st->solo_pending;
    if (!pending) {
        if (tid == 0) bump_seq(st, hflags);
        return;
    }
    const uint32_t nlog = st->solo_nlog;
    if (tid == 0) { s_first = nlog; s_good = 0u; s_scan = 0u; }
    __syncthreads();
    // eight threads per entry
    static_assert(kSoloLogCap * 8u <= (uint32_t)kPubThreads, "eight threads per log entry");
    const uint32_t k = tid >> 3, sub = tid & 7u;
    const bool has_scan = k < nlog && (log[kSoloHeaderWords + (size_t)k * kSoloEntryWords + 1] & 1u);
    if (pending == 1u && k < nlog) {
        uint32_t mx = 0u;
        uint64_t mn = ~0ull;
        const uint32_t* pm = v_max + (size_t)k * nvwg;
        const uint64_t* pn = v_min + (size_t)k * nvwg;
        // this thread's eighth of the partials, sixteen independent loads at a time
        const uint32_t per = (nvwg + 7u) / 8u;
        const uint32_t b_lo = sub * per, b_hi = (b_lo + per < nvwg) ? b_lo + per : nvwg;
        for (uint32_t b0 = b_lo; b0 < b_hi; b0 += 16) {
            uint32_t a[16];
            uint64_t c[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const uint32_t b = b0 + (uint32_t)t;
                a[t] = b < b_hi ? pm[b] : 0u;
                c[t] = b < b_hi ? pn[b] : ~0ull;
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) { mx = a[t] > mx ? a[t] : mx; mn = c[t] < mn ? c[t] : mn; }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)mn, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(mn >> 32), o);
            const uint64_t v = ((uint64_t)hi << 32) | lo;
            mn = v < mn ? v : mn;
        }
        if (sub == 0) {
            const uint32_t* e = log + kSoloHeaderWords + (size_t)k * kSoloEntryWords;
            bool ok = (e[4] & 0x7fffffffu) == mx;                    // lambda = ||c||_inf over all columns
            if (e[1] & 1u) {
                // off-support minimum over all columns, merged with the support's own candidate (logged)
                float g = Lim<float>::max();
                uint32_t idx = 0xffffffffu;
                if (mn != ~0ull) { g = __uint_as_float((uint32_t)(mn >> 32)); idx = (uint32_t)mn; }
                const float gs = __uint_as_float(e[6]);
                const uint32_t is = e[7];
                if (is != 0xffffffffu && better_min(gs, is, g, idx)) { g = gs; idx = is; }
                ok = ok && __float_as_uint(g) == e[5] && idx == e[3];
            }
            if (!ok) atomicMin(&s_first, k);
        }
    }
    __syncthreads();
    const uint32_t first_bad = s_first;
    if (sub == 0 && has_scan) {
        atomicOr(&s_scan, 1u);
        if (k < first_bad) atomicAdd(&s_good, 1u);
    }
    __syncthreads();
    if (pending == 1u && first_bad < nlog) {
        // a column outside the subset would have changed lambda or the pick at entry first_bad
        if (tid == 0) {
            st->solo_fails += 1u;
            if (s_good == 0u) {
                st->solo_off = 1;
                __hip_atomic_store(&hflags[4], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                st->solo_replay = s_good;
            }
            st->solo_pending = 0;
            bump_seq(st, hflags);
        }
        return;
    }

    // ---- commit: the staged hand-over becomes the state (cf. the end of k_la_persist) -----------------
    const uint32_t* sg = ca.stage;
    constexpr uint32_t LP = kSoloListPitch;
    const uint32_t code = sg[9];
    const uint32_t K = sg[0];
    const bool save = sg[1] != 0u;
    if (code != kPsExitNothing) {
        const uint32_t kcap = ca.kcap;
        const uint32_t cur = st->cur;
        float* const Ig = cur ? ca.inv1 : ca.inv0;
        uint32_t* const gam_cur = ca.gam2 + (size_t)cur * kcap;
        uint32_t* const gam_alt = ca.gam2 + (size_t)(cur ^ 1u) * kcap;
        const uint32_t K0 = st->K;
        // the old support leaves the dense vectors and the membership flags ...
        for (uint32_t j = tid; j < K0; j += kPubThreads) {
            const uint32_t cl = gam_cur[j];
            ca.x[cl] = 0.f; ca.d[cl] = 0.f; ca.insup[cl] = 0;
        }
        __syncthreads();
        // ... the new one enters (the inverse: four independent elements per thread and step)
        for (uint32_t e0 = tid; e0 < K * K; e0 += 4 * kPubThreads) {
            uint32_t v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { const uint32_t e = e0 + (uint32_t)t * kPubThreads; v[t] = e < K * K ? sg[kSoloStageHead + 4 * LP + 1 + e] : 0u; }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t e = e0 + (uint32_t)t * kPubThreads;
                if (e < K * K) { const uint32_t a = e / K, b = e - a * K; Ig[(size_t)a * kcap + b] = __uint_as_float(v[t]); }
            }
        }
        for (uint32_t j = tid; j < K; j += kPubThreads) {
            const uint32_t cl = sg[kSoloStageHead + j];
            gam_cur[j] = cl;
            ca.x[cl] = __uint_as_float(sg[kSoloStageHead + LP + j]);
            ca.d[cl] = __uint_as_float(sg[kSoloStageHead + 2 * LP + j]);
            ca.insup[cl] = 1;
        }
        if (save)
            for (uint32_t j = tid; j < K + 1u; j += kPubThreads) gam_alt[j] = sg[kSoloStageHead + 3 * LP + j];
        if (!ca.zero_on_removal) {
            // reference mode: the `touched` list k_la_iter / k_la_cq walk (columns whose x may be non-zero).  A solo
            // launch never commits a removal that leaves a rounding residue (it hands over before), so the list
            // that will be current — the staged support, or the pending one — is all of it.
            uint32_t* const tch = ca.touched2 + (size_t)(save ? (cur ^ 1u) : cur) * kcap;
            const uint32_t Kt = save ? K + 1u : K;
            for (uint32_t j = tid; j < Kt; j += kPubThreads) tch[j] = sg[kSoloStageHead + (save ? 3u : 0u) * LP + j];
        }
    }
    __syncthreads();
    if (tid != 0) return;
    if (code != kPsExitNothing) {
        if (save) ca.insup[sg[5]] = 1;                 // the entering column whose inverse update is pending
        ca.sy->tick = sg[8];
        st->K = save ? K + 1u : K;
        st->iter = sg[2];
        st->c_inf = (double)__uint_as_float(sg[3]);
        st->gamma = (double)__uint_as_float(sg[4]);
        st->idx = sg[5];
        st->rank = sg[6];
        st->added = sg[7];
        if (!ca.zero_on_removal) st->ntouched = save ? K + 1u : K;
        st->subg_active = 0;                           // later solo launches pick a new subset: Gram-column cache
    }
    bool off = pending == 2u;                          // after a replay the resident form goes on
    if (code == kPsExitDone) {
        st->done_round = sg[10];
        st->need_sweep = 0;
        st->done = 1;
        signal_done(hflags, nullptr, 1u, sg[10]);
    } else if (code == kPsExitMiss) {
        st->need_sweep = 1;
        const uint32_t nm = st->nmiss + 1u;
        st->nmiss = nm;
        __hip_atomic_store(&hflags[2], nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (code == kPsExitGrow || code == kPsExitNothing) {
        __hip_atomic_store(&hflags[3], K + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        off = true;
    } else if (code == kPsExitHandOver) {
        off = true;
    }
    if (off) {
        st->solo_off = 1;
        __hip_atomic_store(&hflags[4], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    st->solo_replay = 0;
    if (pending == 1u && s_scan) st->cand_scan = 1;
    st->solo_pending = 0;
    bump_seq(st, hflags);
}

// The columns of the next lookahead sweep, from the per-block candidate tops instead of a scan of all n
// step-length candidates (k_la_top, one workgroup over n values: 31 us at n = 65536): the entering column
// first, then the best-ranked offers that are neither active nor cached.  Same outputs as k_la_top.
constexpr int kTcThreads = 1024;             // one offer per thread up to 1024 candidates (4 per block at n = 65536)
constexpr int kTcStride = 64;                 // sw_list layout: rcols[64] then drows[64] (k_la_top's kSwStride)
// Wider dictionaries have more candidates than threads (4 per 256-column block: 1536 at n = 98304).  Round 2 let a thread
// keep only the BEST of its strided share — and dropped, now and then, a column that was about to enter: the subset then
// missed it, the first speculative launch failed its check on nearly every solve beyond 65536 columns, and after eight
// solves the context stopped speculating (3.5 ms per solve at 98304 columns against 1.5 at 65536: the "cliff" was this,
// not the shape of the passes).  Now a thread keeps up to kTcOffers offers (ncand <= 4096: n <= 262144; only beyond that
// is the excess folded into the last offer) and every offer is ranked by counting over all of them.
constexpr int kTcOffers = 4;

// ranks the valid offers o[0 .. kTcOffers) of every thread among all offers of the workgroup (s_of: kTcOffers * kTcThreads
// words): rank[r] = number of smaller keys; returns the number of valid offers in all.  nper = offers per thread in use.
__device__ __forceinline__ uint32_t rank_offers(const uint64_t (&o)[kTcOffers], uint32_t nper, uint64_t* s_of, uint32_t (&rank)[kTcOffers])
{
    const uint32_t tid = threadIdx.x;
    uint32_t mine = 0;
#pragma unroll
    for (int r = 0; r < kTcOffers; ++r) {
        if ((uint32_t)r < nper) { s_of[(uint32_t)r * kTcThreads + tid] = o[r]; mine += o[r] != ~0ull ? 1u : 0u; }
        rank[r] = 0u;
    }
    __syncthreads();
    const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(s_of);
    const uint32_t words2 = nper * (kTcThreads / 2);
#pragma unroll
    for (int r = 0; r < kTcOffers; ++r) {
        if ((uint32_t)r >= nper || o[r] == ~0ull) continue;
        const uint64_t me = o[r];
        uint32_t c = 0;
#pragma unroll 8
        for (uint32_t u = 0; u < words2; ++u) {
            const ulonglong2 v = p2[u];
            c += v.x < me ? 1u : 0u;
            c += v.y < me ? 1u : 0u;
        }
        rank[r] = c;
    }
    // total number of valid offers (block sum through the ballot counts of the barrier)
    uint32_t total = 0;
#pragma unroll
    for (int r = 0; r < kTcOffers; ++r)
        total += (uint32_t)__syncthreads_count((uint32_t)r < nper && o[r] != ~0ull);
    (void)mine;
    return total;
}

__global__ __launch_bounds__(kTcThreads)
void k_la_top_cand(const uint64_t* __restrict__ cand_top, uint32_t ncand, uint32_t n,
                   const uint8_t* __restrict__ insup, int32_t* __restrict__ slot_of, uint32_t gcap,
                   uint32_t* __restrict__ sw_list, DevState* st, uint32_t* hflags, uint32_t* __restrict__ slot_col,
                   uint32_t nsel /* columns of this sweep: 32, or 64 for the first one of a solve */)
{
    __shared__ __attribute__((aligned(16))) uint64_t s_of[kTcOffers * kTcThreads];
    if (st->done || !st->need_sweep) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t idx = st->idx;
    const uint32_t used = st->cache_used;                                 // (read before the barrier: thread 0 updates it at the end)
    // up to kTcOffers offers per thread, every one ranked by counting over all of them
    const uint32_t nper = ncand <= (uint32_t)kTcThreads ? 1u : (ncand + kTcThreads - 1u) / kTcThreads < (uint32_t)kTcOffers
                              ? (ncand + kTcThreads - 1u) / kTcThreads : (uint32_t)kTcOffers;
    uint64_t o[kTcOffers];
#pragma unroll
    for (int r = 0; r < kTcOffers; ++r) o[r] = ~0ull;
    for (uint32_t e = tid, r = 0; e < ncand; e += kTcThreads, ++r) {
        const uint64_t pk = cand_top[e];
        const uint32_t cl = (uint32_t)pk;
        if (pk == ~0ull || cl >= n || cl == idx || insup[cl] || slot_of[cl] >= 0) continue;
        const uint32_t rr = r < nper ? r : nper - 1u;                     // (beyond kTcOffers * 1024 candidates: the best of the excess)
#pragma unroll
        for (int q = 0; q < kTcOffers; ++q) if ((uint32_t)q == rr && pk < o[q]) o[q] = pk;
    }
    uint32_t rk[kTcOffers];
    const uint32_t total = rank_offers(o, nper, s_of, rk);
    const uint32_t room = gcap > used ? gcap - used : 0u;                 // slots left (the entering column takes the first)
    uint32_t count = total < nsel - 1u ? total : nsel - 1u;
    if (room == 0u) count = 0u; else if (count + 1u > room) count = room - 1u;
#pragma unroll
    for (int r = 0; r < kTcOffers; ++r) {
        if ((uint32_t)r < nper && o[r] != ~0ull && rk[r] < count) {
            const uint32_t r1 = rk[r];
            const uint32_t cl = (uint32_t)o[r], sl = used + 1u + r1;
            sw_list[1 + r1] = cl; sw_list[kTcStride + 1 + r1] = sl; slot_of[cl] = (int32_t)sl;
            if (slot_col != nullptr) slot_col[sl] = cl;
        }
    }
    if (tid >= 1u + count && tid < (uint32_t)kTcStride) { sw_list[tid] = 0xffffffffu; sw_list[kTcStride + tid] = 0xffffffffu; }
    if (tid == 0) {
        if (room > 0u) {
            sw_list[0] = idx; sw_list[kTcStride] = used; slot_of[idx] = (int32_t)used;
            if (slot_col != nullptr) slot_col[used] = idx;
            st->cache_used = used + 1u + count;
            st->nsweeps += 1;
        } else {
            // cache budget exhausted: the host re-runs the solve in residual form (as k_la_top does)
            sw_list[0] = 0xffffffffu; sw_list[kTcStride] = 0xffffffffu;
            st->status = kStatusRetryResidual;
            st->need_sweep = 0;
            st->done = 1;
            signal_done(hflags, nullptr, 1u, st->iter);
        }
    }
}

// ===== early form: the iterations start on the subset Gram matrix while A is still being swept ================
// (homotopy.hip, Lookahead::init_early; csrc/subgram.hip)

// The subset of the first solo launch and the slots of the first 64 Gram columns, from one ranking of the
// per-block candidate tops by |c0|: position 0 / slot 0 = the first pick, positions 1..255 = the best-ranked
// candidates, the first 63 of which also get cache slots 1..63 (two 32-column passes: sw_list[0..31] / [64..95]
// and sw_list[32..63] / [96..127]).
__global__ __launch_bounds__(kTcThreads)
void k_subset_pick(const uint64_t* __restrict__ cand_top, uint32_t ncand, uint32_t n, int32_t* __restrict__ slot_of,
                   uint32_t* __restrict__ sw_list, uint32_t* __restrict__ sub_cols, DevState* st,
                   uint32_t* __restrict__ slot_col, uint32_t subset_cap /* option solo_subset: columns beyond the support */,
                   uint32_t* __restrict__ se_count /* counters of the passes dealt out by shader engine, or null */)
{
    if (se_count != nullptr && threadIdx.x < 2u * (kSeCount + 2u)) se_count[threadIdx.x] = 0u;
    // (the progress hints of the solo launch that follows, one per subset position: nothing known yet)
    if (threadIdx.x < kSoloWidth) reinterpret_cast<float*>(sub_cols + kSoloWidth)[threadIdx.x] = Lim<float>::max();
    __shared__ __attribute__((aligned(16))) uint64_t s_of[kTcOffers * kTcThreads];
    if (st->done) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t idx = st->idx;
    const uint32_t nper = ncand <= (uint32_t)kTcThreads ? 1u : (ncand + kTcThreads - 1u) / kTcThreads < (uint32_t)kTcOffers
                              ? (ncand + kTcThreads - 1u) / kTcThreads : (uint32_t)kTcOffers;
    uint64_t o[kTcOffers];
#pragma unroll
    for (int r = 0; r < kTcOffers; ++r) o[r] = ~0ull;
    for (uint32_t e = tid, r = 0; e < ncand; e += kTcThreads, ++r) {
        const uint64_t pk = cand_top[e];
        const uint32_t cl = (uint32_t)pk;
        if (pk == ~0ull || cl >= n || cl == idx) continue;
        const uint32_t rr = r < nper ? r : nper - 1u;
#pragma unroll
        for (int q = 0; q < kTcOffers; ++q) if ((uint32_t)q == rr && pk < o[q]) o[q] = pk;
    }
    uint32_t rk[kTcOffers];
    const uint32_t total = rank_offers(o, nper, s_of, rk);
    uint32_t nsub = total < kSoloWidth - 1u ? total : kSoloWidth - 1u;             // subset positions 1 .. nsub
    if (nsub > subset_cap) nsub = subset_cap;
    const uint32_t nslot = total < 63u ? total : 63u;                              // cache slots 1 .. nslot are set aside:
    const uint32_t nfirst = nslot < 31u ? nslot : 31u;                             // 1 .. nfirst for the first pass (given here),
#pragma unroll
    for (int r = 0; r < kTcOffers; ++r) {                                          // 32 .. nslot for the second (k_pick_pass_b)
        if ((uint32_t)r < nper && o[r] != ~0ull && rk[r] < nsub) {
            const uint32_t r1 = rk[r];
            const uint32_t cl = (uint32_t)o[r];
            sub_cols[1u + r1] = cl;
            if (r1 < nfirst) {
                sw_list[1u + r1] = cl; sw_list[kTcStride + 1u + r1] = 1u + r1; slot_of[cl] = (int32_t)(1u + r1);
                if (slot_col != nullptr) slot_col[1u + r1] = cl;
            }
        }
    }
    if (tid >= 1u + nsub && tid < kSoloWidth) sub_cols[tid] = 0xffffffffu;
    if (tid >= 1u + nfirst && tid < (uint32_t)kTcStride) { sw_list[tid] = 0xffffffffu; sw_list[kTcStride + tid] = 0xffffffffu; }
    if (tid == 0) {
        sub_cols[0] = idx;
        sw_list[0] = idx; sw_list[kTcStride] = 0u; slot_of[idx] = 0;
        if (slot_col != nullptr) slot_col[0] = idx;
        st->cache_used = 1u + nslot;
        st->nsweeps += nslot >= 32u ? 2u : 1u;
        st->subg_active = 1;
    }
}

// The second pass's columns (second stream, between the two passes; the solo launch is half a millisecond old):
// of the subset's columns that have no Gram row yet, first those that ENTERED the launch's support so far — the
// verification will need their rows whatever happens — then the ones closest to entering by the launch's latest
// step-length candidates, then by |c0| rank (= subset position).  The hints are read while the launch writes them:
// a stale or torn view only changes which columns are fetched, never a result — whatever is still missing afterwards
// is found by k_missing_cols.  adapt = 0: by |c0| rank alone (positions 32..63, the round-2 first version).
__global__ __launch_bounds__(kSoloWidth)
void k_pick_pass_b(const uint32_t* __restrict__ sub_cols, uint32_t n, int32_t* __restrict__ slot_of,
                   uint32_t* __restrict__ slot_col, uint32_t* __restrict__ sw_list, const DevState* __restrict__ st, int adapt)
{
    __shared__ __attribute__((aligned(16))) uint64_t s_key[kSoloWidth];
    if (st->done) return;                                   // (the lists' second halves are empty since k_subset_pick)
    const uint32_t tid = threadIdx.x;
    const uint32_t cl = sub_cols[tid];
    const float* prog = reinterpret_cast<const float*>(sub_cols + kSoloWidth);
    uint64_t key = ~0ull;
    if (cl < n && slot_of[cl] < 0) {
        float p = adapt ? __hip_atomic_load(&prog[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : Lim<float>::max();
        if (!(p == p) || p > Lim<float>::max()) p = Lim<float>::max();
        const float k = p < 0.f ? 0.f : p;                  // entered: first of all
        key = ((uint64_t)__float_as_uint(k) << 32) | (uint64_t)tid;
    }
    s_key[tid] = key;
    __syncthreads();
    uint32_t r = 0;
    if (key != ~0ull) {
        const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(s_key);
#pragma unroll 8
        for (uint32_t u = 0; u < kSoloWidth / 2; ++u) {
            const ulonglong2 v = p2[u];
            r += v.x < key ? 1u : 0u;
            r += v.y < key ? 1u : 0u;
        }
        if (r < 32u) {
            const uint32_t sl = 32u + r;                    // slots 32..63 were set aside by k_subset_pick
            sw_list[32u + r] = cl; sw_list[kTcStride + 32u + r] = sl; slot_of[cl] = (int32_t)sl;
            if (slot_col != nullptr) slot_col[sl] = cl;
        }
    }
}

// second stream: hold the passes over A back until the solo workgroup is resident (it needs a whole CU's LDS; once
// the barrier-free pass has spread its single-wave workgroups over every CU there is no room for it until they
// finish).  Bounded: a solo launch that never starts only costs the overlap.
__global__ void k_wait_started(const DevState* st)
{
    if (threadIdx.x != 0) return;
    for (uint32_t spin = 0; spin < 4000u; ++spin) {
        if (__hip_atomic_load(&st->solo_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        if (__hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        __builtin_amdgcn_s_sleep(32);
    }
}

// After the first solo launch of the early form: every column that was in the support at any of its breakpoints
// (or was its last pick) needs its full Gram row before the verification can re-derive the breakpoints over all n
// columns.  The first 64 ranked columns have theirs (the two passes that ran beside the launch); the others get
// slots here and up to two more passes fetch them (sw_list2, same layout as sw_list).  More than 64 of them, or a
// full cache: the host re-runs the solve in the plain form (kStatusRetryPlain).
constexpr uint32_t kMissThreads = 1024;        // the walk over the log's support lists is what takes the time: 1024 threads on it
__global__ __launch_bounds__(kMissThreads)
void k_missing_cols(const uint32_t* __restrict__ log, const uint8_t* __restrict__ sub_pos, uint32_t n, uint32_t gcap,
                    int32_t* __restrict__ slot_of, uint32_t* __restrict__ slot_col, uint32_t* __restrict__ sw_list2,
                    DevState* st, uint32_t* hflags)
{
    __shared__ uint32_t s_need[kSoloWidth];
    __shared__ uint32_t s_w[kSoloWidth / 64];
    const uint32_t tid = threadIdx.x;
    const bool pos_thread = tid < kSoloWidth;                                        // threads that stand for a subset position
    if (tid < 2u * (uint32_t)kTcStride) sw_list2[tid] = 0xffffffffu;
    if (st->done || st->solo_pending != 1u || !st->subg_active) return;
    const uint32_t nlog = st->solo_nlog;
    if (pos_thread) s_need[tid] = 0u;
    __syncthreads();
    const uint32_t* entries = log + kSoloHeaderWords;
    for (uint32_t pr = tid; pr < nlog * kSoloListPitch; pr += kMissThreads) {
        const uint32_t k = pr / kSoloListPitch, j = pr - k * kSoloListPitch;
        const uint32_t* e = entries + (size_t)k * kSoloEntryWords;
        if (j < e[0]) s_need[e[8 + kSoloListPitch + j] & (kSoloWidth - 1u)] = 1u;   // subset position of a support column
        if (j == 0 && (e[1] & 1u)) {
            const uint32_t cl = e[3];                                                // the pick of a scan entry
            if (cl < n) { const uint32_t p = sub_pos[cl]; if (log[p] == cl) s_need[p] = 1u; }
        }
    }
    __syncthreads();
    const uint32_t cl = pos_thread ? log[tid] : 0xffffffffu;                         // header: column of position tid
    const bool miss = pos_thread && s_need[tid] != 0u && cl < n && slot_of[cl] < 0;
    const uint64_t bal = __ballot(miss);
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    if (pos_thread && lane == 0) s_w[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < kSoloWidth / 64; ++w) { if (w < wave) before += s_w[w]; total += s_w[w]; }
    const uint32_t used = st->cache_used;
    if (total > 64u || used + total > gcap) {
        if (tid == 0) {
            st->status = kStatusRetryPlain;
            st->need_sweep = 0;
            st->done = 1;
            signal_done(hflags, nullptr, 1u, st->iter);
        }
        return;
    }
    if (miss) {
        const uint32_t r = before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        const uint32_t sl = used + r;
        // list 0: entries 0..31 / 64..95, list 1: entries 32..63 / 96..127
        sw_list2[r] = cl; sw_list2[kTcStride + r] = sl; slot_of[cl] = (int32_t)sl;
        if (slot_col != nullptr) slot_col[sl] = cl;
    }
    if (tid == 0 && total != 0u) { st->cache_used = used + total; st->nsweeps += total > 32u ? 2u : 1u; }
}

hipError_t launch_subset_pick_f32(ss_hip_ctx* ctx, Workspace<float>& ws)
{
    if (ws.cand_top == nullptr || ws.nvwg == 0 || ws.sub_cols == nullptr) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k_subset_pick, dim3(1), dim3(kTcThreads), 0, ctx->stream, (const uint64_t*)ws.cand_top, kCandPerBlock * ws.nvwg,
                       (uint32_t)ctx->n, ws.slot_of, ws.sw_list, ws.sub_cols, ws.st, ws.slot_col,
                       (uint32_t)std::max(0, std::min(ctx->solo_subset, (int)kSoloWidth)), ctx->se_count);
    return hipGetLastError();
}

hipError_t launch_pick_pass_b_f32(ss_hip_ctx* ctx, Workspace<float>& ws, hipStream_t on)
{
    if (ws.sub_cols == nullptr || ws.sw_list == nullptr) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k_pick_pass_b, dim3(1), dim3(kSoloWidth), 0, on, (const uint32_t*)ws.sub_cols, (uint32_t)ctx->n, ws.slot_of,
                       ws.slot_col, ws.sw_list, (const DevState*)ws.st, ctx->early_adapt);
    return hipGetLastError();
}

// hold a pass's main launch back until every workgroup of the launch that goes first holds its CU (bounded, like the gate)
__global__ void k_wait_count(const uint32_t* counter, uint32_t target, const DevState* st)
{
    if (threadIdx.x != 0) return;
    for (uint32_t spin = 0; spin < 2000u; ++spin) {
        if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
        if (__hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        __builtin_amdgcn_s_sleep(8);
    }
}

hipError_t launch_wait_count(hipStream_t on, const uint32_t* counter, uint32_t target, const DevState* st)
{
    hipLaunchKernelGGL(k_wait_count, dim3(1), dim3(64), 0, on, counter, target, st);
    return hipGetLastError();
}

hipError_t launch_wait_started(ss_hip_ctx* ctx, Workspace<float>& ws, hipStream_t on)
{
    (void)ctx;
    hipLaunchKernelGGL(k_wait_started, dim3(1), dim3(64), 0, on, (const DevState*)ws.st);
    return hipGetLastError();
}

hipError_t launch_missing_cols_f32(ss_hip_ctx* ctx, Workspace<float>& ws)
{
    if (ws.solo_log == nullptr || ws.sw_list2 == nullptr) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k_missing_cols, dim3(1), dim3(kMissThreads), 0, ctx->stream, (const uint32_t*)ws.solo_log, (const uint8_t*)ws.sub_pos,
                       (uint32_t)ctx->n, ws.gcap, ws.slot_of, ws.slot_col, ws.sw_list2, ws.st, ctx->dev_flags);
    return hipGetLastError();
}

hipError_t launch_la_top_cand_f32(ss_hip_ctx* ctx, Workspace<float>& ws, uint32_t nsel)
{
    if (ws.cand_top == nullptr || ws.nvwg == 0) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k_la_top_cand, dim3(1), dim3(kTcThreads), 0, ctx->stream, (const uint64_t*)ws.cand_top, kCandPerBlock * ws.nvwg,
                       (uint32_t)ctx->n, (const uint8_t*)ws.insup, ws.slot_of, ws.gcap, ws.sw_list, ws.st, ctx->dev_flags,
                       ws.gram_is_full ? (uint32_t*)nullptr : ws.slot_col, nsel > 64u ? 64u : (nsel < 2u ? 2u : nsel));
    return hipGetLastError();
}

hipError_t launch_la_cand_init_f32(ss_hip_ctx* ctx, Workspace<float>& ws)
{
    if (ws.cand_top == nullptr || ws.nvwg == 0) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k_la_cand_init, dim3(ws.nvwg), dim3(kVfThreads), 0, ctx->stream, (const float*)ws.c0, (uint32_t)ctx->n,
                       (const DevState*)ws.st, ws.cand_top);
    return hipGetLastError();
}

// k_la_verify + k_la_vpublish behind a solo launch
hipError_t launch_la_verify_f32(ss_hip_ctx* ctx, Workspace<float>& ws)
{
    if (ws.solo_log == nullptr || ws.v_max == nullptr || ws.nvwg == 0) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k_la_verify, dim3(ws.nvwg), dim3(kVfyThreads), 0, ctx->stream, (uint32_t)ctx->n, ws.gram_is_full ? 1 : 0,
                       (const float*)ws.gcache, ws.gpitch, (const int32_t*)ws.slot_of, (const float*)ws.c0,
                       (const uint32_t*)ws.solo_log, (const uint8_t*)ws.sub_pos, (const DevState*)ws.st, ws.v_max, ws.v_min,
                       ws.nvwg, ws.tcand, ws.cand_top, ctx->tie_guard, ws.gram_is_full ? ctx->n_pad : ws.gcap, ws.la_dbg, &ws.st->tie_stall);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    CommitArgs ca;
    ca.stage = ws.solo_stage;
    ca.x = ws.x; ca.d = ws.d; ca.insup = ws.insup; ca.gam2 = ws.gam; ca.inv0 = ws.inv[0]; ca.inv1 = ws.inv[1];
    ca.kcap = ws.dims.kcap; ca.sy = ws.la_sync;
    ca.touched2 = ws.touched; ca.zero_on_removal = ctx->zero_on_removal;
    hipLaunchKernelGGL(k_la_vpublish, dim3(1), dim3(kPubThreads), 0, ctx->stream, (const uint32_t*)ws.solo_log, ws.st,
                       (const uint32_t*)ws.v_max, (const uint64_t*)ws.v_min, ws.nvwg, ctx->dev_flags, ca);
    return hipGetLastError();
}

}  // namespace sship
