// persist.hip — the lookahead engine's iterations as ONE resident launch.
//
// k_la_iter (activeset.hip) already runs a whole Homotopy iteration per launch, but every
// iteration still pays a launch, two fenced hand-offs and a chain of dependent global-memory
// round trips through the (A_S^T A_S)^-1 matrix: ~34 us for ~2 us of arithmetic.  Here the
// launch stays resident for as many iterations as it can, and every workgroup is the same:
//
//   * it owns a slice of the n columns: c_i = c0_i - sum_j x_j G[j][i], q_i = sum_j d_j G[j][i] from
//     the cached Gram columns, max |c_i|, and find_max_gamma's scan (homotopy-cpu.cpp:122-163);
//   * it keeps a full replica of the active set in LDS — sorted support, cache slots, x_S, d_S and
//     the explicit inverse — and performs the serial part of every iteration on it: loop control,
//     pick, toggle, x update, bordering / deflation of the inverse (online_inverse.h:183-293), new
//     direction (homotopy-cpu.cpp:236-272).  Same code, same inputs, same order in every
//     workgroup, hence the same bits: nothing has to be broadcast.
//
// Two exchanges per iteration remain, both all-to-all through per-workgroup slots in global
// memory: lambda = ||c||_inf (every workgroup posts its maximum and reads all of them) and the
// step length (every workgroup posts its best candidate and reads all of them).  Slots and the
// c, q values read across workgroups are moved with agent-scope atomic (L2-bypassing) loads and
// stores, so no cache write-back / invalidate fences are needed and everything else stays cached.
// A slot is set back to "empty" by its owner two exchanges later, when every reader is provably
// past it (see the loop).
//
// The launch ends when the solve terminates, when the entering column has no cached Gram column
// (the host then runs k_la_top + the lookahead sweep + k_gramupd and launches again) or when the
// support outgrows the LDS tier it was launched with.  Before A is swept for a missing column the
// loop's while-test is answered with a probe exchange (the entering column carries x = 0, so
// lambda does not need its Gram column); if the solve ends there the sweep is never made.
// State is handed over by workgroup 0 in the global-memory layout k_la_iter / k_gramupd use, so
// the three forms can follow one another.
//
// Residency: the grid is sized by the host from the occupancy of this kernel and every wait is a
// bounded spin; a workgroup that gives up returns, and the others give up within the bound.
//
// The arithmetic is the same, in the same order, as k_la_iter's (sorted support, sequential
// accumulation over j, the same reductions), so the two produce identical paths, bit for bit.
// Columns that left the support must carry exact zeros here (the lists hold the support only): with option
// zero_on_removal = 1 they always do; in reference mode (0, the default) a removal whose x + gamma*d is not
// exactly 0 ends the launch and k_la_iter, which walks every column ever touched, goes on.
// Compiled with -ffp-contract=off like activeset.hip.
//
// k_la_persist<true> is the SPECULATIVE ("solo") form, the default for fp32: ONE workgroup runs the same
// iterations on a subset of 256 columns (support, cached columns, best-ranked entrant candidates) with no
// exchange at all — c, q of the subset stay in LDS and come from one fused pass, the inverse is stored
// before the direction is formed — logs every breakpoint and STAGES the state it ends with; k_la_verify /
// k_la_vpublish (solo.hip) re-derive every logged decision over all n columns, bit for bit, before that
// state is committed.  The last step of a path (every column ties within rounding) and the rare endings
// are left to the resident form above.  10 us per iteration instead of 16.
#include "ss_hip_internal.h"
#include "ss_hip_device.h"

#include <algorithm>

namespace sship {

constexpr int kPsThreads = 512;
constexpr uint32_t kPsWidth = 256;              // columns a workgroup owns per pass (threads 0..255 carry them)
constexpr uint32_t kPsLdsBudget = 160 * 1024 - 1024;   // dynamic LDS a workgroup may take (static scratch aside)
constexpr int kPsCols = 2;                      // columns a thread owns at most
constexpr uint32_t kPsSpinLimit = 1u << 20;     // polls (~1 us each) before a wait gives up
constexpr uint32_t kPsGroup = 8;                // rows a Gram-form pass handles per step (the lists are zero-padded to whole groups)

// ---- L2-bypassing accessors for words that cross workgroups inside the launch ----------------
__device__ __forceinline__ uint32_t ld_u32(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_u64(const uint64_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_u32(uint32_t* p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_u64(uint64_t* p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_f32(const float* p)
{
    return __uint_as_float(ld_u32(reinterpret_cast<const uint32_t*>(p)));
}
__device__ __forceinline__ void st_f32(float* p, float v)
{
    st_u32(reinterpret_cast<uint32_t*>(p), __float_as_uint(v));
}
__device__ __forceinline__ void drain_vmem()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LDS carve-up (floats / 32-bit words), P = columns the launch can hold (a multiple of 16)
struct PsLds {
    float* I;        // [P][P + 1] explicit inverse, sorted-support order (master)
    uint32_t* gam;   // [P] sorted support (master)
    uint32_t* slt;   // [P] cache slot of each support column (master: state; workers: staged copy)
    float* xs;       // [P] x on the support
    float* ds;       // [P] direction on the support
    float* u1;       // [P]
    float* u2;       // [P]
    float* sg;       // [P] sign vector
    float* cn;       // [P] correlations after the step, support order
    uint32_t* lrw;   // [P] row of the LDS Gram slice that holds each support column (kNoLdsRow: none)
};
constexpr uint32_t kNoLdsRow = 0xffffffffu;
__host__ __device__ inline size_t ps_lds_words(uint32_t P) { return (size_t)P * (P + 1) + 9 * (size_t)P; }

// four independent wave sums (same order of additions as wave_sum)
__device__ __forceinline__ void wave_sum4(float (&v)[4])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = wave_sum(v[r]);
}

// In-place update of the LDS inverse I (pitch Pp) to the new sorted support; every pass reads all
// it needs before it writes (rows bottom-up for an insertion, top-down for a removal).
//   insertion: [inv + d u2 u2^T, -d u2; -d u2^T, d] with the new row/column at `rank`   (online_inverse.h:229-248)
//   removal:   inv' - u3 u3^T / d over the remaining rows/columns                         (online_inverse.h:275-290)
// The same element expressions are evaluated on the fly by new_inverse_elem() when the new
// direction is formed before this store pass has run.
__device__ __forceinline__ float new_inverse_elem(const float* I, uint32_t Pp, const float* u2, bool added,
                                                  uint32_t rank, float dv, uint32_t a, uint32_t b)
{
    if (added) {
        if (a == rank && b == rank) return dv;
        if (a == rank) return -dv * u2[b - (b > rank ? 1u : 0u)];
        if (b == rank) return -dv * u2[a - (a > rank ? 1u : 0u)];
        const uint32_t oa = a - (a > rank ? 1u : 0u), ob = b - (b > rank ? 1u : 0u);
        return I[oa * Pp + ob] + (dv * u2[oa]) * u2[ob];
    }
    const uint32_t oa = a + (a >= rank ? 1u : 0u), ob = b + (b >= rank ? 1u : 0u);
    return I[oa * Pp + ob] + (-dv * u2[oa]) * u2[ob];          // dv = the removed diagonal entry here
}

__device__ __forceinline__ void store_new_inverse(float* I, uint32_t Pp, const float* u2, bool added,
                                                  uint32_t rank, float dv, uint32_t K_new)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wv = tid >> 6;
    constexpr uint32_t NWv = kPsThreads / 64;
    const uint32_t cch = (K_new + 63u) / 64u;                 // 64-column chunks: 1..3
    const uint32_t er = cch == 1u ? 8u : (cch == 2u ? 4u : 2u);   // row slots per thread (8 values in registers)
    const uint32_t rpp = NWv * er;                            // rows per pass
    const uint32_t npass = (K_new + rpp - 1u) / rpp;
    for (uint32_t ps = 0; ps < npass; ++ps) {
        // insertion: last rows first; removal: first rows first
        const uint32_t r0 = added ? (K_new > (ps + 1u) * rpp ? K_new - (ps + 1u) * rpp : 0u) : ps * rpp;
        const uint32_t r1 = added ? K_new - ps * rpp : (r0 + rpp < K_new ? r0 + rpp : K_new);
        float v[8];
#pragma unroll
        for (uint32_t e = 0; e < 8; ++e) {
            const uint32_t ri = cch == 1u ? e : (cch == 2u ? (e >> 1) : (e >> 2));
            const uint32_t ci = cch == 1u ? 0u : (cch == 2u ? (e & 1u) : (e & 3u));
            const uint32_t a = r0 + wv + NWv * ri, b = lane + 64u * ci;
            v[e] = 0.f;
            if (ri < er && ci < cch && a < r1 && b < K_new) v[e] = new_inverse_elem(I, Pp, u2, added, rank, dv, a, b);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t e = 0; e < 8; ++e) {
            const uint32_t ri = cch == 1u ? e : (cch == 2u ? (e >> 1) : (e >> 2));
            const uint32_t ci = cch == 1u ? 0u : (cch == 2u ? (e & 1u) : (e & 3u));
            const uint32_t a = r0 + wv + NWv * ri, b = lane + 64u * ci;
            if (ri < er && ci < cch && a < r1 && b < K_new) I[a * Pp + b] = v[e];
        }
        __syncthreads();
    }
}

// one exchange: post this workgroup's word, read everybody's.  `slots` = the parity row of the
// exchange (kLaSlotStride words); returns false if a slot stayed empty for the whole bound.
// reduce(pk) is called by each thread for the slots it read.
template <typename F>
__device__ __forceinline__ bool exchange_all(uint64_t* slots, uint32_t nb, uint32_t w, uint64_t mine, F&& reduce)
{
    const uint32_t tid = threadIdx.x;
    if (tid == 0) st_u64(&slots[w], mine);
    bool ok = true;
    for (uint32_t s0 = 0; s0 < nb; s0 += kPsThreads) {
        const uint32_t sidx = s0 + tid;
        if (sidx < nb) {
            uint64_t pk = kLaSlotEmpty;
            for (uint32_t spin = 0; spin < kPsSpinLimit; ++spin) {
                pk = ld_u64(&slots[sidx]);
                if (pk != kLaSlotEmpty) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (pk == kLaSlotEmpty) ok = false;
            else reduce(pk);
        }
    }
    return __syncthreads_or(ok ? 0 : 1) == 0;
}

// SOLO = the speculative form: ONE workgroup runs the same iterations on a subset of kSoloWidth columns
// (the support, the cached columns and the best-ranked entrant candidates) with no exchange at all;
// lambda and the step length come from the subset alone.  Every breakpoint is logged (lists, lambda,
// pick) and k_la_verify recomputes the decisions over ALL columns afterwards, in the same arithmetic
// order; the outcome is staged in DevState and reaches the host only after that check (solo.hip).
struct SoloArgs {
    const uint32_t* slot_col;   // [cache_used] column of each cache slot
    const uint64_t* cand_top;   // [ncand] (ordered key << 32 | column) entrant candidates, ~0 = none
    uint32_t ncand;
    uint32_t subset_cap;        // columns the launch may hold beyond the support (option solo_subset)
    uint32_t* log;              // header + [kSoloLogCap][kSoloEntryWords] (ss_hip_internal.h)
    uint8_t* sub_pos;           // [n_pad] subset position of each of this launch's columns
    uint32_t* stage;            // [kSoloStageWords] staged hand-over (committed by k_la_vpublish)
    // early form (DevState::subg_active): the launch runs on the subset Gram matrix Gs = A_S^T A_S of the given
    // columns (subgram.hip) instead of cache rows — Gram "row" and "column" are subset positions there
    const float* subg;          // [kSoloWidth][kSoloWidth], null = never
    const uint32_t* sub_cols;   // [kSoloWidth] the columns of that subset (position 0 = the first pick)
    float* prog;                // [kSoloWidth] progress for k_pick_pass_b: per subset position, -1 = in the support,
                                // else this breakpoint's step-length candidate (how soon the column would enter)
};

template <bool SOLO>
__global__ __launch_bounds__(kPsThreads)
void k_la_persist(float tol, uint32_t max_iter, uint32_t n, uint32_t P, uint32_t gl_rows, int full_g,
                  const float* __restrict__ gcache, const int32_t* __restrict__ slot_of,
                  const float* __restrict__ c0, uint32_t gpitch,
                  float* c, float* q, float* c_alt, float* q_alt, float* x, float* d, uint8_t* insup,
                  uint32_t* gam2, float* inv0, float* inv1, float* __restrict__ tcand,
                  SlotDims L, DevState* st, LaSync* sy, uint64_t* smax, uint64_t* smin, uint32_t* hflags,
                  TraceEntry* trace, uint32_t trace_cap, int tie_guard, int zero_on_removal, uint32_t* touched2,
                  int after_solo, uint64_t* dbg, SoloArgs sa)
{
    extern __shared__ float smem[];
    __shared__ float sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_cnt[2];
    __shared__ float s_dd;
    __shared__ uint32_t s_sub[SOLO ? kSoloWidth : 1];          // solo: the columns of this launch
    __shared__ uint64_t s_off[SOLO ? kPsThreads : 1];          // solo: candidate offers while the subset is built
    __shared__ uint32_t s_sps[SOLO ? kSoloListPitch : 1];      // solo: subset position of each support column
    __shared__ uint32_t s_ipos;                                // solo: subset position of the entering column
    __shared__ float s_sgv[SOLO ? kPsThreads / 64 : 1];        // solo: per-wave best step-length candidate on the support ...
    __shared__ uint32_t s_sgi[SOLO ? kPsThreads / 64 : 1];     // ... and its column
    __shared__ float s_c0[SOLO ? kSoloWidth : 1];              // solo: c0 of the subset's columns
    __shared__ float s_cq[SOLO ? 2 * kSoloWidth : 1];          // solo: c, q of the subset (the resident form hands them through global memory)

    const uint32_t tid = threadIdx.x;
    const uint32_t nb = SOLO ? 1u : gridDim.x, w = SOLO ? 0u : blockIdx.x;
    const bool lead = w == 0;                    // the workgroup that writes the shared state
    const uint32_t kcap = L.kcap;
    const uint32_t Pp = P + 1u;
    PsLds S;
    S.I = smem;
    S.gam = reinterpret_cast<uint32_t*>(smem + (size_t)P * Pp);
    S.slt = S.gam + P;
    S.xs = reinterpret_cast<float*>(S.slt + P);
    S.ds = S.xs + P;
    S.u1 = S.ds + P;
    S.u2 = S.u1 + P;
    S.sg = S.u2 + P;
    S.cn = S.sg + P;
    // the workgroup's slice of the first gl_rows cache rows: [gl_rows][kPsWidth]
    S.lrw = reinterpret_cast<uint32_t*>(S.cn + P);
    float* const Glds = reinterpret_cast<float*>(S.lrw + P);

    // ---- nothing to do in this launch? (same answer in every workgroup: DevState was written
    // ---- by earlier launches only) --------------------------------------------------------------
    const uint32_t K0 = st->K;
    const bool grow0 = (K0 + 1u > P) && (P < kcap);
    if (SOLO) {
        // early form: the passes over A on the second stream are held back until this workgroup is resident
        if (tid == 0) {
            // where this workgroup runs, for the passes that share the chip with it (they keep off its shader engine)
            uint32_t hw_, xcc_;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
            __hip_atomic_store(&st->solo_where, (((xcc_ & 0xfu) << 16) | (hw_ & 0xffffu)) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&st->solo_started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (k_la_vpublish counts the launch for the host pump, whether it worked or not)
        if (st->done || st->need_sweep || st->solo_off) return;
        if (grow0) {
            // the support does not fit the solo tier: the resident form takes over (nothing staged: exit code 0)
            if (tid == 0) { sa.stage[9] = kPsExitNothing; sa.stage[0] = K0; st->solo_nlog = 0; st->solo_pending = 2; }
            return;
        }
    } else if (after_solo && !st->solo_off) {
        // queued right behind a speculative group so that its hand-over (the last step of a path) needs no trip
        // through the host: nothing to do unless that group has handed over
        if (lead && tid == 0) bump_seq(st, hflags);
        return;
    } else if (st->done || st->need_sweep || grow0) {
        if (lead && tid == 0) {
            if (!st->done && !st->need_sweep)
                __hip_atomic_store(&hflags[3], K0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            bump_seq(st, hflags);
        }
        return;
    }
    uint32_t tick = sy->tick;
    // early form: Gram values come from Gs, indexed by subset positions (rows are handed out in LDS as columns
    // enter, like the full-G mode does it)
    const bool subg = SOLO && sa.subg != nullptr && st->subg_active != 0u;
    const bool by_entry = full_g || subg;                      // LDS rows follow the order of entry, not the cache slots

    // ---- this workgroup's columns -------------------------------------------------------------------
    const uint32_t replay = SOLO ? st->solo_replay : 0u;       // > 0: repeat exactly this many (verified) iterations
    if (SOLO && subg && replay == 0u) {
        if (tid < kSoloWidth) s_sub[tid] = sa.sub_cols[tid];     // chosen before the launch (k_subset_pick)
        __syncthreads();
    } else if (SOLO && replay != 0u) {
        // the same columns as the launch that is being repeated (its header is still in the log)
        if (tid < kSoloWidth) s_sub[tid] = sa.log[tid];
        __syncthreads();
    } else if (SOLO) {
        // the subset: support, then cached columns outside it (latest slots first), then the best-ranked
        // entrant candidates of the last scan (k_la_verify) or of |c0| (k_la_cand_init)
        // (subset_cap counts the columns beyond the support; the support, K0 <= P < kSoloWidth, is always in)
        const uint32_t cap = K0 + sa.subset_cap < kSoloWidth ? K0 + sa.subset_cap : kSoloWidth;
        const uint32_t* gam0 = gam2 + (size_t)st->cur * kcap;
        if (tid < 2) s_cnt[tid] = 0u;
        if (tid < kSoloWidth) s_sub[tid] = 0xffffffffu;
        __syncthreads();
        if (tid < K0) s_sub[tid] = gam0[tid];
        uint32_t cnt = K0;
        if (!full_g) {
            const uint32_t used = st->cache_used;
            const uint32_t quota = cap > cnt + 32u ? cap - cnt - 32u : 0u;       // room kept for candidates
            const uint32_t s0 = used > kPsThreads ? used - kPsThreads : 0u;
            // thread t looks at slot used-1-t; positions by an exclusive prefix count over the threads
            // (ballot per wave, wave totals through LDS): the same subset on every run
            uint32_t cl = 0xffffffffu;
            if (s0 + tid < used) cl = sa.slot_col[used - 1u - tid];
            const bool take = cl < n && !insup[cl];
            const uint64_t bal = __ballot(take);
            const uint32_t lane_ = tid & 63u, wave_ = tid >> 6;
            if (lane_ == 0) si[wave_] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (uint32_t w2 = 0; w2 < kPsThreads / 64; ++w2) { const uint32_t c2 = si[w2]; if (w2 < wave_) before += c2; total += c2; }
            const uint32_t pos = before + (uint32_t)__popcll(bal & ((1ull << lane_) - 1ull));
            if (take && pos < quota) s_sub[cnt + pos] = cl;
            __syncthreads();
            cnt += total < quota ? total : quota;
        }
        // candidates: every thread offers the best of its strided share, the offers are ranked by counting
        uint64_t offer = ~0ull;
        if (n <= 32u * kPsThreads) {
            // few columns: the workgroup ranks all of them itself (the per-block tops are too few here)
            const bool scanned = st->cand_scan != 0;              // tcand of the last verified scan, else |c0|
            for (uint32_t cl = tid; cl < n; cl += kPsThreads) {
                if (insup[cl] || (!full_g && slot_of[cl] >= 0)) continue;
                float key;
                if (scanned) { key = tcand[cl]; if (!(key < Lim<float>::max())) continue; }
                else { const float v = c0[cl]; key = v < 0.f ? v : -v; }
                const uint64_t pk = ((uint64_t)ordered_key(key) << 32) | cl;
                if (pk < offer) offer = pk;
            }
        } else {
            for (uint32_t e = tid; e < sa.ncand; e += kPsThreads) {
                const uint64_t pk = sa.cand_top[e];
                const uint32_t cl = (uint32_t)pk;
                if (pk == ~0ull || cl >= n || insup[cl] || (!full_g && slot_of[cl] >= 0)) continue;
                if (pk < offer) offer = pk;
            }
        }
        s_off[tid] = offer;
        __syncthreads();
        if (offer != ~0ull) {
            uint32_t rank = 0;
            for (uint32_t u = 0; u < kPsThreads; ++u) rank += s_off[u] < offer ? 1u : 0u;
            if (cnt + rank < cap) s_sub[cnt + rank] = (uint32_t)offer;
        }
        __syncthreads();
    }
    uint32_t col[kPsCols];
    bool in[kPsCols], cached[kPsCols];
    uint32_t act[kPsCols];
    float c0v[kPsCols];
#pragma unroll
    for (int k = 0; k < kPsCols; ++k) {
        col[k] = w * kPsWidth + (tid & (kPsWidth - 1u)) + (uint32_t)k * nb * kPsWidth;
        in[k] = tid < kPsWidth && col[k] < n;
        if (SOLO) {
            col[k] = (k == 0 && tid < kPsWidth) ? s_sub[tid] : 0xffffffffu;
            in[k] = col[k] < n;
        }
        c0v[k] = 0.f; act[k] = 0; cached[k] = false;
        if (in[k]) {
            c0v[k] = c0[col[k]];
            act[k] = insup[col[k]];
            cached[k] = slot_of[col[k]] >= 0;
        }
    }

    if (SOLO) {
        if (tid < kSoloWidth) {
            const uint32_t cl = s_sub[tid];
            sa.log[tid] = cl;
            sa.log[kSoloWidth + tid] = cl < n ? (full_g ? cl : (uint32_t)slot_of[cl]) : 0xffffffffu;   // (informational: k_la_verify looks the rows up itself)
            if (cl < n) sa.sub_pos[cl] = (uint8_t)tid;
        }
        if (tid < kSoloListPitch) s_sps[tid] = tid;             // the support leads the subset, in order
        if (tid < kSoloWidth) s_c0[tid] = s_sub[tid] < n ? c0[s_sub[tid]] : 0.f;
    }
    // ---- replica of the active set ---------------------------------------------------------------------
    const uint32_t cur = st->cur;
    float* const Ig = cur ? inv1 : inv0;
    uint32_t* const gam_cur = gam2 + (size_t)cur * kcap;
    uint32_t* const gam_alt = gam2 + (size_t)(cur ^ 1u) * kcap;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NW = kPsThreads / 64;

    uint32_t K = K0;
    for (uint32_t e = tid; e < K * K; e += kPsThreads) {
        const uint32_t a = e / K, b = e - a * K;
        S.I[a * Pp + b] = Ig[(size_t)a * kcap + b];
    }
    if (tid < K) {
        const uint32_t cl = gam_cur[tid];
        S.gam[tid] = cl;
        uint32_t sl = (uint32_t)slot_of[cl];
        if (subg) {                                  // the Gram row of a column is its subset position
            sl = 0u;
            for (uint32_t p2 = 0; p2 < kSoloWidth; ++p2) if (s_sub[p2] == cl) sl = p2;
        }
        S.slt[tid] = sl;
        S.xs[tid] = x[cl];
        S.ds[tid] = d[cl];
    } else if (tid < P) {
        S.xs[tid] = 0.f;                             // padding of the Gram-form loops (whole groups of kPsGroup rows)
        S.ds[tid] = 0.f;
    }
    __syncthreads();
    // the Gram-column cache as a buffer: row offsets go in the scalar offset of the loads (cache mode;
    // with the full Gram matrix as "cache" rows lie up to n * pitch * 4 B apart: 64-bit addressing there)
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gcache), 0, -1, 0x00020000);
    // (every lambda below is forced inline: a closure that survives keeps its captures — K, the list
    // pointers, ... — in scratch memory, which made each stage of the solo instantiation ~1.5x slower)
    auto grow_global = [&](uint32_t row, uint32_t cofs4) __attribute__((always_inline)) -> float {
        if (subg) return sa.subg[(size_t)row * kSoloWidth + (cofs4 >> 2)];       // (row, column: subset positions)
        if (full_g) return gcache[(size_t)row * gpitch + (cofs4 >> 2)];
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grsrc, cofs4, row * (gpitch * 4u), 0));
    };
    // This workgroup's slice of Gram rows, kept in LDS for the whole launch: the per-CU limit on
    // outstanding L1 misses makes K row reads from L2 cost ~40 ns each, every iteration.
    //   cache mode : LDS row r = cache slot r, for the first gl_rows slots handed out;
    //   full-G mode: LDS rows are handed out as columns enter (the support at entry takes rows 0..K0-1),
    //                each workgroup fetching its 1 KiB of row idx of G at that moment.
    const uint32_t gl_used = by_entry ? (K0 < gl_rows ? K0 : gl_rows) : (st->cache_used < gl_rows ? st->cache_used : gl_rows);
    uint32_t lds_rows_used = gl_used;                  // full-G mode: next free LDS row (same in every workgroup)
    if (tid < P) {
        uint32_t lr = kNoLdsRow;
        if (tid < K0) lr = by_entry ? (tid < gl_rows ? tid : kNoLdsRow) : (S.slt[tid] < gl_used ? S.slt[tid] : kNoLdsRow);
        S.lrw[tid] = lr;
    }
    {
        const uint32_t tcol = tid & (kPsWidth - 1u), half = tid / kPsWidth;          // two rows per pass
        const uint32_t cg = SOLO ? s_sub[tcol] : w * kPsWidth + tcol;
        const uint32_t cofs4 = subg ? tcol * 4u : (cg < n ? cg : 0u) * 4u;
        constexpr uint32_t rpp = kPsThreads / kPsWidth;      // rows per pass
        for (uint32_t r0 = 0; r0 < gl_used; r0 += 8 * rpp) {
            float gv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t r = r0 + rpp * t + half;
                gv[t] = 0.f;
                if (r < gl_used) gv[t] = grow_global(by_entry ? S.slt[r] : r, cofs4);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t r = r0 + rpp * t + half;
                if (r < gl_used) Glds[r * kPsWidth + tcol] = gv[t];
            }
        }
    }
    __syncthreads();
    if (dbg != nullptr && lead && tid == 0) { dbg[0] = gl_used; dbg[1] = nb; dbg[2] = gl_rows; dbg[3] = K0; }

    uint32_t iter = st->iter;
    float c_inf_rep = (float)st->c_inf;      // what the report will carry
    float gamma_last = (float)st->gamma;
    const float lambda0 = st->lambda0;       // scale of the tie band (ss_hip_device.h)
    uint32_t last_idx = st->idx, last_rank = st->rank, last_added = st->added;
    uint32_t done_round = 0u, status = 0u;
    // why the launch ends (kPsExit*, ss_hip_internal.h)
    uint32_t exit_code = kPsExitNone;
    bool report_empty = false;               // the support became empty (DevState::K = 0)
    bool save_lists_for_update = false;      // exit 2: the pick is made, the inverse update is pending
    uint32_t pend_rank = 0, pend_idx = 0;
    bool pend = false, pend_added = false;   // the LDS inverse still has to take the last toggle
    uint32_t pend_rk = 0, pend_K = 0;
    float pend_dv = 0.f;
    uint64_t ts[8];

    // sum_j coef[j] * G[slt[j]][col] over the first Kc list entries, for this thread's columns.
    // Each lane fetches one cache slot of the lists; the loop takes them out of the lanes as scalars.
    // Rows below gl_used come from the LDS slice (first column set), the others by buffer loads
    // whose row offset is a scalar register.  Entries Kc .. roundup16(Kc)-1 are padding (coef = 0
    // on a valid row: exact zeros), so the loop runs in whole groups of kPsGroup rows without branches.
    auto gram_pass = [&](const float* coef, uint32_t Kc, float (&out)[kPsCols]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < kPsCols; ++k) {
            out[k] = 0.f;
            if (__ballot(in[k]) == 0ull) continue;            // whole waves only (lanes exchange entries)
            const uint32_t cofs4 = (in[k] ? col[k] : 0u) * 4u;
            const uint32_t K16 = (Kc + (kPsGroup - 1u)) & ~(kPsGroup - 1u);
            const uint32_t tcol = tid & (kPsWidth - 1u);
            float acc = 0.f;
            for (uint32_t j0 = 0; j0 < K16; j0 += 64) {
                const uint32_t jl = j0 + (uint32_t)lane;
                const uint32_t vs = S.slt[jl < Kc ? jl : 0u];                          // global row
                const uint32_t vl = k == 0 ? S.lrw[jl < Kc ? jl : 0u] : kNoLdsRow;     // LDS row (first column set)
                const uint32_t cnt = K16 - j0 < 64u ? K16 - j0 : 64u;
                const float* cf = coef + j0;
                const uint64_t out_of_lds = __ballot(vl == kNoLdsRow);    // lanes whose row must come from L2
                for (uint32_t u = 0; u < cnt; u += kPsGroup) {
                    float gv[kPsGroup];
                    if (((out_of_lds >> u) & ((1ull << kPsGroup) - 1ull)) == 0ull) {
#pragma unroll
                        for (int t = 0; t < (int)kPsGroup; ++t) gv[t] = Glds[__builtin_amdgcn_readlane(vl, u + t) * kPsWidth + tcol];
                    } else {
#pragma unroll
                        for (int t = 0; t < (int)kPsGroup; ++t) {
                            const uint32_t lr = __builtin_amdgcn_readlane(vl, u + t);
                            gv[t] = 0.f;
                            if (lr == kNoLdsRow) gv[t] = grow_global(__builtin_amdgcn_readlane(vs, u + t), cofs4);
                        }
#pragma unroll
                        for (int t = 0; t < (int)kPsGroup; ++t) {
                            const uint32_t lr = __builtin_amdgcn_readlane(vl, u + t);
                            if (lr != kNoLdsRow) gv[t] = Glds[lr * kPsWidth + tcol];
                        }
                    }
#pragma unroll
                    for (int t = 0; t < (int)kPsGroup; ++t) acc += cf[u + t] * gv[t];
                }
            }
            out[k] = acc;
        }
    };

    float cv[kPsCols], qv[kPsCols];
    // c = c0 - sum_j x_j g_j over the lists as they stand (entries with x = 0 add exact zeros), the
    // partial maximum of |c|, and this workgroup's word of the lambda exchange of a new tick
    float cmax_v = -1.f;
    uint32_t cmax_i = 0xffffffffu;
    auto c_pass = [&]() __attribute__((always_inline)) {
        float ax[kPsCols];
        gram_pass(S.xs, K, ax);
        float bv = -1.f;
        uint32_t bi = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < kPsCols; ++k) {
            cv[k] = 0.f;
            if (!in[k]) continue;
            cv[k] = c0v[k] - ax[k];
            const float a = cv[k] < 0.f ? -cv[k] : cv[k];
            if (better_max(a, col[k], bv, bi)) { bv = a; bi = col[k]; }
        }
        block_reduce_pair<float, true>(bv, bi, sv, si);
        cmax_v = bv;
        cmax_i = bi;
    };
    // solo: c and q in ONE pass over the rows (no exchange has to be posted in between, and each Gram
    // value is read once).  Per column the two sums are formed exactly as by the separate passes.
    auto cq_pass = [&]() __attribute__((always_inline)) {
        // all eight waves: wave w, lanes 0..31 form sum_j x_j g_j and lanes 32..63 sum_j d_j g_j of the 32
        // columns 32w .. 32w+31 of the subset (one sum per lane; a packed two-column variant measured the same:
        // the pass is bound by the latency of its LDS reads, not by instruction issue)
        {
            const uint32_t t = (uint32_t)wave * 32u + ((uint32_t)lane & 31u);       // subset position
            const bool dsum = lane >= 32;
            const uint32_t ccol = s_sub[t];
            const uint32_t cofs4 = subg ? t * 4u : (ccol < n ? ccol : 0u) * 4u;
            const float* coef = dsum ? S.ds : S.xs;
            const uint32_t K16 = (K + (kPsGroup - 1u)) & ~(kPsGroup - 1u);
            float acc = 0.f;
            for (uint32_t j0 = 0; j0 < K16; j0 += 64) {
                const uint32_t jl = j0 + (uint32_t)lane;
                const uint32_t vs = S.slt[jl < K ? jl : 0u];
                const uint32_t vl = S.lrw[jl < K ? jl : 0u];
                const uint32_t cnt = K16 - j0 < 64u ? K16 - j0 : 64u;
                const float* cf = coef + j0;
                const uint64_t out_of_lds = __ballot(vl == kNoLdsRow);
                for (uint32_t u = 0; u < cnt; u += kPsGroup) {
                    float gv[kPsGroup];
                    if (((out_of_lds >> u) & ((1ull << kPsGroup) - 1ull)) == 0ull) {
#pragma unroll
                        for (int q2 = 0; q2 < (int)kPsGroup; ++q2) gv[q2] = Glds[__builtin_amdgcn_readlane(vl, u + q2) * kPsWidth + t];
                    } else {
#pragma unroll
                        for (int q2 = 0; q2 < (int)kPsGroup; ++q2) {
                            const uint32_t lr = __builtin_amdgcn_readlane(vl, u + q2);
                            gv[q2] = 0.f;
                            if (lr == kNoLdsRow) gv[q2] = grow_global(__builtin_amdgcn_readlane(vs, u + q2), cofs4);
                        }
#pragma unroll
                        for (int q2 = 0; q2 < (int)kPsGroup; ++q2) {
                            const uint32_t lr = __builtin_amdgcn_readlane(vl, u + q2);
                            if (lr != kNoLdsRow) gv[q2] = Glds[lr * kPsWidth + t];
                        }
                    }
#pragma unroll
                    for (int q2 = 0; q2 < (int)kPsGroup; ++q2) acc += cf[u + q2] * gv[q2];
                }
            }
            s_cq[(dsum ? kSoloWidth : 0u) + t] = dsum ? acc : s_c0[t] - acc;
        }
        __syncthreads();
        float bv = -1.f;
        uint32_t bi = 0xffffffffu;
        cv[0] = 0.f; qv[0] = 0.f; cv[1] = 0.f; qv[1] = 0.f;
        if (in[0]) {
            cv[0] = s_cq[tid];
            qv[0] = s_cq[kSoloWidth + tid];
            const float a = cv[0] < 0.f ? -cv[0] : cv[0];
            if (better_max(a, col[0], bv, bi)) { bv = a; bi = col[0]; }
        }
        block_reduce_pair<float, true>(bv, bi, sv, si);
        cmax_v = bv;
        cmax_i = bi;
    };
    auto post_lambda = [&]() __attribute__((always_inline)) {
        ++tick;
        if (!SOLO && tid == 0)
            st_u64(&smax[(tick & 1u) * kLaSlotStride + w],
                   cmax_i != 0xffffffffu ? (((uint64_t)__float_as_uint(cmax_v) << 32) | (uint64_t)(0xffffffffu - cmax_i)) : 0ull);
    };
    auto c_pass_and_post = [&]() __attribute__((always_inline)) { c_pass(); post_lambda(); };
    // read everybody's word of the lambda exchange of the current tick (false: a wait expired)
    // `early`: what this thread read from slot `tid` before the q pass (usually the word is there by then)
    auto poll_lambda = [&](float& lam, uint64_t early = kLaSlotEmpty) __attribute__((always_inline)) -> bool {
        if (SOLO) { lam = cmax_v; return true; }             // the subset's maximum (uniform after the block reduction)
        const uint32_t par = (tick & 1u) * kLaSlotStride;
        float mv = -1.f;
        uint32_t mi = 0xffffffffu;
        bool ok = true;
        for (uint32_t s0 = 0; s0 < nb; s0 += kPsThreads) {
            const uint32_t sidx = s0 + tid;
            if (sidx < nb) {
                uint64_t pk = s0 == 0 ? early : kLaSlotEmpty;
                for (uint32_t spin = 0; pk == kLaSlotEmpty && spin < kPsSpinLimit; ++spin) {
                    pk = ld_u64(&smax[par + sidx]);
                    if (pk != kLaSlotEmpty) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (pk == kLaSlotEmpty) ok = false;
                else if (pk != 0ull) {
                    const float v = __uint_as_float((uint32_t)(pk >> 32));
                    const uint32_t i2 = 0xffffffffu - (uint32_t)pk;
                    if (better_max(v, i2, mv, mi)) { mv = v; mi = i2; }
                }
            }
        }
        if (__syncthreads_or(ok ? 0 : 1)) return false;
        block_reduce_pair<float, true>(mv, mi, sv, si);
        lam = mv;
        // every workgroup is past the previous step-length exchange (it posted a maximum after it):
        // this workgroup's slot of that exchange can be cleared for its next use
        if (tid == 0) st_u64(&smin[((tick + 1u) & 1u) * kLaSlotStride + w], kLaSlotEmpty);
        return true;
    };

    // solo: one log entry per decision taken from the subset alone (lists as they stand: K entries)
    uint32_t nlog = 0, nscan = 0;
    auto log_entry = [&](uint32_t flags, uint32_t round_, float lam, float g_, uint32_t idx_, float gs_, uint32_t is_) __attribute__((always_inline)) {
        if (!SOLO) return;
        uint32_t* e = sa.log + kSoloHeaderWords + (size_t)nlog * kSoloEntryWords;
        if (tid == 0) {
            e[0] = K; e[1] = flags; e[2] = round_; e[3] = idx_;
            e[4] = __float_as_uint(lam); e[5] = __float_as_uint(g_); e[6] = __float_as_uint(gs_); e[7] = is_;
        }
        if (tid < K) {
            e[8 + tid] = S.gam[tid];
            e[8 + kSoloListPitch + tid] = s_sps[tid];
            e[8 + 2 * kSoloListPitch + tid] = __float_as_uint(S.xs[tid]);
            e[8 + 3 * kSoloListPitch + tid] = __float_as_uint(S.ds[tid]);
        }
        ++nlog;
        nscan += flags & 1u;
    };

    // Software pipeline.  The lambda exchange of an iteration is posted as soon as its x is known
    // (right after the previous pick) and read only after the inverse update and the q pass, which
    // hide its latency; the step-length exchange hides the in-place store pass of the inverse.
    if (K + 1u > P && P < kcap) {
        exit_code = kPsExitGrow;                               // (also caught at entry; kept for clarity)
    } else if (SOLO) {
        cq_pass();
        post_lambda();
    } else {
        c_pass_and_post();
    }
    while (exit_code == kPsExitNone) {
        const uint32_t round = iter + 1u;
        const uint32_t par = (tick & 1u) * kLaSlotStride;
        ts[0] = wall_clock64();
        if (SOLO && (nlog + 2u > kSoloLogCap || (replay != 0u && nscan >= replay))) {
            // the log is full: end the launch at this iteration boundary (the next one goes on)
            if (pend) { store_new_inverse(S.I, Pp, S.u2, pend_added, pend_rk, pend_dv, pend_K); pend = false; }
            exit_code = kPsExitLogFull;
            break;
        }

        // ---- q = sum_j d_j g_j; publish c, q for the reads on the support -----------------------------
        // (buffer of this tick's parity: a workgroup may already be writing the next tick's values
        // while a slower one still reads this tick's)
        float* const cbuf = (tick & 1u) ? c_alt : c;
        float* const qbuf = (tick & 1u) ? q_alt : q;
        // (the lambda words were posted before the inverse update: read this thread's slot now, use it below)
        const uint64_t early = (!SOLO && tid < nb) ? ld_u64(&smax[par + tid]) : kLaSlotEmpty;
        if (!SOLO) gram_pass(S.ds, K, qv);                    // (solo: q came with c, in cq_pass)
        if (SOLO) {
            // (scattered columns: 512 single-line stores per iteration would hold up every later barrier)
            if (tid < kSoloWidth) { s_cq[tid] = cv[0]; s_cq[kSoloWidth + tid] = qv[0]; }
        } else {
#pragma unroll
            for (int k = 0; k < kPsCols; ++k)
                if (in[k]) { st_f32(&cbuf[col[k]], cv[k]); st_f32(&qbuf[col[k]], qv[k]); }
        }
        ts[1] = wall_clock64();

        // ---- lambda (posted before the inverse update) ---------------------------------------------------
        float c_inf;
        if (!poll_lambda(c_inf, early)) { exit_code = kPsExitWait; break; }
        ts[2] = wall_clock64();

        // do { ... } while (iter < max_iter && c_inf > tolerance)   (homotopy-cpu.cpp:236,272)
        if ((round > 1u && !(c_inf > tol)) || round > max_iter) {
            if (pend) { store_new_inverse(S.I, Pp, S.u2, pend_added, pend_rk, pend_dv, pend_K); pend = false; }
            log_entry(0u, round, c_inf, 0.f, 0u, 0.f, 0u);         // the lambda that ended the loop
            c_inf_rep = c_inf;
            iter = round - 1u;
            done_round = round;
            exit_code = kPsExitDone;
            break;
        }

        // ---- step-length candidates: own inactive columns (same expressions as k_scansel) ... ----------
        float best = Lim<float>::max();
        uint32_t best_i = 0xffffffffu;
        float mk[kPsCols];
#pragma unroll
        for (int k = 0; k < kPsCols; ++k) {
            mk[k] = Lim<float>::max();
            if (!in[k]) continue;
            float m = Lim<float>::max();
            if (!act[k]) {
                const float qi = qv[k], ci = cv[k];
                const float dl = 1.f - qi, dr = 1.f + qi;
                if (dl != 0.f) {
                    float t = (c_inf - ci) / dl;
                    if (tie_guard && t == 0.f && dl > 0.f) t = Lim<float>::tiny();
                    // (see DevState::tie_stall; the column that left the support in the previous iteration sits on the boundary by rule)
                    if (t == 0.f && !(last_added == 0u && col[k] == last_idx) && tie_band<float>(c_inf, c_inf_rep, gamma_last, lambda0)) __hip_atomic_store(&st->tie_stall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t > 0.f && t < m) m = t;
                }
                if (dr != 0.f) {
                    float t = (c_inf + ci) / dr;
                    if (tie_guard && t == 0.f && dr > 0.f) t = Lim<float>::tiny();
                    if (t == 0.f && !(last_added == 0u && col[k] == last_idx) && tie_band<float>(c_inf, c_inf_rep, gamma_last, lambda0)) __hip_atomic_store(&st->tie_stall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t > 0.f && t < m) m = t;
                }
            }
            mk[k] = m;
            if (m < Lim<float>::max() && better_min(m, col[k], best, best_i)) { best = m; best_i = col[k]; }
        }
        if (SOLO && subg && sa.prog != nullptr && tid < kSoloWidth && in[0])
            // early form: the second pass over A picks its columns while this launch runs (k_pick_pass_b, another
            // stream): columns that have entered, then those closest to entering.  A hint, not a hand-over: plain
            // write-through stores, no ordering, nothing in this launch depends on them
            __hip_atomic_store(&sa.prog[tid], act[0] ? -1.f : mk[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (SOLO) {
            // the support's own candidates, -x_j / d_j (homotopy-cpu.cpp:128-135): reduced per wave here, the
            // per-wave results ride on the barriers of the block reduction below
            float gw = Lim<float>::max();
            uint32_t iw = 0xffffffffu;
            if (tid < K) {
                const float t = -S.xs[tid] / S.ds[tid];
                if (t > 0.f && t < Lim<float>::max()) { gw = t; iw = S.gam[tid]; }
            }
            wave_reduce_pair<float, false>(gw, iw);
            if (lane == 0) { s_sgv[wave] = gw; s_sgi[wave] = iw; }
        } else {
            drain_vmem();                                  // this wave's c, q stores are performed
        }
        block_reduce_pair<float, false>(best, best_i, sv, si);
        if (!SOLO && tid == 0)
            st_u64(&smin[par + w], best_i != 0xffffffffu ? (((uint64_t)__float_as_uint(best) << 32) | (uint64_t)best_i) : kLaSlotNone);
        // per-column candidates for k_la_top's ranking: nobody reads them inside the launch
        // (solo: k_la_verify writes them for all columns)
#pragma unroll
        for (int k = 0; k < kPsCols; ++k)
            if (!SOLO && in[k]) tcand[col[k]] = (act[k] || cached[k]) ? Lim<float>::max() : mk[k];
        // while the candidates travel: bring the stored inverse up to date with the last toggle
        if (pend) { store_new_inverse(S.I, Pp, S.u2, pend_added, pend_rk, pend_dv, pend_K); pend = false; }
        // ... and the active columns, -x_j / d_j (homotopy-cpu.cpp:128-135), from the replica
        float g = Lim<float>::max();
        uint32_t idx = 0xffffffffu;
        if (!SOLO && tid < K) {
            const float t = -S.xs[tid] / S.ds[tid];
            if (t > 0.f && t < Lim<float>::max()) { g = t; idx = S.gam[tid]; }
        }
        float gs_log = Lim<float>::max();
        uint32_t is_log = 0xffffffffu;
        if (SOLO) {
            // support candidates alone (logged: the verification merges them with ITS candidates), then
            // merged with the subset's best
#pragma unroll
            for (int w2 = 0; w2 < kPsThreads / 64; ++w2)
                if (better_min(s_sgv[w2], s_sgi[w2], g, idx)) { g = s_sgv[w2]; idx = s_sgi[w2]; }
            gs_log = g; is_log = idx;
            if (best_i != 0xffffffffu && better_min(best, best_i, g, idx)) { g = best; idx = best_i; }
        } else {
            bool ok = true;
            for (uint32_t s0 = 0; s0 < nb; s0 += kPsThreads) {
                const uint32_t sidx = s0 + tid;
                if (sidx < nb) {
                    uint64_t pk = kLaSlotEmpty;
                    for (uint32_t spin = 0; spin < kPsSpinLimit; ++spin) {
                        pk = ld_u64(&smin[par + sidx]);
                        if (pk != kLaSlotEmpty) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (pk == kLaSlotEmpty) ok = false;
                    else if (pk != kLaSlotNone) {
                        const float v = __uint_as_float((uint32_t)(pk >> 32));
                        const uint32_t i2 = (uint32_t)pk;
                        if (better_min(v, i2, g, idx)) { g = v; idx = i2; }
                    }
                }
            }
            if (__syncthreads_or(ok ? 0 : 1)) { exit_code = kPsExitWait; break; }
            // final (gamma, idx): smallest positive candidate, left-most index (:123-124)
            block_reduce_pair<float, false>(g, idx, sv, si);
        }
        // every workgroup has read all maxima of this tick (it posted its candidate after that)
        if (!SOLO && tid == 0) st_u64(&smax[par + w], kLaSlotEmpty);
        const bool no_candidate = !(g < Lim<float>::max());
        if (no_candidate) idx = 0u;
        ts[3] = wall_clock64();

        // rank of idx in the sorted support, membership (rank_index.h:65-83)
        if (tid < 2) s_cnt[tid] = 0u;
        if (SOLO && in[0] && col[0] == idx) s_ipos = tid;          // subset position of the pick
        __syncthreads();
        if (tid < K) {
            const uint32_t gj = S.gam[tid];
            if (gj < idx) atomicAdd(&s_cnt[0], 1u);
            if (gj == idx) atomicAdd(&s_cnt[1], 1u);
        }
        __syncthreads();
        const uint32_t rank = s_cnt[0];
        const bool added = s_cnt[1] == 0u;
        const uint32_t K_new = added ? K + 1u : K - 1u;
        // Reference behaviour (option zero_on_removal = 0, the default): the column that leaves keeps
        // x + gamma*d (homotopy-cpu.cpp:252) — 0, or an ulp of residue that stays in x for good and goes on
        // contributing to c.  This kernel's lists hold the support only: an exact 0 is handled here (it is what
        // the lists assume), a residue ends the launch at this iteration boundary and the launch-per-iteration
        // form (k_la_iter, which walks every column ever touched) repeats the iteration and goes on.
        bool residue = false;
        if (!zero_on_removal && !added && !no_candidate && K_new != 0u) residue = (S.xs[rank] + g * S.ds[rank]) != 0.f;
        if (SOLO) {
            if (no_candidate || K_new == 0u || K_new > kcap || !(c_inf - g > tol) || residue) {
                // Left to the resident form (nothing of this iteration has been committed yet): the rare
                // endings (no step, empty support, workspace full) and the LAST step of a path — it takes
                // lambda to ~0, where every column's candidate ties within rounding and no subset can know
                // the winner.
                exit_code = kPsExitHandOver;
                break;
            }
            log_entry(1u, round, c_inf, g, idx, gs_log, is_log);
        } else if (residue) {
            exit_code = kPsExitResidue;                // (same decision in every workgroup: same inputs, same bits)
            break;
        }
        if (lead && trace != nullptr && tid == 0 && round < trace_cap) {
            trace[round].idx = idx;
            trace[round].added = added ? 1u : 0u;
            trace[round].gamma = (double)g;
            trace[round].c_inf = (double)c_inf;
        }
        if (K_new == 0u || K_new > kcap) {
            // homotopy-cpu.cpp:248-249 (support became empty: break before x is updated) / workspace full
            if (K_new == 0u) {
                if (lead && tid == 0) insup[idx] = 0;
                report_empty = true;                     // the lists keep the one column: its x is handed back
                last_idx = idx; last_rank = rank; last_added = 0u; gamma_last = g;
                iter = round;
            } else {
                status = SS_HIP_ECAPACITY;
                iter = round - 1u;
            }
            c_inf_rep = c_inf;
            done_round = round;
            exit_code = kPsExitDone;
            break;
        }

        // loads that only need idx: c, q on the support (and at idx), the cache slot of idx
        float cj = 0.f, qj = 0.f;
        if (SOLO) {
            if (tid < K) { cj = s_cq[s_sps[tid]]; qj = s_cq[kSoloWidth + s_sps[tid]]; }
            else if (tid == K && added) { cj = s_cq[s_ipos]; qj = s_cq[kSoloWidth + s_ipos]; }
        } else if (tid < K) { cj = ld_f32(&cbuf[S.gam[tid]]); qj = ld_f32(&qbuf[S.gam[tid]]); }
        else if (tid == K && added) { cj = ld_f32(&cbuf[idx]); qj = ld_f32(&qbuf[idx]); }
        int32_t slot = 0;
        if (added) slot = subg ? (int32_t)s_ipos : slot_of[idx];
        // u1 = A_S^T a_idx, dot = a_idx.a_idx from the Gram column of idx (online_inverse.h:209-218): issued
        // here (the slot permitting), consumed in the inverse update after the c pass
        float u1v = 0.f;
        if (added && slot >= 0) {
            if (subg) {
                const float* gi = sa.subg + (size_t)slot * kSoloWidth;
                if (tid < K) u1v = gi[s_sps[tid]];
                else if (tid == K) u1v = gi[slot];
            } else {
                const float* gi = gcache + (size_t)slot * gpitch;
                if (tid < K) u1v = gi[S.gam[tid]];
                else if (tid == K) u1v = gi[idx];
            }
        }

        // x += gamma * direction over the OLD support (:252); the leaving column lands on exactly 0
        if (tid < K) {
            const float xn = S.xs[tid] + g * S.ds[tid];
            S.xs[tid] = (!added && tid == rank) ? 0.f : xn;
        }
        if (!SOLO && lead && tid == 0) insup[idx] = added ? 1 : 0;      // (solo: membership flags change at the commit)
#pragma unroll
        for (int k = 0; k < kPsCols; ++k)
            if (in[k] && col[k] == idx) act[k] = added ? 1u : 0u;
        iter = round;
        c_inf_rep = c_inf;
        gamma_last = g;
        last_idx = idx; last_rank = rank; last_added = added ? 1u : 0u;
        __syncthreads();

        // The next iteration's c only needs the x just updated (a column that entered carries x = 0, one
        // that left carries an exact 0 in its old list entry): form it (the loads above are still in
        // flight) and start its lambda exchange now.
        const bool miss = added && slot < 0;
        if (!SOLO || miss) c_pass();                          // (solo: after the new direction, together with q)
        const bool grow_next = !miss && (K_new + 1u > P) && (P < kcap);
        if (!grow_next) post_lambda();

        if (miss) {
            // No cached Gram column.  Before A is swept for it: the while-test of this iteration only
            // needs lambda = ||A^T(y - A x)||_inf for the x just updated — the exchange just started.
            // If the solve ends here (homotopy-cpu.cpp:272) the pending inverse update is never used.
            float lam;
            if (!poll_lambda(lam)) { exit_code = kPsExitWait; break; }
            // this tick has no step-length exchange of its own: an (empty-handed) one keeps the slot
            // discipline — it proves every workgroup has read the maxima
            if (!SOLO) {
                const uint32_t par2 = (tick & 1u) * kLaSlotStride;
                const bool ok = exchange_all(smin + par2, nb, w, kLaSlotNone, [&](uint64_t) {});
                if (!ok) { exit_code = kPsExitWait; break; }
                if (tid == 0) st_u64(&smax[par2 + w], kLaSlotEmpty);
            }
            log_entry(0u, round + 1u, lam, 0.f, 0u, 0.f, 0u);      // lambda after the step (x updated, K old entries)
            if (!(lam > tol) || round + 1u > max_iter) {
                c_inf_rep = lam;                         // iter = round already
                done_round = round + 1u;
                exit_code = kPsExitDone;
                break;
            }
            // the path goes on: hand the pending inverse update to k_gramupd, which reads the
            // iteration's c and q from the primary buffers (the c just formed belongs to the next one)
#pragma unroll
            for (int k = 0; k < kPsCols; ++k)
                if (in[k]) { c[col[k]] = SOLO ? s_cq[tid] : ld_f32(&cbuf[col[k]]); q[col[k]] = qv[k]; }
            save_lists_for_update = true;
            pend_rank = rank;
            pend_idx = idx;
            exit_code = kPsExitMiss;
            break;
        }
        ts[4] = wall_clock64();

        // correlations after the step on the new support, c - gamma*q (their sign is all that is used)
        const float cnv = cj - g * qj;
        float dv;
        if (added) {
            if (tid < K) S.u1[tid] = u1v;
            else if (tid == K) s_dd = u1v;                       // dot, replaced by d below
            __syncthreads();
            const uint32_t nn = K;
            // u2 = inv * u1 (online_inverse.h:224-225), one wave per row
            for (uint32_t i0 = wave; i0 < nn; i0 += 4 * NW) {
                float acc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t i = i0 + (uint32_t)r * NW;
                    acc[r] = 0.f;
                    if (i < nn)
                        for (uint32_t j = lane; j < nn; j += 64) acc[r] += S.I[i * Pp + j] * S.u1[j];
                }
                wave_sum4(acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t i = i0 + (uint32_t)r * NW;
                    if (lane == 0 && i < nn) S.u2[i] = acc[r];
                }
            }
            __syncthreads();
            // d = 1 / (dot - u1.u2) (online_inverse.h:228)
            float part = 0.f;
            for (uint32_t j = tid; j < nn; j += kPsThreads) part += S.u1[j] * S.u2[j];
            const float ssum = block_sum(part, sv);
            __syncthreads();
            if (tid == 0) s_dd = 1.f / (s_dd - ssum);
            __syncthreads();
            dv = s_dd;
            // full-G mode: the entering column gets the next LDS row; this workgroup fetches its slice of row
            // idx of G (used first by the q pass below, after several barriers)
            uint32_t new_lrow = kNoLdsRow;
            if (by_entry) {
                if (lds_rows_used < gl_rows) {
                    new_lrow = lds_rows_used++;
                    if (tid < kPsWidth) {
                        const uint32_t cg = SOLO ? s_sub[tid] : w * kPsWidth + tid;
                        Glds[new_lrow * kPsWidth + tid] = subg ? sa.subg[(size_t)slot * kSoloWidth + tid]
                                                               : gcache[(size_t)idx * gpitch + (cg < n ? cg : 0u)];
                    }
                }
            } else if ((uint32_t)slot < gl_used) {
                new_lrow = (uint32_t)slot;
            }
            // lists: insert at `rank`
            uint32_t ng = 0, ns = 0, nl = kNoLdsRow, np = 0;
            float nx = 0.f, ncn = 0.f;
            if (tid < K_new) {
                const uint32_t o = tid - (tid > rank ? 1u : 0u);
                if (tid == rank) { ng = idx; ns = (uint32_t)slot; nx = 0.f; nl = new_lrow; if (SOLO) np = s_ipos; }
                else { ng = S.gam[o]; ns = S.slt[o]; nx = S.xs[o]; nl = S.lrw[o]; if (SOLO) np = s_sps[o]; }
            }
            // cnv sits in thread j (old position) / thread K (idx): move through LDS
            if (tid <= K) S.cn[tid] = cnv;
            __syncthreads();
            if (tid < K_new) {
                const uint32_t o = tid - (tid > rank ? 1u : 0u);
                ncn = (tid == rank) ? S.cn[K] : S.cn[o];
            }
            __syncthreads();
            if (tid < K_new) { S.gam[tid] = ng; S.slt[tid] = ns; S.xs[tid] = nx; S.cn[tid] = ncn; S.lrw[tid] = nl; if (SOLO) s_sps[tid] = np; }
        } else {
            // remove row/column `rank` (online_inverse.h:275-290)
            const uint32_t nn = K;
            dv = S.I[rank * Pp + rank];                            // d of the reference; (-d u3) u3^T below
            const float sc = -(1.f / dv);
            __syncthreads();
            for (uint32_t i = tid; i < nn; i += kPsThreads) S.u2[i] = S.I[i * Pp + rank] * sc;
            // the column that left: exact zeros in the dense vectors, out of the lists
            if (!SOLO && lead && tid == 0) { x[idx] = 0.f; d[idx] = 0.f; }
            if (tid < K) S.cn[tid] = cnv;
            __syncthreads();
            uint32_t ng = 0, ns = 0, nl = kNoLdsRow, np = 0;
            float nx = 0.f, ncn = 0.f;
            if (tid < K_new) {
                const uint32_t o = tid + (tid >= rank ? 1u : 0u);
                ng = S.gam[o]; ns = S.slt[o]; nx = S.xs[o]; ncn = S.cn[o]; nl = S.lrw[o];
                if (SOLO) np = s_sps[o];
            }
            __syncthreads();
            if (tid < K_new) { S.gam[tid] = ng; S.slt[tid] = ns; S.xs[tid] = nx; S.cn[tid] = ncn; S.lrw[tid] = nl; if (SOLO) s_sps[tid] = np; }
            else if (tid == K_new) { S.xs[tid] = 0.f; S.ds[tid] = 0.f; }     // the vacated entry is padding again
        }
        __syncthreads();
        ts[5] = wall_clock64();
        // sign(c[Gamma]) with dead zone tol (homotopy-cpu.cpp:259-260), direction = inv * sign (:263).
        // The new inverse is not stored yet: its elements are formed on the fly, by the same
        // expressions the store pass uses; the store pass runs during the next step-length exchange.
        if (tid < K_new) S.sg[tid] = sign_tol(S.cn[tid], tol);
        // (solo: no exchange to hide the store pass behind — the inverse is stored first and read back plainly;
        // the stored elements are the values the on-the-fly expressions give)
        if (SOLO) store_new_inverse(S.I, Pp, S.u2, added, rank, dv, K_new);
        else __syncthreads();
        for (uint32_t a0 = wave; a0 < K_new; a0 += 4 * NW) {
            float acc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t a = a0 + (uint32_t)r * NW;
                acc[r] = 0.f;
                if (a < K_new)
                    for (uint32_t b = lane; b < K_new; b += 64)
                        acc[r] += (SOLO ? S.I[a * Pp + b] : new_inverse_elem(S.I, Pp, S.u2, added, rank, dv, a, b)) * S.sg[b];
            }
            wave_sum4(acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t a = a0 + (uint32_t)r * NW;
                if (lane == 0 && a < K_new) S.ds[a] = acc[r];
            }
        }
        __syncthreads();
        pend = !SOLO; pend_added = added; pend_rk = rank; pend_dv = dv; pend_K = K_new;
        K = K_new;
        ts[6] = wall_clock64();
        if (SOLO) cq_pass();                                  // c and q of the next iteration (x updated, new direction)
        if (dbg != nullptr && lead && tid == 0 && round < 1024u) {
            ts[7] = wall_clock64();
            for (int k2 = 0; k2 < 8; ++k2) dbg[(size_t)round * 8 + k2] = ts[k2];
        }
        if (grow_next) {
            if (pend) store_new_inverse(S.I, Pp, S.u2, pend_added, pend_rk, pend_dv, pend_K);
            pend = false;
            exit_code = kPsExitGrow;
        }
    }

    // ---- end of the launch: workgroup 0 hands the state back in the global layout --------------------
    if (!lead) return;
    if (SOLO) {
        // staged, in the layout of ss_hip_internal.h: k_la_vpublish commits it once the log is verified
        uint32_t* sg = sa.stage;
        constexpr uint32_t LP = kSoloListPitch;
        if (tid < K) {
            sg[kSoloStageHead + tid] = S.gam[tid];
            sg[kSoloStageHead + LP + tid] = __float_as_uint(S.xs[tid]);
            sg[kSoloStageHead + 2 * LP + tid] = __float_as_uint(S.ds[tid]);
        }
        if (save_lists_for_update && tid < K + 1u) {
            const uint32_t o = tid - (tid > pend_rank ? 1u : 0u);
            sg[kSoloStageHead + 3 * LP + tid] = (tid == pend_rank) ? pend_idx : S.gam[o];
        }
        for (uint32_t e = tid; e < K * K; e += kPsThreads) {
            const uint32_t a = e / K, b = e - a * K;
            sg[kSoloStageHead + 4 * LP + 1 + e] = __float_as_uint(S.I[a * Pp + b]);
        }
        if (tid == 0) {
            sg[0] = K; sg[1] = save_lists_for_update ? 1u : 0u; sg[2] = iter;
            sg[3] = __float_as_uint(c_inf_rep); sg[4] = __float_as_uint(gamma_last);
            sg[5] = last_idx; sg[6] = last_rank; sg[7] = last_added; sg[8] = tick; sg[9] = exit_code; sg[10] = done_round;
            st->solo_nlog = nlog;
            st->solo_pending = replay != 0u ? 2u : 1u;
        }
        return;
    }
    if (save_lists_for_update) {
        // the pick is made (x updated, K_new columns) but the inverse still describes the old support:
        // buffer `cur` keeps the old support and inverse, buffer cur^1 receives the new sorted support
        // — exactly what select_toggle leaves behind for k_gramupd
        const uint32_t K_new = K + 1u;
        if (tid < K_new) {
            const uint32_t o = tid - (tid > pend_rank ? 1u : 0u);
            gam_alt[tid] = (tid == pend_rank) ? pend_idx : S.gam[o];
        }
    }
    for (uint32_t e = tid; e < K * K; e += kPsThreads) {
        const uint32_t a = e / K, b = e - a * K;
        Ig[(size_t)a * kcap + b] = S.I[a * Pp + b];
    }
    if (tid < K) {
        const uint32_t cl = S.gam[tid];
        gam_cur[tid] = cl;
        x[cl] = S.xs[tid];
        d[cl] = S.ds[tid];
    }
    if (!zero_on_removal) {
        // reference mode: k_la_iter / k_la_cq walk the `touched` list (every column whose x may be non-zero).
        // Columns that left the support inside this kernel carry exact zeros (a residue ends the launch), so
        // the list that will be current — this support, or the pending one — is all of it.
        uint32_t* const tch = touched2 + (size_t)(save_lists_for_update ? (cur ^ 1u) : cur) * kcap;
        const uint32_t Kt = save_lists_for_update ? K + 1u : K;
        if (tid < Kt) {
            const uint32_t o = save_lists_for_update ? tid - (tid > pend_rank ? 1u : 0u) : tid;
            tch[tid] = (save_lists_for_update && tid == pend_rank) ? pend_idx : S.gam[o];
        }
        if (tid == 0) st->ntouched = Kt;
    }
    if (tid == 0) {
        sy->tick = tick;
        st->K = report_empty ? 0u : (save_lists_for_update ? K + 1u : K);
        st->iter = iter;
        st->c_inf = (double)c_inf_rep;
        st->gamma = (double)gamma_last;
        st->idx = last_idx;
        st->rank = last_rank;
        st->added = last_added;
        if (exit_code == kPsExitWait) { status = SS_HIP_ERUNTIME; done_round = iter + 1u; }
        if (status != 0u) st->status = status;
        if (exit_code == kPsExitDone || exit_code == kPsExitWait) {
            st->done_round = done_round;
            st->need_sweep = 0;
            st->done = 1;
            signal_done(hflags, nullptr, 1u, done_round);
        } else if (exit_code == kPsExitMiss) {
            st->need_sweep = 1;
            const uint32_t nm = st->nmiss + 1u;
            st->nmiss = nm;
            __hip_atomic_store(&hflags[2], nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            // the support outgrew the tier (the host picks the next one), or a removal left a rounding residue in
            // x: 0xffffffff = no tier will do, the launch-per-iteration form goes on
            __hip_atomic_store(&hflags[3], exit_code == kPsExitResidue ? 0xffffffffu : K + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        bump_seq(st, hflags);
    }
}

// ---- host side -------------------------------------------------------------------------------------
// LDS of a launch: the replica of the active set for the tier, the rest (whole rows) for the
// workgroup's slice of the cached Gram rows
static uint32_t persist_tier_cols(uint32_t P) { return P > kLaLdsSmall ? kLaLdsLarge : kLaLdsSmall; }
static uint32_t persist_gl_rows(uint32_t P)
{
    const size_t replica = ps_lds_words(persist_tier_cols(P)) * sizeof(float);
    if (replica >= kPsLdsBudget) return 0;
    return (uint32_t)((kPsLdsBudget - replica) / (kPsWidth * sizeof(float))) & ~15u;
}
static size_t persist_lds_bytes(uint32_t P)
{
    return ps_lds_words(persist_tier_cols(P)) * sizeof(float) + (size_t)persist_gl_rows(P) * kPsWidth * sizeof(float);
}

// workgroups of k_la_persist the device can keep resident for LDS tier P (0: unusable)
static int persist_workers(ss_hip_ctx* ctx, uint32_t P)
{
    const int tier = P > kLaLdsSmall ? 1 : 0;
    if (ctx->persist_workers[tier] >= 0) return ctx->persist_workers[tier];
    int result = 0;
    const size_t lds = persist_lds_bytes(P);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_la_persist<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPsLdsBudget) != hipSuccess) {
        (void)hipGetLastError();
        ctx->persist_workers[tier] = 0;
        return 0;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_la_persist<false>, kPsThreads, lds) == hipSuccess && per_cu > 0) {
        // the whole grid must be resident: at most what the occupancy allows (a launch that does not
        // get there gives up through its bounded waits and the solve falls back to k_la_iter)
        const long cap = (long)std::min(per_cu, 2) * ctx->num_cus;
        result = (int)std::min<long>(kLaSlotStride, std::max<long>(0, cap));
    } else {
        (void)hipGetLastError();
    }
    ctx->persist_workers[tier] = result;
    return result;
}

static uint32_t persist_grid_workers(ss_hip_ctx* ctx, uint32_t P)
{
    const uint32_t n = (uint32_t)ctx->n;
    const uint32_t cap = (uint32_t)persist_workers(ctx, P);
    const uint32_t want = (n + kPsWidth - 1) / kPsWidth;
    const uint32_t nw = std::min(want, cap);
    if (nw == 0 || (uint64_t)nw * kPsWidth * kPsCols < n) return 0;
    return nw;
}

bool la_persist_usable(ss_hip_ctx* ctx, uint32_t lds_cols)
{
    if (ctx->n >= (1u << 30) || lds_cols > kLaLdsLarge) return false;      // column indices travel in 30 bits
    return persist_grid_workers(ctx, lds_cols) != 0;
}

hipError_t launch_la_persist_f32(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, uint32_t max_iter, uint32_t lds_cols, bool after_solo)
{
    const uint32_t P = lds_cols;
    const uint32_t nw = persist_grid_workers(ctx, P);
    if (nw == 0 || P > kLaLdsLarge || ws.cq_alt == nullptr) return hipErrorInvalidConfiguration;
    // the LDS tier fixes the allocation (and with it the residency the grid was sized for)
    const size_t lds = persist_lds_bytes(P);
    uint64_t* smax = reinterpret_cast<uint64_t*>(ws.la_sync + 1);
    uint64_t* smin = smax + 2 * kLaSlotStride;
    hipLaunchKernelGGL(k_la_persist<false>, dim3(nw), dim3(kPsThreads), lds, ctx->stream, tol, max_iter, (uint32_t)ctx->n, P, persist_gl_rows(P), ws.gram_is_full ? 1 : 0,
                       (const float*)ws.gcache, (const int32_t*)ws.slot_of, (const float*)ws.c0, ws.gpitch,
                       ws.c, ws.q, ws.cq_alt, ws.cq_alt + ctx->n_pad, ws.x, ws.d, ws.insup, ws.gam, ws.inv[0], ws.inv[1], ws.tcand, ws.dims, ws.st,
                       ws.la_sync, smax, smin, ctx->dev_flags, ws.trace, ws.trace_cap, ctx->tie_guard, ctx->zero_on_removal, ws.touched, after_solo ? 1 : 0, ws.la_dbg, SoloArgs{});
    return hipGetLastError();
}

// ---- speculative form: one workgroup, first LDS tier ---------------------------------------------------
// (its static LDS — subset, offers, position tables, c0 / c / q of the subset — takes ~9 KiB of the 160 KiB:
// the Gram slice gets 96 rows instead of 112; a launch that does not fit is refused, never silently shrunk)
constexpr uint32_t kSoloGlRows = 96;
static size_t solo_lds_bytes()
{
    return ps_lds_words(persist_tier_cols(kLaLdsSmall)) * sizeof(float) + (size_t)kSoloGlRows * kPsWidth * sizeof(float);
}

bool la_solo_usable(ss_hip_ctx* ctx)
{
    if (ctx->n >= (1u << 30)) return false;
    if (ctx->solo_attr_set < 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_la_persist<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)solo_lds_bytes());
        if (e != hipSuccess) (void)hipGetLastError();
        ctx->solo_attr_set = e == hipSuccess ? 1 : 0;
    }
    return ctx->solo_attr_set == 1;
}

hipError_t launch_la_solo_f32(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, uint32_t max_iter)
{
    const uint32_t P = kLaLdsSmall;
    if (!la_solo_usable(ctx) || ws.cq_alt == nullptr || ws.solo_log == nullptr || ws.solo_stage == nullptr) return hipErrorInvalidConfiguration;
    const size_t lds = solo_lds_bytes();
    uint64_t* smax = reinterpret_cast<uint64_t*>(ws.la_sync + 1);
    uint64_t* smin = smax + 2 * kLaSlotStride;
    SoloArgs sa;
    sa.slot_col = ws.slot_col;
    sa.cand_top = ws.cand_top;
    sa.ncand = kCandPerBlock * ws.nvwg;
    sa.subset_cap = (uint32_t)std::max(0, std::min(ctx->solo_subset, (int)kSoloWidth));
    sa.log = ws.solo_log;
    sa.sub_pos = ws.sub_pos;
    sa.stage = ws.solo_stage;
    sa.subg = ws.subg;
    sa.sub_cols = ws.sub_cols;
    sa.prog = (ws.sub_cols != nullptr && ctx->early_adapt) ? reinterpret_cast<float*>(ws.sub_cols + kSoloWidth) : nullptr;
    hipLaunchKernelGGL(k_la_persist<true>, dim3(1), dim3(kPsThreads), lds, ctx->stream, tol, max_iter, (uint32_t)ctx->n, P, kSoloGlRows, ws.gram_is_full ? 1 : 0,
                       (const float*)ws.gcache, (const int32_t*)ws.slot_of, (const float*)ws.c0, ws.gpitch,
                       ws.c, ws.q, ws.cq_alt, ws.cq_alt + ctx->n_pad, ws.x, ws.d, ws.insup, ws.gam, ws.inv[0], ws.inv[1], ws.tcand, ws.dims, ws.st,
                       ws.la_sync, smax, smin, ctx->dev_flags, ws.trace, ws.trace_cap, ctx->tie_guard, ctx->zero_on_removal, ws.touched, 0, ws.la_dbg, sa);
    return hipGetLastError();
}

}  // namespace sship
