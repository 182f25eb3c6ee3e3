// colshard.hip — ONE signal over a dictionary whose COLUMNS are split across the GPUs of a node:
// C-ABI ss_hip_homotopy_colshard_{create,solve}_f32, one process per GPU, RCCL over xGMI.
//
// The reference's loop (src/solvers/homotopy-cpu.cpp:236-272) with the O(m n) work local to the shard and every
// reduction over all columns as a device-side collective:
//
//   every rank owns A[:, col_lo : col_lo + n_local] (its own context: column-contiguous copy, k_sweep);
//   the ACTIVE SET is replicated — sorted support and touched lists (global column indices), the explicit
//   (A_S^T A_S)^-1, x on the touched columns, the direction on the support, and a copy of every column that was ever
//   active (AS, what online_column_inverse keeps as _At: src/linalg/online_inverse.h:35-63) — and every rank
//   performs the same O(K m + K^2) update on it: same kernels, same inputs, same order, hence the same bits;
//   nothing of the active set is ever broadcast except the column that enters.
//
// Per iteration, all enqueued on the context's stream (the device decides, the host only counts rounds):
//   k_cs_rp       r = y - AS x, p = AS d                       replicated, m-vectors          (homotopy-cpu.cpp:94-96,114-116)
//   k_sweep       [c_loc, q_loc] = A_loc^T [r, p]              local, HBM-bound: the pass over the shard (:97,:120)
//   k_cs_lmax     local (max |c|, first global index)          -> one u64
//   all-reduce    MAX of (bits(|c|) << 32 | ~index)            lambda = ||c||_inf and the left-most arg-max (:32-37)
//   k_cs_scan     loop control; find_max_gamma's scan over the local columns (:122-163) -> (bits(t) << 32 | index)
//   all-reduce    MIN                                          smallest step, then smallest global index (left-most rule)
//   k_cs_select   pick, toggle of the replicated lists, x update (:246-252); the owner of the entering column puts it
//                 (m floats) and every owner its c - gamma q on the support into the exchange buffer, zeros elsewhere
//   all-reduce    SUM of m + kcap floats                       (= a broadcast without knowing the root on the host)
//   k_cs_update   AS gains the column; u1, the bordered / deflated inverse (online_inverse.h:209-290), sign and the new
//                 direction (:257-267), replicated
// Three small collectives per iteration (8 B, 8 B, (m + kcap) * 4 B): latency-bound on xGMI, which is why this form is
// for dictionaries beyond one GPU's HBM or n >> 10^6; signals sharded across ranks is the scaling the
// path has at the reference's sizes.
//
// Transport: RCCL (librccl.so opened at run time; the communicator is built from a ncclUniqueId the caller distributes),
// or a table of HOST collectives supplied by the caller (tests: two gloo ranks sharing one GPU; other transports).
// Same results either way; sharded and unsharded runs of this path agree bit for bit (every column's correlation is its
// own chain, the reductions are exact max / min, the active set is replicated arithmetic).
#include "ss_hip_internal.h"
#include "ss_hip_device.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace sship {

// ---- RCCL at run time (no link-time dependency: a single-GPU user never loads it) ---------------------------
typedef struct ncclComm* cs_comm_t;
typedef struct { char internal[SS_HIP_COMM_ID_BYTES]; } cs_unique_id;
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(cs_unique_id*) = nullptr;
    int (*CommInitRank)(cs_comm_t*, int, cs_unique_id, int) = nullptr;
    int (*CommDestroy)(cs_comm_t) = nullptr;
    int (*CommAbort)(cs_comm_t) = nullptr;             // (optional: a communicator with an outstanding collective is aborted, not destroyed)
    int (*AllReduce)(const void*, void*, size_t, int, int, cs_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
// rccl.h: ncclDataType_t ncclUint64 = 5, ncclFloat32 = 7; ncclRedOp_t ncclSum = 0, ncclMax = 2, ncclMin = 3
constexpr int kNcclUint64 = 5, kNcclFloat32 = 7, kNcclSum = 0, kNcclMax = 2, kNcclMin = 3;

static Rccl* rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r.lib ? &r : nullptr;
    tried = true;
    const char* names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
    for (const char* nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return nullptr;
    r.GetUniqueId = reinterpret_cast<int (*)(cs_unique_id*)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<int (*)(cs_comm_t*, int, cs_unique_id, int)>(dlsym(r.lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<int (*)(cs_comm_t)>(dlsym(r.lib, "ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<int (*)(cs_comm_t)>(dlsym(r.lib, "ncclCommAbort"));
    r.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, cs_comm_t, hipStream_t)>(dlsym(r.lib, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(r.lib, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) { r.lib = nullptr; return nullptr; }
    return &r;
}

// ---- state of a column-sharded context (ss_hip_ctx::colshard) ----------------------------------------------------
struct ColShard {
    uint32_t col_lo = 0, n_total = 0, n_local = 0;
    int rank = 0, world = 1;
    cs_comm_t comm = nullptr;
    bool have_host = false;
    ss_hip_collectives host{};
    uint32_t kcap = 0;
    // replicated active set (device)
    uint32_t* gam = nullptr;        // [2][kcap] sorted support, global indices (ping-pong with inv)
    uint32_t* tch = nullptr;        // [2][kcap] sorted touched list
    uint32_t* trow = nullptr;       // [2][kcap] row of AS that holds each touched column
    float* xt = nullptr;            // [2][kcap] x on the touched columns (touched order)
    float* ds = nullptr;            // [2][kcap] direction on the support (support order)
    float* inv = nullptr;           // [2][kcap][kcap]
    float* AS = nullptr;            // [kcap][ldm] every column that was ever active, in order of first entry
    float* u1 = nullptr; float* u2 = nullptr; float* sgn = nullptr;     // [kcap]
    uint64_t* red = nullptr;        // [2] exchange words of the two (value, index) reductions
    float* xbuf = nullptr;          // [ldm + kcap + 8] exchange buffer: the entering column, then c - gamma q on the support
    uint32_t xcount = 0;            // its length in floats
    // pinned staging for host collectives
    unsigned char* hstage = nullptr;
    // the ranks' agreement on "everything this solve needs is allocated" (made at create: it must exist when an allocation fails)
    uint64_t* agree_dev = nullptr;  // [1] device word of the RCCL all-reduce
    uint64_t* agree_host = nullptr; // [1] pinned
    bool dead = false;              // the communicator was aborted in the middle of a solve: the context refuses further solves
};

struct CsState {                     // device-resident, replicated scalars (one 128-byte line)
    uint32_t done, status, iter, K, ntouched, idx, rank, added, cur, seen, newrow, pad0;
    float lambda, gamma, lambda0, dot;
    uint32_t ticket_scan, ticket_upd, pad1[14];
};
static_assert(sizeof(CsState) == 128, "CsState layout");

constexpr int kCsThreads = 256;
constexpr uint32_t kCsNone = 0xffffffffu;

// ---- k_cs_rp: r = y - sum_t xt[t] AS[trow[t]] ; p = sum_j ds[j] AS[row(gam[j])] --------------------------------------
// one thread per row, touched order (sorted by global column): the same arithmetic on every rank
__global__ __launch_bounds__(kCsThreads)
void k_cs_rp(const float* __restrict__ AS, uint32_t ldm, uint32_t m, const float* __restrict__ y, uint32_t kcap,
             const uint32_t* __restrict__ tch2, const uint32_t* __restrict__ trow2, const float* __restrict__ xt2,
             const uint32_t* __restrict__ gam2, const float* __restrict__ ds2, float* __restrict__ r, float* __restrict__ p,
             const CsState* st)
{
    if (st->done) return;
    const uint32_t i = blockIdx.x * kCsThreads + threadIdx.x;
    if (i >= ldm) return;
    const uint32_t cur = st->cur, nt = st->ntouched, K = st->K;
    const uint32_t* tch = tch2 + (size_t)cur * kcap;
    const uint32_t* trow = trow2 + (size_t)cur * kcap;
    const float* xt = xt2 + (size_t)cur * kcap;
    const uint32_t* gam = gam2 + (size_t)cur * kcap;
    const float* ds = ds2 + (size_t)cur * kcap;
    float ar = 0.f, ap = 0.f;
    uint32_t j = 0;                                        // position in the support (a sub-sequence of the touched list)
    for (uint32_t t = 0; t < nt; ++t) {
        const float a = AS[(size_t)trow[t] * ldm + i];
        ar += xt[t] * a;
        if (j < K && gam[j] == tch[t]) { ap += ds[j] * a; ++j; }
    }
    r[i] = i < m ? y[i] - ar : 0.f;
    p[i] = i < m ? ap : 0.f;
}

// ---- k_cs_lmax: the shard's (max |c|, first index) as one ordered word ------------------------------------------------
__global__ __launch_bounds__(kCsThreads)
void k_cs_lmax(const float* __restrict__ pmax_val, const uint32_t* __restrict__ pmax_idx, uint32_t nb, uint32_t col_lo,
               uint32_t n_local, uint64_t* __restrict__ red, const CsState* st)
{
    __shared__ float sv[16];
    __shared__ uint32_t si[16];
    float v;
    uint32_t ix;
    reduce_sweep_partials(pmax_val, pmax_idx, nb, v, ix, sv, si);
    if (threadIdx.x == 0) {
        // (a shard without columns offers nothing; larger |c| first, then the smaller global index: ~index)
        uint64_t w = 0ull;
        if (n_local != 0u && v >= 0.f) w = ((uint64_t)__float_as_uint(v) << 32) | (uint64_t)(~(col_lo + ix));
        red[0] = (st != nullptr && st->done) ? 0ull : w;
    }
}

// ---- k_cs_init: first pick (homotopy-cpu.cpp:217-221) from the reduced word; the owner offers its column ---------------
__global__ __launch_bounds__(kCsThreads)
void k_cs_init(const float* __restrict__ At, uint32_t ldm, uint32_t col_lo, uint32_t n_local, const float* __restrict__ c0,
               const uint64_t* __restrict__ red, float* __restrict__ xbuf, uint32_t xcount, CsState* st)
{
    const uint64_t w = red[0];
    const uint32_t idx = ~(uint32_t)w;
    const float lam = __uint_as_float((uint32_t)(w >> 32));
    const bool mine = idx >= col_lo && idx < col_lo + n_local;
    for (uint32_t i = threadIdx.x; i < xcount; i += blockDim.x) {
        float v = 0.f;
        if (mine && i < ldm) v = At[(size_t)(idx - col_lo) * ldm + i];
        if (mine && i == ldm) v = c0[idx - col_lo];                       // c0[idx] (option strict_sign seeds the first sign with it)
        xbuf[i] = v;
    }
    if (threadIdx.x == 0) {
        st->done = 0; st->status = 0; st->iter = 0; st->K = 0; st->ntouched = 0; st->idx = idx; st->rank = 0; st->added = 1;
        st->cur = 0; st->seen = 0; st->newrow = 0; st->lambda = lam; st->gamma = 0.f; st->lambda0 = lam;
        st->ticket_scan = 0; st->ticket_upd = 0;
    }
}

// ---- k_cs_first: AS[0] = the first column; inv = [1 / ||a||^2] through the norm (online_inverse.h:193-201); first direction
__global__ __launch_bounds__(kUpdThreads)
void k_cs_first(const float* __restrict__ xbuf, uint32_t ldm, uint32_t col_lo, uint32_t n_local, uint32_t kcap, float tol,
                int strict_sign, float* __restrict__ AS, uint32_t* gam, uint32_t* tch, uint32_t* trow, float* xt, float* ds,
                float* inv, float* __restrict__ d_loc, uint8_t* __restrict__ insup, CsState* st, TraceEntry* trace)
{
    __shared__ float sv[16];
    for (uint32_t i = threadIdx.x; i < ldm; i += blockDim.x) AS[i] = xbuf[i];
    __syncthreads();
    float acc = 0.f;
    for (uint32_t i = threadIdx.x; i < ldm; i += blockDim.x) { const float a = xbuf[i]; acc += a * a; }
    const float dot = block_sum(acc, sv);
    if (threadIdx.x == 0) {
        const uint32_t idx = st->idx;
        const float nrm = sqrtf(dot);
        const float inv00 = 1.f / (nrm * nrm);
        const float lam = st->lambda;
        const float seed = strict_sign ? xbuf[ldm] : lam;                 // first-step quirk (homotopy-cpu.cpp:223-227)
        const float d0 = sign_tol(seed, tol) * inv00;
        gam[0] = idx; tch[0] = idx; trow[0] = 0u; xt[0] = 0.f; ds[0] = d0; inv[0] = inv00;
        if (idx >= col_lo && idx < col_lo + n_local) { d_loc[idx - col_lo] = d0; insup[idx - col_lo] = 1; }
        st->K = 1; st->ntouched = 1; st->cur = 0;
        if (trace != nullptr) { trace[0].idx = idx; trace[0].added = 1; trace[0].gamma = 0.0; trace[0].c_inf = (double)lam; }
        (void)kcap;
    }
}

// ---- k_cs_scan: loop control and find_max_gamma's scan (homotopy-cpu.cpp:122-163) over the shard's columns -------------
__global__ __launch_bounds__(kCsThreads)
void k_cs_scan(uint32_t round, float tol, uint32_t max_iter, uint32_t n_local, uint32_t col_lo,
               const float* __restrict__ c, const float* __restrict__ q, const float* __restrict__ x, const float* __restrict__ d,
               const uint8_t* __restrict__ insup, uint64_t* red, uint64_t* pmin, int tie_guard, CsState* st, uint32_t* hflags)
{
    __shared__ float sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_flag;
    if (st->done) return;
    const uint64_t w = red[0];
    const float c_inf = __uint_as_float((uint32_t)(w >> 32));
    // do { ... } while (iter < max_iter && c_inf > tolerance): the test of iteration round-1; every rank reads the same word
    if ((round > 1 && !(c_inf > tol)) || round > max_iter) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->lambda = c_inf;
            st->iter = round - 1;
            st->done = 1;
            red[1] = ~0ull;
            if (hflags) { __hip_atomic_store(&hflags[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&hflags[0], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
        return;
    }
    float best = Lim<float>::max();
    uint32_t best_i = kCsNone;
    for (uint32_t i = blockIdx.x * kCsThreads + threadIdx.x; i < n_local; i += gridDim.x * kCsThreads) {
        float m = Lim<float>::max();
        if (insup[i]) {
            const float t = -x[i] / d[i];
            if (t > 0.f && t < m) m = t;
        } else {
            const float qi = q[i], ci = c[i];
            const float dl = 1.f - qi, dr = 1.f + qi;
            if (dl != 0.f) {
                float t = (c_inf - ci) / dl;
                if (tie_guard && t == 0.f && dl > 0.f) t = Lim<float>::tiny();
                if (t > 0.f && t < m) m = t;
            }
            if (dr != 0.f) {
                float t = (c_inf + ci) / dr;
                if (tie_guard && t == 0.f && dr > 0.f) t = Lim<float>::tiny();
                if (t > 0.f && t < m) m = t;
            }
        }
        if (better_min(m, col_lo + i, best, best_i)) { best = m; best_i = col_lo + i; }
    }
    block_reduce_pair<float, false>(best, best_i, sv, si);
    if (threadIdx.x == 0) {
        // positive floats order like their bit patterns; no candidate: (FLT_MAX, 0) as in the reference (:123-124)
        const uint64_t pk = best < Lim<float>::max() ? (((uint64_t)__float_as_uint(best) << 32) | best_i)
                                                     : ((uint64_t)__float_as_uint(Lim<float>::max()) << 32);
        __hip_atomic_store(&pmin[blockIdx.x], pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!arrive_last_relaxed(&st->ticket_scan, gridDim.x, &s_flag)) return;
    uint64_t mn = ~0ull;
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += blockDim.x) {
        const uint64_t v = __hip_atomic_load(&pmin[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mn = v < mn ? v : mn;
    }
    // block minimum of 64-bit words through the two halves (value first, then index)
    __shared__ uint64_t s_m[kCsThreads];
    s_m[threadIdx.x] = mn;
    __syncthreads();
    for (uint32_t o = kCsThreads / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) { const uint64_t a = s_m[threadIdx.x], b2 = s_m[threadIdx.x + o]; s_m[threadIdx.x] = a < b2 ? a : b2; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { red[1] = s_m[0]; st->lambda = c_inf; }
}

// ---- k_cs_select: the pick, the toggle of the replicated lists and the x update (homotopy-cpu.cpp:246-252) --------------
// one workgroup per rank, the same on every rank; the owners fill the exchange buffer
__global__ __launch_bounds__(kUpdThreads)
void k_cs_select(uint32_t round, const float* __restrict__ At, uint32_t ldm, uint32_t col_lo, uint32_t n_local, uint32_t kcap,
                 const uint64_t* __restrict__ red, const float* __restrict__ c, const float* __restrict__ q,
                 float* __restrict__ x_loc, uint8_t* __restrict__ insup,
                 uint32_t* gam2, uint32_t* tch2, uint32_t* trow2, float* xt2, const float* ds2,
                 float* __restrict__ xbuf, uint32_t xcount, int zero_on_removal, CsState* st, uint32_t* hflags,
                 TraceEntry* trace, uint32_t trace_cap)
{
    __shared__ uint32_t s_cnt[2];
    if (st->done) { for (uint32_t i = threadIdx.x; i < xcount; i += blockDim.x) xbuf[i] = 0.f; return; }
    const uint64_t w = red[1];
    const float g = __uint_as_float((uint32_t)(w >> 32));
    const uint32_t idx = (uint32_t)w;
    const uint32_t cur = st->cur, K = st->K, nt = st->ntouched;
    const uint32_t* gam = gam2 + (size_t)cur * kcap;
    uint32_t* gam_new = gam2 + (size_t)(cur ^ 1u) * kcap;
    const uint32_t* tch = tch2 + (size_t)cur * kcap;
    uint32_t* tch_new = tch2 + (size_t)(cur ^ 1u) * kcap;
    const uint32_t* trow = trow2 + (size_t)cur * kcap;
    uint32_t* trow_new = trow2 + (size_t)(cur ^ 1u) * kcap;
    const float* xt = xt2 + (size_t)cur * kcap;
    float* xt_new = xt2 + (size_t)(cur ^ 1u) * kcap;
    const float* ds = ds2 + (size_t)cur * kcap;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t lr = 0, lt = 0;
    for (uint32_t j = threadIdx.x; j < K; j += blockDim.x) lr += gam[j] < idx ? 1u : 0u;
    for (uint32_t j = threadIdx.x; j < nt; j += blockDim.x) lt += tch[j] < idx ? 1u : 0u;
    if (lr) atomicAdd(&s_cnt[0], lr);
    if (lt) atomicAdd(&s_cnt[1], lt);
    __syncthreads();
    const uint32_t rank = s_cnt[0], trank = s_cnt[1];
    const bool added = !(rank < K && gam[rank] == idx);
    const bool seen = trank < nt && tch[trank] == idx;
    const uint32_t K_new = added ? K + 1u : K - 1u;
    const uint32_t nt_new = (added && !seen) ? nt + 1u : nt;
    if (trace != nullptr && threadIdx.x == 0 && round < trace_cap) {
        trace[round].idx = idx; trace[round].added = added ? 1u : 0u; trace[round].gamma = (double)g; trace[round].c_inf = (double)st->lambda;
    }
    if (K_new == 0u || K_new > kcap || nt_new > kcap) {
        // the support became empty (homotopy-cpu.cpp:248-249: break before x is updated), or the workspace is exhausted
        for (uint32_t i = threadIdx.x; i < xcount; i += blockDim.x) xbuf[i] = 0.f;
        if (threadIdx.x == 0) {
            if (K_new == 0u) { st->K = 0; st->idx = idx; st->added = 0; st->gamma = g; st->iter = round; if (idx >= col_lo && idx < col_lo + n_local) insup[idx - col_lo] = 0; }
            else { st->status = SS_HIP_ECAPACITY; st->iter = round - 1; }
            st->done = 1;
            if (hflags) { __hip_atomic_store(&hflags[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&hflags[0], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
        return;
    }
    // x += gamma * direction over the OLD support (direction is zero elsewhere); touched order keeps the residue of
    // columns that left (reference mode).  The new touched list / x / AS-row tables are written out of place.
    for (uint32_t t = threadIdx.x; t < nt_new; t += blockDim.x) {
        // source position in the old touched list
        const bool ins = added && !seen;
        const uint32_t ot = ins ? (t < trank ? t : (t == trank ? kCsNone : t - 1u)) : t;
        uint32_t col, row;
        float xv;
        if (ot == kCsNone) { col = idx; row = nt; xv = 0.f; }
        else { col = tch[ot]; row = trow[ot]; xv = xt[ot]; }
        // direction of this column under the OLD support (binary search in gam)
        uint32_t lo = 0, hi = K;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (gam[mid] < col) lo = mid + 1u; else hi = mid; }
        if (lo < K && gam[lo] == col) {
            const float xn = xv + g * ds[lo];
            xv = (!added && zero_on_removal && col == idx) ? 0.f : xn;
        }
        tch_new[t] = col; trow_new[t] = row; xt_new[t] = xv;
        if (col >= col_lo && col < col_lo + n_local) x_loc[col - col_lo] = xv;
    }
    if (added) {
        for (uint32_t j = threadIdx.x; j < K_new; j += blockDim.x) gam_new[j] = j < rank ? gam[j] : (j == rank ? idx : gam[j - 1u]);
    } else {
        for (uint32_t j = threadIdx.x; j < K_new; j += blockDim.x) gam_new[j] = gam[j + (j >= rank ? 1u : 0u)];
    }
    // exchange buffer: the entering column (if it has no row of AS yet) from its owner; c - gamma q on the NEW support
    // from each column's owner; zeros elsewhere (the sum over ranks is then a copy)
    const bool mine = idx >= col_lo && idx < col_lo + n_local;
    const bool send_col = added && !seen && mine;
    for (uint32_t i = threadIdx.x; i < ldm; i += blockDim.x) xbuf[i] = send_col ? At[(size_t)(idx - col_lo) * ldm + i] : 0.f;
    for (uint32_t a = threadIdx.x; a < kcap; a += blockDim.x) {
        float v = 0.f;
        if (a < K_new) {
            const uint32_t col = a < rank ? gam[a] : (added ? (a == rank ? idx : gam[a - 1u]) : gam[a + 1u]);
            if (col >= col_lo && col < col_lo + n_local) v = c[col - col_lo] - g * q[col - col_lo];
        }
        xbuf[ldm + a] = v;
    }
    for (uint32_t i = ldm + kcap + threadIdx.x; i < xcount; i += blockDim.x) xbuf[i] = 0.f;
    if (threadIdx.x == 0) {
        if (mine) insup[idx - col_lo] = added ? 1 : 0;
        st->K = K_new; st->ntouched = nt_new; st->idx = idx; st->rank = rank; st->added = added ? 1u : 0u;
        st->seen = seen ? 1u : 0u; st->newrow = seen ? trow[trank] : nt;
        st->gamma = g; st->iter = round;
        if (hflags) __hip_atomic_store(&hflags[0], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- k_cs_update: online_column_inverse::insert / remove on the replicated state (online_inverse.h:183-293), the sign
// ---- vector and the new direction (homotopy-cpu.cpp:257-267).  Workgroup b < K_new forms u1 / the column's own dot
// ---- product from AS; the last to arrive does the rest.
__global__ __launch_bounds__(kUpdThreads)
void k_cs_update(uint32_t ldm, uint32_t col_lo, uint32_t n_local, uint32_t kcap, float tol, const float* __restrict__ xbuf,
                 float* AS, const uint32_t* gam2, const uint32_t* tch2, const uint32_t* trow2, float* ds2, float* inv,
                 float* u1, float* u2, float* sgn, float* __restrict__ d_loc, CsState* st)
{
    __shared__ float sv[16];
    __shared__ float s_d;
    __shared__ uint32_t s_flag;
    if (st->done) return;
    const uint32_t cur = st->cur, K_new = st->K, rank = st->rank, nt = st->ntouched;
    const bool added = st->added != 0u, seen = st->seen != 0u;
    const uint32_t newrow = st->newrow;
    const uint32_t* gam_old = gam2 + (size_t)cur * kcap;
    const uint32_t* gam_new = gam2 + (size_t)(cur ^ 1u) * kcap;
    const uint32_t* tch_new = tch2 + (size_t)(cur ^ 1u) * kcap;
    const uint32_t* trow_new = trow2 + (size_t)(cur ^ 1u) * kcap;
    const uint32_t K_old = added ? K_new - 1u : K_new + 1u;
    if (added && blockIdx.x < K_new) {
        const uint32_t b = blockIdx.x;
        // the entering column: from the exchange buffer (first entry) or from its row of AS (re-insertion)
        const float* cn = seen ? AS + (size_t)newrow * ldm : xbuf;
        // row of AS of support column b: look its column up in the NEW touched list
        const uint32_t col = gam_new[b];
        uint32_t lo = 0, hi = nt;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (tch_new[mid] < col) lo = mid + 1u; else hi = mid; }
        const float* cb = (b == rank) ? cn : AS + (size_t)trow_new[lo] * ldm;
        const float v = block_dot(cb, cn, ldm, sv);
        if (threadIdx.x == 0) {
            if (b == rank) st->dot = v;
            else u1[b - (b > rank ? 1u : 0u)] = v;
        }
        // every rank files the new column (the owner included): block `rank` copies it
        if (b == rank && !seen) for (uint32_t i = threadIdx.x; i < ldm; i += blockDim.x) AS[(size_t)newrow * ldm + i] = xbuf[i];
    }
    if (!arrive_last(&st->ticket_upd, gridDim.x, &s_flag)) return;

    const float* Iold = inv + (size_t)cur * kcap * kcap;
    float* Inew = inv + (size_t)(cur ^ 1u) * kcap * kcap;
    const size_t P = kcap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = (int)(blockDim.x >> 6);
    if (added) {
        const uint32_t nn = K_old;
        for (uint32_t i = wave; i < nn; i += NW) {                        // u2 = inv * u1 (online_inverse.h:224-225)
            float acc = 0.f;
            for (uint32_t j = lane; j < nn; j += 64) acc += Iold[i * P + j] * u1[j];
            acc = wave_sum(acc);
            if (lane == 0) u2[i] = acc;
        }
        __syncthreads();
        float part = 0.f;
        for (uint32_t j = threadIdx.x; j < nn; j += blockDim.x) part += u1[j] * u2[j];
        const float s = block_sum(part, sv);
        if (threadIdx.x == 0) s_d = 1.f / (__hip_atomic_load(&st->dot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - s);   // :228
        __syncthreads();
        const float dv = s_d;
        const uint32_t tot = K_new * K_new;
        for (uint32_t e = threadIdx.x; e < tot; e += blockDim.x) {        // :229-248, directly in sorted order
            const uint32_t a = e / K_new, b = e - a * K_new;
            float v;
            if (a == rank && b == rank) v = dv;
            else if (a == rank) v = -dv * u2[b - (b > rank ? 1u : 0u)];
            else if (b == rank) v = -dv * u2[a - (a > rank ? 1u : 0u)];
            else { const uint32_t oa = a - (a > rank ? 1u : 0u), ob = b - (b > rank ? 1u : 0u); v = Iold[oa * P + ob] + (dv * u2[oa]) * u2[ob]; }
            Inew[a * P + b] = v;
        }
    } else {
        const uint32_t nn = K_old;                                        // :275-290
        const float dd = Iold[rank * P + rank];
        const float sc = -(1.f / dd);
        for (uint32_t i = threadIdx.x; i < nn; i += blockDim.x) u2[i] = Iold[i * P + rank] * sc;
        __syncthreads();
        const uint32_t tot = K_new * K_new;
        for (uint32_t e = threadIdx.x; e < tot; e += blockDim.x) {
            const uint32_t a = e / K_new, b = e - a * K_new;
            const uint32_t oa = a + (a >= rank ? 1u : 0u), ob = b + (b >= rank ? 1u : 0u);
            Inew[a * P + b] = Iold[oa * P + ob] + (-dd * u2[oa]) * u2[ob];
        }
    }
    // sign(c - gamma q) on the new support with the dead zone (homotopy-cpu.cpp:259-260): the owners' values
    for (uint32_t a = threadIdx.x; a < K_new; a += blockDim.x) sgn[a] = sign_tol(xbuf[ldm + a], tol);
    // the old direction leaves the shard's dense vector
    for (uint32_t j = threadIdx.x; j < K_old; j += blockDim.x) {
        const uint32_t col = gam_old[j];
        if (col >= col_lo && col < col_lo + n_local) d_loc[col - col_lo] = 0.f;
    }
    __syncthreads();
    float* ds_new = ds2 + (size_t)(cur ^ 1u) * kcap;
    for (uint32_t a = wave; a < K_new; a += NW) {                          // direction = inv * sign (:263)
        float acc = 0.f;
        for (uint32_t b = lane; b < K_new; b += 64) acc += Inew[a * P + b] * sgn[b];
        acc = wave_sum(acc);
        if (lane == 0) {
            ds_new[a] = acc;
            const uint32_t col = gam_new[a];
            if (col >= col_lo && col < col_lo + n_local) d_loc[col - col_lo] = acc;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) st->cur = cur ^ 1u;
}

}  // namespace sship

using namespace sship;

namespace {

struct CsFail { std::string msg; };
#define CSHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw CsFail{ std::string("HIP error: ") + hipGetErrorString(e_) + " in " #expr }; } while (0)

// one all-reduce of `count` elements at `buf` (device), in place
void cs_allreduce(ss_hip_ctx* ctx, ColShard* cs, void* buf, size_t count, int dtype, int op)
{
    if (cs->comm != nullptr) {
        Rccl* r = rccl();
        const int rc = r->AllReduce(buf, buf, count, dtype, op, cs->comm, ctx->stream);
        if (rc != 0) throw CsFail{ std::string("RCCL all-reduce failed: ") + (r->GetErrorString ? r->GetErrorString(rc) : "?") };
        return;
    }
    if (!cs->have_host) return;                                            // world == 1 without a transport: nothing to do
    const size_t bytes = count * (dtype == kNcclUint64 ? 8 : 4);
    CSHIP(hipMemcpyAsync(cs->hstage, buf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CSHIP(hipStreamSynchronize(ctx->stream));
    int rc;
    if (dtype == kNcclUint64) rc = (op == kNcclMax ? cs->host.allreduce_max_u64 : cs->host.allreduce_min_u64)(cs->host.user, reinterpret_cast<uint64_t*>(cs->hstage), count);
    else rc = cs->host.allreduce_sum_f32(cs->host.user, reinterpret_cast<float*>(cs->hstage), count);
    if (rc != 0) throw CsFail{ "host collective failed" };
    CSHIP(hipMemcpyAsync(buf, cs->hstage, bytes, hipMemcpyHostToDevice, ctx->stream));
}

void cs_free(ColShard* cs)
{
    if (!cs) return;
    void* ptrs[] = { cs->gam, cs->tch, cs->trow, cs->xt, cs->ds, cs->inv, cs->AS, cs->u1, cs->u2, cs->sgn, cs->red, cs->xbuf };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (cs->hstage) (void)hipHostFree(cs->hstage);
    cs->gam = cs->tch = cs->trow = nullptr; cs->xt = cs->ds = cs->inv = cs->AS = cs->u1 = cs->u2 = cs->sgn = nullptr;
    cs->red = nullptr; cs->xbuf = nullptr; cs->hstage = nullptr; cs->kcap = 0;
}

void cs_ensure(ss_hip_ctx* ctx, ColShard* cs, uint32_t kcap)
{
    if (kcap <= cs->kcap) return;
    const uint32_t want = std::max<uint32_t>(kcap, std::min<uint32_t>(kKcapLimit, std::max<uint32_t>(64, cs->kcap * 2)));
    cs_free(cs);
    const size_t K = want, ldm = ctx->ldm;
    CSHIP(hipMalloc(&cs->gam, 2 * K * 4)); CSHIP(hipMalloc(&cs->tch, 2 * K * 4)); CSHIP(hipMalloc(&cs->trow, 2 * K * 4));
    CSHIP(hipMalloc(&cs->xt, 2 * K * 4)); CSHIP(hipMalloc(&cs->ds, 2 * K * 4));
    CSHIP(hipMalloc(&cs->inv, 2 * K * K * 4)); CSHIP(hipMalloc(&cs->AS, K * ldm * 4));
    CSHIP(hipMalloc(&cs->u1, K * 4)); CSHIP(hipMalloc(&cs->u2, K * 4)); CSHIP(hipMalloc(&cs->sgn, K * 4));
    CSHIP(hipMalloc(&cs->red, 64));
    cs->xcount = (uint32_t)((ldm + K + 7) / 8 * 8);
    CSHIP(hipMalloc(&cs->xbuf, (size_t)cs->xcount * 4));
    CSHIP(hipHostMalloc(reinterpret_cast<void**>(&cs->hstage), (size_t)cs->xcount * 4 + 64, hipHostMallocDefault));
    cs->kcap = want;
}

}  // namespace

namespace sship {
void colshard_destroy(ss_hip_ctx* ctx)
{
    ColShard* cs = static_cast<ColShard*>(ctx->colshard);
    if (!cs) return;
    if (cs->comm != nullptr) { Rccl* r = rccl(); if (r) (void)r->CommDestroy(cs->comm); }
    cs_free(cs);
    if (cs->agree_dev) (void)hipFree(cs->agree_dev);
    if (cs->agree_host) (void)hipHostFree(cs->agree_host);
    delete cs;
    ctx->colshard = nullptr;
}
}  // namespace sship

extern "C" {

int ss_hip_comm_unique_id(unsigned char* id, char* err, size_t errlen)
{
    if (!id) { set_err(err, errlen, "comm_unique_id: null argument"); return SS_HIP_EINVAL; }
    Rccl* r = rccl();
    if (!r) { set_err(err, errlen, "comm_unique_id: librccl.so could not be loaded"); return SS_HIP_ERUNTIME; }
    cs_unique_id u;
    std::memset(&u, 0, sizeof(u));
    const int rc = r->GetUniqueId(&u);
    if (rc != 0) { set_err(err, errlen, std::string("ncclGetUniqueId failed: ") + (r->GetErrorString ? r->GetErrorString(rc) : "?")); return SS_HIP_ERUNTIME; }
    std::memcpy(id, u.internal, SS_HIP_COMM_ID_BYTES);
    return SS_HIP_OK;
}

ss_hip_ctx* ss_hip_homotopy_colshard_create_f32(const float* A_local, size_t m, size_t n_local, ptrdiff_t stride_row,
                                                ptrdiff_t stride_col, size_t col_lo, size_t n_total, int device,
                                                const unsigned char* comm_id, int rank, int world,
                                                const ss_hip_collectives* host_collectives, char* err, size_t errlen)
{
    if (world < 1 || rank < 0 || rank >= world || n_total == 0 || col_lo + n_local > n_total || n_total > 0xfffffff0ull) {
        set_err(err, errlen, "colshard_create: bad shard description (rank / world / column range)");
        return nullptr;
    }
    if (world > 1 && comm_id == nullptr && host_collectives == nullptr) {
        set_err(err, errlen, "colshard_create: world > 1 needs a communicator id (RCCL) or host collectives");
        return nullptr;
    }
    if (host_collectives && comm_id == nullptr &&
        (!host_collectives->allreduce_max_u64 || !host_collectives->allreduce_min_u64 || !host_collectives->allreduce_sum_f32)) {
        set_err(err, errlen, "colshard_create: incomplete table of host collectives");
        return nullptr;
    }
    // a shard may be empty (more ranks than columns at the tail): it still takes part in every collective.  The context
    // needs at least one column to exist: an empty shard holds one zero column that never enters (|c| = 0, q = 0).
    static const float zero_col = 0.f;
    std::vector<float> zeros;
    const float* Aptr = A_local;
    size_t ncols = n_local;
    ptrdiff_t rs = stride_row, cs_ = stride_col;
    if (n_local == 0) { zeros.assign(m, 0.f); Aptr = zeros.data(); ncols = 1; rs = 1; cs_ = (ptrdiff_t)m; (void)zero_col; }
    ss_hip_ctx* ctx = ss_hip_homotopy_create_f32(Aptr, m, ncols, rs, cs_, device, err, errlen);
    if (!ctx) return nullptr;
    ColShard* cs = new (std::nothrow) ColShard();
    if (!cs) { set_err(err, errlen, "colshard_create: out of host memory"); ss_hip_homotopy_destroy(ctx); return nullptr; }
    ctx->colshard = cs;
    cs->col_lo = (uint32_t)col_lo; cs->n_total = (uint32_t)n_total; cs->rank = rank; cs->world = world;
    cs->n_local = (uint32_t)n_local;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&cs->agree_dev), 64) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&cs->agree_host), 64, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_err(err, errlen, "colshard_create: out of memory");
        ss_hip_homotopy_destroy(ctx);
        return nullptr;
    }
    if (comm_id != nullptr) {
        Rccl* r = rccl();
        if (!r) { set_err(err, errlen, "colshard_create: librccl.so could not be loaded"); ss_hip_homotopy_destroy(ctx); return nullptr; }
        cs_unique_id u;
        std::memcpy(u.internal, comm_id, SS_HIP_COMM_ID_BYTES);
        if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); }
        const int rc = r->CommInitRank(&cs->comm, world, u, rank);
        if (rc != 0) {
            set_err(err, errlen, std::string("ncclCommInitRank failed: ") + (r->GetErrorString ? r->GetErrorString(rc) : "?"));
            cs->comm = nullptr;
            ss_hip_homotopy_destroy(ctx);
            return nullptr;
        }
    } else if (host_collectives != nullptr) {
        cs->host = *host_collectives;
        cs->have_host = true;
    }
    return ctx;
}

int ss_hip_homotopy_colshard_solve_f32(ss_hip_ctx* ctx, const float* y, ptrdiff_t incy, float tol, uint32_t max_iter,
                                       float* x_local, ptrdiff_t incx, uint32_t* iter_out, double* err_out,
                                       char* err, size_t errlen)
{
    if (!ctx || !ctx->colshard) { set_err(err, errlen, "colshard_solve: not a column-sharded context"); return SS_HIP_EINVAL; }
    if (ctx->is_f64) { set_err(err, errlen, "colshard_solve: fp32 contexts only"); return SS_HIP_ETYPE; }
    ColShard* cs = static_cast<ColShard*>(ctx->colshard);
    if (!y || (!x_local && cs->n_local != 0)) { set_err(err, errlen, "colshard_solve: y and x must not be null"); return SS_HIP_EINVAL; }
    if (max_iter == 0) { set_err(err, errlen, "colshard_solve: max_iterations must be > 0"); return SS_HIP_EINVAL; }
    if (!(tol >= std::numeric_limits<float>::epsilon() && tol < 1.f)) { set_err(err, errlen, "colshard_solve: tolerance must satisfy eps <= tolerance < 1"); return SS_HIP_EINVAL; }
    if (incy <= 0 || incx <= 0) { set_err(err, errlen, "colshard_solve: vector increments must be positive"); return SS_HIP_EINVAL; }
    if (cs->dead) { set_err(err, errlen, "colshard_solve: the communicator of this context was aborted by an earlier failure"); return SS_HIP_ERUNTIME; }
    // ---- everything this solve allocates, BEFORE the first collective; then the ranks AGREE on having it.  The protocol below is
    // collective: a rank that left alone on a failed hipMalloc would leave the others blocked in their next all-reduce for good.
    const uint32_t kcap = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(cs->n_total, (uint64_t)max_iter + 1), kKcapLimit);
    const uint32_t want_trace = ctx->tracing ? (uint32_t)std::min<uint64_t>((uint64_t)max_iter + 2, 1u << 20) : 0u;
    int local = SS_HIP_OK;
    std::string local_msg;
    try {
        CSHIP(hipSetDevice(ctx->device));
        if (ctx->colshard_fail_prepare) throw CsFail{ "preparation failure requested (option colshard_fail_prepare)" };
        cs_ensure(ctx, cs, kcap);
        // the shard's own workspace: y, rhs (r, p), c, q, x, d, insup, sweep partials (homotopy.hip: Workspace<float>)
        const int rc0 = colshard_workspace(ctx, kcap);
        if (rc0 != SS_HIP_OK) throw CsFail{ "workspace allocation failed" };
        Workspace<float>& ws0 = *static_cast<Workspace<float>*>(ctx->ws);
        if (want_trace > ws0.trace_cap) {
            if (ws0.trace) CSHIP(hipFree(ws0.trace));
            ws0.trace = nullptr; ws0.trace_cap = 0;
            CSHIP(hipMalloc(&ws0.trace, (size_t)want_trace * sizeof(TraceEntry)));
            ws0.trace_cap = want_trace;
        }
    } catch (const CsFail& f) {
        (void)hipGetLastError();
        local = SS_HIP_ENOMEM;
        local_msg = f.msg;
    } catch (const std::bad_alloc&) {
        local = SS_HIP_ENOMEM;
        local_msg = "out of host memory";
    }
    {
        // max over the ranks of the local status (0 = ready): one 8-byte all-reduce per solve
        uint64_t agreed = (uint64_t)local;
        bool transport_failed = false;
        if (cs->comm != nullptr) {
            Rccl* r = rccl();
            *cs->agree_host = agreed;
            if (hipMemcpyAsync(cs->agree_dev, cs->agree_host, 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                r->AllReduce(cs->agree_dev, cs->agree_dev, 1, kNcclUint64, kNcclMax, cs->comm, ctx->stream) != 0 ||
                hipMemcpyAsync(cs->agree_host, cs->agree_dev, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); transport_failed = true; }
            else agreed = *cs->agree_host;
        } else if (cs->have_host) {
            if (cs->host.allreduce_max_u64(cs->host.user, &agreed, 1) != 0) transport_failed = true;
        }
        if (transport_failed) { set_err(err, errlen, "colshard_solve: the ranks could not agree on their preparation (collective failed)"); return SS_HIP_ERUNTIME; }
        if (agreed != 0) {
            set_err(err, errlen, local != SS_HIP_OK ? "colshard_solve: preparation failed on this rank: " + local_msg
                                                    : std::string("colshard_solve: preparation failed on another rank; every rank leaves the solve"));
            return (int)agreed;
        }
    }
    bool in_collectives = false;
    try {
        const uint32_t K = cs->kcap;
        Workspace<float>& ws = *static_cast<Workspace<float>*>(ctx->ws);
        hipStream_t st = ctx->stream;
        const uint32_t ldm = ctx->ldm, m = (uint32_t)ctx->m, nl = cs->n_local, np = ctx->n_pad;
        const float* At = static_cast<const float*>(ctx->At);
        CsState* dst = reinterpret_cast<CsState*>(ws.st);                  // (DevState is 640 bytes: room for the 128 of CsState)
        in_collectives = true;
        TraceEntry* trace = ctx->tracing ? ws.trace : nullptr;
        ctx->host_flags[0] = 0; ctx->host_flags[1] = 0;
        // y, zero padded; x, d, membership flags of the shard
        if (incy == 1) CSHIP(hipMemcpyAsync(ws.y, y, (size_t)m * 4, hipMemcpyDefault, st));
        else CSHIP(hipMemcpy2DAsync(ws.y, 4, y, (size_t)incy * 4, 4, m, hipMemcpyDefault, st));
        CSHIP(hipMemsetAsync(ws.x, 0, (size_t)np * 4, st));
        CSHIP(hipMemsetAsync(ws.d, 0, (size_t)np * 4, st));
        CSHIP(hipMemsetAsync(ws.insup, 0, (size_t)np, st));
        CSHIP(hipMemsetAsync(ws.st, 0, sizeof(DevState), st));
        CSHIP(hipMemcpyAsync(ws.rhs, ws.y, (size_t)ldm * 4, hipMemcpyDeviceToDevice, st));
        const size_t rhs_stride = (size_t)ws.dims.b_pad * ldm;
        float* r = ws.rhs;
        float* p = ws.rhs + rhs_stride;
        uint64_t* pmin = reinterpret_cast<uint64_t*>(ws.pmin_val);       // kMaxScanBlocks floats + as many indices: room for 256 words
        // c0 = A_loc^T y, lambda and the first pick over all shards, the first column to everyone
        uint32_t nb = 0;
        CSHIP(launch_sweep<float>(ctx, r, rhs_stride, 1, ws.c, nullptr, ws.pmax_val, ws.pmax_idx, &nb, nullptr));
        hipLaunchKernelGGL(k_cs_lmax, dim3(1), dim3(kCsThreads), 0, st, (const float*)ws.pmax_val, (const uint32_t*)ws.pmax_idx, nb,
                           cs->col_lo, nl, cs->red, (const CsState*)nullptr);
        cs_allreduce(ctx, cs, cs->red, 1, kNcclUint64, kNcclMax);
        hipLaunchKernelGGL(k_cs_init, dim3(1), dim3(kCsThreads), 0, st, At, ldm, cs->col_lo, nl, (const float*)ws.c, (const uint64_t*)cs->red,
                           cs->xbuf, cs->xcount, dst);
        cs_allreduce(ctx, cs, cs->xbuf, cs->xcount, kNcclFloat32, kNcclSum);
        hipLaunchKernelGGL(k_cs_first, dim3(1), dim3(kUpdThreads), 0, st, (const float*)cs->xbuf, ldm, cs->col_lo, nl, K, tol, ctx->strict_sign,
                           cs->AS, cs->gam, cs->tch, cs->trow, cs->xt, cs->ds, cs->inv, ws.d, ws.insup, dst, trace);
        CSHIP(hipGetLastError());
        const uint32_t rp_blocks = (ldm + kCsThreads - 1) / kCsThreads;
        uint32_t scan_blocks = std::max<uint32_t>(1u, std::min<uint32_t>((nl + kCsThreads * 4 - 1) / (kCsThreads * 4), kMaxScanBlocks / 2));
        // Rounds are enqueued in blocks of `block` and the replicated `done` flag is read back after each block: every
        // rank sees the same flag at the same round, so all ranks issue the same collectives.
        const uint32_t block = (uint32_t)std::max(1, std::min(ctx->lookahead * 2, 16));
        CsState hs{};
        uint64_t round = 1;
        const uint64_t last_round = (uint64_t)max_iter + 1;
        for (;;) {
            for (uint32_t b = 0; b < block && round <= last_round; ++b, ++round) {
                hipLaunchKernelGGL(k_cs_rp, dim3(rp_blocks), dim3(kCsThreads), 0, st, (const float*)cs->AS, ldm, m, (const float*)ws.y, K,
                                   (const uint32_t*)cs->tch, (const uint32_t*)cs->trow, (const float*)cs->xt, (const uint32_t*)cs->gam,
                                   (const float*)cs->ds, r, p, (const CsState*)dst);
                CSHIP(launch_sweep<float>(ctx, r, rhs_stride, 2, ws.c, ws.q, ws.pmax_val, ws.pmax_idx, &nb, nullptr));
                hipLaunchKernelGGL(k_cs_lmax, dim3(1), dim3(kCsThreads), 0, st, (const float*)ws.pmax_val, (const uint32_t*)ws.pmax_idx, nb,
                                   cs->col_lo, nl, cs->red, (const CsState*)dst);
                cs_allreduce(ctx, cs, cs->red, 1, kNcclUint64, kNcclMax);
                hipLaunchKernelGGL(k_cs_scan, dim3(scan_blocks), dim3(kCsThreads), 0, st, (uint32_t)round, tol, max_iter, nl, cs->col_lo,
                                   (const float*)ws.c, (const float*)ws.q, (const float*)ws.x, (const float*)ws.d, (const uint8_t*)ws.insup,
                                   cs->red, pmin, ctx->tie_guard, dst, ctx->dev_flags);
                cs_allreduce(ctx, cs, cs->red + 1, 1, kNcclUint64, kNcclMin);
                hipLaunchKernelGGL(k_cs_select, dim3(1), dim3(kUpdThreads), 0, st, (uint32_t)round, At, ldm, cs->col_lo, nl, K,
                                   (const uint64_t*)cs->red, (const float*)ws.c, (const float*)ws.q, ws.x, ws.insup,
                                   cs->gam, cs->tch, cs->trow, cs->xt, (const float*)cs->ds, cs->xbuf, cs->xcount, ctx->zero_on_removal, dst,
                                   ctx->dev_flags, trace, ws.trace_cap);
                cs_allreduce(ctx, cs, cs->xbuf, cs->xcount, kNcclFloat32, kNcclSum);
                uint32_t gb = (uint32_t)std::min<uint64_t>(round + 1, K);
                hipLaunchKernelGGL(k_cs_update, dim3(gb), dim3(kUpdThreads), 0, st, ldm, cs->col_lo, nl, K, tol, (const float*)cs->xbuf,
                                   cs->AS, (const uint32_t*)cs->gam, (const uint32_t*)cs->tch, (const uint32_t*)cs->trow, cs->ds, cs->inv,
                                   cs->u1, cs->u2, cs->sgn, ws.d, dst);
                CSHIP(hipGetLastError());
            }
            CSHIP(hipMemcpyAsync(&hs, dst, sizeof(CsState), hipMemcpyDeviceToHost, st));
            CSHIP(hipStreamSynchronize(st));
            if (hs.done || round > last_round) break;
        }
        if (!hs.done) { set_err(err, errlen, "colshard_solve: internal error, device loop did not terminate"); return SS_HIP_ERUNTIME; }
        if (hs.status != 0) { set_err(err, errlen, "colshard_solve: active set outgrew the workspace capacity"); return (int)hs.status; }
        if (x_local && nl != 0) {
            if (incx == 1) CSHIP(hipMemcpyAsync(x_local, ws.x, (size_t)nl * 4, hipMemcpyDefault, st));
            else CSHIP(hipMemcpy2DAsync(x_local, (size_t)incx * 4, ws.x, 4, 4, nl, hipMemcpyDefault, st));
        }
        CSHIP(hipStreamSynchronize(st));
        if (iter_out) *iter_out = hs.iter;
        if (err_out) *err_out = (double)hs.lambda;
        ctx->last_trace.clear();
        if (ctx->tracing && ws.trace) {
            const size_t cnt = std::min<size_t>((size_t)hs.iter + 1, ws.trace_cap);
            ctx->last_trace.resize(cnt);
            CSHIP(hipMemcpy(ctx->last_trace.data(), ws.trace, cnt * sizeof(TraceEntry), hipMemcpyDeviceToHost));
        }
        ctx->stats.solves += 1;
        ctx->stats.iterations += hs.iter;
    } catch (const CsFail& f) {
        // A failure in the middle of the protocol (a launch error, a collective that returned an error): this rank's communicator may
        // have a collective outstanding — it is ABORTED (ncclCommAbort), not destroyed later, and the context refuses further solves.
        // (The other ranks learn of it from their transport: RCCL reports the aborted peer, a host table returns non-zero.)
        (void)hipGetLastError();
        if (in_collectives && cs->comm != nullptr) {
            Rccl* r = rccl();
            if (r && r->CommAbort) (void)r->CommAbort(cs->comm);
            cs->comm = nullptr;
            cs->dead = true;
        }
        set_err(err, errlen, f.msg);
        return SS_HIP_ERUNTIME;
    } catch (const std::bad_alloc&) {
        set_err(err, errlen, "colshard_solve: out of host memory");
        return SS_HIP_ENOMEM;
    }
    return SS_HIP_OK;
}

}  // extern "C"
