// colshard.hip — ONE signal over a dictionary whose COLUMNS are split across the GPUs of a node:
// C-ABI ss_hip_homotopy_colshard_{create,solve}_f32 / _f64, one process per GPU, RCCL over xGMI.
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
// (fp64: a double and its index do not share a 64-bit word — each of the two reductions is a [world][2] table gathered by one MAX
//  all-reduce, CsRed<double> below.)
// Three small collectives per iteration (8 B, 8 B, (m + kcap) * 4 B; fp64: 16 B x world twice, (m + kcap) * 8 B): latency-bound on xGMI, which is why this form is
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
// rccl.h: ncclDataType_t ncclUint64 = 5, ncclFloat32 = 7, ncclFloat64 = 8; ncclRedOp_t ncclSum = 0, ncclMax = 2, ncclMin = 3
constexpr int kNcclUint64 = 5, kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0, kNcclMax = 2, kNcclMin = 3;

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
    ss_hip_collectives host{};            // fp32 contexts
    ss_hip_collectives_f64 host64{};      // fp64 contexts
    uint32_t esize = 4;                   // element size of the context (4 / 8)
    uint32_t kcap = 0;
    // replicated active set (device)
    uint32_t* gam = nullptr;        // [2][kcap] sorted support, global indices (ping-pong with inv)
    uint32_t* tch = nullptr;        // [2][kcap] sorted touched list
    uint32_t* trow = nullptr;       // [2][kcap] row of AS that holds each touched column
    // (element type of the context: float or double)
    void* xt = nullptr;             // [2][kcap] x on the touched columns (touched order)
    void* ds = nullptr;             // [2][kcap] direction on the support (support order)
    void* inv = nullptr;            // [2][kcap][kcap]
    void* AS = nullptr;             // [kcap][ldm] every column that was ever active, in order of first entry
    void* u1 = nullptr; void* u2 = nullptr; void* sgn = nullptr;     // [kcap]
    uint64_t* red = nullptr;        // exchange words of the two (value, index) reductions: [2][CsRed<T>::words(world)]
    uint64_t* pmin = nullptr;       // [kMaxScanBlocks][2] per-workgroup minima of the step-length scan
    void* xbuf = nullptr;           // [ldm + kcap + 8] exchange buffer: the entering column, then c - gamma q on the support
    uint32_t xcount = 0;            // its length in elements
    // pinned staging for host collectives
    unsigned char* hstage = nullptr;
    // the ranks' agreement on "everything this solve needs is allocated" (made at create: it must exist when an allocation fails)
    uint64_t* agree_dev = nullptr;  // [1] device word of the RCCL all-reduce
    uint64_t* agree_host = nullptr; // [1] pinned
    bool dead = false;              // the communicator was aborted in the middle of a solve: the context refuses further solves
};

template <typename T>
struct CsStateT {                    // device-resident, replicated scalars (one 128-byte line)
    uint32_t done, status, iter, K, ntouched, idx, rank, added, cur, seen, newrow, pad0;
    T lambda, gamma, lambda0, dot;
    uint32_t ticket_scan, ticket_upd, pad1[(128 - 48 - 4 * sizeof(T) - 8) / 4];
};
static_assert(sizeof(CsStateT<float>) == 128 && sizeof(CsStateT<double>) == 128, "CsState layout");

constexpr int kCsThreads = 256;
constexpr uint32_t kCsNone = 0xffffffffu;
constexpr uint32_t kCsMaxWorld = 64;            // (fp64: the reduction tables hold a pair of words per rank)

// ---- (value, index) reductions over the ranks --------------------------------------------------------------------------
// fp32: value and index share one ordered 64-bit word (bits(value) << 32 | index key) and the all-reduce is an exact MAX / MIN.
// fp64: they do not fit one word — every rank writes its pair into ITS slot of a [world][2] table and zeros everywhere else, ONE
// MAX all-reduce gathers the table (each slot has a single contributor, every word is >= 0) and every rank reduces the table the
// same way: largest value first, then the smallest global index — the same order as the packed word's.
template <typename T> struct CsRed;
template <> struct CsRed<float> {
    static constexpr bool gather = false;
    __host__ __device__ static uint32_t words(uint32_t) { return 1u; }
    __device__ static uint64_t bits(float v) { return (uint64_t)__float_as_uint(v); }
    __device__ static void put_max(uint64_t* red, uint32_t, uint32_t, bool has, float v, uint32_t gidx)
    {
        red[0] = has ? ((bits(v) << 32) | (uint64_t)(~gidx)) : 0ull;
    }
    __device__ static void get_max(const uint64_t* red, uint32_t, float& v, uint32_t& gidx)
    {
        const uint64_t w = red[0];
        gidx = ~(uint32_t)w;
        v = __uint_as_float((uint32_t)(w >> 32));
    }
    __device__ static void put_min(uint64_t* red, uint32_t, uint32_t, uint64_t vbits, uint32_t gidx) { red[0] = (vbits << 32) | (uint64_t)gidx; }
    __device__ static void put_min_end(uint64_t* red, uint32_t, uint32_t) { red[0] = ~0ull; }
    __device__ static void get_min(const uint64_t* red, uint32_t, float& t, uint32_t& gidx)
    {
        const uint64_t w = red[0];
        t = __uint_as_float((uint32_t)(w >> 32));
        gidx = (uint32_t)w;
    }
};
template <> struct CsRed<double> {
    static constexpr bool gather = true;
    __host__ __device__ static uint32_t words(uint32_t world) { return 2u * world; }
    __device__ static uint64_t bits(double v) { return (uint64_t)__double_as_longlong(v); }
    __device__ static void clear(uint64_t* red, uint32_t world) { for (uint32_t r = 0; r < 2u * world; ++r) red[r] = 0ull; }
    __device__ static void put_max(uint64_t* red, uint32_t rank, uint32_t world, bool has, double v, uint32_t gidx)
    {
        clear(red, world);
        if (has) { red[2u * rank] = bits(v); red[2u * rank + 1u] = (uint64_t)(~gidx); }
    }
    __device__ static void get_max(const uint64_t* red, uint32_t world, double& v, uint32_t& gidx)
    {
        uint64_t bv = 0ull, bk = 0ull;
        for (uint32_t r = 0; r < world; ++r) {
            const uint64_t a = red[2u * r], k = red[2u * r + 1u];
            if (a > bv || (a == bv && k > bk)) { bv = a; bk = k; }
        }
        gidx = ~(uint32_t)bk;
        v = __longlong_as_double((long long)bv);
    }
    __device__ static void put_min(uint64_t* red, uint32_t rank, uint32_t world, uint64_t vbits, uint32_t gidx)
    {
        clear(red, world);
        red[2u * rank] = vbits; red[2u * rank + 1u] = (uint64_t)gidx;
    }
    __device__ static void put_min_end(uint64_t* red, uint32_t rank, uint32_t world) { put_min(red, rank, world, (uint64_t)0x7ff0000000000000ull, 0u); }
    __device__ static void get_min(const uint64_t* red, uint32_t world, double& t, uint32_t& gidx)
    {
        uint64_t bv = ~0ull, bi = ~0ull;
        for (uint32_t r = 0; r < world; ++r) {
            const uint64_t a = red[2u * r], k = red[2u * r + 1u];
            if (a < bv || (a == bv && k < bi)) { bv = a; bi = k; }
        }
        gidx = (uint32_t)bi;
        t = __longlong_as_double((long long)bv);
    }
};

// ---- k_cs_rp: r = y - sum_t xt[t] AS[trow[t]] ; p = sum_j ds[j] AS[row(gam[j])] --------------------------------------
// one thread per row, touched order (sorted by global column): the same arithmetic on every rank
template <typename T>
__global__ __launch_bounds__(kCsThreads)
void k_cs_rp(const T* __restrict__ AS, uint32_t ldm, uint32_t m, const T* __restrict__ y, uint32_t kcap,
             const uint32_t* __restrict__ tch2, const uint32_t* __restrict__ trow2, const T* __restrict__ xt2,
             const uint32_t* __restrict__ gam2, const T* __restrict__ ds2, T* __restrict__ r, T* __restrict__ p,
             const CsStateT<T>* st)
{
    if (st->done) return;
    const uint32_t i = blockIdx.x * kCsThreads + threadIdx.x;
    if (i >= ldm) return;
    const uint32_t cur = st->cur, nt = st->ntouched, K = st->K;
    const uint32_t* tch = tch2 + (size_t)cur * kcap;
    const uint32_t* trow = trow2 + (size_t)cur * kcap;
    const T* xt = xt2 + (size_t)cur * kcap;
    const uint32_t* gam = gam2 + (size_t)cur * kcap;
    const T* ds = ds2 + (size_t)cur * kcap;
    T ar = T(0), ap = T(0);
    uint32_t j = 0;                                        // position in the support (a sub-sequence of the touched list)
    for (uint32_t t = 0; t < nt; ++t) {
        const T a = AS[(size_t)trow[t] * ldm + i];
        ar += xt[t] * a;
        if (j < K && gam[j] == tch[t]) { ap += ds[j] * a; ++j; }
    }
    r[i] = i < m ? y[i] - ar : T(0);
    p[i] = i < m ? ap : T(0);
}

// ---- k_cs_lmax: the shard's (max |c|, first index) as one ordered word ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kCsThreads)
void k_cs_lmax(const T* __restrict__ pmax_val, const uint32_t* __restrict__ pmax_idx, uint32_t nb, uint32_t col_lo,
               uint32_t n_local, uint64_t* __restrict__ red, uint32_t rank_id, uint32_t world, const CsStateT<T>* st)
{
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    T v;
    uint32_t ix;
    reduce_sweep_partials(pmax_val, pmax_idx, nb, v, ix, sv, si);
    if (threadIdx.x == 0) {
        // (a shard without columns offers nothing; larger |c| first, then the smaller global index: ~index)
        const bool has = n_local != 0u && v >= T(0) && !(st != nullptr && st->done);
        CsRed<T>::put_max(red, rank_id, world, has, v, col_lo + ix);
    }
}

// ---- k_cs_init: first pick (homotopy-cpu.cpp:217-221) from the reduced word; the owner offers its column ---------------
template <typename T>
__global__ __launch_bounds__(kCsThreads)
void k_cs_init(const T* __restrict__ At, uint32_t ldm, uint32_t col_lo, uint32_t n_local, const T* __restrict__ c0,
               const uint64_t* __restrict__ red, uint32_t world, T* __restrict__ xbuf, uint32_t xcount, CsStateT<T>* st)
{
    uint32_t idx;
    T lam;
    CsRed<T>::get_max(red, world, lam, idx);
    const bool mine = idx >= col_lo && idx < col_lo + n_local;
    for (uint32_t i = threadIdx.x; i < xcount; i += blockDim.x) {
        T v = T(0);
        if (mine && i < ldm) v = At[(size_t)(idx - col_lo) * ldm + i];
        if (mine && i == ldm) v = c0[idx - col_lo];                       // c0[idx] (option strict_sign seeds the first sign with it)
        xbuf[i] = v;
    }
    if (threadIdx.x == 0) {
        st->done = 0; st->status = 0; st->iter = 0; st->K = 0; st->ntouched = 0; st->idx = idx; st->rank = 0; st->added = 1;
        st->cur = 0; st->seen = 0; st->newrow = 0; st->lambda = lam; st->gamma = T(0); st->lambda0 = lam;
        st->ticket_scan = 0; st->ticket_upd = 0;
    }
}

// ---- k_cs_first: AS[0] = the first column; inv = [1 / ||a||^2] through the norm (online_inverse.h:193-201); first direction
template <typename T>
__global__ __launch_bounds__(kUpdThreads)
void k_cs_first(const T* __restrict__ xbuf, uint32_t ldm, uint32_t col_lo, uint32_t n_local, uint32_t kcap, T tol,
                int strict_sign, T* __restrict__ AS, uint32_t* gam, uint32_t* tch, uint32_t* trow, T* xt, T* ds,
                T* inv, T* __restrict__ d_loc, uint8_t* __restrict__ insup, CsStateT<T>* st, TraceEntry* trace)
{
    __shared__ T sv[16];
    for (uint32_t i = threadIdx.x; i < ldm; i += blockDim.x) AS[i] = xbuf[i];
    __syncthreads();
    T acc = T(0);
    for (uint32_t i = threadIdx.x; i < ldm; i += blockDim.x) { const T a = xbuf[i]; acc += a * a; }
    const T dot = block_sum(acc, sv);
    if (threadIdx.x == 0) {
        const uint32_t idx = st->idx;
        const T nrm = sqrt(dot);
        const T inv00 = T(1) / (nrm * nrm);
        const T lam = st->lambda;
        const T seed = strict_sign ? xbuf[ldm] : lam;                 // first-step quirk (homotopy-cpu.cpp:223-227)
        const T d0 = sign_tol(seed, tol) * inv00;
        gam[0] = idx; tch[0] = idx; trow[0] = 0u; xt[0] = T(0); ds[0] = d0; inv[0] = inv00;
        if (idx >= col_lo && idx < col_lo + n_local) { d_loc[idx - col_lo] = d0; insup[idx - col_lo] = 1; }
        st->K = 1; st->ntouched = 1; st->cur = 0;
        if (trace != nullptr) { trace[0].idx = idx; trace[0].added = 1; trace[0].gamma = 0.0; trace[0].c_inf = (double)lam; }
        (void)kcap;
    }
}

// ---- k_cs_scan: loop control and find_max_gamma's scan (homotopy-cpu.cpp:122-163) over the shard's columns -------------
template <typename T>
__global__ __launch_bounds__(kCsThreads)
void k_cs_scan(uint32_t round, T tol, uint32_t max_iter, uint32_t n_local, uint32_t col_lo,
               const T* __restrict__ c, const T* __restrict__ q, const T* __restrict__ x, const T* __restrict__ d,
               const uint8_t* __restrict__ insup, const uint64_t* red_max, uint64_t* red_min, uint32_t rank_id, uint32_t world,
               uint64_t* pmin, int tie_guard, CsStateT<T>* st, uint32_t* hflags)
{
    __shared__ T sv[16];
    __shared__ uint32_t si[16];
    __shared__ uint32_t s_flag;
    if (st->done) return;
    T c_inf;
    { uint32_t ix_unused; CsRed<T>::get_max(red_max, world, c_inf, ix_unused); }
    // do { ... } while (iter < max_iter && c_inf > tolerance): the test of iteration round-1; every rank reads the same word
    if ((round > 1 && !(c_inf > tol)) || round > max_iter) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->lambda = c_inf;
            st->iter = round - 1;
            st->done = 1;
            CsRed<T>::put_min_end(red_min, rank_id, world);
            if (hflags) { __hip_atomic_store(&hflags[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&hflags[0], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
        return;
    }
    T best = Lim<T>::max();
    uint32_t best_i = kCsNone;
    for (uint32_t i = blockIdx.x * kCsThreads + threadIdx.x; i < n_local; i += gridDim.x * kCsThreads) {
        T m = Lim<T>::max();
        if (insup[i]) {
            const T t = -x[i] / d[i];
            if (t > T(0) && t < m) m = t;
        } else {
            const T qi = q[i], ci = c[i];
            const T dl = T(1) - qi, dr = T(1) + qi;
            if (dl != T(0)) {
                T t = (c_inf - ci) / dl;
                if (tie_guard && t == T(0) && dl > T(0)) t = Lim<T>::tiny();
                if (t > T(0) && t < m) m = t;
            }
            if (dr != T(0)) {
                T t = (c_inf + ci) / dr;
                if (tie_guard && t == T(0) && dr > T(0)) t = Lim<T>::tiny();
                if (t > T(0) && t < m) m = t;
            }
        }
        if (better_min(m, col_lo + i, best, best_i)) { best = m; best_i = col_lo + i; }
    }
    block_reduce_pair<T, false>(best, best_i, sv, si);
    if (threadIdx.x == 0) {
        // positive values order like their bit patterns; no candidate: (max, 0) as in the reference (:123-124)
        const bool has = best < Lim<T>::max();
        __hip_atomic_store(&pmin[2u * blockIdx.x], CsRed<T>::bits(has ? best : Lim<T>::max()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pmin[2u * blockIdx.x + 1u], (uint64_t)(has ? best_i : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!arrive_last_relaxed(&st->ticket_scan, gridDim.x, &s_flag)) return;
    uint64_t mv = ~0ull, mi = ~0ull;
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += blockDim.x) {
        const uint64_t v = __hip_atomic_load(&pmin[2u * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t k = __hip_atomic_load(&pmin[2u * b + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < mv || (v == mv && k < mi)) { mv = v; mi = k; }
    }
    // block minimum of the (value bits, index) pairs: value first, then index
    __shared__ uint64_t s_v[kCsThreads];
    __shared__ uint64_t s_i[kCsThreads];
    s_v[threadIdx.x] = mv; s_i[threadIdx.x] = mi;
    __syncthreads();
    for (uint32_t o = kCsThreads / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const uint64_t a = s_v[threadIdx.x], b2 = s_v[threadIdx.x + o], ka = s_i[threadIdx.x], kb = s_i[threadIdx.x + o];
            if (b2 < a || (b2 == a && kb < ka)) { s_v[threadIdx.x] = b2; s_i[threadIdx.x] = kb; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { CsRed<T>::put_min(red_min, rank_id, world, s_v[0], (uint32_t)s_i[0]); st->lambda = c_inf; }
}

// ---- k_cs_select: the pick, the toggle of the replicated lists and the x update (homotopy-cpu.cpp:246-252) --------------
// one workgroup per rank, the same on every rank; the owners fill the exchange buffer
template <typename T>
__global__ __launch_bounds__(kUpdThreads)
void k_cs_select(uint32_t round, const T* __restrict__ At, uint32_t ldm, uint32_t col_lo, uint32_t n_local, uint32_t kcap,
                 const uint64_t* __restrict__ red_min, uint32_t world, const T* __restrict__ c, const T* __restrict__ q,
                 T* __restrict__ x_loc, uint8_t* __restrict__ insup,
                 uint32_t* gam2, uint32_t* tch2, uint32_t* trow2, T* xt2, const T* ds2,
                 T* __restrict__ xbuf, uint32_t xcount, int zero_on_removal, CsStateT<T>* st, uint32_t* hflags,
                 TraceEntry* trace, uint32_t trace_cap)
{
    __shared__ uint32_t s_cnt[2];
    if (st->done) { for (uint32_t i = threadIdx.x; i < xcount; i += blockDim.x) xbuf[i] = T(0); return; }
    T g;
    uint32_t idx;
    CsRed<T>::get_min(red_min, world, g, idx);
    const uint32_t cur = st->cur, K = st->K, nt = st->ntouched;
    const uint32_t* gam = gam2 + (size_t)cur * kcap;
    uint32_t* gam_new = gam2 + (size_t)(cur ^ 1u) * kcap;
    const uint32_t* tch = tch2 + (size_t)cur * kcap;
    uint32_t* tch_new = tch2 + (size_t)(cur ^ 1u) * kcap;
    const uint32_t* trow = trow2 + (size_t)cur * kcap;
    uint32_t* trow_new = trow2 + (size_t)(cur ^ 1u) * kcap;
    const T* xt = xt2 + (size_t)cur * kcap;
    T* xt_new = xt2 + (size_t)(cur ^ 1u) * kcap;
    const T* ds = ds2 + (size_t)cur * kcap;
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
        for (uint32_t i = threadIdx.x; i < xcount; i += blockDim.x) xbuf[i] = T(0);
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
        T xv;
        if (ot == kCsNone) { col = idx; row = nt; xv = T(0); }
        else { col = tch[ot]; row = trow[ot]; xv = xt[ot]; }
        // direction of this column under the OLD support (binary search in gam)
        uint32_t lo = 0, hi = K;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (gam[mid] < col) lo = mid + 1u; else hi = mid; }
        if (lo < K && gam[lo] == col) {
            const T xn = xv + g * ds[lo];
            xv = (!added && zero_on_removal && col == idx) ? T(0) : xn;
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
    for (uint32_t i = threadIdx.x; i < ldm; i += blockDim.x) xbuf[i] = send_col ? At[(size_t)(idx - col_lo) * ldm + i] : T(0);
    for (uint32_t a = threadIdx.x; a < kcap; a += blockDim.x) {
        T v = T(0);
        if (a < K_new) {
            const uint32_t col = a < rank ? gam[a] : (added ? (a == rank ? idx : gam[a - 1u]) : gam[a + 1u]);
            if (col >= col_lo && col < col_lo + n_local) v = c[col - col_lo] - g * q[col - col_lo];
        }
        xbuf[ldm + a] = v;
    }
    for (uint32_t i = ldm + kcap + threadIdx.x; i < xcount; i += blockDim.x) xbuf[i] = T(0);
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
template <typename T>
__global__ __launch_bounds__(kUpdThreads)
void k_cs_update(uint32_t ldm, uint32_t col_lo, uint32_t n_local, uint32_t kcap, T tol, const T* __restrict__ xbuf,
                 T* AS, const uint32_t* gam2, const uint32_t* tch2, const uint32_t* trow2, T* ds2, T* inv,
                 T* u1, T* u2, T* sgn, T* __restrict__ d_loc, CsStateT<T>* st)
{
    __shared__ T sv[16];
    __shared__ T s_d;
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
        const T* cn = seen ? AS + (size_t)newrow * ldm : xbuf;
        // row of AS of support column b: look its column up in the NEW touched list
        const uint32_t col = gam_new[b];
        uint32_t lo = 0, hi = nt;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (tch_new[mid] < col) lo = mid + 1u; else hi = mid; }
        const T* cb = (b == rank) ? cn : AS + (size_t)trow_new[lo] * ldm;
        const T v = block_dot(cb, cn, ldm, sv);
        if (threadIdx.x == 0) {
            if (b == rank) st->dot = v;
            else u1[b - (b > rank ? 1u : 0u)] = v;
        }
        // every rank files the new column (the owner included): block `rank` copies it
        if (b == rank && !seen) for (uint32_t i = threadIdx.x; i < ldm; i += blockDim.x) AS[(size_t)newrow * ldm + i] = xbuf[i];
    }
    if (!arrive_last(&st->ticket_upd, gridDim.x, &s_flag)) return;

    const T* Iold = inv + (size_t)cur * kcap * kcap;
    T* Inew = inv + (size_t)(cur ^ 1u) * kcap * kcap;
    const size_t P = kcap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = (int)(blockDim.x >> 6);
    if (added) {
        const uint32_t nn = K_old;
        for (uint32_t i = wave; i < nn; i += NW) {                        // u2 = inv * u1 (online_inverse.h:224-225)
            T acc = T(0);
            for (uint32_t j = lane; j < nn; j += 64) acc += Iold[i * P + j] * u1[j];
            acc = wave_sum(acc);
            if (lane == 0) u2[i] = acc;
        }
        __syncthreads();
        T part = T(0);
        for (uint32_t j = threadIdx.x; j < nn; j += blockDim.x) part += u1[j] * u2[j];
        const T s = block_sum(part, sv);
        if (threadIdx.x == 0) s_d = T(1) / (__hip_atomic_load(&st->dot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - s);   // :228
        __syncthreads();
        const T dv = s_d;
        const uint32_t tot = K_new * K_new;
        for (uint32_t e = threadIdx.x; e < tot; e += blockDim.x) {        // :229-248, directly in sorted order
            const uint32_t a = e / K_new, b = e - a * K_new;
            T v;
            if (a == rank && b == rank) v = dv;
            else if (a == rank) v = -dv * u2[b - (b > rank ? 1u : 0u)];
            else if (b == rank) v = -dv * u2[a - (a > rank ? 1u : 0u)];
            else { const uint32_t oa = a - (a > rank ? 1u : 0u), ob = b - (b > rank ? 1u : 0u); v = Iold[oa * P + ob] + (dv * u2[oa]) * u2[ob]; }
            Inew[a * P + b] = v;
        }
    } else {
        const uint32_t nn = K_old;                                        // :275-290
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
    // sign(c - gamma q) on the new support with the dead zone (homotopy-cpu.cpp:259-260): the owners' values
    for (uint32_t a = threadIdx.x; a < K_new; a += blockDim.x) sgn[a] = sign_tol(xbuf[ldm + a], tol);
    // the old direction leaves the shard's dense vector
    for (uint32_t j = threadIdx.x; j < K_old; j += blockDim.x) {
        const uint32_t col = gam_old[j];
        if (col >= col_lo && col < col_lo + n_local) d_loc[col - col_lo] = T(0);
    }
    __syncthreads();
    T* ds_new = ds2 + (size_t)(cur ^ 1u) * kcap;
    for (uint32_t a = wave; a < K_new; a += NW) {                          // direction = inv * sign (:263)
        T acc = T(0);
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
    const size_t bytes = count * (dtype == kNcclFloat32 ? 4 : 8);
    CSHIP(hipMemcpyAsync(cs->hstage, buf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CSHIP(hipStreamSynchronize(ctx->stream));
    int rc;
    if (cs->esize == 8) {
        // (fp64 contexts: the reductions are gathers by MAX, the exchange buffer a sum of doubles)
        if (dtype == kNcclUint64) rc = cs->host64.allreduce_max_u64(cs->host64.user, reinterpret_cast<uint64_t*>(cs->hstage), count);
        else rc = cs->host64.allreduce_sum_f64(cs->host64.user, reinterpret_cast<double*>(cs->hstage), count);
    } else if (dtype == kNcclUint64) rc = (op == kNcclMax ? cs->host.allreduce_max_u64 : cs->host.allreduce_min_u64)(cs->host.user, reinterpret_cast<uint64_t*>(cs->hstage), count);
    else rc = cs->host.allreduce_sum_f32(cs->host.user, reinterpret_cast<float*>(cs->hstage), count);
    if (rc != 0) throw CsFail{ "host collective failed" };
    CSHIP(hipMemcpyAsync(buf, cs->hstage, bytes, hipMemcpyHostToDevice, ctx->stream));
}

void cs_free(ColShard* cs)
{
    if (!cs) return;
    void* ptrs[] = { cs->gam, cs->tch, cs->trow, cs->xt, cs->ds, cs->inv, cs->AS, cs->u1, cs->u2, cs->sgn, cs->red, cs->pmin, cs->xbuf };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (cs->hstage) (void)hipHostFree(cs->hstage);
    cs->gam = cs->tch = cs->trow = nullptr; cs->xt = cs->ds = cs->inv = cs->AS = cs->u1 = cs->u2 = cs->sgn = nullptr;
    cs->red = nullptr; cs->pmin = nullptr; cs->xbuf = nullptr; cs->hstage = nullptr; cs->kcap = 0;
}

void cs_ensure(ss_hip_ctx* ctx, ColShard* cs, uint32_t kcap)
{
    if (kcap <= cs->kcap) return;
    const uint32_t want = std::max<uint32_t>(kcap, std::min<uint32_t>(kKcapLimit, std::max<uint32_t>(64, cs->kcap * 2)));
    cs_free(cs);
    const size_t K = want, ldm = ctx->ldm, es = cs->esize;
    CSHIP(hipMalloc(&cs->gam, 2 * K * 4)); CSHIP(hipMalloc(&cs->tch, 2 * K * 4)); CSHIP(hipMalloc(&cs->trow, 2 * K * 4));
    CSHIP(hipMalloc(&cs->xt, 2 * K * es)); CSHIP(hipMalloc(&cs->ds, 2 * K * es));
    CSHIP(hipMalloc(&cs->inv, 2 * K * K * es)); CSHIP(hipMalloc(&cs->AS, K * ldm * es));
    CSHIP(hipMalloc(&cs->u1, K * es)); CSHIP(hipMalloc(&cs->u2, K * es)); CSHIP(hipMalloc(&cs->sgn, K * es));
    const size_t red_bytes = 2 * (size_t)(2 * kCsMaxWorld) * 8;
    CSHIP(hipMalloc(&cs->red, red_bytes));
    CSHIP(hipMemsetAsync(cs->red, 0, red_bytes, ctx->stream));
    CSHIP(hipMalloc(&cs->pmin, (size_t)kMaxScanBlocks * 2 * 8));
    cs->xcount = (uint32_t)((ldm + K + 7) / 8 * 8);
    CSHIP(hipMalloc(&cs->xbuf, (size_t)cs->xcount * es));
    CSHIP(hipHostMalloc(reinterpret_cast<void**>(&cs->hstage), std::max((size_t)cs->xcount * es, red_bytes) + 64, hipHostMallocDefault));
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

namespace {

inline bool table_complete(const ss_hip_collectives* t) { return t->allreduce_max_u64 && t->allreduce_min_u64 && t->allreduce_sum_f32; }
inline bool table_complete(const ss_hip_collectives_f64* t) { return t->allreduce_max_u64 && t->allreduce_sum_f64; }
inline void set_table(ColShard* cs, const ss_hip_collectives* t) { cs->host = *t; }
inline void set_table(ColShard* cs, const ss_hip_collectives_f64* t) { cs->host64 = *t; }
inline ss_hip_ctx* create_ctx(const float* A, size_t m, size_t n, ptrdiff_t rs, ptrdiff_t cs, int device, char* err, size_t errlen)
{
    return ss_hip_homotopy_create_f32(A, m, n, rs, cs, device, err, errlen);
}
inline ss_hip_ctx* create_ctx(const double* A, size_t m, size_t n, ptrdiff_t rs, ptrdiff_t cs, int device, char* err, size_t errlen)
{
    return ss_hip_homotopy_create_f64(A, m, n, rs, cs, device, err, errlen);
}

template <typename T, typename Table>
ss_hip_ctx* colshard_create_impl(const T* A_local, size_t m, size_t n_local, ptrdiff_t stride_row,
                                 ptrdiff_t stride_col, size_t col_lo, size_t n_total, int device,
                                 const unsigned char* comm_id, int rank, int world,
                                 const Table* host_collectives, char* err, size_t errlen)
{
    if (world < 1 || rank < 0 || rank >= world || n_total == 0 || col_lo + n_local > n_total || n_total > 0xfffffff0ull) {
        set_err(err, errlen, "colshard_create: bad shard description (rank / world / column range)");
        return nullptr;
    }
    if (world > 1 && comm_id == nullptr && host_collectives == nullptr) {
        set_err(err, errlen, "colshard_create: world > 1 needs a communicator id (RCCL) or host collectives");
        return nullptr;
    }
    if (host_collectives && comm_id == nullptr && !table_complete(host_collectives)) {
        set_err(err, errlen, "colshard_create: incomplete table of host collectives");
        return nullptr;
    }
    if (sizeof(T) == 8 && (uint32_t)world > kCsMaxWorld) {
        set_err(err, errlen, "colshard_create: fp64 contexts take at most 64 ranks");
        return nullptr;
    }
    // a shard may be empty (more ranks than columns at the tail): it still takes part in every collective.  The context
    // needs at least one column to exist: an empty shard holds one zero column that never enters (|c| = 0, q = 0).
    std::vector<T> zeros;
    const T* Aptr = A_local;
    size_t ncols = n_local;
    ptrdiff_t rs = stride_row, cs_ = stride_col;
    if (n_local == 0) { zeros.assign(m, T(0)); Aptr = zeros.data(); ncols = 1; rs = 1; cs_ = (ptrdiff_t)m; }
    ss_hip_ctx* ctx = create_ctx(Aptr, m, ncols, rs, cs_, device, err, errlen);
    if (!ctx) return nullptr;
    ColShard* cs = new (std::nothrow) ColShard();
    if (!cs) { set_err(err, errlen, "colshard_create: out of host memory"); ss_hip_homotopy_destroy(ctx); return nullptr; }
    ctx->colshard = cs;
    cs->col_lo = (uint32_t)col_lo; cs->n_total = (uint32_t)n_total; cs->rank = rank; cs->world = world;
    cs->n_local = (uint32_t)n_local;
    cs->esize = (uint32_t)sizeof(T);
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
        set_table(cs, host_collectives);
        cs->have_host = true;
    }
    return ctx;
}

template <typename T>
int colshard_solve_impl(ss_hip_ctx* ctx, const T* y, ptrdiff_t incy, T tol, uint32_t max_iter,
                        T* x_local, ptrdiff_t incx, uint32_t* iter_out, double* err_out,
                        char* err, size_t errlen)
{
    if (!ctx || !ctx->colshard) { set_err(err, errlen, "colshard_solve: not a column-sharded context"); return SS_HIP_EINVAL; }
    if (ctx->is_f64 != (sizeof(T) == 8)) { set_err(err, errlen, "colshard_solve: element type of the call does not match the context"); return SS_HIP_ETYPE; }
    ColShard* cs = static_cast<ColShard*>(ctx->colshard);
    if (!y || (!x_local && cs->n_local != 0)) { set_err(err, errlen, "colshard_solve: y and x must not be null"); return SS_HIP_EINVAL; }
    if (max_iter == 0) { set_err(err, errlen, "colshard_solve: max_iterations must be > 0"); return SS_HIP_EINVAL; }
    if (!(tol >= std::numeric_limits<T>::epsilon() && tol < T(1))) { set_err(err, errlen, "colshard_solve: tolerance must satisfy eps <= tolerance < 1"); return SS_HIP_EINVAL; }
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
        // the shard's own workspace: y, rhs (r, p), c, q, x, d, insup, sweep partials (homotopy.hip: Workspace<T>)
        const int rc0 = colshard_workspace(ctx, kcap);
        if (rc0 != SS_HIP_OK) throw CsFail{ "workspace allocation failed" };
        Workspace<T>& ws0 = *static_cast<Workspace<T>*>(ctx->ws);
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
            const int rc_a = cs->esize == 8 ? cs->host64.allreduce_max_u64(cs->host64.user, &agreed, 1)
                                            : cs->host.allreduce_max_u64(cs->host.user, &agreed, 1);
            if (rc_a != 0) transport_failed = true;
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
        Workspace<T>& ws = *static_cast<Workspace<T>*>(ctx->ws);
        hipStream_t st = ctx->stream;
        const uint32_t ldm = ctx->ldm, m = (uint32_t)ctx->m, nl = cs->n_local, np = ctx->n_pad;
        const T* At = static_cast<const T*>(ctx->At);
        CsStateT<T>* dst = reinterpret_cast<CsStateT<T>*>(ws.st);                  // (DevState is 640 bytes: room for the 128 of CsState)
        in_collectives = true;
        TraceEntry* trace = ctx->tracing ? ws.trace : nullptr;
        ctx->host_flags[0] = 0; ctx->host_flags[1] = 0;
        // y, zero padded; x, d, membership flags of the shard
        if (incy == 1) CSHIP(hipMemcpyAsync(ws.y, y, (size_t)m * sizeof(T), hipMemcpyDefault, st));
        else CSHIP(hipMemcpy2DAsync(ws.y, sizeof(T), y, (size_t)incy * sizeof(T), sizeof(T), m, hipMemcpyDefault, st));
        CSHIP(hipMemsetAsync(ws.x, 0, (size_t)np * sizeof(T), st));
        CSHIP(hipMemsetAsync(ws.d, 0, (size_t)np * sizeof(T), st));
        CSHIP(hipMemsetAsync(ws.insup, 0, (size_t)np, st));
        CSHIP(hipMemsetAsync(ws.st, 0, sizeof(DevState), st));
        CSHIP(hipMemcpyAsync(ws.rhs, ws.y, (size_t)ldm * sizeof(T), hipMemcpyDeviceToDevice, st));
        const size_t rhs_stride = (size_t)ws.dims.b_pad * ldm;
        T* r = ws.rhs;
        T* p = ws.rhs + rhs_stride;
        uint64_t* pmin = cs->pmin;
        // the two (value, index) reductions: one packed word each in fp32 (MAX / MIN), a gathered [world][2] table each in fp64 (MAX)
        const uint32_t W = (uint32_t)cs->world, rk = (uint32_t)cs->rank, rw = CsRed<T>::words(W);
        uint64_t* red_max = cs->red;
        uint64_t* red_min = cs->red + rw;
        const int op_min = CsRed<T>::gather ? kNcclMax : kNcclMin;
        const int dt = sizeof(T) == 8 ? kNcclFloat64 : kNcclFloat32;
        T* const xbuf = static_cast<T*>(cs->xbuf);
        T* const AS = static_cast<T*>(cs->AS);
        T* const xt = static_cast<T*>(cs->xt);
        T* const ds = static_cast<T*>(cs->ds);
        T* const inv = static_cast<T*>(cs->inv);
        T* const u1 = static_cast<T*>(cs->u1);
        T* const u2 = static_cast<T*>(cs->u2);
        T* const sgn = static_cast<T*>(cs->sgn);
        // c0 = A_loc^T y, lambda and the first pick over all shards, the first column to everyone
        uint32_t nb = 0;
        CSHIP(launch_sweep<T>(ctx, r, rhs_stride, 1, ws.c, nullptr, ws.pmax_val, ws.pmax_idx, &nb, nullptr));
        hipLaunchKernelGGL((k_cs_lmax<T>), dim3(1), dim3(kCsThreads), 0, st, (const T*)ws.pmax_val, (const uint32_t*)ws.pmax_idx, nb,
                           cs->col_lo, nl, red_max, rk, W, (const CsStateT<T>*)nullptr);
        cs_allreduce(ctx, cs, red_max, rw, kNcclUint64, kNcclMax);
        hipLaunchKernelGGL((k_cs_init<T>), dim3(1), dim3(kCsThreads), 0, st, At, ldm, cs->col_lo, nl, (const T*)ws.c, (const uint64_t*)red_max, W,
                           xbuf, cs->xcount, dst);
        cs_allreduce(ctx, cs, xbuf, cs->xcount, dt, kNcclSum);
        hipLaunchKernelGGL((k_cs_first<T>), dim3(1), dim3(kUpdThreads), 0, st, (const T*)xbuf, ldm, cs->col_lo, nl, K, tol, ctx->strict_sign,
                           AS, cs->gam, cs->tch, cs->trow, xt, ds, inv, ws.d, ws.insup, dst, trace);
        CSHIP(hipGetLastError());
        const uint32_t rp_blocks = (ldm + kCsThreads - 1) / kCsThreads;
        uint32_t scan_blocks = std::max<uint32_t>(1u, std::min<uint32_t>((nl + kCsThreads * 4 - 1) / (kCsThreads * 4), kMaxScanBlocks / 2));
        // Rounds are enqueued in blocks of `block` and the replicated `done` flag is read back after each block: every
        // rank sees the same flag at the same round, so all ranks issue the same collectives.
        const uint32_t block = (uint32_t)std::max(1, std::min(ctx->lookahead * 2, 16));
        CsStateT<T> hs{};
        uint64_t round = 1;
        const uint64_t last_round = (uint64_t)max_iter + 1;
        for (;;) {
            for (uint32_t b = 0; b < block && round <= last_round; ++b, ++round) {
                hipLaunchKernelGGL((k_cs_rp<T>), dim3(rp_blocks), dim3(kCsThreads), 0, st, (const T*)AS, ldm, m, (const T*)ws.y, K,
                                   (const uint32_t*)cs->tch, (const uint32_t*)cs->trow, (const T*)xt, (const uint32_t*)cs->gam,
                                   (const T*)ds, r, p, (const CsStateT<T>*)dst);
                CSHIP(launch_sweep<T>(ctx, r, rhs_stride, 2, ws.c, ws.q, ws.pmax_val, ws.pmax_idx, &nb, nullptr));
                hipLaunchKernelGGL((k_cs_lmax<T>), dim3(1), dim3(kCsThreads), 0, st, (const T*)ws.pmax_val, (const uint32_t*)ws.pmax_idx, nb,
                                   cs->col_lo, nl, red_max, rk, W, (const CsStateT<T>*)dst);
                cs_allreduce(ctx, cs, red_max, rw, kNcclUint64, kNcclMax);
                hipLaunchKernelGGL((k_cs_scan<T>), dim3(scan_blocks), dim3(kCsThreads), 0, st, (uint32_t)round, tol, max_iter, nl, cs->col_lo,
                                   (const T*)ws.c, (const T*)ws.q, (const T*)ws.x, (const T*)ws.d, (const uint8_t*)ws.insup,
                                   (const uint64_t*)red_max, red_min, rk, W, pmin, ctx->tie_guard, dst, ctx->dev_flags);
                cs_allreduce(ctx, cs, red_min, rw, kNcclUint64, op_min);
                hipLaunchKernelGGL((k_cs_select<T>), dim3(1), dim3(kUpdThreads), 0, st, (uint32_t)round, At, ldm, cs->col_lo, nl, K,
                                   (const uint64_t*)red_min, W, (const T*)ws.c, (const T*)ws.q, ws.x, ws.insup,
                                   cs->gam, cs->tch, cs->trow, xt, (const T*)ds, xbuf, cs->xcount, ctx->zero_on_removal, dst,
                                   ctx->dev_flags, trace, ws.trace_cap);
                cs_allreduce(ctx, cs, xbuf, cs->xcount, dt, kNcclSum);
                uint32_t gb = (uint32_t)std::min<uint64_t>(round + 1, K);
                hipLaunchKernelGGL((k_cs_update<T>), dim3(gb), dim3(kUpdThreads), 0, st, ldm, cs->col_lo, nl, K, tol, (const T*)xbuf,
                                   AS, (const uint32_t*)cs->gam, (const uint32_t*)cs->tch, (const uint32_t*)cs->trow, ds, inv,
                                   u1, u2, sgn, ws.d, dst);
                CSHIP(hipGetLastError());
            }
            CSHIP(hipMemcpyAsync(&hs, dst, sizeof(CsStateT<T>), hipMemcpyDeviceToHost, st));
            CSHIP(hipStreamSynchronize(st));
            if (hs.done || round > last_round) break;
        }
        if (!hs.done) { set_err(err, errlen, "colshard_solve: internal error, device loop did not terminate"); return SS_HIP_ERUNTIME; }
        if (hs.status != 0) { set_err(err, errlen, "colshard_solve: active set outgrew the workspace capacity"); return (int)hs.status; }
        if (x_local && nl != 0) {
            if (incx == 1) CSHIP(hipMemcpyAsync(x_local, ws.x, (size_t)nl * sizeof(T), hipMemcpyDefault, st));
            else CSHIP(hipMemcpy2DAsync(x_local, (size_t)incx * sizeof(T), ws.x, sizeof(T), sizeof(T), nl, hipMemcpyDefault, st));
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

}  // namespace

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
    return colshard_create_impl<float>(A_local, m, n_local, stride_row, stride_col, col_lo, n_total, device, comm_id, rank, world, host_collectives, err, errlen);
}

ss_hip_ctx* ss_hip_homotopy_colshard_create_f64(const double* A_local, size_t m, size_t n_local, ptrdiff_t stride_row,
                                                ptrdiff_t stride_col, size_t col_lo, size_t n_total, int device,
                                                const unsigned char* comm_id, int rank, int world,
                                                const ss_hip_collectives_f64* host_collectives, char* err, size_t errlen)
{
    return colshard_create_impl<double>(A_local, m, n_local, stride_row, stride_col, col_lo, n_total, device, comm_id, rank, world, host_collectives, err, errlen);
}

int ss_hip_homotopy_colshard_solve_f32(ss_hip_ctx* ctx, const float* y, ptrdiff_t incy, float tol, uint32_t max_iter,
                                       float* x_local, ptrdiff_t incx, uint32_t* iter_out, double* err_out,
                                       char* err, size_t errlen)
{
    return colshard_solve_impl<float>(ctx, y, incy, tol, max_iter, x_local, incx, iter_out, err_out, err, errlen);
}

int ss_hip_homotopy_colshard_solve_f64(ss_hip_ctx* ctx, const double* y, ptrdiff_t incy, double tol, uint32_t max_iter,
                                       double* x_local, ptrdiff_t incx, uint32_t* iter_out, double* err_out,
                                       char* err, size_t errlen)
{
    return colshard_solve_impl<double>(ctx, y, incy, tol, max_iter, x_local, incx, iter_out, err_out, err, errlen);
}

}  // extern "C"
