// homotopy.hip — the C-ABI of libss_hip.so (include/ss_hip.h): context creation (upload +
// re-layout of the sensing matrix), the host side of the device-resident Homotopy loop,
// the standalone sweep entry and the measurement hooks.
//
// Reference being replaced (paths under /root/reference):
//   ss::solver<T,P>::solver / homotopy_policy state   include/ss/ss.h:98-105, policies.h:42
//   solve_homotopy::op<mode,T> / run_solver<T>         src/solvers/homotopy.h:27-38,
//                                                      src/solvers/homotopy-cpu.cpp:186-275
//
// Execution model: one HIP stream per context.  A solve enqueues, per homotopy
// iteration ("round"), one fused sweep kernel and five small kernels; termination is
// decided ON THE DEVICE (k_select raises DevState::done, after which every kernel is a
// no-op), the host only polls a copy of that flag `lookahead` rounds behind the queue
// head, so the GPU never waits for the host between iterations.
#include "ss_hip_internal.h"
#include "resident.h"

#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <new>
#include <thread>

using namespace sship;

namespace sship {

void set_err(char* err, size_t errlen, const std::string& msg)
{
    if (!err || errlen == 0) return;
    const size_t k = std::min(errlen - 1, msg.size());
    std::memcpy(err, msg.data(), k);
    err[k] = '\0';
}

}  // namespace sship

namespace {

struct HipFail {
    hipError_t code;
    const char* what;
};

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) throw HipFail{ e_, #expr };                                     \
    } while (0)

std::string hip_msg(const HipFail& f)
{
    return std::string("HIP error: ") + hipGetErrorString(f.code) + " in " + f.what;
}

template <typename T>
Workspace<T>* ws_of(ss_hip_ctx* ctx) { return static_cast<Workspace<T>*>(ctx->ws); }

bool is_device_pointer(const void* p)
{
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof(attr));
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // unregistered host memory: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged ||
           attr.type == hipMemoryTypeUnified;
}

// ---- re-layout: At[j][i] = src[i*rs + j*cs] for rows [r0, r0+R) --------------------
template <typename T>
__global__ __launch_bounds__(1024)
void k_relayout(const T* __restrict__ src, long long rs, long long cs, uint32_t R, uint32_t n,
                uint32_t r0, T* __restrict__ At, uint32_t ldm)
{
    __shared__ T tile[32][33];
    const uint32_t j0 = blockIdx.x * 32;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (uint32_t i0 = blockIdx.y * 32; i0 < R; i0 += gridDim.y * 32) {   // uniform per block
        {
            const uint32_t i = i0 + ty, j = j0 + tx;
            tile[ty][tx] = (i < R && j < n) ? src[(long long)i * rs + (long long)j * cs] : T(0);
        }
        __syncthreads();
        {
            const uint32_t j = j0 + ty, i = i0 + tx;
            if (j < n && i < R) At[(size_t)j * ldm + r0 + i] = tile[tx][ty];
        }
        __syncthreads();
    }
}

template <typename T>
void upload_matrix(ss_hip_ctx* ctx, const T* A, ptrdiff_t rs, ptrdiff_t cs)
{
    const size_t m = ctx->m, n = ctx->n, s = sizeof(T);
    T* At = static_cast<T*>(ctx->At);
    const uint32_t ldm = ctx->ldm;
    const bool on_device = is_device_pointer(A);

    if (on_device) {
        // one generic strided transpose straight from the caller's device buffer
        const dim3 grid((unsigned)((n + 31) / 32), (unsigned)std::min<size_t>((m + 31) / 32, 32768));
        hipLaunchKernelGGL((k_relayout<T>), grid, dim3(1024), 0, ctx->stream, A, (long long)rs,
                           (long long)cs, (uint32_t)m, (uint32_t)n, 0u, At, ldm);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return;
    }

    if ((rs == 1 || m == 1) && (cs >= (ptrdiff_t)m || n == 1) && cs > 0) {
        // column-major host view: columns are already contiguous
        // (a single column has no meaningful column stride: any pitch >= the width will do)
        const size_t spitch = (n == 1) ? m * s : (size_t)cs * s;
        HIPCHK(hipMemcpy2D(At, (size_t)ldm * s, A, spitch, m * s, n, hipMemcpyHostToDevice));
        return;
    }

    // row panels through a staging buffer, transposed on the device
    const size_t panel_bytes = (size_t)256 << 20;
    size_t R = std::max<size_t>(1, std::min(m, panel_bytes / std::max<size_t>(1, n * s)));
    T* stage = nullptr;
    HIPCHK(hipMalloc(&stage, R * n * s));
    std::vector<T> gather;
    const bool rowmajor = (cs == 1 || n == 1) && (rs >= (ptrdiff_t)n || m == 1) && rs > 0;
    try {
        for (size_t r0 = 0; r0 < m; r0 += R) {
            const size_t rows = std::min(R, m - r0);
            if (rowmajor) {
                const size_t spitch = (m == 1) ? n * s : (size_t)rs * s;
                HIPCHK(hipMemcpy2D(stage, n * s, A + (ptrdiff_t)r0 * rs, spitch, n * s, rows,
                                   hipMemcpyHostToDevice));
            } else {
                // arbitrary (e.g. negative or doubly strided) host view: gather on the host
                gather.resize(rows * n);
                for (size_t i = 0; i < rows; ++i)
                    for (size_t j = 0; j < n; ++j)
                        gather[i * n + j] = A[(ptrdiff_t)(r0 + i) * rs + (ptrdiff_t)j * cs];
                HIPCHK(hipMemcpy(stage, gather.data(), rows * n * s, hipMemcpyHostToDevice));
            }
            const dim3 grid((unsigned)((n + 31) / 32), (unsigned)std::min<size_t>((rows + 31) / 32, 32768));
            hipLaunchKernelGGL((k_relayout<T>), grid, dim3(1024), 0, ctx->stream, stage, (long long)n,
                               1LL, (uint32_t)rows, (uint32_t)n, (uint32_t)r0, At, ldm);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
    } catch (...) {
        (void)hipFree(stage);
        throw;
    }
    HIPCHK(hipFree(stage));
}

// ---- compact output: one fixed-size record per signal, packed from the solver's own lists ----------
// Record layout (include/ss_hip.h): u32 K, u32 iter, f64 err, u32 idx[kmax], T val[kmax].
// The non-zero coefficients of x live on the columns of the slot's `touched` list (every column that was
// ever in the support: reference mode, where a leaving column may keep a rounding residue) or on its
// support list (zero_on_removal = 1: leaving columns carry exact zeros) — both sorted by column, so a
// stable compaction of the entries with x != 0 gives the record without scanning the n coefficients.
inline size_t record_bytes(uint32_t kmax, size_t elem) { return (16 + (size_t)kmax * (4 + elem) + 7) & ~(size_t)7; }

template <typename T>
__global__ __launch_bounds__(256)
void k_pack_records(const T* __restrict__ x, const uint32_t* __restrict__ gam2, const uint32_t* __restrict__ touched2,
                    const DevState* __restrict__ st, SlotDims L, int use_touched, uint32_t kmax,
                    unsigned char* __restrict__ rec, size_t rec_bytes)
{
    const size_t s = blockIdx.x;
    x += s * L.n_pad;
    st += s;
    const uint32_t cur = st->cur & 1u;
    const uint32_t* list = (use_touched ? touched2 : gam2) + s * 2 * (size_t)L.kcap + (size_t)cur * L.kcap;
    // (a solve that ended on an emptied support keeps the one column in its list: its x is handed back)
    uint32_t cnt = use_touched ? st->ntouched : st->K;
    if (cnt == 0u) cnt = 1u;
    if (cnt > L.kcap) cnt = L.kcap;
    // A slot that did not report (a form declined it: the one-workgroup kernels then leave K set and the lists unwritten) packs an
    // EMPTY record — whoever solves it again overwrites the record; its lists must not be walked (stale words as column indices)
    if (st->status != 0u) cnt = 0u;
    unsigned char* r = rec + s * rec_bytes;
    uint32_t* idx = reinterpret_cast<uint32_t*>(r + 16);
    T* val = reinterpret_cast<T*>(r + 16 + (size_t)kmax * 4);
    __shared__ uint32_t s_w[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t base = 0;
    for (uint32_t j0 = 0; j0 < cnt; j0 += 256u) {
        const uint32_t j = j0 + tid;
        uint32_t col = 0;
        T v = T(0);
        if (j < cnt) { col = list[j]; v = col < L.n_pad ? x[col] : T(0); }
        const bool nz = j < cnt && v != T(0);
        const uint64_t bal = __ballot(nz);
        __syncthreads();
        if (lane == 0) s_w[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < 4u; ++w) { if (w < wave) before += s_w[w]; total += s_w[w]; }
        const uint32_t pos = base + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (nz && pos < kmax) { idx[pos] = col; val[pos] = v; }
        base += total;
    }
    for (uint32_t p = (base < kmax ? base : kmax) + tid; p < kmax; p += 256u) { idx[p] = 0u; val[p] = T(0); }
    for (size_t p = 16 + (size_t)kmax * (4 + sizeof(T)) + tid; p < rec_bytes; p += 256u) r[p] = 0;     // alignment padding
    if (tid == 0) {
        reinterpret_cast<uint32_t*>(r)[0] = base;
        reinterpret_cast<uint32_t*>(r)[1] = st->iter;
        *reinterpret_cast<double*>(r + 8) = st->c_inf;
    }
}

// ---- end of a solve in ONE launch: the device state to pinned (device-mapped) host memory and, when the caller's x
// ---- lives on the device, the n coefficients to it — instead of two or three copy commands with 10-20 us between them
template <typename T>
__global__ __launch_bounds__(256)
void k_epilogue(const uint32_t* __restrict__ st_words, uint32_t* __restrict__ hs_mapped, uint32_t st_nwords,
                const T* __restrict__ x_src, T* __restrict__ x_dst, long long incx, uint32_t n, uint32_t status_word, uint32_t flag_word)
{
    // status_word != 0xffffffff (the screened form of one signal): the verdict of its certificate — "a column was not certified"
    // (the word at flag_word, raised by the screening pass) makes a clean status kStatusSubsetFail — is applied to the copy the host
    // reads, instead of a launch of its own (k_sub_finish) before this one
    // (the caller's x receives the coefficients only of a path that was certified: an uncertified one is solved again by the engine
    // behind the form, and if THAT fails the caller must not be left holding uncertified numbers beside an error code)
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    const bool withheld = status_word != 0xffffffffu && (st_words[status_word] != 0u || st_words[flag_word] != 0u);
    if (x_dst != nullptr && !withheld)
        for (uint32_t i = gtid; i < n; i += gsz) x_dst[(long long)i * incx] = x_src[i];
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < st_nwords; i += blockDim.x) {
            uint32_t w = st_words[i];
            if (i == status_word && w == 0u && st_words[flag_word] != 0u) w = kStatusSubsetFail;
            __hip_atomic_store(&hs_mapped[i], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
}

template <typename T>
void release_arrays(Workspace<T>* w)
{
    void* ptrs[] = { w->y, w->rhs, w->cq, w->x, w->d, w->insup, w->pmax_val, w->pmax_idx,
                     w->pmin_val, w->pmin_idx, w->gam, w->touched, w->inv[0], w->u1,
                     w->u2, w->sgn, w->st, w->ndone, w->tile_skip, w->gcache, w->slot_of, w->c0,
                     w->tcand, w->sw_list, w->la_dbg, w->la_sync, w->cq_alt, w->slot_identity,
                     w->slot_col, w->solo_log, w->sub_pos, w->v_max, w->v_min, w->cand_top, w->solo_stage,
                     w->sw_list2, w->sub_cols, w->subg };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    TraceEntry* tr = w->trace;
    const uint32_t tc = w->trace_cap;
    *w = Workspace<T>();
    w->trace = tr;
    w->trace_cap = tc;
}

template <typename T>
void free_ws(Workspace<T>* w)
{
    if (!w) return;
    release_arrays(w);
    if (w->trace) (void)hipFree(w->trace);
    delete w;
}

// Makes sure the workspace holds `nslots` signals and an active set of `kcap` columns each.
// Everything is (re)allocated together; sizes only grow.
template <typename T>
void ensure_workspace(ss_hip_ctx* ctx, uint32_t nslots, uint32_t kcap)
{
    if (!ctx->ws) ctx->ws = new Workspace<T>();
    Workspace<T>* w = ws_of<T>(ctx);
    if (nslots <= w->b_cap && kcap <= w->kcap) return;
    uint32_t want_k = std::max<uint32_t>(kcap, w->kcap);
    if (kcap > w->kcap)   // grow geometrically so slightly larger max_iter does not realloc
        want_k = std::max<uint32_t>(kcap, std::min<uint32_t>(kKcapLimit, std::max<uint32_t>(64, w->kcap * 2)));
    const uint32_t want_b = std::max<uint32_t>(nslots, w->b_cap);
    // one slot keeps the classic [2][ldm] / [2][n_pad] layout; batches are padded to whole
    // 128-row GEMM tiles
    const uint32_t b_pad = want_b == 1 ? 1u : (want_b + 127u) / 128u * 128u;
    release_arrays(w);
    const size_t ldm = ctx->ldm, np = ctx->n_pad, s = sizeof(T), B = want_b, K = want_k;
    SlotDims L{};
    L.n_pad = ctx->n_pad;
    L.ldm = ctx->ldm;
    L.m = (uint32_t)ctx->m;
    L.kcap = want_k;
    L.b_pad = b_pad;
    L.pmax_stride = kMaxSweepBlocks;
    L.pmin_stride = kMaxScanBlocks;
    HIPCHK(hipMalloc(&w->y, B * ldm * s));
    HIPCHK(hipMalloc(&w->rhs, 2 * (size_t)b_pad * ldm * s));
    HIPCHK(hipMalloc(&w->cq, 2 * (size_t)b_pad * np * s));
    HIPCHK(hipMalloc(&w->x, B * np * s));
    HIPCHK(hipMalloc(&w->d, B * np * s));
    HIPCHK(hipMalloc(&w->insup, B * np));
    HIPCHK(hipMalloc(&w->pmax_val, B * L.pmax_stride * s));
    HIPCHK(hipMalloc(&w->pmax_idx, B * L.pmax_stride * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&w->pmin_val, B * L.pmin_stride * s));
    HIPCHK(hipMalloc(&w->pmin_idx, B * L.pmin_stride * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&w->gam, B * 2 * K * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&w->touched, B * 2 * K * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&w->inv[0], B * 2 * K * K * s));
    HIPCHK(hipMalloc(&w->u1, B * K * s));
    HIPCHK(hipMalloc(&w->u2, B * K * s));
    HIPCHK(hipMalloc(&w->sgn, B * K * s));
    HIPCHK(hipMalloc(&w->st, B * sizeof(DevState)));
    HIPCHK(hipMalloc(&w->ndone, 64));
    HIPCHK(hipMalloc(&w->tile_skip, ((size_t)b_pad / 128 + 1) * sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(w->tile_skip, 0, ((size_t)b_pad / 128 + 1) * sizeof(uint32_t), ctx->stream));
    w->inv[1] = w->inv[0] + K * K;
    w->c = w->cq;
    w->q = w->cq + (size_t)b_pad * np;
    w->b_cap = want_b;
    w->kcap = want_k;
    w->dims = L;
    HIPCHK(hipMemsetAsync(w->y, 0, B * ldm * s, ctx->stream));
    HIPCHK(hipMemsetAsync(w->rhs, 0, 2 * (size_t)b_pad * ldm * s, ctx->stream));
    HIPCHK(hipMemsetAsync(w->cq, 0, 2 * (size_t)b_pad * np * s, ctx->stream));
    HIPCHK(hipMemsetAsync(w->st, 0, B * sizeof(DevState), ctx->stream));
    HIPCHK(hipMemsetAsync(w->ndone, 0, 64, ctx->stream));
    // the support / touched lists start as "no column" (0xffffffff), not as whatever the allocation held: a reader that walks a list
    // nobody wrote (a slot a form declined) then meets an index every consumer bounds-checks — the same on every run, not a stale
    // word that happens to be a valid column in one process and an unmapped address in the next
    HIPCHK(hipMemsetAsync(w->gam, 0xff, B * 2 * K * sizeof(uint32_t), ctx->stream));
    HIPCHK(hipMemsetAsync(w->touched, 0xff, B * 2 * K * sizeof(uint32_t), ctx->stream));
}

}  // namespace

namespace sship {
int colshard_workspace(ss_hip_ctx* ctx, uint32_t kcap)
{
    try {
        if (ctx->is_f64) ensure_workspace<double>(ctx, 1, kcap);
        else ensure_workspace<float>(ctx, 1, kcap);
    } catch (const HipFail&) {
        (void)hipGetLastError();
        return SS_HIP_ENOMEM;
    } catch (const std::bad_alloc&) {
        return SS_HIP_ENOMEM;
    }
    return SS_HIP_OK;
}
}  // namespace sship

namespace {

template <typename T>
ss_hip_ctx* create_impl(const T* A, size_t m, size_t n, ptrdiff_t rs, ptrdiff_t cs, int device,
                        char* err, size_t errlen, int kind = 0)
{
    if (kind == 1 && m < n) {
        // irls-cpu.cpp / qr_decomposition.h:101 assert M >= N ("underdetermined systems not supported")
        set_err(err, errlen, "ss_hip_irls_create: IRLS needs a matrix with at least as many rows as columns");
        return nullptr;
    }
    if (!A || m == 0 || n == 0) {
        set_err(err, errlen, "ss_hip_homotopy_create: A must be a non-empty m x n matrix");
        return nullptr;
    }
    if (m > (size_t)1 << 28 || n > (size_t)1 << 30) {
        set_err(err, errlen, "ss_hip_homotopy_create: matrix dimensions too large");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        set_err(err, errlen, "ss_hip_homotopy_create: no HIP device available");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        set_err(err, errlen, "ss_hip_homotopy_create: device index out of range");
        return nullptr;
    }
    ss_hip_ctx* ctx = new (std::nothrow) ss_hip_ctx();
    if (!ctx) {
        set_err(err, errlen, "ss_hip_homotopy_create: out of host memory");
        return nullptr;
    }
    try {
        HIPCHK(hipSetDevice(device));
        ctx->device = device;
        ctx->is_f64 = sizeof(T) == 8;
        ctx->m = m;
        ctx->n = n;
        ctx->ldm = (uint32_t)((m + kRowPad - 1) / kRowPad * kRowPad);
        ctx->n_pad = (uint32_t)((n + kColPad - 1) / kColPad * kColPad);
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        const size_t bytes = (size_t)ctx->n_pad * ctx->ldm * sizeof(T);
        HIPCHK(hipMalloc(&ctx->At, bytes));
        HIPCHK(hipMemsetAsync(ctx->At, 0, bytes, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        upload_matrix<T>(ctx, A, rs, cs);
        ctx->kind = kind;
        // (SS_HIP_SCREEN_SINGLE = 0 / 1 / 2: the initial value of option "screen_single" — the test suite pins the engines it
        // examines one by one with it; ss_hip_set_option overrides)
        if (const char* ev = std::getenv("SS_HIP_SCREEN_SINGLE")) ctx->screen_single = std::max(0, std::min(2, std::atoi(ev)));
        if (kind == 1) HIPCHK(irls_factor<T>(ctx));
        else ensure_workspace<T>(ctx, 1, 64);
        HIPCHK(hipHostMalloc(&ctx->host_flags, 64 * sizeof(uint32_t), hipHostMallocMapped));
        std::memset(ctx->host_flags, 0, 64 * sizeof(uint32_t));
        HIPCHK(hipHostMalloc(&ctx->hs_pinned, sizeof(DevState), hipHostMallocMapped));
        std::memset(ctx->hs_pinned, 0, sizeof(DevState));
        if (hipHostGetDevicePointer(&ctx->hs_mapped, ctx->hs_pinned, 0) != hipSuccess) { (void)hipGetLastError(); ctx->hs_mapped = nullptr; }
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->dev_flags), ctx->host_flags, 0));
        HIPCHK(hipEventCreate(&ctx->ev_solve0));
        HIPCHK(hipEventCreate(&ctx->ev_solve1));
        const uint64_t s = sizeof(T);
        ctx->stats.sweep_bytes = (uint64_t)m * n * s + 2 * (uint64_t)m * s + 2 * (uint64_t)n * s;
        ctx->stats.sweep1_bytes = (uint64_t)m * n * s + (uint64_t)m * s + (uint64_t)n * s;
        ctx->stats.sweep32_bytes = (uint64_t)m * n * s + 32 * (uint64_t)m * s + 32 * (uint64_t)n * s;
        ctx->stats.sweep32_timed_cols = n;
        ctx->stats.sweep64_bytes = (uint64_t)m * n * s + 64 * (uint64_t)m * s + 64 * (uint64_t)n * s;
        ctx->stats.sweep64_flops = 2ull * 64ull * (uint64_t)m * n;
    } catch (const HipFail& f) {
        set_err(err, errlen, hip_msg(f));
        ss_hip_homotopy_destroy(ctx);
        return nullptr;
    } catch (const std::bad_alloc&) {
        set_err(err, errlen, "ss_hip_homotopy_create: out of host memory");
        ss_hip_homotopy_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

// ---- lookahead engine (fp32): Gram-column cache management and round launches -------------
bool ensure_full_gram(ss_hip_ctx* ctx);     // G = A^T A of the context (defined with the batched paths)
void gram_reserve_start(ss_hip_ctx* ctx, size_t B);

// the 32-RHS lookahead sweep of either precision, and the resident iteration kernel (fp32 only)
inline hipError_t launch_gemm32(const ss_hip_ctx* ctx, const uint32_t* rcols, const uint32_t* drows, float* D, uint32_t ldd, const DevState* st)
{ return launch_gemm32_tn_f32(ctx, rcols, drows, D, ldd, st); }
inline hipError_t launch_gemm32(const ss_hip_ctx* ctx, const uint32_t* rcols, const uint32_t* drows, double* D, uint32_t ldd, const DevState* st)
{ return launch_gemm32_tn_f64(ctx, rcols, drows, D, ldd, st); }
inline hipError_t launch_persist(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, uint32_t max_iter, uint32_t lds_cols, bool after_solo = false)
{ return launch_la_persist_f32(ctx, ws, tol, max_iter, lds_cols, after_solo); }
inline hipError_t launch_persist(ss_hip_ctx*, Workspace<double>&, double, uint32_t, uint32_t, bool = false)
{ return hipErrorInvalidConfiguration; }
// speculative form (fp32 only): solo launch + verification + publication; seeding of the subset ranking
inline hipError_t launch_solo_group(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, uint32_t max_iter)
{
    const hipError_t e = launch_la_solo_f32(ctx, ws, tol, max_iter);
    return e != hipSuccess ? e : launch_la_verify_f32(ctx, ws);
}
inline hipError_t launch_solo_group(ss_hip_ctx*, Workspace<double>&, double, uint32_t) { return hipErrorInvalidConfiguration; }
inline hipError_t launch_cand_init(ss_hip_ctx* ctx, Workspace<float>& ws) { return launch_la_cand_init_f32(ctx, ws); }
inline hipError_t launch_cand_init(ss_hip_ctx*, Workspace<double>&) { return hipErrorInvalidConfiguration; }
inline hipError_t launch_top_cand(ss_hip_ctx* ctx, Workspace<float>& ws, uint32_t nsel = 32) { return launch_la_top_cand_f32(ctx, ws, nsel); }
inline hipError_t launch_top_cand(ss_hip_ctx*, Workspace<double>&, uint32_t = 32) { return hipErrorInvalidConfiguration; }
// the first lookahead sweep of a fp32 solve may fetch 64 Gram columns in one (MFMA-bound) pass
inline hipError_t launch_gemm_first(const ss_hip_ctx* ctx, uint32_t nsel, const uint32_t* rcols, const uint32_t* drows, float* D, uint32_t ldd, const DevState* st)
{ return nsel > 32 ? launch_gemm64_tn_f32(ctx, rcols, drows, D, ldd, st) : launch_gemm32_tn_f32(ctx, rcols, drows, D, ldd, st); }
inline hipError_t launch_gemm_first(const ss_hip_ctx* ctx, uint32_t nsel, const uint32_t* rcols, const uint32_t* drows, double* D, uint32_t ldd, const DevState* st)
{ return nsel > 32 ? launch_gemm64_tn_f64(ctx, rcols, drows, D, ldd, st) : launch_gemm32_tn_f64(ctx, rcols, drows, D, ldd, st); }
// columns of a miss sweep: 32 in fp32 (HBM-bound), option sweep_cols_f64 in double precision
// (fetch = how many passes this solve has fetched after its first: in double precision the first two passes are wide;
// later misses come late on the path — at configs[4] the third pass serves the last ~8 of 128 iterations — and take 32)
inline uint32_t miss_cols(const ss_hip_ctx*, const Workspace<float>&, uint32_t = 0) { return 32u; }
inline uint32_t miss_cols(const ss_hip_ctx* ctx, const Workspace<double>& ws, uint32_t fetch = 0)
{
    if (!(ctx->sweep_cols_f64 > 32 && ws.gcap >= 192)) return 32u;
    return (fetch >= 1 && ctx->sweep_cols_f64_late <= 32) ? 32u : 64u;
}

// early form of the speculative engine (fp32): the first solo launch runs on the subset Gram matrix beside the passes over A
inline void early_prologue(ss_hip_ctx* ctx, Workspace<float>& ws, uint32_t nparts, float tol, uint32_t max_iter, uint32_t lds_cols,
                           hipEvent_t pe0, hipEvent_t pe1, hipEvent_t pe2 = nullptr, hipEvent_t pe3 = nullptr);
inline void early_prologue(ss_hip_ctx*, Workspace<double>&, uint32_t, double, uint32_t, uint32_t, hipEvent_t, hipEvent_t, hipEvent_t = nullptr, hipEvent_t = nullptr)
{ throw HipFail{ hipErrorInvalidConfiguration, "early_prologue<double>" }; }

template <typename T> struct Lookahead {
    static constexpr bool supported = true;

    static void ensure(ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t kcap)
    {
        const uint32_t gpitch = (ctx->n_pad + 1023u) / 1024u * 1024u;
        // every sweep caches up to 32 columns; a solve needs at most one sweep per inserted column
        uint64_t want = 32ull * ((uint64_t)kcap + 2);
        // (the resident kernel addresses cache rows with 32-bit byte offsets: stay below 4 GiB)
        const uint64_t budget = std::min<uint64_t>((uint64_t)ctx->cache_mib << 20, (4095ull << 20));
        const uint64_t fit = std::max<uint64_t>(64, budget / ((uint64_t)gpitch * sizeof(T)));
        if (want > fit) want = fit;
        if (ws.gcache && ws.gcap >= want && ws.gpitch == gpitch) return;
        void* olds[] = { ws.gcache, ws.slot_of, ws.c0, ws.tcand, ws.sw_list, ws.slot_col, ws.solo_log, ws.sub_pos, ws.v_max, ws.v_min, ws.cand_top, ws.solo_stage,
                         ws.sw_list2, ws.sub_cols, ws.subg };
        for (void* p : olds) if (p) HIPCHK(hipFree(p));
        ws.sw_list2 = nullptr; ws.sub_cols = nullptr; ws.subg = nullptr;
        ws.gcache = nullptr; ws.slot_of = nullptr; ws.c0 = nullptr; ws.tcand = nullptr; ws.sw_list = nullptr;
        ws.solo_stage = nullptr; ws.slot_col = nullptr; ws.solo_log = nullptr; ws.sub_pos = nullptr; ws.v_max = nullptr; ws.v_min = nullptr; ws.cand_top = nullptr;
        ws.gcap = 0;
        HIPCHK(hipMalloc(&ws.gcache, (size_t)want * gpitch * sizeof(T)));
        HIPCHK(hipMalloc(&ws.slot_of, (size_t)ctx->n_pad * sizeof(int32_t)));
        HIPCHK(hipMalloc(&ws.c0, (size_t)ctx->n_pad * sizeof(T)));
        HIPCHK(hipMalloc(&ws.tcand, (size_t)ctx->n_pad * sizeof(T)));
        HIPCHK(hipMalloc(&ws.sw_list, 128 * sizeof(uint32_t)));
        // (0xffffffff = "no column": k_subset_pick with a small subset writes only the entries it uses)
        HIPCHK(hipMemsetAsync(ws.sw_list, 0xff, 128 * sizeof(uint32_t), ctx->stream));
        if (sizeof(T) == 4) {
            // speculative form: slot -> column map, breakpoint log, verification partials, subset ranking
            ws.nvwg = (uint32_t)((ctx->n + kSoloWidth - 1) / kSoloWidth);
            HIPCHK(hipMalloc(&ws.slot_col, (size_t)want * sizeof(uint32_t)));
            HIPCHK(hipMalloc(&ws.solo_log, ((size_t)kSoloHeaderWords + (size_t)kSoloLogCap * kSoloEntryWords) * sizeof(uint32_t)));
            HIPCHK(hipMalloc(&ws.solo_stage, (size_t)kSoloStageWords * sizeof(uint32_t)));
            HIPCHK(hipMalloc(&ws.sub_pos, (size_t)ctx->n_pad));
            HIPCHK(hipMemsetAsync(ws.sub_pos, 0, (size_t)ctx->n_pad, ctx->stream));
            HIPCHK(hipMalloc(&ws.v_max, (size_t)kSoloLogCap * ws.nvwg * sizeof(uint32_t)));
            HIPCHK(hipMalloc(&ws.v_min, (size_t)kSoloLogCap * ws.nvwg * sizeof(uint64_t)));
            HIPCHK(hipMalloc(&ws.sw_list2, 128 * sizeof(uint32_t)));
            HIPCHK(hipMemsetAsync(ws.sw_list2, 0xff, 128 * sizeof(uint32_t), ctx->stream));
            HIPCHK(hipMalloc(&ws.sub_cols, 2 * (size_t)kSoloWidth * sizeof(uint32_t)));     // columns, then the progress hints (float)
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&ws.subg), (size_t)kSoloWidth * kSoloWidth * sizeof(float)));
            HIPCHK(hipMalloc(&ws.cand_top, kCandPerBlock * (size_t)ws.nvwg * sizeof(uint64_t)));
            HIPCHK(hipMemsetAsync(ws.cand_top, 0xff, kCandPerBlock * (size_t)ws.nvwg * sizeof(uint64_t), ctx->stream));
        }
        ws.gcap = (uint32_t)want;
        ws.gpitch = gpitch;
        // developer aid: SS_HIP_LA_DEBUG=<file> dumps the stage timestamps of k_la_iter after each solve
        if (!ws.la_dbg && std::getenv("SS_HIP_LA_DEBUG")) HIPCHK(hipMalloc(&ws.la_dbg, 2048 * 8 * sizeof(uint64_t)));
        if (!ws.la_sync) HIPCHK(hipMalloc(&ws.la_sync, kLaSyncBytes));
        if (!ws.cq_alt) HIPCHK(hipMalloc(&ws.cq_alt, 2 * (size_t)ctx->n_pad * sizeof(T)));
    }

    // c0 = A^T y has been swept into ws.c0 (partials in pmax): first pick, first lookahead sweep
    // returns the number of columns of the first sweep (0: none — full-G mode); ev0 / ev1: events around it
    static uint32_t init(ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nparts, T tol, bool solo = false,
                         hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr)
    {
        uint32_t first_cols = 0;
        hipStream_t st = ctx->stream;
        const bool full = ws.gram_is_full;                      // every column is "cached": no sweep, ever
        // (the slot map, the hand-off area of the resident kernel and the dense vectors were cleared by
        // k_la_reset before the sweep; the first sign is read from c0 itself)
        if (ws.la_dbg) HIPCHK(hipMemsetAsync(ws.la_dbg, 0, 2048 * 8 * sizeof(uint64_t), st));
        HIPCHK(launch_la_init_pick<T>(ctx, ws, nparts, tol, full));
        if (solo) HIPCHK(launch_cand_init(ctx, ws));           // per-block tops of |c0|: ranking of the first sweep and subset
        if (!full) {
            // the first sweep: the entering column and the largest |c0| — 64 columns in one pass (fp32; MFMA-bound,
            // ~0.45 ms at C2) instead of two 32-column passes and a round trip through the host in between
            const uint32_t nsel = sizeof(T) == 8 ? miss_cols(ctx, ws)
                                : ((ctx->first_sweep_cols > 32 && ctx->sweep32_variant == 0 && ws.gcap >= 128) ? 64u : 32u);
            if (solo && ctx->n > 32u * 512u) HIPCHK(launch_top_cand(ctx, ws, nsel));
            else HIPCHK(launch_la_top<T>(ctx, ws, 1, nsel));
            if (ev0) HIPCHK(hipEventRecord(ev0, ctx->stream));
            HIPCHK(launch_gemm_first(ctx, nsel, ws.sw_list, ws.sw_list + 64, ws.gcache, ws.gpitch, ws.st));
            if (ev1) HIPCHK(hipEventRecord(ev1, ctx->stream));
            first_cols = nsel;
        }
        HIPCHK(launch_la_update<T>(ctx, ws, 0, tol));
        if (ctx->la_fused) return first_cols;      // k_la_iter forms c and q itself
        uint32_t np2 = 0;
        HIPCHK(launch_la_cq<T>(ctx, ws, &np2));
        ws.la_nparts = np2;
        return first_cols;
    }

    // fused form: one launch per iteration ...
    // lds_cols != 0: resident form holding up to that many support columns in LDS
    // solo: the speculative form (one workgroup + verification) while the device has not switched it off
    static void iterate(ss_hip_ctx* ctx, Workspace<T>& ws, T tol, uint32_t max_iter, uint32_t lds_cols, bool solo = false)
    {
        if (solo) HIPCHK(launch_solo_group(ctx, ws, tol, max_iter));
        else if (lds_cols != 0) HIPCHK(launch_persist(ctx, ws, tol, max_iter, lds_cols));
        else HIPCHK(launch_la_iter<T>(ctx, ws, tol, max_iter));
    }
    // ... and, when the device reports an entering column without cached Gram column, the sweep
    // that fetches it (plus 31 likely successors) and the inverse update that was waiting for it
    // from_cand: the speculative form is running — its verification has left the per-block candidate tops of
    // the scan that missed (wide dictionaries only: with few blocks the tops are too few to rank from)
    static void fetch(ss_hip_ctx* ctx, Workspace<T>& ws, T tol, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, bool from_cand = false,
                      uint32_t fetch_index = 0)
    {
        const uint32_t nsel = miss_cols(ctx, ws, fetch_index);
        if (from_cand && ctx->n > 32u * 512u) HIPCHK(launch_top_cand(ctx, ws));
        else HIPCHK(launch_la_top<T>(ctx, ws, 0, nsel));
        if (ev0) HIPCHK(hipEventRecord(ev0, ctx->stream));
        HIPCHK(launch_gemm_first(ctx, nsel, ws.sw_list, ws.sw_list + 64, ws.gcache, ws.gpitch, ws.st));
        if (ev1) HIPCHK(hipEventRecord(ev1, ctx->stream));
        HIPCHK(launch_la_update<T>(ctx, ws, 1, tol));
    }
    // OMP: the next picks are the largest correlations; the pending update is k_gramupd in OMP mode
    static void fetch_omp(ss_hip_ctx* ctx, Workspace<T>& ws, T tol, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr)
    {
        const uint32_t nsel = miss_cols(ctx, ws);
        HIPCHK(launch_la_top<T>(ctx, ws, 2, nsel));
        if (ev0) HIPCHK(hipEventRecord(ev0, ctx->stream));
        HIPCHK(launch_gemm_first(ctx, nsel, ws.sw_list, ws.sw_list + 64, ws.gcache, ws.gpitch, ws.st));
        if (ev1) HIPCHK(hipEventRecord(ev1, ctx->stream));
        HIPCHK(launch_la_omp_update<T>(ctx, ws, tol));
    }

    // one homotopy iteration: scan + select, (sweep if the entering column is not cached),
    // inverse update + direction from the cache, Gram-form c and q
    static void round(ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t rnd, T tol, uint32_t max_iter,
                      hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr)
    {
        HIPCHK(launch_la_scansel<T>(ctx, ws, rnd, ws.la_nparts, tol, max_iter));
        const uint32_t nsel = miss_cols(ctx, ws);
        HIPCHK(launch_la_top<T>(ctx, ws, 0, nsel));
        if (ev0) HIPCHK(hipEventRecord(ev0, ctx->stream));
        HIPCHK(launch_gemm_first(ctx, nsel, ws.sw_list, ws.sw_list + 64, ws.gcache, ws.gpitch, ws.st));
        if (ev1) HIPCHK(hipEventRecord(ev1, ctx->stream));
        HIPCHK(launch_la_update<T>(ctx, ws, rnd, tol));
        uint32_t np2 = 0;
        HIPCHK(launch_la_cq<T>(ctx, ws, &np2));
        ws.la_nparts = np2;
    }
};

// ---- early form (fp32, option early_solo): everything of a typical solve in ONE enqueue ---------------------------------
// After c0 = A^T y the columns that matter are ranked once (|c0| tops).  The best 256 become the subset of the
// first solo launch, whose Gram values all come from Gs = A_S^T A_S (subgram.hip: bit for bit the sweep's values,
// 8 MiB of A instead of two passes over it); the best 64 get cache slots and their full Gram columns are swept on a
// SECOND stream while the solo workgroup iterates (the barrier-free pass: 2048 single-wave workgroups that share
// the chip with it; a gate kernel holds them back until the solo workgroup is resident).  Afterwards, on the main
// stream again: slots + up to two passes for the columns the launch used beyond those 64, then the verification of
// every breakpoint over all n columns (from the swept rows, exactly as in the plain form), the commit, and the
// resident form for the last step of the path.  A solve that does not fit this mould (a pick outside the subset,
// a miss, a failed check) simply continues in the plain pump below.
inline void early_prologue(ss_hip_ctx* ctx, Workspace<float>& ws, uint32_t nparts, float tol, uint32_t max_iter, uint32_t lds_cols,
                           hipEvent_t pe0, hipEvent_t pe1, hipEvent_t pe2, hipEvent_t pe3)
{
    hipStream_t st = ctx->stream;
    const int probe = ctx->early_probe;              // developer aid: 1 = no overlap (the passes first, then the solo launch)
    if (!ctx->stream2) {
        // A stream of another priority class: HIP keeps a pool of hardware queues per class, so this one never
        // shares a queue with the main stream (two streams on ONE hardware queue execute in submission order, and
        // the gate kernel below would then sit in front of the very launch it waits for).
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        if (hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, hi) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        }
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        // a third stream (lowest priority class: again its own hardware queue) for the tiles dealt out by shader engine
        if (hipStreamCreateWithPriority(&ctx->stream3, hipStreamNonBlocking, lo) != hipSuccess) {
            (void)hipGetLastError();
            ctx->stream3 = nullptr;
        }
        if (ctx->stream3) {
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_gate, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_b0, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_join3, hipEventDisableTiming));
            HIPCHK(hipMalloc(&ctx->se_count, 2 * (kSeCount + 2) * sizeof(uint32_t)));
            if (hipStreamCreateWithPriority(&ctx->stream4, hipStreamNonBlocking, lo) != hipSuccess) { (void)hipGetLastError(); ctx->stream4 = nullptr; }
            if (ctx->stream4) HIPCHK(hipEventCreateWithFlags(&ctx->ev_join4, hipEventDisableTiming));
        }
    }
    // Dealing the passes out around the solo workgroup (option early_se).  The hardware distributes a grid statically:
    // every shader engine of every XCD gets the same number of workgroups, whatever its CUs hold (per-workgroup trace of
    // the real pass, tools/probe_pass_trace.py).  The solo workgroup owns one CU of one SE; 512 tiles put 16 on the 7
    // other CUs of that SE, two of them run a third tile, and a pass — bound by what ONE CU can load — ends at 0.49 ms
    // instead of 0.37.  So: the main launch takes 14 tiles per SE (two per CU on the solo SE, room for exactly two
    // workgroups per CU: LDS padding), a second launch on a third stream puts two more workgroups on every SE, which
    // pick their tile by where they run and leave at once on the solo workgroup's SE (k_gemm32_tn_f32<BYSE>), and the
    // two tiles that are left of 512 are formed by the VALU chain (k_cols_gram) on a fourth stream.  Every CU but one
    // then carries exactly two tiles.
    uint32_t main_tiles = 0, se_last = 0, tail_c0 = 0, tail_cols = 0, se_quota = 2u, range_last = 0;
    if (ctx->early_se && ctx->early_pass == 2 && !probe && ctx->stream3 && ctx->num_cus == 8 * (int)kSeCount &&
        ctx->n_pad % 128 == 0 && ctx->ldm % 256 == 0) {
        const uint32_t nt = (uint32_t)(ctx->n_pad / 128);
        if (nt > 14u * kSeCount && nt <= 16u * kSeCount) {
            main_tiles = 14u * kSeCount;                                        // 448
            se_last = std::min<uint32_t>(nt, main_tiles + 2u * (kSeCount - 1u)); // tiles [448, 510) by shader engine
            tail_c0 = se_last * 128u;
            tail_cols = (nt - se_last) * 128u;                                  // the last 256 columns at 8192 x 65536
        } else if (nt > 16u * kSeCount && ctx->early_se == 3) {
            // Wider dictionaries, opt-in (early_se = 3; measured no faster than one launch per pass: 2.43 against 2.21 ms per
            // solve at 98304 columns, 2.79 against 2.78 at 131072 — with three or more tiles per CU the launch evens itself
            // out as workgroups finish; the cliff round 2 saw beyond 65536 columns was the candidate ranking, solo.hip:
            // rank_offers): a pass is bound by what ONE CU can load, so its time is the largest number of
            // tiles any CU gets — u = nt / 255 per CU when the 255 CUs beside the solo workgroup share evenly: the main
            // launch takes 7 u tiles per shader engine (u per CU on the solo workgroup's SE, which has 7), the workgroups
            // dealt out by shader engine come back until their SE has had u more (u per CU there too), and the remainder
            // (< 255 tiles) runs as a partial round behind the main launch — or, when it is at most 16 tiles, on the VALU
            // chain beside the pass like the two left-over tiles at 65536 columns.
            const uint32_t per = 8u * kSeCount - 1u;                            // 255 CUs take tiles
            const uint32_t u = nt / per;
            main_tiles = 7u * kSeCount * u;
            se_quota = u;
            se_last = main_tiles + (kSeCount - 1u) * u;                          // = 255 u
            const uint32_t rest = nt - se_last;
            if (rest <= 16u) { tail_c0 = se_last * 128u; tail_cols = rest * 128u; }
            else range_last = nt;
        }
    }
    ctx->stats.sweep32_timed_cols = main_tiles ? (uint64_t)main_tiles * 128u : (uint64_t)ctx->n;   // (of THIS form's timed launch: kind 5 below)
    if (ws.la_dbg) HIPCHK(hipMemsetAsync(ws.la_dbg, 0, 2048 * 8 * sizeof(uint64_t), st));
    HIPCHK(launch_la_init_pick<float>(ctx, ws, nparts, tol, false));
    HIPCHK(launch_la_cand_init_f32(ctx, ws));
    HIPCHK(launch_subset_pick_f32(ctx, ws));                        // subset of 256, slots 0..63, the two sweep lists
    // Gs, and a_idx . a_idx seeded into the first pick's cache row for the first inverse update
    HIPCHK(launch_subset_gram_f32(ctx, ws.sub_cols, ws.subg, ws.st, ws.gcache, ws.slot_of, ws.gpitch));
    // (the counters of the dealing-out were zeroed by k_subset_pick)
    HIPCHK(hipEventRecord(ctx->ev_fork, st));
    // second stream: the two 32-column passes, held back until the solo workgroup is resident.  (Enqueued AFTER the
    // solo launch: should the two streams ever share a hardware queue after all, the gate then follows the launch it
    // waits for and everything merely runs one after the other.)
    auto enqueue_passes = [&]() {
        HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        if (!probe) HIPCHK(launch_wait_started(ctx, ws, ctx->stream2));
        if (main_tiles) {
            // third stream: the tiles dealt out by shader engine.  Its workgroups must hold their CUs BEFORE the main launch
            // of the pass arrives (a workgroup that finds its SE full holds up every workgroup behind it in its grid; the
            // two that land on the solo workgroup's SE leave at once when they get there first): the main launch waits for
            // their arrival count
            HIPCHK(hipEventRecord(ctx->ev_gate, ctx->stream2));                     // (behind the gate: the solo workgroup is resident)
            HIPCHK(hipStreamWaitEvent(ctx->stream3, ctx->ev_gate, 0));
            HIPCHK(launch_gemm32se_on(ctx, ctx->stream3, ws.sw_list, ws.sw_list + 64, ws.gcache, ws.gpitch, main_tiles, se_last, ws.st,
                                      ctx->se_count, se_quota));
            HIPCHK(launch_wait_count(ctx->stream2, ctx->se_count + kSeCount, early_se_wgs(ctx), ws.st));
        }
        if (pe0) HIPCHK(hipEventRecord(pe0, ctx->stream2));
        HIPCHK(launch_gemm32w_on(ctx, ctx->stream2, ws.sw_list, ws.sw_list + 64, ws.gcache, ws.gpitch, main_tiles));
        if (pe1) HIPCHK(hipEventRecord(pe1, ctx->stream2));
        if (range_last) HIPCHK(launch_gemm32range_on(ctx, ctx->stream2, ws.sw_list, ws.sw_list + 64, ws.gcache, ws.gpitch, se_last, range_last));
        // the second pass's 32 columns are chosen now, half a millisecond into the solo launch: what has entered
        // its support without a Gram row so far, then the columns closest to entering (k_pick_pass_b)
        HIPCHK(launch_pick_pass_b_f32(ctx, ws, ctx->stream2));
        if (main_tiles) {
            HIPCHK(hipEventRecord(ctx->ev_b0, ctx->stream2));                       // (the second list exists, the first pass is complete)
            HIPCHK(hipStreamWaitEvent(ctx->stream3, ctx->ev_b0, 0));
            HIPCHK(launch_gemm32se_on(ctx, ctx->stream3, ws.sw_list + 32, ws.sw_list + 96, ws.gcache, ws.gpitch, main_tiles, se_last, ws.st,
                                      ctx->se_count + (kSeCount + 2), se_quota));
            HIPCHK(hipEventRecord(ctx->ev_join3, ctx->stream3));
            HIPCHK(launch_wait_count(ctx->stream2, ctx->se_count + (kSeCount + 2) + kSeCount, early_se_wgs(ctx), ws.st));
        }
        if (pe2) HIPCHK(hipEventRecord(pe2, ctx->stream2));
        HIPCHK(launch_gemm32w_on(ctx, ctx->stream2, ws.sw_list + 32, ws.sw_list + 96, ws.gcache, ws.gpitch, main_tiles));
        if (pe3) HIPCHK(hipEventRecord(pe3, ctx->stream2));
        if (range_last) HIPCHK(launch_gemm32range_on(ctx, ctx->stream2, ws.sw_list + 32, ws.sw_list + 96, ws.gcache, ws.gpitch, se_last, range_last));
        if (main_tiles) HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_join3, 0));
        HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
    };
    if (probe == 1) { enqueue_passes(); HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0)); }
    // main stream: first inverse + direction (k_gramupd, round 0: reads only the seeded entry), then the solo launch
    HIPCHK(launch_la_update<float>(ctx, ws, 0, tol));
    HIPCHK(launch_la_solo_f32(ctx, ws, tol, max_iter));
    if (probe != 1) enqueue_passes();
    if (tail_cols) {
        // the last columns of both passes by the VALU chain (32 small workgroups each, ~70 us): on a fourth stream beside the
        // passes when there is one, else on this stream behind the solo launch
        hipStream_t ts = ctx->stream4 ? ctx->stream4 : st;
        if (ctx->stream4) HIPCHK(hipStreamWaitEvent(ts, ctx->ev_gate, 0));
        else HIPCHK(hipStreamWaitEvent(ts, ctx->ev_b0, 0));
        HIPCHK(launch_cols_gram_on(ctx, ts, tail_c0, tail_cols, ws.sw_list, ws.sw_list + 64, ws.gcache, ws.gpitch));
        if (ctx->stream4) HIPCHK(hipStreamWaitEvent(ts, ctx->ev_b0, 0));
        HIPCHK(launch_cols_gram_on(ctx, ts, tail_c0, tail_cols, ws.sw_list + 32, ws.sw_list + 96, ws.gcache, ws.gpitch));
        if (ctx->stream4) {
            HIPCHK(hipEventRecord(ctx->ev_join4, ts));
            HIPCHK(hipStreamWaitEvent(st, ctx->ev_join4, 0));
        }
    }
    // ... which the passes have to be complete for from here on
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
    HIPCHK(launch_missing_cols_f32(ctx, ws));
    HIPCHK(launch_gemm32_tn_f32(ctx, ws.sw_list2, ws.sw_list2 + 64, ws.gcache, ws.gpitch, nullptr));
    HIPCHK(launch_gemm32_tn_f32(ctx, ws.sw_list2 + 32, ws.sw_list2 + 96, ws.gcache, ws.gpitch, nullptr));
    HIPCHK(launch_la_verify_f32(ctx, ws));                          // verification + publication
    if (lds_cols != 0) HIPCHK(launch_la_persist_f32(ctx, ws, tol, max_iter, lds_cols, true));
}

// copies a strided vector (host or device) into a contiguous device buffer
template <typename T>
void copy_in(ss_hip_ctx* ctx, T* dst_dev, const T* src, ptrdiff_t inc, size_t len)
{
    if (inc == 1) {
        HIPCHK(hipMemcpyAsync(dst_dev, src, len * sizeof(T), hipMemcpyDefault, ctx->stream));
    } else {
        HIPCHK(hipMemcpy2DAsync(dst_dev, sizeof(T), src, (size_t)inc * sizeof(T), sizeof(T), len,
                                hipMemcpyDefault, ctx->stream));
    }
}

template <typename T>
void copy_out(ss_hip_ctx* ctx, T* dst, ptrdiff_t inc, const T* src_dev, size_t len)
{
    if (inc == 1) {
        HIPCHK(hipMemcpyAsync(dst, src_dev, len * sizeof(T), hipMemcpyDefault, ctx->stream));
    } else {
        HIPCHK(hipMemcpy2DAsync(dst, (size_t)inc * sizeof(T), src_dev, sizeof(T), sizeof(T), len,
                                hipMemcpyDefault, ctx->stream));
    }
}

hipEvent_t prof_event(ss_hip_ctx* ctx, size_t i)
{
    while (ctx->prof_events.size() <= i) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        ctx->prof_events.push_back(e);
    }
    return ctx->prof_events[i];
}

// compact output: pack the records of `nslots` finished slots into the context's staging buffer (device)
template <typename T>
unsigned char* pack_records(ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t kmax)
{
    const size_t rb = record_bytes(kmax, sizeof(T));
    const size_t need = rb * nslots;
    if (ctx->rec_stage_bytes < need) {
        if (ctx->rec_stage) HIPCHK(hipFree(ctx->rec_stage));
        ctx->rec_stage = nullptr;
        ctx->rec_stage_bytes = 0;
        HIPCHK(hipMalloc(&ctx->rec_stage, need));
        ctx->rec_stage_bytes = need;
    }
    hipLaunchKernelGGL((k_pack_records<T>), dim3(nslots), dim3(256), 0, ctx->stream, (const T*)ws.x, (const uint32_t*)ws.gam,
                       (const uint32_t*)ws.touched, (const DevState*)ws.st, ws.dims, ctx->zero_on_removal ? 0 : 1, kmax,
                       ctx->rec_stage, rb);
    HIPCHK(hipGetLastError());
    return ctx->rec_stage;
}

// one signal in the subset form (subbatch.hip) when G = A^T A is at hand: c0 is in ws.c0
inline hipError_t sub_single(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, uint32_t max_iter)
{
    const size_t need = sub_buffer_bytes(1);
    if (ctx->sub_buf_bytes < need) {
        if (ctx->sub_buf) HIPCHK(hipFree(ctx->sub_buf));
        ctx->sub_buf = nullptr;
        ctx->sub_buf_bytes = 0;
        HIPCHK(hipMalloc(&ctx->sub_buf, need));
        ctx->sub_buf_bytes = need;
    }
    return launch_sub_form(ctx, ws, 1, ws.c0, tol, max_iter);
}
inline hipError_t sub_single(ss_hip_ctx*, Workspace<double>&, double, uint32_t) { return hipErrorInvalidConfiguration; }

// one signal in the screened form (screen.hip): no G — the subset's own Gram matrix from A, then one pass over the fp16 copy of A
inline hipError_t scr_single(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, uint32_t max_iter, bool first16, hipEvent_t e0, hipEvent_t e1,
                             hipEvent_t e2, hipEvent_t e3, hipEvent_t e4, hipEvent_t e5, bool omp, bool rescue)
{
    const size_t need = sub_buffer_bytes(1);
    if (ctx->sub_buf_bytes < need) {
        if (ctx->sub_buf) HIPCHK(hipFree(ctx->sub_buf));
        ctx->sub_buf = nullptr;
        ctx->sub_buf_bytes = 0;
        HIPCHK(hipMalloc(&ctx->sub_buf, need));
        ctx->sub_buf_bytes = need;
    }
    if (ctx->sub_dbg == nullptr && std::getenv("SS_HIP_SUB_STAMPS")) {
        HIPCHK(hipMalloc(&ctx->sub_dbg, 16 * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(ctx->sub_dbg, 0, 16 * sizeof(unsigned long long), ctx->stream));
    }
    // (with the state mirrored to mapped host memory the epilogue launch applies the certificate's verdict: no k_sub_finish)
    return launch_screen_form(ctx, ws, tol, max_iter, first16, ctx->hs_mapped == nullptr, e0, e1, e2, e3, e4, e5, omp, rescue);
}
inline hipError_t scr_rescue_scan(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, bool from_recheck, uint32_t* found) { return launch_screen_rescue_scan(ctx, ws, tol, from_recheck, found); }
inline hipError_t scr_rescue_scan(ss_hip_ctx*, Workspace<double>&, double, bool, uint32_t*) { return hipErrorInvalidConfiguration; }
inline hipError_t scr_single(ss_hip_ctx*, Workspace<double>&, double, uint32_t, bool, hipEvent_t, hipEvent_t, hipEvent_t, hipEvent_t, hipEvent_t, hipEvent_t, bool, bool) { return hipErrorInvalidConfiguration; }
// (typed shims of the fp64 screened form: never reached for float)
inline hipError_t scr64_gather(ss_hip_ctx* ctx, const double* c0, const double* y, hipEvent_t e0, hipEvent_t e1) { return screen64_gather(ctx, c0, y, e0, e1); }
inline hipError_t scr64_gather(ss_hip_ctx*, const float*, const float*, hipEvent_t, hipEvent_t) { return hipErrorInvalidConfiguration; }
inline hipError_t scr64_certify(ss_hip_ctx* ctx, Workspace<double>& ws, const double* y, uint32_t T, double tol, double c_inf, uint32_t K, hipEvent_t e2, hipEvent_t e3, bool omp,
                                bool first16)
{
    return screen64_certify(ctx, ws, y, T, tol, c_inf, K, e2, e3, omp, first16);
}
inline hipError_t scr64_certify(ss_hip_ctx*, Workspace<float>&, const float*, uint32_t, float, double, uint32_t, hipEvent_t, hipEvent_t, bool, bool) { return hipErrorInvalidConfiguration; }

inline hipError_t scr64_resident(ss_hip_ctx* ctx, Workspace<double>& ws, double tol, uint32_t max_iter, bool first16, bool omp, hipEvent_t e0, hipEvent_t e1,
                                 hipEvent_t e2, hipEvent_t e3, hipEvent_t e4, hipEvent_t e5, bool rescue)
{
    return launch_screen64_resident(ctx, ws, tol, max_iter, first16, omp, e0, e1, e2, e3, e4, e5, rescue);
}
inline hipError_t scr64_resident(ss_hip_ctx*, Workspace<float>&, float, uint32_t, bool, bool, hipEvent_t, hipEvent_t, hipEvent_t, hipEvent_t, hipEvent_t, hipEvent_t, bool) { return hipErrorInvalidConfiguration; }
inline hipError_t scr64_rescue_scan(ss_hip_ctx* ctx, Workspace<double>& ws, double tol, bool from_recheck, uint32_t* found) { return launch_screen64_rescue_scan(ctx, ws, tol, from_recheck, found); }
inline hipError_t scr64_rescue_scan(ss_hip_ctx*, Workspace<float>&, float, bool, uint32_t*) { return hipErrorInvalidConfiguration; }

// why a subset / screened solve was not reported: DevState::sub_reason's bits into the statistics
inline void count_reasons(ss_hip_ctx* ctx, uint32_t r, bool tie)
{
    ss_hip_stats& S = ctx->stats;
    if (r & kReasonRemoval) S.why_removal += 1;
    if (r & kReasonPositions) S.why_positions += 1;
    if (r & kReasonLog) S.why_breakpoints += 1;
    if (r & kReasonGuard) S.why_guard += 1;
    if (r & kReasonNoCand) S.why_no_candidate += 1;
    if (r & kReasonFirstState) S.why_first_state += 1;
    if (r & kReasonIrregular) S.why_irregular += 1;
    if (r & kReasonOverflow) S.why_overflow += 1;
    if (r & kReasonColumn) S.why_column += 1;
    if (tie || (r & kReasonTie)) S.why_tie += 1;
}

// ---- the host side of a solve in the launch-chain engines ---------------------------------------------------------------------
// The device decides termination (k_scansel / k_la_iter raise DevState::done and mirror it, with the round reached, into pinned host
// memory).  The host keeps at most `lookahead` launches queued beyond the device's position and stops enqueueing as soon as it
// sees `done`; launches already queued behind it are no-ops.
struct PumpState {
    bool solo = false;            // speculative launches still in use (the device may hand over to the resident form)
    bool solo_started = false;
    bool early = false;           // early form: a solo group (and the resident launch behind it) is queued already
    uint32_t early_lds_cols = 0;
    size_t nprof = 0;             // profiling events used so far
    bool enqueued = false;        // the pump queued work behind a speculative epilogue
};

// Fused lookahead engine: every launch of k_la_iter performs the next iteration, or nothing while the device waits for a Gram
// column (host_flags[2] counts those waits).  Returns false if the loop made no progress (internal error).
template <typename T>
bool pump_fused(ss_hip_ctx* ctx, Workspace<T>& ws, T tol, uint32_t max_iter, bool la, bool la_omp, bool prof, PumpState& ps)
{
    hipStream_t st = ctx->stream;
    const uint32_t L = (uint32_t)std::max(1, std::min(ctx->lookahead, 64));
    volatile uint32_t* hf = ctx->host_flags;
    bool& solo = ps.solo;
    const bool solo_started = ps.solo_started, early = ps.early;
    const uint32_t early_lds_cols = ps.early_lds_cols;
    size_t& nprof = ps.nprof;
    bool& pump_enqueued = ps.enqueued;
    // Fused lookahead engine: every launch of k_la_iter performs the next iteration, or
    // nothing while the device waits for a Gram column (hf[2] counts those waits).  The
    // host keeps L launches queued ahead and answers each wait with one fetch.
    uint64_t enq = early ? (early_lds_cols != 0 ? 2u : 1u) : 0u;   // (early form: a solo group and the resident launch behind it are queued)
    uint32_t handled = 0, timed_fetches = 0;
    // resident kernel: LDS tier (support columns it can hold); 0 = one launch per iteration
    uint32_t lds_cols = 0;
    const uint32_t kcap_ws = ws.dims.kcap;       // what the device checks K against (>= this solve's kcap)
    if (la && ctx->la_fused >= 2 && sizeof(T) == 4) {
        lds_cols = std::min<uint32_t>((kcap_ws + 15u) & ~15u, kLaLdsSmall);
        if (!la_persist_usable(ctx, lds_cols)) lds_cols = 0;
    }
    const uint64_t max_launch = 4 * ((uint64_t)max_iter + 2) + 64;
    bool stuck = false;
    for (;;) {
        uint32_t spins = 0;
        // (a solo group runs until the host has to act — fetch, hand-over, end — so one group in
        // flight is enough: every further one is three launches that find nothing to do; after
        // the hand-over, near the end of the path, two resident launches)
        const uint32_t depth = solo ? 1u : (solo_started ? std::min<uint32_t>(L, 2u) : L);
        while (hf[1] == 0 && hf[2] == handled && enq >= (uint64_t)hf[0] + depth) {
            if ((++spins & 0x3ffu) == 0) {
                const hipError_t qs = hipStreamQuery(st);
                if (qs == hipSuccess) break;
                if (qs != hipErrorNotReady) throw HipFail{ qs, "hipStreamQuery(solve loop)" };
            }
            std::this_thread::yield();
        }
        if (hf[1] != 0) break;
        pump_enqueued = true;
        if (hf[2] != handled) {
            const bool timed_la = prof && (timed_fetches++ % (uint32_t)std::max(1, ctx->profile_every) == 0);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (timed_la) { e0 = prof_event(ctx, 2 * nprof); e1 = prof_event(ctx, 2 * nprof + 1); }
            if (la_omp) Lookahead<T>::fetch_omp(ctx, ws, tol, e0, e1);
            else Lookahead<T>::fetch(ctx, ws, tol, e0, e1, solo, handled);
            if (timed_la) { ctx->prof_kind.push_back(3); ++nprof; }
            ++handled;
        }
        if (lds_cols != 0 && hf[3] > lds_cols) {
            // the support outgrew the tier: take the large one, or go on one launch per iteration
            const uint32_t big = std::min<uint32_t>((kcap_ws + 15u) & ~15u, kLaLdsLarge);
            lds_cols = (hf[3] <= big && big > lds_cols && la_persist_usable(ctx, big)) ? big : 0u;
        }
        if (enq >= max_launch) { stuck = true; break; }
        if (solo && hf[4] != 0) solo = false;   // the device handed over to the resident / launch-per-iteration form
        if (la_omp) HIPCHK(launch_la_omp<T>(ctx, ws, tol, max_iter));
        else Lookahead<T>::iterate(ctx, ws, tol, max_iter, lds_cols, solo);
        ++enq;
        if (solo && lds_cols != 0) {
            // the resident form queued right behind the speculative group: a no-op unless that group hands
            // over (the last step of a path), which then costs no trip through the host
            HIPCHK(launch_persist(ctx, ws, tol, max_iter, lds_cols, true));
            ++enq;
        }
    }
    if (stuck) {
        HIPCHK(hipStreamSynchronize(st));
        if (hf[1] == 0) return false;
    }
    return true;
}

// One launch chain per round (engine 0: the fused sweep + tail; the lookahead engine without in-kernel grid synchronisation; the
// reference-order engine; OMP in residual form).
template <typename T>
void pump_rounds(ss_hip_ctx* ctx, Workspace<T>& ws, T tol, uint32_t max_iter, bool la, bool ro, bool omp, uint32_t ro_parts,
                 size_t rhs_stride, bool prof, PumpState& ps)
{
    hipStream_t st = ctx->stream;
    const uint32_t L = (uint32_t)std::max(1, std::min(ctx->lookahead, 64));
    volatile uint32_t* hf = ctx->host_flags;
    const uint64_t last_round = (uint64_t)max_iter + 1;
    size_t& nprof = ps.nprof;
    for (uint64_t round = 1; round <= last_round; ++round) {
        if (round > L) {
            const uint32_t need = (uint32_t)(round - L);
            uint32_t spins = 0;
            while (hf[1] == 0 && hf[0] < need) {
                if ((++spins & 0x3ffu) == 0) {
                    const hipError_t q = hipStreamQuery(st);
                    if (q == hipSuccess) break;               // queue drained
                    if (q != hipErrorNotReady) throw HipFail{ q, "hipStreamQuery(solve loop)" };
                }
                std::this_thread::yield();
            }
            if (hf[1] != 0) break;
        }
        if (la) {
            // the launch is a no-op unless a column without cached Gram column enters: time
            // every `profile_every`-th launch and keep the ones that did work (see below)
            const bool timed_la = prof && (round % (uint64_t)std::max(1, ctx->profile_every) == 0);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (timed_la) { e0 = prof_event(ctx, 2 * nprof); e1 = prof_event(ctx, 2 * nprof + 1); }
            Lookahead<T>::round(ctx, ws, (uint32_t)round, tol, max_iter, e0, e1);
            if (timed_la) { ctx->prof_kind.push_back(3); ++nprof; }
            continue;
        }
        if (ro) {
            // homotopy-cpu.cpp:236-272 with ONE pass over A per iteration: r = y - A x and p = A d (the direction built
            // from the signs of c - gamma q), the fused sweep [c, q] = A^T [r, p], lambda + the while-test + the check of
            // those signs against the re-computed c (a mismatch rebuilds the direction; p and q are then formed again),
            // the scan + toggle + x update, the inverse update and the next direction
            HIPCHK(launch_ro_round<T>(ctx, ws, 1u, (uint32_t)round, ro_parts, tol, max_iter));
            continue;
        }
        if (omp) {
            // orthogonal matching pursuit round: c = A^T r (one right-hand side), pick,
            // bordered inverse + least squares on the support, new residual
            uint32_t nbo = 0;
            HIPCHK(launch_sweep<T>(ctx, ws.rhs, rhs_stride, 1, ws.c, nullptr, ws.pmax_val, ws.pmax_idx, &nbo, ws.st));
            HIPCHK(launch_omp_tail<T>(ctx, ws, 1, (uint32_t)round, nbo, tol, max_iter));
            continue;
        }
        uint32_t nb = 0;
        // HIP events cost tens of microseconds of stream time each: time every
        // `profile_every`-th fused sweep only (option), still inside the solve
        const bool timed = prof && (round % (uint64_t)std::max(1, ctx->profile_every) == 0);
        if (timed) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof), st)); }
        HIPCHK(launch_sweep<T>(ctx, ws.rhs, rhs_stride, 2, ws.c, ws.q, ws.pmax_val, ws.pmax_idx, &nb, ws.st));
        if (timed) {
            HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof + 1), st));
            ctx->prof_kind.push_back((int)round + 16);     // >= 16: fused sweep of round (kind-16)
            ++nprof;
        }
        HIPCHK(launch_iteration_tail<T>(ctx, ws, 1, (uint32_t)round, nb, tol, max_iter));
    }
}

// The HIP events of a profiled solve into the context's statistics (prof_kind says what each pair bracketed)
template <typename T>
void account_profile(ss_hip_ctx* ctx, size_t nprof, uint32_t scr_launches, const DevState& hs)
{
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev_solve0, ctx->ev_solve1));
    ctx->stats.solve_ms += ms;
    // only sweeps that did real work: the initial one and rounds 1..done_round
    for (size_t i = 0; i < nprof; ++i) {
        HIPCHK(hipEventElapsedTime(&ms, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]));
        if (ctx->prof_kind[i] == 1) {
            ctx->stats.sweep1_launches += 1;
            ctx->stats.sweep1_ms += ms;
        } else if (ctx->prof_kind[i] == 6) {
            // the screening pass: fp16 copy of A + the residual block (re-read from L2 by every workgroup: not counted) + norms
            ctx->stats.screen_launches += scr_launches;
            ctx->stats.screen_ms += ms;
            ctx->stats.screen_bytes += (uint64_t)scr_launches * ((uint64_t)ctx->ldm * ctx->n_pad * 2ull + 96ull * ctx->ldm * 2ull + (uint64_t)ctx->n_pad * 4ull);
        } else if (ctx->prof_kind[i] == 8) {
            ctx->stats.res_solve_launches += 1;
            ctx->stats.res_solve_ms += ms;
        } else if (ctx->prof_kind[i] == 7) {
            ctx->stats.first16_launches += 1;
            ctx->stats.first16_ms += ms;
            ctx->stats.first16_bytes += (uint64_t)ctx->ldm * ctx->n_pad * (uint64_t)ctx->first_pass_elem_bytes + (uint64_t)ctx->ldm * sizeof(T) + (uint64_t)ctx->n_pad * 4ull;
        } else if (ctx->prof_kind[i] == 4) {
            if (ms > 0.02f) {                          // (a launch of a solve that ended at the first pick is a no-op)
                ctx->stats.sweep64_launches += 1;
                ctx->stats.sweep64_ms += ms;
            }
        } else if (ctx->prof_kind[i] == 3 || ctx->prof_kind[i] == 5) {
            // lookahead sweep: a launch that found nothing to do returns in microseconds.  Bytes per EVENT: a
            // plain pass (3) covers all n columns, the early form's main launch (5) its share of them
            const uint64_t sz = sizeof(T);
            const uint64_t cols = ctx->prof_kind[i] == 5 ? ctx->stats.sweep32_timed_cols : (uint64_t)ctx->n;
            const uint64_t bytes = (uint64_t)ctx->m * cols * sz + 32ull * ctx->m * sz + 32ull * cols * sz;
            if ((double)bytes / (ms * 1e-3) < 50e12) {   // < 50 TB/s: it streamed A
                ctx->stats.sweep32_launches += 1;
                ctx->stats.sweep32_ms += ms;
                ctx->stats.sweep32_bytes_timed += bytes;
            }
        } else if ((uint32_t)(ctx->prof_kind[i] - 16) <= hs.done_round) {
            ctx->stats.sweep_launches += 1;
            ctx->stats.sweep_ms += ms;
        }
    }
}

// developer aid (SS_HIP_SOLO_DEBUG = a path): the log and the verification partials of the last speculative launch
template <typename T>
void dump_solo_debug(Workspace<T>& ws, const DevState& hs, const char* path)
{
    // developer aid: the log and the verification partials of the last solo launch
    const size_t lw = (size_t)kSoloHeaderWords + (size_t)kSoloLogCap * kSoloEntryWords;
    std::vector<uint32_t> lg(lw), vm((size_t)kSoloLogCap * ws.nvwg);
    std::vector<uint64_t> vn((size_t)kSoloLogCap * ws.nvwg);
    HIPCHK(hipMemcpy(lg.data(), ws.solo_log, lw * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(vm.data(), ws.v_max, vm.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(vn.data(), ws.v_min, vn.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (FILE* fp = std::fopen(path, "wb")) {
        const uint32_t hdr[4] = { hs.solo_nlog, ws.nvwg, kSoloEntryWords, kSoloHeaderWords };
        std::fwrite(hdr, 4, 4, fp);
        std::fwrite(lg.data(), 4, lg.size(), fp);
        std::fwrite(vm.data(), 4, vm.size(), fp);
        std::fwrite(vn.data(), 8, vn.size(), fp);
        std::fclose(fp);
    }
}

// ---- which way a single-signal solve goes ---------------------------------------------------------------------------------
// A solve is ONE attempt in one form (solve_once); what the attempt reports decides whether another form takes the signal
// (the ladder in solve_once's verdict section).  A Route is everything a later attempt is told about the earlier ones.
struct Route {
    bool omp = false;              // orthogonal matching pursuit (ss_hip_omp_solve_*)
    bool force_residual = false;   // the Gram form's tolerance guard tripped: residual form (engine 0) for this signal
    bool no_solo = false;          // no speculative launches (callers that run many contexts side by side)
    bool force_ro = false;         // a step-length scan met an exact tie: the reference-order engine (reforder.hip) arbitrates
    bool no_sub = false;           // the subset / screened forms declined the signal: the default engine
    bool no_res = false;           // fp64: the resident tier declined it: the sub-dictionary tier
    bool plain = false;            // the early form asked for the plain speculative form (sticky for the rest of this solve)
    bool rescue = false;           // the screened form declined the signal for a column its ranking missed: once more, with the columns its log names
    uint32_t rescue_why = 0;       // ... the declined attempt's reason bits (kReason*)
    Route after() const { Route r; r.plain = plain; return r; }     // a fresh route that keeps what is sticky
};

// The forms an attempt can take (at most one of sub1 / scr1 / scr64 / scr64r; la / la_omp / ro say which engine stands behind)
struct Forms {
    bool y_direct = false;   // the lookahead engine reads the signal from the caller's device buffer (no copy command)
    bool ro = false;         // reference-order engine (option engine = 3, and the arbiter of tie stalls)
    bool la = false;         // lookahead (Gram-form) engine: the default
    bool la_omp = false;     // OMP in Gram form (k_la_omp)
    bool sub1 = false;       // G = A^T A at hand: the subset form of the batches for ONE signal (subbatch.hip)
    bool scr1 = false;       // fp32 screened form (screen.hip + resident.hip): the default on large dictionaries
    bool scr64 = false;      // fp64 screened form, sub-dictionary tier (2048 columns, a context of its own)
    bool scr64r = false;     // fp64 screened form, resident tier (256 columns, one workgroup)
};

// What a context's options, its size and the route so far allow.  (No side effects: the step-aside counters are applied by the
// caller, so that asking twice gives the same answer.)
template <typename T>
Forms choose_forms(ss_hip_ctx* ctx, const Route& route, const T* y, void* rec_out)
{
    Forms f;
    const bool omp = route.omp;
    f.y_direct = !omp && !route.force_ro && ctx->engine != 3 && Lookahead<T>::supported && ctx->engine >= 1 && !route.force_residual &&
                 is_device_pointer(y);
    f.ro = !omp && (route.force_ro || ctx->engine == 3);
    f.la = !omp && Lookahead<T>::supported && ctx->engine >= 1 && !route.force_residual && !f.ro;
    f.la_omp = omp && Lookahead<T>::supported && ctx->engine >= 1 && !route.force_residual && ctx->la_fused >= 1;
    // With G = A^T A at hand (a large batch has run on the context, or option gram_full_after) a single signal takes the
    // subset form of the batches (subbatch.hip): A^T y, one workgroup on 448 columns, the check over all columns — no
    // pass over A beyond A^T y.  What the form does not vouch for is solved again the usual way (no_sub).
    // (where the screened form below applies it is the faster of the two since its first pass reads the fp16 copy — 0.71 ms against
    // 1.0 at configs[1] — and takes the signal; option screen_single = 0 leaves it to this form)
    f.sub1 = f.la && sizeof(T) == 4 && !route.no_sub && ctx->batch_subset && ctx->gram_single && ctx->gram_full != nullptr && sub_form_usable(ctx);
    // Without G: the screened form (screen.hip) — the same subset solve on the subset's own Gram matrix (formed from A), every
    // state of its path then screened against all columns by ONE pass over a half-precision copy of A with a rigorous error
    // bound, instead of the default engine's two fp32 passes.  It stands in for the default speculative engine only
    // (la_fused = 3 with the early form: contexts whose options ask for another engine get that engine).
    // (OMP — ss::omp<float> — takes it too: the resident kernel's OMP statement on the same subset, the same certificate; not with a trace)
    f.scr1 = (f.la || (f.la_omp && ctx->screen_resident && !ctx->tracing)) && sizeof(T) == 4 && !route.no_sub && ctx->la_fused >= 3 && ctx->early_solo &&
             !ctx->early_probe && ctx->solo_subset == 256 && (!f.sub1 || screen_first16_usable(ctx)) && screen_form_usable(ctx);
    if (f.scr1) f.sub1 = false;
    // fp64: the same certificate around the fp64 engine — the path is solved by a context of its own on the 2048 columns with
    // the largest |c0| (passes and iterations on 1.6 % of the dictionary), its logged states are screened against all columns
    // (not with a trace or compact records asked for: the sub-context's lists are over ITS columns)
    // (OMP too: the sub-context runs k_la_omp, the certificate is the same — nothing outside the sub-dictionary reaches the pick's |c|)
    f.scr64 = (f.la || f.la_omp) && sizeof(T) == 8 && !route.no_sub && !ctx->tracing && rec_out == nullptr && ctx->la_fused >= 1 && screen64_usable(ctx);
    // ... and before that tier, the RESIDENT tier (resident.hip): the path on the 256 best-ranked columns in ONE workgroup with the
    // Gram values in registers — no sub-context, no host round trip, everything queued in one go; its lists are over the
    // dictionary's own columns, so a trace and compact records work too.  What it does not report goes to the tier above.
    f.scr64r = (f.la || f.la_omp) && sizeof(T) == 8 && !route.no_sub && !route.no_res && ctx->screen_resident && ctx->la_fused >= 1 &&
               !(omp && ctx->tracing) && screen64_usable(ctx) && screen64_resident_usable(ctx);
    return f;
}

// ---- what an attempt reported: the ladder ---------------------------------------------------------------------------------------
// Reads the device state of a finished attempt and decides: report it (*report), fail (the returned status), or hand the signal to
// another form (*again, *next).  Rungs, in the order they are tried:
//   exact tie in a step-length scan            -> reference-order engine (the arbiter; option tie_rerun)
//   fp64 resident tier declined                -> sub-dictionary tier                      (no_res)
//   fp32 screened form declined, a missed column -> the same form once more with it        (rescue)
//   screened form (fp32 / fp64 tier 2) declined -> default engine                           (no_sub)
//   subset form on G declined                  -> default engine                           (no_sub)
//   early form used too many unfetched columns -> plain speculative form                   (plain)
//   Gram-form tolerance guard                  -> residual form                            (force_residual)
//   a resident grid's wait expired             -> launch per iteration, then no in-kernel grid synchronisation (context options)
// Every rung also keeps the statistics (ss_hip_stats) and the step-aside counters of its form.
template <typename T>
int attempt_verdict(ss_hip_ctx* ctx, const Route& route, const Forms& f, const DevState& hs, Route* next, bool* again, bool* report,
                    char* err, size_t errlen)
{
    const bool omp = route.omp, force_residual = route.force_residual, no_solo = route.no_solo;
    const bool ro = f.ro, la = f.la, la_omp = f.la_omp, sub1 = f.sub1, scr1 = f.scr1, scr64 = f.scr64, scr64r = f.scr64r;
    auto retry = [&](const Route& r) { *next = r; *again = true; return SS_HIP_OK; };

    if (!hs.done) {
        set_err(err, errlen, "solve: internal error, device loop did not terminate");
        return SS_HIP_ERUNTIME;
    }
    // (a tie met by the screened form's subset solve is a tie of the SUBSET's view — on a subset that cannot be certified it may not
    // exist over all columns: such a signal goes to the default engine below, which meets the tie itself if it is real)
    const bool scr_tie = (scr1 || scr64r) && ctx->tie_rerun && !ctx->tie_guard && (hs.tie_stall != 0 || hs.status == kStatusTieRerun);
    if (!ro && !omp && !scr_tie && ctx->tie_rerun && !ctx->tie_guard && (hs.tie_stall != 0 || hs.status == kStatusTieRerun)) {
        // a step-length scan met an exact tie (DevState::tie_stall): whether the strict t > 0 of the reference then
        // derails the path is decided by rounding — the reference-order engine is the arbiter
        ctx->stats.tie_reruns += 1;
        Route r = route.after(); r.no_solo = no_solo; r.force_ro = true;
        return retry(r);
    }
    if (sub1 || scr1 || scr64) {
        // (a context whose signals the form hands back more often than not stops trying for a while)
        ctx->sub_seen += 1;
        if (hs.status == kStatusSubsetDecline || hs.status == kStatusSubsetFail || scr_tie) ctx->sub_failed += 1;
        if (ctx->sub_seen >= 8) {
            if (2 * ctx->sub_failed > ctx->sub_seen) ctx->sub_off_solves = 64;
            ctx->sub_seen = 0;
            ctx->sub_failed = 0;
        }
    }
    if (scr64r) {
        const bool back = hs.status == kStatusSubsetDecline || hs.status == kStatusSubsetFail || scr_tie;
        ctx->res_seen += 1;
        if (back) ctx->res_failed += 1;
        if (ctx->res_seen >= 8) {
            if (2 * ctx->res_failed > ctx->res_seen) ctx->res_off_solves = 64;
            ctx->res_seen = 0;
            ctx->res_failed = 0;
        }
        if (back) {
            const uint32_t rs = hs.sub_reason;
            if (!route.rescue) count_reasons(ctx, rs, scr_tie);
            if (std::getenv("SS_HIP_SUB_DEBUG"))
                std::fprintf(stderr, "[screened form, fp64 resident tier%s] status %u reason 0x%x after %u iterations, %u states logged, K = %u, lambda %g\n",
                             route.rescue ? ", rescue" : "", hs.status, hs.sub_reason, hs.iter, hs.solo_nlog, hs.K, hs.c_inf);
            // (the rescue, as in fp32: a column the ranking left out of the 256 — the tier once more with the columns its log names)
            const bool rescuable = !omp && !route.rescue && !scr_tie && ctx->screen_rescue && !ctx->tracing && screen_first16_usable(ctx) &&
                                   (rs & (kReasonPositions | kReasonLog | kReasonColumn)) != 0u &&
                                   (rs & (kReasonRemoval | kReasonIrregular | kReasonTie | kReasonGuard | kReasonNoCand | kReasonFirstState | kReasonOverflow)) == 0u;
            if (rescuable) { Route r = route; r.rescue = true; r.rescue_why = rs; return retry(r); }
            // the resident tier does not report this signal: the sub-dictionary tier (2048 columns) takes it next
            ctx->stats.screen_tier2 += 1;
            Route r = route; r.rescue = false; r.no_res = true;
            return retry(r);
        }
        if (hs.status == 0 && route.rescue) ctx->stats.screen_rescued += 1;
        if (hs.status == 0) { ctx->stats.screen_signals += 1; ctx->stats.screen_resident += 1; }
    }
    if ((scr1 || scr64) && (hs.status == kStatusSubsetDecline || hs.status == kStatusSubsetFail || scr_tie)) {
        const uint32_t rs = hs.sub_reason;
        if (!route.rescue) count_reasons(ctx, rs, scr_tie);            // (why the FIRST attempt declined)
        if (std::getenv("SS_HIP_SUB_DEBUG")) {
            std::fprintf(stderr, "[screened form%s] status %u reason 0x%x after %u iterations, %u states logged, K = %u, lambda %g, lambda0 %g\n",
                         route.rescue ? ", rescue" : "", hs.status, hs.sub_reason, hs.iter, hs.solo_nlog, hs.K, hs.c_inf, (double)hs.lambda0);
            if (scr1) screen_debug_recheck(ctx);
        }
        // The rescue (fp32 Homotopy, option screen_rescue): the path ran out of positions / states, or an outside column beat a state, and
        // nothing else was wrong with it — typically a planted column the ranking left out of the subset.  The next attempt scans this
        // attempt's log for such columns (screen.hip: launch_screen_rescue_scan) and repeats the form with them; once.
        const bool rescuable = scr1 && !omp && !route.rescue && !scr_tie && ctx->screen_rescue && ctx->screen_resident && !ctx->tracing &&
                               screen_first16_usable(ctx) && (rs & (kReasonPositions | kReasonLog | kReasonColumn)) != 0u &&
                               (rs & (kReasonRemoval | kReasonIrregular | kReasonTie | kReasonGuard | kReasonNoCand | kReasonFirstState | kReasonOverflow)) == 0u;
        if (rescuable) { Route r = route; r.rescue = true; r.rescue_why = rs; return retry(r); }
        ctx->stats.screen_redone += 1;
        Route r = route; r.rescue = false; r.no_sub = true;
        return retry(r);
    }
    if (scr1 && hs.status == 0 && route.rescue) ctx->stats.screen_rescued += 1;
    if ((scr1 || scr64) && hs.status == 0) ctx->stats.screen_signals += 1;
    if (scr1 && hs.status == 0 && ctx->screen_resident && res_solve_usable<float>()) ctx->stats.screen_resident += 1;
    if ((scr1 || scr64r) && hs.status == 0 && (hs.sub_reason & kReasonRechecked)) ctx->stats.screen_recheck += 1;
    if (scr1 && ctx->sub_dbg != nullptr) {
        unsigned long long tp[9];
        HIPCHK(hipMemcpy(tp, ctx->sub_dbg, sizeof(tp), hipMemcpyDeviceToHost));
        const double r = tp[8] ? (double)tp[8] : 1.0;
        std::fprintf(stderr, "[k_sub_solve, cycles per round over %llu rounds] chain %.0f  max|c| %.0f  log+scan %.0f  arg-min %.0f  hand-shake+x %.0f  u1/u2 %.0f  "
                             "inverse+signs %.0f  direction %.0f\n", tp[8], tp[0] / r, tp[1] / r, tp[2] / r, tp[3] / r, tp[4] / r, tp[5] / r, tp[6] / r, tp[7] / r);
    }
    if (sub1 && (hs.status == kStatusSubsetDecline || hs.status == kStatusSubsetFail)) {
        ctx->stats.subset_redone += 1;
        count_reasons(ctx, hs.sub_reason, false);
        Route r = route; r.no_sub = true;
        return retry(r);
    }
    if (sub1 && hs.status == 0) ctx->stats.subset_signals += 1;
    if (la && hs.status == kStatusRetryPlain) {
        // the early form's first launch used too many columns beyond the prefetched ones: plain form for this solve
        // (solve_impl switches option early_solo off until this solve has returned)
        Route r = route.after(); r.omp = omp; r.force_residual = force_residual; r.no_solo = no_solo; r.plain = true;
        return retry(r);
    }
    if ((la || la_omp) && hs.status == kStatusRetryResidual) {
        // tolerance too tight for Gram-form correlations (see k_la_init_pick): residual form
        ctx->stats.gram_fallbacks += 1;
        Route r = route.after(); r.omp = omp; r.force_residual = true; r.no_solo = no_solo;
        return retry(r);
    }
    if (la && hs.status == SS_HIP_ERUNTIME && ctx->la_fused >= 2 && (ctx->persist_workers[0] != 0 || ctx->persist_workers[1] != 0)) {
        // the resident kernel gave up on a wait (its grid was not fully resident): this context
        // goes on with one launch per iteration
        ctx->persist_workers[0] = 0;
        ctx->persist_workers[1] = 0;
        ctx->stats.persist_fallbacks += 1;
        Route r = route.after(); r.omp = omp; r.force_residual = force_residual; r.no_solo = no_solo;
        return retry(r);
    }
    if ((la || la_omp) && hs.status == SS_HIP_ERUNTIME && ctx->la_fused >= 1) {
        // k_la_iter's grid barrier expired as well (the GPU is shared with another resident grid):
        // from here on this context uses the form without in-kernel grid synchronisation
        ctx->la_fused = 0;
        ctx->stats.persist_fallbacks += 1;
        Route r = route.after(); r.omp = omp; r.force_residual = force_residual; r.no_solo = no_solo;
        return retry(r);
    }
    if (hs.status != 0) {
        set_err(err, errlen, hs.status == SS_HIP_ECAPACITY
                                 ? "solve: active set outgrew the workspace capacity (4096 columns)"
                                 : "solve: internal error, a device-side wait expired");
        return (int)hs.status;
    }
    *report = true;
    return SS_HIP_OK;
}

template <typename T>
int solve_impl(ss_hip_ctx* ctx, const T* y, ptrdiff_t incy, T tol, uint32_t max_iter, T* x,
               ptrdiff_t incx, uint32_t* iter_out, double* err_out, char* err, size_t errlen,
               Route route = Route(), void* rec_out = nullptr, uint32_t kmax = 0);

// One attempt.  Returns the status of the solve, or — with *again set — the route of the next attempt in *next.
template <typename T>
int solve_once(ss_hip_ctx* ctx, const T* y, ptrdiff_t incy, T tol, uint32_t max_iter, T* x,
               ptrdiff_t incx, uint32_t* iter_out, double* err_out, char* err, size_t errlen,
               const Route& route, void* rec_out, uint32_t kmax, Route* next, bool* again)
{
    const bool omp = route.omp, force_residual = route.force_residual, no_solo = route.no_solo;
    auto retry = [&](const Route& r) { *next = r; *again = true; return SS_HIP_OK; };
    try {
        HIPCHK(hipSetDevice(ctx->device));
        const size_t m = ctx->m, n = ctx->n;
        const uint32_t kcap = (uint32_t)std::min<uint64_t>(
            std::min<uint64_t>(n, (uint64_t)max_iter + 1), kKcapLimit);
        ensure_workspace<T>(ctx, 1, kcap);
        Workspace<T>& ws = *ws_of<T>(ctx);
        hipStream_t st = ctx->stream;
        const uint32_t want_trace = ctx->tracing ? (uint32_t)std::min<uint64_t>((uint64_t)max_iter + 2, 1u << 20) : 0u;
        if (want_trace > ws.trace_cap) {
            if (ws.trace) HIPCHK(hipFree(ws.trace));
            ws.trace = nullptr;
            ws.trace_cap = 0;
            HIPCHK(hipMalloc(&ws.trace, (size_t)want_trace * sizeof(TraceEntry)));
            ws.trace_cap = want_trace;
        }
        TraceEntry* const trace_keep = ws.trace;
        if (!ctx->tracing) ws.trace = nullptr;          // kernels skip the stores
        struct Restore { Workspace<T>& w; TraceEntry* p; ~Restore() { w.trace = p; } } restore{ ws, trace_keep };
        // (option profile_solve_every = k: with profiling on, only every k-th solve carries the HIP events — each costs
        // stream time, ~0.07 ms per solve in all at 8192 x 65536)
        const bool prof = ctx->profiling != 0 && (ctx->profile_solve_every <= 1 || (ctx->prof_solve_tick++ % (uint64_t)ctx->profile_solve_every) == 0);
        size_t nprof = 0;
        ctx->prof_kind.clear();

        ctx->host_flags[0] = 0;     // the stream is idle here: the previous solve synchronised
        ctx->host_flags[1] = 0;
        ctx->host_flags[2] = 0;
        ctx->host_flags[3] = 0;
        ctx->host_flags[4] = 0;
        if (prof) HIPCHK(hipEventRecord(ctx->ev_solve0, st));
        Forms forms = choose_forms<T>(ctx, route, y, rec_out);
        // (lookahead engine with the signal on the device: k_la_reset reads it from the caller's buffer — no copy command)
        const bool y_direct = forms.y_direct;
        if (!y_direct) copy_in<T>(ctx, ws.y, y, incy, m);
        // reference-order engine (reforder.hip; option engine = 3, and the arbiter of tie stalls): the reference's
        // iteration with every reduction in the documented 8-partial order, two passes over A per iteration
        const bool ro = forms.ro;
        const bool la_path = forms.la;
        if (!la_path) {
            HIPCHK(hipMemsetAsync(ws.x, 0, (size_t)ctx->n_pad * sizeof(T), st));
            HIPCHK(hipMemsetAsync(ws.d, 0, (size_t)ctx->n_pad * sizeof(T), st));
            HIPCHK(hipMemsetAsync(ws.insup, 0, (size_t)ctx->n_pad, st));
            HIPCHK(hipMemsetAsync(ws.st, 0, sizeof(DevState), st));
            HIPCHK(hipMemsetAsync(ws.ndone, 0, sizeof(uint32_t), st));
        }
        if (!la_path) HIPCHK(hipMemcpyAsync(ws.rhs, ws.y, (size_t)ctx->ldm * sizeof(T), hipMemcpyDeviceToDevice, st));
        const size_t rhs_stride = (size_t)ws.dims.b_pad * ctx->ldm;   // r-block -> p-block

        const bool la = forms.la;
        // orthogonal matching pursuit in Gram form (k_la_omp): same cache, same sweeps
        const bool la_omp = forms.la_omp;
        bool solo = false, solo_started = false, early = false;
        uint32_t early_lds_cols = 0, ro_parts = 0;
        bool sub1 = forms.sub1, scr1 = forms.scr1, scr64 = forms.scr64, scr64r = forms.scr64r;
        // (a context whose signals a form hands back more often than not steps that form aside for a while: the counters below)
        if ((sub1 || scr1 || scr64 || scr64r) && ctx->sub_off_solves > 0) { ctx->sub_off_solves -= 1; sub1 = false; scr1 = false; scr64 = false; scr64r = false; }
        if (scr64r && ctx->res_off_solves > 0) { ctx->res_off_solves -= 1; scr64r = false; }
        if (route.rescue && !scr1 && !scr64r) ctx->stats.screen_redone += 1;      // (the form has stepped aside meanwhile: the signal is the default engine's after all)
        if (scr64r) scr64 = false;
        uint32_t scr_launches = 1;
        // end of a solve: the device state to pinned memory, x (and the compact record) to the caller
        bool spec_epilogue = false, pump_enqueued = false;
        const bool x_on_device = x != nullptr && is_device_pointer(x);
        auto enqueue_epilogue = [&]() {
            if (ctx->hs_mapped != nullptr) {
                // one launch: state -> pinned memory, x -> the caller's device buffer (a host x still takes a copy command)
                const uint32_t grid = x_on_device ? std::min<uint32_t>(((uint32_t)n + 1023u) / 1024u, 256u) : 1u;
                hipLaunchKernelGGL((k_epilogue<T>), dim3(std::max(1u, grid)), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(ws.st),
                                   static_cast<uint32_t*>(ctx->hs_mapped), (uint32_t)(sizeof(DevState) / 4), (const T*)ws.x,
                                   x_on_device ? x : (T*)nullptr, (long long)incx, (uint32_t)n,
                                   (scr1 || scr64r) ? (uint32_t)(offsetof(DevState, status) / 4) : 0xffffffffu, (uint32_t)(offsetof(DevState, need_sweep) / 4));
                HIPCHK(hipGetLastError());
                if (x && !x_on_device) copy_out<T>(ctx, x, incx, ws.x, n);
            } else {
                HIPCHK(hipMemcpyAsync(ctx->hs_pinned, ws.st, sizeof(DevState), hipMemcpyDeviceToHost, st));
                if (x) copy_out<T>(ctx, x, incx, ws.x, n);
            }
            if (rec_out) {
                // compact output (a record that is superseded by a retry below is simply overwritten)
                const unsigned char* stage = pack_records<T>(ctx, ws, 1, kmax);
                HIPCHK(hipMemcpyAsync(rec_out, stage, record_bytes(kmax, sizeof(T)), hipMemcpyDefault, st));
            }
            if (prof) HIPCHK(hipEventRecord(ctx->ev_solve1, st));
        };
        // full-G mode (fp32): G = A^T A of the context as the cache.  G exists once a large batch has run
        // on the context, or — opt-in, option gram_full_after > 0 — is made here after that many single-signal
        // solves (17 GiB and a few tenths of a second at C2 against ~0.6 ms saved per solve from then on).
        struct FullGramView {
            Workspace<T>& w; bool on = false;
            ~FullGramView() { if (on) { w.gcache = w.gcache_own; w.gpitch = w.gpitch_own; w.slot_of = w.slot_of_own; w.gram_is_full = false; } }
        } full_view{ ws };
        auto enter_full_gram = [&]() {
            if (sizeof(T) != 4 || ctx->engine < 1 || ws.gram_is_full || !ctx->gram_single) return;     // (a retry runs inside the outer view)
            if (!ctx->gram_full && ctx->gram_full_after > 0 && ctx->single_solves + 1 >= (uint64_t)ctx->gram_full_after)
                (void)ensure_full_gram(ctx);
            if (!ctx->gram_full) return;
            if (!ws.slot_identity) {
                std::vector<int32_t> iota(ctx->n_pad);
                for (uint32_t i = 0; i < ctx->n_pad; ++i) iota[i] = (int32_t)i;
                HIPCHK(hipMalloc(&ws.slot_identity, (size_t)ctx->n_pad * sizeof(int32_t)));
                HIPCHK(hipMemcpy(ws.slot_identity, iota.data(), (size_t)ctx->n_pad * sizeof(int32_t), hipMemcpyHostToDevice));
            }
            ws.gcache_own = ws.gcache; ws.gpitch_own = ws.gpitch; ws.slot_of_own = ws.slot_of;
            ws.gcache = reinterpret_cast<T*>(ctx->gram_full);
            ws.gpitch = ctx->gram_pitch;
            ws.slot_of = ws.slot_identity;
            ws.gram_is_full = true;
            full_view.on = true;
        };
        if (scr64r) {
            Lookahead<T>::ensure(ctx, ws, kcap);
            bool rescue = false;
            if (route.rescue) {
                uint32_t found = 0;
                const bool from_recheck = (route.rescue_why & kReasonRechecked) != 0u && (route.rescue_why & (kReasonPositions | kReasonLog)) == 0u;
                HIPCHK(scr64_rescue_scan(ctx, ws, tol, from_recheck, &found));
                rescue = found >= 1u && found <= screen_rescue_cap();
                if (std::getenv("SS_HIP_SUB_DEBUG")) std::fprintf(stderr, "[screened form, fp64 resident tier, rescue] the scan lists %u columns the ranking missed\n", found);
                if (!rescue) { ctx->stats.screen_tier2 += 1; Route r = route; r.rescue = false; r.no_res = true; return retry(r); }
                ctx->stats.screen_rescue_tried += 1;
            }
            if (!omp) HIPCHK(launch_la_reset<T>(ctx, ws, false, y_direct ? y : (const T*)nullptr, incy));     // x, d, flags, DevState, r = y (OMP: done above)
            const bool first16 = screen_first16_usable(ctx);
            uint32_t nb1 = 0;
            if (!first16) {
                if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof), st)); }
                HIPCHK(launch_sweep<T>(ctx, ws.rhs, rhs_stride, 1, ws.c0, nullptr, ws.pmax_val, ws.pmax_idx, &nb1, ws.st));
                if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof + 1), st)); ctx->prof_kind.push_back(1); ++nprof; }
            }
            hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr, e4 = nullptr, e5 = nullptr;
            if (prof && first16 && !rescue) { e0 = prof_event(ctx, 2 * nprof); e1 = prof_event(ctx, 2 * nprof + 1); ctx->prof_kind.push_back(7); ++nprof; }
            if (prof) { e2 = prof_event(ctx, 2 * nprof); e3 = prof_event(ctx, 2 * nprof + 1); e4 = prof_event(ctx, 2 * nprof + 2); e5 = prof_event(ctx, 2 * nprof + 3); }
            HIPCHK(scr64_resident(ctx, ws, tol, max_iter, first16, omp, e0, e1, e2, e3, e4, e5, rescue));
            if (prof) { ctx->prof_kind.push_back(6); ctx->prof_kind.push_back(8); nprof += 2; }      // (6 = the screening pass, 8 = the path kernel)
        } else if (scr64) {
            Lookahead<T>::ensure(ctx, ws, kcap);
            if (!omp) HIPCHK(launch_la_reset<T>(ctx, ws, false, y_direct ? y : (const T*)nullptr, incy));     // x, d, flags, DevState, r = y (OMP: done above)
            // (the first pass — A^T y over all columns, which here only ranks them — over the fp16 copy: k_scr_first; 7 = that pass)
            const bool first16 = screen_first16_usable(ctx);
            uint32_t nb1 = 0;
            if (first16) {
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (prof) { e0 = prof_event(ctx, 2 * nprof); e1 = prof_event(ctx, 2 * nprof + 1); ctx->prof_kind.push_back(7); ++nprof; }
                HIPCHK(scr64_gather(ctx, (const T*)nullptr, ws.rhs, e0, e1));
            } else {
                if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof), st)); }
                HIPCHK(launch_sweep<T>(ctx, ws.rhs, rhs_stride, 1, ws.c0, nullptr, ws.pmax_val, ws.pmax_idx, &nb1, ws.st));
                if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof + 1), st)); ctx->prof_kind.push_back(1); ++nprof; }
                HIPCHK(scr64_gather(ctx, ws.c0, ws.rhs, nullptr, nullptr));
            }
            HIPCHK(hipStreamSynchronize(st));
            bool handed_back = true;
            if constexpr (sizeof(T) == 8) {
                ss_hip_ctx* sub = screen64_sub(ctx);
                sub->strict_sign = ctx->strict_sign; sub->zero_on_removal = ctx->zero_on_removal; sub->tie_guard = ctx->tie_guard;
                sub->tie_rerun = ctx->tie_rerun; sub->engine = ctx->engine; sub->lookahead = ctx->lookahead;
                sub->sweep_f64_variant = 2;               // (128-column tiles: the sub-dictionary has 16 of them)
                const uint64_t ties0 = sub->stats.tie_reruns, gf0 = sub->stats.gram_fallbacks, pf0 = sub->stats.persist_fallbacks;
                uint32_t it_s = 0;
                double e_s = 0.0;
                Route rs; rs.omp = omp; rs.no_sub = true;
                const int rc_s = solve_impl<T>(sub, ws.rhs, 1, tol, max_iter, screen64_xsub(ctx), 1, &it_s, &e_s, err, errlen, rs, nullptr, 0);
                HIPCHK(hipSetDevice(ctx->device));
                const bool clean = rc_s == SS_HIP_OK && sub->stats.tie_reruns == ties0 && sub->stats.gram_fallbacks == gf0 &&
                                   sub->stats.persist_fallbacks == pf0 && it_s >= 1u && it_s <= 192u;
                if (clean) {
                    const DevState hsub = *static_cast<const DevState*>(sub->hs_pinned);
                    hipEvent_t e2 = nullptr, e3 = nullptr;
                    if (prof) { e2 = prof_event(ctx, 2 * nprof); e3 = prof_event(ctx, 2 * nprof + 1); }
                    HIPCHK(scr64_certify(ctx, ws, ws.rhs, it_s, tol, e_s, hsub.K, e2, e3, omp, first16));
                    if (prof) { ctx->prof_kind.push_back(6); ++nprof; }
                    scr_launches = (it_s + 95u) / 96u;
                    handed_back = false;
                }
            }
            if (handed_back && std::getenv("SS_HIP_SUB_DEBUG")) {
                ss_hip_ctx* sub = screen64_sub(ctx);
                const DevState hsub = *static_cast<const DevState*>(sub->hs_pinned);
                std::fprintf(stderr, "[screened form, fp64] handed back: sub-context status %u iter %u K %u ties %llu gram fallbacks %llu persist fallbacks %llu (%s)\n",
                             hsub.status, hsub.iter, hsub.K, (unsigned long long)sub->stats.tie_reruns, (unsigned long long)sub->stats.gram_fallbacks,
                             (unsigned long long)sub->stats.persist_fallbacks, err ? err : "");
            }
            if (handed_back) {
                // the sub-context's solve left its common path (a tie re-run, a residual-form retry, too many states): the usual engine
                ctx->stats.screen_redone += 1;
                Route r = route; r.no_sub = true;
                return retry(r);
            }
        } else if (la_omp && !scr1) {
            Lookahead<T>::ensure(ctx, ws, kcap);
            enter_full_gram();
            uint32_t nb1 = 0;
            if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof), st)); }
            HIPCHK(launch_sweep<T>(ctx, ws.rhs, rhs_stride, 1, ws.c0, nullptr, ws.pmax_val, ws.pmax_idx, &nb1, ws.st));
            if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof + 1), st)); ctx->prof_kind.push_back(1); ++nprof; }
            if (!ws.gram_is_full) HIPCHK(hipMemsetAsync(ws.slot_of, 0xff, (size_t)ctx->n_pad * sizeof(int32_t), st));   // nothing cached yet
        } else if (sub1 || scr1) {
            Lookahead<T>::ensure(ctx, ws, kcap);
            bool rescue = false;
            if (route.rescue) {
                // the scan of the declined attempt's log — its state, logs and c~0 are still in place: nothing has been reset yet
                uint32_t found = 0;
                // (a decline by the exact re-check alone left the list of the columns that failed it: no scan)
                const bool from_recheck = (route.rescue_why & kReasonRechecked) != 0u && (route.rescue_why & (kReasonPositions | kReasonLog)) == 0u;
                if (scr1) HIPCHK(scr_rescue_scan(ctx, ws, tol, from_recheck, &found));
                rescue = scr1 && found >= 1u && found <= screen_rescue_cap();
                if (std::getenv("SS_HIP_SUB_DEBUG")) std::fprintf(stderr, "[screened form, rescue] the scan lists %u columns the ranking missed\n", found);
                if (!rescue) { ctx->stats.screen_redone += 1; Route r = route; r.rescue = false; r.no_sub = true; return retry(r); }
                ctx->stats.screen_rescue_tried += 1;
            }
            if (!omp) HIPCHK(launch_la_reset<T>(ctx, ws, false, y_direct ? y : (const T*)nullptr, incy));     // x, d, flags, DevState, r = y (OMP: done above)
            // (the screened form's first pass — A^T y over all columns — reads the half-precision copy too: screen.hip, k_scr_first)
            const bool first16 = scr1 && screen_first16_usable(ctx);
            uint32_t nb1 = 0;
            if (!first16) {
                if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof), st)); }
                HIPCHK(launch_sweep<T>(ctx, ws.rhs, rhs_stride, 1, ws.c0, nullptr, ws.pmax_val, ws.pmax_idx, &nb1, ws.st));
                if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof + 1), st)); ctx->prof_kind.push_back(1); ++nprof; }
            }
            if (sub1) {
                HIPCHK(sub_single(ctx, ws, tol, max_iter));
            } else {
                // (6 = the screening pass over the fp16 copy of A, 7 = the first pass when it runs there too)
                hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr, e4 = nullptr, e5 = nullptr;
                if (prof && first16 && !rescue) { e0 = prof_event(ctx, 2 * nprof); e1 = prof_event(ctx, 2 * nprof + 1); ctx->prof_kind.push_back(7); ++nprof; }
                if (prof) { e2 = prof_event(ctx, 2 * nprof); e3 = prof_event(ctx, 2 * nprof + 1); e4 = prof_event(ctx, 2 * nprof + 2); e5 = prof_event(ctx, 2 * nprof + 3); }
                HIPCHK(scr_single(ctx, ws, tol, max_iter, first16, e0, e1, e2, e3, e4, e5, omp, rescue));
                if (prof) { ctx->prof_kind.push_back(6); ctx->prof_kind.push_back(8); nprof += 2; }      // (6 = the screening pass, 8 = the path kernel)
            }
        } else if (la) {
            Lookahead<T>::ensure(ctx, ws, kcap);
            enter_full_gram();
            HIPCHK(launch_la_reset<T>(ctx, ws, !ws.gram_is_full, y_direct ? y : (const T*)nullptr, incy));     // x, d, flags, slot map, exchange area, DevState, r = y
            uint32_t nb1 = 0;
            if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof), st)); }
            HIPCHK(launch_sweep<T>(ctx, ws.rhs, rhs_stride, 1, ws.c0, nullptr, ws.pmax_val, ws.pmax_idx, &nb1, ws.st));
            if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof + 1), st)); ctx->prof_kind.push_back(1); ++nprof; }
            // Speculative form (fp32): the default
            // (la_fused = 3).  It also runs where the resident kernel cannot (dictionaries too wide for one
            // launch to own every column).
            const bool solo_wanted = ctx->la_fused >= 3;
            // (not with the full Gram matrix as the cache: there every entering column's row slice is a gather of
            // 256 scattered entries of a 256-KiB row in the iteration's chain — 1.81 ms per C2 solve against 1.55 ms)
            solo = solo_wanted && !no_solo && ctx->solo_off_solves == 0 && sizeof(T) == 4 &&
                   (!ws.gram_is_full || ctx->solo_full_gram) && la_solo_usable(ctx);
            if (solo_wanted && !solo && ctx->solo_off_solves > 0 && !no_solo) ctx->solo_off_solves -= 1;
            solo_started = solo;
            early = solo && ctx->early_solo && sizeof(T) == 4 && !ws.gram_is_full && ctx->n > 32u * 512u && ws.gcap >= 160 &&
                    ctx->sweep32_variant == 0 && ws.subg != nullptr && !ws.la_dbg;
            if (early) {
                // resident tier of the launch queued behind the solo group (the last step of the path)
                uint32_t lc = std::min<uint32_t>((ws.dims.kcap + 15u) & ~15u, kLaLdsSmall);
                if (ctx->la_fused < 2 || !la_persist_usable(ctx, lc)) lc = 0;
                early_lds_cols = lc;
                hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
                if (prof) { e0 = prof_event(ctx, 2 * nprof); e1 = prof_event(ctx, 2 * nprof + 1); e2 = prof_event(ctx, 2 * nprof + 2); e3 = prof_event(ctx, 2 * nprof + 3); }
                early_prologue(ctx, ws, nb1, tol, max_iter, lc, e0, e1, e2, e3);
                // 5 = the main launch of an early-form pass (its share of the columns): both passes of the solve are timed
                if (prof) { ctx->prof_kind.push_back(5); ctx->prof_kind.push_back(5); nprof += 2; }
                // The typical solve is complete with what is queued now: its epilogue (state, x, record) goes right
                // behind instead of after a trip through the host (host notices `done`, three enqueues: ~70 us).
                // Should the pump below have to queue more work, the epilogue is simply issued again at the end.
                enqueue_epilogue();
                spec_epilogue = true;
            } else {
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (prof) { e0 = prof_event(ctx, 2 * nprof); e1 = prof_event(ctx, 2 * nprof + 1); }
                const uint32_t fc = Lookahead<T>::init(ctx, ws, nb1, tol, solo, e0, e1);
                // (events were recorded only if a sweep was launched; 4 = the 64-column first sweep, 3 = a 32-column one)
                if (prof && fc != 0) { ctx->prof_kind.push_back(fc > 32 ? 4 : 3); ++nprof; }
            }
        } else if (ro) {
            // c = A^T y in reference order, first pick with the column norm as a chain dot product
            HIPCHK(launch_ro_sweep<T>(ctx, ws.rhs, 0, ws.c, 0, ws.dims.n_pad, 1, 1u, ws.pmax_val, ws.pmax_idx, ws.dims.pmax_stride, &ro_parts, ws.st, false));
            HIPCHK(launch_ro_init<T>(ctx, ws, 1u, ro_parts, tol));
        } else if (!omp) {
            // c = A^T y  (residual_vector with x = 0, homotopy-cpu.cpp:215)
            uint32_t nb1 = 0;
            if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof), st)); }
            HIPCHK(launch_sweep<T>(ctx, ws.rhs, rhs_stride, 1, ws.c, nullptr, ws.pmax_val, ws.pmax_idx, &nb1, ws.st));
            if (prof) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * nprof + 1), st)); ctx->prof_kind.push_back(1); ++nprof; }
            HIPCHK(launch_init<T>(ctx, ws, 1, nb1, tol));
            HIPCHK(launch_rp<T>(ctx, ws, 1));
        }

        PumpState ps;
        ps.solo = solo; ps.solo_started = solo_started; ps.early = early; ps.early_lds_cols = early_lds_cols; ps.nprof = nprof;
        if (sub1 || scr1 || scr64 || scr64r) {
            // (everything is queued: selection, the solve, the check)
        } else if ((la && ctx->la_fused) || la_omp) {
            if (!pump_fused<T>(ctx, ws, tol, max_iter, la, la_omp, prof, ps)) {
                set_err(err, errlen, "solve: internal error, lookahead loop made no progress");
                return SS_HIP_ERUNTIME;
            }
        } else {
            pump_rounds<T>(ctx, ws, tol, max_iter, la, ro, omp, ro_parts, rhs_stride, prof, ps);
        }
        solo = ps.solo; nprof = ps.nprof; pump_enqueued = ps.enqueued;

        if (!(spec_epilogue && !pump_enqueued)) enqueue_epilogue();
        HIPCHK(hipStreamSynchronize(st));
        const DevState hs = *static_cast<const DevState*>(ctx->hs_pinned);
        Forms eff = forms;
        eff.sub1 = sub1; eff.scr1 = scr1; eff.scr64 = scr64; eff.scr64r = scr64r;      // (after the step-aside counters)
        {
            bool report = false;
            const int vrc = attempt_verdict<T>(ctx, route, eff, hs, next, again, &report, err, errlen);
            if (!report) return vrc;
        }
        if (iter_out) *iter_out = hs.iter;
        if (err_out) *err_out = hs.c_inf;
        if (la && ws.la_dbg && !scr64) {                 // (fp64 screened form: the stamps are the sub-context's, dumped by its solve)
            if (const char* path = std::getenv("SS_HIP_LA_DEBUG")) {
                std::vector<uint64_t> tsb(2048 * 8);
                HIPCHK(hipMemcpy(tsb.data(), ws.la_dbg, tsb.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
                if (FILE* fp = std::fopen(path, "wb")) { std::fwrite(tsb.data(), sizeof(uint64_t), tsb.size(), fp); std::fclose(fp); }
            }
        }
        ctx->last_trace.clear();
        if (ctx->tracing && ws.trace) {
            // entry 0 = the initial pick, entry t = the toggle of iteration t
            const size_t cnt = std::min<size_t>((size_t)hs.iter + 1, ws.trace_cap);
            ctx->last_trace.resize(cnt);
            HIPCHK(hipMemcpy(ctx->last_trace.data(), ws.trace, cnt * sizeof(TraceEntry), hipMemcpyDeviceToHost));
        }

        ctx->stats.solves += 1;
        ctx->single_solves += 1;
        if (solo_started) {
            // Speculative launches that failed their check cost a replay and the rest of the solve in the
            // resident form; contexts whose problems do that on most solves stop speculating for a while.
            ctx->stats.solo_solves += 1;
            ctx->stats.solo_retries += hs.solo_fails;
            // (private counters: the decision must not depend on whether the caller reset the statistics)
            ctx->solo_seen += 1;
            ctx->solo_failed += hs.solo_fails;
            if (ctx->solo_seen >= 8 && 2 * ctx->solo_failed > ctx->solo_seen) ctx->solo_off_solves = 64;
            const char* path = hs.solo_fails ? std::getenv("SS_HIP_SOLO_DEBUG") : nullptr;
            if (path != nullptr) dump_solo_debug<T>(ws, hs, path);
        }
        ctx->stats.iterations += hs.iter;
        if (la || la_omp) ctx->stats.lookahead_sweeps += hs.nsweeps;
        if (ro) ctx->stats.ro_resweeps += hs.nsweeps;
        if (prof) account_profile<T>(ctx, nprof, scr_launches, hs);
    } catch (const HipFail& f) {
        set_err(err, errlen, hip_msg(f));
        return SS_HIP_ERUNTIME;
    } catch (const std::bad_alloc&) {
        set_err(err, errlen, "solve: out of host memory");
        return SS_HIP_ENOMEM;
    }
    return SS_HIP_OK;
}

// The solve: validate once, then attempts until one of them reports (at most one per rung of the ladder: every retry sets a
// flag of the route or changes the context so that the same rung cannot be taken twice).
template <typename T>
int solve_impl(ss_hip_ctx* ctx, const T* y, ptrdiff_t incy, T tol, uint32_t max_iter, T* x,
               ptrdiff_t incx, uint32_t* iter_out, double* err_out, char* err, size_t errlen,
               Route route, void* rec_out, uint32_t kmax)
{
    if (!ctx) { set_err(err, errlen, "solve: null context"); return SS_HIP_EINVAL; }
    if (ctx->kind != 0) { set_err(err, errlen, "solve: this context was created for IRLS"); return SS_HIP_EINVAL; }
    if (ctx->is_f64 != (sizeof(T) == 8)) {
        set_err(err, errlen, "solve: element type of the call does not match the context");
        return SS_HIP_ETYPE;
    }
    if (!y || (!x && !rec_out)) { set_err(err, errlen, "solve: y and x must not be null"); return SS_HIP_EINVAL; }
    // preconditions the reference asserts (homotopy-cpu.cpp:193-199)
    if (max_iter == 0) { set_err(err, errlen, "solve: max_iterations must be > 0"); return SS_HIP_EINVAL; }
    if (!(tol >= std::numeric_limits<T>::epsilon() && tol < T(1))) {
        set_err(err, errlen, "solve: tolerance must satisfy eps <= tolerance < 1");
        return SS_HIP_EINVAL;
    }
    if (incy <= 0 || incx <= 0) {
        set_err(err, errlen, "solve: vector increments must be positive");
        return SS_HIP_EINVAL;
    }
    // (a retry in the plain speculative form runs — with everything it may fall back to in turn — with option early_solo off)
    struct EarlyKeep { ss_hip_ctx* c; int keep; bool on = false; ~EarlyKeep() { if (on) c->early_solo = keep; } } early_keep{ ctx, ctx->early_solo };
    for (int attempt = 0; attempt < 16; ++attempt) {
        Route next;
        bool again = false;
        if (route.plain && !early_keep.on) { early_keep.on = true; ctx->early_solo = 0; }
        const int rc = solve_once<T>(ctx, y, incy, tol, max_iter, x, incx, iter_out, err_out, err, errlen, route, rec_out, kmax, &next, &again);
        if (!again) return rc;
        route = next;
    }
    set_err(err, errlen, "solve: internal error, the forms kept handing the signal on");
    return SS_HIP_ERUNTIME;
}

// ---- batched solve: B signals share the sensing matrix and advance in lock-step ---------
// Per round the 2*B correlation GEMVs are two MFMA GEMMs ([c] = R·Atᵀ, [q] = P·Atᵀ, gemm.hip);
// the active-set tail runs for all signals at once (grid.y = slot).  A signal that has
// terminated turns its kernels into no-ops; the solve ends when every slot is done.
// the full Gram matrix G = A^T A for the batched Gram form: one GEMM of 2 m n^2 flops on the MFMA units
// (0.55 s at C2), kept in the context for later batches; false if it does not fit the budget
// G's memory ahead of time, on a helper thread (ss_hip_ctx::gram_reserve_thread): started by the first batch of >= 4 signals a
// context receives, where G would fit the budget and is at most an eighth of the device's memory
void gram_reserve_start(ss_hip_ctx* ctx, size_t B)
{
    if (!ctx->gram_reserve || ctx->gram_full || ctx->gram_reserve_thread || B < 4 || ctx->engine < 1 || ctx->batch_gram_min <= 0) return;
    const size_t np = ctx->n_pad;
    const size_t bytes = np * ((np + 1023) / 1024 * 1024) * sizeof(float);
    if (ctx->gram_full_gib <= 0 || bytes > ((size_t)ctx->gram_full_gib << 30)) return;
    size_t free_b = 0, total_b = 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
    if (bytes > total_b / 8 || bytes + ((size_t)16 << 30) > free_b) return;
    const int device = ctx->device;
    float** slot = &ctx->gram_reserved;
    std::thread* th = new (std::nothrow) std::thread([device, bytes, slot]() {
        float* p = nullptr;
        if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
        *slot = p;
    });
    ctx->gram_reserve_thread = th;
}

bool ensure_full_gram(ss_hip_ctx* ctx)
{
    if (ctx->gram_full) return true;
    const size_t np = ctx->n_pad;
    const uint32_t pitch = (uint32_t)((np + 1023) / 1024 * 1024);
    const size_t bytes = np * (size_t)pitch * sizeof(float);
    if (ctx->gram_full_gib <= 0 || bytes > ((size_t)ctx->gram_full_gib << 30)) return false;
    float* G = nullptr;
    const auto t_alloc = std::chrono::steady_clock::now();
    if (ctx->gram_reserve_thread != nullptr) {
        // (reserved ahead on a helper thread: what is left of that allocation is all this batch waits for)
        std::thread* th = static_cast<std::thread*>(ctx->gram_reserve_thread);
        th->join();
        delete th;
        ctx->gram_reserve_thread = nullptr;
        G = ctx->gram_reserved;
        ctx->gram_reserved = nullptr;
    }
    if (G == nullptr) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes + ((size_t)8 << 30) > free_b) {
            (void)hipGetLastError();
            return false;
        }
        if (hipMalloc(&G, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    ctx->stats.gram_alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc).count();
    hipEvent_t g0 = nullptr, g1 = nullptr;
    if (hipEventCreate(&g0) != hipSuccess || hipEventCreate(&g1) != hipSuccess) { (void)hipGetLastError(); g0 = g1 = nullptr; }
    if (g0) (void)hipEventRecord(g0, ctx->stream);
    const hipError_t e = ctx->gram_symmetric
        ? launch_gemm_sym_f32(ctx, G, pitch)
        : launch_gemm_tn_f32(ctx, static_cast<const float*>(ctx->At), (uint32_t)np, ctx->ldm, G, pitch, nullptr, false);
    if (g1) (void)hipEventRecord(g1, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(G); throw HipFail{ e, "launch_gemm_tn_f32(full Gram)" }; }
    if (g0 && g1 && hipEventSynchronize(g1) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g0, g1) == hipSuccess) ctx->stats.gram_build_ms += ms;
    }
    (void)hipGetLastError();
    if (g0) (void)hipEventDestroy(g0);
    if (g1) (void)hipEventDestroy(g1);
    ctx->gram_full = G;
    ctx->gram_pitch = pitch;
    ctx->stats.gram_full_builds += 1;
    return true;
}

// Column form of mid-size batches: makes sure the context holds the Gram-column cache ((max_iter + 2) * per rows of
// n_pad fp32, rounded up to 1024 columns), the row tables and the pass lists for chunks of `per` signals.  Returns
// false — nothing thrown, the sticky error cleared — when that does not fit the budget (option gram_full_gib) or the
// free HBM, or an allocation fails: the dispatcher then runs the batch another way.  Sizes only grow; the buffers are
// released with the context.
bool ensure_bcol(ss_hip_ctx* ctx, size_t per, uint32_t max_iter)
{
    const size_t np = ctx->n_pad;
    const size_t pitchc = (np + 1023) / 1024 * 1024;
    const size_t rows_needed = ((size_t)max_iter + 2) * per;
    auto try_malloc = [](void** p, size_t bytes) {
        if (hipMalloc(p, bytes) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return false; }
        return true;
    };
    if (ctx->bcol_cache_rows < rows_needed) {
        const size_t bytes = rows_needed * pitchc * sizeof(float);
        const size_t held = ctx->bcol_cache_rows * pitchc * sizeof(float);
        if (ctx->gram_full_gib <= 0 || bytes > ((size_t)ctx->gram_full_gib << 30)) return false;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (bytes + ((size_t)4 << 30) > free_b + held) return false;          // keep 4 GiB for everything else
        if (ctx->bcol_cache) (void)hipFree(ctx->bcol_cache);
        ctx->bcol_cache = nullptr;
        ctx->bcol_cache_rows = 0;
        if (!try_malloc(reinterpret_cast<void**>(&ctx->bcol_cache), bytes)) return false;
        ctx->bcol_cache_rows = rows_needed;
    }
    if (ctx->bcol_slot_rows < per) {
        if (ctx->bcol_slot) (void)hipFree(ctx->bcol_slot);
        ctx->bcol_slot = nullptr;
        ctx->bcol_slot_rows = 0;
        if (!try_malloc(reinterpret_cast<void**>(&ctx->bcol_slot), per * np * sizeof(int32_t))) return false;
        ctx->bcol_slot_rows = per;
    }
    if (!ctx->bcol_lists && !try_malloc(reinterpret_cast<void**>(&ctx->bcol_lists), 2 * 1024 * sizeof(uint32_t))) return false;
    return true;
}

// Several signals in the reference-order engine at once (reforder.hip): up to ro_slots_max() of them share every pass
// over A — the sweep carries [r, p] of each — and run in lock-step like the batched forms; each signal's words are
// exactly those of a solve on its own (the slots share nothing but the dictionary tile in LDS).  Used for a batch's tie
// re-runs and for batches in engine 3.  sig: the signals' indices into Y / X / the outputs (nullptr: 0 .. count-1).
template <typename T>
int solve_batch_ro(ss_hip_ctx* ctx, const T* Y, const size_t* sig, size_t count, ptrdiff_t y_stride, ptrdiff_t incy, T tol,
                   uint32_t max_iter, T* X, ptrdiff_t x_stride, ptrdiff_t incx, uint32_t* iter_out, double* err_out,
                   char* err, size_t errlen, void* rec_out, uint32_t kmax)
{
    if (max_iter == 0) { set_err(err, errlen, "solve_batch: max_iterations must be > 0"); return SS_HIP_EINVAL; }
    if (!(tol >= std::numeric_limits<T>::epsilon() && tol < T(1))) {
        set_err(err, errlen, "solve_batch: tolerance must satisfy eps <= tolerance < 1");
        return SS_HIP_EINVAL;
    }
    if (incy <= 0 || incx <= 0) { set_err(err, errlen, "solve_batch: increments must be positive"); return SS_HIP_EINVAL; }
    try {
        HIPCHK(hipSetDevice(ctx->device));
        const size_t m = ctx->m, n = ctx->n, ldm = ctx->ldm, np = ctx->n_pad;
        const uint32_t kcap = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(n, (uint64_t)max_iter + 1), kKcapLimit);
        const size_t per = std::max<uint32_t>(1u, std::min<uint32_t>(ro_slots_max(ctx, sizeof(T) == 8), (uint32_t)std::max(1, ctx->ro_slots)));
        hipStream_t st = ctx->stream;
        std::vector<DevState> hs;
        for (size_t j0 = 0; j0 < count; j0 += per) {
            const uint32_t R = (uint32_t)std::min(per, count - j0);
            const uint32_t Rg = R <= 2 ? R : (R <= 4 ? 4u : 8u);  // the sweep carries 1, 2, 4 or 8 slots: the workspace holds that many
            auto gidx = [&](uint32_t b) { return sig ? sig[j0 + b] : j0 + b; };
            ensure_workspace<T>(ctx, Rg, kcap);
            Workspace<T>& ws = *ws_of<T>(ctx);
            const uint32_t want_trace = (ctx->tracing && gidx(0) == 0) ? (uint32_t)std::min<uint64_t>((uint64_t)max_iter + 2, 1u << 20) : 0u;
            if (want_trace > ws.trace_cap) {
                if (ws.trace) HIPCHK(hipFree(ws.trace));
                ws.trace = nullptr;
                ws.trace_cap = 0;
                HIPCHK(hipMalloc(&ws.trace, (size_t)want_trace * sizeof(TraceEntry)));
                ws.trace_cap = want_trace;
            }
            TraceEntry* const trace_keep = ws.trace;
            if (want_trace == 0u) ws.trace = nullptr;
            struct RestoreTrace { Workspace<T>& w; TraceEntry* p; ~RestoreTrace() { w.trace = p; } } restore_trace{ ws, trace_keep };

            ctx->host_flags[0] = 0;
            ctx->host_flags[1] = 0;
            HIPCHK(hipMemsetAsync(ws.y, 0, (size_t)Rg * ldm * sizeof(T), st));
            for (uint32_t b = 0; b < R; ++b) copy_in<T>(ctx, ws.y + (size_t)b * ldm, Y + (ptrdiff_t)gidx(b) * y_stride, incy, m);
            HIPCHK(hipMemsetAsync(ws.x, 0, (size_t)R * np * sizeof(T), st));
            HIPCHK(hipMemsetAsync(ws.d, 0, (size_t)R * np * sizeof(T), st));
            HIPCHK(hipMemsetAsync(ws.insup, 0, (size_t)R * np, st));
            HIPCHK(hipMemsetAsync(ws.st, 0, (size_t)Rg * sizeof(DevState), st));
            // (a slot of the group that carries no signal is "done" from the start: DevState::done is its first word)
            for (uint32_t b = R; b < Rg; ++b) HIPCHK(hipMemsetAsync(&ws.st[b], 0x01, sizeof(uint32_t), st));
            HIPCHK(hipMemsetAsync(ws.ndone, 0, sizeof(uint32_t), st));
            // (the sweep reads whole slot groups of 1, 2, 4, 8 right-hand sides: the slots beyond R carry zeros)
            const size_t bp = ws.dims.b_pad;
            HIPCHK(hipMemsetAsync(ws.rhs, 0, 2 * bp * ldm * sizeof(T), st));

            // c = A^T y of every slot in one pass (the slots' y rows are the right-hand sides), first picks
            uint32_t nparts = 0;
            HIPCHK(launch_ro_sweep<T>(ctx, ws.y, 0, ws.c, 0, ws.dims.n_pad, 1, R, ws.pmax_val, ws.pmax_idx, ws.dims.pmax_stride, &nparts,
                                      ws.st, false));
            HIPCHK(launch_ro_init<T>(ctx, ws, R, nparts, tol));
            const uint32_t L = (uint32_t)std::max(1, std::min(ctx->lookahead, 64));
            volatile uint32_t* hf = ctx->host_flags;
            const uint64_t last_round = (uint64_t)max_iter + 1;
            for (uint64_t round = 1; round <= last_round; ++round) {
                if (round > L) {
                    const uint32_t need = (uint32_t)(round - L);
                    uint32_t spins = 0;
                    while (hf[1] == 0 && hf[0] < need) {
                        if ((++spins & 0x3ffu) == 0) {
                            const hipError_t q = hipStreamQuery(st);
                            if (q == hipSuccess) break;
                            if (q != hipErrorNotReady) throw HipFail{ q, "hipStreamQuery(reference-order batch loop)" };
                        }
                        std::this_thread::yield();
                    }
                    if (hf[1] != 0) break;
                }
                HIPCHK(launch_ro_round<T>(ctx, ws, R, (uint32_t)round, nparts, tol, max_iter));
            }
            hs.resize(R);
            HIPCHK(hipMemcpyAsync(hs.data(), ws.st, (size_t)R * sizeof(DevState), hipMemcpyDeviceToHost, st));
            if (rec_out) {
                const size_t rb = record_bytes(kmax, sizeof(T));
                const unsigned char* stage = pack_records<T>(ctx, ws, R, kmax);
                for (uint32_t b = 0; b < R; ++b)
                    HIPCHK(hipMemcpyAsync(static_cast<unsigned char*>(rec_out) + gidx(b) * rb, stage + (size_t)b * rb, rb, hipMemcpyDefault, st));
            }
            if (X)
                for (uint32_t b = 0; b < R; ++b) copy_out<T>(ctx, X + (ptrdiff_t)gidx(b) * x_stride, incx, ws.x + (size_t)b * np, n);
            HIPCHK(hipStreamSynchronize(st));
            if (want_trace != 0u && ws.trace) {
                const size_t cnt = std::min<size_t>((size_t)hs[0].iter + 1, ws.trace_cap);
                ctx->last_trace.resize(cnt);
                HIPCHK(hipMemcpy(ctx->last_trace.data(), ws.trace, cnt * sizeof(TraceEntry), hipMemcpyDeviceToHost));
            }
            for (uint32_t b = 0; b < R; ++b) {
                if (!hs[b].done) { set_err(err, errlen, "solve_batch: internal error, a signal did not terminate"); return SS_HIP_ERUNTIME; }
                if (hs[b].status != 0) {
                    set_err(err, errlen, hs[b].status == SS_HIP_ECAPACITY ? "solve_batch: active set outgrew the workspace capacity"
                                                                          : "solve_batch: internal error, a device-side wait expired");
                    return (int)hs[b].status;
                }
                if (iter_out) iter_out[gidx(b)] = hs[b].iter;
                if (err_out) err_out[gidx(b)] = hs[b].c_inf;
                ctx->stats.iterations += hs[b].iter;
                ctx->stats.ro_resweeps += hs[b].nsweeps;
            }
            ctx->stats.solves += R;
        }
    } catch (const HipFail& f) {
        set_err(err, errlen, hip_msg(f));
        return SS_HIP_ERUNTIME;
    } catch (const std::bad_alloc&) {
        set_err(err, errlen, "solve_batch: out of host memory");
        return SS_HIP_ENOMEM;
    }
    return SS_HIP_OK;
}

// `gram`: Gram form — the correlations of every signal come from rows of G = A^T A
// (c = c0 - sum_j x_j G[j], q = sum_j d_j G[j]) instead of two GEMMs per round
int solve_batch_gemm_f32(ss_hip_ctx* ctx, const float* Y, size_t B, ptrdiff_t y_stride, ptrdiff_t incy,
                         float tol, uint32_t max_iter, float* X, ptrdiff_t x_stride, ptrdiff_t incx,
                         uint32_t* iter_out, double* err_out, char* err, size_t errlen, int form = 0,
                         void* rec_out = nullptr, uint32_t kmax = 0, bool no_subset = false)
{
    // form: 0 = two GEMMs per round (residual form), 1 = Gram form on the full G = A^T A, 2 = column form: Gram form
    // on a cache of the entering columns' Gram columns, formed round by round (mid-size batches, no G), 3 = screened form
    // (screen.hip): c0 by the batch GEMM, one workgroup per signal on its subset's own Gram matrix, one screening launch per chunk
    using T = float;
    const bool gram = form != 0, cols_form = form == 2;
    if (max_iter == 0) { set_err(err, errlen, "solve_batch: max_iterations must be > 0"); return SS_HIP_EINVAL; }
    if (!(tol >= std::numeric_limits<T>::epsilon() && tol < T(1))) {
        set_err(err, errlen, "solve_batch: tolerance must satisfy eps <= tolerance < 1");
        return SS_HIP_EINVAL;
    }
    if (incy <= 0 || incx <= 0) { set_err(err, errlen, "solve_batch: increments must be positive"); return SS_HIP_EINVAL; }
    try {
        HIPCHK(hipSetDevice(ctx->device));
        const size_t m = ctx->m, n = ctx->n, ldm = ctx->ldm, np = ctx->n_pad;
        const uint32_t kcap = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(n, (uint64_t)max_iter + 1), kKcapLimit);
        // (column form: chunks of at most 448 signals = 7 full passes per round, fewer if the cache budget says so)
        const bool scr_form = form == 3;
        const size_t chunk = scr_form ? (size_t)screen_batch_cap()
                                      : cols_form ? (size_t)std::max(1, ctx->bcol_chunk) : (size_t)std::max(4, ctx->batch_chunk);
        hipStream_t st = ctx->stream;
        std::vector<DevState> hs;
        std::vector<size_t> all_ties;                        // signals whose scan met a tie stall, over all chunks
        std::vector<size_t> redo;                            // subset form: signals it declined or whose check failed
        for (size_t b0 = 0; b0 < B; b0 += chunk) {
            const uint32_t Bc = (uint32_t)std::min(chunk, B - b0);
            ensure_workspace<T>(ctx, Bc, kcap);
            Workspace<T>& ws = *ws_of<T>(ctx);
            // option "trace": the path of the batch's FIRST signal (slot 0 of the first chunk) is recorded like a single solve's
            const uint32_t want_trace = (ctx->tracing && b0 == 0) ? (uint32_t)std::min<uint64_t>((uint64_t)max_iter + 2, 1u << 20) : 0u;
            if (want_trace > ws.trace_cap) {
                if (ws.trace) HIPCHK(hipFree(ws.trace));
                ws.trace = nullptr;
                ws.trace_cap = 0;
                HIPCHK(hipMalloc(&ws.trace, (size_t)want_trace * sizeof(TraceEntry)));
                ws.trace_cap = want_trace;
            }
            TraceEntry* const trace_keep = ws.trace;
            if (want_trace == 0u) ws.trace = nullptr;
            struct RestoreTrace { Workspace<T>& w; TraceEntry* p; ~RestoreTrace() { w.trace = p; } } restore_trace{ ws, trace_keep };
            const uint32_t rows = (Bc + 127u) / 128u * 128u;          // GEMM rows of each block
            const size_t bp = ws.dims.b_pad;
            T* const Rblk = ws.rhs;
            T* const Pblk = ws.rhs + bp * ldm;

            ctx->host_flags[0] = 0;
            ctx->host_flags[1] = 0;
            const T* Yc = Y + (ptrdiff_t)b0 * y_stride;
            if (incy == 1) {
                HIPCHK(hipMemcpy2DAsync(ws.y, ldm * sizeof(T), Yc, (size_t)y_stride * sizeof(T), m * sizeof(T), Bc,
                                        hipMemcpyDefault, st));
            } else {
                for (uint32_t b = 0; b < Bc; ++b) copy_in<T>(ctx, ws.y + (size_t)b * ldm, Yc + (ptrdiff_t)b * y_stride, incy, m);
            }
            HIPCHK(hipMemsetAsync(ws.x, 0, (size_t)Bc * np * sizeof(T), st));
            HIPCHK(hipMemsetAsync(ws.d, 0, (size_t)Bc * np * sizeof(T), st));
            HIPCHK(hipMemsetAsync(ws.insup, 0, (size_t)Bc * np, st));
            HIPCHK(hipMemsetAsync(ws.st, 0, (size_t)Bc * sizeof(DevState), st));
            HIPCHK(hipMemsetAsync(ws.ndone, 0, sizeof(uint32_t), st));
            HIPCHK(hipMemsetAsync(ws.rhs, 0, 2 * bp * ldm * sizeof(T), st));
            HIPCHK(hipMemcpyAsync(Rblk, ws.y, (size_t)Bc * ldm * sizeof(T), hipMemcpyDeviceToDevice, st));

            // c_b = A^T y_b for every signal (residual_vector with x = 0, homotopy-cpu.cpp:215)
            uint32_t nparts = 0;
            const bool time_c0 = ctx->profiling != 0;
            if (time_c0 && !ctx->ev_c0a) { HIPCHK(hipEventCreate(&ctx->ev_c0a)); HIPCHK(hipEventCreate(&ctx->ev_c0b)); }
            if (time_c0) HIPCHK(hipEventRecord(ctx->ev_c0a, st));
            HIPCHK(launch_gemm_tn_f32(ctx, Rblk, rows, (uint32_t)ldm, ws.c, (uint32_t)np, nullptr));
            if (time_c0) HIPCHK(hipEventRecord(ctx->ev_c0b, st));
            HIPCHK(launch_absmax<T>(ctx, ws, Bc, &nparts));
            HIPCHK(launch_init<T>(ctx, ws, Bc, nparts, tol));
            bool gram_chunk = gram;
            if (gram_chunk && ctx->engine == 1) {
                // the tolerance guard of engine 1, for every signal of the chunk at once
                ctx->host_flags[4] = 0;
                HIPCHK(launch_gram_guard_batched<T>(ctx, ws, Bc, tol));
                HIPCHK(hipStreamSynchronize(st));
                if (ctx->host_flags[4] != 0) { gram_chunk = false; ctx->stats.gram_fallbacks += 1; }
            }
            if (gram_chunk) {
                // keep c0 = A^T y of every signal: the Gram-form rounds subtract from it
                if (ctx->c0_batch_rows < bp) {
                    if (ctx->c0_batch) HIPCHK(hipFree(ctx->c0_batch));
                    ctx->c0_batch = nullptr;
                    ctx->c0_batch_rows = 0;
                    HIPCHK(hipMalloc(&ctx->c0_batch, bp * np * sizeof(T)));
                    ctx->c0_batch_rows = bp;
                }
                HIPCHK(hipMemcpyAsync(ctx->c0_batch, ws.c, (size_t)Bc * np * sizeof(T), hipMemcpyDeviceToDevice, st));
            } else {
                HIPCHK(launch_rp<T>(ctx, ws, Bc));
            }
            BatchCols bc;
            const bool cols_chunk = gram_chunk && cols_form;
            if (cols_chunk) {
                // one cache row per slot and round (round 0 = the first pick), a row table per slot, the pass lists:
                // sized by the dispatcher (ensure_bcol), which sends the batch another way when they do not fit
                const size_t rows_needed = ((size_t)max_iter + 2) * Bc;
                const size_t pitchc = (np + 1023) / 1024 * 1024;          // (k_la_cq reads whole 1024-column chunks of a row)
                if (ctx->bcol_cache_rows < rows_needed || ctx->bcol_slot_rows < Bc || !ctx->bcol_lists)
                    throw HipFail{ hipErrorOutOfMemory, "column form: cache not sized for this chunk" };
                bc.pitch = (uint32_t)pitchc;
                HIPCHK(hipMemsetAsync(ctx->bcol_slot, 0xff, (size_t)Bc * np * sizeof(int32_t), st));
                bc.cache = ctx->bcol_cache;
                bc.bslot = ctx->bcol_slot;
                bc.rcols = ctx->bcol_lists;
                bc.drows = ctx->bcol_lists + 1024;
                bc.cap = (Bc + 63u) / 64u * 64u;
                bc.row_base = 0;
                HIPCHK(launch_batch_cols(ctx, ws.st, Bc, 0, true, bc.bslot, bc.rcols, bc.drows, bc.cap));
                HIPCHK(launch_batch_passes(ctx, &bc, Bc));
                ctx->stats.batch_col_rounds += 1;
            }
            const T* const Gsrc = cols_chunk ? bc.cache : ctx->gram_full;
            const uint32_t Gpitch = cols_chunk ? bc.pitch : ctx->gram_pitch;

            // Subset form (subbatch.hip): with G at hand every signal is solved by one workgroup on the 448 columns with the
            // largest |c0| and then checked against all columns — 16.8 MB of G per signal instead of 545
            bool sub_chunk = gram_chunk && form == 1 && !no_subset && ctx->batch_subset && ctx->gram_full != nullptr && sub_form_usable(ctx);
            if (sub_chunk && ctx->sub_off_chunks > 0) { ctx->sub_off_chunks -= 1; sub_chunk = false; }     // (it handed back too much lately)
            const bool scr_chunk = gram_chunk && scr_form && !no_subset;
            // (a chunk whose tolerance is too tight for Gram-form correlations runs the residual-form rounds below, as in every form)
            if (sub_chunk || scr_chunk) {
                const size_t need = sub_buffer_bytes(Bc);
                if (ctx->sub_buf_bytes < need) {
                    if (ctx->sub_buf) HIPCHK(hipFree(ctx->sub_buf));
                    ctx->sub_buf = nullptr;
                    ctx->sub_buf_bytes = 0;
                    if (hipMalloc(&ctx->sub_buf, need) != hipSuccess) {
                        (void)hipGetLastError(); ctx->sub_buf = nullptr; sub_chunk = false;
                        if (scr_chunk) throw HipFail{ hipErrorOutOfMemory, "screened batch form: log buffers" };
                    } else ctx->sub_buf_bytes = need;
                }
            }
            const uint32_t L = (uint32_t)std::max(1, std::min(ctx->lookahead, 64));
            volatile uint32_t* hf = ctx->host_flags;
            const uint64_t last_round = (uint64_t)max_iter + 1;
            uint64_t rounds_run = 0;
            size_t ncq = 0;                                  // timed k_la_cq launches of this chunk (profiling on)
            if (sub_chunk) {
                const bool timed = ctx->profiling != 0;
                hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
                if (timed) { e0 = prof_event(ctx, 0); e1 = prof_event(ctx, 1); e2 = prof_event(ctx, 2); if (!ctx->ev_sub_sel) HIPCHK(hipEventCreate(&ctx->ev_sub_sel)); }
                HIPCHK(launch_sub_form(ctx, ws, Bc, ctx->c0_batch, tol, max_iter, e0, e1, e2));
                if (timed) {
                    HIPCHK(hipEventSynchronize(e2));
                    float ms = 0.f;
                    HIPCHK(hipEventElapsedTime(&ms, e0, e1)); ctx->stats.sub_solve_ms += ms;
                    if (std::getenv("SS_HIP_SUB_DEBUG")) { float ms2 = 0.f; HIPCHK(hipEventElapsedTime(&ms2, e0, ctx->ev_sub_sel)); std::fprintf(stderr, "[subset form] select %.3f ms, solve %.3f ms\n", ms2, ms - ms2); }
                    HIPCHK(hipEventElapsedTime(&ms, e1, e2)); ctx->stats.sub_verify_ms += ms;
                    HIPCHK(hipEventElapsedTime(&ms, ctx->ev_c0a, ctx->ev_c0b));
                    ctx->stats.c0_gemm_ms += ms;
                    ctx->stats.c0_gemm_flops += 2.0 * (double)rows * (double)ldm * (double)np;
                }
            }
            if (scr_chunk) HIPCHK(launch_screen_batch(ctx, ws, Bc, ctx->c0_batch, tol, max_iter));
            for (uint64_t round = 1; round <= last_round && !sub_chunk && !scr_chunk; ++round) {
                if (round > L) {
                    const uint32_t need = (uint32_t)(round - L);
                    uint32_t spins = 0;
                    while (hf[1] == 0 && hf[0] < need) {
                        if ((++spins & 0x3ffu) == 0) {
                            const hipError_t q = hipStreamQuery(st);
                            if (q == hipSuccess) break;
                            if (q != hipErrorNotReady) throw HipFail{ q, "hipStreamQuery(batch loop)" };
                        }
                        std::this_thread::yield();
                    }
                    if (hf[1] != 0) break;
                }
                if (gram_chunk) {
                    const bool timed_cq = ctx->profiling != 0 && ncq < 4096;
                    if (timed_cq) HIPCHK(hipEventRecord(prof_event(ctx, 2 * ncq), st));
                    // (fused form: the scan and the pick of this round happen inside the Gram-form pass, k_la_cqs)
                    const bool fused = cqs_usable(ctx, ws.dims.pmin_stride);
                    HIPCHK(launch_cq_gram_batched<T>(ctx, ws, Bc, Gsrc, Gpitch, ctx->c0_batch, &nparts, cols_chunk ? bc.bslot : nullptr,
                                                     fused ? (uint32_t)round : 0u, tol, max_iter));
                    if (timed_cq) { HIPCHK(hipEventRecord(prof_event(ctx, 2 * ncq + 1), st)); ++ncq; }
                    bc.row_base = (uint32_t)(round * Bc);
                    HIPCHK(launch_tail_gram_batched<T>(ctx, ws, Bc, (uint32_t)round, nparts, tol, max_iter,
                                                       Gsrc, Gpitch, cols_chunk ? &bc : nullptr, fused));
                    if (cols_chunk) ctx->stats.batch_col_rounds += 1;
                } else {
                    HIPCHK(launch_tile_list(ctx, ws.st, Bc, rows, ws.tile_skip));
                    HIPCHK(launch_gemm_tn_f32(ctx, Rblk, rows, (uint32_t)ldm, ws.c, (uint32_t)np, ws.tile_skip));
                    HIPCHK(launch_gemm_tn_f32(ctx, Pblk, rows, (uint32_t)ldm, ws.q, (uint32_t)np, ws.tile_skip));
                    HIPCHK(launch_absmax<T>(ctx, ws, Bc, &nparts));
                    HIPCHK(launch_iteration_tail<T>(ctx, ws, Bc, (uint32_t)round, nparts, tol, max_iter));
                }
                ++rounds_run;
            }
            hs.resize(Bc);
            HIPCHK(hipMemcpyAsync(hs.data(), ws.st, (size_t)Bc * sizeof(DevState), hipMemcpyDeviceToHost, st));
            if (rec_out) {
                const size_t rb = record_bytes(kmax, sizeof(T));
                const unsigned char* stage = pack_records<T>(ctx, ws, Bc, kmax);
                HIPCHK(hipMemcpyAsync(static_cast<unsigned char*>(rec_out) + b0 * rb, stage, rb * Bc, hipMemcpyDefault, st));
            }
            float* Xc = X ? X + (ptrdiff_t)b0 * x_stride : nullptr;
            if (!X) {
            } else if (incx == 1) {
                HIPCHK(hipMemcpy2DAsync(Xc, (size_t)x_stride * sizeof(T), ws.x, np * sizeof(T), n * sizeof(T), Bc,
                                        hipMemcpyDefault, st));
            } else {
                for (uint32_t b = 0; b < Bc; ++b) copy_out<T>(ctx, Xc + (ptrdiff_t)b * x_stride, incx, ws.x + (size_t)b * np, n);
            }
            HIPCHK(hipStreamSynchronize(st));
            if (want_trace != 0u && ws.trace) {
                const size_t cnt = std::min<size_t>((size_t)hs[0].iter + 1, ws.trace_cap);
                ctx->last_trace.resize(cnt);
                HIPCHK(hipMemcpy(ctx->last_trace.data(), ws.trace, cnt * sizeof(TraceEntry), hipMemcpyDeviceToHost));
            }
            // The fused scan's meeting of a signal's workgroups is a bounded wait (k_la_cqs): should it ever expire — the
            // workgroups of a signal not resident together — the slot carries SS_HIP_ERUNTIME; this context then goes on with
            // the two-kernel form and the chunk is solved again (nothing of it has been reported yet)
            if (gram && ctx->batch_fused_scan) {
                bool expired = false;
                for (uint32_t b = 0; b < Bc; ++b) expired = expired || hs[b].status == SS_HIP_ERUNTIME;
                if (expired) {
                    ctx->batch_fused_scan = 0;
                    ctx->stats.persist_fallbacks += 1;
                    b0 -= chunk;                                 // (the loop adds it back: unsigned arithmetic)
                    continue;
                }
            }
            std::vector<uint32_t> ties;                      // slots whose scan met a tie stall (DevState::tie_stall)
            uint32_t n_redo_chunk = 0;
            for (uint32_t b = 0; b < Bc; ++b) {
                if (!hs[b].done) { set_err(err, errlen, "solve_batch: internal error, a signal did not terminate"); return SS_HIP_ERUNTIME; }
                if (hs[b].status == kStatusSubsetDecline || hs[b].status == kStatusSubsetFail) {
                    if (std::getenv("SS_HIP_SUB_DEBUG"))
                        std::fprintf(stderr, "[subset form] signal %zu: status %u after %u iterations, %u breakpoints logged, K = %u, lambda %g\n",
                                     b0 + b, hs[b].status, hs[b].iter, hs[b].solo_nlog, hs[b].K, hs[b].c_inf);
                    redo.push_back(b0 + b);                  // (nothing of it is reported: solved again in the lock-step form below)
                    ++n_redo_chunk;
                    continue;
                }
                if (ctx->tie_rerun && !ctx->tie_guard && (hs[b].status == kStatusTieRerun || (hs[b].status == 0 && hs[b].tie_stall != 0))) {
                    ties.push_back(b);
                    continue;
                }
                if (hs[b].status != 0) {
                    set_err(err, errlen, hs[b].status == SS_HIP_ECAPACITY ? "solve_batch: active set outgrew the workspace capacity"
                                                                          : "solve_batch: internal error, a device-side wait expired");
                    return (int)hs[b].status;
                }
                if (iter_out) iter_out[b0 + b] = hs[b].iter;
                if (err_out) err_out[b0 + b] = hs[b].c_inf;
                ctx->stats.iterations += hs[b].iter;
            }
            ctx->stats.solves += Bc - (uint32_t)ties.size() - n_redo_chunk;
            if (sub_chunk) {
                ctx->stats.subset_signals += Bc - n_redo_chunk;
                ctx->stats.subset_redone += n_redo_chunk;
                if (Bc >= 16 && 3u * n_redo_chunk > Bc) ctx->sub_off_chunks = 8;
            }
            if (scr_chunk) {
                ctx->stats.screen_signals += Bc - n_redo_chunk - (uint32_t)ties.size();
                ctx->stats.screen_redone += n_redo_chunk;
                // (a context whose signals the form mostly hands back — dense supports — goes the other way for the next 8 batches)
                if (Bc >= 8 && 3u * n_redo_chunk > Bc) ctx->sub_off_chunks = 8;
            }
            ctx->stats.batch_rounds += rounds_run;
            if (ncq != 0) {
                // rounds enqueued behind the end of the batch are no-ops (microseconds): only launches that
                // belong to a round some signal was still running in are counted, with their bytes
                uint32_t live_rounds = 0;
                for (uint32_t b = 0; b < Bc; ++b) live_rounds = std::max(live_rounds, std::min<uint32_t>(hs[b].iter + 1u, max_iter));
                for (size_t i = 0; i < ncq && i < live_rounds; ++i) {
                    float ms = 0.f;
                    HIPCHK(hipEventElapsedTime(&ms, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]));
                    ctx->stats.cq_ms += ms;
                    ctx->stats.cq_launches += 1;
                }
                // a signal with `it` iterations is live in rounds 1 .. it + 1 (the last one finds the end); in
                // round r its list holds r columns (one per iteration; removals make this an upper bound)
                uint64_t rows = 0;
                for (uint32_t b = 0; b < Bc; ++b) {
                    const uint64_t r = std::min<uint64_t>(std::min<uint32_t>(hs[b].iter + 1u, max_iter), ncq);
                    rows += r * (r + 1) / 2 + 3 * r;
                }
                ctx->stats.cq_bytes += rows * (uint64_t)n * sizeof(T);
            }
            // Tie stalls: which implementation's rounding derails on an exact tie is luck (homotopy-cpu.cpp:143-153), so
            // these signals are solved again in the reference-order engine — after the last chunk, all of them together
            // (up to 4 share every pass over A: solve_batch_ro).  (A handful per 4096 signals at 8192 x 65536.)
            for (uint32_t b : ties) all_ties.push_back(b0 + b);
        }
        if (!redo.empty() && scr_form) {
            // screened batch form: what it hands back is solved by the default single-signal engine, one by one
            const size_t rb = record_bytes(kmax, sizeof(T));
            for (size_t g : redo) {
                uint32_t it = 0;
                double e = 0.0;
                Route r; r.no_sub = true;                  // (the batch's screened form has declined it: the default engine)
                const int rc = solve_impl<T>(ctx, Y + (ptrdiff_t)g * y_stride, incy, tol, max_iter, X ? X + (ptrdiff_t)g * x_stride : nullptr, incx, &it, &e,
                                             err, errlen, r, rec_out ? static_cast<unsigned char*>(rec_out) + g * rb : nullptr, kmax);
                if (rc != SS_HIP_OK) return rc;
                if (iter_out) iter_out[g] = it;
                if (err_out) err_out[g] = e;
            }
            redo.clear();
        }
        if (!redo.empty()) {
            // the signals the subset form did not vouch for, gathered and solved in the lock-step Gram form (their own ties
            // are arbitrated inside that call), results scattered back
            const size_t nr = redo.size(), rb = record_bytes(kmax, sizeof(T));
            T* Yg = nullptr; T* Xg = nullptr; unsigned char* Rg = nullptr;
            struct FreeTmp { T*& a; T*& b; unsigned char*& c; ~FreeTmp() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (c) (void)hipFree(c); } } free_tmp{ Yg, Xg, Rg };
            HIPCHK(hipMalloc(&Yg, nr * m * sizeof(T)));
            if (X) HIPCHK(hipMalloc(&Xg, nr * n * sizeof(T)));
            if (rec_out) HIPCHK(hipMalloc(&Rg, nr * rb));
            for (size_t r = 0; r < nr; ++r) copy_in<T>(ctx, Yg + r * m, Y + (ptrdiff_t)redo[r] * y_stride, incy, m);
            HIPCHK(hipStreamSynchronize(st));
            std::vector<uint32_t> it_r(nr, 0u);
            std::vector<double> er_r(nr, 0.0);
            const int rc = solve_batch_gemm_f32(ctx, Yg, nr, (ptrdiff_t)m, 1, tol, max_iter, Xg, (ptrdiff_t)n, 1, it_r.data(), er_r.data(), err, errlen,
                                                form, Rg, kmax, true);
            if (rc != SS_HIP_OK) return rc;
            for (size_t r = 0; r < nr; ++r) {
                const size_t g = redo[r];
                if (X) copy_out<T>(ctx, X + (ptrdiff_t)g * x_stride, incx, Xg + r * n, n);
                if (rec_out) HIPCHK(hipMemcpyAsync(static_cast<unsigned char*>(rec_out) + g * rb, Rg + r * rb, rb, hipMemcpyDefault, st));
                if (iter_out) iter_out[g] = it_r[r];
                if (err_out) err_out[g] = er_r[r];
            }
            HIPCHK(hipStreamSynchronize(st));
        }
        if (!all_ties.empty()) {
            ctx->stats.tie_reruns += all_ties.size();
            const int rc = solve_batch_ro<T>(ctx, Y, all_ties.data(), all_ties.size(), y_stride, incy, tol, max_iter, X, x_stride, incx,
                                             iter_out, err_out, err, errlen, rec_out, kmax);
            if (rc != SS_HIP_OK) return rc;
        }
    } catch (const HipFail& f) {
        set_err(err, errlen, hip_msg(f));
        return SS_HIP_ERUNTIME;
    } catch (const std::bad_alloc&) {
        set_err(err, errlen, "solve_batch: out of host memory");
        return SS_HIP_ENOMEM;
    }
    return SS_HIP_OK;
}

template <typename T>
int solve_batch_seq(ss_hip_ctx* ctx, const T* Y, size_t B, ptrdiff_t y_stride, ptrdiff_t incy, T tol,
                    uint32_t max_iter, T* X, ptrdiff_t x_stride, ptrdiff_t incx, uint32_t* iter_out,
                    double* err_out, char* err, size_t errlen, void* rec_out = nullptr, uint32_t kmax = 0)
{
    const size_t rb = record_bytes(kmax, sizeof(T));
    for (size_t b = 0; b < B; ++b) {
        uint32_t it = 0;
        double e = 0.0;
        const int rc = solve_impl<T>(ctx, Y + (ptrdiff_t)b * y_stride, incy, tol, max_iter,
                                     X ? X + (ptrdiff_t)b * x_stride : nullptr, incx, &it, &e, err, errlen, Route(),
                                     rec_out ? static_cast<unsigned char*>(rec_out) + b * rb : nullptr, kmax);
        if (rc != SS_HIP_OK) return rc;
        if (iter_out) iter_out[b] = it;
        if (err_out) err_out[b] = e;
    }
    return SS_HIP_OK;
}

int solve_batch_dispatch(ss_hip_ctx* ctx, const float* Y, size_t B, ptrdiff_t y_stride, ptrdiff_t incy, float tol,
                         uint32_t max_iter, float* X, ptrdiff_t x_stride, ptrdiff_t incx, uint32_t* iter_out,
                         double* err_out, char* err, size_t errlen, void* rec_out = nullptr, uint32_t kmax = 0)
{
    // reference-order engine: in lock-step, up to 4 signals per pass over A
    if (ctx->engine == 3)
        return solve_batch_ro<float>(ctx, Y, nullptr, B, y_stride, incy, tol, max_iter, X, x_stride, incx, iter_out, err_out, err, errlen, rec_out, kmax);
    // lock-step MFMA path once enough signals share the matrix (batch_min option, default 192) — or, with G = A^T A already
    // in HBM, from four signals on: the subset form then runs every signal on a workgroup of its own (subbatch.hip)
    const bool have_g = ctx->gram_full != nullptr && ctx->batch_subset && ctx->engine >= 1 && ctx->batch_gram_min > 0;
    gram_reserve_start(ctx, B);
    // (from four: the batch GEMM that forms c0 works on 128 rows at a time — 1.1 ms at 8192 x 65536, three single sweeps)
    const bool lockstep = B >= (size_t)std::max(2, ctx->batch_min) || (have_g && B >= 4);
    int form = 0;
    // Gram form when G = A^T A is at hand, or the batch is large enough to pay for making it
    // (2 m n^2 flops once, against 4 m n flops per signal and round); same tolerance guard as engine 1
    // (where the screened batch form applies, G = A^T A — 0.28 s at 8192 x 65536 — pays later: a one-off batch breaks even at
    // ~1700 signals (5 300 signals/s against 31 000 + the build), a context that keeps receiving batches after ~3000 in all)
    const bool scr_batches = !ctx->gram_full && ctx->batch_screen && ctx->engine >= 1 && ctx->la_fused >= 3 && ctx->early_solo &&
                             ctx->solo_subset == 256 && !ctx->tracing && ctx->sub_off_chunks == 0 && screen_form_usable(ctx);
    const size_t gram_min = scr_batches ? std::max<size_t>((size_t)ctx->batch_gram_min, 1536) : (size_t)ctx->batch_gram_min;
    const bool gram_pays = B >= gram_min || (scr_batches && ctx->batch_signals_seen + B >= 3072 && B >= (size_t)ctx->batch_gram_min);
    ctx->batch_signals_seen += B;
    if (lockstep && ctx->engine >= 1 && ctx->batch_gram_min > 0 && (ctx->gram_full || gram_pays)) {
        try {
            HIPCHK(hipSetDevice(ctx->device));
            if (ensure_full_gram(ctx)) form = 1;
        } catch (const HipFail& f) {
            set_err(err, errlen, hip_msg(f));
            return SS_HIP_ERUNTIME;
        }
    }
    // screened form for the batches in between (4 .. batch_gram_min - 1 signals, no G; option batch_screen): c0 of a chunk by the
    // batch GEMM, every signal solved by one workgroup on its subset's own Gram matrix, one screening launch per chunk of 64
    if (form == 0 && !ctx->gram_full && ctx->batch_screen && ctx->engine >= 1 && B >= 4 && ctx->la_fused >= 3 && ctx->early_solo &&
        ctx->solo_subset == 256 && !ctx->tracing && screen_form_usable(ctx)) {
        if (ctx->sub_off_chunks > 0) ctx->sub_off_chunks -= 1;      // (it handed back too much lately)
        else form = 3;
    }
    // column form for the batches in between (batch_cols_min .. batch_cols_max signals, no G): in lock-step, one pass
    // over A per round and 64 signals forms the Gram columns of the entering columns (half the flops of the two GEMMs
    // of form 0, and one pass serves 64 signals where a single solve spends three on one).  The cache holds one row
    // per slot and round; a budget (option gram_full_gib) it does not fit sends the batch the old way.
    if (form == 0 && !ctx->gram_full && ctx->engine >= 1 && ctx->batch_cols_min > 0 && B >= (size_t)std::max(2, ctx->batch_cols_min) &&
        (ctx->batch_cols_max <= 0 || B <= (size_t)ctx->batch_cols_max) && ctx->n_pad % 256 == 0) {
        // signals per chunk: at most 448 (7 full passes per round), at most what the cache budget AND the free HBM hold
        // (the cache is (max_iter + 2) rows per signal: a large max_iter with many signals does not fit — such a batch
        // goes the old way instead of failing), whole passes
        const double row_bytes = ((double)max_iter + 2.0) * (double)((ctx->n_pad + 1023) / 1024 * 1024) * 4.0;
        double budget = (double)ctx->gram_full_gib * 1073741824.0;
        size_t free_b = 0, total_b = 0;
        if (hipSetDevice(ctx->device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double held = (double)ctx->bcol_cache_rows * (double)((ctx->n_pad + 1023) / 1024 * 1024) * 4.0;
            budget = std::min(budget, std::max(0.0, (double)free_b + held - 4.0 * 1073741824.0));
        } else (void)hipGetLastError();
        const double fit = budget / row_bytes;
        size_t per = (size_t)std::min<double>(448.0, std::max(0.0, fit));
        if (per >= B) per = B; else per = per / 64 * 64;
        if ((per >= 64 || (per == B && per > 0)) && ensure_bcol(ctx, per, max_iter)) { form = 2; ctx->bcol_chunk = (int)std::max<size_t>(per, 1); }
    }
    if (lockstep || form == 2 || form == 3)
        return solve_batch_gemm_f32(ctx, Y, B, y_stride, incy, tol, max_iter, X, x_stride, incx, iter_out, err_out, err, errlen, form,
                                    rec_out, kmax);
    return solve_batch_seq<float>(ctx, Y, B, y_stride, incy, tol, max_iter, X, x_stride, incx, iter_out, err_out, err, errlen, rec_out, kmax);
}

// fp64 batch, resident tier, in chunks of screen64_batch_cap() signals
int solve_batch_res64(ss_hip_ctx* ctx, const double* Y, size_t B, ptrdiff_t y_stride, ptrdiff_t incy, double tol, uint32_t max_iter, double* X,
                      ptrdiff_t x_stride, ptrdiff_t incx, uint32_t* iter_out, double* err_out, char* err, size_t errlen, void* rec_out, uint32_t kmax)
{
    if (max_iter == 0) { set_err(err, errlen, "solve_batch: max_iterations must be > 0"); return SS_HIP_EINVAL; }
    if (!(tol >= std::numeric_limits<double>::epsilon() && tol < 1.0)) { set_err(err, errlen, "solve_batch: tolerance must satisfy eps <= tolerance < 1"); return SS_HIP_EINVAL; }
    if (incy <= 0 || incx <= 0) { set_err(err, errlen, "solve_batch: vector increments must be positive"); return SS_HIP_EINVAL; }
    const size_t rb = record_bytes(kmax, sizeof(double));
    const uint32_t cap = screen64_batch_cap();
    std::vector<size_t> redo;
    try {
        HIPCHK(hipSetDevice(ctx->device));
        const size_t m = ctx->m, n = ctx->n;
        const uint32_t kcap = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(n, (uint64_t)max_iter + 1), kKcapLimit);
        std::vector<DevState> hst(cap);
        for (size_t b0 = 0; b0 < B; b0 += cap) {
            const uint32_t nb = (uint32_t)std::min<size_t>(cap, B - b0);
            ensure_workspace<double>(ctx, cap, kcap);
            Workspace<double>& ws = *ws_of<double>(ctx);
            hipStream_t st = ctx->stream;
            HIPCHK(hipMemsetAsync(ws.y, 0, (size_t)cap * ctx->ldm * sizeof(double), st));
            for (uint32_t b = 0; b < nb; ++b) copy_in<double>(ctx, ws.y + (size_t)b * ctx->ldm, Y + (ptrdiff_t)(b0 + b) * y_stride, incy, m);
            HIPCHK(hipMemsetAsync(ws.x, 0, (size_t)nb * ctx->n_pad * sizeof(double), st));
            HIPCHK(hipMemsetAsync(ws.st, 0, (size_t)nb * sizeof(DevState), st));
            HIPCHK(launch_screen64_batch(ctx, ws, nb, tol, max_iter));
            const unsigned char* stage = rec_out ? pack_records<double>(ctx, ws, nb, kmax) : nullptr;
            HIPCHK(hipMemcpyAsync(hst.data(), ws.st, (size_t)nb * sizeof(DevState), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (uint32_t b = 0; b < nb; ++b) {
                const DevState& hs = hst[b];
                const bool good = hs.done && hs.status == 0 && hs.tie_stall == 0;
                if (!good) {
                    redo.push_back(b0 + b);
                    ctx->stats.screen_tier2 += 1;
                    count_reasons(ctx, hs.sub_reason, hs.tie_stall != 0 || hs.status == kStatusTieRerun);
                    continue;
                }
                if (X) copy_out<double>(ctx, X + (ptrdiff_t)(b0 + b) * x_stride, incx, ws.x + (size_t)b * ctx->n_pad, n);
                if (rec_out) HIPCHK(hipMemcpyAsync(static_cast<unsigned char*>(rec_out) + (b0 + b) * rb, stage + (size_t)b * rb, rb, hipMemcpyDefault, st));
                if (iter_out) iter_out[b0 + b] = hs.iter;
                if (err_out) err_out[b0 + b] = hs.c_inf;
                ctx->stats.solves += 1;
                ctx->stats.iterations += hs.iter;
                ctx->stats.screen_signals += 1;
                ctx->stats.screen_resident += 1;
            }
            HIPCHK(hipStreamSynchronize(st));
        }
    } catch (const HipFail& f) {
        set_err(err, errlen, hip_msg(f));
        return SS_HIP_ERUNTIME;
    } catch (const std::bad_alloc&) {
        set_err(err, errlen, "solve_batch: out of host memory");
        return SS_HIP_ENOMEM;
    }
    // what the chunks did not report: alone, through the sub-dictionary tier and the engine behind it
    for (size_t b : redo) {
        uint32_t it = 0;
        double e = 0.0;
        Route r; r.no_res = true;                          // (the batch's resident tier has declined it: the tiers behind)
        const int rc = solve_impl<double>(ctx, Y + (ptrdiff_t)b * y_stride, incy, tol, max_iter, X ? X + (ptrdiff_t)b * x_stride : nullptr, incx, &it, &e, err, errlen,
                                          r, rec_out ? static_cast<unsigned char*>(rec_out) + b * rb : nullptr, kmax);
        if (rc != SS_HIP_OK) return rc;
        if (iter_out) iter_out[b] = it;
        if (err_out) err_out[b] = e;
    }
    return SS_HIP_OK;
}

int solve_batch_dispatch(ss_hip_ctx* ctx, const double* Y, size_t B, ptrdiff_t y_stride, ptrdiff_t incy, double tol,
                         uint32_t max_iter, double* X, ptrdiff_t x_stride, ptrdiff_t incx, uint32_t* iter_out,
                         double* err_out, char* err, size_t errlen, void* rec_out = nullptr, uint32_t kmax = 0)
{
    // reference-order engine: in lock-step, up to 4 signals per pass over A
    if (ctx->engine == 3)
        return solve_batch_ro<double>(ctx, Y, nullptr, B, y_stride, incy, tol, max_iter, X, x_stride, incx, iter_out, err_out, err, errlen, rec_out, kmax);
    // fp64 batches of four signals or more on dictionaries the fp64 screened form takes: the resident tier with the signals of a chunk
    // side by side (screen.hip: launch_screen64_batch) — one pass over the fp16 copy ranks every signal's columns, the paths run in as
    // many workgroups at once, every signal's states are certified by a screening pass of its own; what a slot's certificate does not
    // cover is solved again alone, through the remaining tiers
    if (B >= 4 && !ctx->tracing && ctx->engine >= 1 && ctx->la_fused >= 1 && ctx->screen_resident && ctx->sub_off_solves == 0 && ctx->res_off_solves == 0) {
        bool usable = false;
        try { HIPCHK(hipSetDevice(ctx->device)); usable = screen64_batch_usable(ctx); } catch (const HipFail&) { usable = false; }
        if (usable) return solve_batch_res64(ctx, Y, B, y_stride, incy, tol, max_iter, X, x_stride, incx, iter_out, err_out, err, errlen, rec_out, kmax);
    }
    // otherwise one signal at a time
    return solve_batch_seq<double>(ctx, Y, B, y_stride, incy, tol, max_iter, X, x_stride, incx, iter_out, err_out, err, errlen, rec_out, kmax);
}

template <typename T>
int solve_batch_impl(ss_hip_ctx* ctx, const T* Y, size_t B, ptrdiff_t y_stride, ptrdiff_t incy, T tol,
                     uint32_t max_iter, T* X, ptrdiff_t x_stride, ptrdiff_t incx, uint32_t* iter_out,
                     double* err_out, char* err, size_t errlen, void* rec_out = nullptr, uint32_t kmax = 0)
{
    if (!ctx) { set_err(err, errlen, "solve_batch: null context"); return SS_HIP_EINVAL; }
    if (ctx->kind != 0) { set_err(err, errlen, "solve_batch: this context was created for IRLS"); return SS_HIP_EINVAL; }
    if (ctx->is_f64 != (sizeof(T) == 8)) { set_err(err, errlen, "solve_batch: element type mismatch"); return SS_HIP_ETYPE; }
    if (!Y || (!X && !rec_out)) { set_err(err, errlen, "solve_batch: Y and the output must not be null"); return SS_HIP_EINVAL; }
    if (rec_out && (kmax == 0 || kmax > kKcapLimit || (reinterpret_cast<uintptr_t>(rec_out) & 7u))) {
        set_err(err, errlen, "solve_batch_compact: kmax must be 1..4096 and records 8-byte aligned");
        return SS_HIP_EINVAL;
    }
    if (B == 0) return SS_HIP_OK;
    return solve_batch_dispatch(ctx, Y, B, y_stride, incy, tol, max_iter, X, x_stride, incx, iter_out, err_out, err, errlen, rec_out, kmax);
}

template <typename T>
int gemv_t_impl(ss_hip_ctx* ctx, const T* r, T* c, int repeats, float* ms_out, char* err, size_t errlen)
{
    if (ctx && ctx->kind != 0) { set_err(err, errlen, "this entry point needs a Homotopy context (an IRLS context holds the factorised matrix)"); return SS_HIP_EINVAL; }
    if (!ctx || !r || !c) { set_err(err, errlen, "gemv_t: null argument"); return SS_HIP_EINVAL; }
    if (ctx->is_f64 != (sizeof(T) == 8)) { set_err(err, errlen, "gemv_t: type mismatch"); return SS_HIP_ETYPE; }
    if (repeats < 1) repeats = 1;
    try {
        HIPCHK(hipSetDevice(ctx->device));
        Workspace<T>& ws = *ws_of<T>(ctx);
        hipStream_t st = ctx->stream;
        copy_in<T>(ctx, ws.rhs, r, 1, ctx->m);
        if (ctx->ldm > ctx->m)
            HIPCHK(hipMemsetAsync(ws.rhs + ctx->m, 0, (ctx->ldm - ctx->m) * sizeof(T), st));
        uint32_t nb = 0;
        HIPCHK(hipEventRecord(ctx->ev_solve0, st));
        for (int i = 0; i < repeats; ++i) {
            // (option engine = 3: the reference-order sweep, reforder.hip)
            if (ctx->engine == 3) HIPCHK(launch_ro_sweep<T>(ctx, ws.rhs, 0, ws.c, 0, ws.dims.n_pad, 1, 1u, ws.pmax_val, ws.pmax_idx, ws.dims.pmax_stride, &nb, nullptr, false));
            else HIPCHK(launch_sweep<T>(ctx, ws.rhs, 0, 1, ws.c, nullptr, ws.pmax_val, ws.pmax_idx, &nb, nullptr));
        }
        HIPCHK(hipEventRecord(ctx->ev_solve1, st));
        copy_out<T>(ctx, c, 1, ws.c, ctx->n);
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ctx->ev_solve0, ctx->ev_solve1));
        if (ms_out) *ms_out = ms / (float)repeats;
    } catch (const HipFail& f) {
        set_err(err, errlen, hip_msg(f));
        return SS_HIP_ERUNTIME;
    }
    return SS_HIP_OK;
}

// C[b][:] = A^T R[b][:] for B right-hand sides through the MFMA GEMM (fp32 contexts)
int gemm_t_impl(ss_hip_ctx* ctx, const float* R, size_t B, ptrdiff_t ldR, float* C, ptrdiff_t ldC,
                int repeats, float* ms_out, char* err, size_t errlen)
{
    if (ctx && ctx->kind != 0) { set_err(err, errlen, "this entry point needs a Homotopy context (an IRLS context holds the factorised matrix)"); return SS_HIP_EINVAL; }
    if (!ctx || !R || !C || B == 0) { set_err(err, errlen, "gemm_t: null/empty argument"); return SS_HIP_EINVAL; }
    if (ctx->is_f64) { set_err(err, errlen, "gemm_t: fp32 contexts only"); return SS_HIP_ETYPE; }
    if (repeats < 1) repeats = 1;
    float* Rd = nullptr;
    float* Dd = nullptr;
    int rc = SS_HIP_OK;
    try {
        HIPCHK(hipSetDevice(ctx->device));
        const size_t Bp = (B + 127) / 128 * 128;
        const size_t ldm = ctx->ldm, np = ctx->n_pad;
        HIPCHK(hipMalloc(&Rd, Bp * ldm * sizeof(float)));
        HIPCHK(hipMalloc(&Dd, Bp * np * sizeof(float)));
        HIPCHK(hipMemsetAsync(Rd, 0, Bp * ldm * sizeof(float), ctx->stream));
        HIPCHK(hipMemcpy2DAsync(Rd, ldm * sizeof(float), R, (size_t)ldR * sizeof(float),
                                ctx->m * sizeof(float), B, hipMemcpyDefault, ctx->stream));
        HIPCHK(hipEventRecord(ctx->ev_solve0, ctx->stream));
        for (int i = 0; i < repeats; ++i)
            HIPCHK(launch_gemm_tn_f32(ctx, Rd, (uint32_t)Bp, (uint32_t)ldm, Dd, (uint32_t)np, nullptr));
        HIPCHK(hipEventRecord(ctx->ev_solve1, ctx->stream));
        HIPCHK(hipMemcpy2DAsync(C, (size_t)ldC * sizeof(float), Dd, np * sizeof(float),
                                ctx->n * sizeof(float), B, hipMemcpyDefault, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ctx->ev_solve0, ctx->ev_solve1));
        if (ms_out) *ms_out = ms / (float)repeats;
    } catch (const HipFail& f) {
        set_err(err, errlen, hip_msg(f));
        rc = SS_HIP_ERUNTIME;
    }
    if (Rd) (void)hipFree(Rd);
    if (Dd) (void)hipFree(Dd);
    return rc;
}

// G[s][:] = A^T a_{cols[s]} for up to 32 columns in one HBM-bound pass (lookahead sweep kernel)
template <typename T>
int gram_cols_impl(ss_hip_ctx* ctx, const uint32_t* cols, size_t S, T* G, ptrdiff_t ldG, int repeats,
                   float* ms_out, char* err, size_t errlen)
{
    if (ctx && ctx->kind != 0) { set_err(err, errlen, "this entry point needs a Homotopy context (an IRLS context holds the factorised matrix)"); return SS_HIP_EINVAL; }
    if (!ctx || !cols || !G || S == 0 || S > 32) { set_err(err, errlen, "gram_cols: need 1..32 columns"); return SS_HIP_EINVAL; }
    if (ctx->is_f64 != (sizeof(T) == 8)) { set_err(err, errlen, "gram_cols: element type of the call does not match the context"); return SS_HIP_ETYPE; }
    for (size_t s = 0; s < S; ++s)
        if (cols[s] >= ctx->n) { set_err(err, errlen, "gram_cols: column index out of range"); return SS_HIP_EINVAL; }
    if (repeats < 1) repeats = 1;
    uint32_t* dlist = nullptr;
    T* Dd = nullptr;
    int rc = SS_HIP_OK;
    try {
        HIPCHK(hipSetDevice(ctx->device));
        uint32_t h[64];
        for (int s = 0; s < 32; ++s) {
            h[s] = (size_t)s < S ? cols[s] : 0xffffffffu;
            h[32 + s] = (size_t)s < S ? (uint32_t)s : 0xffffffffu;
        }
        const size_t np = ctx->n_pad;
        HIPCHK(hipMalloc(&dlist, sizeof(h)));
        HIPCHK(hipMalloc(&Dd, 32 * np * sizeof(T)));
        HIPCHK(hipMemcpyAsync(dlist, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipEventRecord(ctx->ev_solve0, ctx->stream));
        for (int i = 0; i < repeats; ++i)
            HIPCHK(launch_gemm32(ctx, dlist, dlist + 32, Dd, (uint32_t)np, nullptr));
        HIPCHK(hipEventRecord(ctx->ev_solve1, ctx->stream));
        HIPCHK(hipMemcpy2DAsync(G, (size_t)ldG * sizeof(T), Dd, np * sizeof(T), ctx->n * sizeof(T), S,
                                hipMemcpyDefault, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ctx->ev_solve0, ctx->ev_solve1));
        if (ms_out) *ms_out = ms / (float)repeats;
    } catch (const HipFail& f) {
        set_err(err, errlen, hip_msg(f));
        rc = SS_HIP_ERUNTIME;
    }
    if (dlist) (void)hipFree(dlist);
    if (Dd) (void)hipFree(Dd);
    return rc;
}

int subset_gram_impl(ss_hip_ctx* ctx, const uint32_t* cols, float* Gs, int repeats, float* ms_out, char* err, size_t errlen)
{
    if (ctx && ctx->kind != 0) { set_err(err, errlen, "this entry point needs a Homotopy context (an IRLS context holds the factorised matrix)"); return SS_HIP_EINVAL; }
    if (!ctx || !cols || !Gs) { set_err(err, errlen, "subset_gram: null argument"); return SS_HIP_EINVAL; }
    if (ctx->is_f64) { set_err(err, errlen, "subset_gram: fp32 contexts only"); return SS_HIP_ETYPE; }
    if (repeats < 1) repeats = 1;
    uint32_t* dcols = nullptr;
    float* dG = nullptr;
    int rc = SS_HIP_OK;
    try {
        HIPCHK(hipSetDevice(ctx->device));
        HIPCHK(hipMalloc(&dcols, kSoloWidth * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&dG, (size_t)kSoloWidth * kSoloWidth * sizeof(float)));
        HIPCHK(hipMemcpyAsync(dcols, cols, kSoloWidth * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipEventRecord(ctx->ev_solve0, ctx->stream));
        for (int i = 0; i < repeats; ++i)
            HIPCHK(launch_subset_gram_f32(ctx, dcols, dG, nullptr, nullptr, nullptr, 0));
        HIPCHK(hipEventRecord(ctx->ev_solve1, ctx->stream));
        HIPCHK(hipMemcpyAsync(Gs, dG, (size_t)kSoloWidth * kSoloWidth * sizeof(float), hipMemcpyDefault, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ctx->ev_solve0, ctx->ev_solve1));
        if (ms_out) *ms_out = ms / (float)repeats;
    } catch (const HipFail& f) {
        set_err(err, errlen, hip_msg(f));
        rc = SS_HIP_ERUNTIME;
    }
    if (dcols) (void)hipFree(dcols);
    if (dG) (void)hipFree(dG);
    return rc;
}

template <typename T>
int reconstruct_impl(ss_hip_ctx* ctx, const T* x, T* y, char* err, size_t errlen)
{
    if (ctx && ctx->kind != 0) { set_err(err, errlen, "this entry point needs a Homotopy context (an IRLS context holds the factorised matrix)"); return SS_HIP_EINVAL; }
    if (!ctx || !x || !y) { set_err(err, errlen, "reconstruct: null argument"); return SS_HIP_EINVAL; }
    if (ctx->is_f64 != (sizeof(T) == 8)) { set_err(err, errlen, "reconstruct: type mismatch"); return SS_HIP_ETYPE; }
    try {
        HIPCHK(hipSetDevice(ctx->device));
        Workspace<T>& ws = *ws_of<T>(ctx);
        copy_in<T>(ctx, ws.q, x, 1, ctx->n);
        HIPCHK(launch_gemv_n<T>(ctx, ws.q, ws.rhs));
        copy_out<T>(ctx, y, 1, ws.rhs, ctx->m);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        // rhs padding must stay zero for the sweeps: rows >= m were not written
    } catch (const HipFail& f) {
        set_err(err, errlen, hip_msg(f));
        return SS_HIP_ERUNTIME;
    }
    return SS_HIP_OK;
}


// ---- IRLS (irls.hip): y up, one launch, x and the report down ---------------------------------
template <typename T>
int irls_solve_impl(ss_hip_ctx* ctx, const T* y, ptrdiff_t incy, T tol, uint32_t max_iter, T* x, ptrdiff_t incx,
                    uint32_t* iter_out, double* err_out, int* spd_failure, char* err, size_t errlen)
{
    if (!ctx) { set_err(err, errlen, "irls_solve: null context"); return SS_HIP_EINVAL; }
    if (ctx->kind != 1) { set_err(err, errlen, "irls_solve: this context was not created for IRLS"); return SS_HIP_EINVAL; }
    if (ctx->is_f64 != (sizeof(T) == 8)) {
        set_err(err, errlen, "irls_solve: element type of the call does not match the context");
        return SS_HIP_ETYPE;
    }
    if (!y || !x) { set_err(err, errlen, "irls_solve: y and x must not be null"); return SS_HIP_EINVAL; }
    if (max_iter == 0) { set_err(err, errlen, "irls_solve: max_iterations must be > 0"); return SS_HIP_EINVAL; }   // irls-cpu.cpp:78
    if (incy <= 0 || incx <= 0) { set_err(err, errlen, "irls_solve: vector increments must be positive"); return SS_HIP_EINVAL; }
    try {
        HIPCHK(hipSetDevice(ctx->device));
        copy_in<T>(ctx, irls_y_buffer<T>(ctx), y, incy, ctx->m);
        IrlsResult res{};
        HIPCHK(irls_solve<T>(ctx, tol, max_iter, &res));
        copy_out<T>(ctx, x, incx, irls_x_buffer<T>(ctx), ctx->n);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (iter_out) *iter_out = res.iter;
        if (err_out) *err_out = res.solution_error;
        if (spd_failure) *spd_failure = (int)res.spd_failure;
        ctx->stats.solves += 1;
        ctx->stats.iterations += res.iter;
    } catch (const HipFail& f) {
        set_err(err, errlen, hip_msg(f));
        return SS_HIP_ERUNTIME;
    }
    return SS_HIP_OK;
}

}  // namespace

// ---- C-ABI ----------------------------------------------------------------------------

extern "C" {

int ss_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* ss_hip_version(void) { return "0.1.0"; }

ss_hip_ctx* ss_hip_homotopy_create_f32(const float* A, size_t m, size_t n, ptrdiff_t stride_row,
                                       ptrdiff_t stride_col, int device, char* err, size_t errlen)
{
    return create_impl<float>(A, m, n, stride_row, stride_col, device, err, errlen);
}

ss_hip_ctx* ss_hip_homotopy_create_f64(const double* A, size_t m, size_t n, ptrdiff_t stride_row,
                                       ptrdiff_t stride_col, int device, char* err, size_t errlen)
{
    return create_impl<double>(A, m, n, stride_row, stride_col, device, err, errlen);
}

ss_hip_ctx* ss_hip_irls_create_f32(const float* A, size_t m, size_t n, ptrdiff_t stride_row, ptrdiff_t stride_col,
                                   int device, char* err, size_t errlen)
{
    return create_impl<float>(A, m, n, stride_row, stride_col, device, err, errlen, 1);
}

ss_hip_ctx* ss_hip_irls_create_f64(const double* A, size_t m, size_t n, ptrdiff_t stride_row, ptrdiff_t stride_col,
                                   int device, char* err, size_t errlen)
{
    return create_impl<double>(A, m, n, stride_row, stride_col, device, err, errlen, 1);
}

int ss_hip_irls_solve_f32(ss_hip_ctx* ctx, const float* y, ptrdiff_t incy, float tol, uint32_t max_iter, float* x,
                          ptrdiff_t incx, uint32_t* iter_out, double* err_out, int* spd_failure, char* err, size_t errlen)
{
    return irls_solve_impl<float>(ctx, y, incy, tol, max_iter, x, incx, iter_out, err_out, spd_failure, err, errlen);
}

int ss_hip_irls_solve_f64(ss_hip_ctx* ctx, const double* y, ptrdiff_t incy, double tol, uint32_t max_iter, double* x,
                          ptrdiff_t incx, uint32_t* iter_out, double* err_out, int* spd_failure, char* err, size_t errlen)
{
    return irls_solve_impl<double>(ctx, y, incy, tol, max_iter, x, incx, iter_out, err_out, spd_failure, err, errlen);
}

void ss_hip_irls_destroy(ss_hip_ctx* ctx) { ss_hip_homotopy_destroy(ctx); }

void ss_hip_homotopy_destroy(ss_hip_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream4) (void)hipStreamSynchronize(ctx->stream4);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws) {
        if (ctx->is_f64) free_ws(static_cast<Workspace<double>*>(ctx->ws));
        else free_ws(static_cast<Workspace<float>*>(ctx->ws));
    }
    sship::irls_free(ctx);
    sship::colshard_destroy(ctx);
    if (ctx->gram_reserve_thread != nullptr) {
        std::thread* th = static_cast<std::thread*>(ctx->gram_reserve_thread);
        th->join();
        delete th;
        ctx->gram_reserve_thread = nullptr;
    }
    if (ctx->gram_reserved) (void)hipFree(ctx->gram_reserved);
    if (ctx->gram_full) (void)hipFree(ctx->gram_full);
    if (ctx->c0_batch) (void)hipFree(ctx->c0_batch);
    if (ctx->sub_buf) (void)hipFree(ctx->sub_buf);
    if (ctx->sub_dbg) (void)hipFree(ctx->sub_dbg);
    sship::screen_free(ctx);
    if (ctx->ev_sub_sel) (void)hipEventDestroy(ctx->ev_sub_sel);
    if (ctx->ev_c0a) (void)hipEventDestroy(ctx->ev_c0a);
    if (ctx->ev_c0b) (void)hipEventDestroy(ctx->ev_c0b);
    if (ctx->bcol_cache) (void)hipFree(ctx->bcol_cache);
    if (ctx->bcol_slot) (void)hipFree(ctx->bcol_slot);
    if (ctx->bcol_lists) (void)hipFree(ctx->bcol_lists);
    if (ctx->rec_stage) (void)hipFree(ctx->rec_stage);
    if (ctx->At) (void)hipFree(ctx->At);
    if (ctx->host_flags) (void)hipHostFree(ctx->host_flags);
    if (ctx->hs_pinned) (void)hipHostFree(ctx->hs_pinned);
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    if (ctx->ev_solve0) (void)hipEventDestroy(ctx->ev_solve0);
    if (ctx->ev_solve1) (void)hipEventDestroy(ctx->ev_solve1);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->se_count) (void)hipFree(ctx->se_count);
    if (ctx->ev_gate) (void)hipEventDestroy(ctx->ev_gate);
    if (ctx->ev_b0) (void)hipEventDestroy(ctx->ev_b0);
    if (ctx->ev_join3) (void)hipEventDestroy(ctx->ev_join3);
    if (ctx->ev_join4) (void)hipEventDestroy(ctx->ev_join4);
    if (ctx->stream4) (void)hipStreamDestroy(ctx->stream4);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int ss_hip_homotopy_solve_f32(ss_hip_ctx* ctx, const float* y, ptrdiff_t incy, float tol,
                              uint32_t max_iter, float* x, ptrdiff_t incx, uint32_t* iter_out,
                              double* err_out, char* err, size_t errlen)
{
    return solve_impl<float>(ctx, y, incy, tol, max_iter, x, incx, iter_out, err_out, err, errlen);
}

int ss_hip_homotopy_solve_f64(ss_hip_ctx* ctx, const double* y, ptrdiff_t incy, double tol,
                              uint32_t max_iter, double* x, ptrdiff_t incx, uint32_t* iter_out,
                              double* err_out, char* err, size_t errlen)
{
    return solve_impl<double>(ctx, y, incy, tol, max_iter, x, incx, iter_out, err_out, err, errlen);
}

int ss_hip_omp_solve_f32(ss_hip_ctx* ctx, const float* y, ptrdiff_t incy, float tol, uint32_t max_iter,
                         float* x, ptrdiff_t incx, uint32_t* iter_out, double* err_out, char* err, size_t errlen)
{
    Route r; r.omp = true;
    return solve_impl<float>(ctx, y, incy, tol, max_iter, x, incx, iter_out, err_out, err, errlen, r);
}

int ss_hip_omp_solve_f64(ss_hip_ctx* ctx, const double* y, ptrdiff_t incy, double tol, uint32_t max_iter,
                         double* x, ptrdiff_t incx, uint32_t* iter_out, double* err_out, char* err, size_t errlen)
{
    Route r; r.omp = true;
    return solve_impl<double>(ctx, y, incy, tol, max_iter, x, incx, iter_out, err_out, err, errlen, r);
}

int ss_hip_homotopy_solve_batch_f32(ss_hip_ctx* ctx, const float* Y, size_t B, ptrdiff_t y_stride,
                                    ptrdiff_t incy, float tol, uint32_t max_iter, float* X,
                                    ptrdiff_t x_stride, ptrdiff_t incx, uint32_t* iter_out,
                                    double* err_out, char* err, size_t errlen)
{
    return solve_batch_impl<float>(ctx, Y, B, y_stride, incy, tol, max_iter, X, x_stride, incx,
                                   iter_out, err_out, err, errlen);
}

int ss_hip_homotopy_solve_batch_f64(ss_hip_ctx* ctx, const double* Y, size_t B, ptrdiff_t y_stride,
                                    ptrdiff_t incy, double tol, uint32_t max_iter, double* X,
                                    ptrdiff_t x_stride, ptrdiff_t incx, uint32_t* iter_out,
                                    double* err_out, char* err, size_t errlen)
{
    return solve_batch_impl<double>(ctx, Y, B, y_stride, incy, tol, max_iter, X, x_stride, incx,
                                    iter_out, err_out, err, errlen);
}

size_t ss_hip_record_bytes(uint32_t kmax, int is_f64) { return record_bytes(kmax, is_f64 ? 8 : 4); }

int ss_hip_homotopy_solve_batch_compact_f32(ss_hip_ctx* ctx, const float* Y, size_t B, ptrdiff_t y_stride, ptrdiff_t incy,
                                            float tol, uint32_t max_iter, uint32_t kmax, void* records, char* err, size_t errlen)
{
    if (!records) { set_err(err, errlen, "solve_batch_compact: records must not be null"); return SS_HIP_EINVAL; }
    return solve_batch_impl<float>(ctx, Y, B, y_stride, incy, tol, max_iter, nullptr, 0, 1, nullptr, nullptr, err, errlen, records, kmax);
}

int ss_hip_homotopy_solve_batch_compact_f64(ss_hip_ctx* ctx, const double* Y, size_t B, ptrdiff_t y_stride, ptrdiff_t incy,
                                            double tol, uint32_t max_iter, uint32_t kmax, void* records, char* err, size_t errlen)
{
    if (!records) { set_err(err, errlen, "solve_batch_compact: records must not be null"); return SS_HIP_EINVAL; }
    return solve_batch_impl<double>(ctx, Y, B, y_stride, incy, tol, max_iter, nullptr, 0, 1, nullptr, nullptr, err, errlen, records, kmax);
}

int ss_hip_gemv_t_f32(ss_hip_ctx* ctx, const float* r, float* c, int repeats, float* ms_out,
                      char* err, size_t errlen)
{
    return gemv_t_impl<float>(ctx, r, c, repeats, ms_out, err, errlen);
}

int ss_hip_gemv_t_f64(ss_hip_ctx* ctx, const double* r, double* c, int repeats, float* ms_out,
                      char* err, size_t errlen)
{
    return gemv_t_impl<double>(ctx, r, c, repeats, ms_out, err, errlen);
}

int ss_hip_gemm_t_f32(ss_hip_ctx* ctx, const float* R, size_t B, ptrdiff_t ldR, float* C, ptrdiff_t ldC,
                      int repeats, float* ms_out, char* err, size_t errlen)
{
    return gemm_t_impl(ctx, R, B, ldR, C, ldC, repeats, ms_out, err, errlen);
}

int ss_hip_gram_cols_f32(ss_hip_ctx* ctx, const uint32_t* cols, size_t S, float* G, ptrdiff_t ldG, int repeats,
                         float* ms_out, char* err, size_t errlen)
{
    return gram_cols_impl<float>(ctx, cols, S, G, ldG, repeats, ms_out, err, errlen);
}

int ss_hip_gram_cols_f64(ss_hip_ctx* ctx, const uint32_t* cols, size_t S, double* G, ptrdiff_t ldG, int repeats,
                         float* ms_out, char* err, size_t errlen)
{
    return gram_cols_impl<double>(ctx, cols, S, G, ldG, repeats, ms_out, err, errlen);
}

int ss_hip_subset_gram_f32(ss_hip_ctx* ctx, const uint32_t* cols, float* Gs, int repeats, float* ms_out, char* err, size_t errlen)
{
    return subset_gram_impl(ctx, cols, Gs, repeats, ms_out, err, errlen);
}

int ss_hip_reconstruct_f32(ss_hip_ctx* ctx, const float* x, float* y, char* err, size_t errlen)
{
    return reconstruct_impl<float>(ctx, x, y, err, errlen);
}

int ss_hip_reconstruct_f64(ss_hip_ctx* ctx, const double* x, double* y, char* err, size_t errlen)
{
    return reconstruct_impl<double>(ctx, x, y, err, errlen);
}

int ss_hip_set_profiling(ss_hip_ctx* ctx, int profiling)
{
    if (!ctx) return SS_HIP_EINVAL;
    ctx->profiling = profiling ? 1 : 0;
    return SS_HIP_OK;
}

int ss_hip_get_stats(ss_hip_ctx* ctx, ss_hip_stats* out)
{
    if (!ctx || !out) return SS_HIP_EINVAL;
    if (ctx->screen != nullptr && ctx->stats.screen_signals + ctx->stats.screen_redone != 0) {
        (void)hipSetDevice(ctx->device);
        ctx->stats.screen_headroom = sship::screen_read_headroom(ctx);
    }
    *out = ctx->stats;
    return SS_HIP_OK;
}

int ss_hip_reset_stats(ss_hip_ctx* ctx)
{
    if (!ctx) return SS_HIP_EINVAL;
    const uint64_t b2 = ctx->stats.sweep_bytes, b1 = ctx->stats.sweep1_bytes, b32 = ctx->stats.sweep32_bytes;
    const uint64_t tc32 = ctx->stats.sweep32_timed_cols;
    const uint64_t b64 = ctx->stats.sweep64_bytes, f64 = ctx->stats.sweep64_flops;
    ctx->stats = ss_hip_stats{};
    ctx->stats.sweep_bytes = b2;
    ctx->stats.sweep1_bytes = b1;
    ctx->stats.sweep32_bytes = b32;
    ctx->stats.sweep32_timed_cols = tc32;
    ctx->stats.sweep64_bytes = b64;
    ctx->stats.sweep64_flops = f64;
    return SS_HIP_OK;
}

int ss_hip_set_option(ss_hip_ctx* ctx, const char* key, long value)
{
    if (!ctx || !key) return SS_HIP_EINVAL;
    if (!std::strcmp(key, "sweep_variant")) { ctx->sweep_variant = (int)value; return SS_HIP_OK; }
    if (!std::strcmp(key, "lookahead"))     { ctx->lookahead = (int)value; return SS_HIP_OK; }
    if (!std::strcmp(key, "temporal_cols")) { ctx->temporal_cols = std::max<long>(0, value); return SS_HIP_OK; }
    if (!std::strcmp(key, "strict_sign"))   { ctx->strict_sign = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "screen_first8")) { ctx->screen_first8 = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "screen_rescue")) { ctx->screen_rescue = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "trace"))         { ctx->tracing = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "zero_on_removal")) { ctx->zero_on_removal = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "tie_guard"))     { ctx->tie_guard = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "profile_every")) { ctx->profile_every = (int)std::max<long>(1, value); return SS_HIP_OK; }
    if (!std::strcmp(key, "profile_solve_every")) { ctx->profile_solve_every = (int)std::max<long>(1, value); ctx->prof_solve_tick = 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "engine"))        { ctx->engine = (int)std::max<long>(0, std::min<long>(3, value)); return SS_HIP_OK; }
    if (!std::strcmp(key, "tie_rerun"))     { ctx->tie_rerun = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "ro_force_resweep")) { ctx->ro_force_resweep = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "ro_staged"))     { ctx->ro_staged = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_subset"))  { ctx->batch_subset = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "ro_slots"))      { if (value < 1 || value > 8) return SS_HIP_EINVAL; ctx->ro_slots = (int)value; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_fused_scan")) { ctx->batch_fused_scan = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "cq_vec4"))       { ctx->cq_vec4 = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "cq_cols"))       { ctx->cq_cols = (int)value; return SS_HIP_OK; }
    if (!std::strcmp(key, "cq_rows"))       { ctx->cq_rows = (int)value; return SS_HIP_OK; }
    if (!std::strcmp(key, "sweep32_variant")) { ctx->sweep32_variant = (int)std::max<long>(0, std::min<long>(9, value)); return SS_HIP_OK; }
    if (!std::strcmp(key, "first_sweep_cols")) { ctx->first_sweep_cols = value > 32 ? 64 : 32; return SS_HIP_OK; }
    if (!std::strcmp(key, "early_solo"))    { ctx->early_solo = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "sweep_cols_f64")) { ctx->sweep_cols_f64 = value > 32 ? 64 : 32; return SS_HIP_OK; }
    if (!std::strcmp(key, "sweep_cols_f64_late")) { ctx->sweep_cols_f64_late = value > 32 ? 64 : 32; return SS_HIP_OK; }
    if (!std::strcmp(key, "early_probe"))   { ctx->early_probe = (int)value; return SS_HIP_OK; }
    if (!std::strcmp(key, "early_pass"))    { ctx->early_pass = (int)value; return SS_HIP_OK; }
    if (!std::strcmp(key, "early_adapt"))   { ctx->early_adapt = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "scan_blocks"))   { ctx->scan_blocks = (int)std::max<long>(0, value); return SS_HIP_OK; }
    if (!std::strcmp(key, "early_se"))      { ctx->early_se = (int)std::max<long>(0, std::min<long>(3, value)); return SS_HIP_OK; }
    if (!std::strcmp(key, "pass_dbg_ptr"))  {   // developer aid: device buffer of 1 + 4 * 4096 u64 (0 = off), tools/probe_pass_trace.py
        (void)hipSetDevice(ctx->device);
        return sship::set_pass_debug(reinterpret_cast<uint64_t*>(static_cast<uintptr_t>(value))) == hipSuccess ? SS_HIP_OK : SS_HIP_ERUNTIME;
    }
    if (!std::strcmp(key, "sweep_f64_variant")) { ctx->sweep_f64_variant = (int)std::max<long>(0, std::min<long>(2, value)); return SS_HIP_OK; }
    if (!std::strcmp(key, "la_fused"))      { ctx->la_fused = (int)std::max<long>(0, std::min<long>(3, value)); return SS_HIP_OK; }
    if (!std::strcmp(key, "solo_subset"))   { ctx->solo_subset = (int)std::max<long>(0, std::min<long>(256, value)); return SS_HIP_OK; }
    if (!std::strcmp(key, "solo_full_gram")) { ctx->solo_full_gram = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "cache_mib"))     { ctx->cache_mib = std::max<long>(16, value); return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_min"))     { ctx->batch_min = (int)std::max<long>(2, value); return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_gram_min")) { ctx->batch_gram_min = (int)std::max<long>(0, value); return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_cols_min")) { ctx->batch_cols_min = (int)std::max<long>(0, value); return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_cols_max")) { ctx->batch_cols_max = (int)std::max<long>(0, value); return SS_HIP_OK; }
    if (!std::strcmp(key, "gram_full_gib")) { ctx->gram_full_gib = std::max<long>(0, value); return SS_HIP_OK; }
    if (!std::strcmp(key, "gram_full_after")) { ctx->gram_full_after = std::max<long>(0, value); return SS_HIP_OK; }
    if (!std::strcmp(key, "gram_single"))   { ctx->gram_single = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "gram_symmetric")) { ctx->gram_symmetric = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_chunk"))   { ctx->batch_chunk = (int)std::max<long>(4, value); return SS_HIP_OK; }
    if (!std::strcmp(key, "screen_single")) {
        // (setting the option also forgets what the context has learnt about its signals: the step-aside counters start again)
        ctx->screen_single = (int)std::max<long>(0, std::min<long>(2, value));
        ctx->sub_off_solves = 0; ctx->sub_seen = 0; ctx->sub_failed = 0; ctx->res_off_solves = 0; ctx->res_seen = 0; ctx->res_failed = 0;
        return SS_HIP_OK;
    }
    if (!std::strcmp(key, "screen_first16")) { ctx->screen_first16 = value != 0 ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_screen"))  { ctx->batch_screen = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "screen_resident")) { ctx->screen_resident = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "screen_recheck")) { ctx->screen_recheck = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "gram_reserve")) { ctx->gram_reserve = value ? 1 : 0; return SS_HIP_OK; }
    if (!std::strcmp(key, "colshard_fail_prepare")) { ctx->colshard_fail_prepare = value ? 1 : 0; return SS_HIP_OK; }
    return SS_HIP_EINVAL;
}

int ss_hip_get_trace(ss_hip_ctx* ctx, uint32_t capacity, uint32_t* idx, uint8_t* added, double* gamma,
                     double* c_inf, uint32_t* count)
{
    if (!ctx || !count) return SS_HIP_EINVAL;
    const uint32_t n = (uint32_t)std::min<size_t>(capacity, ctx->last_trace.size());
    for (uint32_t i = 0; i < n; ++i) {
        if (idx) idx[i] = ctx->last_trace[i].idx;
        if (added) added[i] = (uint8_t)ctx->last_trace[i].added;
        if (gamma) gamma[i] = ctx->last_trace[i].gamma;
        if (c_inf) c_inf[i] = ctx->last_trace[i].c_inf;
    }
    *count = (uint32_t)ctx->last_trace.size();
    return SS_HIP_OK;
}

int ss_hip_get_option(ss_hip_ctx* ctx, const char* key, long* value)
{
    if (!ctx || !key || !value) return SS_HIP_EINVAL;
    if (!std::strcmp(key, "sweep_variant")) { *value = ctx->sweep_variant; return SS_HIP_OK; }
    if (!std::strcmp(key, "lookahead"))     { *value = ctx->lookahead; return SS_HIP_OK; }
    if (!std::strcmp(key, "temporal_cols")) { *value = ctx->temporal_cols; return SS_HIP_OK; }
    if (!std::strcmp(key, "strict_sign"))   { *value = ctx->strict_sign; return SS_HIP_OK; }
    if (!std::strcmp(key, "screen_first8")) { *value = ctx->screen_first8; return SS_HIP_OK; }
    if (!std::strcmp(key, "screen_rescue")) { *value = ctx->screen_rescue; return SS_HIP_OK; }
    if (!std::strcmp(key, "trace"))         { *value = ctx->tracing; return SS_HIP_OK; }
    if (!std::strcmp(key, "zero_on_removal")) { *value = ctx->zero_on_removal; return SS_HIP_OK; }
    if (!std::strcmp(key, "tie_guard"))     { *value = ctx->tie_guard; return SS_HIP_OK; }
    if (!std::strcmp(key, "profile_every")) { *value = ctx->profile_every; return SS_HIP_OK; }
    if (!std::strcmp(key, "dbg_ndone") || !std::strcmp(key, "dbg_skip_sum")) {
        // debugging aids: device-side counters of the last batched solve (fp32 contexts)
        if (ctx->is_f64 || !ctx->ws) return SS_HIP_EINVAL;
        Workspace<float>* w = static_cast<Workspace<float>*>(ctx->ws);
        if (hipSetDevice(ctx->device) != hipSuccess) return SS_HIP_ERUNTIME;
        if (!std::strcmp(key, "dbg_ndone")) {
            uint32_t v = 0;
            if (hipMemcpy(&v, w->ndone, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return SS_HIP_ERUNTIME;
            *value = (long)v;
        } else {
            *value = -1;   // (the tile list is rebuilt every round; nothing meaningful to report)
        }
        return SS_HIP_OK;
    }
    if (!std::strcmp(key, "engine"))        { *value = ctx->engine; return SS_HIP_OK; }
    if (!std::strcmp(key, "tie_rerun"))     { *value = ctx->tie_rerun; return SS_HIP_OK; }
    if (!std::strcmp(key, "ro_force_resweep")) { *value = ctx->ro_force_resweep; return SS_HIP_OK; }
    if (!std::strcmp(key, "ro_staged"))     { *value = ctx->ro_staged; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_subset"))  { *value = ctx->batch_subset; return SS_HIP_OK; }
    if (!std::strcmp(key, "ro_slots"))      { *value = ctx->ro_slots; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_fused_scan")) { *value = ctx->batch_fused_scan; return SS_HIP_OK; }
    if (!std::strcmp(key, "cq_vec4"))       { *value = ctx->cq_vec4; return SS_HIP_OK; }
    if (!std::strcmp(key, "cq_cols"))       { *value = ctx->cq_cols; return SS_HIP_OK; }
    if (!std::strcmp(key, "cq_rows"))       { *value = ctx->cq_rows; return SS_HIP_OK; }
    if (!std::strcmp(key, "la_fused"))      { *value = ctx->la_fused; return SS_HIP_OK; }
    if (!std::strcmp(key, "solo_subset"))   { *value = ctx->solo_subset; return SS_HIP_OK; }
    if (!std::strcmp(key, "sweep32_variant")) { *value = ctx->sweep32_variant; return SS_HIP_OK; }
    if (!std::strcmp(key, "first_sweep_cols")) { *value = ctx->first_sweep_cols; return SS_HIP_OK; }
    if (!std::strcmp(key, "early_solo"))    { *value = ctx->early_solo; return SS_HIP_OK; }
    if (!std::strcmp(key, "early_pass"))    { *value = ctx->early_pass; return SS_HIP_OK; }
    if (!std::strcmp(key, "early_adapt"))   { *value = ctx->early_adapt; return SS_HIP_OK; }
    if (!std::strcmp(key, "sweep_cols_f64")) { *value = ctx->sweep_cols_f64; return SS_HIP_OK; }
    if (!std::strcmp(key, "sweep_cols_f64_late")) { *value = ctx->sweep_cols_f64_late; return SS_HIP_OK; }
    if (!std::strcmp(key, "cache_mib"))     { *value = ctx->cache_mib; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_min"))     { *value = ctx->batch_min; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_gram_min")) { *value = ctx->batch_gram_min; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_cols_min")) { *value = ctx->batch_cols_min; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_cols_max")) { *value = ctx->batch_cols_max; return SS_HIP_OK; }
    if (!std::strcmp(key, "gram_full_gib")) { *value = ctx->gram_full_gib; return SS_HIP_OK; }
    if (!std::strcmp(key, "gram_full_after")) { *value = ctx->gram_full_after; return SS_HIP_OK; }
    if (!std::strcmp(key, "gram_single"))   { *value = ctx->gram_single; return SS_HIP_OK; }
    if (!std::strcmp(key, "gram_symmetric")) { *value = ctx->gram_symmetric; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_chunk"))   { *value = ctx->batch_chunk; return SS_HIP_OK; }
    if (!std::strcmp(key, "screen_single")) { *value = ctx->screen_single; return SS_HIP_OK; }
    if (!std::strcmp(key, "screen_first16")) { *value = ctx->screen_first16; return SS_HIP_OK; }
    if (!std::strcmp(key, "batch_screen"))  { *value = ctx->batch_screen; return SS_HIP_OK; }
    if (!std::strcmp(key, "screen_resident")) { *value = ctx->screen_resident; return SS_HIP_OK; }
    if (!std::strcmp(key, "screen_recheck")) { *value = ctx->screen_recheck; return SS_HIP_OK; }
    if (!std::strcmp(key, "gram_reserve")) { *value = ctx->gram_reserve; return SS_HIP_OK; }
    if (!std::strcmp(key, "colshard_fail_prepare")) { *value = ctx->colshard_fail_prepare; return SS_HIP_OK; }
    return SS_HIP_EINVAL;
}

int ss_hip_ctx_info(const ss_hip_ctx* ctx, size_t* m, size_t* n, int* is_f64, int* device)
{
    if (!ctx) return SS_HIP_EINVAL;
    if (m) *m = ctx->m;
    if (n) *n = ctx->n;
    if (is_f64) *is_f64 = ctx->is_f64;
    if (device) *device = ctx->device;
    return SS_HIP_OK;
}

}  // extern "C"
