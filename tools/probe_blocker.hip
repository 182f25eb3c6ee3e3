// Developer probe: what slows the 32-column pass down beside the speculative launch?  The pass (tiling chosen by
// sweep32_variant) is timed alone and beside a BLOCKER on another stream: one workgroup of 512 threads holding
// 150 KB of LDS (as the solo launch does) that (a) only sleeps, (b) also keeps reading a small buffer, (c) also
// keeps storing with agent scope.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_blocker.hip -Iinclude -Lsparse-solvers_amd/lib -lss_hip -Wl,-rpath,$PWD/sparse-solvers_amd/lib -o /tmp/probe_blocker
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <ctime>
#include <vector>
#include "ss_hip.h"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_blocker(uint32_t ms, int mode, float* buf, uint32_t words)
{
    extern __shared__ float lds[];
    const uint64_t t0 = wall_clock64();
    lds[threadIdx.x] = 1.f;
    float acc = 0.f;
    uint32_t i = threadIdx.x;
    while (wall_clock64() - t0 < (uint64_t)ms * 100000ull) {          // 100 MHz clock
        if (mode == 0) __builtin_amdgcn_s_sleep(32);
        else if (mode == 1) { acc += __builtin_nontemporal_load(&buf[i % words]); i += 512; }
        else if (mode == 2) { __hip_atomic_store(&buf[i % words], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); i += 512; }
        else if (mode == 3) { acc += lds[(i * 33u) % 8192u]; i += 1; }
        else {
            // mode 4: what the solo launch looks like to the CU — barriers, LDS traffic, ~60 live registers
            float r[48];
#pragma unroll
            for (int k = 0; k < 48; ++k) r[k] = lds[(threadIdx.x + 33u * k) % 8192u];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 48; ++k) acc = fmaf(r[k], r[(k + 7) % 48], acc);
            lds[(threadIdx.x * 5u + i) % 8192u] = acc;
            __syncthreads();
            i += 1;
        }
    }
    if (acc == 123.456f) buf[0] = acc;
}

__device__ __forceinline__ uint32_t hw_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}
__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v;
}
// every workgroup records where it ran and when it started, then lingers so that the whole grid is resident at once
__global__ void k_census(uint32_t* where, uint64_t* t_start, uint32_t spin_us)
{
    extern __shared__ float lds[];
    const uint64_t t0 = wall_clock64();
    if (threadIdx.x == 0) {
        where[blockIdx.x] = (xcc_id() & 0xf) << 16 | (hw_id() & 0xffff);
        t_start[blockIdx.x] = t0;
        lds[0] = 1.f;
    }
    while (wall_clock64() - t0 < (uint64_t)spin_us * 100ull) __builtin_amdgcn_s_sleep(8);
}

// a stand-in for the pass: every workgroup streams `mb` MiB of its own with 16-byte loads (what one CU can load is
// the bound), and records where and when it ran
typedef float pb_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream(const float* __restrict__ src, uint32_t mb, uint32_t* where, uint64_t* t0s, uint64_t* t1s, float* sink)
{
    extern __shared__ float lds[];
    const uint64_t t0 = wall_clock64();
    const pb_v4f* p = reinterpret_cast<const pb_v4f*>(src + (size_t)blockIdx.x * mb * 262144u);
    const uint32_t n16 = mb * 65536u;
    pb_v4f acc = { 0.f, 0.f, 0.f, 0.f };
    for (uint32_t i = threadIdx.x; i < n16; i += 1024u) {
        const pb_v4f a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + 256u),
                     c = __builtin_nontemporal_load(p + i + 512u), d = __builtin_nontemporal_load(p + i + 768u);
        acc += a + b + c + d;
    }
    lds[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    __syncthreads();
    if (threadIdx.x == 0) {
        where[blockIdx.x] = (xcc_id() & 0xf) << 16 | (hw_id() & 0xffff);
        t0s[blockIdx.x] = t0;
        t1s[blockIdx.x] = wall_clock64();
        if (lds[7] == 123.25f) sink[0] = lds[7];
    }
}

static int stream_probe(hipStream_t side, const float* src, bool blocker, int blocker_kb, float* buf)
{
    const int nwg = 512;
    uint32_t* d_where; uint64_t *d_t0, *d_t1;
    if (hipMalloc(&d_where, nwg * 4) != hipSuccess || hipMalloc(&d_t0, nwg * 8) != hipSuccess || hipMalloc(&d_t1, nwg * 8) != hipSuccess) return 1;
    std::vector<uint32_t> w(nwg); std::vector<uint64_t> t0(nwg), t1(nwg);
    hipFuncSetAttribute((const void*)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        if (blocker) {
            hipLaunchKernelGGL(k_blocker, dim3(1), dim3(512), blocker_kb * 1024, side, 20u, 0, buf, (1u << 18));
            const uint64_t w0 = (uint64_t)clock();
            while ((uint64_t)clock() - w0 < (uint64_t)CLOCKS_PER_SEC / 500) { }
        }
        hipLaunchKernelGGL(k_stream, dim3(nwg), dim3(256), 46080, nullptr, src, 4u, d_where, d_t0, d_t1, buf);
        hipDeviceSynchronize();
    }
    hipMemcpy(w.data(), d_where, nwg * 4, hipMemcpyDeviceToHost);
    hipMemcpy(t0.data(), d_t0, nwg * 8, hipMemcpyDeviceToHost);
    hipMemcpy(t1.data(), d_t1, nwg * 8, hipMemcpyDeviceToHost);
    uint64_t tmin = ~0ull, tmax = 0;
    for (int i = 0; i < nwg; ++i) { tmin = t0[i] < tmin ? t0[i] : tmin; tmax = t1[i] > tmax ? t1[i] : tmax; }
    // per XCC: number of workgroups, mean and max duration, latest end
    printf("  streaming stand-in (512 WGs x 4 MiB, 46 KB LDS)%s: total %.1f us\n", blocker ? (blocker_kb > 100 ? ", beside a 150-KB blocker" : ", beside a 66-KB blocker") : "", (double)(tmax - tmin) / 100.0);
    std::vector<int> cnt(16 * 256, 0);
    for (int i = 0; i < nwg; ++i) { const uint32_t v = w[i]; cnt[(v >> 16) * 256 + ((v >> 13) & 7) * 32 + ((v >> 12) & 1) * 16 + ((v >> 8) & 0xf)] += 1; }
    for (int x = 0; x < 8; ++x) {
        double sum = 0, mx = 0, end = 0; int c = 0, three = 0;
        for (int i = 0; i < nwg; ++i) if ((int)(w[i] >> 16) == x) {
            const double d = (double)(t1[i] - t0[i]) / 100.0;
            sum += d; mx = d > mx ? d : mx; c += 1;
            const double e = (double)(t1[i] - tmin) / 100.0; end = e > end ? e : end;
        }
        for (int k = 0; k < 256; ++k) if (cnt[x * 256 + k] >= 3) three += 1;
        printf("    XCC %d: %3d WGs, mean %.1f us, max %.1f us, last end %.1f us, CUs with 3+ WGs: %d\n", x, c, sum / (c ? c : 1), mx, end, three);
    }
    hipFree(d_where); hipFree(d_t0); hipFree(d_t1);
    return 0;
}

static int census(hipStream_t on, hipStream_t side, int nwg, int threads, int lds_bytes, bool blocker, float* buf)
{
    uint32_t* d_where; uint64_t* d_t;
    if (hipMalloc(&d_where, nwg * 4) != hipSuccess || hipMalloc(&d_t, nwg * 8) != hipSuccess) return 1;
    std::vector<uint32_t> w(nwg); std::vector<uint64_t> t(nwg);
    hipFuncSetAttribute((const void*)k_census, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (blocker) {
        hipLaunchKernelGGL(k_blocker, dim3(1), dim3(512), 150 * 1024, side, 20u, 0, buf, (1u << 18));
        const uint64_t w0 = (uint64_t)clock();
        while ((uint64_t)clock() - w0 < (uint64_t)CLOCKS_PER_SEC / 500) { }          // 2 ms: the blocker is resident
    }
    hipLaunchKernelGGL(k_census, dim3(nwg), dim3(threads), lds_bytes, on, d_where, d_t, 300u);
    hipStreamSynchronize(on);
    hipStreamSynchronize(side);
    hipMemcpy(w.data(), d_where, nwg * 4, hipMemcpyDeviceToHost);
    hipMemcpy(t.data(), d_t, nwg * 8, hipMemcpyDeviceToHost);
    // per (xcc, se, sh/cu) count
    std::vector<int> cnt(16 * 256, 0), per_xcc(16, 0);
    uint64_t tmin = ~0ull, tmax = 0;
    for (int i = 0; i < nwg; ++i) {
        const uint32_t v = w[i];
        const uint32_t xcc = v >> 16, se = (v >> 13) & 7, sh = (v >> 12) & 1, cu = (v >> 8) & 0xf;
        cnt[xcc * 256 + se * 32 + sh * 16 + cu] += 1;
        per_xcc[xcc] += 1;
        tmin = t[i] < tmin ? t[i] : tmin; tmax = t[i] > tmax ? t[i] : tmax;
    }
    int hist[32] = { 0 }, used = 0;
    for (int c : cnt) if (c) { hist[c < 31 ? c : 31] += 1; used += 1; }
    printf("  census %4d WGs x %3d threads, %5d B LDS%s: %d CUs used; WGs per CU histogram:", nwg, threads, lds_bytes,
           blocker ? ", beside the blocker" : "", used);
    for (int k = 1; k < 32; ++k) if (hist[k]) printf(" %dx%d", hist[k], k);
    printf("; per XCC:");
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("; start spread %.1f us\n", (double)(tmax - tmin) / 100.0);
    hipFree(d_where); hipFree(d_t);
    return 0;
}

int main(int argc, char** argv)
{
    const size_t m = 8192, n = 65536;
    std::vector<float> A(m * n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < A.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; A[i] = ((int64_t)(s >> 40) - (1 << 23)) * (1.0f / (1 << 23)) * 0.011f; }
    char err[512] = { 0 };
    ss_hip_ctx* ctx = ss_hip_homotopy_create_f32(A.data(), m, n, (ptrdiff_t)n, 1, 0, err, sizeof(err));
    if (!ctx) { printf("create failed: %s\n", err); return 1; }
    std::vector<uint32_t> cols(32);
    for (int i = 0; i < 32; ++i) cols[i] = (uint32_t)(i * 2039 + 17);
    std::vector<float> G((size_t)32 * n);
    float* buf;
    CHK(hipMalloc(&buf, 1 << 20));
    CHK(hipMemset(buf, 0, 1 << 20));
    hipStream_t side;
    CHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    CHK(hipFuncSetAttribute((const void*)k_blocker, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int b = 0; b < 2; ++b) {
        census(ctx ? nullptr : nullptr, side, 512, 256, 46080, b != 0, buf);        // the 128-column LDS tiling, 3 per CU
        census(nullptr, side, 2048, 64, 9216, b != 0, buf);                          // the single-wave tiling
        census(nullptr, side, 256, 512, 82944, b != 0, buf);                         // one 512-thread workgroup per CU
    }
    {
        float* src;
        CHK(hipMalloc(&src, (size_t)512 * 4 * 1048576));
        CHK(hipMemset(src, 0, (size_t)512 * 4 * 1048576));
        stream_probe(side, src, false, 0, buf);
        stream_probe(side, src, true, 150, buf);
        stream_probe(side, src, true, 66, buf);
        CHK(hipFree(src));
    }
    for (int variant : { 2 }) {
        ss_hip_set_option(ctx, "sweep32_variant", variant);
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) ss_hip_gram_cols_f32(ctx, cols.data(), 32, G.data(), (ptrdiff_t)n, 1, &ms, err, sizeof(err));
        printf("variant %d alone: %.3f ms\n", variant, ms);
        for (int mode = 0; mode < 5; ++mode) {
            for (int lds_kb : { 150, 110, 100, 90, 70, 66, 48, 8 }) {
                if (mode != 0 && mode != 4 && lds_kb != 150 && lds_kb != 8) continue;
                if (mode == 4 && lds_kb != 150 && lds_kb != 66 && lds_kb != 48) continue;
                float best = 1e9f, worst = 0.f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipLaunchKernelGGL(k_blocker, dim3(1), dim3(512), lds_kb * 1024, side, 200u, mode, buf, (1u << 18));
                    // (the pass call uploads nothing large: the blocker is resident by the time the pass starts; it lingers 200 ms)
                    ss_hip_gram_cols_f32(ctx, cols.data(), 32, G.data(), (ptrdiff_t)n, 1, &ms, err, sizeof(err));
                    CHK(hipStreamSynchronize(side));
                    best = ms < best ? ms : best; worst = ms > worst ? ms : worst;
                }
                printf("  beside a blocker (mode %d: %s, %3d KB LDS): %.3f .. %.3f ms\n", mode,
                       mode == 0 ? "sleeps" : mode == 1 ? "global loads" : mode == 2 ? "agent-scope stores" : mode == 3 ? "LDS reads" : "barriers + LDS + registers", lds_kb, best, worst);
            }
        }
    }
    ss_hip_homotopy_destroy(ctx);
    return 0;
}
