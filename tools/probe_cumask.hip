// Developer probe: does hipExtStreamCreateWithCUMask keep a stream's kernels off chosen CUs on this box, and
// does a 150-KB-LDS workgroup launched on another stream start at once while a long, LDS-light kernel fills
// the (masked) rest of the chip?   hipcc --offload-arch=gfx950 -O2 tools/probe_cumask.hip -o /tmp/probe_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

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

// every workgroup records where it ran, then lingers `spin_us` so that the grid spreads over the chip
__global__ void k_census(uint32_t* where, uint64_t* t_start, uint32_t spin_us)
{
    extern __shared__ float lds[];
    const uint64_t t0 = wall_clock64();
    if (threadIdx.x == 0) {
        const uint32_t h = hw_id();
        where[blockIdx.x] = (xcc_id() & 0xf) << 16 | (h & 0xffff);
        t_start[blockIdx.x] = t0;
        lds[0] = 1.f;
    }
    while (wall_clock64() - t0 < (uint64_t)spin_us * 100ull) __builtin_amdgcn_s_sleep(8);      // 100 MHz clock
}

static size_t distinct_cus(const std::vector<uint32_t>& w)
{
    std::set<uint32_t> s;
    for (uint32_t v : w) s.insert(((v >> 16) << 8) | (((v >> 13) & 7) << 5) | (((v >> 12) & 1) << 4) | ((v >> 8) & 0xf));   // xcc, se, sh, cu
    return s.size();
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    printf("CUs %d\n", prop.multiProcessorCount);
    const int NB = 4096;
    uint32_t* d_where; uint64_t* d_t;
    CHK(hipMalloc(&d_where, NB * 4)); CHK(hipMalloc(&d_t, NB * 8));
    std::vector<uint32_t> w(NB); std::vector<uint64_t> t(NB);
    hipStream_t s0;
    CHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_census, dim3(NB), dim3(64), 9216, s0, d_where, d_t, 200u);
    CHK(hipStreamSynchronize(s0));
    CHK(hipMemcpy(w.data(), d_where, NB * 4, hipMemcpyDeviceToHost));
    printf("unmasked stream: %zu distinct CUs\n", distinct_cus(w));
    for (int variant = 0; variant < 3; ++variant) {
        uint32_t mask[8];
        for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu;
        if (variant == 0) mask[0] &= ~1u;                         // bit 0 off
        if (variant == 1) mask[0] &= ~0xffu;                      // bits 0..7 off
        if (variant == 2) for (int i = 0; i < 8; ++i) mask[i] &= ~1u;   // bit 0 of every word off
        hipStream_t sm;
        hipError_t e = hipExtStreamCreateWithCUMask(&sm, 8, mask);
        if (e != hipSuccess) { printf("variant %d: hipExtStreamCreateWithCUMask failed: %s\n", variant, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipLaunchKernelGGL(k_census, dim3(NB), dim3(64), 9216, sm, d_where, d_t, 200u);
        CHK(hipStreamSynchronize(sm));
        CHK(hipMemcpy(w.data(), d_where, NB * 4, hipMemcpyDeviceToHost));
        printf("variant %d: masked stream used %zu distinct CUs\n", variant, distinct_cus(w));
        // concurrency: the long LDS-light kernel on the masked stream first, then one 150-KB workgroup elsewhere
        uint32_t* d_w2; uint64_t* d_t2;
        CHK(hipMalloc(&d_w2, 4)); CHK(hipMalloc(&d_t2, 8));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_census), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipLaunchKernelGGL(k_census, dim3(2048), dim3(64), 9216, sm, d_where, d_t, 400u);         // "sweep": 8 waves x 9 KB per CU, 0.4 ms
        hipLaunchKernelGGL(k_census, dim3(1), dim3(512), 150 * 1024, s0, d_w2, d_t2, 50u);        // "solo": a whole CU's LDS
        CHK(hipStreamSynchronize(s0)); CHK(hipStreamSynchronize(sm));
        uint64_t ts; CHK(hipMemcpy(&ts, d_t2, 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(t.data(), d_t, 2048 * 8, hipMemcpyDeviceToHost));
        uint64_t tmin = ~0ull;
        for (int i = 0; i < 2048; ++i) tmin = t[i] < tmin ? t[i] : tmin;
        printf("variant %d: 150-KB workgroup started %.1f us after the first sweep workgroup (sweep lasts 400 us)\n", variant, ((double)ts - (double)tmin) / 100.0);
        CHK(hipStreamDestroy(sm));
    }
    // the same without a mask: how long does the big workgroup wait?
    {
        uint32_t* d_w2; uint64_t* d_t2; hipStream_t s1;
        CHK(hipMalloc(&d_w2, 4)); CHK(hipMalloc(&d_t2, 8));
        CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        hipLaunchKernelGGL(k_census, dim3(2048), dim3(64), 9216, s1, d_where, d_t, 400u);
        hipLaunchKernelGGL(k_census, dim3(1), dim3(512), 150 * 1024, s0, d_w2, d_t2, 50u);
        CHK(hipStreamSynchronize(s0)); CHK(hipStreamSynchronize(s1));
        uint64_t ts; CHK(hipMemcpy(&ts, d_t2, 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(t.data(), d_t, 2048 * 8, hipMemcpyDeviceToHost));
        uint64_t tmin = ~0ull;
        for (int i = 0; i < 2048; ++i) tmin = t[i] < tmin ? t[i] : tmin;
        printf("no mask: 150-KB workgroup started %.1f us after the first sweep workgroup\n", ((double)ts - (double)tmin) / 100.0);
        // and the other order: big workgroup first
        hipLaunchKernelGGL(k_census, dim3(1), dim3(512), 150 * 1024, s0, d_w2, d_t2, 600u);
        hipLaunchKernelGGL(k_census, dim3(2048), dim3(64), 9216, s1, d_where, d_t, 400u);
        CHK(hipStreamSynchronize(s0)); CHK(hipStreamSynchronize(s1));
        CHK(hipMemcpy(&ts, d_t2, 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(t.data(), d_t, 2048 * 8, hipMemcpyDeviceToHost));
        uint64_t tmax = 0; tmin = ~0ull;
        for (int i = 0; i < 2048; ++i) { tmin = t[i] < tmin ? t[i] : tmin; tmax = t[i] > tmax ? t[i] : tmax; }
        printf("big first: sweep workgroups started %.1f .. %.1f us after the big one (it lingers 600 us)\n", ((double)tmin - (double)ts) / 100.0, ((double)tmax - (double)ts) / 100.0);
    }
    return 0;
}
