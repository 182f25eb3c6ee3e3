// Measurement aid (not part of the product): how fast can a CU-persistent workgroup stream the
// column-contiguous A (n columns x m floats, pitch m) with different per-step access shapes?
//   LPR   lanes per contiguous row segment (segment = LPR*16 bytes of one column per K-step)
//   HN    columns per workgroup tile
//   DEPTH K-steps of loads kept in flight (register ring)
//   BAR   workgroup barrier per K-step (as an LDS-staged GEMM has)
//   LDSW  also write the step to LDS and read it back (staging cost)
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/bw_pattern.hip -o /tmp/bw && /tmp/bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int LPR, int HN, int DEPTH, bool BAR, bool LDSW, int HT>
__global__ __launch_bounds__(HT, 1)
void k_stream(const float* __restrict__ At, float* __restrict__ out, uint32_t K, uint32_t ldq, uint32_t ntiles)
{
    constexpr int RPP = HT / LPR;
    constexpr int NJ = HN / RPP;
    constexpr int KS = LPR * 4;
    constexpr int LD = KS + 4;
    __shared__ __attribute__((aligned(16))) float sQ[LDSW ? 2 : 1][LDSW ? HN : 1][LD];
    const uint32_t tid = threadIdx.x;
    const uint32_t srow = tid / LPR, squad = tid % LPR;
    const uint32_t nk = K / KS;
    v4f sum = { 0.f, 0.f, 0.f, 0.f };
    for (uint32_t bn = blockIdx.x; bn < ntiles; bn += gridDim.x) {
        const float* gQ = At + (size_t)(bn * HN + srow) * ldq + squad * 4;
        v4f r[DEPTH][NJ];
#pragma unroll
        for (int s = 0; s < DEPTH; ++s)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                r[s][j] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(gQ + (size_t)(RPP * j) * ldq + s * KS));
        for (uint32_t kt = 0; kt < nk; kt += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const uint32_t k = kt + u;
                if (LDSW) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        *reinterpret_cast<v4f*>(&sQ[u & 1][srow + RPP * j][squad * 4]) = r[u][j];
                } else {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) sum += r[u][j];
                }
                if (k + DEPTH < nk) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        r[u][j] = __builtin_nontemporal_load(
                            reinterpret_cast<const v4f*>(gQ + (size_t)(RPP * j) * ldq + (k + DEPTH) * KS));
                }
                if (BAR) __syncthreads();
                if (LDSW) {
                    // read back like an MFMA feed: every wave reads 32 rows x the whole K-step
                    const uint32_t lane = tid & 63u, wave = tid >> 6;
                    const uint32_t row = (wave * 32 + (lane & 31u)) % HN;
#pragma unroll
                    for (int g = 0; g < KS / 8; ++g)
                        sum += *reinterpret_cast<const v4f*>(&sQ[u & 1][row][(2 * g + (lane >> 5)) * 4]);
                }
            }
        }
    }
    if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.678f) out[tid] = sum[0];
}

// k_sweep-like: every wave streams whole columns, 1 KiB contiguous per instruction
template <int CPW, int DEPTH, int HT>
__global__ __launch_bounds__(HT, 1)
void k_cols(const float* __restrict__ At, float* __restrict__ out, uint32_t K, uint32_t ldq, uint32_t ncols)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr int WPB = HT / 64;
    v4f sum = { 0.f, 0.f, 0.f, 0.f };
    const uint32_t ngroups = ncols / CPW;
    const uint32_t nk = K / 256;
    for (uint32_t grp = blockIdx.x * WPB + wave; grp < ngroups; grp += gridDim.x * WPB) {
        const float* g = At + (size_t)(grp * CPW) * ldq + lane * 4;
        v4f r[DEPTH][CPW];
#pragma unroll
        for (int s = 0; s < DEPTH; ++s)
#pragma unroll
            for (int c = 0; c < CPW; ++c)
                r[s][c] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(g + (size_t)c * ldq + s * 256));
        for (uint32_t kt = 0; kt < nk; kt += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
#pragma unroll
                for (int c = 0; c < CPW; ++c) sum += r[u][c];
                if (kt + u + DEPTH < nk) {
#pragma unroll
                    for (int c = 0; c < CPW; ++c)
                        r[u][c] = __builtin_nontemporal_load(
                            reinterpret_cast<const v4f*>(g + (size_t)c * ldq + (kt + u + DEPTH) * 256));
                }
            }
        }
    }
    if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.678f) out[threadIdx.x] = sum[0];
}


// the lookahead sweep's structure, feature by feature: RT = R tile (32 RHS rows from L2) staged too,
// MF = 0 nothing, 1 LDS read-back of both operands + VALU, 2 MFMA
typedef float v16f __attribute__((ext_vector_type(16)));
template <int DEPTH, bool RT, int MF>
__global__ __launch_bounds__(512, 1)
void k_g32(const float* __restrict__ At, float* __restrict__ out, uint32_t K, uint32_t ldq, uint32_t ntiles)
{
    constexpr int HN = 256, LD = 36, RPP = 64, NJ = 4, KS = 32;
    __shared__ __attribute__((aligned(16))) float sR[2][32][LD];
    __shared__ __attribute__((aligned(16))) float sQ[2][HN][LD];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t h = lane >> 5, l31 = lane & 31u;
    const uint32_t srow = tid >> 3, squad = tid & 7u;
    const bool has_r = RT && tid < 256;
    const float* gR = At + (size_t)((srow & 31u) * 1000u) * ldq + squad * 4;
    const uint32_t nk = K / KS;
    v16f acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (uint32_t bn = blockIdx.x; bn < ntiles; bn += gridDim.x) {
        const float* gQ = At + (size_t)(bn * HN + srow) * ldq + squad * 4;
        v4f rR[DEPTH], rQ[DEPTH][NJ];
#define LOADS(SET, KT) { if (has_r) rR[SET] = *reinterpret_cast<const v4f*>(gR + (KT) * KS); \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) rQ[SET][j] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(gQ + (size_t)(RPP * j) * ldq + (KT) * KS)); }
#define STORES(SET, BUF) { if (has_r) *reinterpret_cast<v4f*>(&sR[BUF][srow][squad * 4]) = rR[SET]; \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) *reinterpret_cast<v4f*>(&sQ[BUF][srow + RPP * j][squad * 4]) = rQ[SET][j]; }
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) LOADS(s, (uint32_t)s)
        __syncthreads();
        STORES(0, 0)
        if ((uint32_t)DEPTH < nk) LOADS(0, (uint32_t)DEPTH)
        __syncthreads();
        for (uint32_t kt = 0; kt < nk; kt += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const uint32_t k = kt + u;
                const int buf = u & 1;
                if (MF > 0) {
#pragma unroll
                    for (int g = 0; g < KS / 8; ++g) {
                        const uint32_t kq = (2u * g + h) * 4u;
                        v4f a = { 1.f, 1.f, 1.f, 1.f };
                        if (RT) a = *reinterpret_cast<const v4f*>(&sR[buf][l31][kq]);
                        const v4f b = *reinterpret_cast<const v4f*>(&sQ[buf][wave * 32 + l31][kq]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            if (MF == 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
                            else acc[t] += a[t] + b[t];
                        }
                    }
                }
                if (k + 1 < nk) {
                    const int set = (u + 1) % DEPTH;
                    STORES(set, buf ^ 1)
                    if (k + 1 + DEPTH < nk) LOADS(set, k + 1 + DEPTH)
                }
                __syncthreads();
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) t += acc[e];
    if (t == 12345.678f) out[tid] = t;
}


// variant: wave = (column group of 64, k half): 2 accumulators per wave, R operand read half as often
template <int DEPTH>
__global__ __launch_bounds__(512, 1)
void k_g32w(const float* __restrict__ At, float* __restrict__ out, uint32_t K, uint32_t ldq, uint32_t ntiles)
{
    constexpr int HN = 256, LD = 36, RPP = 64, NJ = 4, KS = 32;
    __shared__ __attribute__((aligned(16))) float sR[2][32][LD];
    __shared__ __attribute__((aligned(16))) float sQ[2][HN][LD];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t h = lane >> 5, l31 = lane & 31u;
    const uint32_t cg = wave & 3u, kh = wave >> 2;
    const uint32_t srow = tid >> 3, squad = tid & 7u;
    const bool has_r = tid < 256;
    const float* gR = At + (size_t)((srow & 31u) * 1000u) * ldq + squad * 4;
    const uint32_t nk = K / KS;
    v16f acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    for (uint32_t bn = blockIdx.x; bn < ntiles; bn += gridDim.x) {
        const float* gQ = At + (size_t)(bn * HN + srow) * ldq + squad * 4;
        v4f rR[DEPTH], rQ[DEPTH][NJ];
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) LOADS(s, (uint32_t)s)
        __syncthreads();
        STORES(0, 0)
        if ((uint32_t)DEPTH < nk) LOADS(0, (uint32_t)DEPTH)
        __syncthreads();
        for (uint32_t kt = 0; kt < nk; kt += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const uint32_t k = kt + u;
                const int buf = u & 1;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const uint32_t kq = (2u * (2u * kh + g2) + h) * 4u;
                    const v4f a = *reinterpret_cast<const v4f*>(&sR[buf][l31][kq]);
                    const v4f b0 = *reinterpret_cast<const v4f*>(&sQ[buf][cg * 64 + l31][kq]);
                    const v4f b1 = *reinterpret_cast<const v4f*>(&sQ[buf][cg * 64 + 32 + l31][kq]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b0[t], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b1[t], acc1, 0, 0, 0);
                    }
                }
                if (k + 1 < nk) {
                    const int set = (u + 1) % DEPTH;
                    STORES(set, buf ^ 1)
                    if (k + 1 + DEPTH < nk) LOADS(set, k + 1 + DEPTH)
                }
                __syncthreads();
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) t += acc0[e] + acc1[e];
    if (t == 12345.678f) out[tid] = t;
}

__global__ void k_fill(float* A, size_t nel, int mode)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nel; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32) * 40503u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        A[i] = mode == 0 ? 0.0115f : ((float)(int32_t)x) * (0.02f / 2147483648.f);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename F>
static float time_it(F launch, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); launch();
    hipDeviceSynchronize();
    float best = 1e9f, tot = 0.f;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a, 0);
        launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        tot += ms; if (ms < best) best = ms;
        // idle gap so that sustained-load clock effects do not dominate
        hipDeviceSynchronize();
    }
    hipEventDestroy(a); hipEventDestroy(b);
    printf("  avg %.4f ms  best %.4f ms", tot / reps, best);
    return tot / reps;
}

int main()
{
    int cus = 0;
    const uint32_t m = 8192, n = 65536;
    const size_t bytes = (size_t)m * n * 4;
    float *A, *out;
    const int reps = 15;
    CK(hipMalloc(&A, bytes));
    CK(hipMalloc(&out, 4096 * 4));
    for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, (size_t)m * n, mode);
    CK(hipDeviceSynchronize());
    printf("== data: %s\n", mode ? "random" : "constant");
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    cus = p.multiProcessorCount;
    printf("CUs %d\n", cus);
#define RUN_STREAM(LPR, HN, DEPTH, BAR, LDSW, HT)                                                       \
    {                                                                                                   \
        printf("stream LPR=%2d HN=%3d DEPTH=%d BAR=%d LDS=%d HT=%d:", LPR, HN, DEPTH, BAR, LDSW, HT);   \
        float ms = time_it([&] { hipLaunchKernelGGL((k_stream<LPR, HN, DEPTH, BAR, LDSW, HT>), dim3(cus), dim3(HT), 0, 0, A, out, m, m, n / HN); }, reps); \
        printf("  %.0f GB/s\n", bytes / ms / 1e6);                                                      \
    }
#define RUN_COLS(CPW, DEPTH, HT, GRIDMUL)                                                               \
    {                                                                                                   \
        printf("cols   CPW=%d DEPTH=%d HT=%d grid=%dxCU:", CPW, DEPTH, HT, GRIDMUL);                    \
        float ms = time_it([&] { hipLaunchKernelGGL((k_cols<CPW, DEPTH, HT>), dim3(cus * GRIDMUL), dim3(HT), 0, 0, A, out, m, m, n); }, reps); \
        printf("  %.0f GB/s\n", bytes / ms / 1e6);                                                      \
    }

#define RUN_G32(DEPTH, RT, MF)                                                                          \
    {                                                                                                   \
        printf("g32    DEPTH=%d RT=%d MF=%d:", DEPTH, RT, MF);                                         \
        float ms = time_it([&] { hipLaunchKernelGGL((k_g32<DEPTH, RT, MF>), dim3(cus), dim3(512), 0, 0, A, out, m, m, n / 256); }, reps); \
        printf("  %.0f GB/s\n", bytes / ms / 1e6);                                                      \
    }
    { printf("g32w   DEPTH=4 (64 cols x k-half per wave):");
      float ms = time_it([&] { hipLaunchKernelGGL((k_g32w<4>), dim3(cus), dim3(512), 0, 0, A, out, m, m, n / 256); }, reps);
      printf("  %.0f GB/s\n", bytes / ms / 1e6); }
    RUN_G32(4, true, 2)
    { printf("g32w   DEPTH=4 (64 cols x k-half per wave):");
      float ms = time_it([&] { hipLaunchKernelGGL((k_g32w<4>), dim3(cus), dim3(512), 0, 0, A, out, m, m, n / 256); }, reps);
      printf("  %.0f GB/s\n", bytes / ms / 1e6); }
    RUN_G32(4, false, 0)
    RUN_G32(4, false, 1)
    RUN_G32(4, true, 1)
    RUN_G32(4, false, 2)
    RUN_G32(4, true, 2)
    RUN_G32(8, true, 2)
    RUN_G32(2, true, 2)
    RUN_G32(4, true, 0)
    RUN_COLS(4, 2, 1024, 1)
    }
    const int mode_done = 1; (void)mode_done;
    RUN_COLS(4, 4, 512, 1)
    RUN_COLS(2, 4, 512, 1)
    RUN_STREAM(8, 256, 4, false, false, 512)
    RUN_STREAM(8, 256, 4, true, false, 512)
    RUN_STREAM(8, 256, 4, true, true, 512)
    RUN_STREAM(8, 256, 8, true, true, 512)
    RUN_STREAM(16, 256, 2, false, false, 512)
    RUN_STREAM(16, 256, 4, false, false, 512)
    RUN_STREAM(16, 128, 4, false, false, 512)
    RUN_STREAM(16, 128, 4, true, true, 512)
    RUN_STREAM(32, 64, 4, false, false, 512)
    RUN_STREAM(32, 64, 4, true, true, 512)
    RUN_STREAM(64, 32, 4, false, false, 512)
    RUN_STREAM(64, 32, 8, false, false, 512)
    RUN_STREAM(64, 64, 4, false, false, 512)
    RUN_STREAM(64, 32, 4, true, true, 512)
    RUN_STREAM(64, 64, 4, true, true, 512)
    RUN_STREAM(8, 256, 4, false, false, 1024)
    RUN_STREAM(64, 64, 4, false, false, 1024)
    hipFree(A); hipFree(out);
    return 0;
}
