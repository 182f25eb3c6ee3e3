// Measurement aid (not part of the product): how fast can ONE workgroup run the arithmetic skeleton of a
// Homotopy iteration in Gram form on 256 columns with everything in LDS?
//   2 Gram passes  (acc_i = sum_j coef_j * G[row_j][i], sequential in j, K rows)      variants A/B/C
//   2 block reductions of (value, index) pairs
//   2 K x K matrix-vector products + 1 rank-1 style update of the K x K matrix
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/iter_floor.hip -o /tmp/itf && /tmp/itf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int T = 512, W = 256, KMAX = 96;

__device__ __forceinline__ float wsum(float v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// variant 0: row index per list entry fetched one per lane, broadcast with readlane (as k_la_persist does)
// variant 1: row byte offsets and coefficients read as uniform LDS values (ds_read broadcast), 4 rows per step
// variant 2: like 1, but the 256 columns are spread over all 8 waves as 32 columns x 2 row-halves per wave?  (not order preserving: skipped)
template <int VAR>
__device__ __forceinline__ float gram_pass(const float* G, const uint32_t* rows, const float* coef, int K, int tcol, int lane)
{
    float acc = 0.f;
    if (VAR == 0) {
        const int K16 = (K + 15) & ~15;
        for (int j0 = 0; j0 < K16; j0 += 64) {
            const int jl = j0 + lane;
            const uint32_t vl = rows[jl < K ? jl : 0];
            const int cnt = K16 - j0 < 64 ? K16 - j0 : 64;
            for (int u = 0; u < cnt; u += 16) {
                float gv[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) gv[t] = G[__builtin_amdgcn_readlane(vl, u + t) * W + tcol];
#pragma unroll
                for (int t = 0; t < 16; ++t) acc += coef[j0 + u + t] * gv[t];
            }
        }
    } else {
        const int K8 = (K + 7) & ~7;
        for (int j = 0; j < K8; j += 8) {
            float gv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) gv[t] = G[rows[j + t] * W + tcol];      // rows[] uniform: scalar-ish LDS broadcast
#pragma unroll
            for (int t = 0; t < 8; ++t) acc += coef[j + t] * gv[t];
        }
    }
    return acc;
}

template <int VAR, int NWAVES_MV>
__global__ __launch_bounds__(T, 1)
void k_iter(int iters, int K, float* out, uint64_t* ts)
{
    extern __shared__ float smem[];
    float* G = smem;                          // [KMAX][W]
    float* I = G + KMAX * W;                  // [KMAX][KMAX+1]
    float* xs = I + KMAX * (KMAX + 1);        // [KMAX + 16]
    float* ds = xs + KMAX + 16;
    float* u1 = ds + KMAX + 16;
    float* u2 = u1 + KMAX + 16;
    uint32_t* rows = reinterpret_cast<uint32_t*>(u2 + KMAX + 16);   // [KMAX + 16]
    __shared__ float sv[16];
    __shared__ uint32_t si[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < KMAX * W; e += T) G[e] = 1e-3f * (float)((e * 2654435761u) >> 20);
    for (int e = tid; e < KMAX * (KMAX + 1); e += T) I[e] = (e % (KMAX + 2) == 0) ? 1.f : 1e-3f;
    if (tid < KMAX + 16) { xs[tid] = tid < K ? 0.5f : 0.f; ds[tid] = tid < K ? 0.25f : 0.f; u1[tid] = 0.1f; u2[tid] = 0.f; rows[tid] = (uint32_t)((tid * 7) % KMAX); }
    __syncthreads();
    const int tcol = tid & (W - 1);
    float total = 0.f;
    const uint64_t t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        // q pass, c pass (waves 0..3 own the 256 columns)
        float q = 0.f, c = 0.f;
        if (tid < W) {
            q = gram_pass<VAR>(G, rows, ds, K, tcol, lane);
            c = 1.f - gram_pass<VAR>(G, rows, xs, K, tcol, lane);
        }
        // two (value, index) block reductions
        for (int r = 0; r < 2; ++r) {
            float v = r ? c : (1.f - c) / (1.f - q + 2.f);
            uint32_t ix = tid;
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(v, o);
                const uint32_t oi = __shfl_xor(ix, o);
                if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
            }
            __syncthreads();
            if (lane == 0) { sv[wave] = v; si[wave] = ix; }
            __syncthreads();
            v = sv[0]; ix = si[0];
            for (int w2 = 1; w2 < T / 64; ++w2) if (sv[w2] < v) { v = sv[w2]; ix = si[w2]; }
            total += v + (float)ix * 1e-9f;
        }
        // u2 = I u1, one wave per row
        for (int i = wave; i < K; i += NWAVES_MV) {
            if (wave >= NWAVES_MV) break;
            float a = 0.f;
            for (int j = lane; j < K; j += 64) a += I[i * (KMAX + 1) + j] * u1[j];
            a = wsum(a);
            if (lane == 0) u2[i] = a;
        }
        __syncthreads();
        // I += d u2 u2^T (all threads)
        for (int e = tid; e < K * K; e += T) {
            const int a = e / K, b = e - a * K;
            I[a * (KMAX + 1) + b] += (1e-6f * u2[a]) * u2[b];
        }
        __syncthreads();
        // d = I s
        for (int i = wave; i < K; i += NWAVES_MV) {
            if (wave >= NWAVES_MV) break;
            float a = 0.f;
            for (int j = lane; j < K; j += 64) a += I[i * (KMAX + 1) + j] * xs[j];
            a = wsum(a);
            if (lane == 0) ds[i] = a * 1e-3f;
        }
        __syncthreads();
    }
    const uint64_t t1 = wall_clock64();
    if (tid == 0) { ts[0] = t1 - t0; out[0] = total; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main()
{
    float* out; uint64_t* ts;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&ts, 64));
    const size_t lds = (size_t)(KMAX * W + KMAX * (KMAX + 1) + 5 * (KMAX + 16)) * 4;
    const int iters = 200;
#define RUN(VAR, NW, K)                                                                                   \
    {                                                                                                     \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_iter<VAR, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_iter<VAR, NW>), dim3(1), dim3(T), lds, 0, iters, K, out, ts); \
        CK(hipDeviceSynchronize());                                                                       \
        uint64_t h; CK(hipMemcpy(&h, ts, 8, hipMemcpyDeviceToHost));                                      \
        printf("gram variant %d, matvec waves %d, K=%2d: %.2f us per iteration\n", VAR, NW, K, (double)h / 100.0 / iters); \
    }
    RUN(0, 8, 32) RUN(0, 8, 64) RUN(1, 8, 32) RUN(1, 8, 64) RUN(1, 4, 64)
    return 0;
}
