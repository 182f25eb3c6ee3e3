// utils.hip — the two utilities either side of the solver in the reference's public API, on the device:
//
//   ss::norm_l1(A)            src/linalg/norms.h:22-27, src/lib.cpp:106-112   -> ss_hip_norm_l1_*
//       sums = sum_i |A(i, j)| per column, A(i, j) /= sums[j]  (a zero column becomes NaN, like the
//       reference's 0 / 0)
//
// A is normalised IN PLACE where it lives: a device buffer is reduced and scaled by two kernels; a host
// matrix streams through a 256 MiB staging buffer in row panels (column sums first, then scale + copy back).
// Any element strides are accepted (the layout rules of src/linalg/blas_wrapper.h:63-94 generalised);
// column sums are formed in a fixed order (per 1024-row chunk ascending rows, chunks ascending), so the
// result does not depend on the launch geometry.
#include "ss_hip_internal.h"
#include "ss_hip_device.h"

#include <algorithm>
#include <cstring>
#include <vector>

using namespace sship;

namespace {

constexpr uint32_t kL1Chunk = 1024;          // rows per partial sum

// partial[c][j] = sum over rows [c*kL1Chunk, ...) of |A(i, j)|, one thread per column (coalesced for row-major)
template <typename T>
__global__ __launch_bounds__(256)
void k_l1_partial(const T* __restrict__ A, long long rs, long long cs, uint32_t rows, uint32_t n, T* __restrict__ partial)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    const uint32_t c = blockIdx.y;
    if (j >= n) return;
    const uint32_t i0 = c * kL1Chunk, i1 = min(rows, i0 + kL1Chunk);
    T acc = T(0);
    for (uint32_t i = i0; i < i1; ++i) {
        const T v = A[(long long)i * rs + (long long)j * cs];
        acc += v < T(0) ? -v : v;
    }
    partial[(size_t)c * n + j] = acc;
}

// the same with one workgroup per column (column-contiguous matrices: rows are the unit stride)
template <typename T>
__global__ __launch_bounds__(256)
void k_l1_partial_col(const T* __restrict__ A, long long rs, long long cs, uint32_t rows, uint32_t n, T* __restrict__ partial)
{
    __shared__ T sv[16];
    const uint32_t j = blockIdx.x, c = blockIdx.y;
    const uint32_t i0 = c * kL1Chunk, i1 = min(rows, i0 + kL1Chunk);
    T acc = T(0);
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += 256u) {
        const T v = A[(long long)i * rs + (long long)j * cs];
        acc += v < T(0) ? -v : v;
    }
    const T s = block_sum(acc, sv);
    if (threadIdx.x == 0) partial[(size_t)c * n + j] = s;
}

// sums[j] (+)= partial[0][j] + partial[1][j] + ...   (ascending chunks)
template <typename T>
__global__ __launch_bounds__(256)
void k_l1_reduce(const T* __restrict__ partial, uint32_t chunks, uint32_t n, T* __restrict__ sums, int accumulate)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    T acc = accumulate ? sums[j] : T(0);
    for (uint32_t c = 0; c < chunks; ++c) acc += partial[(size_t)c * n + j];
    sums[j] = acc;
}

// A(i, j) /= sums[j]; `fast_cols`: consecutive threads walk columns (row-major) or rows (column-major)
template <typename T>
__global__ __launch_bounds__(256)
void k_l1_scale(T* __restrict__ A, long long rs, long long cs, uint32_t rows, uint32_t n, const T* __restrict__ sums, int fast_cols)
{
    const size_t total = (size_t)rows * n;
    for (size_t e = (size_t)blockIdx.x * 256u + threadIdx.x; e < total; e += (size_t)gridDim.x * 256u) {
        const size_t i = fast_cols ? e / n : e % rows, j = fast_cols ? e % n : e / rows;
        const long long off = (long long)i * rs + (long long)j * cs;
        A[off] = A[off] / sums[j];
    }
}

struct HipFail2 { hipError_t code; const char* what; };
#define HIPCHK2(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw HipFail2{ e_, #expr }; } while (0)

bool on_device(const void* p)
{
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof(attr));
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeUnified;
}

// column sums of a device-resident strided block of `rows` rows into sums (accumulate: add to what is there)
template <typename T>
void device_col_sums(const T* A, long long rs, long long cs, uint32_t rows, uint32_t n, T* partial, T* sums, int accumulate,
                     hipStream_t st)
{
    const uint32_t chunks = (rows + kL1Chunk - 1) / kL1Chunk;
    const bool col_contig = (rs == 1 && cs != 1);
    if (col_contig) hipLaunchKernelGGL((k_l1_partial_col<T>), dim3(n, chunks), dim3(256), 0, st, A, rs, cs, rows, n, partial);
    else hipLaunchKernelGGL((k_l1_partial<T>), dim3((n + 255u) / 256u, chunks), dim3(256), 0, st, A, rs, cs, rows, n, partial);
    HIPCHK2(hipGetLastError());
    hipLaunchKernelGGL((k_l1_reduce<T>), dim3((n + 255u) / 256u), dim3(256), 0, st, (const T*)partial, chunks, n, sums, accumulate);
    HIPCHK2(hipGetLastError());
}

template <typename T>
void device_scale(T* A, long long rs, long long cs, uint32_t rows, uint32_t n, const T* sums, hipStream_t st)
{
    const bool col_contig = (rs == 1 && cs != 1);
    const size_t total = (size_t)rows * n;
    const uint32_t grid = (uint32_t)std::min<size_t>((total + 255) / 256, (size_t)1 << 20);
    hipLaunchKernelGGL((k_l1_scale<T>), dim3(grid), dim3(256), 0, st, A, rs, cs, rows, n, sums, col_contig ? 0 : 1);
    HIPCHK2(hipGetLastError());
}

template <typename T>
int norm_l1_impl(T* A, size_t m, size_t n, ptrdiff_t rs, ptrdiff_t cs, int device, char* err, size_t errlen)
{
    if (!A || m == 0 || n == 0) { set_err(err, errlen, "norm_l1: A must be a non-empty m x n matrix"); return SS_HIP_EINVAL; }
    if (m > ((size_t)1 << 28) || n > ((size_t)1 << 30) || m > 65535u * (size_t)kL1Chunk) { set_err(err, errlen, "norm_l1: matrix dimensions too large"); return SS_HIP_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); set_err(err, errlen, "norm_l1: no HIP device available"); return SS_HIP_ENODEVICE; }
    if (device < 0 || device >= ndev) { set_err(err, errlen, "norm_l1: device index out of range"); return SS_HIP_EINVAL; }
    T* partial = nullptr; T* sums = nullptr; T* stage = nullptr;
    hipStream_t st = nullptr;
    int rc = SS_HIP_OK;
    try {
        HIPCHK2(hipSetDevice(device));
        HIPCHK2(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIPCHK2(hipMalloc(&sums, n * sizeof(T)));
        if (on_device(A)) {
            const uint32_t chunks = (uint32_t)((m + kL1Chunk - 1) / kL1Chunk);
            HIPCHK2(hipMalloc(&partial, (size_t)chunks * n * sizeof(T)));
            device_col_sums<T>(A, rs, cs, (uint32_t)m, (uint32_t)n, partial, sums, 0, st);
            device_scale<T>(A, rs, cs, (uint32_t)m, (uint32_t)n, sums, st);
            HIPCHK2(hipStreamSynchronize(st));
        } else {
            // host matrix: row panels through a staging buffer (row-major there), two passes
            const size_t panel_bytes = (size_t)256 << 20;
            const size_t R = std::max<size_t>(1, std::min(m, panel_bytes / std::max<size_t>(1, n * sizeof(T))));
            const uint32_t chunks = (uint32_t)((R + kL1Chunk - 1) / kL1Chunk);
            HIPCHK2(hipMalloc(&stage, R * n * sizeof(T)));
            HIPCHK2(hipMalloc(&partial, (size_t)chunks * n * sizeof(T)));
            const bool rowmajor = (cs == 1 || n == 1) && (rs >= (ptrdiff_t)n || m == 1) && rs > 0;
            std::vector<T> gather;
            auto upload = [&](size_t r0, size_t rows) {
                if (rowmajor) {
                    const size_t spitch = (m == 1) ? n * sizeof(T) : (size_t)rs * sizeof(T);
                    HIPCHK2(hipMemcpy2DAsync(stage, n * sizeof(T), A + (ptrdiff_t)r0 * rs, spitch, n * sizeof(T), rows, hipMemcpyHostToDevice, st));
                } else {
                    gather.resize(rows * n);
                    for (size_t i = 0; i < rows; ++i)
                        for (size_t j = 0; j < n; ++j) gather[i * n + j] = A[(ptrdiff_t)(r0 + i) * rs + (ptrdiff_t)j * cs];
                    HIPCHK2(hipMemcpyAsync(stage, gather.data(), rows * n * sizeof(T), hipMemcpyHostToDevice, st));
                    HIPCHK2(hipStreamSynchronize(st));           // `gather` is reused by the next panel
                }
            };
            for (size_t r0 = 0; r0 < m; r0 += R) {
                const size_t rows = std::min(R, m - r0);
                upload(r0, rows);
                device_col_sums<T>(stage, (long long)n, 1, (uint32_t)rows, (uint32_t)n, partial, sums, r0 != 0, st);
                HIPCHK2(hipStreamSynchronize(st));
            }
            for (size_t r0 = 0; r0 < m; r0 += R) {
                const size_t rows = std::min(R, m - r0);
                if (m > R) upload(r0, rows);                     // (a single panel is still there)
                device_scale<T>(stage, (long long)n, 1, (uint32_t)rows, (uint32_t)n, sums, st);
                if (rowmajor) {
                    const size_t dpitch = (m == 1) ? n * sizeof(T) : (size_t)rs * sizeof(T);
                    HIPCHK2(hipMemcpy2DAsync(A + (ptrdiff_t)r0 * rs, dpitch, stage, n * sizeof(T), n * sizeof(T), rows, hipMemcpyDeviceToHost, st));
                    HIPCHK2(hipStreamSynchronize(st));
                } else {
                    gather.resize(rows * n);
                    HIPCHK2(hipMemcpyAsync(gather.data(), stage, rows * n * sizeof(T), hipMemcpyDeviceToHost, st));
                    HIPCHK2(hipStreamSynchronize(st));
                    for (size_t i = 0; i < rows; ++i)
                        for (size_t j = 0; j < n; ++j) A[(ptrdiff_t)(r0 + i) * rs + (ptrdiff_t)j * cs] = gather[i * n + j];
                }
            }
        }
    } catch (const HipFail2& f) {
        set_err(err, errlen, std::string("HIP error: ") + hipGetErrorString(f.code) + " in " + f.what);
        rc = SS_HIP_ERUNTIME;
    } catch (const std::bad_alloc&) {
        set_err(err, errlen, "norm_l1: out of host memory");
        rc = SS_HIP_ENOMEM;
    }
    if (partial) (void)hipFree(partial);
    if (sums) (void)hipFree(sums);
    if (stage) (void)hipFree(stage);
    if (st) (void)hipStreamDestroy(st);
    return rc;
}

}  // namespace

extern "C" {

int ss_hip_norm_l1_f32(float* A, size_t m, size_t n, ptrdiff_t stride_row, ptrdiff_t stride_col, int device, char* err, size_t errlen)
{
    return norm_l1_impl<float>(A, m, n, stride_row, stride_col, device, err, errlen);
}

int ss_hip_norm_l1_f64(double* A, size_t m, size_t n, ptrdiff_t stride_row, ptrdiff_t stride_col, int device, char* err, size_t errlen)
{
    return norm_l1_impl<double>(A, m, n, stride_row, stride_col, device, err, errlen);
}

}  // extern "C"
