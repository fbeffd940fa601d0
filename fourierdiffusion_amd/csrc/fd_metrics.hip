// fd_metrics.hip -- sliced / marginal Wasserstein-2 evaluation metrics on the GPU.
//
// Reference: src/fdiff/sampling/metrics.py:100-217 + src/fdiff/utils/wasserstein.py:95-199.  Per direction the reference
// projects both sample sets (data @ direction), and calls POT's emd2_1d (uniform weights, squared Euclidean cost), i.e. the
// exact 1-D optimal transport = the L2 distance of the two quantile functions.  Here:
//   fd_project_rows    P[k, i] = <dir_k, x_i>            one fp32-MFMA GEMM for all directions          (wasserstein.py:150-153)
//   fd_transpose_rows  P[f, i] = x_i[f]                  marginals = the standard basis directions      (wasserstein.py:77-89)
//   fd_sort_rows       each row of P sorted ascending    rocPRIM segmented radix sort (the one library call of the engine:
//                                                        a row is up to the size of the training set, beyond one workgroup's LDS)
//   fd_w2_sorted_rows  W2 per row between a sorted row of n and a sorted row of m values, any n, m: on the common grid of
//                      n*m cells element i of the larger set meets at most two elements of the smaller one; the overlaps are
//                      exact integers, the sum runs in double                                            (wasserstein.py:112-113, 139-141)
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "fd_common.h"
#include "fd_gemm_f32.h"

namespace {

struct RowOffset {
    int n;
    __host__ __device__ int operator()(int i) const { return i * n; }
};

__global__ __launch_bounds__(256) void k_transpose_rows(const float* __restrict__ x, float* __restrict__ out, int n, int d) {
    __shared__ float tile[32][33];
    const int i0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, f = f0 + tx;
        tile[r][tx] = (i < n && f < d) ? x[(size_t)i * d + f] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int f = f0 + r, i = i0 + tx;
        if (f < d && i < n) out[(size_t)f * n + i] = tile[tx][r];
    }
}

// one workgroup per row pair
__global__ __launch_bounds__(256) void k_w2_sorted_rows(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, int n, int m) {
    __shared__ double red[4];
    const int row = blockIdx.x;
    const float* big = (n >= m) ? a + (size_t)row * n : b + (size_t)row * m;
    const float* small = (n >= m) ? b + (size_t)row * m : a + (size_t)row * n;
    const long long N = (n >= m) ? n : m, M = (n >= m) ? m : n;
    double acc = 0.0;
    for (long long i = threadIdx.x; i < N; i += 256) {
        const long long lo = i * M, hi = lo + M;               // cells [lo, hi) of the N*M grid; small element j covers [j*N, (j+1)*N)
        const long long j0 = lo / N, j1 = (hi - 1) / N;
        const double v = (double)big[i];
        for (long long j = j0; j <= j1; ++j) {
            const long long s = max(lo, j * N), e = min(hi, (j + 1) * N);
            const double dlt = v - (double)small[j];
            acc += (double)(e - s) * dlt * dlt;
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = ((red[0] + red[1]) + (red[2] + red[3])) / ((double)N * (double)M);
        out[row] = (float)sqrt(tot);
    }
}

}  // namespace

extern "C" int fd_project_rows(fd_ctx* ctx, const float* x, const float* dirs, float* out, int n, int d, int K, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, x && dirs && out, "fd_project_rows: null pointer");
    FD_REQUIRE(ctx, n > 0 && d > 0 && K > 0, "fd_project_rows: bad shape n=%d d=%d K=%d", n, d, K);
    FD_REQUIRE(ctx, (long long)K * n < 2147483647LL && (long long)n * d < 2147483647LL, "fd_project_rows: too large");
    fdgemm::linear_fwd(dirs, x, nullptr, out, K, n, d, false, (hipStream_t)stream);   // out (K, n) = dirs (K, d) . x (n, d)^T
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_transpose_rows(fd_ctx* ctx, const float* x, float* out, int n, int d, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, x && out && x != out, "fd_transpose_rows: null or aliased pointer");
    FD_REQUIRE(ctx, n > 0 && d > 0, "fd_transpose_rows: bad shape n=%d d=%d", n, d);
    FD_REQUIRE(ctx, (d + 31) / 32 <= 65535, "fd_transpose_rows: d=%d too large", d);
    hipLaunchKernelGGL(k_transpose_rows, dim3((n + 31) / 32, (d + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, out, n, d);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_sort_rows_temp_bytes(fd_ctx* ctx, int K, int n, size_t* bytes) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, bytes && K > 0 && n > 0 && (long long)K * n < 2147483647LL, "fd_sort_rows_temp_bytes: bad arguments");
    auto offs = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), RowOffset{n});
    size_t need = 0;
    const hipError_t e = rocprim::segmented_radix_sort_keys(nullptr, need, (const float*)nullptr, (float*)nullptr,
                                                            (unsigned)((size_t)K * n), (unsigned)K, offs, offs + 1, 0, 32,
                                                            (hipStream_t)0);
    if (e != hipSuccess) return fd_fail(ctx, FD_ERR_HIP, "fd_sort_rows_temp_bytes: %s", hipGetErrorString(e));
    *bytes = need < 256 ? 256 : need;
    return FD_OK;
}

extern "C" int fd_sort_rows(fd_ctx* ctx, const float* in, float* out, int K, int n, void* temp, size_t temp_bytes, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, in && out && temp && in != out, "fd_sort_rows: null or aliased pointer");
    FD_REQUIRE(ctx, K > 0 && n > 0 && (long long)K * n < 2147483647LL, "fd_sort_rows: bad shape K=%d n=%d", K, n);
    auto offs = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), RowOffset{n});
    const hipError_t e = rocprim::segmented_radix_sort_keys(temp, temp_bytes, in, out, (unsigned)((size_t)K * n), (unsigned)K, offs,
                                                            offs + 1, 0, 32, (hipStream_t)stream);
    if (e != hipSuccess) return fd_fail(ctx, FD_ERR_HIP, "fd_sort_rows: %s", hipGetErrorString(e));
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_w2_sorted_rows(fd_ctx* ctx, const float* a, const float* b, float* out, int K, int n, int m, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, a && b && out, "fd_w2_sorted_rows: null pointer");
    FD_REQUIRE(ctx, K > 0 && n > 0 && m > 0, "fd_w2_sorted_rows: bad shape K=%d n=%d m=%d", K, n, m);
    hipLaunchKernelGGL(k_w2_sorted_rows, dim3(K), dim3(256), 0, (hipStream_t)stream, a, b, out, n, m);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}
