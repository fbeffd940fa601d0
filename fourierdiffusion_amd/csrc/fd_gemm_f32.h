// fd_gemm_f32.h -- exact-f32 GEMMs for the parity / training path: a plain VALU kernel (FDIFF_GEMM=valu) and the
// fp32-MFMA kernel used by default (same arithmetic class, ~30x faster on the training shapes).
//
//   C[m,n] = alpha * sum_k A(m,k) * B(k,n)  (+ bias[n]) (relu) (+ C[m,n] if accumulate)
//
// A(m,k) = A[m*a_rs + k*a_cs], B(k,n) = B[k*b_rs + n*b_cs]: every transposition the forward and
// backward passes need is a choice of strides (x.W^T, dY.W, dY^T.X).  64x64x16 LDS tiles, 256
// threads, 4x4 register tile per thread, plain v_fma_f32 accumulation in k order (bitwise the same
// rounding class as the reference's fp32 CPU GEMMs).  This is the correctness anchor, not the
// fast path: the bf16 MFMA kernels in fd_score_bf16.hip carry the throughput.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

namespace fdgemm {

constexpr int BM = 64, BN = 64, BK = 16, NT = 256;

struct Args {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   // per n, nullable
    int M, N, K;
    long long a_rs, a_cs, b_rs, b_cs, c_rs;
    float alpha;
    int relu;
    int accumulate;
};

// A_KFAST: a_cs == 1 (k contiguous) -> load with k fastest across threads; else m fastest.
// B_NFAST: b_cs == 1 (n contiguous) -> n fastest; else k fastest.
template <bool A_KFAST, bool B_NFAST>
__global__ __launch_bounds__(NT) void k_gemm_f32(Args g) {
    __shared__ float As[BK][BM + 4];
    __shared__ float Bs[BK][BN + 4];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tm = (tid / 16) * 4, tn = (tid % 16) * 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int k0 = 0; k0 < g.K; k0 += BK) {
        // stage A tile (BM x BK) and B tile (BK x BN): 1024 elements each, 4 per thread
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * NT;
            int m, k;
            if (A_KFAST) { k = id % BK; m = id / BK; } else { m = id % BM; k = id / BM; }
            const int gm = m0 + m, gk = k0 + k;
            As[k][m] = (gm < g.M && gk < g.K) ? g.A[gm * g.a_rs + gk * g.a_cs] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * NT;
            int n, k;
            if (B_NFAST) { n = id % BN; k = id / BN; } else { k = id % BK; n = id / BK; }
            const int gn = n0 + n, gk = k0 + k;
            Bs[k][n] = (gn < g.N && gk < g.K) ? g.B[gk * g.b_rs + gn * g.b_cs] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BK; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(&As[k][tm]);
            const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tn]);
            const float av[4] = {a.x, a.y, a.z, a.w};
            const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + tm + i;
        if (gm >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tn + j;
            if (gn >= g.N) continue;
            float v = g.alpha * acc[i][j];
            if (g.bias) v += g.bias[gn];
            if (g.relu) v = fmaxf(v, 0.f);
            float* c = g.C + gm * g.c_rs + gn;
            if (g.accumulate) v += *c;
            *c = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// fp32 MFMA variant (v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulate -- the same rounding class as the VALU
// kernel, only the summation order differs).  128x64x16 LDS tiles, 4 waves, each wave a 32x64 strip = two 32x32
// accumulator tiles sharing the A operand; next tile's global loads are in flight while the current one is
// multiplied.  grid.z splits K: long reductions with small outputs (dW = dY^T.X, K = all tokens) would otherwise
// run on a handful of workgroups; partial sums go to `partial[z][M][N]` and are added in fixed z order by
// k_splitk_reduce (deterministic, unlike atomics).
constexpr int MBM = 128, MBN = 64, MBK = 16;
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <bool A_KFAST, bool B_NFAST>
__global__ __launch_bounds__(NT) void k_gemm_mfma_f32(Args g, int klen, float* __restrict__ partial) {
    __shared__ float As[MBK][MBM + 4];
    __shared__ float Bs[MBK][MBN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * MBM, n0 = blockIdx.x * MBN;
    const int kbeg = blockIdx.z * klen, kend = min(g.K, kbeg + klen);
    f32x16_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float ra[8], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = tid + i * NT;
            int m, k;
            if (A_KFAST) { k = id % MBK; m = id / MBK; } else { m = id % MBM; k = id / MBM; }
            const int gm = m0 + m, gk = k0 + k;
            ra[i] = (gm < g.M && gk < kend) ? g.A[gm * g.a_rs + gk * g.a_cs] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * NT;
            int n, k;
            if (B_NFAST) { n = id % MBN; k = id / MBN; } else { k = id % MBK; n = id / MBK; }
            const int gn = n0 + n, gk = k0 + k;
            rb[i] = (gn < g.N && gk < kend) ? g.B[gk * g.b_rs + gn * g.b_cs] : 0.f;
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = tid + i * NT;
            int m, k;
            if (A_KFAST) { k = id % MBK; m = id / MBK; } else { m = id % MBM; k = id / MBM; }
            As[k][m] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * NT;
            int n, k;
            if (B_NFAST) { n = id % MBN; k = id / MBN; } else { k = id % MBK; n = id / MBK; }
            Bs[k][n] = rb[i];
        }
    };
    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += MBK) {
        sstore();
        __syncthreads();
        if (k0 + MBK < kend) gload(k0 + MBK);
        // operand layout of 32x32x2: lane l holds row/col l&31 of k-slot l>>5
#pragma unroll
        for (int kk = 0; kk < MBK / 2; ++kk) {
            const int k = 2 * kk + (lane >> 5);
            const float a = As[k][wave * 32 + (lane & 31)];
            const float b0 = Bs[k][lane & 31], b1 = Bs[k][32 + (lane & 31)];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        }
        __syncthreads();
    }
    // C layout: register i -> row 8*(i/4) + 4*(lane>>5) + (i%4), col lane&31
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int gn = n0 + 32 * t + (lane & 31);
        if (gn >= g.N) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int gm = m0 + wave * 32 + 8 * (i / 4) + 4 * (lane >> 5) + (i % 4);
            if (gm >= g.M) continue;
            if (partial) {
                partial[((size_t)blockIdx.z * g.M + gm) * g.N + gn] = acc[t][i];
            } else {
                float v = g.alpha * acc[t][i];
                if (g.bias) v += g.bias[gn];
                if (g.relu) v = fmaxf(v, 0.f);
                float* c = g.C + gm * g.c_rs + gn;
                if (g.accumulate) v += *c;
                *c = v;
            }
        }
    }
}

static __global__ __launch_bounds__(256) void k_splitk_reduce(Args g, const float* __restrict__ partial, int splits) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)g.M * g.N) return;
    const int gm = (int)(i / g.N), gn = (int)(i - (size_t)gm * g.N);
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += partial[(size_t)z * g.M * g.N + i];
    v *= g.alpha;
    if (g.bias) v += g.bias[gn];
    if (g.relu) v = fmaxf(v, 0.f);
    float* c = g.C + gm * g.c_rs + gn;
    if (g.accumulate) v += *c;
    *c = v;
}

inline bool use_valu_gemm() {
    static const bool v = [] { const char* e = getenv("FDIFF_GEMM"); return e && e[0] == 'v'; }();
    return v;
}

// scratch (optional, caller-owned device memory, `scratch_floats` long) enables split-K for long reductions
inline void launch(const Args& g, hipStream_t s, float* scratch = nullptr, size_t scratch_floats = 0) {
    const bool ak = (g.a_cs == 1), bn = (g.b_cs == 1);
    if (use_valu_gemm()) {
        dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM), block(NT);
        if (ak && bn) hipLaunchKernelGGL((k_gemm_f32<true, true>), grid, block, 0, s, g);
        else if (ak && !bn) hipLaunchKernelGGL((k_gemm_f32<true, false>), grid, block, 0, s, g);
        else if (!ak && bn) hipLaunchKernelGGL((k_gemm_f32<false, true>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((k_gemm_f32<false, false>), grid, block, 0, s, g);
        return;
    }
    const int tiles = ((g.N + MBN - 1) / MBN) * ((g.M + MBM - 1) / MBM);
    int splits = 1;
    if (scratch && tiles < 512 && g.K >= 512) {
        splits = (1024 + tiles - 1) / tiles;
        splits = std::min(splits, g.K / 128);
        splits = (int)std::min<size_t>((size_t)splits, scratch_floats / ((size_t)g.M * g.N));
        splits = std::max(splits, 1);
    }
    int klen = (g.K + splits - 1) / splits;
    klen = (klen + MBK - 1) / MBK * MBK;
    splits = (g.K + klen - 1) / klen;
    float* partial = splits > 1 ? scratch : nullptr;
    dim3 grid((g.N + MBN - 1) / MBN, (g.M + MBM - 1) / MBM, splits), block(NT);
    if (ak && bn) hipLaunchKernelGGL((k_gemm_mfma_f32<true, true>), grid, block, 0, s, g, klen, partial);
    else if (ak && !bn) hipLaunchKernelGGL((k_gemm_mfma_f32<true, false>), grid, block, 0, s, g, klen, partial);
    else if (!ak && bn) hipLaunchKernelGGL((k_gemm_mfma_f32<false, true>), grid, block, 0, s, g, klen, partial);
    else hipLaunchKernelGGL((k_gemm_mfma_f32<false, false>), grid, block, 0, s, g, klen, partial);
    if (splits > 1) {
        const size_t n = (size_t)g.M * g.N;
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, partial, splits);
    }
}

// y[M,N] = x[M,K] . W[N,K]^T + bias   (nn.Linear forward)
inline void linear_fwd(const float* x, const float* W, const float* bias, float* y, int M, int N, int K, bool relu,
                       hipStream_t s, float* scratch = nullptr, size_t scratch_floats = 0) {
    Args g{x, W, y, bias, M, N, K, (long long)K, 1, 1, (long long)K, (long long)N, 1.0f, relu ? 1 : 0, 0};
    launch(g, s, scratch, scratch_floats);
}
// dx[M,K] (+)= dy[M,N] . W[N,K]
inline void linear_bwd_input(const float* dy, const float* W, float* dx, int M, int N, int K, bool accumulate,
                             hipStream_t s, float* scratch = nullptr, size_t scratch_floats = 0) {
    Args g{dy, W, dx, nullptr, M, K, N, (long long)N, 1, (long long)K, 1, (long long)K, 1.0f, 0, accumulate ? 1 : 0};
    launch(g, s, scratch, scratch_floats);
}
// dW[N,K] (+)= dy[M,N]^T . x[M,K]
inline void linear_bwd_weight(const float* dy, const float* x, float* dW, int M, int N, int K, bool accumulate,
                              hipStream_t s, float* scratch = nullptr, size_t scratch_floats = 0) {
    Args g{dy, x, dW, nullptr, N, K, M, 1, (long long)N, (long long)K, 1, (long long)K, 1.0f, 0, accumulate ? 1 : 0};
    launch(g, s, scratch, scratch_floats);
}

}  // namespace fdgemm
