// fd_gemm_f32.h -- exact-f32 GEMM for the parity / training path.
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

inline void launch(const Args& g, hipStream_t s) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM), block(NT);
    const bool ak = (g.a_cs == 1), bn = (g.b_cs == 1);
    if (ak && bn) hipLaunchKernelGGL((k_gemm_f32<true, true>), grid, block, 0, s, g);
    else if (ak && !bn) hipLaunchKernelGGL((k_gemm_f32<true, false>), grid, block, 0, s, g);
    else if (!ak && bn) hipLaunchKernelGGL((k_gemm_f32<false, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((k_gemm_f32<false, false>), grid, block, 0, s, g);
}

// y[M,N] = x[M,K] . W[N,K]^T + bias   (nn.Linear forward)
inline void linear_fwd(const float* x, const float* W, const float* bias, float* y, int M, int N, int K, bool relu,
                       hipStream_t s) {
    Args g{x, W, y, bias, M, N, K, (long long)K, 1, 1, (long long)K, (long long)N, 1.0f, relu ? 1 : 0, 0};
    launch(g, s);
}
// dx[M,K] (+)= dy[M,N] . W[N,K]
inline void linear_bwd_input(const float* dy, const float* W, float* dx, int M, int N, int K, bool accumulate,
                             hipStream_t s) {
    Args g{dy, W, dx, nullptr, M, K, N, (long long)N, 1, (long long)K, 1, (long long)K, 1.0f, 0, accumulate ? 1 : 0};
    launch(g, s);
}
// dW[N,K] (+)= dy[M,N]^T . x[M,K]
inline void linear_bwd_weight(const float* dy, const float* x, float* dW, int M, int N, int K, bool accumulate,
                              hipStream_t s) {
    Args g{dy, x, dW, nullptr, N, K, M, 1, (long long)N, (long long)K, 1, (long long)K, 1.0f, 0, accumulate ? 1 : 0};
    launch(g, s);
}

}  // namespace fdgemm
