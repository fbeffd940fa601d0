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

#include "fd_philox.h"

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
    // optional inverted dropout on the output (after bias / relu), fd_k_dropout's stream: element e = m * N + n uses Philox
    // counter drop_offset + e / 4, component e % 4.  Needs N % 4 == 0 and a contiguous C (c_rs == N); MFMA path only.
    float drop_p = 0.f;
    uint64_t drop_seed = 0, drop_offset = 0;
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
// NACC = accumulator tiles per wave = 32-column panels per workgroup: 2 (128x64) in general, 3 (128x96) when that pads N
// less -- the model width D = 72 then is ONE panel, so the big operand of the "-> D" GEMMs (the (M, F) FFN activations and
// their gradients, 52 MB at the training batch) is streamed once instead of once per 64-column panel.
constexpr int MBM = 128, MBK = 16;
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <bool A_KFAST, bool B_NFAST, int NACC>
__global__ __launch_bounds__(NT) void k_gemm_mfma_f32(Args g, int klen, float* __restrict__ partial) {
    constexpr int MBN = 32 * NACC, NB = MBK * MBN / NT;
    __shared__ float As[MBK][MBM + 4];
    __shared__ float Bs[MBK][MBN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * MBM, n0 = blockIdx.x * MBN;
    const int kbeg = blockIdx.z * klen, kend = min(g.K, kbeg + klen);
    f32x16_t acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float ra[8], rb[NB];
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
        for (int i = 0; i < NB; ++i) {
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
        for (int i = 0; i < NB; ++i) {
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
#pragma unroll
            for (int t = 0; t < NACC; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[k][32 * t + (lane & 31)], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
    // C layout: register i -> row 8*(i/4) + 4*(lane>>5) + (i%4), col lane&31
    const bool drop = !partial && g.drop_p > 0.f;
    const float keep_scale = drop ? 1.0f / (1.0f - g.drop_p) : 1.0f;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        const int gn = n0 + 32 * t + (lane & 31);
        if (gn >= g.N) continue;                               // (N % 4 == 0 with dropout: uniform over a quad)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const int gm0 = m0 + wave * 32 + 8 * ib + 4 * (lane >> 5);
            // dropout bits of this lane's 4 rows: the quad's lanes hold the 4 columns of one Philox group, so lane qi
            // evaluates the group of row gm0 + qi and a 4x4 transpose over DPP quad broadcasts hands every lane its own
            // component for each row -- one Philox per 4 outputs, like the stand-alone kernel
            uint32_t rv[4] = {~0u, ~0u, ~0u, ~0u};
            if (drop) {
                const int qi = lane & 3;
                const uint64_t e = (uint64_t)(gm0 + qi) * g.N + (gn & ~3);
                const fd_u4 r = fd_philox4x32_10(g.drop_offset + (e >> 2), g.drop_seed);
#define FD_QUAD_PICK(M_)                                                                                               \
    {                                                                                                                  \
        constexpr int c_ = M_ | (M_ << 2) | (M_ << 4) | (M_ << 6);                                                     \
        const uint32_t a0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)r.x, c_, 0xf, 0xf, true);                          \
        const uint32_t a1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)r.y, c_, 0xf, 0xf, true);                          \
        const uint32_t a2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)r.z, c_, 0xf, 0xf, true);                          \
        const uint32_t a3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)r.w, c_, 0xf, 0xf, true);                          \
        rv[M_] = qi == 0 ? a0 : qi == 1 ? a1 : qi == 2 ? a2 : a3;                                                      \
    }
                FD_QUAD_PICK(0) FD_QUAD_PICK(1) FD_QUAD_PICK(2) FD_QUAD_PICK(3)
#undef FD_QUAD_PICK
            }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = 4 * ib + ii;
                const int gm = gm0 + ii;
                if (gm >= g.M) continue;
                if (partial) {
                    partial[((size_t)blockIdx.z * g.M + gm) * g.N + gn] = acc[t][i];
                } else {
                    float v = g.alpha * acc[t][i];
                    if (g.bias) v += g.bias[gn];
                    if (g.relu) v = fmaxf(v, 0.f);
                    if (drop) v = (fd_u01(rv[ii]) >= g.drop_p) ? v * keep_scale : 0.f;
                    float* c = g.C + gm * g.c_rs + gn;
                    if (g.accumulate) v += *c;
                    *c = v;
                }
            }
        }
    }
}

// 64 outputs per block; the 4 waves each add every 4th partial (independent loads in flight), then wave 0 adds the four
// strands in fixed order: deterministic, and the dependent-add chain is splits/4 long instead of splits (the 72x72 weight
// gradients with 50 splits ran 14 us on 21 blocks)
static __global__ __launch_bounds__(256) void k_splitk_reduce(Args g, const float* __restrict__ partial, int splits) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t mn = (size_t)g.M * g.N;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    float v = 0.f;
    if (i < mn) {
#pragma unroll 4
        for (int z = w; z < splits; z += 4) v += partial[(size_t)z * mn + i];
    }
    red[w][lane] = v;
    __syncthreads();
    if (w != 0 || i >= mn) return;
    v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    const int gm = (int)(i / g.N), gn = (int)(i - (size_t)gm * g.N);
    v *= g.alpha;
    if (g.bias) v += g.bias[gn];
    if (g.relu) v = fmaxf(v, 0.f);
    if (g.drop_p > 0.f) {                                      // (one Philox per output here: these outputs are small)
        const fd_u4 r = fd_philox4x32_10(g.drop_offset + (i >> 2), g.drop_seed);
        const uint32_t rv = (i & 3) == 0 ? r.x : (i & 3) == 1 ? r.y : (i & 3) == 2 ? r.z : r.w;
        v = (fd_u01(rv) >= g.drop_p) ? v * (1.0f / (1.0f - g.drop_p)) : 0.f;
    }
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
    // panel width: whichever of 64 / 96 columns pads N less (ties: the wider one, fewer passes over A) -- unless that
    // leaves an unsplit GEMM with too few workgroups (D = 72 at the training batch: 50 tiles of 128x96 ran 1.3 % slower
    // per optimizer step than 100 of 128x64; at T = 252, 126 tiles, the wide panel is 2.8 % faster)
    const int pad2 = (g.N + 63) / 64 * 64, pad3 = (g.N + 95) / 96 * 96;
    const bool may_split = scratch && g.K >= 512;
    const int tiles3 = (pad3 / 96) * ((g.M + MBM - 1) / MBM);
    const int nacc = (pad3 <= pad2 && (may_split || tiles3 >= 96)) ? 3 : 2;
    const int MBN = 32 * nacc;
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
#define FD_GEMM_GO(AK, BN_, NA) hipLaunchKernelGGL((k_gemm_mfma_f32<AK, BN_, NA>), grid, block, 0, s, g, klen, partial)
    if (nacc == 3) {
        if (ak && bn) FD_GEMM_GO(true, true, 3);
        else if (ak && !bn) FD_GEMM_GO(true, false, 3);
        else if (!ak && bn) FD_GEMM_GO(false, true, 3);
        else FD_GEMM_GO(false, false, 3);
    } else {
        if (ak && bn) FD_GEMM_GO(true, true, 2);
        else if (ak && !bn) FD_GEMM_GO(true, false, 2);
        else if (!ak && bn) FD_GEMM_GO(false, true, 2);
        else FD_GEMM_GO(false, false, 2);
    }
#undef FD_GEMM_GO
    if (splits > 1) {
        const size_t n = (size_t)g.M * g.N;
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, g, partial, splits);
    }
}

// y[M,N] = x[M,K] . W[N,K]^T + bias   (nn.Linear forward)
inline void linear_fwd(const float* x, const float* W, const float* bias, float* y, int M, int N, int K, bool relu,
                       hipStream_t s, float* scratch = nullptr, size_t scratch_floats = 0) {
    Args g{x, W, y, bias, M, N, K, (long long)K, 1, 1, (long long)K, (long long)N, 1.0f, relu ? 1 : 0, 0};
    launch(g, s, scratch, scratch_floats);
}
// true when linear_fwd_dropout can apply the dropout in the GEMM's own epilogue (else: GEMM, then fd_k_dropout)
inline bool can_fuse_dropout(int N) { return N % 4 == 0 && !use_valu_gemm(); }
// y = dropout(x . W^T + bias [relu]) with fd_k_dropout's mask for (seed, offset)
inline void linear_fwd_dropout(const float* x, const float* W, const float* bias, float* y, int M, int N, int K, bool relu,
                               float p, uint64_t seed, uint64_t offset, hipStream_t s, float* scratch = nullptr,
                               size_t scratch_floats = 0) {
    Args g{x, W, y, bias, M, N, K, (long long)K, 1, 1, (long long)K, (long long)N, 1.0f, relu ? 1 : 0, 0};
    g.drop_p = p;
    g.drop_seed = seed;
    g.drop_offset = offset;
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
