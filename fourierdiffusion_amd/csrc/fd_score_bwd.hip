// fd_score_bwd.hip -- backward pass of the score network (exact-f32 training path).
//
// The reference trains through torch autograd (Lightning `training_step`, src/fdiff/models/score_models.py:96-108);
// here the gradient of every op of SURVEY.md A.3 is written out: GEMM-shaped pieces reuse fd_gemm_f32.h with
// transposed strides, the rest (LayerNorm, attention, dropout/relu masks, bias/positional/time-embedding reductions)
// are the kernels below.  Activations come from the ctx workspace filled by fd_score_forward_train; dropout masks
// are regenerated from the same Philox counters (never stored).  Gradients accumulate into ONE flat fp32 buffer
// with the parameter layout (so the data-parallel exchange is one all-reduce).
#include "fd_gemm_f32.h"
#include "fd_philox.h"
#include "fd_score.h"

void fd_dropout_inplace(fd_ctx* ctx, float* x, size_t n, float p, uint64_t seed, uint64_t offset, hipStream_t s);
uint64_t fd_dropout_site_offset(uint64_t base, int layer, int site);

namespace {

// Column sums without atomics (two runs must give bit-identical gradients): stage 1 writes one partial row per block
// (rows m0 .. m0+rows_per_block-1, 4 row strands added in a fixed order), stage 2 (k_sum_rows) adds the blocks' rows in
// ascending order.
__global__ __launch_bounds__(256) void k_colsum_part(const float* __restrict__ x, float* __restrict__ part, int M, int N,
                                                      int rows_per_block) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;                 // 4 row-strands per block
    const int m0 = blockIdx.y * rows_per_block;
    const int m1 = min(M, m0 + rows_per_block);
    float acc = 0.f;
    if (n < N)
        for (int m = m0 + sub; m < m1; m += 4) acc += x[(size_t)m * N + n];
    __shared__ float red[4][64];
    red[sub][threadIdx.x & 63] = acc;
    __syncthreads();
    if (sub == 0 && n < N)
        part[(size_t)blockIdx.y * N + n] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// 64 columns x 4 row strands per block; a strand adds rows strand, strand + 4, ... in ascending order with eight loads in
// flight, the strands are combined in a fixed order (one thread per column walking R rows was a chain of R dependent round
// trips on 72 threads: 14-32 us for the 126 partial rows of a 16 128-token batch).
__device__ __forceinline__ float sum_rows_strand(const float* __restrict__ p, int R, size_t stride, int sub) {
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    int r = sub;
    for (; r + 28 < R; r += 32) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += p[(size_t)(r + 4 * k) * stride];
    }
    for (; r < R; r += 4) a[0] += p[(size_t)r * stride];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
__global__ __launch_bounds__(256) void k_sum_rows(const float* __restrict__ part, int R, int N, float* __restrict__ out, int accumulate) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + col;
    red[sub][col] = (n < N) ? sum_rows_strand(part + n, R, (size_t)N, sub) : 0.f;
    __syncthreads();
    if (sub == 0 && n < N) {
        const float v = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        out[n] = accumulate ? out[n] + v : v;
    }
}

// out[n] += sum_r part[r * stride + col0 + n], n < N
__global__ __launch_bounds__(256) void k_sum_rows_strided(const float* __restrict__ part, int R, int stride, int col0, int N,
                                                           float* __restrict__ out) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + col;
    red[sub][col] = (n < N) ? sum_rows_strand(part + col0 + n, R, (size_t)stride, sub) : 0.f;
    __syncthreads();
    if (sub == 0 && n < N) out[n] += (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

// out = dropout(in) with the forward's mask (same Philox stream as fd_k_dropout: 4 elements per counter)
__global__ __launch_bounds__(256) void k_dropout_copy(const float* __restrict__ in, float* __restrict__ out, size_t n, float p,
                                                       uint64_t seed, uint64_t offset) {
    const size_t ng = (n + 3) / 4;
    const float sc = (p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;
    for (size_t g = blockIdx.x * (size_t)256 + threadIdx.x; g < ng; g += (size_t)gridDim.x * 256) {
        uint32_t rv[4] = {~0u, ~0u, ~0u, ~0u};
        if (p > 0.f) {
            const fd_u4 r = fd_philox4x32_10(offset + g, seed);
            rv[0] = r.x; rv[1] = r.y; rv[2] = r.z; rv[3] = r.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t e = g * 4 + i;
            if (e < n) out[e] = (p <= 0.f || fd_u01(rv[i]) >= p) ? in[e] * sc : 0.f;
        }
    }
}

// LayerNorm backward.  y = (x - mean) * rstd * gamma + beta, x = pre-norm sum saved by the forward.
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
//   dgamma += sum_tokens dy * xhat ; dbeta += sum_tokens dy
// One wave per token, 4 tokens per block pass; per-block partial parameter gradients go through LDS.
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                                 const float* __restrict__ mr, const float* __restrict__ gamma,
                                                 float* __restrict__ dx, float* __restrict__ part, int M, int D,
                                                 int tokens_per_block) {
    extern __shared__ float sh[];          // [4 waves][2][D] partial dgamma, dbeta of the waves
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m0 = blockIdx.x * tokens_per_block;
    const int m1 = min(M, m0 + tokens_per_block);
    float pg[16], pb[16];                  // this lane's columns (d = lane + 64 n, D <= 1024), summed over the wave's tokens
#pragma unroll
    for (int n = 0; n < 16; ++n) { pg[n] = 0.f; pb[n] = 0.f; }
    for (int m = m0 + w; m < m1; m += 4) {
        const float mean = mr[(size_t)m * 2], rstd = mr[(size_t)m * 2 + 1];
        float g[16], xh[16];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int d = lane + 64 * n;
            if (d < D) {
                const float dyv = dy[(size_t)m * D + d];
                xh[n] = (x[(size_t)m * D + d] - mean) * rstd;
                g[n] = dyv * gamma[d];
                sg += g[n];
                sgx += g[n] * xh[n];
                pg[n] = fmaf(dyv, xh[n], pg[n]);
                pb[n] += dyv;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            sg += __shfl_xor(sg, o);
            sgx += __shfl_xor(sgx, o);
        }
        const float mg = sg / (float)D, mgx = sgx / (float)D;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int d = lane + 64 * n;
            if (d < D) dx[(size_t)m * D + d] = rstd * (g[n] - mg - xh[n] * mgx);
        }
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const int d = lane + 64 * n;
        if (d < D) {
            sh[(w * 2 + 0) * D + d] = pg[n];
            sh[(w * 2 + 1) * D + d] = pb[n];
        }
    }
    __syncthreads();
    // block partial row [dgamma | dbeta], waves added in a fixed order; the blocks' rows are added by k_sum_rows
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const int which = i / D, d = i - which * D;
        part[(size_t)blockIdx.x * 2 * D + i] = (sh[(0 * 2 + which) * D + d] + sh[(1 * 2 + which) * D + d]) +
                                               (sh[(2 * 2 + which) * D + d] + sh[(3 * 2 + which) * D + d]);
    }
}

// d(relu + inverted dropout): g *= (act != 0) / (1 - p)  (act = drop(relu(.)): nonzero <=> kept and positive)
__global__ __launch_bounds__(256) void k_relu_drop_bwd(float* __restrict__ g, const float* __restrict__ act, size_t n,
                                                        float inv_keep) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        g[i] = (act[i] != 0.f) ? g[i] * inv_keep : 0.f;
}

__global__ __launch_bounds__(256) void k_add_inplace(float* __restrict__ a, const float* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] += b[i];
}

// Dq[b,h,q] = sum_d dO[q, h, d] * O[q, h, d]   (row term of the softmax Jacobian).  Kept as its own 6 us launch: folded
// into the two attention-backward kernels (strided 8-load chains in their prologues) it cost 11 us.
__global__ __launch_bounds__(256) void k_attn_rowdot(const float* __restrict__ dO, const float* __restrict__ O,
                                                      float* __restrict__ Dq, int B, int T, int H, int hd) {
    const size_t id = blockIdx.x * (size_t)256 + threadIdx.x;
    if (id >= (size_t)B * H * T) return;
    const int q = (int)(id % T), h = (int)((id / T) % H), b = (int)(id / ((size_t)T * H));
    const size_t base = ((size_t)b * T + q) * (H * hd) + h * hd;
    float a = 0.f;
    for (int d = 0; d < hd; ++d) a = fmaf(dO[base + d], O[base + d], a);
    Dq[id] = a;
}

// Attention-probability dropout mask, same stream as the forward (fd_score_f32.hip k_attention_f32): counter
// offset + ((b*H+h)*T + q) * ceil(T/4) + key/4, component key%4.  One Philox4x32-10 costs ~900 cycles (40 quarter-rate
// integer multiplies), so both backward kernels evaluate one per FOUR scores instead of one per score.
__device__ __forceinline__ fd_u4 drop_bits(uint64_t seed, uint64_t offset, int bh, int T, int q, int key_group) {
    const int groups_per_row = (T + 3) / 4;
    return fd_philox4x32_10(offset + (((uint64_t)bh * T + q) * groups_per_row) + key_group, seed);
}
__device__ __forceinline__ float keep_of(uint32_t rv, float drop_p, float keep_scale) {
    return (fd_u01(rv) >= drop_p) ? keep_scale : 0.f;
}
// value of `v` held by lane M of this lane's quad (DPP quad_perm broadcast)
template <int M>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, M | (M << 2) | (M << 4) | (M << 6), 0xf, 0xf, true);
}

// dq: lane per query, keys/values streamed through LDS.  dS = P * (dP - Dq), dq = scale * dS . K
// As in the forward, the key tiles are dealt to the NWV waves of the block (private LDS slices, no block barrier in the
// loop) and the partial dq are summed through LDS at the end.
constexpr int kAttnWaves = 4;
template <int HDP, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_attn_bwd_q(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                    const float* __restrict__ lse, const float* __restrict__ Dq,
                                                    float* __restrict__ dqkv, int T, int H, int hd, float scale,
                                                    float drop_p, uint64_t seed, uint64_t offset) {
    constexpr int KT = 32;
    __shared__ float Ks[NWV][KT][HDP];
    __shared__ float Vs[NWV][KT][HDP];
    __shared__ float part[NWV][HDP][64];
    const int D = H * hd;
    const int b = blockIdx.z, h = blockIdx.y;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 64 + lane;
    const bool active = q < T;
    const size_t row0 = (size_t)b * T;
    float qr[HDP], dor[HDP], dq[HDP];
#pragma unroll
    for (int d = 0; d < HDP; ++d) { qr[d] = 0.f; dor[d] = 0.f; dq[d] = 0.f; }
    float L = 0.f, Dv = 0.f;
    if (active) {
        const float* qp = qkv + (row0 + q) * 3 * D + h * hd;
        const float* dp = dO + (row0 + q) * D + h * hd;
        for (int d = 0; d < hd; ++d) { qr[d] = qp[d] * scale; dor[d] = dp[d]; }
        L = lse[((size_t)b * H + h) * T + q];
        Dv = Dq[((size_t)b * H + h) * T + q];
    }
    const float keep_scale = (drop_p > 0.f) ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (int k0 = w * KT; k0 < T; k0 += NWV * KT) {
        const int kn = min(KT, T - k0);
        for (int id = lane; id < KT * HDP; id += 64) {
            const int j = id / HDP, d = id % HDP;
            float kv = 0.f, vv = 0.f;
            if (j < kn && d < hd) {
                const float* base = qkv + (row0 + k0 + j) * 3 * D + h * hd + d;
                kv = base[D];
                vv = base[2 * D];
            }
            Ks[w][j][d] = kv;
            Vs[w][j][d] = vv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int j4 = 0; j4 < kn; j4 += 4) {                  // k0 is a multiple of 4: one Philox group per pass
            uint32_t rv[4] = {0u, 0u, 0u, 0u};
            if (drop_p > 0.f) {
                const fd_u4 r = drop_bits(seed, offset, b * H + h, T, active ? q : 0, (k0 + j4) >> 2);
                rv[0] = r.x; rv[1] = r.y; rv[2] = r.z; rv[3] = r.w;
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j4 + jj;                          // rows j >= kn of Ks/Vs are zero: ds is masked below
                float s = 0.f, dpd = 0.f;
#pragma unroll
                for (int d = 0; d < HDP; ++d) {
                    s = fmaf(qr[d], Ks[w][j][d], s);
                    dpd = fmaf(dor[d], Vs[w][j][d], dpd);
                }
                const float p = expf(s - L);
                const float keep = drop_p > 0.f ? keep_of(rv[jj], drop_p, keep_scale) : 1.0f;
                const float ds = (j < kn) ? p * (keep * dpd - Dv) : 0.f;
#pragma unroll
                for (int d = 0; d < HDP; ++d) dq[d] = fmaf(ds, Ks[w][j][d], dq[d]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // tile reads done before the next fill
    }
    if (NWV > 1) {
#pragma unroll
        for (int d = 0; d < HDP; ++d) part[w][d][lane] = dq[d];
        __syncthreads();
        if (w != 0) return;
#pragma unroll
        for (int v = 1; v < NWV; ++v)
#pragma unroll
            for (int d = 0; d < HDP; ++d) dq[d] += part[v][d][lane];
    }
    if (active) {
        float* out = dqkv + (row0 + q) * 3 * D + h * hd;
        for (int d = 0; d < hd; ++d) out[d] = dq[d] * scale;
    }
}

// dk, dv: lane per key, queries streamed through LDS; query tiles dealt to the NWV waves, partial dk/dv summed at the end.
template <int HDP, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_attn_bwd_kv(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                     const float* __restrict__ lse, const float* __restrict__ Dq,
                                                     float* __restrict__ dqkv, int T, int H, int hd, float scale,
                                                     float drop_p, uint64_t seed, uint64_t offset) {
    constexpr int QT = 32;
    __shared__ float Qs_[NWV][QT][HDP];
    __shared__ float dOs_[NWV][QT][HDP];
    __shared__ float Ls_[NWV][QT], Ds_[NWV][QT];
    __shared__ float part[NWV][2 * HDP][64];
    const int D = H * hd;
    const int b = blockIdx.z, h = blockIdx.y;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float (*Qs)[HDP] = Qs_[w];
    float (*dOs)[HDP] = dOs_[w];
    float* Ls = Ls_[w];
    float* Ds = Ds_[w];
    const int key = blockIdx.x * 64 + lane;
    const bool active = key < T;
    const size_t row0 = (size_t)b * T;
    float kr[HDP], vr[HDP], dk[HDP], dv[HDP];
#pragma unroll
    for (int d = 0; d < HDP; ++d) { kr[d] = 0.f; vr[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    if (active) {
        const float* kp = qkv + (row0 + key) * 3 * D + h * hd;
        for (int d = 0; d < hd; ++d) { kr[d] = kp[D + d]; vr[d] = kp[2 * D + d]; }
    }
    const float keep_scale = (drop_p > 0.f) ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (int q0 = w * QT; q0 < T; q0 += NWV * QT) {
        const int qn = min(QT, T - q0);
        for (int id = lane; id < QT * HDP; id += 64) {
            const int j = id / HDP, d = id % HDP;
            float qv = 0.f, dv_ = 0.f;
            if (j < qn && d < hd) {
                qv = qkv[(row0 + q0 + j) * 3 * D + h * hd + d] * scale;
                dv_ = dO[(row0 + q0 + j) * D + h * hd + d];
            }
            Qs[j][d] = qv;
            dOs[j][d] = dv_;
        }
        if (lane < QT) {
            const int j = lane;
            Ls[j] = (j < qn) ? lse[((size_t)b * H + h) * T + q0 + j] : 0.f;
            Ds[j] = (j < qn) ? Dq[((size_t)b * H + h) * T + q0 + j] : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int j4 = 0; j4 < qn; j4 += 4) {
            // The 4 lanes of a quad hold the 4 keys of one Philox group.  Lane i evaluates the group for query j4+i; the
            // 4x4 transpose through DPP quad broadcasts gives every lane its own component for each of the 4 queries.
            uint32_t rv[4] = {0u, 0u, 0u, 0u};
            if (drop_p > 0.f) {
                const int qi = lane & 3;                       // == key & 3
                const fd_u4 r = drop_bits(seed, offset, b * H + h, T, q0 + j4 + qi, key >> 2);
#define FD_QUAD_PICK(M)                                                                                      \
    {                                                                                                        \
        const uint32_t a0 = quad_bcast<M>(r.x), a1 = quad_bcast<M>(r.y), a2 = quad_bcast<M>(r.z),            \
                       a3 = quad_bcast<M>(r.w);                                                              \
        rv[M] = qi == 0 ? a0 : qi == 1 ? a1 : qi == 2 ? a2 : a3;                                             \
    }
                FD_QUAD_PICK(0) FD_QUAD_PICK(1) FD_QUAD_PICK(2) FD_QUAD_PICK(3)
#undef FD_QUAD_PICK
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j4 + jj;                          // rows j >= qn of Qs/dOs are zero: masked below
                float s = 0.f, dpd = 0.f;
#pragma unroll
                for (int d = 0; d < HDP; ++d) {
                    s = fmaf(Qs[j][d], kr[d], s);
                    dpd = fmaf(dOs[j][d], vr[d], dpd);
                }
                const float p = (j < qn) ? expf(s - Ls[j]) : 0.f;
                const float keep = drop_p > 0.f ? keep_of(rv[jj], drop_p, keep_scale) : 1.0f;
                const float pd = p * keep;                        // dropped / rescaled probability
                const float ds = p * (keep * dpd - Ds[j]);
#pragma unroll
                for (int d = 0; d < HDP; ++d) {
                    dv[d] = fmaf(pd, dOs[j][d], dv[d]);
                    dk[d] = fmaf(ds, Qs[j][d], dk[d]);          // Qs already carries the softmax scale
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // tile reads done before the next fill
    }
    if (NWV > 1) {
#pragma unroll
        for (int d = 0; d < HDP; ++d) { part[w][d][lane] = dk[d]; part[w][HDP + d][lane] = dv[d]; }
        __syncthreads();
        if (w != 0) return;
#pragma unroll
        for (int v = 1; v < NWV; ++v)
#pragma unroll
            for (int d = 0; d < HDP; ++d) { dk[d] += part[v][d][lane]; dv[d] += part[v][HDP + d][lane]; }
    }
    if (active) {
        float* out = dqkv + (row0 + key) * 3 * D + h * hd;
        for (int d = 0; d < hd; ++d) {
            out[D + d] = dk[d];
            out[2 * D + d] = dv[d];
        }
    }
}

// dpos[t, :] += sum_b dh[b, t, :] ; dtemb[b, :] = sum_t dh[b, t, :]
// Four independent strands per output (loads in flight instead of one dependent chain), combined in a fixed order.
__global__ __launch_bounds__(256) void k_embed_reduce(const float* __restrict__ dh, float* __restrict__ dpos,
                                                       float* __restrict__ dtemb, int B, int T, int D) {
    // grid: blocks [0, nb_pos) own the positional-table gradient, the rest the time-embedding gradient; eight loads in flight
    // per output (four left the longer of the two sums a chain of T / 4 dependent round trips on B * D threads)
    const int nb_pos = (T * D + 255) / 256;
    if ((int)blockIdx.x < nb_pos) {
        const size_t id = blockIdx.x * (size_t)256 + threadIdx.x;
        if (id >= (size_t)T * D) return;
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = 0.f;
        int b = 0;
        for (; b + 7 < B; b += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += dh[(size_t)(b + k) * T * D + id];
        }
        for (; b < B; ++b) a[0] += dh[(size_t)b * T * D + id];
        dpos[id] += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    } else {
        // 64 columns x 4 strands of t per block, fixed-order combine
        __shared__ float red[4][64];
        const int col = threadIdx.x & 63, sub = threadIdx.x >> 6;
        const size_t id = (size_t)(blockIdx.x - nb_pos) * 64 + col;          // (b, d) flattened
        const bool ok = id < (size_t)B * D;
        const int b = ok ? (int)(id / D) : 0, d = ok ? (int)(id % D) : 0;
        red[sub][col] = ok ? sum_rows_strand(dh + (size_t)b * T * D + d, T, (size_t)D, sub) : 0.f;
        __syncthreads();
        if (sub == 0 && ok) dtemb[id] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    }
}

// ---- the embedding-side backward in two launches (fd_embed_backward) ----
// Per series b: wpart[b][d * C + c] = sum_t dh[b, t, d] * x[b, t, c] (its share of the embedder weight gradient) and
// dtemb[b, d] = sum_t dh[b, t, d].  grid (ceil((D * C + D) / 16), B): 16 outputs x 16 strands of t per block (a strand is T / 16
// rows, all of its loads in flight: the kernel is one or two memory round trips long), outputs [0, D * C) are the weight
// entries, [D * C, D * C + D) the time-embedding gradient; fixed-order combine.
__global__ __launch_bounds__(256) void k_embed_bwd_series(const float* __restrict__ dh, const float* __restrict__ x,
                                                           float* __restrict__ wpart, float* __restrict__ dtemb, int T, int C,
                                                           int D) {
    __shared__ float red[16][17];
    const int col = threadIdx.x & 15, sub = threadIdx.x >> 4;
    const int b = blockIdx.y, o = blockIdx.x * 16 + col, nW = D * C;
    const float* dhb = dh + (size_t)b * T * D;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    if (o < nW) {
        const int d = o / C, c = o - d * C;
        const float* pd = dhb + d;
        const float* px = x + (size_t)b * T * C + c;
        int t = sub;
        for (; t + 16 * 7 < T; t += 16 * 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = fmaf(pd[(size_t)(t + 16 * k) * D], px[(size_t)(t + 16 * k) * C], a[k]);
        }
        for (; t < T; t += 16) a[0] = fmaf(pd[(size_t)t * D], px[(size_t)t * C], a[0]);
    } else if (o < nW + D) {
        const float* pd = dhb + (o - nW);
        int t = sub;
        for (; t + 16 * 7 < T; t += 16 * 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += pd[(size_t)(t + 16 * k) * D];
        }
        for (; t < T; t += 16) a[0] += pd[(size_t)t * D];
    }
    red[sub][col] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (sub != 0 || o >= nW + D) return;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][col];
    if (o < nW) wpart[(size_t)b * nW + o] = v;
    else dtemb[(size_t)b * D + (o - nW)] = v;
}

// Everything that sums over the batch: 64 outputs x 4 strands of b per block, fixed-order combine
//   [0, T*D)                dpos[t, d]   += sum_b dh[b, t, d]
//   [.., + D*C)             demb_w[d, c] += sum_b wpart[b][d, c]
//   [.., + D*D)             dtd_w[i, j]  += sum_b dtemb[b, i] * emb[b, j]
//   [.., + D)               s = sum_b dtemb[b, d];  demb_b[d] += s;  dtd_b[d] += s   (sum_{b,t} dh = sum_b dtemb)
struct EmbedFinalArgs {
    const float *dh, *wpart, *dtemb, *emb;
    float *dpos, *demb_w, *demb_b, *dtd_w, *dtd_b;
    int B, T, C, D;
};
__global__ __launch_bounds__(256) void k_embed_bwd_final(EmbedFinalArgs A) {
    __shared__ float red[4][64];
    const int B = A.B, D = A.D;
    const size_t n0 = (size_t)A.T * D, n1 = n0 + (size_t)D * A.C, n2 = n1 + (size_t)D * D, n3 = n2 + D;
    const int col = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const size_t id = blockIdx.x * (size_t)64 + col;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    if (id < n3) {
        const float* p;
        size_t stride;
        if (id < n0) { p = A.dh + id; stride = n0; }
        else if (id < n1) { p = A.wpart + (id - n0); stride = (size_t)D * A.C; }
        else if (id < n2) { p = A.dtemb + (id - n1) / D; stride = D; }
        else { p = A.dtemb + (id - n2); stride = D; }
        int b = sub;
        if (id >= n1 && id < n2) {
            const float* q = A.emb + (id - n1) % D;
            for (; b + 4 * 7 < B; b += 4 * 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = fmaf(p[(size_t)(b + 4 * k) * stride], q[(size_t)(b + 4 * k) * D], a[k]);
            }
            for (; b < B; b += 4) a[0] = fmaf(p[(size_t)b * stride], q[(size_t)b * D], a[0]);
        } else {
            for (; b + 4 * 7 < B; b += 4 * 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += p[(size_t)(b + 4 * k) * stride];
            }
            for (; b < B; b += 4) a[0] += p[(size_t)b * stride];
        }
    }
    red[sub][col] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (sub != 0 || id >= n3) return;
    const float v = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    if (id < n0) A.dpos[id] += v;
    else if (id < n1) A.demb_w[id - n0] += v;
    else if (id < n2) A.dtd_w[id - n1] += v;
    else { A.demb_b[id - n2] += v; A.dtd_b[id - n2] += v; }
}

inline unsigned ew_grid(fd_ctx* ctx, size_t n) {
    size_t b = (n + 255) / 256;
    const size_t cap = (size_t)ctx->num_cu * 8;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

void colsum(fd_ctx* ctx, const float* x, float* out, int M, int N, hipStream_t s) { fd_defer(ctx, fd_colsum_det(ctx, x, out, M, N, s)); }

// tmp = dropout-backward(src) (site mask), bias gradient += column sums of tmp (fixed-order reduction)
void dropout_bwd_colsum(fd_ctx* ctx, const float* src, float* tmp, int M, int N, float p, uint64_t seed, uint64_t offset,
                        float* bias_grad, hipStream_t s) {
    const size_t n = (size_t)M * N;
    hipLaunchKernelGGL(k_dropout_copy, dim3(ew_grid(ctx, (n + 3) / 4)), dim3(256), 0, s, src, tmp, n, p, seed, offset);
    colsum(ctx, tmp, bias_grad, M, N, s);
}

void ln_bwd(fd_ctx* ctx, const float* dy, const float* x, const float* mr, const float* gamma, float* dx, float* dgamma,
            float* dbeta, int M, int D, hipStream_t s) {
    // about one block per CU at the training batch, never more than 64 tokens per block
    int tokens_per_block = (M + ctx->num_cu - 1) / ctx->num_cu;
    tokens_per_block = std::min(64, std::max(8, (tokens_per_block + 3) & ~3));
    const int nblk = (M + tokens_per_block - 1) / tokens_per_block;
    float* part = fd_red_scratch(ctx, (size_t)nblk * 2 * D);
    if (!part) {                             // nothing is launched, so no launch check would see it: the API call returns it
        fd_defer(ctx, fd_fail(ctx, FD_ERR_HIP, "ln_bwd: reduction scratch allocation failed"));
        return;
    }
    hipLaunchKernelGGL(k_ln_bwd, dim3(nblk), dim3(256), 8 * D * sizeof(float), s, dy, x, mr, gamma, dx, part, M, D, tokens_per_block);
    // rows are [dgamma | dbeta]: two strided sums
    hipLaunchKernelGGL(k_sum_rows_strided, dim3((D + 63) / 64), dim3(256), 0, s, part, nblk, 2 * D, 0, D, dgamma);
    hipLaunchKernelGGL(k_sum_rows_strided, dim3((D + 63) / 64), dim3(256), 0, s, part, nblk, 2 * D, D, D, dbeta);
}

template <int HDP>
void attn_bwd_t(const float* qkv, const float* dO, const float* lse, const float* Dq, float* dqkv, int B, int T, int H,
                int hd, float p, uint64_t seed, uint64_t offset, hipStream_t s) {
    dim3 grid((T + 63) / 64, H, B);
    const float scale = 1.0f / sqrtf((float)hd);
    constexpr int NWV = HDP <= 16 ? kAttnWaves : 1;            // (LDS: tiles + merge area per wave)
    hipLaunchKernelGGL((k_attn_bwd_q<HDP, NWV>), grid, dim3(64 * NWV), 0, s, qkv, dO, lse, Dq, dqkv, T, H, hd, scale, p, seed,
                       offset);
    hipLaunchKernelGGL((k_attn_bwd_kv<HDP, NWV>), grid, dim3(64 * NWV), 0, s, qkv, dO, lse, Dq, dqkv, T, H, hd, scale, p, seed,
                       offset);
}

}  // namespace

void fd_sum_rows(const float* part, int R, int N, float* out, bool accumulate, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_rows, dim3((N + 63) / 64), dim3(256), 0, s, part, R, N, out, accumulate ? 1 : 0);
}

int fd_colsum_det(fd_ctx* ctx, const float* x, float* out, int M, int N, hipStream_t s) {
    const int rows_per_block = 128;
    const int nblk = (M + rows_per_block - 1) / rows_per_block;
    float* part = fd_red_scratch(ctx, (size_t)nblk * N);
    if (!part) return fd_fail(ctx, FD_ERR_HIP, "fd_colsum_det: reduction scratch allocation failed");
    dim3 grid((N + 63) / 64, nblk);
    hipLaunchKernelGGL(k_colsum_part, grid, dim3(256), 0, s, x, part, M, N, rows_per_block);
    fd_sum_rows(part, nblk, N, out, true, s);
    return FD_OK;
}

// split-K partial sums of the weight-gradient GEMMs (dW = dY^T X reduces over all B*T tokens): 16 x the largest dW
constexpr size_t kSplitKFloats = (size_t)4 << 20;

size_t fd_score_bwd_workspace(const fd_score* m, int B) {
    const size_t M = (size_t)B * m->d.max_len, D = m->d.d_model, F = m->d.dim_ff, H = m->d.n_head, T = m->d.max_len;
    auto fl = [](size_t n) { return fd_ws::padded(n * sizeof(float)); };
    return 3 * fl(M * D) + fl(M * F) + fl(M * 3 * D) + fl((size_t)B * H * T) + fl((size_t)B * D) + fl(kSplitKFloats) + 4096;
}

// Gradients of the input side of the network from dh = d loss / d h0 (B*T, D): embedder, positional table, time embedding
// (h0 = X We^T + be + pe[t] + temb[b]; temb = emb Wd^T + bd; time_encoder.W is frozen: transformer.py:72-74).
// Shared by the exact-f32 and the bf16 backward; accumulates into `grads`.
int fd_embed_backward(fd_score* m, const float* dh, const float* emb, float* dtemb, float* grads, int B, float* skp,
                      size_t skp_floats, hipStream_t s) {
    fd_ctx* ctx = m->ctx;
    const int T = m->d.max_len, C = m->d.n_channels, D = m->d.d_model;
    const int M = B * T;
    static const bool unfused = getenv("FDIFF_EMBED_BWD_UNFUSED") != nullptr;
    if (!unfused && skp && (size_t)B * D * C <= skp_floats) {
        // two launches instead of eight (two split-K GEMMs + their reduces, two column sums in two stages each, the batch / time
        // reduce): everything here is a reduction of the (B*T, D) gradient over t or over b, ~100 us of 5 us kernels before
        hipLaunchKernelGGL(k_embed_bwd_series, dim3((unsigned)((D * C + D + 15) / 16), B), dim3(256), 0, s, dh, m->saved_x, skp, dtemb, T,
                           C, D);
        EmbedFinalArgs A{dh, skp, dtemb, emb, grads + m->pos, grads + m->emb_w, grads + m->emb_b, grads + m->td_w, grads + m->td_b,
                         B, T, C, D};
        const size_t n = (size_t)T * D + (size_t)D * C + (size_t)D * D + D;
        hipLaunchKernelGGL(k_embed_bwd_final, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, A);
        FD_LAUNCH_CHECK(ctx);
        return fd_take_deferred(ctx);
    }
    fdgemm::linear_bwd_weight(dh, m->saved_x, grads + m->emb_w, M, D, C, true, s, skp, skp_floats);
    colsum(ctx, dh, grads + m->emb_b, M, D, s);
    hipLaunchKernelGGL(k_embed_reduce, dim3((unsigned)((T * D + 255) / 256 + (B * D + 63) / 64)), dim3(256), 0, s, dh, grads + m->pos,
                       dtemb, B, T, D);
    fdgemm::linear_bwd_weight(dtemb, emb, grads + m->td_w, B, D, D, true, s, skp, skp_floats);
    colsum(ctx, dtemb, grads + m->td_b, B, D, s);
    FD_LAUNCH_CHECK(ctx);
    return fd_take_deferred(ctx);
}

extern "C" int fd_score_backward(fd_score* m, const float* dout, float* grads, int accumulate, void* stream) {
    if (!m) return FD_ERR_ARG;
    fd_ctx* ctx = m->ctx;
    FD_REQUIRE(ctx, dout && grads, "fd_score_backward: null pointer");
    if (!m->have_saved) return fd_fail(ctx, FD_ERR_STATE, "fd_score_backward: no training forward to differentiate");
    hipStream_t s = (hipStream_t)stream;
    if (m->backbone != FD_BACKBONE_TRANSFORMER) {
        if (ctx->ws_gen != m->saved_ws_gen || ctx->ws != m->saved_ws)
            return fd_fail(ctx, FD_ERR_STATE, "fd_score_backward: another engine call used the context workspace after "
                           "fd_score_forward_train (the saved activations live there)");
        m->have_saved = false;
        return fd_bb_backward(m, dout, grads, accumulate, s);
    }
    if (m->saved_bf16) {
        if (ctx->ws_gen != m->saved_ws_gen || ctx->ws != m->saved_ws)
            return fd_fail(ctx, FD_ERR_STATE, "fd_score_backward: another engine call used the context workspace after "
                           "fd_score_forward_train (the saved activations live there)");
        m->have_saved = false;
        return fd_score_backward_bf16(m, dout, grads, accumulate, s);
    }
    const int B = m->saved_B;
    const int T = m->d.max_len, C = m->d.n_channels, D = m->d.d_model, H = m->d.n_head, F = m->d.dim_ff;
    const int L = m->d.num_layers, hd = D / H;
    const int M = B * T;
    const float p = m->saved_p;
    const float* P = m->params;
    // the training forward reserved room for both carve-outs; re-derive the same pointers
    const size_t fwd_bytes = fd_score_f32_workspace(m, B, true);
    if (ctx->ws_bytes < fwd_bytes + fd_score_bwd_workspace(m, B))
        return fd_fail(ctx, FD_ERR_STATE, "fd_score_backward: workspace was resized since the training forward");
    if (ctx->ws_gen != m->saved_ws_gen || ctx->ws != m->saved_ws)
        return fd_fail(ctx, FD_ERR_STATE, "fd_score_backward: another engine call used the context workspace after "
                       "fd_score_forward_train (the saved activations live there); run forward_train -> loss -> backward "
                       "without other workspace-using calls on this context in between");
    m->have_saved = false;            // one backward per training forward (its scratch overlays nothing, but the dropout
                                      // masks / inputs belong to that forward only)
    fd_ws ws(ctx, /*reader=*/true);
    fd_saved sv;
    fd_score_carve_saved(m, B, ws, sv);
    ws.off = fwd_bytes;
    float* dh = ws.take<float>((size_t)M * D);
    float* ds = ws.take<float>((size_t)M * D);
    float* tmp = ws.take<float>((size_t)M * D);
    float* dact = ws.take<float>((size_t)M * F);
    float* dqkv = ws.take<float>((size_t)M * 3 * D);
    float* Dq = ws.take<float>((size_t)B * H * T);
    float* dtemb = ws.take<float>((size_t)B * D);
    float* skp = ws.take<float>(kSplitKFloats);
    size_t gsk_n = 0;
    float* gsk = fd_gemm_scratch(ctx, &gsk_n);

    if (!accumulate) FD_HIP(ctx, hipMemsetAsync(grads, 0, sizeof(float) * (size_t)m->nparams, s));
    const float inv_keep = (p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;

    // ---- unembedder: out = hL Wu^T + bu
    fdgemm::linear_bwd_weight(dout, sv.hL, grads + m->un_w, M, C, D, true, s, skp, kSplitKFloats);
    colsum(ctx, dout, grads + m->un_b, M, C, s);
    fdgemm::linear_bwd_input(dout, P + m->un_w, dh, M, C, D, false, s);

    for (int i = L - 1; i >= 0; --i) {
        const fd_layer_off& lo = m->layers[i];
        const fd_saved_layer& A = sv.layers[i];
        // x_next = LN2(s2): dh -> ds (= d s2)
        ln_bwd(ctx, dh, A.s2, A.mr2, P + lo.n2_w, ds, grads + lo.n2_w, grads + lo.n2_b, M, D, s);
        // s2 = x1 + drop(f2), f2 = hact W2^T + b2
        dropout_bwd_colsum(ctx, ds, tmp, M, D, p, m->saved_seed, fd_dropout_site_offset(m->saved_offset, i, 3), grads + lo.l2_b, s);
        fdgemm::linear_bwd_weight(tmp, A.hact, grads + lo.l2_w, M, D, F, true, s, skp, kSplitKFloats);
        fdgemm::linear_bwd_input(tmp, P + lo.l2_w, dact, M, D, F, false, s);
        // hact = drop(relu(x1 W1^T + b1))
        // (one fused pass with the bias column sums measured slower than these two: 46-79 us vs 25 + 13 us)
        hipLaunchKernelGGL(k_relu_drop_bwd, dim3(ew_grid(ctx, (size_t)M * F)), dim3(256), 0, s, dact, A.hact, (size_t)M * F,
                           inv_keep);
        colsum(ctx, dact, grads + lo.l1_b, M, F, s);
        fdgemm::linear_bwd_weight(dact, A.x1, grads + lo.l1_w, M, F, D, true, s, skp, kSplitKFloats);
        fdgemm::linear_bwd_input(dact, P + lo.l1_w, ds, M, F, D, true, s, gsk, gsk_n);   // ds = d x1 (residual + FFN branch); K = F: split
        // x1 = LN1(s1): ds -> dh (= d s1)
        ln_bwd(ctx, ds, A.s1, A.mr1, P + lo.n1_w, dh, grads + lo.n1_w, grads + lo.n1_b, M, D, s);
        // s1 = x0 + drop(proj), proj = att Wo^T + bo
        dropout_bwd_colsum(ctx, dh, tmp, M, D, p, m->saved_seed, fd_dropout_site_offset(m->saved_offset, i, 1), grads + lo.out_b, s);
        fdgemm::linear_bwd_weight(tmp, A.att, grads + lo.out_w, M, D, D, true, s, skp, kSplitKFloats);
        fdgemm::linear_bwd_input(tmp, P + lo.out_w, ds, M, D, D, false, s);      // ds = d att
        // attention core
        {
            const size_t n = (size_t)B * H * T;
            hipLaunchKernelGGL(k_attn_rowdot, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ds, A.att, Dq, B, T, H, hd);
            const uint64_t off0 = fd_dropout_site_offset(m->saved_offset, i, 0);
            if (hd <= 8) attn_bwd_t<8>(A.qkv, ds, A.lse, Dq, dqkv, B, T, H, hd, p, m->saved_seed, off0, s);
            else if (hd <= 16) attn_bwd_t<16>(A.qkv, ds, A.lse, Dq, dqkv, B, T, H, hd, p, m->saved_seed, off0, s);
            else if (hd <= 32) attn_bwd_t<32>(A.qkv, ds, A.lse, Dq, dqkv, B, T, H, hd, p, m->saved_seed, off0, s);
            else attn_bwd_t<64>(A.qkv, ds, A.lse, Dq, dqkv, B, T, H, hd, p, m->saved_seed, off0, s);
        }
        // qkv = x0 Win^T + bin
        fdgemm::linear_bwd_weight(dqkv, A.x0, grads + lo.in_w, M, 3 * D, D, true, s, skp, kSplitKFloats);
        colsum(ctx, dqkv, grads + lo.in_b, M, 3 * D, s);
        fdgemm::linear_bwd_input(dqkv, P + lo.in_w, dh, M, 3 * D, D, true, s);   // dh = d x0 (residual + attention branch)
    }

    if (int rc = fd_embed_backward(m, dh, sv.emb, dtemb, grads, B, skp, kSplitKFloats, s)) return rc;
    FD_LAUNCH_CHECK(ctx);
    return fd_take_deferred(ctx);
}
