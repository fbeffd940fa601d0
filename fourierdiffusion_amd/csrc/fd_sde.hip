// fd_sde.hip -- HBM-bound elementwise kernels of the SDE (reference: src/fdiff/schedulers/sde.py,
// src/fdiff/utils/losses.py).  The reference spells its row scalings as diag_embed + (T,T)@(B,T,C)
// matmuls; here every one is a fused per-element multiply by G[t] with on-device Philox noise.
//
// Layout: (B,T,C) float32 row-major; thread i owns the 16-byte group of elements [4i, 4i+4) so that
// every access is a coalesced dwordx4 and one Philox counter feeds exactly one group.
#include "fd_common.h"
#include "fd_philox.h"
#include "fd_sde.h"

namespace {

constexpr int kBlock = 256;

// (tails by predicated scalar accesses: indexing the float4 with a run-time subscript put it in scratch memory, and with a
// private segment k_sde_step took 18.5 us for 2.4 MB instead of 5.8, 73 us instead of 59 at 117 MB x 3 -- rocprofv3 kernel
// times, profiles/r03_scratch_sde_step.txt; scripts/scratch_report.py lists every kernel of the library that uses scratch)
__device__ __forceinline__ float4 ld4(const float* p, size_t e, size_t n) {
    if (e + 4 <= n) return *reinterpret_cast<const float4*>(p + e);
    float4 v;
    v.x = e < n ? p[e] : 0.f;
    v.y = e + 1 < n ? p[e + 1] : 0.f;
    v.z = e + 2 < n ? p[e + 2] : 0.f;
    v.w = 0.f;
    return v;
}
__device__ __forceinline__ void st4(float* p, size_t e, size_t n, float4 v) {
    if (e + 4 <= n) {
        *reinterpret_cast<float4*>(p + e) = v;
        return;
    }
    if (e < n) p[e] = v.x;
    if (e + 1 < n) p[e + 1] = v.y;
    if (e + 2 < n) p[e + 2] = v.z;
}

__global__ __launch_bounds__(kBlock) void k_randn(float* __restrict__ out, size_t n, uint64_t seed,
                                                    uint64_t offset) {
    const size_t ngroups = (n + 3) / 4;
    for (size_t g = blockIdx.x * (size_t)kBlock + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * kBlock) {
        float z[4];
        fd_randn4(offset + g, seed, z);
        st4(out, g * 4, n, float4{z[0], z[1], z[2], z[3]});
    }
}

// Time index t of the four elements of group e .. e+3 of a (B, T, C) tensor: ONE division for the row and one for its remainder
// by T, then carries (eight runtime divisions per group were ~200 VALU instructions, more than the Philox evaluation).
__device__ __forceinline__ void group_rows(size_t e, size_t n, int T, int C, int (&t)[4]) {
    if ((n >> 32) == 0 && C >= 4) {           // wave-uniform
        const unsigned e32 = (unsigned)e;
        const unsigned row = e32 / (unsigned)C;
        unsigned c = e32 - row * (unsigned)C;
        unsigned tt = row % (unsigned)T;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t[i] = (int)tt;
            if (++c == (unsigned)C) { c = 0; tt = (tt + 1 == (unsigned)T) ? 0u : tt + 1; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = (int)(((e + i) / (size_t)C) % (size_t)T);
    }
}

// Euler-Maruyama reverse step (sde.py:129-165, 215-246), one pass: 12 bytes per element (x, score in; x' out), noise from the
// Philox stream.  Two 16-byte groups per thread and iteration, all four loads issued before the first Philox round, so that
// the ~350 VALU instructions of two counters run under the loads' latency (one group per iteration left the kernel at 3.3 TB/s
// where the 28-byte-per-parameter AdamW pass streams 5.0).
// out = scale * G[t] * z
__global__ __launch_bounds__(kBlock) void k_prior(const float* __restrict__ G, const float* __restrict__ zin,
                                                    float* __restrict__ out, size_t n, int T, int C,
                                                    float scale, uint64_t seed, uint64_t offset) {
    const size_t ngroups = (n + 3) / 4;
    for (size_t g = blockIdx.x * (size_t)kBlock + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * kBlock) {
        const size_t e = g * 4;
        float z[4];
        if (zin) {
            const float4 v = ld4(zin, e, n);
            z[0] = v.x; z[1] = v.y; z[2] = v.z; z[3] = v.w;
        } else {
            fd_randn4(offset + g, seed, z);
        }
        int t[4];
        group_rows(e, n, T, C, t);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (e + i < n) ? scale * (G[t[i]] * z[i]) : 0.f;
        st4(out, e, n, float4{o[0], o[1], o[2], o[3]});
    }
}

typedef __attribute__((ext_vector_type(4))) float f32x4_nt;
__device__ __forceinline__ float4 ldnt4(const float* p) {
    const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
    return float4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void stnt4(float* p, float a, float b, float c, float d) {
    __builtin_nontemporal_store(f32x4_nt{a, b, c, d}, reinterpret_cast<f32x4_nt*>(p));
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void k_sde_step(const float* __restrict__ G, const float* __restrict__ x,
                                                       const float* __restrict__ score,
                                                       const float* __restrict__ zin, float* __restrict__ out,
                                                       size_t n, int T, int C, SdeCoef cf, uint64_t seed,
                                                       uint64_t offset) {
    const size_t ngroups = (n + 3) / 4;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t g0 = blockIdx.x * (size_t)kBlock + threadIdx.x; g0 < ngroups; g0 += 2 * stride) {
        const size_t g1 = g0 + stride;
        const bool two = g1 < ngroups;
        const size_t e0 = g0 * 4, e1 = (two ? g1 : g0) * 4;
        const bool full = e0 + 4 <= n && e1 + 4 <= n;
        float4 xv0, sv0, xv1, sv1;
        if (NT && full) {       // streamed once: keep the lines out of the way of the score network's weights in L2 / MALL
            xv0 = ldnt4(x + e0); sv0 = ldnt4(score + e0);
            xv1 = ldnt4(x + e1); sv1 = ldnt4(score + e1);
        } else {
            xv0 = ld4(x, e0, n); sv0 = ld4(score, e0, n);
            xv1 = ld4(x, e1, n); sv1 = ld4(score, e1, n);
        }
        float z0[4], z1[4];
        if (zin) {
            const float4 v0 = ld4(zin, e0, n), v1 = ld4(zin, e1, n);
            z0[0] = v0.x; z0[1] = v0.y; z0[2] = v0.z; z0[3] = v0.w;
            z1[0] = v1.x; z1[1] = v1.y; z1[2] = v1.z; z1[3] = v1.w;
        } else {
            fd_randn4(offset + g0, seed, z0);
            fd_randn4(offset + (two ? g1 : g0), seed, z1);
        }
        int t0[4], t1[4];
        group_rows(e0, n, T, C, t0);
        group_rows(e1, n, T, C, t1);
        const float xs0[4] = {xv0.x, xv0.y, xv0.z, xv0.w}, ss0[4] = {sv0.x, sv0.y, sv0.z, sv0.w};
        const float xs1[4] = {xv1.x, xv1.y, xv1.z, xv1.w}, ss1[4] = {sv1.x, sv1.y, sv1.z, sv1.w};
        float o0[4], o1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o0[i] = fd_sde_apply(xs0[i], ss0[i], z0[i], G[t0[i]], cf);
            o1[i] = fd_sde_apply(xs1[i], ss1[i], z1[i], G[t1[i]], cf);
        }
        if (NT && full) {
            stnt4(out + e0, o0[0], o0[1], o0[2], o0[3]);
            if (two) stnt4(out + e1, o1[0], o1[1], o1[2], o1[3]);
        } else {
            st4(out, e0, n, float4{o0[0], o0[1], o0[2], o0[3]});
            if (two) st4(out, e1, n, float4{o1[0], o1[1], o1[2], o1[3]});
        }
    }
}

// Perturbation kernel of the loss (losses.py:66-85): per-sample t.
__global__ __launch_bounds__(kBlock) void k_perturb(const float* __restrict__ G, const float* __restrict__ x,
                                                      const float* __restrict__ tvec,
                                                      const float* __restrict__ zin, float* __restrict__ xn,
                                                      float* __restrict__ target, float* __restrict__ std_out,
                                                      size_t n, int T, int C, int kind, float p0, float p1,
                                                      uint64_t seed, uint64_t offset) {
    const size_t ngroups = (n + 3) / 4;
    const size_t per_b = (size_t)T * C;
    for (size_t g = blockIdx.x * (size_t)kBlock + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * kBlock) {
        const size_t e = g * 4;
        const float4 xv = ld4(x, e, n);
        float z[4];
        if (zin) {
            const float4 v = ld4(zin, e, n);
            z[0] = v.x; z[1] = v.y; z[2] = v.z; z[3] = v.w;
        } else {
            fd_randn4(offset + g, seed, z);
        }
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        float o[4], tg[4];
        // (sample, time, channel) of the group's first element by two divisions, the other three by carries; the per-sample
        // coefficients (two exponentials, or a power) are evaluated once per group and again only where the group crosses into the
        // next sample -- per element they were ~600 instructions around 12 bytes of traffic
        const size_t row0 = e / (size_t)C;
        int c = (int)(e - row0 * (size_t)C);
        int b = (int)(row0 / (size_t)T);
        int t = (int)(row0 - (size_t)b * T);
        auto coef = [&](int bb, float& mcoef, float& sdev) {
            const float tt = tvec[bb];
            if (kind == 0) {
                const float lmc = -0.25f * tt * tt * (p1 - p0) - 0.5f * tt * p0;   // sde.py:195-197
                mcoef = expf(lmc);
                sdev = sqrtf(1.0f - expf(2.0f * lmc));                              // sde.py:203-205
            } else {
                mcoef = 1.0f;
                sdev = p0 * powf(p1 / p0, tt);                                      // sde.py:117
            }
        };
        const int Bn = (int)(n / per_b);
        float mcoef, s;
        coef(b < Bn ? b : Bn - 1, mcoef, s);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = e + i < n;
            const float sd = s * G[t];
            o[i] = mcoef * xs[i] + sd * z[i];
            tg[i] = z[i] / sd;
            if (std_out && c == 0 && in) std_out[(size_t)b * T + t] = sd;
            if (++c == C) {
                c = 0;
                if (++t == T) {
                    t = 0;
                    ++b;
                    if (i < 3 && e + i + 1 < n) coef(b, mcoef, s);
                }
            }
        }
        st4(xn, e, n, float4{o[0], o[1], o[2], o[3]});
        if (target) st4(target, e, n, float4{tg[0], tg[1], tg[2], tg[3]});
    }
}

// Denoising score-matching loss (losses.py:92-124) + gradient wrt score.
// One block per sample b: w_b from std row, then sum over (t,c).
__global__ __launch_bounds__(kBlock) void k_dsm_loss(const float* __restrict__ score,
                                                       const float* __restrict__ target,
                                                       const float* __restrict__ stdv, int lw,
                                                       float* __restrict__ loss_out, float* __restrict__ dscore,
                                                       int B, int T, int C, float* __restrict__ partial) {
    __shared__ float red[kBlock / 64];
    __shared__ float w_sh;
    const int b = blockIdx.x;
    const float* sd = stdv + (size_t)b * T;
    // w_b = 1 / sum_k std^-2   (losses.py:96)
    float acc = 0.f;
    for (int k = threadIdx.x; k < T; k += kBlock) acc += 1.0f / (sd[k] * sd[k]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < kBlock / 64; ++i) s += red[i];
        w_sh = 1.0f / s;
    }
    __syncthreads();
    const float w = w_sh;
    const size_t per_b = (size_t)T * C;
    const float inv_cnt = 1.0f / ((float)per_b * (float)B);
    float part = 0.f;
    for (size_t i = threadIdx.x; i < per_b; i += kBlock) {
        const size_t e = (size_t)b * per_b + i;
        const int k = (int)(i / C);
        const float d = score[e] + target[e];
        float coef;                       // loss element = coef * d^2
        if (lw) coef = sd[k] * sd[k];     // (std*(s+target))^2, losses.py:115-121
        else coef = w;                    // w_b (s+target)^2,   losses.py:100-102
        part += coef * d * d;
        if (dscore) dscore[e] = 2.0f * coef * d * inv_cnt;
    }
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < kBlock / 64; ++i) s += red[i];
        if (partial) partial[b] = s * inv_cnt;                 // summed in fixed order by k_loss_sum
        else atomicAdd(loss_out, s * inv_cnt);
    }
}

// loss = sum_b partial[b] in a fixed order (lane-strided strands, then a fixed shuffle tree): the same inputs give the
// same loss bit for bit, which float atomics across the per-sample blocks did not
__global__ __launch_bounds__(64) void k_loss_sum(const float* __restrict__ partial, int B, float* __restrict__ loss_out) {
    float v = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) v += partial[b];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (threadIdx.x == 0) *loss_out = v;
}

inline int grid_for(size_t ngroups, int num_cu) {
    size_t blocks = (ngroups + kBlock - 1) / kBlock;
    static const int mult = getenv("FDIFF_SDE_GRIDMULT") ? atoi(getenv("FDIFF_SDE_GRIDMULT")) : 64;
    size_t cap = (size_t)num_cu * mult;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int check_btc(fd_ctx* ctx, int B, int T, int C) {
    FD_REQUIRE(ctx, B > 0 && T > 0 && C > 0, "bad shape B=%d T=%d C=%d", B, T, C);
    return FD_OK;
}

}  // namespace

extern "C" int fd_randn(fd_ctx* ctx, float* out, size_t n, uint64_t seed, uint64_t offset, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, out && n > 0, "fd_randn: null output or n == 0");
    const size_t ng = (n + 3) / 4;
    hipLaunchKernelGGL(k_randn, dim3(grid_for(ng, ctx->num_cu)), dim3(kBlock), 0, (hipStream_t)stream, out, n,
                       seed, offset);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

// Raw generator output and the dropout decisions derived from it (tests pin both to the oracle's numpy restatement and to the
// published Philox4x32-10 known-answer vector).
namespace {
__global__ __launch_bounds__(kBlock) void k_philox_words(uint32_t* __restrict__ out, size_t n, uint64_t seed, uint64_t offset) {
    for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const fd_u4 r = fd_philox4x32_10(offset + i, seed);
        *reinterpret_cast<uint4*>(out + 4 * i) = uint4{r.x, r.y, r.z, r.w};
    }
}
__global__ __launch_bounds__(kBlock) void k_drop16(unsigned short* __restrict__ out, size_t n, uint64_t seed, uint64_t offset,
                                                    unsigned thr16) {
    for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
        out[i] = (unsigned short)fd_drop16(offset + i, seed, thr16);
}
}  // namespace

extern "C" int fd_philox_words(fd_ctx* ctx, uint32_t* out, size_t n_counters, uint64_t seed, uint64_t offset, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, out && n_counters > 0, "fd_philox_words: null output or no counters");
    hipLaunchKernelGGL(k_philox_words, dim3(grid_for(n_counters, ctx->num_cu)), dim3(kBlock), 0, (hipStream_t)stream, out, n_counters,
                       seed, offset);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_dropout_decisions(fd_ctx* ctx, uint16_t* out, size_t n_counters, float p, uint64_t seed, uint64_t offset,
                                    void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, out && n_counters > 0, "fd_dropout_decisions: null output or no counters");
    FD_REQUIRE(ctx, p >= 0.f && p < 1.f, "fd_dropout_decisions: p=%f", p);
    unsigned thr16 = (unsigned)((double)p * 65536.0 + 0.5);      // (the training path's rounding: fd_train_bf16.hip make_dims)
    if (p > 0.f && thr16 == 0) thr16 = 1;
    hipLaunchKernelGGL(k_drop16, dim3(grid_for(n_counters, ctx->num_cu)), dim3(kBlock), 0, (hipStream_t)stream, out, n_counters, seed,
                       offset, thr16);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_prior_sample(fd_ctx* ctx, const fd_sde_params* sde, const float* G, const float* z,
                               uint64_t seed, uint64_t offset, float* out, int B, int T, int C, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, sde && G && out, "fd_prior_sample: null pointer");
    if (int rc = check_btc(ctx, B, T, C)) return rc;
    const size_t n = (size_t)B * T * C;
    const float scale = (sde->kind == 1) ? sde->p1 : 1.0f;   // sde.py:125-127
    hipLaunchKernelGGL(k_prior, dim3(grid_for((n + 3) / 4, ctx->num_cu)), dim3(kBlock), 0, (hipStream_t)stream,
                       G, z, out, n, T, C, scale, seed, offset);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_sde_step(fd_ctx* ctx, const fd_sde_params* sde, const float* G, const float* x,
                           const float* score, const float* z, uint64_t seed, uint64_t offset, double t,
                           float dt, float* out, int B, int T, int C, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, sde && G && x && score && out, "fd_sde_step: null pointer");
    FD_REQUIRE(ctx, sde->kind == 0 || sde->kind == 1, "fd_sde_step: unknown SDE kind %d", sde->kind);
    FD_REQUIRE(ctx, dt > 0.f, "fd_sde_step: step size must be > 0 (sde.py:158)");
    if (int rc = check_btc(ctx, B, T, C)) return rc;
    const size_t n = (size_t)B * T * C;
    const SdeCoef cf = fd_sde_coef(*sde, t, dt);
    // nontemporal loads / stores when the three streams exceed what the 256 MB Infinity Cache can hold between kernels
    // ((4096, 256, 28) = 352 MB: 3.6 -> 4.4 TB/s); smaller steps -- the sampler's x and score were just written by the score
    // network and are still cache resident -- run 15-25 % faster with ordinary accesses.  FDIFF_SDE_NT=0/1 forces either form.
    const char* nte = getenv("FDIFF_SDE_NT");
    const bool nt = nte ? nte[0] == '1' : (12.0 * (double)n > 160.0e6);
    const size_t ngroups2 = ((n + 3) / 4 + 1) / 2;                  // two groups per thread and iteration
    if (nt) hipLaunchKernelGGL(k_sde_step<true>, dim3(grid_for(ngroups2, ctx->num_cu)), dim3(kBlock), 0,
                               (hipStream_t)stream, G, x, score, z, out, n, T, C, cf, seed, offset);
    else hipLaunchKernelGGL(k_sde_step<false>, dim3(grid_for(ngroups2, ctx->num_cu)), dim3(kBlock), 0,
                            (hipStream_t)stream, G, x, score, z, out, n, T, C, cf, seed, offset);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

// Langevin corrector step of a predictor-corrector sampler (Song et al. 2021, Alg. 4/5; NOT in the reference, whose sampler is
// predictor-only: src/fdiff/sampling/sampler.py:24-43 -- default off, parity unpinned).  In the coordinates whitened by G
// (x = G x_w, score_w = G score): per series  eps = 2 alpha (snr |z| / |G score|)^2 ;  x <- x + eps G^2 score + sqrt(2 eps) G z.
// One workgroup per series: two fixed-order norm reductions, then the update.
namespace {
__global__ __launch_bounds__(256) void k_langevin(const float* __restrict__ G, const float* __restrict__ x,
                                                   const float* __restrict__ score, const float* __restrict__ z,
                                                   float* __restrict__ out, int T, int C, float snr, float alpha, uint64_t seed,
                                                   uint64_t offset) {
    __shared__ float red[2][4];
    const int b = blockIdx.x;
    const size_t per = (size_t)T * C, base = (size_t)b * per;
    const size_t ng = (per + 3) / 4;                           // groups of 4 elements: one Philox counter each
    float sg = 0.f, sz = 0.f;
    for (size_t gi = threadIdx.x; gi < ng; gi += 256) {
        float zz[4];
        if (!z) fd_randn4(offset + (base >> 2) + gi, seed, zz);    // (per % 4 == 0 is required for on-device noise)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t e = gi * 4 + i;
            if (e < per) {
                const float gk = G[e / C];
                const float sv = gk * score[base + e];
                const float zv = z ? z[base + e] : zz[i];
                sg += sv * sv;
                sz += zv * zv;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { sg += __shfl_down(sg, o); sz += __shfl_down(sz, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sg; red[1][threadIdx.x >> 6] = sz; }
    __syncthreads();
    const float gn = sqrtf((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    const float zn = sqrtf((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    const float r = snr * zn / fmaxf(gn, 1e-20f);
    const float eps = 2.0f * alpha * r * r;
    const float sq = sqrtf(2.0f * eps);
    for (size_t gi = threadIdx.x; gi < ng; gi += 256) {
        float zz[4];
        if (!z) fd_randn4(offset + (base >> 2) + gi, seed, zz);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t e = gi * 4 + i;
            if (e < per) {
                const float gk = G[e / C];
                const float zv = z ? z[base + e] : zz[i];
                out[base + e] = x[base + e] + eps * (gk * gk) * score[base + e] + sq * (gk * zv);
            }
        }
    }
}
}  // namespace

extern "C" int fd_langevin_step(fd_ctx* ctx, const float* G, const float* x, const float* score, const float* z, uint64_t seed,
                                uint64_t offset, float snr, float alpha, float* out, int B, int T, int C, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, G && x && score && out, "fd_langevin_step: null pointer");
    FD_REQUIRE(ctx, snr > 0.f && alpha > 0.f, "fd_langevin_step: snr=%f alpha=%f", snr, alpha);
    if (int rc = check_btc(ctx, B, T, C)) return rc;
    FD_REQUIRE(ctx, z || ((size_t)T * C) % 4 == 0, "fd_langevin_step: on-device noise needs T*C %% 4 == 0 (inject z otherwise)");
    hipLaunchKernelGGL(k_langevin, dim3(B), dim3(256), 0, (hipStream_t)stream, G, x, score, z, out, T, C, snr, alpha, seed, offset);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_perturb(fd_ctx* ctx, const fd_sde_params* sde, const float* G, const float* x, const float* t,
                          const float* z, uint64_t seed, uint64_t offset, float* x_noisy, float* target,
                          float* std_out, int B, int T, int C, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, sde && G && x && t && x_noisy, "fd_perturb: null pointer");
    FD_REQUIRE(ctx, sde->kind == 0 || sde->kind == 1, "fd_perturb: unknown SDE kind %d", sde->kind);
    if (int rc = check_btc(ctx, B, T, C)) return rc;
    const size_t n = (size_t)B * T * C;
    hipLaunchKernelGGL(k_perturb, dim3(grid_for((n + 3) / 4, ctx->num_cu)), dim3(kBlock), 0, (hipStream_t)stream,
                       G, x, t, z, x_noisy, target, std_out, n, T, C, sde->kind, sde->p0, sde->p1, seed, offset);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_dsm_loss(fd_ctx* ctx, const float* score, const float* target, const float* std,
                           int likelihood_weighting, float* loss_out, float* dscore, int B, int T, int C,
                           void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, score && target && std && loss_out, "fd_dsm_loss: null pointer");
    if (int rc = check_btc(ctx, B, T, C)) return rc;
    size_t nscr = 0;
    float* partial = fd_gemm_scratch(ctx, &nscr);              // ctx-owned, stream-ordered with the GEMMs that share it
    if (partial && (size_t)B <= nscr) {
        hipLaunchKernelGGL(k_dsm_loss, dim3(B), dim3(kBlock), 0, (hipStream_t)stream, score, target, std,
                           likelihood_weighting, loss_out, dscore, B, T, C, partial);
        hipLaunchKernelGGL(k_loss_sum, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, B, loss_out);
    } else {
        FD_HIP(ctx, hipMemsetAsync(loss_out, 0, sizeof(float), (hipStream_t)stream));
        hipLaunchKernelGGL(k_dsm_loss, dim3(B), dim3(kBlock), 0, (hipStream_t)stream, score, target, std,
                           likelihood_weighting, loss_out, dscore, B, T, C, (float*)nullptr);
    }
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}
