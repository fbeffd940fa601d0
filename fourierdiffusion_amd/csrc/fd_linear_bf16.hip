// fd_linear_bf16.hip -- bf16 MFMA projections of the step-by-step path for the widths OUTSIDE the persistent kernel's family
// (d_model 32, 64, 96, 128 ...): the attention in-projection  qkv = h W_in^T + b_in  and the out-projection fused with the
// residual add and LayerNorm1  x = LN1(h + att W_o^T + b_o)  (torch's TransformerEncoderLayer, post-norm;
// src/fdiff/models/score_models.py:57-62).  They replace the exact-f32 GEMM (+ add_layernorm) launches of those widths:
// at d_model 64 the fp32-MFMA projections were 45 % of a layer's time.
//
// Operand convention as everywhere (v_mfma_f32_16x16x32_bf16, transposed GEMM): weights are the A operand in 1 KiB fragment
// blocks [row tile][k-step] (rows of W, bias in k-slot K: fd_score_bf16.hip IMG_EMB layout), the 16 tokens of a tile ride on
// lane & 15, activations are converted fp32 -> bf16 B fragments on the fly with the constant 1.0 in k-slot K.  fp32 accumulate,
// fp32 residual and LayerNorm.  One wave = one token tile at a time; a workgroup (4 waves) stages the weight image in LDS
// once and walks its share of the token tiles.
#include <hip/hip_runtime.h>

#include "fd_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

namespace {

constexpr int NWL = 4;

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ f32x4 f4zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void swap32(float v, float& a, float& b) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned r0 = r.x, r1 = r.y;
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void swap16(float v, float& a, float& b) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned r0 = r.x, r1 = r.y;
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float group_sum(float v) {      // over the 4 lane groups holding one token
    float a, b;
    swap32(v, a, b);
    swap16(a + b, a, b);
    return a + b;
}

// B fragments of one token tile from fp32 rows x (M, K), K % 4 == 0: lane (tok, g) holds k-slots 32 ks + 8 g .. +7; slot K = 1.0
template <int KS>
__device__ __forceinline__ void x_frags(const float* __restrict__ x, int m, bool valid, int K, int g, bf16x8 (&xf)[KS]) {
    const int mc = valid ? m : 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k0 = 32 * ks + 8 * g;
        // (clamped addresses + select: a load inside a divergent branch is waited for at the end of the branch)
        float4 a = *reinterpret_cast<const float4*>(x + (size_t)mc * K + min(k0, K - 4));
        float4 c = *reinterpret_cast<const float4*>(x + (size_t)mc * K + min(k0 + 4, K - 4));
        const float4 one = {1.f, 0.f, 0.f, 0.f}, zero = {0.f, 0.f, 0.f, 0.f};
        if (!(valid && k0 + 4 <= K)) a = (valid && k0 == K) ? one : zero;
        if (!(valid && k0 + 8 <= K)) c = (valid && k0 + 4 == K) ? one : zero;
        const u32x4 pk = {cvt_pk_bf16(a.x, a.y), cvt_pk_bf16(a.z, a.w), cvt_pk_bf16(c.x, c.y), cvt_pk_bf16(c.z, c.w)};
        xf[ks] = __builtin_bit_cast(bf16x8, pk);
    }
}

// weight image -> LDS, 1 KiB per wave instruction
__device__ __forceinline__ void stage_image(const char* __restrict__ img, char* lds, int nblk, int wave, int lane) {
    for (int b = wave; b < nblk; b += NWL)
        __builtin_amdgcn_global_load_lds(GLB_PTR(img + ((size_t)b * 64 + lane) * 16), LDS_PTR(lds + (size_t)b * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// out (M, N) = x (M, K) W^T + b, N % 4 == 0.  LDSW: the whole image (NRT x KS KiB) fits the LDS.
template <int KS, bool LDSW>
__global__ __launch_bounds__(NWL * 64) void k_linear_bf16(const float* __restrict__ x, const char* __restrict__ img, float* __restrict__ out,
                                                          int M, int N, int K, int NRT) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (LDSW) stage_image(img, smem, NRT * KS, wave, lane);
    const int ntile = (M + 15) >> 4;
    for (int tile = blockIdx.x * NWL + wave; tile < ntile; tile += gridDim.x * NWL) {
        const int m = tile * 16 + tok;
        const bool valid = m < M;
        bf16x8 xf[KS];
        x_frags<KS>(x, m, valid, K, g, xf);
        for (int rt = 0; rt < NRT; ++rt) {
            f32x4 acc = f4zero();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const size_t off = ((size_t)(rt * KS + ks) * 64 + lane) * 16;
                const bf16x8 w = LDSW ? *reinterpret_cast<const bf16x8*>(smem + off) : *reinterpret_cast<const bf16x8*>(img + off);
                acc = MFMA(w, xf[ks], acc);
            }
            const int n0 = 16 * rt + 4 * g;
            if (valid && n0 < N) *reinterpret_cast<float4*>(out + (size_t)m * N + n0) = float4{acc[0], acc[1], acc[2], acc[3]};
        }
    }
}

// out (M, D) = LayerNorm(res + x W^T + b; gamma, beta), eps 1e-5 (norm1 of the encoder layer).  DT row tiles cover D (+ pad).
template <int KS, int DT, bool LDSW>
__global__ __launch_bounds__(NWL * 64) void k_linear_res_ln_bf16(const float* __restrict__ x, const char* __restrict__ img,
                                                                 const float* __restrict__ res, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ out, int M, int D) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (LDSW) stage_image(img, smem, DT * KS, wave, lane);
    const int ntile = (M + 15) >> 4;
    for (int tile = blockIdx.x * NWL + wave; tile < ntile; tile += gridDim.x * NWL) {
        const int m = tile * 16 + tok;
        const bool valid = m < M;
        const int mc = valid ? m : 0;
        bf16x8 xf[KS];
        x_frags<KS>(x, m, valid, D, g, xf);
        f32x4 o[DT];
        float4 r4[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            r4[dt] = *reinterpret_cast<const float4*>(res + (size_t)mc * D + (d0 < D ? d0 : 0));
            o[dt] = f4zero();
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const size_t off = ((size_t)(dt * KS + ks) * 64 + lane) * 16;
                const bf16x8 w = LDSW ? *reinterpret_cast<const bf16x8*>(smem + off) : *reinterpret_cast<const bf16x8*>(img + off);
                o[dt] = MFMA(w, xf[ks], o[dt]);
            }
        float s = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            if (16 * dt + 4 * g < D) {
                o[dt][0] += r4[dt].x; o[dt][1] += r4[dt].y; o[dt][2] += r4[dt].z; o[dt][3] += r4[dt].w;
                s += (o[dt][0] + o[dt][1]) + (o[dt][2] + o[dt][3]);
            } else {
                o[dt] = f4zero();
            }
        }
        const float invD = 1.0f / (float)D;
        const float mean = group_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            if (16 * dt + 4 * g < D) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float c = o[dt][r] - mean;
                    q += c * c;
                }
            }
        const float rstd = __builtin_amdgcn_rsqf(group_sum(q) * invD + 1e-5f);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            if (valid && d0 < D) {
                const float4 gm = *reinterpret_cast<const float4*>(gamma + d0), bt = *reinterpret_cast<const float4*>(beta + d0);
                *reinterpret_cast<float4*>(out + (size_t)m * D + d0) =
                    float4{(o[dt][0] - mean) * rstd * gm.x + bt.x, (o[dt][1] - mean) * rstd * gm.y + bt.y,
                           (o[dt][2] - mean) * rstd * gm.z + bt.z, (o[dt][3] - mean) * rstd * gm.w + bt.w};
            }
        }
    }
}

template <class Kern>
int set_lds(fd_ctx* ctx, Kern kern, unsigned long long& mask) {
    if (fd_first_on_device(mask, ctx->device))
        FD_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return FD_OK;
}

int grid_tiles(fd_ctx* ctx, int M, size_t lds) {
    const int ntile = (M + 15) / 16;
    const int per_cu = lds > 76 * 1024 ? 1 : (lds > 50 * 1024 ? 2 : 4);
    return std::max(1, std::min((ntile + NWL - 1) / NWL, ctx->num_cu * per_cu));
}

}  // namespace

// qkv-style projection: out (M, N) = x (M, K) W^T + b from the [N/16 row tiles][ks] image; ks = k-steps of K + 1.
int fd_linear_bf16(fd_ctx* ctx, const float* x, const char* img, float* out, int M, int N, int K, int ks, hipStream_t s) {
    if ((N & 3) || (K & 3) || ks < 1 || ks > 5) return FD_ERR_UNSUPPORTED;
    const int NRT = (N + 15) / 16;
    const size_t bytes = (size_t)NRT * ks * 1024;
    const bool ldsw = bytes <= 150 * 1024;
    const size_t lds = ldsw ? bytes : 0;
    const int grid = grid_tiles(ctx, M, lds);
    static unsigned long long attr[6][2] = {};
#define FD_LIN(KS_)                                                                                                   \
    if (ks == KS_) {                                                                                                  \
        if (ldsw) {                                                                                                   \
            if (int rc = set_lds(ctx, k_linear_bf16<KS_, true>, attr[KS_][1])) return rc;                             \
            hipLaunchKernelGGL((k_linear_bf16<KS_, true>), dim3(grid), dim3(NWL * 64), lds, s, x, img, out, M, N, K, NRT); \
        } else {                                                                                                      \
            hipLaunchKernelGGL((k_linear_bf16<KS_, false>), dim3(grid), dim3(NWL * 64), 0, s, x, img, out, M, N, K, NRT); \
        }                                                                                                             \
    }
    FD_LIN(1) FD_LIN(2) FD_LIN(3) FD_LIN(4) FD_LIN(5)
#undef FD_LIN
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

// out-projection + residual + LayerNorm1: out = LN(res + x W^T + b), x / res / out (M, D); image [dt row tiles][ks].
int fd_linear_res_ln_bf16(fd_ctx* ctx, const float* x, const char* img, const float* res, const float* gamma, const float* beta,
                          float* out, int M, int D, int ks, int dt, hipStream_t s) {
    if (D & 3) return FD_ERR_UNSUPPORTED;
    const size_t lds = (size_t)dt * ks * 1024;
    const int grid = grid_tiles(ctx, M, lds);
    static unsigned long long attr[16] = {};
    int slot = 0;
#define FD_LRL(KS_, DT_)                                                                                             \
    if (ks == KS_ && dt == DT_) {                                                                                    \
        if (int rc = set_lds(ctx, k_linear_res_ln_bf16<KS_, DT_, true>, attr[slot])) return rc;                      \
        hipLaunchKernelGGL((k_linear_res_ln_bf16<KS_, DT_, true>), dim3(grid), dim3(NWL * 64), lds, s, x, img, res, gamma, beta, \
                           out, M, D);                                                                               \
        FD_LAUNCH_CHECK(ctx);                                                                                        \
        return FD_OK;                                                                                                \
    }                                                                                                                \
    ++slot;
    FD_LRL(1, 1) FD_LRL(1, 2) FD_LRL(2, 3) FD_LRL(2, 4) FD_LRL(3, 5) FD_LRL(3, 6) FD_LRL(4, 7) FD_LRL(4, 8) FD_LRL(5, 9)
#undef FD_LRL
    return FD_ERR_UNSUPPORTED;
}
