// fd_attn_wide.hip -- bf16 MFMA self-attention for head_dim 8 .. 32 (the sizes most transformers use: d_model 64 / 8 heads,
// 128 / 8, 96 / 12 ...).  fd_attn_bf16.hip and the persistent kernel pack TWO heads of head_dim <= 7 into the 16 k-slots of
// one v_mfma_f32_16x16x16_bf16 and hide the softmax denominator / shift in the free slot; a head of 8 or more dims fills
// the slots, so this kernel gives every head its own contraction:
//   HDS = 16: head_dim <= 16, S^T = K Q^T by the K=16 MFMA;   HDS = 32: head_dim <= 32, by the K=32 MFMA.
// One workgroup = (series, head, slice of the query tiles), 4 waves.  K and V^T of the head for the whole series are staged
// once in LDS as bf16 MFMA fragments (T = 1024, HDS = 32: 128 KiB); a wave's unit is two consecutive query tiles against all
// keys: pass 1 = exact row maxima (QK^T MFMAs + max), pass 2 = exp2(S - max) with -max riding in the MFMA's C operand, P
// packed in registers as the B operand of O^T = V^T P^T, row sums on the VALU (no free slot to take them from the matrix
// pipe).  fp32 accumulate / maxima / sums; only the MFMA operands are bf16 -- the same arithmetic contract as the other two
// attention kernels (oracle parity <= 1e-2 of scale in tests/test_gpu_widths.py).
// Reference arithmetic: torch.nn.MultiheadAttention inside nn.TransformerEncoderLayer (src/fdiff/models/score_models.py:57-62),
// eval mode: softmax(q k^T / sqrt(head_dim)) v per head.
#include <hip/hip_runtime.h>

#include "fd_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

namespace {

constexpr float kNegBig = -1.0e30f;
constexpr int NWW = 4, NQ = 2;

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
    u32x4 r = {cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3]), cvt_pk_bf16(b[0], b[1]), cvt_pk_bf16(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, r);
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
__device__ __forceinline__ float group_max(float v) {      // over the 4 lane groups (rows of 16 lanes) of one query
    float a, b;
    swap32(v, a, b);
    swap16(fmaxf(a, b), a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float group_sum(float v) {
    float a, b;
    swap32(v, a, b);
    swap16(a + b, a, b);
    return a + b;
}

// operand of the score MFMA for one lane: HDS = 16 -> 4 bf16 (k-slots 4g .. 4g+3), HDS = 32 -> 8 bf16 (k-slots 8g .. 8g+7)
template <int HDS>
struct KFrag;
template <>
struct KFrag<16> {
    typedef s16x4 type;
    static constexpr int bytes = 8, per_lane = 4;
};
template <>
struct KFrag<32> {
    typedef bf16x8 type;
    static constexpr int bytes = 16, per_lane = 8;
};
__device__ __forceinline__ f32x4 score_mfma(s16x4 k, s16x4 q, f32x4 c) { return MFMA16(k, q, c); }
__device__ __forceinline__ f32x4 score_mfma(bf16x8 k, bf16x8 q, f32x4 c) { return MFMA32(k, q, c); }

// qkv (B*T, 3D) fp32 rows [q | k | v]; out (B*T, D) fp32.  grid = B * H * slices workgroups of 256 threads.
template <int HDS>
__global__ __launch_bounds__(NWW * 64) void k_attention_wide(const float* __restrict__ qkv, float* __restrict__ out, int T, int H, int hd,
                                                             int D, float qscale, int du_per_block, int slices) {
    using KF = KFrag<HDS>;
    using kfrag_t = typename KF::type;
    constexpr int RT = HDS / 16;                  // 16-row tiles of the head's dims in V^T / O^T
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = blockIdx.x % slices, head = (blockIdx.x / slices) % H, b = blockIdx.x / (slices * H);
    const int KT = (T + 15) >> 4, NJ = (KT + 1) >> 1, NTOK = KT * 16;
    char* const kbf = smem;                                        // [NTOK][4 g][KF::bytes]
    char* const vbf = smem + (size_t)NTOK * 4 * KF::bytes;         // [NJ][RT][4 g][16 dim rows][16 B]: (half, r) -> key (2jj+half)*16 + 4g + r
    const float* base = qkv + (size_t)b * T * 3 * D;

    // ---- stage K (one thread per (token, lane group)) and V^T.  head_dim % 4 == 0 (every common size): 16-byte loads of four
    //      consecutive dims of one token (the fp32 rows are 16-byte aligned: D % 4 == 0); V^T is then scattered into its
    //      [key block][row tile][lane group][dim row][8 keys] layout by 2-byte LDS stores.  Otherwise guarded scalar loads.
    const bool v4 = (hd & 3) == 0;
    for (int i = threadIdx.x; i < NTOK * 4; i += NWW * 64) {
        const int t = i >> 2, gq = i & 3;
        float kv[KF::per_lane];
#pragma unroll
        for (int e = 0; e < KF::per_lane; ++e) kv[e] = 0.f;
        if (t < T) {
            const float* kp = base + (size_t)t * 3 * D + D + head * hd + KF::per_lane * gq;
            if (v4) {
#pragma unroll
                for (int e4 = 0; e4 < KF::per_lane / 4; ++e4)
                    if (KF::per_lane * gq + 4 * e4 < hd) {
                        const float4 q4 = *reinterpret_cast<const float4*>(kp + 4 * e4);
                        kv[4 * e4] = q4.x; kv[4 * e4 + 1] = q4.y; kv[4 * e4 + 2] = q4.z; kv[4 * e4 + 3] = q4.w;
                    }
            } else {
#pragma unroll
                for (int e = 0; e < KF::per_lane; ++e)
                    if (KF::per_lane * gq + e < hd) kv[e] = kp[e];
            }
        }
        if constexpr (HDS == 16) {
            *reinterpret_cast<u32x2*>(kbf + (size_t)i * 8) = u32x2{cvt_pk_bf16(kv[0], kv[1]), cvt_pk_bf16(kv[2], kv[3])};
        } else {
            *reinterpret_cast<u32x4*>(kbf + (size_t)i * 16) = u32x4{cvt_pk_bf16(kv[0], kv[1]), cvt_pk_bf16(kv[2], kv[3]),
                                                                   cvt_pk_bf16(kv[4 % KF::per_lane], kv[5 % KF::per_lane]),
                                                                   cvt_pk_bf16(kv[6 % KF::per_lane], kv[7 % KF::per_lane])};
        }
    }
    if (v4) {
        // one thread per (token slot, group of 4 dims): dims d .. d+3 of key t -> element e = (half, r) of rows d .. d+3
        // (every slot of every 32-key block is written, also the 16 of a missing odd tile: P is 0 there, V must be finite)
        for (int i = threadIdx.x; i < NJ * 32 * (HDS / 4); i += NWW * 64) {
            const int t = i / (HDS / 4), d = 4 * (i - t * (HDS / 4));
            float4 v = {0.f, 0.f, 0.f, 0.f};
            if (t < T && d < hd) v = *reinterpret_cast<const float4*>(base + (size_t)t * 3 * D + 2 * D + head * hd + d);
            const int jj = t >> 5, half = (t >> 4) & 1, gg = (t >> 2) & 3, r = t & 3;
            __bf16* dst = reinterpret_cast<__bf16*>(vbf + ((size_t)(((jj * RT + (d >> 4)) * 4 + gg) * 16 + (d & 15))) * 16) + (half * 4 + r);
            dst[0] = (__bf16)v.x; dst[8] = (__bf16)v.y; dst[16] = (__bf16)v.z; dst[24] = (__bf16)v.w;     // next dim row = +16 bytes
        }
    } else {
    for (int i = threadIdx.x; i < NJ * RT * 64; i += NWW * 64) {
        const int row = i & 15, gg = (i >> 4) & 3, rt = (i >> 6) % RT, jj = i / (64 * RT);
        const int d = 16 * rt + row;
        float vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = (2 * jj + (e >> 2)) * 16 + 4 * gg + (e & 3);
            vv[e] = (t < T && d < hd) ? base[(size_t)t * 3 * D + 2 * D + head * hd + d] : 0.f;
        }
        *reinterpret_cast<u32x4*>(vbf + (size_t)i * 16) = u32x4{cvt_pk_bf16(vv[0], vv[1]), cvt_pk_bf16(vv[2], vv[3]),
                                                               cvt_pk_bf16(vv[4], vv[5]), cvt_pk_bf16(vv[6], vv[7])};
    }
    }
    __syncthreads();

    f32x4 cmask;                                                   // keys beyond T in the ragged last tile
#pragma unroll
    for (int r = 0; r < 4; ++r) cmask[r] = ((KT - 1) * 16 + 4 * g + r >= T) ? kNegBig : 0.f;
    const f32x4 allneg = {kNegBig, kNegBig, kNegBig, kNegBig};
    auto kfrag = [&](int kt) { return *reinterpret_cast<const kfrag_t*>(kbf + ((size_t)(kt * 16 + tok) * 4 + g) * KF::bytes); };
    auto vfrag = [&](int jb, int rt) { return *reinterpret_cast<const bf16x8*>(vbf + ((size_t)((jb * RT + rt) * 4 + g) * 16 + tok) * 16); };
    const int DUS = (KT + NQ - 1) / NQ;
    const int du0 = slice * du_per_block, du1 = min(DUS, du0 + du_per_block);
    for (int du = du0 + wave; du < du1; du += NWW) {
        // Q^T operands of the unit's two query tiles (the second may not exist: it re-reads the first and is not written)
        kfrag_t qf[NQ];
        int qt[NQ];
        bool qv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            qv[q] = du * NQ + q < KT;
            qt[q] = qv[q] ? du * NQ + q : du * NQ;
            const int t = qt[q] * 16 + tok;
            float v[KF::per_lane];
#pragma unroll
            for (int e = 0; e < KF::per_lane; ++e) {
                const int d = KF::per_lane * g + e;
                v[e] = (t < T && d < hd) ? base[(size_t)t * 3 * D + head * hd + d] * qscale : 0.f;
            }
            if constexpr (HDS == 16) {
                const u32x2 w = {cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
                qf[q] = __builtin_bit_cast(kfrag_t, w);
            } else {
                const u32x4 w = {cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4 % KF::per_lane], v[5 % KF::per_lane]),
                                 cvt_pk_bf16(v[6 % KF::per_lane], v[7 % KF::per_lane])};
                qf[q] = __builtin_bit_cast(kfrag_t, w);
            }
        }
        // pass 1: exact row maxima of the (base-2, scaled) scores
        float bm[NQ] = {kNegBig, kNegBig};
        for (int kt = 0; kt < KT; ++kt) {
            const kfrag_t kf = kfrag(kt);
            const f32x4 c0 = (kt == KT - 1) ? cmask : f4zero();
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x4 v = score_mfma(kf, qf[q], c0);
                bm[q] = fmaxf(fmaxf(fmaxf(bm[q], v[0]), v[1]), fmaxf(v[2], v[3]));
            }
        }
        f32x4 negm[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float m = group_max(bm[q]);
            negm[q] = f32x4{-m, -m, -m, -m};
        }
        // pass 2: P = exp2(S - max) per pair of key tiles (= one V^T block), row sums, O^T += V^T P^T
        float ls[NQ] = {0.f, 0.f};
        f32x4 o2[NQ][RT];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) o2[q][rt] = f4zero();
        for (int jb = 0; jb < NJ; ++jb) {
            const int ka = 2 * jb, kb = (2 * jb + 1 < KT) ? 2 * jb + 1 : ka;
            const kfrag_t kfa = kfrag(ka), kfb = kfrag(kb);
            bf16x8 vf[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) vf[rt] = vfrag(jb, rt);
            const f32x4 ma = (ka == KT - 1) ? cmask : f4zero();
            const f32x4 mb = (2 * jb + 1 >= KT) ? allneg : ((kb == KT - 1) ? cmask : f4zero());
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                f32x4 pa = score_mfma(kfa, qf[q], ma + negm[q]);
                f32x4 pb = score_mfma(kfb, qf[q], mb + negm[q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pa[r] = __builtin_amdgcn_exp2f(pa[r]);
                    pb[r] = __builtin_amdgcn_exp2f(pb[r]);
                }
                ls[q] += ((pa[0] + pa[1]) + (pa[2] + pa[3])) + ((pb[0] + pb[1]) + (pb[2] + pb[3]));
                const bf16x8 pk = pack8(pa, pb);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) o2[q][rt] = MFMA32(vf[rt], pk, o2[q][rt]);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float inv = __builtin_amdgcn_rcpf(group_sum(ls[q]));
            const int t = qt[q] * 16 + tok;
            if (qv[q] && t < T) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int d = 16 * rt + 4 * g + r;
                        if (d < hd) out[((size_t)b * T + t) * D + head * hd + d] = o2[q][rt][r] * inv;
                    }
            }
        }
    }
}

}  // namespace

// head_dim 8 .. 32 from the packed projections (B*T, 3D).  FD_ERR_UNSUPPORTED when K / V^T of one head and series exceed the LDS.
int fd_attention_bf16_wide(fd_ctx* ctx, const float* qkv, float* out, int B, int T, int H, int hd, hipStream_t s) {
    if (hd < 1 || hd > 32) return FD_ERR_UNSUPPORTED;
    const int HDS = hd <= 16 ? 16 : 32;
    const int KT = (T + 15) / 16, NJ = (KT + 1) / 2, D = H * hd;
    const size_t lds = (size_t)KT * 16 * 4 * (HDS == 16 ? 8 : 16) + (size_t)NJ * (HDS / 16) * 1024;
    if (lds > 160 * 1024) return FD_ERR_UNSUPPORTED;
    const int DUS = (KT + NQ - 1) / NQ;
    // query slices: enough workgroups for two per CU, at least one unit per wave in a slice
    int slices = 1;
    while ((long long)B * H * slices < 2LL * ctx->num_cu && slices * 2 * NWW <= DUS) slices *= 2;
    const int du_per_block = (DUS + slices - 1) / slices;
    const float qscale = 1.4426950408889634f / sqrtf((float)hd);
    const dim3 grid((unsigned)((long long)B * H * slices)), block(NWW * 64);
    static unsigned long long attr[2] = {};
    if (HDS == 16) {
        if (fd_first_on_device(attr[0], ctx->device))
            FD_HIP(ctx, hipFuncSetAttribute((const void*)k_attention_wide<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k_attention_wide<16>, grid, block, lds, s, qkv, out, T, H, hd, D, qscale, du_per_block, slices);
    } else {
        if (fd_first_on_device(attr[1], ctx->device))
            FD_HIP(ctx, hipFuncSetAttribute((const void*)k_attention_wide<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k_attention_wide<32>, grid, block, lds, s, qkv, out, T, H, hd, D, qscale, du_per_block, slices);
    }
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}
