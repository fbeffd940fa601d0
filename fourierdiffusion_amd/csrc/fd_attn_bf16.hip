// fd_attn_bf16.hip -- bf16 MFMA self-attention for the step-by-step path (series too long for the persistent
// kernel: T > 256).  Same formulation as the attention units of fd_mega.hip, as a standalone kernel:
//   * one workgroup = (series, head pair, slice of the query tiles); K and V^T of the pair for the WHOLE series are
//     converted to bf16 fragments in LDS once (T = 1024: 64 KiB) and every wave then walks 128-key blocks from LDS;
//   * unit = 2 consecutive query tiles x both heads of the pair: S^T = K Q^T by the K=16 MFMA (two heads share each K
//     fragment, Q masked per head), exact two-pass online softmax per key block (row max, then exp2(S - max) with -max
//     riding in the MFMA C operand), P re-packed in registers as the B operand of O^T = V^T P^T; the softmax
//     denominator is the ones row V^T carries in the free dim slot hd (so head_dim <= 7 here).
// Reference arithmetic: torch.nn.MultiheadAttention inside nn.TransformerEncoderLayer
// (src/fdiff/models/score_models.py:57-62), eval mode, softmax(q k^T / sqrt(hd)) v per head.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fd_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

namespace {

constexpr float kNegBig = -1.0e30f;
constexpr int NW = 8, NTH = NW * 64, NQ = 2;

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

// row-max over the 16 lanes of a row (DPP rotations)
template <int N>
__device__ __forceinline__ float row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_max16(float v) {
    v = fmaxf(v, row_ror<8>(v));
    v = fmaxf(v, row_ror<4>(v));
    v = fmaxf(v, row_ror<2>(v));
    v = fmaxf(v, row_ror<1>(v));
    return v;
}

// Weight fragments of one head pair for the fused projections (KS1 1-KiB blocks each, the persistent kernel's images:
// W_q carries log2(e)/sqrt(hd), biases ride in k-slot D against the activation's constant 1.0).
struct fd_attn_w {
    const char* wk;
    const char* wv;
    const char* wq;
};

// KS1 == 0: `in` = packed projections qkv (B*T, 3D) fp32 rows [q | k | v].
// KS1 > 0 : `in` = the layer input h (B*T, D) fp32; Q, K, V of the pair are projected inside the kernel by MFMA (one
//           launch and the (B*T, 3D) round trip less per layer).
// out (B*T, D) fp32.  grid (query slices, head pairs, B), 512 threads.
template <int KS1>
__global__ __launch_bounds__(NTH, 2) void k_attention_bf16(const float* __restrict__ in, float* __restrict__ out, int T,
                                                           int H, int hd, int D, float qscale, int du_per_block, int exact_only,
                                                           fd_attn_w wimg, size_t pair_stride) {
    constexpr bool PROJ = KS1 > 0;
    const float* __restrict__ qkv = in;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KT = (T + 15) >> 4, NJ = (KT + 1) >> 1, NTOK = KT * 16;
    const int pair = blockIdx.y, b = blockIdx.z;
    char* const kbf = smem;                       // [NTOK][4 g][8 B]: lane group g = 2*hs + (d >> 2), element d & 3
    char* const vbf = smem + (size_t)NTOK * 32;   // [NJ][4 g][16 dim slots][16 B]: (half, r) -> token (2jj+half)*16 + 4g + r
    unsigned* const kmax = reinterpret_cast<unsigned*>(vbf + (size_t)NJ * 1024);   // [2] max_j |k_j|^2 per head (bits)
    const float* base = qkv + (size_t)b * T * (PROJ ? 1 : 3) * D;
    // activation B fragment of token tile `tile` (PROJ): lane (tok, g) holds features 32ks + 8g .. +7, 1.0 in slot D
    auto xfrag = [&](int tile, int ks) -> bf16x8 {
        const int t = tile * 16 + tok, k0 = 32 * ks + 8 * g;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (t < T) {
            if (k0 + 8 <= D) {
                const float4 a = *reinterpret_cast<const float4*>(base + (size_t)t * D + k0);
                const float4 c = *reinterpret_cast<const float4*>(base + (size_t)t * D + k0 + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = k0 + e;
                    if (k < D) v[e] = base[(size_t)t * D + k];
                    else if (k == D) v[e] = 1.0f;
                }
            }
        }
        const u32x4 pk = {cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7])};
        return __builtin_bit_cast(bf16x8, pk);
    };
    auto wfrag = [&](const char* img, int ks) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(img + (size_t)pair * pair_stride + ((size_t)ks * 64 + lane) * 16);
    };

    // ---- stage K and V^T of this (series, pair) as bf16 fragments; every byte of both regions is written here
    if (threadIdx.x < 2) kmax[threadIdx.x] = 0u;
    __syncthreads();
    if (PROJ) {
        // K^T = W_k x^T (C tile rows = the pair's 16 dim slots) and V = x W_v^T (C tile rows = tokens: already the
        // V^T A-fragment layout; its ones row comes from the bias slot) per token tile, exactly as in fd_mega.hip
        bf16x8 wkf[PROJ ? KS1 : 1], wvf[PROJ ? KS1 : 1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            wkf[ks] = wfrag(wimg.wk, ks);
            wvf[ks] = wfrag(wimg.wv, ks);
        }
        for (int kt = wave; kt < KT; kt += NW) {
            f32x4 a = f4zero(), c = f4zero();
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const bf16x8 xf = xfrag(kt, ks);
                a = MFMA(wkf[ks], xf, a);
                c = MFMA(xf, wvf[ks], c);
            }
            *reinterpret_cast<u32x2*>(kbf + ((size_t)(kt * 16 + tok) * 4 + g) * 8) = u32x2{cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3])};
            char* dst = vbf + ((size_t)((kt >> 1) * 4 + g) * 16 + tok) * 16;
            *reinterpret_cast<u32x2*>(dst + 8 * (kt & 1)) = u32x2{cvt_pk_bf16(c[0], c[1]), cvt_pk_bf16(c[2], c[3])};
            if ((KT & 1) && kt == KT - 1) *reinterpret_cast<u32x2*>(dst + 8) = u32x2{0u, 0u};
            float n2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
            float ea, eb;
            swap16(n2, ea, eb);
            n2 = row_max16(ea + eb);
            if (tok == 0 && (g & 1) == 0)
                __hip_atomic_fetch_max(&kmax[g >> 1], __builtin_bit_cast(unsigned, n2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
    // K: one thread per (token, lane group gq): the 4 dims 4(gq&1)..+3 of head gq>>1 -> one 8-byte row
    for (int i = threadIdx.x; i < NTOK * 4; i += NTH) {
        const int t = i >> 2, gq = i & 3, head = 2 * pair + (gq >> 1);
        float kv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 4 * (gq & 1) + r;
            kv[r] = (t < T && d < hd && head < H) ? base[(size_t)t * 3 * D + D + head * hd + d] : 0.f;
        }
        *reinterpret_cast<u32x2*>(kbf + (size_t)i * 8) = u32x2{cvt_pk_bf16(kv[0], kv[1]), cvt_pk_bf16(kv[2], kv[3])};
        // |k_t|^2 of head gq>>1: the two threads (gq even / odd) holding a token's halves are lane neighbours
        float n2 = kv[0] * kv[0] + kv[1] * kv[1] + kv[2] * kv[2] + kv[3] * kv[3];
        n2 += __shfl_xor(n2, 1);
        // wave maximum per head (lanes with gq>>1 equal): xor-shuffles over the other lane bits
        n2 = fmaxf(n2, __shfl_xor(n2, 4));
        n2 = fmaxf(n2, __shfl_xor(n2, 8));
        n2 = fmaxf(n2, __shfl_xor(n2, 16));
        n2 = fmaxf(n2, __shfl_xor(n2, 32));
        if ((threadIdx.x & 0x3d) == 0)     // lanes 0 (head 0) and 2 (head 1) of each wave
            __hip_atomic_fetch_max(&kmax[gq >> 1], __builtin_bit_cast(unsigned, n2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // V^T: one thread per (32-key block jj, lane group gg, dim slot row): 8 keys (half, r) of that dim -> one 16-byte
    // vector.  Slot hd of each head is the ones row: the P V MFMAs then also produce sum_j P (softmax denominator).
    for (int i = threadIdx.x; i < NJ * 64; i += NTH) {
        const int row = i & 15, gg = (i >> 4) & 3, jj = i >> 6;
        const int hs = row >> 3, d = row & 7, head = 2 * pair + hs;
        float vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = (2 * jj + (e >> 2)) * 16 + 4 * gg + (e & 3);
            float x = 0.f;
            if (t < T && head < H) x = (d < hd) ? base[(size_t)t * 3 * D + 2 * D + head * hd + d] : (d == hd ? 1.0f : 0.f);
            vv[e] = x;
        }
        *reinterpret_cast<u32x4*>(vbf + (size_t)i * 16) = u32x4{cvt_pk_bf16(vv[0], vv[1]), cvt_pk_bf16(vv[2], vv[3]),
                                                               cvt_pk_bf16(vv[4], vv[5]), cvt_pk_bf16(vv[6], vv[7])};
    }
    }
    __syncthreads();
    bf16x8 wqf[PROJ ? KS1 : 1];
    if (PROJ) {
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) wqf[ks] = wfrag(wimg.wq, ks);
    }

    const bool lo_grp = (g >> 1) == 0;
    const int myhead = 2 * pair + (g >> 1);
    f32x4 cmask;                                                     // keys beyond T in the ragged last tile
#pragma unroll
    for (int r = 0; r < 4; ++r) cmask[r] = ((KT - 1) * 16 + 4 * g + r >= T) ? kNegBig : 0.f;
    const int DUS = (KT + NQ - 1) / NQ;
    const int du0 = blockIdx.x * du_per_block, du1 = min(DUS, du0 + du_per_block);
    for (int du = du0 + wave; du < du1; du += NW) {
        int qt[NQ];
        bool qv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            qv[q] = du * NQ + q < KT;
            qt[q] = qv[q] ? du * NQ + q : du * NQ;
        }
        // Q^T B operands: lane (query tok, g) holds dims 4(g&1)..+3 of head g>>1, pre-scaled by log2(e)/sqrt(hd)
        s16x4 qb[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int t = qt[q] * 16 + tok;
            float qvv[4];
            if (PROJ) {
                f32x4 qa = f4zero();
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) qa = MFMA(wqf[ks], xfrag(qt[q], ks), qa);
#pragma unroll
                for (int r = 0; r < 4; ++r) qvv[r] = qa[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 4 * (g & 1) + r;
                    qvv[r] = (t < T && d < hd && myhead < H) ? base[(size_t)t * 3 * D + myhead * hd + d] * qscale : 0.f;
                }
            }
            const unsigned q01 = cvt_pk_bf16(qvv[0], qvv[1]), q23 = cvt_pk_bf16(qvv[2], qvv[3]);
            const u32x2 qe = {lo_grp ? q01 : 0u, lo_grp ? q23 : 0u};
            const u32x2 qo = {lo_grp ? 0u : q01, lo_grp ? 0u : q23};
            qb[q][0] = __builtin_bit_cast(s16x4, qe);
            qb[q][1] = __builtin_bit_cast(s16x4, qo);
        }
        // Softmax shift: the bound |q| max_j |k_j| >= max_j q.k_j (2 % headroom for the bf16 rounding) instead of a
        // max pass; any shift cancels in P V / sum P.  A row sum below 2^-100 (bound > max + ~100) sends the unit
        // through the exact two-pass form (see fd_mega.hip, tests/test_gpu_baseline_shapes.py).
        float bq[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // this lane's 4 dims of head g>>1, already scaled and rounded like the MFMA operand
            const u32x2 qraw = lo_grp ? __builtin_bit_cast(u32x2, qb[q][0]) : __builtin_bit_cast(u32x2, qb[q][1]);
            const float q0 = __builtin_bit_cast(float, qraw[0] << 16), q1 = __builtin_bit_cast(float, qraw[0] & 0xffff0000u);
            const float q2 = __builtin_bit_cast(float, qraw[1] << 16), q3 = __builtin_bit_cast(float, qraw[1] & 0xffff0000u);
            float ea, eb;
            swap16(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3, ea, eb);
            const float k2 = __builtin_bit_cast(float, kmax[g >> 1]);
            swap32(__builtin_sqrtf((ea + eb) * k2) * 1.02f, bq[q][0], bq[q][1]);
        }
        float m2[NQ][2];
        f32x4 o2[NQ][2];
        auto run_unit = [&](auto exact_c) {
        constexpr bool EXACT = decltype(exact_c)::value;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                m2[q][hs] = EXACT ? kNegBig : bq[q][hs];
                o2[q][hs] = f4zero();
            }
        // One 128-key block.  FULL (8 key tiles) and LAST (the block holds the series' final, possibly ragged tile) are
        // compile-time: with run-time tile guards every MFMA sits in its own basic block and the hand-made software
        // pipeline below falls apart (measured 3x slower).
        auto key_block = [&](int kb, auto full_c, auto last_c) {
            constexpr bool FULL = decltype(full_c)::value, LAST = decltype(last_c)::value;
            s16x4 kf[8];
            bf16x8 vf[4];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (FULL || kb + j < KT) kf[j] = *reinterpret_cast<const s16x4*>(kbf + ((size_t)((kb + j) * 16 + tok) * 4 + g) * 8);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if (FULL || (kb >> 1) + jj < NJ)
                    vf[jj] = *reinterpret_cast<const bf16x8*>(vbf + ((size_t)(((kb >> 1) + jj) * 4 + g) * 16 + tok) * 16);
            const int nk = FULL ? 8 : min(8, KT - kb);
            constexpr int NKT = 16 * NQ;          // score tiles per block: k = ((hs*4 + jj)*NQ + q)*2 + jl, key tile 2jj+jl
            // pass 1 (exact path only): row maxima of this key block (software-pipelined by hand, see fd_mega.hip)
            float bm[NQ][2];
#pragma unroll
            for (int q = 0; q < NQ; ++q) bm[q][0] = bm[q][1] = kNegBig;
            if (EXACT) {
                constexpr int LAG = 3;
                f32x4 t4[NKT];
#pragma unroll
                for (int k = 0; k < NKT + LAG; ++k) {
                    if (k < NKT) {
                        const int jl = k & 1, q = (k >> 1) % NQ, jj = ((k >> 1) / NQ) & 3, hs = (k >> 1) / NQ >> 2;
                        const int j = 2 * jj + jl;
                        if (j < nk) t4[k] = MFMA16(kf[j], qb[q][hs], (LAST && kb + j == KT - 1) ? cmask : f4zero());
                    }
                    if (k >= LAG) {
                        const int e = k - LAG;
                        const int jl = e & 1, q = (e >> 1) % NQ, jj = ((e >> 1) / NQ) & 3, hs = (e >> 1) / NQ >> 2;
                        if (2 * jj + jl < nk) {
                            const f32x4 v = t4[e];
                            float bb = bm[q][hs];
                            bb = fmaxf(fmaxf(bb, v[0]), v[1]);
                            bb = fmaxf(fmaxf(bb, v[2]), v[3]);
                            bm[q][hs] = bb;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            f32x4 negm[NQ][2], clast[NQ][2];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) {
                    float mnew = m2[q][hs];
                    if (EXACT) {
                        mnew = fmaxf(m2[q][hs], group_max(bm[q][hs]));
                        const float alpha = __builtin_amdgcn_exp2f(m2[q][hs] - mnew);
                        o2[q][hs] = o2[q][hs] * alpha;
                        m2[q][hs] = mnew;
                    }
                    negm[q][hs] = f32x4{-mnew, -mnew, -mnew, -mnew};
                    clast[q][hs] = cmask - mnew;
                }
            // pass 2: P = exp2(S - max), packed to bf16 B fragments, then P V
            {
                constexpr int LAG = 2;
                f32x4 pe[NKT];
                bf16x8 pk[NKT / 2];
#pragma unroll
                for (int k = 0; k < NKT + 2 * LAG; ++k) {
                    if (k < NKT) {
                        const int jl = k & 1, q = (k >> 1) % NQ, jj = ((k >> 1) / NQ) & 3, hs = (k >> 1) / NQ >> 2;
                        const int j = 2 * jj + jl;
                        if (j < nk) pe[k] = MFMA16(kf[j], qb[q][hs], (LAST && kb + j == KT - 1) ? clast[q][hs] : negm[q][hs]);
                        else pe[k] = f4zero();
                    }
                    if (k >= LAG && k - LAG < NKT) {
                        const int e = k - LAG;
                        const int jl = e & 1, jj = ((e >> 1) / NQ) & 3;
                        if (2 * jj + jl < nk) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) pe[e][r] = __builtin_amdgcn_exp2f(pe[e][r]);
                        }
                        if (jl) pk[e >> 1] = pack8(pe[e - 1], pe[e]);
                    }
                    if (k >= 2 * LAG && ((k - 2 * LAG) & 1)) {
                        const int e = k - 2 * LAG;
                        const int q = (e >> 1) % NQ, jj = ((e >> 1) / NQ) & 3, hs = (e >> 1) / NQ >> 2;
                        if (2 * jj < nk) o2[q][hs] = MFMA(vf[jj], pk[e >> 1], o2[q][hs]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // The series' last, partial key block (1-7 tiles): a rolled loop over key-tile PAIRS with a static body (a
        // missing odd tile re-reads the previous one and is masked through the C operand) instead of the guarded
        // unrolled pipeline -- same finding as in fd_mega.hip's run-time-shape path.
        auto ragged_block = [&](int kb) {
            const f32x4 allneg = {kNegBig, kNegBig, kNegBig, kNegBig};
            auto kfrag = [&](int kt) { return *reinterpret_cast<const s16x4*>(kbf + ((size_t)(kt * 16 + tok) * 4 + g) * 8); };
            f32x4 negm[NQ][2];
            if (EXACT) {
                float bm[NQ][2];
#pragma unroll
                for (int q = 0; q < NQ; ++q) bm[q][0] = bm[q][1] = kNegBig;
                for (int kt = kb; kt < KT; ++kt) {
                    const s16x4 kfa = kfrag(kt);
                    const f32x4 ca = (kt == KT - 1) ? cmask : f4zero();
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int hs = 0; hs < 2; ++hs) {
                            const f32x4 v = MFMA16(kfa, qb[q][hs], ca);
                            bm[q][hs] = fmaxf(fmaxf(fmaxf(bm[q][hs], v[0]), v[1]), fmaxf(v[2], v[3]));
                        }
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int hs = 0; hs < 2; ++hs) {
                        const float mnew = fmaxf(m2[q][hs], group_max(bm[q][hs]));
                        const float alpha = __builtin_amdgcn_exp2f(m2[q][hs] - mnew);
                        o2[q][hs] = o2[q][hs] * alpha;
                        m2[q][hs] = mnew;
                    }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) negm[q][hs] = f32x4{-m2[q][hs], -m2[q][hs], -m2[q][hs], -m2[q][hs]};
            for (int jb = kb >> 1; jb < NJ; ++jb) {
                const int ka = 2 * jb, kb2 = (2 * jb + 1 < KT) ? 2 * jb + 1 : ka;
                const s16x4 kfa = kfrag(ka), kfb = kfrag(kb2);
                const bf16x8 vfj = *reinterpret_cast<const bf16x8*>(vbf + ((size_t)(jb * 4 + g) * 16 + tok) * 16);
                const f32x4 ma = (ka == KT - 1) ? cmask : f4zero();
                const f32x4 mb = (2 * jb + 1 >= KT) ? allneg : ((kb2 == KT - 1) ? cmask : f4zero());
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int hs = 0; hs < 2; ++hs) {
                        f32x4 pa = MFMA16(kfa, qb[q][hs], ma + negm[q][hs]);
                        f32x4 pb = MFMA16(kfb, qb[q][hs], mb + negm[q][hs]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pa[r] = __builtin_amdgcn_exp2f(pa[r]);
                            pb[r] = __builtin_amdgcn_exp2f(pb[r]);
                        }
                        o2[q][hs] = MFMA(vfj, pack8(pa, pb), o2[q][hs]);
                    }
            }
        };
        {
            int kb = 0;
            for (; kb + 8 < KT; kb += 8) key_block(kb, std::true_type{}, std::false_type{});
            if (kb + 8 == KT) key_block(kb, std::true_type{}, std::true_type{});
            else ragged_block(kb);
        }
        };
        // row sums of P (the ones row): hd in [4,7]: register hd-4 of the odd lane group; hd < 4: register hd of the even one
        float lrow[NQ];
        auto row_sums = [&]() -> bool {
            bool bad = false;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float cand = lo_grp ? o2[q][0][0] : o2[q][1][0];
#pragma unroll
                for (int r = 1; r < 4; ++r) cand = ((hd & 3) == r) ? (lo_grp ? o2[q][0][r] : o2[q][1][r]) : cand;
                float row_even, row_odd;
                swap16(cand, row_even, row_odd);
                lrow[q] = (hd >= 4) ? row_odd : row_even;
                bad |= !(lrow[q] > 7.8e-31f);
            }
            return bad;
        };
        if (exact_only) {
            run_unit(std::true_type{});
            (void)row_sums();
        } else {
            run_unit(std::false_type{});
            if (__builtin_amdgcn_ballot_w64(row_sums()) != 0ull) {
                run_unit(std::true_type{});
                (void)row_sums();
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float o_sel[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o_sel[r] = lo_grp ? o2[q][0][r] : o2[q][1][r];
            const float inv = 1.0f / lrow[q];
            const int t = qt[q] * 16 + tok;
            if (qv[q] && t < T && myhead < H) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 4 * (g & 1) + r;
                    if (d < hd) out[((size_t)b * T + t) * D + myhead * hd + d] = o_sel[r] * inv;
                }
            }
        }
    }
}

}  // namespace

// bf16 MFMA attention.  wk/wv/wq == nullptr: `in` holds the packed projections (B*T, 3D); otherwise `in` is the layer
// input (B*T, D) and the images are the persistent kernel's per-layer W_k / W_v / W_q fragment blocks (ks1 blocks per
// head pair).  Returns FD_ERR_UNSUPPORTED when the shape does not fit (head_dim > 7, K/V^T of one series exceed the
// LDS, or no instantiation for ks1): the caller then runs the unfused / exact-f32 path.
int fd_attention_bf16(fd_ctx* ctx, const float* in, float* out, int B, int T, int H, int hd, hipStream_t s, const char* wk,
                      const char* wv, const char* wq, int ks1) {
    const int KT = (T + 15) / 16, NJ = (KT + 1) / 2, D = H * hd;
    const size_t lds = (size_t)KT * 16 * 32 + (size_t)NJ * 1024 + 16;
    if (hd > 7 || lds > 160 * 1024) return FD_ERR_UNSUPPORTED;
    const bool proj = wk != nullptr;
    if (proj && ks1 != 3 && ks1 != 2) return FD_ERR_UNSUPPORTED;
    const void* kern = proj ? (ks1 == 3 ? (const void*)k_attention_bf16<3> : (const void*)k_attention_bf16<2>)
                            : (const void*)k_attention_bf16<0>;
    FD_HIP(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int NP = (H + 1) / 2, DUS = (KT + NQ - 1) / NQ;
    // Query slices per (series, pair): every slice restages K/V (~10 % of a full slice's work); pick the count that
    // minimises rounds x work per round (one workgroup per CU: 8 waves at the 256-VGPR budget).
    int slices = 1;
    double best = 1e30;
    for (int sl = 1; sl <= 8; sl *= 2) {
        if (sl > 1 && DUS / sl < NW) break;                  // at least one unit per wave
        const double rounds = (double)(((size_t)B * NP * sl + ctx->num_cu - 1) / ctx->num_cu);
        const double cost = rounds * (1.0 / sl + 0.1);
        if (cost < best - 1e-9) { best = cost; slices = sl; }
    }
    const int du_per_block = (DUS + slices - 1) / slices;
    const float qscale = 1.4426950408889634f / sqrtf((float)hd);
    const int exact = getenv("FDIFF_ATTN_EXACT") ? 1 : 0;
    const fd_attn_w w{wk, wv, wq};
    const size_t pair_stride = (size_t)ks1 * 1024;
    const dim3 grid(slices, NP, B), block(NTH);
    if (!proj) hipLaunchKernelGGL(k_attention_bf16<0>, grid, block, lds, s, in, out, T, H, hd, D, qscale, du_per_block, exact, w, pair_stride);
    else if (ks1 == 3) hipLaunchKernelGGL(k_attention_bf16<3>, grid, block, lds, s, in, out, T, H, hd, D, qscale, du_per_block, exact, w, pair_stride);
    else hipLaunchKernelGGL(k_attention_bf16<2>, grid, block, lds, s, in, out, T, H, hd, D, qscale, du_per_block, exact, w, pair_stride);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}
