// fd_train_bf16.hip -- bf16 MFMA training path of the score network: forward with dropout + backward, fp32 accumulate.
//
// Reference: ScoreModule.training_step (src/fdiff/models/score_models.py:96-108) through torch autograd over
// nn.TransformerEncoderLayer (post-LN, relu, dropout 0.1 at the attention probabilities, after the out-projection, after
// relu and after linear2; score_models.py:57-62), loss of src/fdiff/utils/losses.py:39-125.
//
// Five kernels per encoder layer instead of ~40 launches of the exact-f32 path (fd_score_f32.hip / fd_score_bwd.hip):
//   forward   k_tr_attn_fwd   (series, head pair): Q/K/V projections, softmax, dropout, P V            -> att
//             k_tr_ffn_fwd    128 tokens: out-proj + dropout + residual + LN1 + FFN (hidden in registers, dropout)
//                             + dropout + residual + LN2                                               -> next layer input
//   backward  k_tr_ffn_bwd    128 tokens: LN2 bwd, FFN input gradient (d hidden never leaves registers), LN1 bwd,
//                             out-proj input gradient                                                  -> d att, d residual
//             k_tr_attn_bwd   (series, head pair) -- from 12 token tiles on (series, head) --: recomputes Q/K/V and P, d Q/K/V,
//                             input gradient of in_proj (per pair / per head partial tensors, summed by the next k_tr_ffn_bwd)
//   once      k_tr_wgrad      every weight gradient of every layer: each output element is owned by ONE wave that walks the
//                             tokens of its split in a fixed order (the FFN hidden / d hidden are recomputed per 32-token
//                             block, never materialised); k_tr_reduce adds the token splits in a fixed order.
// No float atomics anywhere: two runs give bit-identical gradients.
//
// Conventions (v_mfma_f32_16x16x32_bf16 / _16x16x16): token on lane&15, g = lane>>4; C tile [row 4g+r][col lane&15];
// weights are the A operand in 1 KiB fragment blocks (fd_bf16_images.h).  Activations that a later kernel needs with the
// TOKEN axis as the contraction (weight gradients) are also kept as bf16 "T-blocks" [32-token block][feature row][32 tokens];
// activations needed as MFMA operands with the FEATURE axis as the contraction are kept as bf16 rows [token][32*KS1 k-slots]
// with the constant 1.0 in k-slot D (the bias row of the weight images).  Dropout decisions are stored as bits by the forward
// (hidden: kept AND h > 0; attention: kept) -- the two orientations in which the backward needs them (token on lane /
// feature on lane) cannot both be regenerated from one counter layout without 4x the Philox evaluations.
#include <hip/hip_ext.h>

#include "fd_gemm_f32.h"
#include "fd_train_dev.h"

uint64_t fd_dropout_site_offset(uint64_t base, int layer, int site);   // fd_score_f32.hip
namespace fdf32 {
void time_embed(const float* t, const float* W, const float* Wd, const float* bd, float* temb, int B, int D, hipStream_t s);
void embed(const float* x, const float* We, const float* be, const float* pe, const float* temb, float* h, int M, int T, int C,
           int D, hipStream_t s);
}  // namespace fdf32
int fd_time_embed_train(const float* t, const float* W, const float* Wd, const float* bd, float* emb, float* temb, int B, int D,
                        hipStream_t s);                                  // fd_score_f32.hip
int fd_embed_backward(fd_score* m, const float* dh, const float* emb, float* dtemb, float* grads, int B, float* skp,
                      size_t skp_floats, hipStream_t s);                 // fd_score_bwd.hip

namespace {


// Every dropout decision of one encoder layer (16 per Philox4x32-10 evaluation), generated AHEAD of the kernels that use
// them on the context's side stream: the RNG has no data dependency, and inside the latency-bound forward kernels a Philox
// evaluation (~560 issue cycles) per 32-wide FFN chunk was the longest item of the loop.  Byte layouts:
//   hkeep (Mpad, 4, F/32)      hidden units of token m, lane group g, chunk: bits 0-3 units 4g+r, bits 4-7 units 16+4g+r
//   pmask (B, H, T, NJ, 4)     attention probabilities of query t, key block jb, lane group g (keys 4g+r | 16+4g+r)
//   rb1 / rb3 (Mpad, (DT+1)/2, 4)   out-projection / FFN output rows
struct MaskArgs {
    unsigned char* hkeep; unsigned char* pmask; unsigned char* rb1; unsigned char* rb3;
    unsigned long long off0, off1, off2, off3;
    long long n_h, n_p, n_r;      // byte counts
    int NS2;                      // chunks per token (F / 32)
};
__global__ __launch_bounds__(256) void k_tr_masks(const TrDims d, const MaskArgs a) {
    // one evaluation = two adjacent bytes of a buffer (every byte count is even: each has a factor 4); counters are per pair
    const long long h2 = a.n_h / 2, p2 = a.n_p / 2, r2 = a.n_r / 2;
    const long long total = h2 + p2 + 2 * r2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        unsigned short* dst;
        unsigned long long ctr;
        if (i < h2) { dst = reinterpret_cast<unsigned short*>(a.hkeep) + i; ctr = a.off2 + (unsigned long long)i; }
        else if (i < h2 + p2) { dst = reinterpret_cast<unsigned short*>(a.pmask) + (i - h2); ctr = a.off0 + (unsigned long long)(i - h2); }
        else if (i < h2 + p2 + r2) { dst = reinterpret_cast<unsigned short*>(a.rb1) + (i - h2 - p2); ctr = a.off1 + (unsigned long long)(i - h2 - p2); }
        else { dst = reinterpret_cast<unsigned short*>(a.rb3) + (i - h2 - p2 - r2); ctr = a.off3 + (unsigned long long)(i - h2 - p2 - r2); }
        *dst = (unsigned short)fd_drop16(ctr, d.seed, d.thr16);
    }
}

// ------------------------------------------------------------------------------------------------ layer-input preparation
// fp32 (M, D) -> bf16 rows (ones in slot D) + T-block (ones row), for the first layer's input (the embedding kernel is the
// exact-f32 one).  One wave per 16-token tile.
template <int KS1, int DT>
__global__ __launch_bounds__(256) void k_tr_prep(const float* __restrict__ x, __bf16* __restrict__ rb, __bf16* __restrict__ tb,
                                                  int M, int Mpad, int D) {
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int m = tile * 16 + tok;
    if (tile * 16 >= Mpad) return;
    const bool valid = m < M;
    f32x4 v[DT];
    load_ctile<DT>(x, m, valid, D, g, v);
    store_rows<DT, KS1>(rb, m, valid, D, g, v, true);
    // T-block through the wave-private LDS transpose (32-byte runs instead of 16 DT scattered 2-byte stores per lane: the
    // kernel was 24 us of a 1.35 ms step at 6 400 tokens)
    __shared__ __attribute__((aligned(16))) char tscr_all[4 * 32 * DT * 16];
    const int m0 = tile * 16;
    store_T16<DT>(tscr_all + (threadIdx.x >> 6) * (32 * DT * 16), tb + ((size_t)(m0 >> 5) * (16 * DT)) * 32 + (m0 & 31), 32, lane, D, v, true, valid);
}

// ------------------------------------------------------------------------------------------------ attention forward
// grid (NP, B), 256 threads.  K (both heads of the pair, 8-byte rows) and V^T (16-byte rows of 8 keys) of the series in LDS.
struct AttnFwdArgs {
    const __bf16* x0rb;       // (Mpad, RBW) layer input rows
    float* att;               // (M, D)
    __bf16* attT;             // T-block, ones row
    float* lse2;              // (B, H, T): row maximum + log2(row sum) of the scaled scores (base-2 logits)
    const unsigned char* pmask;   // (B, H, T, NJ, 4) keep bits of the attention dropout (k_tr_masks)
    const char* wk; const char* wv; const char* wq;   // pair images (KS1 blocks per pair)
};

// NW waves per workgroup: a wave owns the token tiles wave, wave + NW, ... (NW = 8 from five tiles on: the kernels are latency-
// bound chains per tile, so twice the waves per (series, head pair) halve the critical path at T = 100 and double the waves
// per SIMD that hide each other's LDS / MFMA / exp latencies at T = 252)
#ifndef FD_TR_WG_DMA_AT
#define FD_TR_WG_DMA_AT 1      // k_tr_wgrad: where a block issues the staging DMA of block + 2: 0 behind the barrier (52.1 us per layer), 1 behind the
                               // H / d H MFMAs (50.0), 2 behind every MFMA of the block (53.0); profiles/r04_train_wgrad_experiments.txt
#endif
#ifndef FD_TR_ATTN_MINW
#define FD_TR_ATTN_MINW 2
#endif
#ifndef FD_TR_ABL_FWD
#define FD_TR_ABL_FWD 0           // k_tr_ffn_fwd timing ablations (wrong results): 1 no weight DMA in the loop, 2 no barrier, 4 no mask /
#endif                            // activity block, 16 no chunk loop at all (prologue + epilogue only), 64 no ballots / activity words, 128 no
                                  // activity byte, 256 no keep-mask table
#ifndef FD_TR_BALLOTS
#define FD_TR_BALLOTS 0           // (rounds 3-5: k_tr_ffn_fwd also wrote the activity bits as 16-token ballots per hidden unit for k_tr_wgrad, 4.8 us
#endif                            // of its chunk loop; k_tr_wgrad reads them out of the activity bytes now.  1 keeps the ballots for timing A/Bs only.)
#ifndef FD_TR_ATTN_OH_MINW
#define FD_TR_ATTN_OH_MINW 3          // one-head attention backward: three 4-wave workgroups per CU
#endif
#ifdef FD_TR_PROF_AF        // variant build: phase clocks of k_tr_attn_fwd (workgroup (5, 0), every wave), printed after 30 launches
__device__ unsigned long long fd_tr_af_dbg[8 * 8];
#define TAF_STAMP(slot, t_prev)                                                                          \
    do {                                                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter();                                    \
        if (blockIdx.x == 5 && blockIdx.y == 0 && lane == 0) fd_tr_af_dbg[wave * 8 + (slot)] += now_ - (t_prev); \
        (t_prev) = now_;                                                                                 \
    } while (0)
#else
#define TAF_STAMP(slot, t_prev) do { } while (0)
#endif
template <int KS1, int NW>
__global__ __launch_bounds__(NW * 64) void k_tr_attn_fwd(const TrDims d, const AttnFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long tprev = __builtin_readcyclecounter();
    (void)tprev;
    int pair, b;
    xcd_deal(d, pair, b);                            // (the pairs of a series share an L2)
    const int T = d.T, KT = d.KT, NJ = d.NJ, NTOK = KT * 16, hd = d.hd, H = d.H, D = d.D;
    char* const kbf = smem;                          // [NTOK][4][8 B]
    char* const vbf = smem + (size_t)NTOK * 32;      // [NJ][4][16][16 B]
    // keep masks of four packed probabilities by nibble of keep bits: klut[n] = two dwords of bf16 lane masks (bit r of n keeps the
    // r-th of the four).  The dropped P is cleared AFTER packing: one 8-byte LDS read + two ANDs per four scores instead of and +
    // compare + select per score (24 of ~60 VALU instructions per (key block, head) iteration of pass 2)
    unsigned* const klut = reinterpret_cast<unsigned*>(vbf + (size_t)NJ * 1024);
    if (threadIdx.x < 32) {
        const unsigned n = threadIdx.x >> 1, hi = threadIdx.x & 1;
        klut[threadIdx.x] = ((n >> (2 * hi)) & 1u ? 0x0000ffffu : 0u) | ((n >> (2 * hi + 1)) & 1u ? 0xffff0000u : 0u);
    }
    const size_t pstride = (size_t)KS1 * 1024;
    auto wfrag = [&](const char* img, int ks) { return *reinterpret_cast<const bf16x8*>(img + pair * pstride + ((size_t)ks * 64 + lane) * 16); };
    // a token tile's B fragments from the bf16 rows: unconditional loads from a clamped row, cleared afterwards (inside row_frag's
    // `if (!valid)` every fragment was waited for where it was requested: 4.1 K cycles per staged tile, 2.4 K per Q projection in
    // the phase clocks, profiles/r05_train_attn_fwd_keep_bytes.txt)
    auto xload = [&](int tile, u32x4 (&r)[KS1]) {
        const int t = tile * 16 + tok, tc = t < T ? t : T - 1;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) r[ks] = *reinterpret_cast<const u32x4*>(a.x0rb + (size_t)(b * T + tc) * d.RBW + 32 * ks + 8 * g);
    };
    auto xfrag_of = [&](int tile, const u32x4 (&r)[KS1], int ks) -> bf16x8 {
        const unsigned keep = (tile * 16 + tok < T) ? ~0u : 0u;
        return __builtin_bit_cast(bf16x8, u32x4{r[ks][0] & keep, r[ks][1] & keep, r[ks][2] & keep, r[ks][3] & keep});
    };
    {
        bf16x8 wkf[KS1], wvf[KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) { wkf[ks] = wfrag(a.wk, ks); wvf[ks] = wfrag(a.wv, ks); }
        u32x4 xc[KS1], xn[KS1];
        xload(wave < KT ? wave : KT - 1, xc);
        for (int kt = wave; kt < KT; kt += NW) {
            xload(kt + NW < KT ? kt + NW : kt, xn);           // (the next tile's rows in flight under this tile's MFMAs)
            f32x4 ka = f4zero(), vc = f4zero();
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const bf16x8 xf = xfrag_of(kt, xc, ks);
                ka = MFMA(wkf[ks], xf, ka);
                vc = MFMA(xf, wvf[ks], vc);
            }
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) xc[ks] = xn[ks];
            *reinterpret_cast<u32x2*>(kbf + ((size_t)(kt * 16 + tok) * 4 + g) * 8) = u32x2{cvt_pk_bf16(ka[0], ka[1]), cvt_pk_bf16(ka[2], ka[3])};
            char* dst = vbf + ((size_t)((kt >> 1) * 4 + g) * 16 + tok) * 16;
            *reinterpret_cast<u32x2*>(dst + 8 * (kt & 1)) = u32x2{cvt_pk_bf16(vc[0], vc[1]), cvt_pk_bf16(vc[2], vc[3])};
            if ((KT & 1) && kt == KT - 1) *reinterpret_cast<u32x2*>(dst + 8) = u32x2{0u, 0u};
        }
    }
    TAF_STAMP(0, tprev);          // K / V staging of this wave's tiles
    __syncthreads();
    TAF_STAMP(1, tprev);          // barrier
    bf16x8 wqf[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) wqf[ks] = wfrag(a.wq, ks);
    const bool lo_grp = (g >> 1) == 0;
    const int myhead = 2 * pair + (g >> 1);
    f32x4 cmask;
#pragma unroll
    for (int r = 0; r < 4; ++r) cmask[r] = ((KT - 1) * 16 + 4 * g + r >= T) ? kNegBig : 0.f;
    const f32x4 allneg = {kNegBig, kNegBig, kNegBig, kNegBig};
    auto kfrag = [&](int kt) { return *reinterpret_cast<const s16x4*>(kbf + ((size_t)(kt * 16 + tok) * 4 + g) * 8); };
    for (int qt = wave; qt < KT; qt += NW) {
        const int t = qt * 16 + tok;
        f32x4 qa = f4zero();
        {
            u32x4 xq[KS1];
            xload(qt, xq);
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) qa = MFMA(wqf[ks], xfrag_of(qt, xq, ks), qa);
        }
        const unsigned q01 = cvt_pk_bf16(qa[0], qa[1]), q23 = cvt_pk_bf16(qa[2], qa[3]);
        s16x4 qb[2];
        qb[0] = __builtin_bit_cast(s16x4, u32x2{lo_grp ? q01 : 0u, lo_grp ? q23 : 0u});
        qb[1] = __builtin_bit_cast(s16x4, u32x2{lo_grp ? 0u : q01, lo_grp ? 0u : q23});
        TAF_STAMP(2, tprev);      // Q projection of the tile
        // pass 1: exact row maxima (base-2 logits: log2(e)/sqrt(hd) is folded into W_q)
        float mx[2] = {kNegBig, kNegBig};
        // (round 6: the last key tile alone carries the padding mask -- peeled, no select per tile; the same in pass 2 below)
        auto maxtile = [&](int kt, const f32x4 c0) {
            const s16x4 kf = kfrag(kt);
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                const f32x4 v = MFMA16(kf, qb[hs], c0);
                mx[hs] = fmaxf(fmaxf(fmaxf(mx[hs], v[0]), v[1]), fmaxf(v[2], v[3]));
            }
        };
#ifdef FD_TR_ABL_AF1          // (timing ablation, wrong results: no pass 1)
        mx[0] = mx[1] = 0.f;
#else
        for (int kt = 0; kt < KT - 1; ++kt) maxtile(kt, f4zero());
        maxtile(KT - 1, cmask);
#endif
        f32x4 negm[2];
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            mx[hs] = group_max(mx[hs]);
            negm[hs] = f32x4{-mx[hs], -mx[hs], -mx[hs], -mx[hs]};
        }
        TAF_STAMP(3, tprev);      // pass 1
        // pass 2: P = exp2(S - max); row sums of the UNDROPPED P; dropped P (unscaled) times V.  The keep bytes of key block jb + 1
        // are requested while block jb runs (unconditional loads from clamped addresses: a padded query row or a missing odd head
        // reads some other row's bits and its output is never stored).  Inside `if (t < T && head < H)` each byte was waited for
        // where it was requested, a dependent L2 round trip between the exponentials and the P V MFMA of every (block, head)
        // iteration: the timing ablation without the loads says 4.4 of the kernel's 31.3 us (profiles/r05_train_attn_fwd_keep_bytes.txt).
        float ls[2] = {0.f, 0.f};
        f32x4 o2[2] = {f4zero(), f4zero()};
        const unsigned char* prow[2];
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            const int head = 2 * pair + hs, hc = head < H ? head : H - 1;
            prow[hs] = a.pmask + ((((size_t)b * H + hc) * T + (t < T ? t : 0)) * NJ) * 4 + g;
        }
        unsigned bnext[2] = {0xffu, 0xffu};
        if (d.p > 0.f) {
            bnext[0] = prow[0][0];
            bnext[1] = prow[1][0];
        }
        // one key block; only the LAST one carries a mask in the C operand of its score MFMAs (x + 0.0f = x bit for bit, and the eight
        // adds per (block, head) were 14 % of the loop's VALU cycles)
        auto keyblock = [&](int jb, auto last_c) {
            constexpr bool LAST = decltype(last_c)::value;
            const unsigned bcur[2] = {bnext[0], bnext[1]};
            if (d.p > 0.f) {
                const int jn = jb + 1 < NJ ? jb + 1 : jb;
                bnext[0] = prow[0][(size_t)jn * 4];
                bnext[1] = prow[1][(size_t)jn * 4];
            }
            const int ka = 2 * jb, kb = (2 * jb + 1 < KT) ? 2 * jb + 1 : ka;
            const s16x4 kfa = kfrag(ka), kfb = kfrag(kb);
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vbf + ((size_t)(jb * 4 + g) * 16 + tok) * 16);
            const f32x4 ma = (ka == KT - 1) ? cmask : f4zero();
            const f32x4 mb = (2 * jb + 1 >= KT) ? allneg : ((kb == KT - 1) ? cmask : f4zero());
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                f32x4 pa = MFMA16(kfa, qb[hs], LAST ? (ma + negm[hs]) : negm[hs]);
                f32x4 pb = MFMA16(kfb, qb[hs], LAST ? (mb + negm[hs]) : negm[hs]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pa[r] = __builtin_amdgcn_exp2f(pa[r]);
                    pb[r] = __builtin_amdgcn_exp2f(pb[r]);
                }
                ls[hs] += (pa[0] + pa[1]) + (pa[2] + pa[3]) + (pb[0] + pb[1]) + (pb[2] + pb[3]);
                const unsigned bits = bcur[hs];
                const u32x2 ka2 = *reinterpret_cast<const u32x2*>(klut + 2 * (bits & 15u)), kb2 = *reinterpret_cast<const u32x2*>(klut + 2 * (bits >> 4));
                const u32x4 pk = __builtin_bit_cast(u32x4, pack8(pa, pb));
                o2[hs] = MFMA(vf, __builtin_bit_cast(bf16x8, u32x4{pk[0] & ka2[0], pk[1] & ka2[1], pk[2] & kb2[0], pk[3] & kb2[1]}), o2[hs]);
            }
        };
        for (int jb = 0; jb < NJ - 1; ++jb) keyblock(jb, std::false_type{});
        keyblock(NJ - 1, std::true_type{});
        TAF_STAMP(4, tprev);      // pass 2
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) ls[hs] = group_sum(ls[hs]);
        const float lmine = lo_grp ? ls[0] : ls[1], mmine = lo_grp ? mx[0] : mx[1];
        const float inv = d.keep_scale * __builtin_amdgcn_rcpf(lmine);     // (v_rcp_f32, 1 ulp; the product is rounded to bf16)
        const int m = b * T + t;
        if (t < T && myhead < H) {
            float ov[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (lo_grp ? o2[0][r] : o2[1][r]) * inv;
            // row form: the lane's dims as one 16- or 8-byte store where head_dim allows (four scalar stores otherwise)
            float* orow = a.att + (size_t)m * D + myhead * hd + 4 * (g & 1);
            const int nv = min(4, max(0, hd - 4 * (g & 1)));
            if (nv == 4) *reinterpret_cast<f32x4_a4*>(orow) = f32x4_a4{ov[0], ov[1], ov[2], ov[3]};
            else if (nv == 2) *reinterpret_cast<f32x2_a4*>(orow) = f32x2_a4{ov[0], ov[1]};
            else
                for (int r = 0; r < nv; ++r) orow[r] = ov[r];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int dd = 4 * (g & 1) + r;
                if (dd < hd) a.attT[((size_t)(m >> 5) * d.NFT + myhead * hd + dd) * 32 + (m & 31)] = (__bf16)ov[r];
            }
            if ((g & 1) == 0) a.lse2[((size_t)b * H + myhead) * T + t] = mmine + __builtin_amdgcn_logf(lmine);
        }
        if (pair == 0 && t < T) {          // ones row (bias column of d W_o) and the zero rows of the T-block padding
            for (int f = D + g; f < d.NFT; f += 4)
                a.attT[((size_t)(m >> 5) * d.NFT + f) * 32 + (m & 31)] = (__bf16)((f == D) ? 1.0f : 0.f);
        }
        TAF_STAMP(5, tprev);      // epilogue stores of the tile
    }
}

// ------------------------------------------------------------------------------------------------ FFN-side forward
#ifdef FD_TR_PROF_FFN       // variant build: in-kernel phase clocks of k_tr_ffn_fwd (workgroup 7, waves 0 and 4), printed after 30 launches
__device__ unsigned long long fd_tr_ffn_dbg[2 * 8];
#define TRF_STAMP(slot, t_prev)                                                                          \
    do {                                                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter();                                    \
        if (blockIdx.x == 7 && lane == 0 && (wave & 3) == 0) fd_tr_ffn_dbg[(wave >> 2) * 8 + (slot)] += now_ - (t_prev); \
        (t_prev) = now_;                                                                                 \
    } while (0)
#else
#define TRF_STAMP(slot, t_prev) do { } while (0)
#endif
struct FfnFwdArgs {
    const float* x0;          // (M, D) layer input (residual)
    const float* att;         // (M, D)
    float* s1; float* s2;     // (M, D) pre-LayerNorm sums (saved)
    float* out;               // (M, D) layer output
    char* stage;              // StageL records of the layer (x1 rows here, d f rows by the backward)
    __bf16* outrb; __bf16* outT;   // next layer's input in operand form (null for the last layer)
    unsigned char* active;    // (Mpad, 4, F/32): bit e of byte (m, g, chunk): hidden unit kept by dropout AND > 0
    const char* wo_img;       // [DT][KSO]
    const char* ffn_img;      // chunk-major forward image of the layer
    const float* bo; const float* g1; const float* be1; const float* b2; const float* g2; const float* be2;
    const unsigned char* hkeep; const unsigned char* rb1; const unsigned char* rb3;   // dropout decisions (k_tr_masks)
    FSplit fs;
};

// 8 waves = 4 token tiles (64 tokens) x 2 halves of F.  The weight stream (one 32-wide chunk per F-half per step, 2*NB KiB)
// runs through a ring of 4 LDS buffers filled 3 steps ahead by global_load_lds; nothing else touches vector memory inside
// the loop (mask bits go through LDS), so `s_waitcnt vmcnt(2*NDMA)` means "all but the two newest buffers have landed".
template <int KS1, int DT, int KSO>
__global__ __launch_bounds__(TW * 64, 2) void k_tr_ffn_fwd(const TrDims d, const FfnFwdArgs a) {
    constexpr int NB = 2 * KS1 + DT, WB = 2 * NB * 1024, NBUF = 4, NDMA = (2 * NB + TW - 1) / TW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = wave & 3, fhw = wave >> 2;
    const int D = d.D, M = d.M, NS = d.F / 64;
    unsigned long long tprev = __builtin_readcyclecounter();
    (void)tprev;
    char* const ring = smem;
    char* const scratch = smem + NBUF * WB + wave * KS1 * 1024;            // wave-private fragment scratch (prologue)
    f32x4* const xch = reinterpret_cast<f32x4*>(smem + NBUF * WB);          // [4 tiles][DT][64] (after the loop: aliases scratch)
    constexpr int SCR = (TW * KS1 * 1024 > 4 * DT * 1024) ? TW * KS1 * 1024 : 4 * DT * 1024;
    unsigned char* const actB = reinterpret_cast<unsigned char*>(smem + NBUF * WB + SCR) + wave * (64 * NS);   // [64 lanes][NS]
    unsigned short* const actT = reinterpret_cast<unsigned short*>(smem + NBUF * WB + SCR + TW * 64 * NS) + wave * (NS * 32);   // [NS][32]
    (void)actT;               // (only -DFD_TR_BALLOTS=1 builds fill it)
    char* const tscr = smem + NBUF * WB + SCR + TW * 64 * NS + TW * NS * 32 * 2 + (wave & 3) * (32 * DT * 16);   // owners' T-store transpose
    // keep masks of four packed bf16 values by nibble of keep bits (see the chunk loop): klut[2 n], klut[2 n + 1] = lane masks of values 0-1 / 2-3
    unsigned* const klut = reinterpret_cast<unsigned*>(smem + NBUF * WB + SCR + TW * 64 * NS + TW * NS * 32 * 2 + 4 * (32 * DT * 16));
    if (threadIdx.x < 32) {
        const unsigned n = threadIdx.x >> 1, hi = threadIdx.x & 1;
        klut[threadIdx.x] = ((n >> (2 * hi)) & 1u ? 0x0000ffffu : 0u) | ((n >> (2 * hi + 1)) & 1u ? 0xffff0000u : 0u);
    }
    // the epilogue's small vectors (b2, gamma2, beta2) through LDS, [vector][4 DT float4], zero beyond D: read where they were used
    // -- inside the epilogue's lane-divergent `if (d0 < D)` branches -- every one of the ten loads was waited for at the end of its
    // branch, ten dependent L2 round trips in the owner's epilogue (9 K of the kernel's 88 K cycles; the finding of k_ffn_ln,
    // profiles/r05_long_ffn_ln_phase_clocks.txt).  Requested here, written in front of the barrier that precedes the chunk loop.
    float4* const lvec = reinterpret_cast<float4*>(klut + 32);
    float4 lval = {0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x < 3 * 4 * DT) {
        const int vq = threadIdx.x / (4 * DT), cq = threadIdx.x - vq * (4 * DT);
        const float* src = vq == 0 ? a.b2 : vq == 1 ? a.g2 : a.be2;
        if (4 * cq < d.D) lval = *reinterpret_cast<const float4*>(src + 4 * cq);
    }
    // F-split (struct FSplit): token block, chunk range and role of this workgroup
    const int nsp = d.fsplit, blk = nsp == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x, fq = nsp == 2 ? (int)(blockIdx.x & 1) : 0;
    const int NSH = NS / nsp, cbase = fq * NSH;
    const bool finisher = fq == nsp - 1;
    const int m = (blk * 4 + tile) * 16 + tok;
    const bool valid = m < M;
    const bool owner = fhw == 0;
    // Every workgroup streams the SAME weights: marching through them in lockstep makes all CUs hit the same L2 channel at the
    // same time (measured in the persistent kernel: ~25 GB/s per CU).  The chunks are summed, so each workgroup walks them in
    // its own rotated order.
    const int rot = d.norot ? 0 : (int)(((unsigned)blk * 5u) % (unsigned)NSH);
    auto issue = [&](int st) {
        int ce = st + rot;
        ce -= (ce >= NSH) ? NSH : 0;
        ce += cbase;
        const char* src = a.ffn_img + (size_t)ce * WB + lane * 16;
        char* dst = ring + (st % NBUF) * WB;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            int bb = wave + i * TW;
            bb %= 2 * NB;                 // padding copies repeat a block (uniform vmcnt bookkeeping)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + bb * 1024), LDS_PTR(dst + bb * 1024), 16, 0, 0);
        }
    };
    issue(0);
    if (NSH > 1) issue(1);
    if (NSH > 2) issue(2);
    // ---- out-projection + bias + dropout + residual -> s1  (both waves of a tile compute it; the owner stores)
    // Every global read of the prologue is issued before the first MFMA waits (attention rows, W_o fragments, residual rows,
    // dropout bytes, bias and LayerNorm vectors): read where they were used, hipcc put an `s_waitcnt vmcnt(0)` behind each
    // group -- about ten dependent L2 round trips in front of the chunk loop of a kernel whose 100 workgroups cannot hide them.
    f32x4 v[DT];
    float4 bo4[DT], g14[DT], be14[DT];
    {
        float e8[KSO][8];
        bf16x8 wof[DT][KSO];
        const int mc = valid ? m : 0;                      // (clamped row: unconditional loads, selected afterwards)
#pragma unroll
        for (int ks = 0; ks < KSO; ++ks) {
            const int head = 4 * ks + g, hc = head < d.H ? head : d.H - 1;
            // the head's 8 dim slots as two dword-aligned 16-byte loads (eight scalar loads per (token, head): every one of the 24
            // instructions walked 64 different cache lines -- issuing them took 6.7 K clocks of a 29 K-clock prologue); slots
            // >= head_dim belong to the next head / row (16 bytes of slack behind the buffer) and are zeroed below
            {
                const f32x4_a4* p8 = reinterpret_cast<const f32x4_a4*>(a.att + (size_t)mc * D + hc * d.hd);
                const f32x4_a4 lo = p8[0], hi = p8[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) { e8[ks][e] = lo[e]; e8[ks][4 + e] = hi[e]; }
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                wof[dt][ks] = *reinterpret_cast<const bf16x8*>(a.wo_img + ((size_t)(dt * KSO + ks) * 64 + lane) * 16);
        }
        unsigned bits[DT];
        row_drop_bits<DT>(d, a.rb1, m, valid, g, bits);
        load_ctile<DT>(a.x0, m, valid, D, g, v);
#pragma unroll
        for (int ks = 0; ks < KSO; ++ks) {
            const int head = 4 * ks + g;
#pragma unroll
            for (int e = 0; e < 8; ++e) e8[ks][e] = (valid && head < d.H && e < d.hd) ? e8[ks][e] : 0.f;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g, dr = d0 < D ? d0 : 0;
            bo4[dt] = *reinterpret_cast<const float4*>(a.bo + dr);
            g14[dt] = *reinterpret_cast<const float4*>(a.g1 + dr);
            be14[dt] = *reinterpret_cast<const float4*>(a.be1 + dr);
        }
        __builtin_amdgcn_sched_barrier(0);
        TRF_STAMP(0, tprev);      // entry, DMA issue, every prologue load issued
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f4zero();
#pragma unroll
        for (int ks = 0; ks < KSO; ++ks) {
            const u32x4 pk = {cvt_pk_bf16(e8[ks][0], e8[ks][1]), cvt_pk_bf16(e8[ks][2], e8[ks][3]), cvt_pk_bf16(e8[ks][4], e8[ks][5]),
                              cvt_pk_bf16(e8[ks][6], e8[ks][7])};
            const bf16x8 af = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] = MFMA(wof[dt][ks], af, o[dt]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            if (d0 < D) {
                const float bv[4] = {bo4[dt].x, bo4[dt].y, bo4[dt].z, bo4[dt].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[dt][r] += ((bits[dt] >> r) & 1u) ? (o[dt][r] + bv[r]) * d.keep_scale : 0.f;
            }
        }
    }
    // (global stores of the prologue are issued AFTER the FFN loop: `s_waitcnt vmcnt(0)` in front of the loop would otherwise
    // wait for their write acknowledgements -- measured 17 us of a 57 us kernel -- and the data is live in registers anyway)
    f32x4 s1keep[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) s1keep[dt] = v[dt];
    {
        float mean, rstd;
        ln_stats<DT>(v, D, g, mean, rstd);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            if (d0 < D) {
                const float4 gm = g14[dt], bt = be14[dt];
                v[dt][0] = (v[dt][0] - mean) * rstd * gm.x + bt.x;
                v[dt][1] = (v[dt][1] - mean) * rstd * gm.y + bt.y;
                v[dt][2] = (v[dt][2] - mean) * rstd * gm.z + bt.z;
                v[dt][3] = (v[dt][3] - mean) * rstd * gm.w + bt.w;
            } else {
                v[dt] = f4zero();
            }
        }
    }
    bf16x8 xf[KS1];
    ctile_to_frags<DT, KS1>(scratch, lane, D, v, true, xf);
    TRF_STAMP(1, tprev);          // loads landed, out-projection, LayerNorm1, fragments
    // this wave's dropout bytes of its F-half -> LDS (overwritten chunk by chunk with kept-AND-positive), and the epilogue's
    // dropout bits, fetched now (a dependent global round trip after the loop otherwise)
    if (d.p > 0.f && valid) {
        const unsigned char* srcb = a.hkeep + ((size_t)m * 4 + g) * (2 * NS) + fhw * NS;
        for (int c = 0; c < NS; c += 16) *reinterpret_cast<u32x4*>(actB + lane * NS + c) = *reinterpret_cast<const u32x4*>(srcb + c);
    } else {
        // (no dropout: every unit kept; a token beyond M: nothing kept, so the chunk loop needs no validity test)
        const unsigned fill = valid ? ~0u : 0u;
        for (int c = 0; c < NS; c += 16) *reinterpret_cast<u32x4*>(actB + lane * NS + c) = u32x4{fill, fill, fill, fill};
    }
    unsigned bits3[DT];
    row_drop_bits<DT>(d, a.rb3, m, valid, g, bits3);
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f4zero();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x < 3 * 4 * DT) lvec[threadIdx.x] = lval;
    TRF_STAMP(2, tprev);          // dropout bytes staged, first weight chunks landed (own share)
    __syncthreads();
    TRF_STAMP(3, tprev);          // barrier
    // ---- FFN: this wave's F-half, hidden in registers.  A step's W1 / W2 fragments are read from the ring into registers
    // during the PREVIOUS step (its buffer became visible one barrier earlier: the wait below leaves only the newest DMA batch
    // in flight), so no LDS round trip sits between the barrier and the step's MFMAs (five of them per step before: the
    // wave's time was its stalls, not its instruction count -- 14 % fewer instructions changed nothing).  Two steps per
    // trip = two register sets, the read for the step after the last one is clamped (no branch, no copies at the back edge).
    unsigned bits_cur = 0u;
    auto frags = [&](int c, bf16x8 (&w1)[2 * KS1], bf16x8 (&w2)[DT]) {
        const int cc = c < NSH ? c : NSH - 1;
        const char* wb = ring + (cc % NBUF) * WB + fhw * NB * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < 2 * KS1; ++i) w1[i] = *reinterpret_cast<const bf16x8*>(wb + i * 1024);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) w2[dt] = *reinterpret_cast<const bf16x8*>(wb + (2 * KS1 + dt) * 1024);
    };
    auto step = [&](int c, const bf16x8 (&w1)[2 * KS1], const bf16x8 (&w2)[DT], bf16x8 (&n1)[2 * KS1], bf16x8 (&n2)[DT]) {
        if (!(FD_TR_ABL_FWD & 1) && c + 3 < NSH) issue(c + 3);
        frags(c + 1, n1, n2);
        f32x4 h0 = f4zero(), h1 = f4zero();
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            h0 = MFMA(w1[ks], xf[ks], h0);
            h1 = MFMA(w1[KS1 + ks], xf[ks], h1);
        }
#if FD_TR_ABL_FWD & 4
        {
#pragma unroll
            for (int r = 0; r < 4; ++r) { h0[r] = fmaxf(h0[r], 0.f); h1[r] = fmaxf(h1[r], 0.f); }
            const bf16x8 hb0 = pack8(h0, h1);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] = MFMA(w2[dt], hb0, acc[dt]);
            if (!(FD_TR_ABL_FWD & 1) && c + 3 < NSH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (!(FD_TR_ABL_FWD & 2)) __builtin_amdgcn_s_barrier();
            return;
        }
#endif
        int ce = c + rot;
        ce -= (ce >= NSH) ? NSH : 0;
        int cn = ce + 1;                                           // next step's chunk (clamped read after the last step)
        cn -= (cn >= NSH) ? NSH : 0;
        ce += cbase;
        cn += cbase;
        const unsigned bits = bits_cur;                            // dropout decisions of this chunk (staged before the loop),
        bits_cur = actB[lane * NS + cn];                           // read one step ahead like the fragments
        // relu + dropout on the PACKED hidden values: v_cvt_pk_bf16_f32, v_pk_max_i16 against 0 (a negative bf16 is a negative
        // int16), and the keep decisions as bf16 lane masks from a 16-entry LDS table indexed by the byte's nibbles -- 16 VALU
        // and two 8-byte LDS reads per 8 values.  The per-value form (compare, bit test, s_and of the two ballots, two selects)
        // was ~90 VALU + ~60 SALU in dependent VALU -> SGPR -> SALU -> VALU chains: 12.6 of the kernel's 45 us at 16 128 tokens
        // (timing ablation -DFD_TR_ABL_FWD=4, profiles/r05_train_ffn_fwd_ablations.txt).
        u32x4 pk;
        {
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
            pk = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, pack8(h0, h1)), z8));
#if !(FD_TR_ABL_FWD & 256)
            const u32x2 k0 = *reinterpret_cast<const u32x2*>(klut + 2 * (bits & 15u)), k1 = *reinterpret_cast<const u32x2*>(klut + 2 * (bits >> 4));
            pk = u32x4{pk[0] & k0[0], pk[1] & k0[1], pk[2] & k1[0], pk[3] & k1[1]};
#endif
        }
        // activity = kept AND > 0 = a non-zero packed value.  As a byte per (token, lane group) for the token-on-lane backward:
        // v_pk_min_i16 against 1 turns each half into 0 / 1, shifts gather the eight bits ...
#if !(FD_TR_ABL_FWD & 128)
        {
            // (one 8-wide min: hipcc folded the per-dword two-wide form of this into "all four dwords equal" -- t = mm[0] * 0x55)
            // (signed: the values are non-negative after the relu, and the signed form selects v_pk_min_i16)
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x8 one8 = {1, 1, 1, 1, 1, 1, 1, 1};
            const u32x4 mm = __builtin_bit_cast(u32x4, __builtin_elementwise_min(__builtin_bit_cast(s16x8, pk), one8));
            const unsigned t = mm[0] | (mm[1] << 2) | (mm[2] << 4) | (mm[3] << 6);     // low halves on bits 0, 2, 4, 6; high halves on 16, 18, 20, 22
            actB[lane * NS + ce] = (unsigned char)((t & 0x55u) | ((t >> 15) & 0xAAu));
        }
#endif
        // ... and with the 16 tokens of the tile as the bits of one word per hidden unit (weight-gradient kernel): the compare's
        // SGPR pair IS the ballot -- bit 16 g + tok of unit 4 g + r (+16 for the second row tile); one compare per half
#if FD_TR_BALLOTS && !(FD_TR_ABL_FWD & 64)
        {
            unsigned long long bal[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bal[2 * q] = __builtin_amdgcn_ballot_w64((pk[q] & 0xffffu) != 0u);
                bal[2 * q + 1] = __builtin_amdgcn_ballot_w64(pk[q] > 0xffffu);
            }
            if (lane == 0) {
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) reinterpret_cast<unsigned long long*>(actT)[ce * 8 + w8] = bal[w8];
            }
        }
#endif
        const bf16x8 hb = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] = MFMA(w2[dt], hb, acc[dt]);
        if (!(FD_TR_ABL_FWD & 1) && c + 3 < NSH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the prefetch reads and this step's LDS writes are done
        if (!(FD_TR_ABL_FWD & 2)) __builtin_amdgcn_s_barrier();
    };
    {
        bf16x8 wa1[2 * KS1], wa2[DT], wb1[2 * KS1], wb2[DT];
        frags(0, wa1, wa2);
        bits_cur = actB[lane * NS + cbase + rot];
        for (int c = 0; c < ((FD_TR_ABL_FWD & 16) ? 0 : NSH); c += 2) {       // (NS = F / 64 is a multiple of 4: F % 1024 == 0)
            step(c, wa1, wa2, wb1, wb2);
            step(c + 1, wb1, wb2, wa1, wa2);
        }
    }
    TRF_STAMP(4, tprev);          // chunk loop
    if (owner && finisher) {
        store_ctile<DT>(a.s1, m, valid, D, g, s1keep);
        stage_rows<DT, KS1>(a.stage, StageL<KS1, DT>::off_xr, m, valid, D, g, v, true);
    }
    // ---- mask bits out: bytes [token][g][chunk] (token-on-lane backward), words [32-token block][half][hidden unit]
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (F-split: the chunks this workgroup ran, [cbase, cbase + NSH): NSH is a multiple of 8, so both ranges keep 16-byte accesses
    //  when it is a multiple of 16 and fall back to 8-byte ones otherwise)
    if (valid) {
        unsigned char* dstb = a.active + ((size_t)m * 4 + g) * (2 * NS) + fhw * NS;
        if ((NSH & 15) == 0) {
            for (int c = cbase; c < cbase + NSH; c += 16)
                *reinterpret_cast<u32x4*>(dstb + c) = *reinterpret_cast<const u32x4*>(actB + lane * NS + c);
        } else {
            for (int c = cbase; c < cbase + NSH; c += 8)
                *reinterpret_cast<u32x2*>(dstb + c) = *reinterpret_cast<const u32x2*>(actB + lane * NS + c);
        }
    }
    TRF_STAMP(5, tprev);          // s1 / stage stores issued, mask bits out
    // ---- combine the F-halves; the owner finishes the tile
    __syncthreads();
    if (!owner) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) xch[(tile * DT + dt) * 64 + lane] = acc[dt];
    }
    __syncthreads();
    TRF_STAMP(6, tprev);          // two barriers + exchange
    if (!owner) return;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = (acc[dt] + xch[(tile * DT + dt) * 64 + lane]) * d.keep_scale;      // hidden-unit keep scale
    if (nsp == 2) {
        if (!finisher) { fsplit_hand_over<DT>(a.fs, blk, tile, lane, acc); return; }
        fsplit_take_over<DT>(a.fs, blk, tile, lane, acc);
    }
    {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            if (d0 < D) {
                const float4 bb = lvec[0 * 4 * DT + 4 * dt + g];
                const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[dt][r] += ((bits3[dt] >> r) & 1u) ? (acc[dt][r] + bv[r]) * d.keep_scale : 0.f;
            }
        }
    }
    store_ctile<DT>(a.s2, m, valid, D, g, v);
    {
        float mean, rstd;
        ln_stats<DT>(v, D, g, mean, rstd);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            if (d0 < D) {
                const float4 gm = lvec[1 * 4 * DT + 4 * dt + g], bt = lvec[2 * 4 * DT + 4 * dt + g];
                v[dt][0] = (v[dt][0] - mean) * rstd * gm.x + bt.x;
                v[dt][1] = (v[dt][1] - mean) * rstd * gm.y + bt.y;
                v[dt][2] = (v[dt][2] - mean) * rstd * gm.z + bt.z;
                v[dt][3] = (v[dt][3] - mean) * rstd * gm.w + bt.w;
            } else {
                v[dt] = f4zero();
            }
        }
    }
    store_ctile<DT>(a.out, m, valid, D, g, v);
    if (a.outrb) {
        const int m0w = (blk * 4 + tile) * 16;
        store_rows<DT, KS1>(a.outrb, m, valid, D, g, v, true);
        store_T16<DT>(tscr, a.outT + ((size_t)(m0w >> 5) * (16 * DT)) * 32 + (m0w & 31), 32, lane, D, v, true, valid);
    }
    TRF_STAMP(7, tprev);          // epilogue of the owner (hand-over, LN2, stores issued)
}

// ------------------------------------------------------------------------------------------------ FFN-side backward
struct FfnBwdArgs {
    const float* dy0;         // (M, D) gradient of the layer output: residual-path part ...
    const float* dyp;         // ... plus `npart` partial tensors (the next layer's attention backward, one per head pair)
    int npart; size_t part_stride;
    int part_bf16;            // the partial tensors are bf16 (one per head), part_stride in elements either way
    int rowsum;               // 1: the partial tensors are summed as contiguous 16-byte units through LDS (default); 0: C-tile loads
    const __bf16* dqkvR;      // dxg: the next layer's d(q | k | v) rows (see AttnBwdArgs) instead of partial tensors ...
    const char* winT;         // ... and that layer's in_proj^T image ([pair][which][DT] half blocks): d y += rows . in_proj^T here
    int dxg;
    const float* s1; const float* s2;
    const unsigned char* active;
    float* datt;              // (M, D) gradient of the attention output
    float* dres;              // (M, D) gradient of the layer input through the residual path (= d s1)
    char* stage; __bf16* doT;
    float* vecpart;           // [grid][5][D]: column sums of this workgroup: d b2, d beta2, d gamma2, d beta1, d gamma1
    const char* bffn;         // backward FFN image (chunk-major)
    const char* wot;          // [DT][KS1]
    const float* g1; const float* be1; const float* g2;
    const unsigned char* rb1; const unsigned char* rb3;
    FSplit fs;
};

// LayerNorm backward on a C-layout tile: dy -> ds (in place), xhat given; returns nothing (column sums done by the caller)
#ifdef FD_TR_PROF_FB        // variant build: in-kernel phase clocks of k_tr_ffn_bwd (workgroup 7, waves 0 and 4), printed after 30 launches
__device__ unsigned long long fd_tr_fb_dbg[2 * 8];
#define TRFB_STAMP(slot, t_prev)                                                                          \
    do {                                                                                                  \
        const unsigned long long now_ = __builtin_readcyclecounter();                                     \
        if (blockIdx.x == 7 && lane == 0 && (wave & 3) == 0) fd_tr_fb_dbg[(wave >> 2) * 8 + (slot)] += now_ - (t_prev); \
        (t_prev) = now_;                                                                                  \
    } while (0)
#else
#define TRFB_STAMP(slot, t_prev) do { } while (0)
#endif
// (gamma: unconditional loads from clamped addresses, all issued before the first use -- inside the lane-divergent branch of
//  ln_bwd_tile each of them was waited for at the end of its branch, DT dependent L2 round trips per call)
template <int DT>
__device__ __forceinline__ void ln_gamma_load(const float* __restrict__ gamma, int D, int g, float4 (&gm4)[DT]) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d0 = 16 * dt + 4 * g;
        gm4[dt] = *reinterpret_cast<const float4*>(gamma + (d0 < D ? d0 : 0));
    }
}
template <int DT>
__device__ __forceinline__ void ln_bwd_tile(f32x4 (&dy)[DT], const f32x4 (&xhat)[DT], const float4 (&gm4)[DT], float rstd, int D, int g) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d0 = 16 * dt + 4 * g;
        if (d0 < D) {
            const float4 gm = gm4[dt];
            const float gv[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dy[dt][r] *= gv[r];
                s1 += dy[dt][r];
                s2 += dy[dt][r] * xhat[dt][r];
            }
        } else {
            dy[dt] = f4zero();
        }
    }
    const float invD = 1.0f / (float)D;
    const float m1 = group_sum(s1) * invD, m2 = group_sum(s2) * invD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
        if (16 * dt + 4 * g < D) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dy[dt][r] = rstd * (dy[dt][r] - m1 - xhat[dt][r] * m2);
        }
}

template <int KS1, int DT>
__global__ __launch_bounds__(TW * 64, 2) void k_tr_ffn_bwd(const TrDims d, const FfnBwdArgs a) {
    constexpr int NB = 2 * KS1 + DT, WB = 2 * NB * 1024, NBUF = 4, NDMA = (2 * NB + TW - 1) / TW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = wave & 3, fhw = wave >> 2;
    const bool owner = fhw == 0;
    char* const ring = smem;
    char* const scratch = smem + NBUF * WB + wave * KS1 * 1024;
    f32x4* const xch = reinterpret_cast<f32x4*>(smem + NBUF * WB);
    constexpr int SCR = (TW * KS1 * 1024 > 4 * DT * 1024) ? TW * KS1 * 1024 : 4 * DT * 1024;
    const int D = d.D, M = d.M, NS = d.F / 64;
    unsigned char* const actB = reinterpret_cast<unsigned char*>(smem + NBUF * WB + SCR) + wave * (64 * NS);
    float* const colred = reinterpret_cast<float*>(smem + NBUF * WB + SCR + TW * 64 * NS);   // [4 tiles][5][16*DT]
    char* const scratch2 = smem + NBUF * WB + SCR + TW * 64 * NS + 4 * 5 * 16 * DT * sizeof(float) + wave * KS1 * 1024;
    // (the second half of the scratch2 area belongs to the non-owner waves, which never use it: 4 KS1 KiB >= 4 x 512 DT bytes)
    char* const tscr = smem + NBUF * WB + SCR + TW * 64 * NS + 4 * 5 * 16 * DT * sizeof(float) + 4 * KS1 * 1024 + (wave & 3) * (32 * DT * 16);
    // lane masks of four packed bf16 values by nibble of activity bits (as in the forward kernels): the recomputed d hidden values are
    // cleared AFTER packing with table entries looked up a step ahead -- 4 ANDs between the H and the W1^T MFMAs instead of 8 x (bit
    // test, compare, select)
    unsigned* const klut = reinterpret_cast<unsigned*>(smem + NBUF * WB + SCR + TW * 64 * NS + 4 * 5 * 16 * DT * sizeof(float) + TW * KS1 * 1024);
    if (threadIdx.x < 32) {
        const unsigned n = threadIdx.x >> 1, hi = threadIdx.x & 1;
        klut[threadIdx.x] = ((n >> (2 * hi)) & 1u ? 0x0000ffffu : 0u) | ((n >> (2 * hi + 1)) & 1u ? 0xffff0000u : 0u);
    }
    // gamma2 | gamma1 behind the table: read by every lane as float4 where the LayerNorm backward needs them (as registers loaded at
    // the top they cost 20 VGPRs each across the prologue -- the kernel spilled; read from global where used, a memory round trip each)
    float* const gvec = reinterpret_cast<float*>(klut + 32);
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 2 * 16 * DT) {
        const int i = threadIdx.x - 64, which = i / (16 * DT), f = i - which * (16 * DT);
        gvec[i] = f < d.D ? (which ? a.g1 : a.g2)[f] : 0.f;
    }
    // F-split (struct FSplit): token block, chunk range and role of this workgroup
    const int nsp = d.fsplit, blk = nsp == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x, fq = nsp == 2 ? (int)(blockIdx.x & 1) : 0;
    const int NSH = NS / nsp, cbase = fq * NSH;
    const bool finisher = fq == nsp - 1;
    const int m = (blk * 4 + tile) * 16 + tok;
    const bool valid = m < M;
    const int rot = (int)(((unsigned)blk * 5u) % (unsigned)NSH);      // rotated chunk order per workgroup (see k_tr_ffn_fwd)
    auto issue = [&](int st) {
        int ce = st + rot;
        ce -= (ce >= NSH) ? NSH : 0;
        ce += cbase;
        const char* src = a.bffn + (size_t)ce * WB + lane * 16;
        char* dst = ring + (st % NBUF) * WB;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            int bb = wave + i * TW;
            bb %= 2 * NB;                 // padding copies repeat a block (uniform vmcnt bookkeeping)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + bb * 1024), LDS_PTR(dst + bb * 1024), 16, 0, 0);
        }
    };
    unsigned long long tprev = __builtin_readcyclecounter();
    (void)tprev;
    // (with the row-linear sums below the first weight DMAs are issued BEHIND the prologue's register loads: vmcnt retires in order, and
    //  a register load issued behind 66 KiB of DMA is only usable when those have landed)
    const bool dxg = a.dxg != 0;
    const bool rowsum = !dxg && a.rowsum && a.npart > 0;
    if (!rowsum && !dxg) {
        issue(0);
        if (NSH > 1) issue(1);
        if (NSH > 2) issue(2);
    }
    // column sums over this tile's 16 tokens -> colred[tile][slot][feature] (owner waves only)
    auto colsum = [&](int slot, const f32x4 (&t)[DT]) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sres = row_sum16(t[dt][r]);
                if (owner && tok == 0) colred[(tile * 5 + slot) * (16 * DT) + 16 * dt + 4 * g + r] = sres;
            }
    };
    // this wave's keep bytes of the whole F-half -> LDS (no vector-memory traffic inside the loop): loaded here, stored below
    constexpr int NAB = 4;                                  // 16-byte pieces held in registers (dim_ff <= 4096; longer rows: the loop below)
    u32x4 abr[NAB];
    {
        const unsigned char* srcb = a.active + ((size_t)(valid ? m : 0) * 4 + g) * (2 * NS) + fhw * NS;
#pragma unroll
        for (int i = 0; i < NAB; ++i) abr[i] = *reinterpret_cast<const u32x4*>(srcb + (16 * i < NS ? 16 * i : 0));
    }
    auto store_act = [&]() {
#pragma unroll
        for (int i = 0; i < NAB; ++i)
            if (16 * i < NS) *reinterpret_cast<u32x4*>(actB + lane * NS + 16 * i) = valid ? abr[i] : u32x4{0u, 0u, 0u, 0u};
        if (NS > 16 * NAB) {
            const unsigned char* srcb = a.active + ((size_t)(valid ? m : 0) * 4 + g) * (2 * NS) + fhw * NS;
            for (int c = 16 * NAB; c < NS; c += 16)
                *reinterpret_cast<u32x4*>(actB + lane * NS + c) = valid ? *reinterpret_cast<const u32x4*>(srcb + c) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    // ---- gradient of the layer output.  EVERY global read of the prologue is issued here, in front of the first wait: the LayerNorm2
    // input, the dropout bits of the FFN output, gamma2 and the owner's LayerNorm1 input / out-projection dropout bits were read where
    // they are used -- three dependent memory round trips inside "LN2 bwd + d f fragments" (14 K of the kernel's 86 K cycles at T = 252,
    // profiles/r06_train_ffn_bwd_phase_clocks.txt) and one more in front of the loop.
    f32x4 dy[DT];
    load_ctile<DT>(a.dy0, m, valid, D, g, dy);
    f32x4 xh[DT];
    load_ctile<DT>(a.s2, m, valid, D, g, xh);
    unsigned bits3[DT];
    row_drop_bits<DT>(d, a.rb3, m, valid, g, bits3);
    f32x4 s1t[DT];
    unsigned bits1[DT];
    if (owner) {
        load_ctile<DT>(a.s1, m, valid, D, g, s1t);
        row_drop_bits<DT>(d, a.rb1, m, valid, g, bits1);
    }
    if (!rowsum && !dxg) store_act();
    if (dxg) {
        // ---- d y += (d q | d k | d v of the next layer, all heads) . in_proj^T: the K = 16 MFMAs k_tr_attn_bwd used to run per head, here
        // once per token tile with fp32 accumulation over all heads.  The image (NP x 3 x DT half blocks, 45 KiB at 12 heads) goes
        // through the first three ring buffers, which the FFN weights take over afterwards; the two waves of a tile split the
        // (which, pair) combinations and exchange their partial sums through LDS like the row-linear sums below.
        const int NC = 3 * d.NP, NI = (NC + 1) / 2, NI0 = (NI + 1) / 2;      // combinations; 16-byte pairs of them; the first wave's share
        const int img_bytes = NC * DT * 512;
        for (int i = wave; i * 1024 < img_bytes; i += TW) {
            const int off = i * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds(GLB_PTR(a.winT + (off < img_bytes ? off : 0)), LDS_PTR(ring + i * 1024), 16, 0, 0);
        }
        constexpr int MAXI = 6;                                              // NP <= 8: at most 12 pairs of combinations, 6 per wave
        u32x4 bb[MAXI];
        const int i0 = fhw ? NI0 : 0, cnt = fhw ? NI - NI0 : NI0;
        {
            const char* rowp = reinterpret_cast<const char*>(a.dqkvR) + ((size_t)(valid ? m : 0) * 4 + g) * NC * 8;
#pragma unroll
            for (int i = 0; i < MAXI; ++i) bb[i] = *reinterpret_cast<const u32x4*>(rowp + (size_t)(i < cnt ? i0 + i : i0) * 16);
        }
        store_act();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f4zero();
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
            if (i < cnt) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = 2 * (i0 + i) + h;
                    if (c < NC) {
                        const int wh = c / d.NP, pr = c - wh * d.NP;
                        const u32x2 bw = {valid ? bb[i][2 * h] : 0u, valid ? bb[i][2 * h + 1] : 0u};
                        const s16x4 bf = __builtin_bit_cast(s16x4, bw);
                        const char* ab = ring + ((size_t)((pr * 3 + wh) * DT) * 64 + lane) * 8;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) o[dt] = MFMA16(*reinterpret_cast<const s16x4*>(ab + (size_t)dt * 512), bf, o[dt]);
                    }
                }
            }
        f32x4* const xs = reinterpret_cast<f32x4*>(fhw == 0 ? ring + 3 * WB + tile * (16 * 16 * DT * 4)
                                                            : smem + NBUF * WB + SCR + TW * 64 * NS + 4 * 5 * 16 * DT * sizeof(float) + tile * (16 * 16 * DT * 4));
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) xs[dt * 64 + lane] = o[dt];
        __syncthreads();
        {
            const f32x4* const x0 = reinterpret_cast<const f32x4*>(ring + 3 * WB + tile * (16 * 16 * DT * 4));
            const f32x4* const x1 = reinterpret_cast<const f32x4*>(smem + NBUF * WB + SCR + TW * 64 * NS + 4 * 5 * 16 * DT * sizeof(float) + tile * (16 * 16 * DT * 4));
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dy[dt] += x0[dt * 64 + lane];
                dy[dt] += x1[dt * 64 + lane];
            }
        }
        issue(0);                                      // (the image has been read: every wave is behind the barrier above)
        if (NSH > 1) issue(1);
        if (NSH > 2) issue(2);
    }
    if (rowsum) {
        // The partial tensors (one per head or head pair, written by the next layer's k_tr_attn_bwd) as C tiles are DT 8-byte (bf16) /
        // 16-byte (fp32) loads per lane and part, each walking 16 rows x 4 lane groups -- 60 narrow gathers per lane for 12 heads,
        // issued by BOTH waves of a tile: 16.6 K of this kernel's 86 K cycles at T = 252 (phase clocks, -DFD_TR_PROF_FB,
        // profiles/r06_train_ffn_bwd_phase_clocks.txt).  A tile's 16 rows of a part are ONE contiguous run (32 D or 64 D bytes):
        // the two waves of a tile take alternate parts, add them up as 16-byte units in that linear order (the sum does not care about
        // the layout), and exchange the two sums through LDS, from where they are read as C tiles.  Order of the additions:
        // d y = (d y0 + (p0 + p2 + ...)) + (p1 + p3 + ...).
        constexpr int NLB = (2 * 16 * DT + 63) / 64, NLF = (4 * 16 * DT + 63) / 64;       // 16-byte units per lane: bf16 / fp32 parts
        const int m0t = (blk * 4 + tile) * 16;
        // LDS: the fourth ring buffer (first written by step 0's DMA, behind the barrier in front of the loop) for the fhw = 0 waves,
        // the epilogue's scratch2 area for the others
        float* const rs = reinterpret_cast<float*>(fhw == 0 ? ring + 3 * WB + tile * (16 * 16 * DT * 4)
                                                            : smem + NBUF * WB + SCR + TW * 64 * NS + 4 * 5 * 16 * DT * sizeof(float) + tile * (16 * 16 * DT * 4));
        bool started = false;
        auto start_ring = [&]() {
            issue(0);
            if (NSH > 1) issue(1);
            if (NSH > 2) issue(2);
            store_act();
            started = true;
        };
        if (a.part_bf16) {
            const int NU = 2 * D;                                          // 16-byte units of the tile's rows
            const char* const pb = reinterpret_cast<const char*>(a.dyp);
            // (a unit that BEGINS inside the part is read in place -- with 2 D % 16 != 0 the last valid row's last unit reaches up to 8 bytes
            //  into the next part, or into the unused half of the fp32-sized parts buffer; units of rows beyond M read offset 0)
            const size_t pbytes = a.part_stride * 2, tot = (size_t)M * D * 2;
            float sum[NLB][8];
#pragma unroll
            for (int k = 0; k < NLB; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) sum[k][j] = 0.f;
            for (int pi = fhw; pi < a.npart; pi += 12) {                   // six parts of this wave per trip (12 heads: one trip), all loads in flight
                u32x4 raw[6][NLB];
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int pq = pi + 2 * q < a.npart ? pi + 2 * q : pi;
#pragma unroll
                    for (int k = 0; k < NLB; ++k) {
                        const size_t off = (size_t)m0t * D * 2 + (size_t)(lane + 64 * k) * 16;
                        raw[q][k] = *reinterpret_cast<const u32x4*>(pb + (size_t)pq * pbytes + (off < tot ? off : 0));
                    }
                }
                if (!started) start_ring();                                // behind the first trip's loads: the weight ring and the keep bytes
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    if (pi + 2 * q < a.npart) {
#pragma unroll
                        for (int k = 0; k < NLB; ++k)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                sum[k][2 * j] += __builtin_bit_cast(float, raw[q][k][j] << 16);
                                sum[k][2 * j + 1] += __builtin_bit_cast(float, raw[q][k][j] & 0xffff0000u);
                            }
                    }
            }
#pragma unroll
            for (int k = 0; k < NLB; ++k)
                if (lane + 64 * k < NU) {
                    float4* dst = reinterpret_cast<float4*>(rs + (size_t)(lane + 64 * k) * 8);
                    dst[0] = float4{sum[k][0], sum[k][1], sum[k][2], sum[k][3]};
                    dst[1] = float4{sum[k][4], sum[k][5], sum[k][6], sum[k][7]};
                }
        } else {
            const int NU = 4 * D;
            const char* const pb = reinterpret_cast<const char*>(a.dyp);
            const size_t pbytes = a.part_stride * 4, tot = (size_t)M * D * 4;          // (D % 4 == 0: no unit straddles two rows)
            f32x4 sum[NLF];
#pragma unroll
            for (int k = 0; k < NLF; ++k) sum[k] = f4zero();
            for (int pi = fhw; pi < a.npart; pi += 6) {                    // three parts of this wave per trip (6 head pairs: one trip)
                f32x4 raw[3][NLF];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int pq = pi + 2 * q < a.npart ? pi + 2 * q : pi;
#pragma unroll
                    for (int k = 0; k < NLF; ++k) {
                        const size_t off = (size_t)m0t * D * 4 + (size_t)(lane + 64 * k) * 16;
                        raw[q][k] = *reinterpret_cast<const f32x4*>(pb + (size_t)pq * pbytes + (off < tot ? off : 0));
                    }
                }
                if (!started) start_ring();
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (pi + 2 * q < a.npart) {
#pragma unroll
                        for (int k = 0; k < NLF; ++k) sum[k] += raw[q][k];
                    }
            }
#pragma unroll
            for (int k = 0; k < NLF; ++k)
                if (lane + 64 * k < NU) *reinterpret_cast<f32x4*>(rs + (size_t)(lane + 64 * k) * 4) = sum[k];
        }
        if (!started) start_ring();                                        // (a wave without a part: one head pair)
        __syncthreads();
        const float* const r0 = reinterpret_cast<const float*>(ring + 3 * WB + tile * (16 * 16 * DT * 4));
        const float* const r1 = reinterpret_cast<const float*>(smem + NBUF * WB + SCR + TW * 64 * NS + 4 * 5 * 16 * DT * sizeof(float) + tile * (16 * 16 * DT * 4));
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            if (d0 < D && valid) {
                dy[dt] += *reinterpret_cast<const f32x4*>(r0 + tok * D + d0);
                if (a.npart > 1) dy[dt] += *reinterpret_cast<const f32x4*>(r1 + tok * D + d0);
            }
        }
    }
    // (four partial tensors per trip, every load of the trip in flight at once: one exposed round trip per four parts; the order
    // of the additions is the part order either way)
    for (int pi = 0; pi < (a.rowsum ? 0 : a.npart); pi += 4) {
        f32x4 t[4][DT];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pq = pi + q < a.npart ? pi + q : a.npart - 1;
            if (a.part_bf16) load_ctile_bf16<DT>(reinterpret_cast<const __bf16*>(a.dyp) + (size_t)pq * a.part_stride, m, valid, D, g, t[q]);
            else load_ctile<DT>(a.dyp + (size_t)pq * a.part_stride, m, valid, D, g, t[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (pi + q < a.npart) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) dy[dt] += t[q][dt];
            }
    }
    if (!rowsum && !dxg) __syncthreads();  // (gvec: the other forms above have their own barriers)
    TRFB_STAMP(0, tprev);          // d y + its partial tensors
    // ---- LayerNorm2 backward
    float rstd2;
    {
        float mean;
        ln_stats<DT>(xh, D, g, mean, rstd2);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) xh[dt][r] = (16 * dt + 4 * g < D && valid) ? (xh[dt][r] - mean) * rstd2 : 0.f;
    }
    colsum(1, dy);                                   // d beta2
    {
        f32x4 t[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) t[dt] = dy[dt] * xh[dt];
        colsum(2, t);                                // d gamma2
    }
    {
        float4 g2v[DT];
        ln_gamma_load<DT>(gvec, D, g, g2v);          // (LDS: written at the top, two barriers ago)
        ln_bwd_tile<DT>(dy, xh, g2v, rstd2, D, g);   // dy = d s2
    }
    // ---- d f (FFN output after its dropout)
    f32x4 df[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) df[dt][r] = ((bits3[dt] >> r) & 1u) ? dy[dt][r] * d.keep_scale : 0.f;
    colsum(0, df);                                   // d b2
    // (the stores of d f follow the loop -- see k_tr_ffn_fwd -- and the epilogue's inputs are fetched now)
    bf16x8 dfr[KS1];
    ctile_to_frags<DT, KS1>(scratch, lane, D, df, false, dfr);
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f4zero();
    TRFB_STAMP(1, tprev);          // LN2 backward, column sums, d f fragments
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TRFB_STAMP(2, tprev);          // DMA wait + barrier
    // ---- d x1 += W1^T (active . W2^T d f) over this wave's F-half: d hidden lives in registers only.  As in k_tr_ffn_fwd the
    // fragments and the activity byte of a chunk are read one step ahead (two register sets, two steps per trip); the keep
    // scale of the hidden units is applied once to the accumulators behind the loop.
    unsigned act_cur = 0u;
    u32x2 kq0 = {0u, 0u}, kq1 = {0u, 0u};
    auto frags = [&](int c, bf16x8 (&w1)[2 * KS1], bf16x8 (&w2)[DT]) {
        const int cc = c < NSH ? c : NSH - 1;
        const char* wb = ring + (cc % NBUF) * WB + fhw * NB * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < 2 * KS1; ++i) w1[i] = *reinterpret_cast<const bf16x8*>(wb + i * 1024);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) w2[dt] = *reinterpret_cast<const bf16x8*>(wb + (2 * KS1 + dt) * 1024);
    };
    auto step = [&](int c, const bf16x8 (&w1)[2 * KS1], const bf16x8 (&w2)[DT], bf16x8 (&n1)[2 * KS1], bf16x8 (&n2)[DT]) {
#ifndef FD_TR_ABL_NODMA
        if (c + 3 < NSH) issue(c + 3);
#endif
        frags(c + 1, n1, n2);
        f32x4 h0 = f4zero(), h1 = f4zero();
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            h0 = MFMA(w1[ks], dfr[ks], h0);
            h1 = MFMA(w1[KS1 + ks], dfr[ks], h1);
        }
        int cn = c + 2 + rot;                         // (two steps deep: this step's table entries were looked up a step ago with the
        cn -= (cn >= NSH) ? NSH : 0;                  //  byte that had arrived by then; the byte of step c + 2 is requested now)
        cn -= (cn >= NSH) ? NSH : 0;                  // (beyond the last step: clamped reads, unused)
        cn -= (cn >= NSH) ? NSH : 0;
        const u32x2 k0 = kq0, k1 = kq1;
        kq0 = *reinterpret_cast<const u32x2*>(klut + 2 * (act_cur & 15u));
        kq1 = *reinterpret_cast<const u32x2*>(klut + 2 * (act_cur >> 4));
        act_cur = actB[lane * NS + cbase + cn];
        const u32x4 pkh = __builtin_bit_cast(u32x4, pack8(h0, h1));
        const bf16x8 hb = __builtin_bit_cast(bf16x8, u32x4{pkh[0] & k0[0], pkh[1] & k0[1], pkh[2] & k1[0], pkh[3] & k1[1]});
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] = MFMA(w2[dt], hb, acc[dt]);
        if (c + 3 < NSH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);
#ifndef FD_TR_ABL_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
    };
    {
        bf16x8 wa1[2 * KS1], wa2[DT], wb1[2 * KS1], wb2[DT];
        frags(0, wa1, wa2);
        act_cur = actB[lane * NS + cbase + rot];
        kq0 = *reinterpret_cast<const u32x2*>(klut + 2 * (act_cur & 15u));
        kq1 = *reinterpret_cast<const u32x2*>(klut + 2 * (act_cur >> 4));
        {
            int c1 = 1 + rot;
            c1 -= (c1 >= NSH) ? NSH : 0;
            act_cur = actB[lane * NS + cbase + c1];
        }
        for (int c = 0; c < NSH; c += 2) {       // (NS = F / 64 is a multiple of 4: F % 1024 == 0)
            step(c, wa1, wa2, wb1, wb2);
            step(c + 1, wb1, wb2, wa1, wa2);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] *= d.keep_scale;
    }
    TRFB_STAMP(3, tprev);          // chunk loop
    __syncthreads();
    if (!owner) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) xch[(tile * DT + dt) * 64 + lane] = acc[dt];
    }
    __syncthreads();
    TRFB_STAMP(4, tprev);          // exchange of the halves
    if (nsp == 2 && !finisher) {          // F-split producer: hand the FFN branch's partial d x1 over and leave (every wave of it)
        if (owner) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] += xch[(tile * DT + dt) * 64 + lane];
            fsplit_hand_over<DT>(a.fs, blk, tile, lane, acc);
        }
        return;
    }
    // The tile's second wave used to idle from here to the last barrier (14.7 K of the kernel's 79 K cycles at T = 252): it now takes
    // the out-projection^T product d att = d o W_o -- its fragments requested here, d o read from the owner's LDS fragments behind a
    // barrier -- while the owner stores d s1 and the d o T-blocks.
    bf16x8 wotf[DT][KS1];
    if (!owner) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) wotf[dt][ks] = *reinterpret_cast<const bf16x8*>(a.wot + ((size_t)(dt * KS1 + ks) * 64 + lane) * 16);
    }
    const int m0w = (blk * 4 + tile) * 16;
    if (owner) {
        stage_rows<DT, KS1>(a.stage, StageL<KS1, DT>::off_dr, m, valid, D, g, df, false);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] += xch[(tile * DT + dt) * 64 + lane];
        if (nsp == 2) fsplit_take_over<DT>(a.fs, blk, tile, lane, acc);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dy[dt] += acc[dt];                                          // d x1 = residual path + FFN branch
        // ---- LayerNorm1 backward
        float rstd1;
        {
            float mean;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) xh[dt] = s1t[dt];
            ln_stats<DT>(xh, D, g, mean, rstd1);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) xh[dt][r] = (16 * dt + 4 * g < D && valid) ? (xh[dt][r] - mean) * rstd1 : 0.f;
        }
        colsum(3, dy);                                   // d beta1
        {
            f32x4 t[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) t[dt] = dy[dt] * xh[dt];
            colsum(4, t);                                // d gamma1
        }
        {
            float4 g1v[DT];
            ln_gamma_load<DT>(gvec + 16 * DT, D, g, g1v);
            ln_bwd_tile<DT>(dy, xh, g1v, rstd1, D, g);   // dy = d s1
        }
        TRFB_STAMP(5, tprev);      // stage rows, LN1 backward
        store_ctile<DT>(a.dres, m, valid, D, g, dy);
        // ---- d o (out-projection output after its dropout) -> d att = d o W_o
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) df[dt][r] = ((bits1[dt] >> r) & 1u) ? dy[dt][r] * d.keep_scale : 0.f;
        // (a second scratch: the first one is aliased by the exchange area, which other owners may still be reading)
        ctile_to_frags<DT, KS1>(scratch2, lane, D, df, false, dfr);
    }
    __syncthreads();                   // the d o fragments of the four tiles are in LDS
    if (owner) {
        store_T16<DT>(tscr, a.doT + ((size_t)(m0w >> 5) * (16 * DT)) * 32 + (m0w & 31), 32, lane, D, df, false, valid);
    } else {
        const char* const osc = scratch2 - 4 * KS1 * 1024;          // the owner's (wave `tile`) scratch
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) dfr[ks] = *reinterpret_cast<const bf16x8*>(osc + (ks * 64 + lane) * 16);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 o = f4zero();
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) o = MFMA(wotf[dt][ks], dfr[ks], o);
            acc[dt] = o;
        }
        store_ctile<DT>(a.datt, m, valid, D, g, acc);
    }
    TRFB_STAMP(6, tprev);          // d o T-blocks, out-proj^T, stores
    // ---- column sums of the workgroup, tiles added in a fixed order
    __syncthreads();
    for (int i = threadIdx.x; i < 5 * 16 * DT; i += TW * 64) {
        const int slot = i / (16 * DT), f = i - slot * (16 * DT);
        float sres = 0.f;
        for (int w = 0; w < 4; ++w) sres += colred[(w * 5 + slot) * (16 * DT) + f];
        if (f < D) a.vecpart[((size_t)blk * 5 + slot) * D + f] = sres;
    }
    TRFB_STAMP(7, tprev);          // barrier + column sums
}

// ------------------------------------------------------------------------------------------------ attention backward
struct AttnBwdArgs {
    const __bf16* x0rb;
    const float* att;         // (M, D) forward output O (after dropout scaling)
    const float* datt;        // (M, D)
    const float* lse2;
    const unsigned char* pmask;
    float* dxp;               // [NP][M, D]: this pair's contribution to the layer-input gradient (OH: [2 NP][M, D], fp32 or bf16)
    int part_bf16;
    __bf16* dqkvT;            // T-block with 3*NP*16 rows: row which*(NP*16) + pair*16 + (8 hs + dim)
    const char* wk; const char* wv; const char* wq;
    const char* winT;         // [pair][which][DT] half blocks
    size_t part_stride;
    // dxg != 0 (default): no partial tensors -- the d(q | k | v) values of every token go out ONCE more, as rows [token][lane group g]
    // [which * NP + pair][4 bf16] (the K = 16 MFMA's B operand of lane (token, g), 144 contiguous bytes per (token, g) at 12 heads), and
    // the consumer of d x (the next k_tr_ffn_bwd, k_tr_dx0 behind layer 0) multiplies them by in_proj^T itself: 576 B per token written
    // and read instead of H x 144 B (12 heads: 1728 B), and the in-proj^T MFMAs of all heads accumulate in fp32.
    __bf16* dqkvR; int dxg;
};

#ifdef FD_TR_PROF_ATTN      // variant build: in-kernel phase clocks of k_tr_attn_bwd (workgroup (0, 0), per wave), printed after 30 launches
__device__ unsigned long long fd_tr_attn_dbg[8 * 8];
#define TRA_STAMP(slot, t_prev)                                                                          \
    do {                                                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter();                                    \
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) fd_tr_attn_dbg[wave * 8 + (slot)] += now_ - (t_prev); \
        (t_prev) = now_;                                                                                 \
    } while (0)
#else
#define TRA_STAMP(slot, t_prev) do { } while (0)
#endif
// OH = 0: one workgroup per (head pair, series), both heads of the pair in every sweep iteration.
// OH = 1: one workgroup per (head, series): H x B workgroups of NW waves with the LDS images of ONE head (row forms 16 B per
//         token, column forms 8 dim rows), so that three or four workgroups share a CU and the grid is a whole number of rounds
//         (H = 12, B = 64: 768 workgroups = 3 per CU; the pair form is 384 workgroups of 8 waves at one per CU: 1.5 rounds).
//         A lane whose k-slots (row forms) / dim row (column forms) belong to the pair's other head supplies zeros without
//         reading the LDS; d x parts are per head (a.dxp holds H of them).
template <int KS1, int DT, int NW, int OH>
__global__ __launch_bounds__(NW * 64, OH ? FD_TR_ATTN_OH_MINW : (NW == 8 ? FD_TR_ATTN_MINW : 1)) void k_tr_attn_bwd(const TrDims d, const AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NHS = OH ? 1 : 2;                              // heads swept by this workgroup
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bx, b;
    xcd_deal(d, bx, b);                                          // (the heads of a series share an L2)
    const int pair = OH ? (bx >> 1) : bx;
    const int hs0 = OH ? (bx & 1) : 0;                           // OH: the head of the pair this workgroup owns
    const int T = d.T, KT = d.KT, NJ = d.NJ, NTOK = KT * 16, hd = d.hd, H = d.H, D = d.D;
    // "row" form [token][4 g][8 B] (16x16x16 operand with the pair's 16 dim slots as k; OH: [token][2][8 B]) and "column" form
    // [32-token block][4 g][16 dim rows][16 B] (16x16x32 A operand with 32 tokens as k; OH: 8 dim rows) of q, k, v, dO
    const size_t RSZ = (size_t)NTOK * (OH ? 16 : 32), CSZ = (size_t)NJ * (OH ? 512 : 1024);
    char* const qR = smem;            char* const kR = qR + RSZ;  char* const vR = kR + RSZ;  char* const oR = vR + RSZ;
    char* const qC = oR + RSZ;        char* const kC = qC + CSZ;  char* const oC = kC + CSZ;
    float* const drow = reinterpret_cast<float*>(oC + CSZ);      // [NHS][NTOK]  rowsum(dO . O) per (head, query)
    float* const lse = drow + NHS * NTOK;                        // [NHS][NTOK]
    // keep bits of the swept heads, [NHS][T][NJ][4] bytes, staged once: the key-owner sweep reads 8 scattered bytes per
    // (query pair, head) and a global gather there was a dependent L2 round trip per iteration
    unsigned char* const pm = reinterpret_cast<unsigned char*>(lse + NHS * NTOK);
    const int PMH = T * NJ * 4;                                  // bytes per head
    // keep multipliers of four scores by nibble of keep bits: lut[n] = {bit r of n ? keep_scale : 0}.  One 16-byte LDS read per
    // four scores instead of and + compare + select per score (24 of the 94 VALU instructions of a key-sweep iteration)
    float* const lut = reinterpret_cast<float*>(pm + (((size_t)NHS * PMH + 15) & ~(size_t)15));
    if (threadIdx.x < 64) lut[threadIdx.x] = ((threadIdx.x >> 2) >> (threadIdx.x & 3)) & 1u ? d.keep_scale : 0.f;
    // key-oriented copy of the same bits for the key-owner sweep, [NHS][NTOK keys][NJ][4]: byte (key, jq, gq) = bits of queries
    // 32 jq + 4 gq + r (bit r) and 32 jq + 16 + 4 gq + r (bit 4 + r); built from `pm` by the workgroup itself once it has landed
    // (round 6: [query block jq][key + key / 32] dwords, byte gq -- a key's dword sat at a 32-byte stride, so the 16 keys a wave reads
    //  per sweep iteration shared four banks and the 64 lanes of a transposition store ONE; now consecutive keys are consecutive
    //  dwords, and the + key / 32 spreads the transposition's keys 32 apart over different banks)
    unsigned char* const pmT = reinterpret_cast<unsigned char*>(lut + 64);
    const int PTRS = NTOK + (NTOK >> 5) + 1;                     // dwords per query block
    const int PTH = NJ * PTRS * 4;                               // bytes per head
    const size_t pstride = (size_t)KS1 * 1024;
    auto wfrag = [&](const char* img, int ks) { return *reinterpret_cast<const bf16x8*>(img + pair * pstride + ((size_t)ks * 64 + lane) * 16); };
    unsigned long long tprev = 0;
    (void)tprev;
#ifdef FD_TR_PROF_ATTN
    tprev = __builtin_readcyclecounter();
#endif
    const bool lo_grp = (g >> 1) == 0;
    const int myhead = 2 * pair + (g >> 1);
    const bool mineR = !OH || (g >> 1) == hs0;                   // this lane's k-slots / C rows belong to a swept head
    const bool mineC = !OH || (tok >> 3) == hs0;                 // this lane's dim row / column belongs to a swept head
    if (d.p > 0.f) {
        for (int hi = 0; hi < NHS; ++hi) {
            const int head = 2 * pair + (OH ? hs0 : hi);
            if (head >= H) continue;
            const unsigned char* src = a.pmask + ((size_t)b * H + head) * PMH;        // contiguous per (series, head)
            if ((PMH & 1023) == 0 || (PMH & 15) == 0) {
                // straight into the LDS (1 KiB per wave and instruction, no registers, nothing waits here: the copy runs under
                // the staging below and is awaited in front of its barrier; it was 3.5 K clocks of a 70 K-clock workgroup)
                const int nkib = PMH >> 10;
                for (int c = wave; c < nkib; c += NW)
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src + (size_t)c * 1024 + lane * 16), LDS_PTR(pm + hi * PMH + c * 1024), 16, 0, 0);
                for (int i = nkib * 1024 + threadIdx.x * 16; i < PMH; i += NW * 64 * 16)      // the last partial KiB
                    *reinterpret_cast<u32x4*>(pm + hi * PMH + i) = *reinterpret_cast<const u32x4*>(src + i);
            } else {
                for (int i = threadIdx.x * 4; i < PMH; i += NW * 64 * 4)                   // PMH is a multiple of 4
                    *reinterpret_cast<unsigned*>(pm + hi * PMH + i) = *reinterpret_cast<const unsigned*>(src + i);
            }
        }
    }
    // LDS offsets of this lane's row-form / column-form words of a token tile (writers; the readers are rfrag / cfrag)
    auto row_off = [&](int tile) -> size_t {
        return OH ? ((size_t)(tile * 16 + tok) * 2 + (g & 1)) * 8 : ((size_t)(tile * 16 + tok) * 4 + g) * 8;
    };
    auto col_off = [&](int tile) -> size_t {
        return OH ? ((size_t)((tile >> 1) * 4 + g) * 8 + (tok & 7)) * 16 + 8 * (tile & 1)
                  : ((size_t)((tile >> 1) * 4 + g) * 16 + tok) * 16 + 8 * (tile & 1);
    };
    // ---- stage q, k, v (recomputed), dO, rowsum(dO.O), lse
    {
        bf16x8 wqf[KS1], wkf[KS1], wvf[KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) { wqf[ks] = wfrag(a.wq, ks); wkf[ks] = wfrag(a.wk, ks); wvf[ks] = wfrag(a.wv, ks); }
        // Every global operand of a tile (x rows, dO / O rows, lse, the dO column values) is requested one tile ahead, from clamped
        // addresses without branches: a tile was load -> wait -> 15 MFMAs -> load -> wait ..., 8 K clocks each with two tiles
        // per wave (in-kernel clocks: staging 16 K of a 70 K-clock workgroup).
        auto fetch = [&](int kt, bf16x8 (&xf)[KS1], f32x4_a4& dv, f32x4_a4& av, float& ls, f32x4& doc) {
            const int t = kt * 16 + tok, tc = t < T ? t : T - 1;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
                xf[ks] = *reinterpret_cast<const bf16x8*>(a.x0rb + (size_t)(b * T + tc) * d.RBW + 32 * ks + 8 * g);
            const int hc = myhead < H ? myhead : H - 1;
            const size_t off = (size_t)(b * T + tc) * D + hc * hd + 4 * (g & 1);
            dv = *reinterpret_cast<const f32x4_a4*>(a.datt + off);
            av = *reinterpret_cast<const f32x4_a4*>(a.att + off);
            ls = a.lse2[((size_t)b * H + hc) * T + tc];
            const int hs = tok >> 3, dd = tok & 7, head = min(2 * pair + hs, H - 1), ddc = dd < hd ? dd : hd - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tt = min(kt * 16 + 4 * g + r, T - 1);
                doc[r] = a.datt[((size_t)b * T + tt) * D + head * hd + ddc];
            }
        };
        bf16x8 cxf[KS1], nxf[KS1];
        f32x4_a4 cdv, cav, ndv, nav;
        float cls = 0.f, nls = 0.f;
        f32x4 cdoc = f4zero(), ndoc = f4zero();
        if (wave < KT) fetch(wave, cxf, cdv, cav, cls, cdoc);
        for (int kt = wave; kt < KT; kt += NW) {
            fetch(kt + NW < KT ? kt + NW : kt, nxf, ndv, nav, nls, ndoc);
            f32x4 qr = f4zero(), kr = f4zero(), vr = f4zero(), qc = f4zero(), kc = f4zero();
            const int t = kt * 16 + tok;
            const bool tv = t < T;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const u32x4 raw = __builtin_bit_cast(u32x4, cxf[ks]);
                const bf16x8 xf = __builtin_bit_cast(bf16x8, u32x4{tv ? raw[0] : 0u, tv ? raw[1] : 0u, tv ? raw[2] : 0u, tv ? raw[3] : 0u});
                qr = MFMA(wqf[ks], xf, qr);       // [dim rows][token col] -> row form
                kr = MFMA(wkf[ks], xf, kr);
                vr = MFMA(wvf[ks], xf, vr);
                qc = MFMA(xf, wqf[ks], qc);       // [token rows][dim col] -> column form
                kc = MFMA(xf, wkf[ks], kc);
            }
            const size_t ro = row_off(kt);
            if (mineR) {
                *reinterpret_cast<s16x4*>(qR + ro) = pack4(qr);
                *reinterpret_cast<s16x4*>(kR + ro) = pack4(kr);
                *reinterpret_cast<s16x4*>(vR + ro) = pack4(vr);
            }
            const size_t co = col_off(kt);
            const bool odd_tail = (KT & 1) && kt == KT - 1;
            if (mineC) {
                *reinterpret_cast<s16x4*>(qC + co) = pack4(qc);
                *reinterpret_cast<s16x4*>(kC + co) = pack4(kc);
                if (odd_tail) {
                    *reinterpret_cast<u32x2*>(qC + co + 8) = u32x2{0u, 0u};
                    *reinterpret_cast<u32x2*>(kC + co + 8) = u32x2{0u, 0u};
                }
            }
            // dO rows and rowsum(dO . O) of this lane's head (4 of its dims per lane, the lane pair g, g^1 holds all 8); dims >=
            // head_dim were read from the next head / row (the buffers are followed by others in the arena) and are dropped
            f32x4 dor = f4zero();
            float part = 0.f;
            const bool ok = tv && myhead < H;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool use = ok && 4 * (g & 1) + r < hd;
                dor[r] = use ? cdv[r] : 0.f;
                part += use ? cdv[r] * cav[r] : 0.f;
            }
            if (mineR) *reinterpret_cast<s16x4*>(oR + ro) = pack4(dor);
            float ea, eb;
            swap16(part, ea, eb);
            if ((g & 1) == 0 && mineR) {
                const int hi = OH ? 0 : (g >> 1);
                // both stored NEGATED: -lse is the C operand of the score MFMAs (S - lse leaves the matrix pipe, no subtraction per
                // score), -rowsum(dO . O) the addend of d S = P (d P keep - D)
                drow[hi * NTOK + kt * 16 + tok] = -(ea + eb);
                // padded queries / a missing odd head: lse = +1e30 makes every P = exp2(s - lse) of that row exactly 0, so the
                // sweeps need no validity selects (padded KEYS have all-zero K / V / dO operands instead)
                lse[hi * NTOK + kt * 16 + tok] = ok ? -cls : -1.0e30f;
            }
            // dO column form: lane (dim row = tok, g) holds tokens 4g+r of this tile
            if (mineC) {
                const int hs = tok >> 3, dd = tok & 7, head = 2 * pair + hs;
                f32x4 doc = f4zero();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tt = kt * 16 + 4 * g + r;
                    doc[r] = (tt < T && head < H && dd < hd) ? cdoc[r] : 0.f;
                }
                *reinterpret_cast<s16x4*>(oC + co) = pack4(doc);
                if (odd_tail) *reinterpret_cast<u32x2*>(oC + co + 8) = u32x2{0u, 0u};
            }
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) cxf[ks] = nxf[ks];
            cdv = ndv; cav = nav; cls = nls; cdoc = ndoc;
        }
    }
    TRA_STAMP(0, tprev);                                       // staging (this wave's tiles)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the keep-bit DMA)
    __syncthreads();
    TRA_STAMP(1, tprev);                                       // DMA wait + barrier
    // The key-owner sweep needs, per key, the keep bits of eight QUERIES (gathering them from the query-oriented bytes inside the
    // sweep cost eight LDS byte reads with their address arithmetic per iteration).  Rounds 2-4 had a side-stream kernel write a
    // key-oriented copy to global memory (k_tr_masks_T: 25 us per layer at T = 252 beside the forward chain, and a global load
    // per sweep iteration); the workgroup now transposes its own head's bits in LDS: an item = the 8 x 8 bit block {8 queries
    // of lane group gq} x {8 keys of lane group gk} of (query block jq, key block jb), read as eight bytes (one per query),
    // transposed in registers (three masked exchange steps on a 64-bit word), written as eight bytes (one per key).
    if (d.p > 0.f) {
        const int RB = NJ * 4;
        // tasks (head, query block jq, key block jb, query lane group gq) over all threads of the workgroup: NHS x NJ x NJ x 4 of them
        // (256 at T = 252 with one head per workgroup, 128 at T = 100 with a head pair); a quad = gq 0..3 of one (head, jq, jb)
        const float inv_rb = 1.0f / (float)RB;
        for (int task = threadIdx.x; task < NHS * NJ * RB; task += NW * 64) {
            {
                const int hj = (int)(((float)task + 0.5f) * inv_rb);      // (head, jq): exact for these small integers
                const int rest = task - hj * RB;
                const int hi = NHS == 1 ? 0 : (hj >= NJ ? 1 : 0), jq = hj - hi * NJ;
                const unsigned char* src = pm + hi * PMH;
                unsigned char* dst = pmT + hi * PTH;
                // a lane = (key block jb, query lane group gq): the eight query rows' DWORDS hold the bytes of all four key lane groups gk
                // (eight 4-byte reads for four items; the first form read eight single bytes per item)
                const int jb = rest >> 2, gq = rest & 3;
                unsigned qw[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q0 = min(32 * jq + 4 * gq + e, T - 1), q1 = min(32 * jq + 16 + 4 * gq + e, T - 1);
                    qw[e] = *reinterpret_cast<const unsigned*>(src + q0 * RB + jb * 4);
                    qw[4 + e] = *reinterpret_cast<const unsigned*>(src + q1 * RB + jb * 4);
                }
                unsigned* const dw = reinterpret_cast<unsigned*>(dst) + jq * PTRS;
#pragma unroll
                for (int gk = 0; gk < 4; ++gk) {
                    unsigned lo = 0u, hi32 = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo |= ((qw[e] >> (8 * gk)) & 0xffu) << (8 * e);
                        hi32 |= ((qw[4 + e] >> (8 * gk)) & 0xffu) << (8 * e);
                    }
                    unsigned long long x = (unsigned long long)lo | ((unsigned long long)hi32 << 32), t;
                    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;  x = x ^ t ^ (t << 7);
                    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x = x ^ t ^ (t << 14);
                    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x = x ^ t ^ (t << 28);
                    // byte f of x = the bits of key f of this item for the queries of lane group gq; the four lanes of a quad are gq = 0..3
                    // of one jb: a 4 x 4 byte transposition inside the quad (two masked exchanges per 32-bit half) leaves lane gq with the
                    // complete dwords of keys f = gq and f = 4 + gq -- two 4-byte stores per item instead of eight single bytes
                    auto quad_t = [&](unsigned v) {
                        const unsigned a = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
                        v = (gq & 1) ? (((a >> 8) & 0x00FF00FFu) | (v & 0xFF00FF00u)) : ((v & 0x00FF00FFu) | ((a & 0x00FF00FFu) << 8));
                        const unsigned c = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
                        return (gq & 2) ? ((c >> 16) | (v & 0xFFFF0000u)) : ((v & 0x0000FFFFu) | (c << 16));
                    };
                    const unsigned w0 = quad_t((unsigned)x), w1 = quad_t((unsigned)(x >> 32));
                    const int k0 = 32 * jb + 4 * gk + gq, k1 = k0 + 16;
                    if (k0 < NTOK) dw[k0 + (k0 >> 5)] = w0;
                    if (k1 < NTOK) dw[k1 + (k1 >> 5)] = w1;
                }
            }
        }
        __syncthreads();
    }
    TRA_STAMP(2, tprev);                                       // keep-bit transposition + barrier
    auto rfrag = [&](const char* base, int tile) {
        if constexpr (OH) {
            const u32x2 v = *reinterpret_cast<const u32x2*>(base + ((size_t)(tile * 16 + tok) * 2 + (g & 1)) * 8);
            return __builtin_bit_cast(s16x4, u32x2{mineR ? v[0] : 0u, mineR ? v[1] : 0u});
        } else {
            return *reinterpret_cast<const s16x4*>(base + ((size_t)(tile * 16 + tok) * 4 + g) * 8);
        }
    };
    // OH, operands read inside the sweeps: no zero fill.  A lane of the other head's k-slots meets the zeros of the per-tile operand
    // (q / dO rows in the query sweep, k / v rows in the key sweep: rfrag); a lane of the other head's dim ROW of a column form
    // produces a C row that the epilogue drops.  What such a lane reads is its own head's (finite) data at the same g & 1 / tok & 7.
    auto rfrag_loop = [&](const char* base, int tile) {
        if constexpr (OH) return *reinterpret_cast<const s16x4*>(base + ((size_t)(tile * 16 + tok) * 2 + (g & 1)) * 8);
        else return *reinterpret_cast<const s16x4*>(base + ((size_t)(tile * 16 + tok) * 4 + g) * 8);
    };
    auto cfrag = [&](const char* base, int jb) {
        if constexpr (OH) return *reinterpret_cast<const bf16x8*>(base + ((size_t)(jb * 4 + g) * 8 + (tok & 7)) * 16);
        else return *reinterpret_cast<const bf16x8*>(base + ((size_t)(jb * 4 + g) * 16 + tok) * 16);
    };
    auto headmask = [&](s16x4 v, int hs) {        // keep only the k-slots of head hs of the pair (OH: rfrag has done it)
        if constexpr (OH) return v;
        const u32x2 u = __builtin_bit_cast(u32x2, v);
        const bool mine = (g >> 1) == hs;
        return __builtin_bit_cast(s16x4, u32x2{mine ? u[0] : 0u, mine ? u[1] : 0u});
    };
    const float ln2 = 0.6931471805599453f;
    const float inv_sqrt_hd = __builtin_amdgcn_rsqf((float)hd);
    // each wave owns token tiles tt = wave, wave+NW, ...: as QUERY tile (d q), then as KEY tile (d k, d v), then the
    // input gradient of in_proj for those 16 tokens
    for (int tt = wave; tt < KT; tt += NW) {
        f32x4 dq[NHS], dk[NHS], dv[NHS];
#pragma unroll
        for (int hi = 0; hi < NHS; ++hi) { dq[hi] = f4zero(); dk[hi] = f4zero(); dv[hi] = f4zero(); }
        // ---------------- as query tile: S^T tiles [key rows 4g+r][query col]
        {
            const int t = tt * 16 + tok;
            s16x4 qb[NHS], ob[NHS];
            f32x4 lq[NHS];           // -lse of this lane's query in all four C rows
            float dr[NHS];           // -rowsum(dO . O)
            {
                const s16x4 qf = rfrag(qR, tt), of = rfrag(oR, tt);
#pragma unroll
                for (int hi = 0; hi < NHS; ++hi) {
                    qb[hi] = headmask(qf, hi);
                    ob[hi] = headmask(of, hi);
                    const float nl = lse[hi * NTOK + tt * 16 + tok];
                    lq[hi] = f32x4{nl, nl, nl, nl};
                    dr[hi] = drow[hi * NTOK + tt * 16 + tok];
                }
            }
            const int tcl = t < T ? t : T - 1;                       // (padded query columns are discarded at the end)
            for (int jb = 0; jb < NJ; ++jb) {
                // a missing odd key tile re-reads tile ka: its half of the K column block is zero, so it adds nothing to d q;
                // padded keys likewise (zero K columns); no per-score validity selects
                const int ka = 2 * jb, kb = (2 * jb + 1 < KT) ? 2 * jb + 1 : ka;
                const s16x4 kfa = rfrag_loop(kR, ka), kfb = rfrag_loop(kR, kb), vfa = rfrag_loop(vR, ka), vfb = rfrag_loop(vR, kb);
                const bf16x8 kcf = cfrag(kC, jb);
#pragma unroll
                for (int hi = 0; hi < NHS; ++hi) {
                    f32x4 sa = MFMA16(kfa, qb[hi], lq[hi]), sb = MFMA16(kfb, qb[hi], lq[hi]);        // S - lse
                    f32x4 pa = MFMA16(vfa, ob[hi], f4zero()), pb = MFMA16(vfb, ob[hi], f4zero());    // dP (dropped P's gradient)
                    unsigned bits = 0xffu;
                    if (d.p > 0.f) bits = pm[hi * PMH + (tcl * NJ + jb) * 4 + g];
                    const f32x4 ma = *reinterpret_cast<const f32x4*>(lut + 4 * (bits & 15u));      // keep multipliers of the scores
                    const f32x4 mb = *reinterpret_cast<const f32x4*>(lut + 4 * (bits >> 4));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float Pa = __builtin_amdgcn_exp2f(sa[r]);
                        const float Pb = __builtin_amdgcn_exp2f(sb[r]);
                        sa[r] = Pa * __builtin_fmaf(pa[r], ma[r], dr[hi]);
                        sb[r] = Pb * __builtin_fmaf(pb[r], mb[r], dr[hi]);
                    }
                    dq[hi] = MFMA(kcf, pack8(sa, sb), dq[hi]);       // [dim rows][query col] += K^T dS^T
                }
            }
        }
        TRA_STAMP(3, tprev);                                   // query-owner sweeps
        // in-proj^T fragments of the epilogue, requested here: their L2 round trip runs under the key-owner sweep (as 15 loads at the
        // head of every tile's epilogue they were exposed: 3.1 K of a tile's 20 K clocks went into the epilogue)
        s16x4 wie[3][DT];
        if (!a.dxg) {
            const char* wbase = a.winT + (size_t)pair * 3 * DT * 512;
#pragma unroll
            for (int wh = 0; wh < 3; ++wh)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) wie[wh][dt] = *reinterpret_cast<const s16x4*>(wbase + ((size_t)(wh * DT + dt) * 64 + lane) * 8);
        }
        // ---------------- as key tile: S tiles [query rows 4g+r][key col]
        {
            const int kt = tt;
            // the head's k-slots are selected on the key side once per tile (masking either operand of the contraction does)
            s16x4 kfh[NHS], vfh[NHS];
            {
                const s16x4 kf = rfrag(kR, kt), vf = rfrag(vR, kt);
#pragma unroll
                for (int hi = 0; hi < NHS; ++hi) { kfh[hi] = headmask(kf, hi); vfh[hi] = headmask(vf, hi); }
            }
            // keep bits in the key-oriented layout: one 32-bit word per (key, query block) and head holds the four lane groups'
            // bytes (padded keys: never written, their K / V / dO operands are zero); read one block ahead
            const unsigned char* pt[NHS];
            unsigned wn[NHS];
#pragma unroll
            for (int hi = 0; hi < NHS; ++hi) {
                pt[hi] = pmT + hi * PTH + ((kt * 16 + tok) + ((kt * 16 + tok) >> 5)) * 4;
                wn[hi] = 0xffffffffu;
                if (d.p > 0.f) wn[hi] = *reinterpret_cast<const unsigned*>(pt[hi]);
            }
            for (int jq = 0; jq < NJ; ++jq) {
                const int qa_t = 2 * jq, qb_t = (2 * jq + 1 < KT) ? 2 * jq + 1 : qa_t;
                unsigned wc[NHS];
#pragma unroll
                for (int hi = 0; hi < NHS; ++hi) {
                    wc[hi] = wn[hi];
                    if (d.p > 0.f && jq + 1 < NJ) wn[hi] = *reinterpret_cast<const unsigned*>(pt[hi] + (size_t)(jq + 1) * PTRS * 4);
                }
                const s16x4 qfa = rfrag_loop(qR, qa_t), qfb = rfrag_loop(qR, qb_t), ofa = rfrag_loop(oR, qa_t), ofb = rfrag_loop(oR, qb_t);
                const bf16x8 qcf = cfrag(qC, jq), ocf = cfrag(oC, jq);
#pragma unroll
                for (int hi = 0; hi < NHS; ++hi) {
                    // padded query rows carry lse = 1e30 (P = 0); a missing odd query tile re-reads tile qa_t against zero halves
                    // of the Q / dO column blocks
                    const f32x4 la = *reinterpret_cast<const f32x4*>(lse + hi * NTOK + qa_t * 16 + 4 * g);      // (-lse)
                    const f32x4 lb = *reinterpret_cast<const f32x4*>(lse + hi * NTOK + qb_t * 16 + 4 * g);
                    const f32x4 da = *reinterpret_cast<const f32x4*>(drow + hi * NTOK + qa_t * 16 + 4 * g);     // (-rowsum)
                    const f32x4 db = *reinterpret_cast<const f32x4*>(drow + hi * NTOK + qb_t * 16 + 4 * g);
                    f32x4 sa = MFMA16(qfa, kfh[hi], la), sb = MFMA16(qfb, kfh[hi], lb);               // S - lse
                    f32x4 pa = MFMA16(ofa, vfh[hi], f4zero()), pb = MFMA16(ofb, vfh[hi], f4zero());
                    const unsigned bits = (wc[hi] >> (8 * g)) & 0xffu;
                    const f32x4 ma = *reinterpret_cast<const f32x4*>(lut + 4 * (bits & 15u));      // keep multipliers of the scores
                    const f32x4 mb = *reinterpret_cast<const f32x4*>(lut + 4 * (bits >> 4));
                    f32x4 pda, pdb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float Pa = __builtin_amdgcn_exp2f(sa[r]);
                        const float Pb = __builtin_amdgcn_exp2f(sb[r]);
                        pda[r] = Pa * ma[r];
                        pdb[r] = Pb * mb[r];
                        sa[r] = Pa * __builtin_fmaf(pa[r], ma[r], da[r]);
                        sb[r] = Pb * __builtin_fmaf(pb[r], mb[r], db[r]);
                    }
                    dv[hi] = MFMA(ocf, pack8(pda, pdb), dv[hi]);     // [dim rows][key col] += dO^T P_drop
                    dk[hi] = MFMA(qcf, pack8(sa, sb), dk[hi]);       // += Q^T dS
                }
            }
        }
        TRA_STAMP(4, tprev);                                   // key-owner sweeps
        // ---------------- combine the heads' rows, scale, write d(qkv) and the in_proj input gradient
        {
            const int t = tt * 16 + tok, mm = b * T + t;
            const bool tv = t < T;
            f32x4 gq, gk, gv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool dv_ok = (4 * (g & 1) + r < hd) && myhead < H && tv && mineR;
                // s_nat = q_raw . k / sqrt(hd);  q_img = q_raw log2(e)/sqrt(hd)
                gq[r] = dv_ok ? (lo_grp || OH ? dq[0][r] : dq[NHS - 1][r]) * inv_sqrt_hd : 0.f;      // d q_raw
                gk[r] = dv_ok ? (lo_grp || OH ? dk[0][r] : dk[NHS - 1][r]) * ln2 : 0.f;              // d k = sum dS q_img ln2
                gv[r] = dv_ok ? (lo_grp || OH ? dv[0][r] : dv[NHS - 1][r]) : 0.f;
            }
            const s16x4 bq = pack4(gq), bk = pack4(gk), bv = pack4(gv);
            const int rows = d.NP * 16;
            __bf16* tcol = a.dqkvT + ((size_t)(mm >> 5) * (3 * rows)) * 32 + (mm & 31);
            if (t < ((T + 15) & ~15)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = pair * 16 + 4 * g + r;
                    if (tv && mineR) {
                        tcol[(size_t)(0 * rows + j) * 32] = (__bf16)gq[r];
                        tcol[(size_t)(1 * rows + j) * 32] = (__bf16)gk[r];
                        tcol[(size_t)(2 * rows + j) * 32] = (__bf16)gv[r];
                    }
                }
            }
            if (a.dxg) {
                if (tv && mineR) {
                    u32x2* row = reinterpret_cast<u32x2*>(a.dqkvR) + ((size_t)mm * 4 + g) * (3 * d.NP) + pair;
                    row[0] = __builtin_bit_cast(u32x2, bq);
                    row[d.NP] = __builtin_bit_cast(u32x2, bk);
                    row[2 * d.NP] = __builtin_bit_cast(u32x2, bv);
                }
            }
            const size_t pidx = OH ? (size_t)bx : (size_t)pair;
#pragma unroll
            for (int dt = 0; dt < (a.dxg ? 0 : DT); ++dt) {
                f32x4 o = f4zero();
                o = MFMA16(wie[0][dt], bq, o);
                o = MFMA16(wie[1][dt], bk, o);
                o = MFMA16(wie[2][dt], bv, o);
                const int d0 = 16 * dt + 4 * g;
                if (tv && d0 < D) {
                    if (OH && a.part_bf16)
                        *reinterpret_cast<s16x4*>(reinterpret_cast<__bf16*>(a.dxp) + pidx * a.part_stride + (size_t)mm * D + d0) = pack4(o);
                    else
                        *reinterpret_cast<float4*>(a.dxp + pidx * a.part_stride + (size_t)mm * D + d0) = float4{o[0], o[1], o[2], o[3]};
                }
            }
        }
        TRA_STAMP(5, tprev);                                   // epilogues
    }
}

// ------------------------------------------------------------------------------------------------ weight gradients
struct WgLayer {
    const __bf16* x0T; const __bf16* attT; const __bf16* doT; const __bf16* dqkvT;
    const char* stage;
    const unsigned char* active;      // (Mpad, 4, F/32) activity bytes of the layer (k_tr_ffn_fwd / k_tr_fwd_layers)
    const char* ffn_img; const char* bffn;
    long long in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w;
};
struct WgArgs {
    float* part;              // [TS][nparams]
    long long nparams;
    int TS, nblk;             // token splits; 32-token blocks in total
};

// T-block fragment straight from global memory: rows 16*rt + (lane&15), tokens 8g..8g+7 of the block (natural order)
__device__ __forceinline__ bf16x8 t_frag(const __bf16* __restrict__ tb, int blk, int NF, int rt, int lane, int nvalid) {
    const int row = lane & 15, g = lane >> 4;
    u32x4 v = *reinterpret_cast<const u32x4*>(tb + ((size_t)blk * NF + 16 * rt + row) * 32 + 8 * g);
    if (nvalid < 32) {       // only the very last block of the batch; the empty asm keeps hipcc from if-converting the (uniform)
        asm volatile("");    // branch into 16 selects executed for every fragment of every block
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (8 * g + e >= nvalid) v[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
    }
    return __builtin_bit_cast(bf16x8, v);
}

// One launch per layer: grid (F/128 + 4, TS), 256 threads.
//   blockIdx.x < F/128 : linear1 / linear2.  A wave owns one 32-wide chunk of hidden units (two 16-wide tiles) and walks the
//                        32-token blocks of its split; the operands every wave needs (the x1 / d f rows of the block, one
//                        StageL record of 4 KS1 + 1 KiB) are staged once per workgroup in an LDS ring by global_load_lds;
//                        the feature-major operands of the d W products are transpose reads of the same rows.
//   the other four     : in_proj rows of q | k | v, and out_proj (operands straight from the T-blocks, next block's
//                        fragments prefetched into registers).
template <int KS1, int DT>
__global__ __launch_bounds__(256, 2) void k_tr_wgrad(const TrDims d, const WgLayer L, const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bx, ts;
    xcd_deal(d, bx, ts);                                      // (the role workgroups of a token split share an L2)
    const int D = d.D, F = d.F, M = d.M, NFT = d.NFT;
    const int blk0 = (int)(((long long)a.nblk * ts) / a.TS), blk1 = (int)(((long long)a.nblk * (ts + 1)) / a.TS);
    float* const part = a.part + (size_t)ts * a.nparams;
    constexpr int NB = 2 * KS1 + DT;
    if (bx < F / 128) {
        using SL = StageL<KS1, DT>;
        constexpr int SB = SL::bytes;                         // staged bytes per 32-token block (one StageL record)
        constexpr int NBUF = FD_TR_WG_NBUF, PD = NBUF - 1, NDMA = (SB / 1024 + 3) / 4;       // ring slots, blocks in flight
        const int NS = F / 64;
        const int chunk = bx * 4 + wave;              // hidden units 32 chunk .. +31
        const int fh = chunk / NS, c = chunk - fh * NS;
        bf16x8 w1[2][KS1], w2[2][KS1];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                w1[ft][ks] = *reinterpret_cast<const bf16x8*>(L.ffn_img + ((size_t)((c * 2 + fh) * NB + ft * KS1 + ks) * 64 + lane) * 16);
                w2[ft][ks] = *reinterpret_cast<const bf16x8*>(L.bffn + ((size_t)((c * 2 + fh) * NB + ft * KS1 + ks) * 64 + lane) * 16);
            }
        f32x4 a1[2][DT], a2[2][DT];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { a1[ft][dt] = f4zero(); a2[ft][dt] = f4zero(); }
        unsigned* const klut = reinterpret_cast<unsigned*>(smem + NBUF * SB);      // bf16 lane masks by nibble (see the block below)
        if (threadIdx.x < 32) {
            const unsigned n = threadIdx.x >> 1, hi = threadIdx.x & 1;
            klut[threadIdx.x] = ((n >> (2 * hi)) & 1u ? 0x0000ffffu : 0u) | ((n >> (2 * hi + 1)) & 1u ? 0xffff0000u : 0u);
        }
        // staging: every wave issues exactly NDMA 1-KiB copies + one 256-byte copy of activity words per block (uniform vmcnt
        // bookkeeping).  The activity words of the workgroup's 128 hidden units (two token halves x 128 words) ride the same
        // DMA path into a small LDS ring: as four 2-byte loads into registers per block they were 5 us of the 52 us launch, and
        // hipcc answered their pending registers with an s_waitcnt vmcnt(0) in front of every third barrier -- which drained
        // the staging DMA issued a block earlier.  No vector-memory instruction of the loop returns to a register now.
        char* const mring = smem + NBUF * SB + 128;                   // [NBUF][2 halves][128 words]
        auto issue = [&](int blk, int slot) {
            char* dst = smem + slot * SB;
            const char* src = L.stage + (size_t)blk * SB + lane * 16;
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                int bb = wave + 4 * i;
                bb %= SB / 1024;
                __builtin_amdgcn_global_load_lds(GLB_PTR(src + bb * 1024), LDS_PTR(dst + bb * 1024), 16, 0, 0);
            }
#ifndef FD_TR_ABL_WG_NOMASK
            {   // the block's activity BYTES of the workgroup's four 32-unit chunks: one dword per (token, lane group) = 512 bytes; waves
                // 0 / 2 copy the tokens 0-15, waves 1 / 3 the tokens 16-31 (the duplicates write the same bytes).  (Round 6: the forward used
                // to write a second, transposed copy -- a 16-token ballot per hidden unit -- for this kernel; the persistent forward's
                // tiles are not 16-aligned in the flat token index, and the bits can be picked out of the bytes here for ~20 VALU per block.)
                const int half = wave & 1;
                const unsigned char* msrc = L.active + (((size_t)blk * 32 + half * 16 + (lane >> 2)) * 4 + (lane & 3)) * (size_t)(F / 32) + (size_t)bx * 4;
                __builtin_amdgcn_global_load_lds(GLB_PTR(msrc), LDS_PTR(mring + slot * 512 + half * 256), 4, 0, 0);
            }
#endif
        };
        constexpr int NVM = NDMA + 1;                                 // vector-memory instructions per wave and block
        const int nb = blk1 - blk0;
        // the 16 workgroups of a (split, layer) read the same token blocks: each starts at its own block (fixed per workgroup,
        // so the summation order -- and the result -- is still reproducible)
        const int brot = nb > 0 ? (int)(((unsigned)bx * 3u) % (unsigned)nb) : 0;
        auto blk_of = [&](int ib) { int bq = ib + brot; bq -= (bq >= nb) ? nb : 0; return blk0 + bq; };
#pragma unroll
        for (int q = 0; q < PD; ++q)
            if (nb > q) issue(blk_of(q), q);
        // One block: the ring slot is a compile-time constant (the loop below is unrolled by NBUF), so neither the mask words
        // nor the LDS addresses go through run-time selects.  Tokens beyond M need no masking here: k_tr_ffn_fwd / k_tr_ffn_bwd
        // write ZERO rows and T-block columns for them into the stage records (every workgroup covers 64 tokens up to Mpad), and
        // a zero x1 / d f column contributes nothing to either gradient.  The dropout keep scale is linear in both products and
        // is applied once to the accumulators after the loop.
        auto block = [&](int ib, auto slot_c) {
            constexpr int slot = decltype(slot_c)::value;
            // block ib must have landed (this wave's share), then everybody's
            {
                const int after = min(PD - 1, nb - 1 - ib);                 // later blocks already issued (wave-uniform)
                if (after >= 3 && PD >= 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * NVM) : "memory");
                else if (after == 2 && PD >= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NVM) : "memory");
                else if (after == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NVM) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
#ifndef FD_TR_ABL_WG_NOBAR
            // bare s_barrier: __syncthreads() carries a workgroup fence that may be lowered to s_waitcnt vmcnt(0), which would drain
            // the DMA of the next block.  What must be ordered is ordered by hand: this wave's share of block ib has landed (above),
            // its LDS reads of block ib - 1 were consumed by that block's MFMAs, the table writes of the prologue have completed.
            __builtin_amdgcn_s_barrier();
#endif
            // this wave's activity words of the block: [half][tile] (token on tok & 3 / tok >> 2 as in the T-blocks)
            // this wave's activity nibbles [half][tile]: bit r = token 4 g + r of the half, hidden unit 16 tile + tok -- bit 4 tile + (tok & 3)
            // of the wave's byte of dword (token, lane group tok >> 2)
            unsigned mkw[4];
            {
                unsigned dwr[2][4];
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dwr[half][r] = *reinterpret_cast<const unsigned*>(mring + slot * 512 + half * 256 + ((4 * g + r) * 4 + (tok >> 2)) * 4);
                const unsigned sh = 8u * (unsigned)wave + (unsigned)(tok & 3);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned sq = sh + 4u * (unsigned)(q & 1);
                    mkw[q] = ((dwr[q >> 1][0] >> sq) & 1u) | (((dwr[q >> 1][1] >> sq) & 1u) << 1) | (((dwr[q >> 1][2] >> sq) & 1u) << 2) |
                             (((dwr[q >> 1][3] >> sq) & 1u) << 3);
                }
            }
            auto prefetch = [&]() {
#ifdef FD_TR_ABL_WG_NODMA
                if (false) {
#else
                if (ib + PD < nb) {
#endif
                    const int bn2 = blk_of(ib + PD);
                    issue(bn2, (slot + PD) % NBUF);           // (the slot of block ib - 1: every wave is past its reads)
                }
            };
#if FD_TR_WG_DMA_AT == 0
            prefetch();
#endif
            const char* sx = smem + slot * SB + SL::off_xr;
            const char* sd = smem + slot * SB + SL::off_dr;
            f32x4 hh[2][2], dh[2][2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 h[2] = {f4zero(), f4zero()}, e[2] = {f4zero(), f4zero()};
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    // A operands: rows = the 16 tokens of this half (row stride 64 KS1 bytes), 8 k-slots per lane group
                    const bf16x8 ax = *reinterpret_cast<const bf16x8*>(sx + (size_t)(half * 16 + tok) * (SL::RBS * 2) + (32 * ks + 8 * g) * 2);
                    const bf16x8 ad = *reinterpret_cast<const bf16x8*>(sd + (size_t)(half * 16 + tok) * (SL::RBS * 2) + (32 * ks + 8 * g) * 2);
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft) {
                        h[ft] = MFMA(ax, w1[ft][ks], h[ft]);      // [token rows 4g+r][hidden col]
                        e[ft] = MFMA(ad, w2[ft][ks], e[ft]);
                    }
                }
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    hh[ft][half] = h[ft];
                    dh[ft][half] = e[ft];
                }
            }
#if FD_TR_WG_DMA_AT == 1
            prefetch();          // (behind the H / d H MFMAs of the block)
#endif
            // inactive (dropped or h <= 0) units are cleared AFTER packing: the lane's nibble of the activity word selects two
            // dwords of bf16 lane masks from a 16-entry LDS table (one 8-byte read + four ANDs per (tile, half) instead of and +
            // compare + two selects per value: 64 of the ~100 VALU instructions of a block)
            bf16x8 hB[2], dB[2];
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                const u32x2 k0 = *reinterpret_cast<const u32x2*>(klut + 2 * mkw[ft]);
                const u32x2 k1 = *reinterpret_cast<const u32x2*>(klut + 2 * mkw[2 + ft]);
                const u32x4 ph = __builtin_bit_cast(u32x4, pack8(hh[ft][0], hh[ft][1])), pd = __builtin_bit_cast(u32x4, pack8(dh[ft][0], dh[ft][1]));
                hB[ft] = __builtin_bit_cast(bf16x8, u32x4{ph[0] & k0[0], ph[1] & k0[1], ph[2] & k1[0], ph[3] & k1[1]});
                dB[ft] = __builtin_bit_cast(bf16x8, u32x4{pd[0] & k0[0], pd[1] & k0[1], pd[2] & k1[0], pd[3] & k1[1]});
            }
#ifdef FD_TR_ABL_WG_NOT
            if (ib < 0)
#endif
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                // feature-major A operands (rows = features 16 dt + tok) in the token order of the packed C tiles: slots 0-3 =
                // tokens 4g.., slots 4-7 = tokens 16+4g...  ds_read_b64_tr_b16: lane 4j + q of a 16-lane group hands in the
                // address of features 16 dt + 4q.. of token row 4g + j, lane tok gets feature 16 dt + tok of the four rows.
                const int toff = (4 * g + (tok >> 2)) * (SL::RBS * 2) + (16 * dt + 4 * (tok & 3)) * 2;
                const s16x4 xl = lds_read_tr16(sx + toff), xh = lds_read_tr16(sx + toff + 16 * (SL::RBS * 2));
                const s16x4 dl = lds_read_tr16(sd + toff), dhh = lds_read_tr16(sd + toff + 16 * (SL::RBS * 2));
                const bf16x8 ax = __builtin_bit_cast(bf16x8, __builtin_shufflevector(xl, xh, 0, 1, 2, 3, 4, 5, 6, 7));
                const bf16x8 ad = __builtin_bit_cast(bf16x8, __builtin_shufflevector(dl, dhh, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    a2[ft][dt] = MFMA(ad, hB[ft], a2[ft][dt]);        // d W2[d][f]
                    a1[ft][dt] = MFMA(ax, dB[ft], a1[ft][dt]);        // d W1[f][d], row D = d b1[f]
                }
            }
#if FD_TR_WG_DMA_AT == 2
            prefetch();          // (behind every MFMA of the block)
#endif
        };
        static_assert(NBUF >= 3 && NBUF <= 5, "ring depth");
        for (int ib = 0; ib < nb; ib += NBUF) {
            block(ib, std::integral_constant<int, 0>{});
            if (ib + 1 < nb) block(ib + 1, std::integral_constant<int, 1>{});
            if (ib + 2 < nb) block(ib + 2, std::integral_constant<int, 2>{});
            if constexpr (NBUF > 3) { if (ib + 3 < nb) block(ib + 3, std::integral_constant<int, 3>{}); }
            if constexpr (NBUF > 4) { if (ib + 4 < nb) block(ib + 4, std::integral_constant<int, 4>{}); }
        }
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { a1[ft][dt][r] *= d.keep_scale; a2[ft][dt][r] *= d.keep_scale; }
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            const int f = chunk * 32 + ft * 16 + tok;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dd = 16 * dt + 4 * g + r;
                    if (dd < D) {
                        part[L.l2_w + (size_t)dd * F + f] = a2[ft][dt][r];
                        part[L.l1_w + (size_t)f * D + dd] = a1[ft][dt][r];
                    } else if (dd == D) {
                        part[L.l1_b + f] = a1[ft][dt][r];
                    }
                }
        }
    } else {
        // ------------------------------------------------ in_proj (+ bias through the ones row of x0T) and out_proj (+ bias)
        const int role = bx - F / 128;              // 0..2: q | k | v rows of in_proj, 3: out_proj
        const int NRT = (role < 3) ? d.NP : DT;             // 16-row tiles of this role
        const __bf16* At = (role < 3) ? L.dqkvT : L.doT;
        const int ANF = (role < 3) ? 3 * d.NP * 16 : NFT;
        const int rt_base = (role < 3) ? role * d.NP : 0;
        const __bf16* Bt = (role < 3) ? L.x0T : L.attT;
        constexpr int MAXR = 2;
        const int nr = (wave < NRT ? 1 : 0) + (wave + 4 < NRT ? 1 : 0);
        f32x4 acc[MAXR][DT];
#pragma unroll
        for (int i = 0; i < MAXR; ++i)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[i][dt] = f4zero();
        if (nr == 0 || blk1 <= blk0) {
            // (nothing owned: still nothing to write -- every output element has exactly one owner wave)
        } else {
            // operands straight from the T-blocks in global memory (L2 / Infinity Cache: ~2 us per round trip and only 2-10
            // MFMAs per block to hide it behind), so the fragments of the next PF blocks are in flight in registers
            constexpr int PF = 3;
            bf16x8 bn[PF][DT], an[PF][MAXR];
            auto fetch = [&](int blk, bf16x8 (&bq)[DT], bf16x8 (&aq)[MAXR]) {
                const int nvalid = min(32, M - blk * 32);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) bq[dt] = t_frag(Bt, blk, NFT, dt, lane, nvalid);
#pragma unroll
                for (int i = 0; i < MAXR; ++i)
                    if (i < nr) aq[i] = t_frag(At, blk, ANF, rt_base + wave + 4 * i, lane, nvalid);
            };
#pragma unroll
            for (int q = 0; q < PF; ++q)
                if (blk0 + q < blk1) fetch(blk0 + q, bn[q], an[q]);
            for (int blk = blk0; blk < blk1; blk += PF) {
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    if (blk + q < blk1) {
                        bf16x8 bc[DT], ac[MAXR];
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) bc[dt] = bn[q][dt];
#pragma unroll
                        for (int i = 0; i < MAXR; ++i) ac[i] = an[q][i];
                        if (blk + q + PF < blk1) fetch(blk + q + PF, bn[q], an[q]);
#pragma unroll
                        for (int i = 0; i < MAXR; ++i)
                            if (i < nr) {
#pragma unroll
                                for (int dt = 0; dt < DT; ++dt) acc[i][dt] = MFMA(ac[i], bc[dt], acc[i][dt]);
                            }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            if (i >= nr) continue;
            const int rt = wave + 4 * i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 4 * g + r;
                if (role < 3) {
                    const int hs = j >> 3, dd = j & 7, head = 2 * rt + hs;
                    if (head < d.H && dd < d.hd) {
                        const long long row = (long long)role * D + head * d.hd + dd;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            const int col = 16 * dt + tok;
                            if (col < D) part[L.in_w + row * D + col] = acc[i][dt][r];
                            else if (col == D) part[L.in_b + row] = acc[i][dt][r];
                        }
                    }
                } else {
                    const int dd = 16 * rt + j;
                    if (dd < D) {
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            const int col = 16 * dt + tok;
                            if (col < D) part[L.out_w + (size_t)dd * D + col] = acc[i][dt][r];
                            else if (col == D) part[L.out_b + dd] = acc[i][dt][r];
                        }
                    }
                }
            }
        }
    }
}

// grads[i] (+)= sum over the token splits (fixed order) for the matrices / biases the weight-gradient kernel owns and 0 for the
// alignment gaps of the flat layout (the fused AdamW and the gradient norm run over them).  ONE LAYER per launch, on the side
// stream right behind that layer's k_tr_wgrad (its partials are still in the Infinity Cache).  The five vector parameters of
// the FFN-side backward (per-workgroup column sums) are reduced by k_tr_vecreduce.
constexpr int kMaxTS = 32;
struct RedArgs {
    const float* part; long long nparams; int TS;      // nparams = parameters per layer (the stride of a split's partials)
    int D;
    long long rel[5];          // offsets of l2_b, n2_b, n2_w, n1_b, n1_w relative to the layer's first parameter
    long long wrel[7], wnum[7];   // in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w: offsets / element counts
    float* grads; int accumulate;                       // grads = the layer's first parameter
};
__global__ __launch_bounds__(256) void k_tr_reduce(const RedArgs a) {
    // 4 consecutive parameters per thread: every tensor of the flat layout starts on a 16-byte boundary and every owned range
    // has a multiple of 4 elements (d_model % 4 == 0), so a group is owned / vector / gap as a whole
    const long long rel = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (rel >= a.nparams) return;
#pragma unroll
    for (int sidx = 0; sidx < 5; ++sidx) {
        const long long o = rel - a.rel[sidx];
        if (o >= 0 && o < a.D) return;                 // k_tr_vecreduce owns these
    }
    float4 v = {0.f, 0.f, 0.f, 0.f};
    bool owned = false;
#pragma unroll
    for (int k = 0; k < 7; ++k) owned |= (rel >= a.wrel[k] && rel < a.wrel[k] + a.wnum[k]);
    if (owned) {
        // all splits' loads in flight together (a rolled loop waits for each load before issuing the next), added in order
        float4 p4[kMaxTS];
#pragma unroll
        for (int t = 0; t < kMaxTS; ++t)
            p4[t] = (t < a.TS) ? *reinterpret_cast<const float4*>(a.part + (size_t)t * a.nparams + rel) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < kMaxTS; ++t) { v.x += p4[t].x; v.y += p4[t].y; v.z += p4[t].z; v.w += p4[t].w; }
    }
    float4* gp = reinterpret_cast<float4*>(a.grads + rel);
    if (a.accumulate) {
        if (owned) { const float4 g4 = *gp; *gp = float4{g4.x + v.x, g4.y + v.y, g4.z + v.z, g4.w + v.w}; }
    } else {
        *gp = v;
    }
}

// The five vector parameters per layer whose gradients k_tr_ffn_bwd leaves as per-workgroup column sums (l2_b, n2_b, n2_w,
// n1_b, n1_w): grid (5, L), 1024 threads = 32 float4 columns x 32 strands.  A strand adds workgroups strand, strand + 32, ... in
// ascending order (all of its loads in flight), the strands are combined through LDS in ascending order: fixed order, no
// atomics.  (As one thread per column group inside k_tr_reduce this was a serial chain of nwg / 8 dependent round trips:
// 110 us at 252 workgroups.)
struct VecRedArgs {
    const float* vecpart;      // [L][nwg][5][D]
    int nwg, D;
    long long begin, layer_stride;
    long long rel[5];
    float* grads; int accumulate;
};
__global__ __launch_bounds__(1024) void k_tr_vecreduce(const VecRedArgs a) {
    __shared__ float4 red[32][32];
    const int vs = blockIdx.x, li = blockIdx.y;
    const int col = threadIdx.x & 31, strand = threadIdx.x >> 5;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    if (4 * col < a.D) {
        const float* vp = a.vecpart + ((size_t)li * a.nwg * 5 + vs) * a.D + 4 * col;
        for (int w0 = strand; w0 < a.nwg; w0 += 32 * 4) {
            float4 q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                q[k] = (w0 + 32 * k < a.nwg) ? *reinterpret_cast<const float4*>(vp + (size_t)(w0 + 32 * k) * 5 * a.D) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc.x += q[k].x; acc.y += q[k].y; acc.z += q[k].z; acc.w += q[k].w; }
        }
    }
    red[strand][col] = acc;
    __syncthreads();
    if (strand == 0 && 4 * col < a.D) {
        float4 v = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 32; ++k) { const float4 q = red[k][col]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        float4* gp = reinterpret_cast<float4*>(a.grads + a.begin + (long long)li * a.layer_stride + a.rel[vs] + 4 * col);
        if (a.accumulate) { const float4 g4 = *gp; v.x += g4.x; v.y += g4.y; v.z += g4.z; v.w += g4.w; }
        *gp = v;
    }
}

// out = a + sum of `np` <= 8 partial tensors (fixed order); 16-byte accesses, every partial's load in flight at once
__global__ __launch_bounds__(256) void k_tr_sum_parts(const float* __restrict__ a0, const float* __restrict__ parts, int np,
                                                       size_t stride, float* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        f32x4 v = *reinterpret_cast<const f32x4*>(a0 + i);
        f32x4 q[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int pp = p < np ? p : 0;
            q[p] = *reinterpret_cast<const f32x4*>(parts + (size_t)pp * stride + i);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p)
            if (p < np) v += q[p];
        for (int p = 8; p < np; ++p) v += *reinterpret_cast<const f32x4*>(parts + (size_t)p * stride + i);
        *reinterpret_cast<f32x4*>(out + i) = v;
    } else {
        for (size_t j = i; j < n; ++j) {
            float v = a0[j];
            for (int p = 0; p < np; ++p) v += parts[(size_t)p * stride + j];
            out[j] = v;
        }
    }
}

// the same with bf16 partial tensors (k_tr_attn_bwd OH form)
__global__ __launch_bounds__(256) void k_tr_sum_parts_bf16(const float* __restrict__ a0, const __bf16* __restrict__ parts, int np,
                                                            size_t stride, float* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        f32x4 v = *reinterpret_cast<const f32x4*>(a0 + i);
        for (int p0 = 0; p0 < np; p0 += 4) {
            u32x2 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const u32x2*>(parts + (size_t)(p0 + u < np ? p0 + u : np - 1) * stride + i);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (p0 + u < np)
                    v += f32x4{__builtin_bit_cast(float, q[u][0] << 16), __builtin_bit_cast(float, q[u][0] & 0xffff0000u),
                               __builtin_bit_cast(float, q[u][1] << 16), __builtin_bit_cast(float, q[u][1] & 0xffff0000u)};
        }
        *reinterpret_cast<f32x4*>(out + i) = v;
    } else {
        for (size_t j = i; j < n; ++j) {
            float v = a0[j];
            for (int p = 0; p < np; ++p) v += (float)parts[(size_t)p * stride + j];
            out[j] = v;
        }
    }
}

// Behind layer 0 (dxg form): gradient of the first layer's input = residual path + d(q | k | v) rows . in_proj^T (k_tr_ffn_bwd's prologue
// does the same for the layers above).  One token tile per wave; the image (NP x 3 x DT half blocks) goes through LDS once per workgroup
// (as fragments straight from the L2 every one of the NC x DT products waited for its own load: the kernel is on the step's tail).
template <int DT>
__global__ __launch_bounds__(256) void k_tr_dx0(const TrDims d, const float* __restrict__ dres, const __bf16* __restrict__ dqkvR,
                                                const char* __restrict__ winT, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = ((int)blockIdx.x * 4 + wave) * 16 + tok;
    const bool valid = m < d.M;
    const int NC = 3 * d.NP, NI = (NC + 1) / 2, img_bytes = NC * DT * 512;
    for (int i = wave; i * 1024 < img_bytes; i += 4) {
        const int off = i * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds(GLB_PTR(winT + (off < img_bytes ? off : 0)), LDS_PTR(smem + i * 1024), 16, 0, 0);
    }
    f32x4 dy[DT], o[DT];
    load_ctile<DT>(dres, m, valid, d.D, g, dy);
    constexpr int MAXI = 12;                                   // NP <= 8
    u32x4 bb[MAXI];
    {
        const char* rowp = reinterpret_cast<const char*>(dqkvR) + ((size_t)(valid ? m : 0) * 4 + g) * NC * 8;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) bb[i] = *reinterpret_cast<const u32x4*>(rowp + (size_t)(i < NI ? i : 0) * 16);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f4zero();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        if (i < NI) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = 2 * i + h;
                if (c < NC) {
                    const int wh = c / d.NP, pr = c - wh * d.NP;
                    const u32x2 bw = {valid ? bb[i][2 * h] : 0u, valid ? bb[i][2 * h + 1] : 0u};
                    const char* ab = smem + ((size_t)((pr * 3 + wh) * DT) * 64 + lane) * 8;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        o[dt] = MFMA16(*reinterpret_cast<const s16x4*>(ab + (size_t)dt * 512), __builtin_bit_cast(s16x4, bw), o[dt]);
                }
            }
        }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dy[dt] += o[dt];
    store_ctile<DT>(out, m, valid, d.D, g, dy);
}

}  // namespace

// ================================================================================================ host side
namespace {

struct TrLayerBufs {
    float *x0, *att, *s1, *s2, *lse2;
    __bf16 *x0rb, *x0T, *attT, *doT, *dqkvT;
    char* stage;
    unsigned char* active;
    unsigned char *pmask, *hkeep, *rb1, *rb3;      // the decision buffers of the step's set (tr_carve picks it)
};
struct TrBufs {
    std::vector<TrLayerBufs> layers;
    float *emb, *temb, *hL;
    // backward transients
    float *dh, *datt, *dres[2], *dxp[2], *dtemb, *skp, *vecpart, *part;
    int Mpad, nwg, TS;
    size_t part_stride, layer_params;
};

constexpr size_t kSkpFloats = (size_t)1 << 20;

// Token splits of the weight-gradient launch: (F/128 + 4) x TS workgroups of 4 waves.  The kernel is latency-bound (one
// L2 / Infinity-Cache round trip per 32-token block and wave), so it wants every SIMD of the chip to hold a wave: 16 splits
// give 320 workgroups at dim_ff 2048 (round 2 ran 4 = 80 workgroups on 256 CUs: 277 us per layer at 16 128 tokens).  At
// least 6 blocks per split so that the per-workgroup prologue (64 KiB of weight fragments) stays amortised.
int tr_TS(const fd_score* m, int B) {
    const long long nblk = ((long long)B * m->d.max_len + 31) / 32;
    // (20 splits = 400 workgroups since the chain kernels got faster: 2.67 -> 2.61 ms per step at 16 128 tokens against 16;
    // 18 / 22 / 24: 2.62 / 2.62 / 2.64)
    int ts = (int)std::min<long long>(20, std::max<long long>(1, nblk / 6));
    if (const char* e = getenv("FDIFF_TR_TS")) ts = std::max(1, std::min(kMaxTS, atoi(e)));     // experiments
    return ts;
}

size_t al(size_t b) { return fd_ws::padded(b); }

// one carve routine for size computation (base == nullptr) and pointer assignment
size_t tr_carve(const fd_score* m, int B, char* base, TrBufs* out, int mask_set = 0) {
    const fd_bf16_images* im = m->bf16;
    const size_t T = m->d.max_len, D = m->d.d_model, F = m->d.dim_ff, H = m->d.n_head, L = m->d.num_layers;
    const size_t M = (size_t)B * T, Mpad = (M + 63) & ~size_t(63);
    const size_t NFT = 16 * (size_t)im->dt, RBW = 32 * (size_t)im->ks1, NP = im->np, NJ = ((T + 15) / 16 + 1) / 2;
    const size_t stage_bytes = ((size_t)2 * 32 * (RBW + 8) * 2 + 1023) & ~size_t(1023);        // == StageL<KS1, DT>::bytes
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al(bytes); return p; };
    TrBufs tb;
    tb.Mpad = (int)Mpad;
    tb.nwg = (int)(Mpad / 64);
    tb.TS = tr_TS(m, B);
    tb.part_stride = M * D;
    tb.emb = (float*)take(sizeof(float) * B * D);
    tb.temb = (float*)take(sizeof(float) * B * D);
    tb.hL = (float*)take(sizeof(float) * M * D);
    tb.layers.resize(L);
    for (size_t l = 0; l < L; ++l) {
        TrLayerBufs& b = tb.layers[l];
        b.x0 = (float*)take(sizeof(float) * M * D);
        b.att = (float*)take(sizeof(float) * M * D + 16);      // (+16: k_tr_ffn_fwd reads whole 8-slot head groups)
        b.s1 = (float*)take(sizeof(float) * M * D);
        b.s2 = (float*)take(sizeof(float) * M * D);
        b.lse2 = (float*)take(sizeof(float) * B * H * T);
        b.x0rb = (__bf16*)take(2 * Mpad * RBW);
        b.stage = take((Mpad / 32) * stage_bytes + 1024);
        b.x0T = (__bf16*)take(2 * Mpad * NFT);
        b.attT = (__bf16*)take(2 * Mpad * NFT);
        b.doT = (__bf16*)take(2 * Mpad * NFT);
        b.dqkvT = (__bf16*)take(2 * Mpad * 3 * NP * 16);
        b.active = (unsigned char*)take(Mpad * (F / 32) * 4);
        // Two sets of dropout-decision buffers, used by alternating training steps: the decisions of step n + 1 are generated (on
        // their own stream) while step n still reads its own set -- beside step n's backward instead of in front of step n + 1's
        // forward, where the persistent forward (which leaves no register for a decision kernel beside it) had to wait for all of them
        for (int set = 0; set < 2; ++set) {
            unsigned char* pm = (unsigned char*)take((size_t)B * H * T * NJ * 4);
            unsigned char* hk = (unsigned char*)take(Mpad * (F / 32) * 4);
            unsigned char* r1 = (unsigned char*)take(Mpad * ((NFT / 16 + 1) / 2) * 4);
            unsigned char* r3 = (unsigned char*)take(Mpad * ((NFT / 16 + 1) / 2) * 4);
            if (set == mask_set) { b.pmask = pm; b.hkeep = hk; b.rb1 = r1; b.rb3 = r3; }
        }
    }
    tb.dh = (float*)take(sizeof(float) * M * D);
    tb.datt = (float*)take(sizeof(float) * M * D);
    for (int i = 0; i < 2; ++i) {
        tb.dres[i] = (float*)take(sizeof(float) * M * D);
        tb.dxp[i] = (float*)take(sizeof(float) * 2 * NP * M * D);     // per head pair, or per head (tr_attn_one_head)
    }
    tb.dtemb = (float*)take(sizeof(float) * B * D);
    tb.skp = (float*)take(sizeof(float) * kSkpFloats);
    tb.vecpart = (float*)take(sizeof(float) * L * tb.nwg * 5 * D);
    // per-split partials of ONE layer's weight gradients, one buffer per side stream (two layers in flight)
    tb.layer_params = L > 1 ? (size_t)(m->layers[1].in_w - m->layers[0].in_w) : (size_t)(m->nparams - (L ? m->layers[0].in_w : 0));
    tb.part = (float*)take(sizeof(float) * 2 * (size_t)kMaxTS * tb.layer_params);
    if (out) *out = tb;
    return off + 4096;
}

// Events that order the step's streams against each other.  (A device-scope release per record -- hipEventReleaseToDevice -- instead of
// the default system-scope fence was measured at +-0: 2.34 / 1.43 ms per step either way, scripts/archive/gpu_r04_events.sh.)
// Events of the training path order work between streams of ONE device (never inspected by the host): without the system-scope fence
// (FDIFF_TR_EVENT_FENCE=1 restores it).
static const unsigned kTrEventFlags = hipEventDisableTiming | ((getenv("FDIFF_TR_EVENT_FENCE") && atoi(getenv("FDIFF_TR_EVENT_FENCE")) != 0) ? 0u : (unsigned)hipEventDisableSystemFence);
// FDIFF_TR_LEAN_EVENTS (bits; default 26 = all three): 2 = the forward's "readers done" event is the persistent launch's stop event,
// 8 = one join of the side streams in front of the optimizer, 16 = the image rebuild waits for fd_score_prepare's own stop event.
// (Measured and dropped, profiles/r06_train_event_packets_ab.txt: bit 1 = `s` waits once for decisions + weight images through the
// decision stream, +30 us; bit 4 = the backward's "readers done" event as layer 0's k_tr_attn_bwd's stop event, +8 / +18 us: the record
// behind that kernel gives layer 0's weight-gradient launch a head start over the embedding backward.)
static int tr_lean_bits() {
    static const int v = getenv("FDIFF_TR_LEAN_EVENTS") ? atoi(getenv("FDIFF_TR_LEAN_EVENTS")) : 26;
    return v;
}

// Side streams carry work that is OFF the step's critical path (dropout decisions one layer ahead, weight gradients behind the
// input-gradient chain): lowest priority, so that the dispatcher hands free CU slots to the chain's workgroups first.
hipError_t side_stream_create(hipStream_t* st) {
    int least = 0, greatest = 0;
    if (!getenv("FDIFF_TR_NOPRIO") && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
        return hipStreamCreateWithPriority(st, hipStreamNonBlocking, least);
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

int tr_attn_waves(int KT) {
    if (const char* e = getenv("FDIFF_TR_ATTN_NW")) return atoi(e) == 8 ? 8 : 4;     // experiments
    return KT > 4 ? 8 : 4;
}

// Attention backward per (head, series) instead of per (head pair, series): see k_tr_attn_bwd.  FDIFF_TR_ATTN_OH=0 keeps the
// pair form (A/B measurements).
// 0: pair form; 1: one head per workgroup, fp32 parts; 2: one head per workgroup, bf16 parts.  Measured at B = 64 (same box,
// alternating runs, profiles/r04_train_attn_bwd_forms.txt): T = 252 (16 tiles) 2.65 / 2.63 / 2.61 ms per optimizer step, T = 100
// (7 tiles) 1.49 / 1.55 / 1.53 -- with few tiles the duplicated staging of the one-head form outweighs its even rounds.
int tr_attn_oh_mode(int KT) {
    const char* e = getenv("FDIFF_TR_ATTN_OH");               // (read per call: the tests switch forms inside one process)
    const int v = e ? atoi(e) : -1;
    return v >= 0 ? v : (KT >= 12 ? 2 : 0);
}

// dynamic LDS of k_tr_attn_bwd: q / k / v / dO row forms, q / k / dO column forms, -lse and -rowsum(dO . O), the keep bits in
// both orientations and the keep-multiplier table
size_t tr_attn_bwd_lds(int T, bool one_head) {
    const size_t KT = (size_t)(T + 15) / 16, NJ = (KT + 1) / 2, nhs = one_head ? 1 : 2;
    return 4 * KT * 16 * (one_head ? 16 : 32) + 3 * NJ * (one_head ? 512 : 1024) + 2 * nhs * KT * 16 * sizeof(float) +
           nhs * (size_t)T * NJ * 4 + 16 + 256 + nhs * NJ * (KT * 16 + (KT * 16) / 32 + 1) * 4;
}

// the context's error word of the device-side bounded waits (pinned, mapped into the device)
int tr_err_word(fd_ctx* ctx) {
    if (!ctx->tr_err_host) {
        FD_HIP(ctx, hipHostMalloc((void**)&ctx->tr_err_host, 64, hipHostMallocMapped));
        *ctx->tr_err_host = 0u;
        FD_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->tr_err_dev, ctx->tr_err_host, 0));
        FD_HIP(ctx, hipMalloc((void**)&ctx->tr_err_gpu, 64));
        FD_HIP(ctx, hipMemset(ctx->tr_err_gpu, 0, 64));
    }
    return FD_OK;
}
unsigned long long tr_wait_timeout_ticks() {      // bound of a device-side wait in ticks of the constant 100 MHz clock
    const char* e = getenv("FDIFF_TR_TIMEOUT_MS");           // (read per call: the tests set it inside one process)
    if (!e) e = getenv("FDIFF_TR_FSPLIT_TIMEOUT_MS");
    const double ms = e ? std::max(1.0, atof(e)) : 2000.0;
    return (unsigned long long)(ms * 1.0e5);
}

// buffers + a fresh epoch for one F-split launch (fs stays empty when the launch is not split).
// Constraints of the split, stated here because nothing in the signature does: (1) the partial-sum buffer, the flags, the epoch
// counter and the error word belong to the CONTEXT -- one training step at a time per context, on one stream (the library's
// general rule: a context is not thread-safe), two models may alternate on a context but not overlap; (2) the epoch is a kernel
// argument, so a training step must not be stream-captured and replayed (a replay would meet flags that already equal its
// epoch and add stale partial sums; the Philox offsets of the dropout decisions change per step as well); (3) the buffers
// are sized once for CUs / 2 token blocks -- the rule below never splits more -- so the steady state neither allocates nor
// synchronises.
int tr_fsplit_prepare(fd_ctx* ctx, const TrDims& d, int blocks, int DT, FSplit* fs, hipStream_t s) {
    *fs = FSplit{};
    if (d.fsplit != 2) return FD_OK;
    if (int rc = tr_err_word(ctx)) return rc;
    if ((size_t)blocks > ctx->tr_fsplit_blocks) {
        // (first use on this context; or a device whose CU count changed under us: the old buffers may still be read by a launch in flight)
        FD_HIP(ctx, hipStreamSynchronize(s));
        if (ctx->tr_ypart) (void)hipFree(ctx->tr_ypart);
        if (ctx->tr_yflag) (void)hipFree(ctx->tr_yflag);
        ctx->tr_ypart = nullptr; ctx->tr_yflag = nullptr; ctx->tr_fsplit_blocks = 0;
        const size_t nb = (size_t)std::max(blocks, ctx->num_cu / 2);
        FD_HIP(ctx, hipMalloc((void**)&ctx->tr_ypart, nb * 4 * 9 * 64 * sizeof(float) * 4));      // (DT <= 9 in every class)
        FD_HIP(ctx, hipMalloc((void**)&ctx->tr_yflag, nb * 4 * sizeof(unsigned)));
        FD_HIP(ctx, hipMemsetAsync(ctx->tr_yflag, 0, nb * 4 * sizeof(unsigned), s));
        ctx->tr_fsplit_blocks = nb;
    }
    (void)DT;
    fs->ypart = ctx->tr_ypart; fs->flag = ctx->tr_yflag; fs->epoch = ++ctx->tr_epoch;
    if (fs->epoch == 0u) fs->epoch = ++ctx->tr_epoch;      // (0 is the flags' initial value)
    fs->err = ctx->tr_err_dev; fs->err_gpu = ctx->tr_err_gpu;
    fs->timeout = tr_wait_timeout_ticks();
    const char* e = getenv("FDIFF_TR_FSPLIT_FENCE");
    fs->fence = (e && atoi(e) != 0) ? 1 : 0;
    e = getenv("FDIFF_TR_FSPLIT_TEST_STALL");
    fs->stall = (e && atoi(e) != 0) ? 1 : 0;
    return FD_OK;
}

// F-split of the FFN kernels (struct FSplit): every workgroup of the doubled grid must find its own CU (the finisher of a pair spins
// until its producer has handed over), and the halves must keep whole two-step trips.  The BACKWARD kernel splits only while the
// doubled grid leaves half of the CUs free: the weight-gradient launches run beside it on the CUs it does not use, and at 100 blocks
// (T = 100, B = 64) the faster FFN kernels made the step slower (1.43 -> 1.45 ms, profiles/r04_train_fsplit_ab.txt).  The FORWARD
// kernel has only the register-light decision kernels beside it and splits up to CUs / 2 blocks.
// FDIFF_TR_FSPLIT: 0 never; 1 (default) forward <= CUs / 2, backward <= CUs / 4; 2 both <= CUs / 2; 3 both <= CUs / 4.
int tr_fsplit_rule(const fd_score* m, int M, int F, bool forward) {
    const char* e = getenv("FDIFF_TR_FSPLIT");
    const int mode = e ? atoi(e) : 1;
    const long long blocks = ((long long)M + 63) / 64;
    const int cu = m->ctx->num_cu;
    const long long cap = mode == 2 ? cu / 2 : mode == 3 ? cu / 4 : (forward ? cu / 2 : cu / 4);
    return (mode != 0 && blocks <= cap && (F / 64) % 4 == 0) ? 2 : 1;
}

TrDims make_dims(const fd_score* m, int B, float p, uint64_t seed) {
    const fd_bf16_images* im = m->bf16;
    TrDims d{};
    d.B = B; d.T = m->d.max_len; d.M = B * d.T; d.D = m->d.d_model; d.F = m->d.dim_ff; d.H = m->d.n_head; d.hd = d.D / d.H;
    d.NP = im->np; d.KT = (d.T + 15) / 16; d.NJ = (d.KT + 1) / 2; d.NFT = 16 * im->dt; d.RBW = 32 * im->ks1;
    d.p = p;
    d.thr16 = (unsigned)std::lround((double)p * 65536.0);
    if (p > 0.f && d.thr16 == 0) d.thr16 = 1;
    d.keep_scale = (p > 0.f) ? (float)(65536.0 / (65536.0 - (double)d.thr16)) : 1.0f;
    d.seed = seed;
    static const int xcd_env = getenv("FDIFF_TR_XCD") ? atoi(getenv("FDIFF_TR_XCD")) : 1;      // (0: hardware order, A/B runs)
    d.xcd = xcd_env;
    d.fsplit = tr_fsplit_rule(m, d.M, d.F, false);      // (the forward launch applies its own rule, see there)
    { const char* e = getenv("FDIFF_TR_ROT"); d.norot = (e && atoi(e) == 0) ? 1 : 0; }      // (read per call: a test switches it)
    return d;
}

// ---- the loss head: unembedder forward + DSM loss + unembedder backward for the tokens [tiles ts, ts + TS, ...] of series b ----
// score = hL Wu^T + bu; d = score + target; loss element = coef d^2 with coef = 1 / sum_k std_k^-2 (default) or std_t^2
// (likelihood weighting) -- losses.py:92-124 as in k_dsm_loss; dscore = 2 coef d / (B T C) * grad_weight;
// dh = dscore Wu; per-workgroup partials of dWu = dscore^T hL, dbu = sum_t dscore and of the loss.  fp32 on the VALU: the
// whole head is 6 C D FLOP per token on (B*T, C <= 40) data.
struct HeadArgs {
    const float *hL, *Wu, *bu, *target, *stdv;
    float *dh, *part;      // part[(b * TS + ts)][C*D (dWu, row-major like the parameter) | C (dbu) | 1 (loss)]
    int T, C, D, TS, lw;
    float inv_cnt, gw;
};
// One 32-token tile per loop trip (the host picks TS = number of tiles whenever the partials fit, so a workgroup normally runs
// one trip): every global read of the trip -- the tile's hL rows, its targets, the std row, Wu, bu -- is issued before the
// first use, i.e. ONE memory round trip, three barriers, then the stores.  (The first version walked its tiles with a
// load -> barrier -> compute chain per phase: 26 us for 64 x 4 workgroups.)
template <int KMAX>
__global__ __launch_bounds__(256) void k_tr_head(HeadArgs A) {
    extern __shared__ float head_lds[];
    constexpr int HMAX = 10;                   // 32 * D / 256 register slots for the tile's rows: d_model <= 80 (host check)
    const int T = A.T, C = A.C, D = A.D, D1 = D + 1;
    float* Wu_s = head_lds;                    // [C][D + 1]
    float* hs = Wu_s + C * D1;                 // [32][D + 1]
    float* ds = hs + 32 * D1;                  // [32][C]
    float* red = ds + 32 * C;                  // [4] + w
    const int tid = threadIdx.x, ts = blockIdx.x, b = blockIdx.y;
    const float* sd = A.stdv + (size_t)b * T;
    const int ntile = (T + 31) / 32;
    // --- everything the first tile needs, in flight together
    float wreg[KMAX], hreg[HMAX], sreg[4];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { const int i = tid + 256 * k; wreg[k] = i < C * D ? A.Wu[i] : 0.f; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = tid + 256 * k; sreg[k] = i < T ? sd[i] : 0.f; }
    int tile = ts;
    {
        const size_t row0 = (size_t)b * T + tile * 32;
        const int nrow = min(32, T - tile * 32);
#pragma unroll
        for (int k = 0; k < HMAX; ++k) { const int i = tid + 256 * k; hreg[k] = i < nrow * D ? A.hL[row0 * D + i] : 0.f; }
    }
    float acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
    float bacc = 0.f, loss = 0.f;
    {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid + 256 * k < T) a += 1.0f / (sreg[k] * sreg[k]);
        for (int k = tid + 1024; k < T; k += 256) a += 1.0f / (sd[k] * sd[k]);
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
        if ((tid & 63) == 0) red[tid >> 6] = a;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { const int i = tid + 256 * k; if (i < C * D) Wu_s[(i / D) * D1 + (i % D)] = wreg[k]; }
    }
    for (; tile < ntile; tile += A.TS) {
        const int t0 = tile * 32, nrow = min(32, T - t0);
        const size_t row0 = (size_t)b * T + t0;
        // the tile's targets and bias for this thread's (row, channel) items: issued here, consumed after the barrier
        float treg[KMAX], breg[KMAX];           // 32 * C / 256 <= KMAX items (C * D <= 256 * KMAX, D >= 8)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = tid + 256 * k;
            const bool ok = i < nrow * C;
            treg[k] = ok ? A.target[row0 * C + i] : 0.f;
            breg[k] = ok ? A.bu[i % C] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < HMAX; ++k) { const int i = tid + 256 * k; if (i < 32 * D) hs[(i / D) * D1 + (i % D)] = hreg[k]; }
        __syncthreads();
        const float w = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = tid + 256 * k;
            if (i < 32 * C) {
                const int r = i / C, c = i - r * C;
                float dout = 0.f;
                if (r < nrow) {
                    float d0 = breg[k], d1 = 0.f;
                    const float* hr = hs + r * D1;
                    const float* wr = Wu_s + c * D1;
                    int d = 0;
                    for (; d + 1 < D; d += 2) { d0 = fmaf(hr[d], wr[d], d0); d1 = fmaf(hr[d + 1], wr[d + 1], d1); }
                    if (d < D) d0 = fmaf(hr[d], wr[d], d0);
                    const float dd = (d0 + d1) + treg[k];
                    const float sdt = sd[t0 + r];
                    const float coef = A.lw ? sdt * sdt : w;
                    loss = fmaf(coef * dd, dd, loss);
                    dout = 2.0f * coef * dd * A.inv_cnt * A.gw;
                }
                ds[i] = dout;
            }
        }
        __syncthreads();
        // next trip's rows (rare: only when the partials of one tile per workgroup do not fit the scratch)
        if (tile + A.TS < ntile) {
            const size_t rown = (size_t)b * T + (tile + A.TS) * 32;
            const int nn = min(32, T - (tile + A.TS) * 32);
#pragma unroll
            for (int k = 0; k < HMAX; ++k) { const int i = tid + 256 * k; hreg[k] = i < nn * D ? A.hL[rown * D + i] : 0.f; }
        }
#pragma unroll
        for (int k = 0; k < HMAX; ++k) {
            const int i = tid + 256 * k;
            if (i < nrow * D) {
                const int r = i / D, d = i - r * D;
                float v = 0.f;
                for (int c = 0; c < C; ++c) v = fmaf(ds[r * C + c], Wu_s[c * D1 + d], v);
                A.dh[row0 * D + i] = v;
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int o = tid + 256 * k;
            if (o < C * D) {
                const int c = o / D, d = o - c * D;
                float v0 = acc[k], v1 = 0.f;
#pragma unroll 8
                for (int r = 0; r < 32; r += 2) {
                    v0 = fmaf(ds[r * C + c], hs[r * D1 + d], v0);
                    v1 = fmaf(ds[(r + 1) * C + c], hs[(r + 1) * D1 + d], v1);
                }
                acc[k] = v0 + v1;
            }
        }
        if (tid < C)
            for (int r = 0; r < 32; ++r) bacc += ds[r * C + tid];
        __syncthreads();
    }
    float* out = A.part + ((size_t)b * A.TS + ts) * ((size_t)C * D + C + 1);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int o = tid + 256 * k;
        if (o < C * D) out[o] = acc[k];
    }
    if (tid < C) out[C * D + tid] = bacc;
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_down(loss, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = loss;
    __syncthreads();
    if (tid == 0) out[C * D + C] = ((red[0] + red[1]) + (red[2] + red[3])) * A.inv_cnt;
}
// dWu += sum_r part[r][0 .. CD), dbu += sum_r part[r][CD .. CD + C), loss = sum_r part[r][CD + C]: 16 outputs x 16 strands of
// r per block (a strand is R / 16 rows, eight loads in flight), combined in a fixed order
__global__ __launch_bounds__(256) void k_tr_head_final(const float* __restrict__ part, int R, int CD, int C, float* __restrict__ dWu,
                                                        float* __restrict__ dbu, float* __restrict__ loss_out) {
    __shared__ float red[16][17];
    const int n = CD + C + 1;
    const int col = threadIdx.x & 15, sub = threadIdx.x >> 4;
    const int id = blockIdx.x * 16 + col;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    if (id < n) {
        int r = sub;
        for (; r + 16 * 7 < R; r += 16 * 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += part[(size_t)(r + 16 * k) * n + id];
        }
        for (; r < R; r += 16) a[0] += part[(size_t)r * n + id];
    }
    red[sub][col] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (sub != 0 || id >= n) return;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][col];
    if (id < CD) dWu[id] += v;
    else if (id < CD + C) dbu[id - CD] += v;
    else *loss_out = v;
}

template <int KS1, int DT, int KSO>
int tr_forward_t(fd_score* m, const float* x, const float* t, float* out, int B, float p, uint64_t seed, uint64_t offset,
                 hipStream_t s, TrBufs& tb, uint64_t gen_in, bool img_forked, int mask_set) {
    fd_ctx* ctx = m->ctx;
    const fd_bf16_images* im = m->bf16;
    const int T = m->d.max_len, C = m->d.n_channels, D = m->d.d_model, L = m->d.num_layers;
    const int M = B * T;
    const float* P = m->params;
    const TrDims d = make_dims(m, B, p, seed);
    if (fd_time_embed_train(t, P + m->tW, P + m->td_w, P + m->td_b, tb.emb, tb.temb, B, D, s)) return FD_ERR_HIP;
    float* h0 = L > 0 ? tb.layers[0].x0 : tb.hL;
    fdf32::embed(x, P + m->emb_w, P + m->emb_b, P + m->pos, tb.temb, h0, M, T, C, D, s);
    if (L > 0)
        hipLaunchKernelGGL((k_tr_prep<KS1, DT>), dim3((tb.Mpad / 16 + 3) / 4), dim3(256), 0, s, h0, tb.layers[0].x0rb, tb.layers[0].x0T,
                           M, tb.Mpad, D);
    // Every event packet on `s` between two kernels of the chain is 4-5 us of an idle chip (round 6: 2.139 -> 2.094 ms per step at
    // T = 252 for the ten records behind k_tr_attn_bwd alone, profiles/r06_train_event_packets_ab.txt).  FDIFF_TR_LEAN_EVENTS=0 keeps
    // the round-5 form: a wait for the rebuilt weight images here, one for the dropout decisions in front of the first layer, a
    // record behind the forward, two joins in front of the optimizer.
    const int lean = tr_lean_bits();
    // ---- every encoder layer as ONE persistent launch (fd_train_persist.hip) where it applies: T <= 256, the workgroups of a launch
    // all resident.  FDIFF_TR_PERSIST: 0 = the per-layer kernels below, 1 (default) = persistent, 2 = the same kernel launched once
    // per layer (A/B runs and debugging: no inter-workgroup wait is ever exercised).  Read per call: the tests compare the forms.
    int trp_nq = 0, trp_spl = 0;
    const int trp_mode = [] { const char* e = getenv("FDIFF_TR_PERSIST"); return e ? atoi(e) : 1; }();
    const int trp_nt = (trp_mode != 0 && L > 0 && !ctx->trp_disabled) ? fd_trp_tiles(m, B, &trp_nq, &trp_spl) : 0;
    // weight images rebuilt (side stream 2).  (Letting the decision stream wait for them, so that `s` waits once for that stream, made
    // the step 30 us SLOWER, 2.09 -> 2.12 ms: profiles/r06_train_event_packets_ab.txt, bit 1.)
    if (img_forked) FD_HIP(ctx, hipStreamWaitEvent(s, m->img_event, 0));
    const size_t lds_attn = (size_t)d.KT * 16 * 32 + (size_t)d.NJ * 1024 + 128;      // K rows, V^T blocks, keep-mask table
    const int attn_nw = tr_attn_waves(d.KT);
    const size_t scr = std::max((size_t)TW * KS1 * 1024, (size_t)4 * DT * 1024);
    const size_t NSh = (size_t)m->d.dim_ff / 64;
    const size_t lds_ffn = (size_t)4 * 2 * (2 * KS1 + DT) * 1024 + scr + (size_t)TW * 64 * NSh + (size_t)TW * NSh * 32 * sizeof(unsigned short) +
                           (size_t)4 * 32 * DT * 16 + 128 + (size_t)3 * 4 * DT * 16;      // (+ the keep-mask table, + the epilogue's vectors)
    static unsigned long long attr = 0;
    if (fd_first_on_device(attr, ctx->device)) {
        FD_HIP(ctx, hipFuncSetAttribute((const void*)k_tr_attn_fwd<KS1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FD_HIP(ctx, hipFuncSetAttribute((const void*)k_tr_attn_fwd<KS1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FD_HIP(ctx, hipFuncSetAttribute((const void*)k_tr_ffn_fwd<KS1, DT, KSO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (p > 0.f) {
        // dropout decisions of every layer on the side stream, layer by layer, ahead of the kernels that read them
        if (!ctx->side_stream) FD_HIP(ctx, side_stream_create(&ctx->side_stream));
        // (round 6: the decision kernels have a stream of their own.  On side stream 1 they queued behind the previous step's last
        // weight-gradient launch -- the very end of that step -- although all they wait for is the last READER of the decision buffers,
        // the attention backward of layer 0: now they run beside that step's tail (layer 0's weight gradients, the embedding
        // backward, AdamW) and this step's prologue.  FDIFF_TR_MASK_STREAM=0: side stream 1 as before.)
        static const bool own_mask_stream = !(getenv("FDIFF_TR_MASK_STREAM") && atoi(getenv("FDIFF_TR_MASK_STREAM")) == 0);
        if (own_mask_stream && !ctx->mask_stream) FD_HIP(ctx, side_stream_create(&ctx->mask_stream));
        const hipStream_t mstream = own_mask_stream ? ctx->mask_stream : ctx->side_stream;
        while ((int)ctx->side_events.size() < L + 3) {
            hipEvent_t e;
            FD_HIP(ctx, hipEventCreateWithFlags(&e, kTrEventFlags));
            ctx->side_events.push_back(e);
        }
        // The mask buffers were last read by the previous training forward / backward on `s`: tr_readers_event was recorded
        // behind that reader, so the decisions of this step are generated while `s` still runs this step's prologue kernels
        // (perturbation, weight-image rebuild, embedding).  Recording an event HERE makes the first attention kernel wait for
        // a cross-stream hand-off plus the first mask kernel (an 80 us hole in every step), so that is only done when some
        // other call has carved the arena since (an eval forward, the sampler, another model on this context: gen_in differs)
        // and may still be running on `s`.
        // (round 6: two sets of decision buffers; this step's set was last read two training steps ago)
        (void)gen_in;
        hipEvent_t& rev = ctx->tr_readers_event[mask_set];
        if (!rev) FD_HIP(ctx, hipEventCreateWithFlags(&rev, kTrEventFlags));
        if (!ctx->tr_readers_event_valid[mask_set]) FD_HIP(ctx, hipEventRecord(rev, s));
        FD_HIP(ctx, hipStreamWaitEvent(mstream, rev, 0));
        for (int l = 0; l < L; ++l) {
            TrLayerBufs& b = tb.layers[l];
            MaskArgs ma{};
            ma.hkeep = b.hkeep; ma.pmask = b.pmask; ma.rb1 = b.rb1; ma.rb3 = b.rb3;
            ma.off0 = fd_dropout_site_offset(offset, l, 0); ma.off1 = fd_dropout_site_offset(offset, l, 1);
            ma.off2 = fd_dropout_site_offset(offset, l, 2); ma.off3 = fd_dropout_site_offset(offset, l, 3);
            ma.NS2 = m->d.dim_ff / 32;
            ma.n_h = (long long)tb.Mpad * 4 * ma.NS2;
            ma.n_p = (long long)B * m->d.n_head * T * d.NJ * 4;
            ma.n_r = (long long)tb.Mpad * ((DT + 1) / 2) * 4;
            const long long tot = ma.n_h + ma.n_p + 2 * ma.n_r;
            // Workgroups per CU of the decision kernel (persistent workgroups, VALU-bound, 66 VGPRs).  16 per CU hold most registers of
            // the chip while a decision kernel runs, and the forward chain's kernels beside them take 36-50 us instead of 29-41
            // (rocprofv3 time line) -- but fewer workgroups make the decisions later and the step slower: 16 / 8 / 4 / 2 / 1 per CU
            // = 2.36 / 2.36 / 2.39 / 2.40 / 2.40 ms per step at T = 252 (scripts/archive/gpu_r04_maskwgs.sh).
            static const int mask_wgs = getenv("FDIFF_TR_MASK_WGS") ? std::max(1, atoi(getenv("FDIFF_TR_MASK_WGS"))) : 16;
            const unsigned grid = (unsigned)std::min<long long>((tot / 2 + 255) / 256, (long long)ctx->num_cu * mask_wgs);
            hipLaunchKernelGGL(k_tr_masks, dim3(grid), dim3(256), 0, mstream, d, ma);
            FD_HIP(ctx, hipEventRecord(ctx->side_events[l], mstream));
        }
    }
    hipEvent_t readers_done = nullptr;      // set when a launch carries the decision buffers' "readers done" event as its stop event
    if (trp_nt) {
        if (int rc = tr_err_word(ctx)) return rc;
        const size_t nflag = (size_t)B * d.KT;
        if (nflag > ctx->trp_flag_count) {
            FD_HIP(ctx, hipStreamSynchronize(s));              // (first use, or a larger batch: a launch in flight may still poll the old flags)
            if (ctx->trp_flags) (void)hipFree(ctx->trp_flags);
            ctx->trp_flags = nullptr; ctx->trp_flag_count = 0;
            FD_HIP(ctx, hipMalloc((void**)&ctx->trp_flags, nflag * sizeof(unsigned long long)));
            FD_HIP(ctx, hipMemsetAsync(ctx->trp_flags, 0, nflag * sizeof(unsigned long long), s));
            ctx->trp_flag_count = nflag;
        }
        // The dropout decisions of EVERY layer must be written before the launch: its workgroups hold every register of every CU
        // (8 waves x 256 VGPRs), so a decision kernel on the side stream could not become resident beside them, and a wait inside the
        // kernel for decisions that cannot be produced would only end at its timeout.  They were launched above, ahead of the step's
        // prologue kernels on `s` (time embedding, embedding, operand preparation, the weight-image rebuild beside them).
        if (p > 0.f) FD_HIP(ctx, hipStreamWaitEvent(s, ctx->side_events[L - 1], 0));
        const TrLayerBufs& b0 = tb.layers[0];
        const fd_layer_off& l0 = m->layers[0];
        fd_trp_args ta{};
        ta.x0 = b0.x0; ta.x0rb = b0.x0rb; ta.x0T = b0.x0T; ta.att = b0.att; ta.attT = b0.attT; ta.lse2 = b0.lse2;
        ta.s1 = b0.s1; ta.s2 = b0.s2; ta.stage = b0.stage; ta.active = b0.active;
        ta.pmask = b0.pmask; ta.hkeep = b0.hkeep; ta.rb1 = b0.rb1; ta.rb3 = b0.rb3;
        ta.lstride = L > 1 ? (size_t)((const char*)tb.layers[1].x0 - (const char*)b0.x0) : 0;
        ta.hL = tb.hL;
        ta.limg = im->mimg + im->off_layers; ta.limg_stride = im->layer_stride;
        ta.off_wk = im->off_wk; ta.off_wv = im->off_wv; ta.off_wq = im->off_wq; ta.off_wo = im->off_wo; ta.off_ffn = im->off_ffn;
        ta.P = P; ta.pstride = (long long)tb.layer_params;
        ta.o_bo = l0.out_b; ta.o_g1 = l0.n1_w; ta.o_be1 = l0.n1_b; ta.o_b2 = l0.l2_b; ta.o_g2 = l0.n2_w; ta.o_be2 = l0.n2_b;
        ta.L = L; ta.Mpad = tb.Mpad;
        ta.xflag = ctx->trp_flags; ta.mflag = nullptr;
        ta.err = ctx->tr_err_dev; ta.err_gpu = ctx->tr_err_gpu; ta.timeout = tr_wait_timeout_ticks();
        { const char* e = getenv("FDIFF_TR_PERSIST_TEST_STALL"); ta.stall = (e && atoi(e) != 0) ? 1 : 0; }
        {
            // measurement hook: every matrix product of the L layers (projections, scores, P V, out-projection, FFN) of M tokens
            fd_prof_scope scope(ctx, s, "k_tr_fwd_layers (all encoder layers, training forward, one persistent launch)",
                                (double)L * M * (8.0 * D * D + 4.0 * (double)T * D + 4.0 * D * m->d.dim_ff));
            if (trp_mode == 2) {
                for (int l = 0; l < L; ++l) {
                    ta.l0 = l; ta.l1 = l + 1; ta.epoch = ++ctx->trp_epoch;
                    if (int rc = fd_trp_forward(m, d, ta, trp_nt, trp_nq, trp_spl, s, nullptr)) return rc;
                }
            } else {
                ta.l0 = 0; ta.l1 = L; ta.epoch = ++ctx->trp_epoch;
                // the launch is the forward's last reader of the decision buffers: their "readers done" event is its stop event
                if ((lean & 2) && p > 0.f) readers_done = ctx->tr_readers_event[mask_set];
                if (int rc = fd_trp_forward(m, d, ta, trp_nt, trp_nq, trp_spl, s, readers_done)) return rc;
            }
        }
    }
    int mask_waited = -1;                 // highest layer whose dropout decisions `s` has waited for
    for (int l = 0; l < (trp_nt ? 0 : L); ++l) {
        const fd_layer_off& lo = m->layers[l];
        TrLayerBufs& b = tb.layers[l];
        const char* limg = im->mimg + im->off_layers + (size_t)l * im->layer_stride;
        AttnFwdArgs aa{};
        aa.x0rb = b.x0rb; aa.att = b.att; aa.attT = b.attT; aa.lse2 = b.lse2; aa.pmask = b.pmask;
        aa.wk = limg + im->off_wk; aa.wv = limg + im->off_wv; aa.wq = limg + im->off_wq;
        // This layer's dropout decisions must be ready.  The decision kernels run ahead of the chain (rocprofv3 time line at T = 252:
        // layer l's at 32 + 40 l us, the chain reaches layer l at 120 + 76 l us), and every wait packet between two chain kernels
        // costs ~6 us of idle queue: wait for layer min(L - 1, 2 l + 1)'s event and skip the waits that one covers (layers 0, 2, 6
        // of 10 wait; FDIFF_TR_MASK_WAIT_ALL=1 waits in front of every layer, A/B runs)
        if (p > 0.f && l > mask_waited) {
            static const bool wait_all = getenv("FDIFF_TR_MASK_WAIT_ALL") != nullptr;
            const int upto = wait_all ? l : std::min(L - 1, 2 * l + 1);
            FD_HIP(ctx, hipStreamWaitEvent(s, ctx->side_events[upto], 0));
            mask_waited = upto;
        }
        {
            // measurement hook: Q / K / V projections + scores + P V of M tokens (the softmax itself is not matrix work)
            fd_prof_scope scope(ctx, s, "k_tr_attn_fwd (Q/K/V projections + softmax attention, training forward)",
                                (double)M * (6.0 * D * D + 4.0 * (double)T * D));
            if (attn_nw == 8) hipLaunchKernelGGL((k_tr_attn_fwd<KS1, 8>), dim3(d.NP, B), dim3(512), lds_attn, s, d, aa);
            else hipLaunchKernelGGL((k_tr_attn_fwd<KS1, 4>), dim3(d.NP, B), dim3(256), lds_attn, s, d, aa);
#ifdef FD_TR_PROF_AF
            {
                static int calls = 0;
                if (++calls == 30) {
                    unsigned long long h[64];
                    hipStreamSynchronize(s);
                    hipMemcpyFromSymbol(h, HIP_SYMBOL(fd_tr_af_dbg), sizeof(h));
                    for (int w = 0; w < attn_nw; ++w)
                        fprintf(stderr, "[attn_fwd dbg] wave %d over %d launches: staging %llu, barrier %llu, Q projection %llu, pass 1 %llu, pass 2 %llu, epilogue %llu cycles\n",
                                w, calls, h[w * 8 + 0] / calls, h[w * 8 + 1] / calls, h[w * 8 + 2] / calls, h[w * 8 + 3] / calls, h[w * 8 + 4] / calls, h[w * 8 + 5] / calls);
                }
            }
#endif
        }
        FfnFwdArgs fa{};
        fa.x0 = b.x0; fa.att = b.att; fa.s1 = b.s1; fa.s2 = b.s2;
        fa.out = (l + 1 < L) ? tb.layers[l + 1].x0 : tb.hL;
        fa.stage = b.stage;
        fa.outrb = (l + 1 < L) ? tb.layers[l + 1].x0rb : nullptr;
        fa.outT = (l + 1 < L) ? tb.layers[l + 1].x0T : nullptr;
        fa.active = b.active;
        fa.wo_img = limg + im->off_wo; fa.ffn_img = limg + im->off_ffn;
        fa.bo = P + lo.out_b; fa.g1 = P + lo.n1_w; fa.be1 = P + lo.n1_b; fa.b2 = P + lo.l2_b; fa.g2 = P + lo.n2_w; fa.be2 = P + lo.n2_b;
        fa.hkeep = b.hkeep; fa.rb1 = b.rb1; fa.rb3 = b.rb3;
        {
            // measurement hook (bench.py --mode train): algorithmic flops of this launch = out-projection + FFN of M tokens
            fd_prof_scope scope(ctx, s, "k_tr_ffn_fwd (out-proj + LN1 + FFN + LN2, training forward)",
                                (double)M * (4.0 * D * m->d.dim_ff + 2.0 * D * D));
            TrDims df = d;
            df.fsplit = tr_fsplit_rule(m, d.M, d.F, true);
            if (int rc = tr_fsplit_prepare(ctx, df, tb.nwg, DT, &fa.fs, s)) return rc;
            hipLaunchKernelGGL((k_tr_ffn_fwd<KS1, DT, KSO>), dim3(tb.nwg * df.fsplit), dim3(TW * 64), lds_ffn, s, df, fa);
#ifdef FD_TR_PROF_FFN
            {
                static int calls = 0;
                if (++calls == 60) {
                    unsigned long long h[16];
                    hipStreamSynchronize(s);
                    hipMemcpyFromSymbol(h, HIP_SYMBOL(fd_tr_ffn_dbg), sizeof(h));
                    static const char* nm[8] = {"entry + loads issued", "loads landed + out-proj + LN1", "staging + DMA wait", "barrier", "chunk loop",
                                                "stores + mask bits out", "exchange barriers", "owner epilogue"};
                    fprintf(stderr, "[k_tr_ffn_fwd phase clocks, workgroup 7, average of %d launches]\n", calls);
                    for (int w = 0; w < 2; ++w) {
                        fprintf(stderr, "  wave %d:", 4 * w);
                        for (int q = 0; q < 8; ++q) fprintf(stderr, " %s %.1f K |", nm[q], (double)h[w * 8 + q] / calls / 1000.0);
                        fprintf(stderr, "\n");
                    }
                }
            }
#endif
        }
    }
    if (out) fdgemm::linear_fwd(tb.hL, P + m->un_w, P + m->un_b, out, M, C, D, false, s);      // (null: the fused loss head reads hL)
    FD_LAUNCH_CHECK(ctx);
    if (p > 0.f) {      // the dropout-decision buffers may be rewritten once everything enqueued so far has run
        if (!readers_done) FD_HIP(ctx, hipEventRecord(ctx->tr_readers_event[mask_set], s));
        ctx->tr_readers_event_valid[mask_set] = true;
        ctx->tr_readers_gen = ctx->ws_gen;
    }
    return FD_OK;
}

template <int KS1, int DT, int KSO>
int tr_backward_t(fd_score* m, const float* dout, float* grads, int accumulate, hipStream_t s, TrBufs& tb, bool head_done) {
    fd_ctx* ctx = m->ctx;
    const fd_bf16_images* im = m->bf16;
    const int B = m->saved_B;
    const int T = m->d.max_len, C = m->d.n_channels, D = m->d.d_model, L = m->d.num_layers, F = m->d.dim_ff;
    const int M = B * T;
    const float* P = m->params;
    const TrDims d = make_dims(m, B, m->saved_p, m->saved_seed);
    if (!accumulate && !head_done)
        FD_HIP(ctx, hipMemsetAsync(grads, 0, sizeof(float) * (size_t)(L > 0 ? m->layers[0].in_w : m->nparams), s));
    // ---- unembedder (the fused loss head, fd_score_train_dsm, has already produced tb.dh and these two gradients)
    if (!head_done) {
        fdgemm::linear_bwd_weight(dout, tb.hL, grads + m->un_w, M, C, D, true, s, tb.skp, kSkpFloats);
        if (int rc = fd_colsum_det(ctx, dout, grads + m->un_b, M, C, s)) return rc;
        fdgemm::linear_bwd_input(dout, P + m->un_w, tb.dh, M, C, D, false, s);
    }
    const size_t scr = std::max((size_t)TW * KS1 * 1024, (size_t)4 * DT * 1024);
    const size_t NSh = (size_t)m->d.dim_ff / 64;
    const size_t lds_bwd = (size_t)4 * 2 * (2 * KS1 + DT) * 1024 + scr + (size_t)TW * 64 * NSh + (size_t)4 * 5 * 16 * DT * sizeof(float) +
                           (size_t)TW * KS1 * 1024 + 128 + (size_t)2 * 16 * DT * sizeof(float);      // (+ the lane-mask table, + gamma2 | gamma1)
    const int attn_nw = tr_attn_waves(d.KT);
    const bool attn_oh = tr_attn_oh_mode(d.KT) != 0;
    const int attn_parts = attn_oh ? 2 * d.NP : d.NP;             // partial tensors of d x written by k_tr_attn_bwd
    const int part_bf16 = tr_attn_oh_mode(d.KT) == 2 ? 1 : 0;
    // d x of a layer's attention side as ONE product with in_proj^T in its consumer instead of per-head partial tensors: the default where
    // the attention backward runs one head per workgroup (>= 12 token tiles: 12 bf16 tensors of d x per layer at 12 heads) -- same box,
    // B = 64: T = 252 2.091 -> 2.042 ms per step; T = 100 (head pairs: 6 fp32 tensors, 100 workgroups in k_tr_ffn_bwd: its prologue is
    // not bandwidth-bound there) 1.184 -> 1.195 ms (profiles/r06_train_dx_product_ab.txt).  FDIFF_TR_DX_GEMM=0 / 1 forces a form.
    const bool dxg = [&] { const char* e = getenv("FDIFF_TR_DX_GEMM"); return e ? atoi(e) != 0 : attn_oh; }() && d.NP <= 8;
    const size_t lds_ab = tr_attn_bwd_lds(d.T, attn_oh);
    if (lds_ab > 160 * 1024)
        return fd_fail(ctx, FD_ERR_UNSUPPORTED, "k_tr_attn_bwd: %zu bytes of LDS for max_len %d (form %d)", lds_ab, d.T, tr_attn_oh_mode(d.KT));
    static unsigned long long attr = 0;
    if (fd_first_on_device(attr, ctx->device)) {
        FD_HIP(ctx, hipFuncSetAttribute((const void*)k_tr_ffn_bwd<KS1, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FD_HIP(ctx, hipFuncSetAttribute((const void*)k_tr_attn_bwd<KS1, DT, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FD_HIP(ctx, hipFuncSetAttribute((const void*)k_tr_attn_bwd<KS1, DT, 8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FD_HIP(ctx, hipFuncSetAttribute((const void*)k_tr_attn_bwd<KS1, DT, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FD_HIP(ctx, hipFuncSetAttribute((const void*)k_tr_wgrad<KS1, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    // side stream: the weight gradients of layer l only need that layer's k_tr_ffn_bwd / k_tr_attn_bwd outputs, so they run
    // beside the input-gradient chain of layers l-1 .. 0 (both are latency-bound and leave most CUs idle on their own)
    if (!ctx->side_stream) FD_HIP(ctx, side_stream_create(&ctx->side_stream));
    if (!ctx->side_stream2) FD_HIP(ctx, side_stream_create(&ctx->side_stream2));
    while ((int)ctx->side_events.size() < L + 3) {
        hipEvent_t e;
        FD_HIP(ctx, hipEventCreateWithFlags(&e, kTrEventFlags));
        ctx->side_events.push_back(e);
    }
    // The request is never below 56 KiB: with the 13-KiB records the ring alone is 41-54 KiB, and at 41 KiB a third kind of
    // workgroup (the chain's backward kernels run beside this launch) became resident on the CUs next to the two
    // weight-gradient ones -- the T = 252 step went 2.32 -> 2.39 ms, against 2.28 ms with the footprint held at >= 56 KiB
    // (profiles/r05_tr16_stage.txt).  FDIFF_TR_WG_LDS_KB overrides the floor (experiments; above 80 KiB only one
    // weight-gradient workgroup fits a CU).
    static const size_t wg_pad = getenv("FDIFF_TR_WG_LDS_KB") ? (size_t)atoi(getenv("FDIFF_TR_WG_LDS_KB")) * 1024 : (size_t)56 * 1024;
    const size_t lds_wg = std::max((size_t)FD_TR_WG_NBUF * StageL<KS1, DT>::bytes + 128 + FD_TR_WG_NBUF * 512, wg_pad);      // (ring + lane-mask table + activity words)
    static const bool serial = getenv("FDIFF_TR_SERIAL") != nullptr;
    static const bool ext_event = !(getenv("FDIFF_TR_EXT_EVENT") && atoi(getenv("FDIFF_TR_EXT_EVENT")) == 0);
    const int lean_b = tr_lean_bits();
    const bool fb_rowsum = [] { const char* e = getenv("FDIFF_TR_FB_ROWSUM"); return !(e && atoi(e) == 0); }();      // (read per call: A/B, tests)
    WgArgs wa{};
    wa.nparams = (long long)tb.layer_params; wa.TS = tb.TS; wa.nblk = (M + 31) / 32;
    RedArgs ra{};
    ra.nparams = (long long)tb.layer_params; ra.TS = tb.TS; ra.D = D; ra.accumulate = accumulate;
    if (L > 0) {
        const fd_layer_off& l0 = m->layers[0];
        ra.rel[0] = l0.l2_b - l0.in_w; ra.rel[1] = l0.n2_b - l0.in_w; ra.rel[2] = l0.n2_w - l0.in_w;
        ra.rel[3] = l0.n1_b - l0.in_w; ra.rel[4] = l0.n1_w - l0.in_w;
        const long long wr[7] = {0, l0.in_b - l0.in_w, l0.out_w - l0.in_w, l0.out_b - l0.in_w, l0.l1_w - l0.in_w, l0.l1_b - l0.in_w, l0.l2_w - l0.in_w};
        const long long wn[7] = {3LL * D * D, 3LL * D, (long long)D * D, D, (long long)F * D, F, (long long)D * F};
        for (int k = 0; k < 7; ++k) { ra.wrel[k] = wr[k]; ra.wnum[k] = wn[k]; }
    }
    for (int l = L - 1; l >= 0; --l) {
        const fd_layer_off& lo = m->layers[l];
        TrLayerBufs& b = tb.layers[l];
        const char* limg = im->mimg + im->off_layers + (size_t)l * im->layer_stride;
        const char* bl = im->bimg + (size_t)l * im->b_layer_stride;
        const int par = l & 1;
        FfnBwdArgs fa{};
        if (l == L - 1) { fa.dy0 = tb.dh; fa.dyp = nullptr; fa.npart = 0; }
        else { fa.dy0 = tb.dres[par ^ 1]; fa.dyp = tb.dxp[par ^ 1]; fa.npart = attn_parts; }
        fa.part_stride = tb.part_stride; fa.part_bf16 = part_bf16;
        fa.rowsum = fb_rowsum && (D % 4 == 0) ? 1 : 0;
        fa.dxg = (dxg && l + 1 < L) ? 1 : 0;
        fa.dqkvR = reinterpret_cast<const __bf16*>(tb.dxp[par ^ 1]);      // (the rows of layer l + 1 live where its partial tensors would)
        fa.winT = l + 1 < L ? im->bimg + (size_t)(l + 1) * im->b_layer_stride + im->boff_win : nullptr;
        fa.s1 = b.s1; fa.s2 = b.s2; fa.active = b.active;
        fa.datt = tb.datt; fa.dres = tb.dres[par];
        fa.stage = b.stage; fa.doT = b.doT;
        fa.vecpart = tb.vecpart + (size_t)l * tb.nwg * 5 * D;
        fa.bffn = bl + im->boff_ffn; fa.wot = bl + im->boff_wot;
        fa.g1 = P + lo.n1_w; fa.be1 = P + lo.n1_b; fa.g2 = P + lo.n2_w;
        fa.rb1 = b.rb1; fa.rb3 = b.rb3;
        {
            // measurement hook: the input-gradient GEMMs W2^T, W1^T and out-proj^T of M tokens (the recomputed hidden chunk is
            // not algorithmic work)
            fd_prof_scope scope(ctx, s, "k_tr_ffn_bwd (LN2 bwd + FFN input gradient + LN1 bwd + out-proj^T, training backward)",
                                (double)M * (4.0 * D * F + 2.0 * D * D));
            if (int rc = tr_fsplit_prepare(ctx, d, tb.nwg, DT, &fa.fs, s)) return rc;
            hipLaunchKernelGGL((k_tr_ffn_bwd<KS1, DT>), dim3(tb.nwg * d.fsplit), dim3(TW * 64), lds_bwd, s, d, fa);
#ifdef FD_TR_PROF_FB
            {
                static int calls = 0;
                if (++calls == 300) {
                    unsigned long long h[16];
                    hipStreamSynchronize(s);
                    hipMemcpyFromSymbol(h, HIP_SYMBOL(fd_tr_fb_dbg), sizeof(h));
                    static const char* nm[8] = {"d y + partial tensors", "LN2 bwd + d f fragments", "DMA wait + barrier", "chunk loop", "exchange",
                                                "stage rows + LN1 bwd", "d o T-blocks + out-proj^T + stores", "barrier + column sums"};
                    fprintf(stderr, "[k_tr_ffn_bwd phase clocks, workgroup 7, K cycles, average of %d launches]\n", calls);
                    for (int w = 0; w < 2; ++w) {
                        fprintf(stderr, "  wave %d:", 4 * w);
                        for (int q = 0; q < 8; ++q) fprintf(stderr, " %s %.1f |", nm[q], (double)h[w * 8 + q] / calls / 1000.0);
                        fprintf(stderr, "\n");
                    }
                }
            }
#endif
        }
        AttnBwdArgs ab{};
        ab.x0rb = b.x0rb; ab.att = b.att; ab.datt = tb.datt; ab.lse2 = b.lse2; ab.pmask = b.pmask;
        ab.dxp = tb.dxp[par]; ab.dqkvT = b.dqkvT;
        ab.wk = limg + im->off_wk; ab.wv = limg + im->off_wv; ab.wq = limg + im->off_wq;
        ab.winT = bl + im->boff_win; ab.part_stride = tb.part_stride; ab.part_bf16 = part_bf16;
        ab.dqkvR = reinterpret_cast<__bf16*>(tb.dxp[par]); ab.dxg = dxg ? 1 : 0;
        {
            // measurement hook: dP = dO V^T, dV = P^T dO, dQ = dS K, dK = dS^T Q (2 T D each per token) + the in-proj^T GEMM; the
            // recomputed scores are not algorithmic work
            fd_prof_scope scope(ctx, s, "k_tr_attn_bwd (attention backward + in-proj^T, training backward)",
                                (double)M * (8.0 * (double)T * D + 6.0 * D * D));
            // The layer's weight-gradient launch (side stream) starts when this kernel has ended.  An event recorded behind the
            // kernel is a packet of its own between this kernel and the next layer's k_tr_ffn_bwd (6-7 us of an idle chip per
            // layer in the kernel trace); bound to the launch itself (hipExtLaunchKernelGGL's stop event) it is the kernel's own
            // completion signal.  FDIFF_TR_EXT_EVENT=0: the recorded event.
            hipEvent_t stop_ev = (ext_event && !serial) ? ctx->side_events[l] : nullptr;
            if (attn_oh) hipExtLaunchKernelGGL((k_tr_attn_bwd<KS1, DT, 4, 1>), dim3(2 * d.NP, B), dim3(256), lds_ab, s, nullptr, stop_ev, 0, d, ab);
            else if (attn_nw == 8) hipExtLaunchKernelGGL((k_tr_attn_bwd<KS1, DT, 8, 0>), dim3(d.NP, B), dim3(512), lds_ab, s, nullptr, stop_ev, 0, d, ab);
            else hipExtLaunchKernelGGL((k_tr_attn_bwd<KS1, DT, 4, 0>), dim3(d.NP, B), dim3(256), lds_ab, s, nullptr, stop_ev, 0, d, ab);
#ifdef FD_TR_PROF_ATTN
            {
                static int calls = 0;
                if (++calls == 30) {
                    unsigned long long h[64];
                    hipStreamSynchronize(s);
                    hipMemcpyFromSymbol(h, HIP_SYMBOL(fd_tr_attn_dbg), sizeof(h));
                    static const char* nm[6] = {"staging", "DMA wait + barrier", "bit transposition", "query sweeps", "key sweeps", "epilogues"};
                    fprintf(stderr, "[k_tr_attn_bwd phase clocks, workgroup (0,0), average of %d launches, per wave]\n", calls);
                    for (int w = 0; w < (attn_oh ? 4 : attn_nw); ++w) {
                        fprintf(stderr, "  wave %d:", w);
                        for (int q = 0; q < 6; ++q) fprintf(stderr, " %s %.1f K |", nm[q], (double)h[w * 8 + q] / calls / 1000.0);
                        fprintf(stderr, "\n");
                    }
                }
            }
#endif
        }
        WgLayer w{};
        w.x0T = b.x0T; w.attT = b.attT; w.doT = b.doT; w.dqkvT = b.dqkvT;
        w.stage = b.stage; w.active = b.active;
        w.ffn_img = limg + im->off_ffn; w.bffn = bl + im->boff_ffn;
        // offsets relative to the layer's first parameter: the partials hold one layer
        w.in_w = 0; w.in_b = lo.in_b - lo.in_w; w.out_w = lo.out_w - lo.in_w; w.out_b = lo.out_b - lo.in_w;
        w.l1_w = lo.l1_w - lo.in_w; w.l1_b = lo.l1_b - lo.in_w; w.l2_w = lo.l2_w - lo.in_w;
        hipStream_t ws = (l & 1) ? ctx->side_stream2 : ctx->side_stream;     // two layers' weight gradients in flight
        if (serial) ws = s;                                                   // measurement: solo kernel times
        wa.part = tb.part + (size_t)(l & 1) * kMaxTS * tb.layer_params;
        // Layer 0's launch is the step's tail: the input-gradient chain has ended, nothing shares the chip with it and the
        // optimizer waits for it.  More token splits shorten each workgroup's block chain (experiments: FDIFF_TR_TS_LAST).
        static const int ts_last_env = getenv("FDIFF_TR_TS_LAST") ? atoi(getenv("FDIFF_TR_TS_LAST")) : 0;
        const int ts_l = (l == 0 && ts_last_env > 0) ? std::max(1, std::min({kMaxTS, ts_last_env, wa.nblk})) : tb.TS;
        wa.TS = ra.TS = ts_l;
        if (!(ext_event && !serial)) FD_HIP(ctx, hipEventRecord(ctx->side_events[l], s));
        if (ws != s) FD_HIP(ctx, hipStreamWaitEvent(ws, ctx->side_events[l], 0));
        {
            // measurement hook (bench.py --mode train): every weight gradient of one layer -- in_proj 2 M 3D D, out_proj 2 M D D,
            // linear1 + linear2 2 x 2 M D F (the recomputation of the hidden / d hidden blocks is not algorithmic work)
            fd_prof_scope scope(ctx, ws, "k_tr_wgrad (all weight gradients of one encoder layer, training backward)",
                                (double)M * (8.0 * D * D + 4.0 * D * F));
            hipLaunchKernelGGL((k_tr_wgrad<KS1, DT>), dim3(F / 128 + 4, ts_l), dim3(256), lds_wg, ws, d, w, wa);
        }
        ra.part = wa.part;
        ra.grads = grads + lo.in_w;
        hipLaunchKernelGGL(k_tr_reduce, dim3((unsigned)((tb.layer_params / 4 + 255) / 256)), dim3(256), 0, ws, ra);
    }
    if (m->saved_p > 0.f && L > 0) {      // last reader of the dropout-decision buffers on `s` (layer 0's attention backward)
        hipEvent_t& rev = ctx->tr_readers_event[m->saved_mask_set & 1];
        if (!rev) FD_HIP(ctx, hipEventCreateWithFlags(&rev, kTrEventFlags));
        FD_HIP(ctx, hipEventRecord(rev, s));
        ctx->tr_readers_event_valid[m->saved_mask_set & 1] = true;
        ctx->tr_readers_gen = ctx->ws_gen;
    }
    // The input-gradient chain is complete; the last weight-gradient launches are still running on the side streams.  The
    // embedding-side backward (first layer's input gradient, positional / time / embedder gradients: ~60 us of small kernels)
    // needs none of them and runs first, the fixed-order reduce of the weight-gradient partials after the side streams join.
    if (L > 0) {
        // gradient of the first layer's input = residual path + the pairs' in_proj contributions
        const size_t nn = (size_t)M * D;
        if (dxg)
            hipLaunchKernelGGL((k_tr_dx0<DT>), dim3((unsigned)((M + 63) / 64)), dim3(256), ((size_t)3 * d.NP * DT * 512 + 1023) & ~(size_t)1023, s, d, (const float*)tb.dres[0],
                               reinterpret_cast<const __bf16*>(tb.dxp[0]), im->bimg + im->boff_win, tb.dh);
        else if (part_bf16)
            hipLaunchKernelGGL(k_tr_sum_parts_bf16, dim3((unsigned)((nn / 4 + 256) / 256)), dim3(256), 0, s, tb.dres[0],
                               reinterpret_cast<const __bf16*>(tb.dxp[0]), attn_parts, tb.part_stride, tb.dh, nn);
        else
            hipLaunchKernelGGL(k_tr_sum_parts, dim3((unsigned)((nn / 4 + 256) / 256)), dim3(256), 0, s, tb.dres[0], tb.dxp[0], attn_parts,
                           tb.part_stride, tb.dh, nn);
    }
    if (L > 0) {
        VecRedArgs va{};
        va.vecpart = tb.vecpart; va.nwg = tb.nwg; va.D = D; va.begin = m->layers[0].in_w; va.layer_stride = (long long)tb.layer_params;
        for (int k = 0; k < 5; ++k) va.rel[k] = ra.rel[k];
        va.grads = grads; va.accumulate = accumulate;
        hipLaunchKernelGGL(k_tr_vecreduce, dim3(5, L), dim3(1024), 0, s, va);
    }
    const int rc_embed = fd_embed_backward(m, tb.dh, tb.emb, tb.dtemb, grads, B, tb.skp, kSkpFloats, s);
    if (rc_embed) {
        // the weight-gradient launches on the side streams are still writing tb.part (arena memory): let them finish before the
        // caller sees the error and reuses or frees the arena
        (void)hipStreamSynchronize(ctx->side_stream);
        (void)hipStreamSynchronize(ctx->side_stream2);
        return rc_embed;
    }
    if (L > 0 && (lean_b & 8) && !serial) {
        // one join packet on `s`, not two: side stream 1 (layer 0's weight gradients, the step's last) first waits for side stream 2
        FD_HIP(ctx, hipEventRecord(ctx->side_events[L + 2], ctx->side_stream2));
        FD_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->side_events[L + 2], 0));
        FD_HIP(ctx, hipEventRecord(ctx->side_events[L], ctx->side_stream));
        FD_HIP(ctx, hipStreamWaitEvent(s, ctx->side_events[L], 0));
    } else if (L > 0) {
        FD_HIP(ctx, hipEventRecord(ctx->side_events[L], ctx->side_stream));
        FD_HIP(ctx, hipStreamWaitEvent(s, ctx->side_events[L], 0));
        FD_HIP(ctx, hipEventRecord(ctx->side_events[L + 2], ctx->side_stream2));
        FD_HIP(ctx, hipStreamWaitEvent(s, ctx->side_events[L + 2], 0));
    }
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

}  // namespace

bool fd_train_bf16_supported(const fd_score* m) {
    const fd_bf16_images* im = m->bf16;
    // dim_ff: the FFN kernels keep one keep-byte per 32-wide chunk per lane in 16-byte groups (F % 1024 == 0) and their LDS
    // budget covers F <= 2048 (torch's default 2048 is the only value the reference uses)
    // max_len: the attention backward holds a head's q / k / v / dO images and its keep bits (both orientations) for the whole
    // series in LDS: 592 time steps in the one-head form (37 token tiles), which is the form from 12 tiles on
    const int KT = (m->d.max_len + 15) / 16;
    return im && im->train && im->bimg && m->d.dim_ff % 1024 == 0 && m->d.dim_ff <= 2048 && m->d.num_layers > 0 &&
           tr_attn_bwd_lds(m->d.max_len, KT >= 12) <= 160 * 1024;
}

size_t fd_train_bf16_workspace(const fd_score* m, int B) { return tr_carve(m, B, nullptr, nullptr); }

#define FD_TR_DISPATCH(CALL)                                                                         \
    do {                                                                                             \
        const fd_bf16_images* im_ = m->bf16;                                                         \
        if (im_->ks1 == 3 && im_->dt == 5 && im_->kso == 3) return CALL(3, 5, 3);                    \
        if (im_->ks1 == 3 && im_->dt == 5 && im_->kso == 2) return CALL(3, 5, 2);                    \
        if (im_->ks1 == 2 && im_->dt == 3 && im_->kso == 1) return CALL(2, 3, 1);                    \
        if (im_->ks1 == 2 && im_->dt == 4 && im_->kso == 3) return CALL(2, 4, 3);                    \
        if (im_->ks1 == 1 && im_->dt == 2 && im_->kso == 1) return CALL(1, 2, 1);                    \
        if (im_->ks1 == 1 && im_->dt == 1 && im_->kso == 1) return CALL(1, 1, 1);                    \
        return fd_fail(m->ctx, FD_ERR_UNSUPPORTED, "bf16 training kernels not instantiated for this model"); \
    } while (0)

// Rebuild of the bf16 weight images on side stream 2, behind everything `s` holds now; completes m->img_event.
static int tr_fork_image_rebuild(fd_score* m, hipStream_t s) {
    fd_ctx* ctx = m->ctx;
    if (!ctx->side_stream2) FD_HIP(ctx, side_stream_create(&ctx->side_stream2));
    if (!m->img_event) FD_HIP(ctx, hipEventCreateWithFlags(&m->img_event, kTrEventFlags));
    while ((int)ctx->side_events.size() < m->d.num_layers + 3) {
        hipEvent_t e;
        FD_HIP(ctx, hipEventCreateWithFlags(&e, kTrEventFlags));
        ctx->side_events.push_back(e);
    }
    // what the rebuild waits for: fd_score_prepare's kernel on this stream (behind the optimizer's), by that launch's own stop
    // event where there is one -- no event packet on `s`
    if ((tr_lean_bits() & 16) && m->prep_event_bound && m->prep_stream == (void*)s) {
        FD_HIP(ctx, hipStreamWaitEvent(ctx->side_stream2, m->prep_event, 0));
    } else {
        FD_HIP(ctx, hipEventRecord(ctx->side_events[m->d.num_layers], s));
        FD_HIP(ctx, hipStreamWaitEvent(ctx->side_stream2, ctx->side_events[m->d.num_layers], 0));
    }
    if (int rc = fd_bf16_refresh(m, ctx->side_stream2, true)) return rc;
    FD_HIP(ctx, hipEventRecord(m->img_event, ctx->side_stream2));
    return FD_OK;
}

int fd_score_forward_train_bf16(fd_score* m, const float* x, const float* t, float* out, int B, float p, uint64_t seed,
                                uint64_t offset, hipStream_t s) {
    fd_ctx* ctx = m->ctx;
    if (int rc = fd_train_async_check(ctx)) return rc;
    if (!fd_train_bf16_supported(m)) return fd_fail(ctx, FD_ERR_UNSUPPORTED, "bf16 training path unsupported for this model");
    // The optimizer step made the bf16 weight images stale.  Rebuilding them (~50 us of small kernels) needs nothing but the
    // parameters, and the step's prologue on `s` (time embedding, embedding, first layer's operand preparation) does not need
    // the images: the rebuild runs beside it on the second side stream and joins `s` in front of the first attention kernel.
    bool img_forked = false;
    if (m->bf16_stale && m->d.num_layers > 0 && !getenv("FDIFF_TR_SERIAL")) {
        if (int rc = tr_fork_image_rebuild(m, s)) return rc;
        img_forked = true;          // (tr_forward_t waits for m->img_event in front of the first kernel that reads the images)
    } else if (int rc = fd_bf16_refresh(m, s, true)) {
        return rc;
    }
    const size_t need = fd_train_bf16_workspace(m, B);
    // ws_gen before this call touches the arena; the reserve + carve below advance it by exactly two, so "nobody else used the
    // arena since the last training reader" reads gen_in == tr_readers_gen
    const uint64_t gen_in = ctx->ws_gen;
    const void* ws_before = ctx->ws;
    if (int rc = fd_ws_reserve(ctx, need)) return rc;
    fd_ws ws(ctx);
    // Somebody else carved the arena since the last training call (an eval forward, the sampler, another model on this context) and may
    // still be running on `s` over memory the decision kernels are about to write, or the arena moved: both sets take the conservative
    // path at their next use (an event recorded on `s` in front of the decision kernels instead of behind the set's last reader).
    // (A different model or batch size also counts: its carve puts the decision buffers where the other layout keeps activations.)
    if (gen_in != ctx->tr_readers_gen || ctx->ws != ws_before || ctx->tr_last_model != (const void*)m || ctx->tr_last_B != B)
        ctx->tr_readers_event_valid[0] = ctx->tr_readers_event_valid[1] = false;
    ctx->tr_last_model = m; ctx->tr_last_B = B;
    const int mask_set = (int)(ctx->tr_mask_steps++ & 1u);
    m->saved_mask_set = mask_set;
    TrBufs tb;
    tr_carve(m, B, (char*)ctx->ws, &tb, mask_set);
    ctx->tr_readers_gen = ctx->ws_gen;      // (own carve accounted for: a training forward that ends without reaching its last event record keeps the sets usable)
#define CALL_F(K, T_, O) tr_forward_t<K, T_, O>(m, x, t, out, B, p, seed, offset, s, tb, gen_in, img_forked, mask_set)
    FD_TR_DISPATCH(CALL_F);
#undef CALL_F
}

int fd_score_backward_bf16(fd_score* m, const float* dout, float* grads, int accumulate, hipStream_t s) {
    fd_ctx* ctx = m->ctx;
    if (int rc = fd_train_async_check(ctx)) return rc;
    const size_t need = fd_train_bf16_workspace(m, m->saved_B);
    if (ctx->ws_bytes < need) return fd_fail(ctx, FD_ERR_STATE, "fd_score_backward: workspace was resized since the training forward");
    TrBufs tb;
    tr_carve(m, m->saved_B, (char*)ctx->ws, &tb, m->saved_mask_set);
#define CALL_B(K, T_, O) tr_backward_t<K, T_, O>(m, dout, grads, accumulate, s, tb, false)
    FD_TR_DISPATCH(CALL_B);
#undef CALL_B
}

// ---- forward + denoising score-matching loss + backward as ONE call (fd_score_train_dsm) ----
// The three-call form (fd_score_forward_train -> fd_dsm_loss -> fd_score_backward) runs the unembedder, the loss and the
// unembedder's backward as eight launches (three fp32 GEMMs with split-K reduces, a two-stage column sum, the loss and its
// sum: ~75 us of 5-20 us kernels on (B*T, C <= 40) data).  Here the loss head is one kernel per (series, token split) that
// keeps the unembedder weight in LDS, plus one fixed-order reduce of its partials.
// The fused loss head's launch plan for a batch of B series; false = not instantiated for this (model, B) (the caller then runs
// forward -> fd_dsm_loss -> backward).  No side effects: fd_score_train_dsm_supported answers from it before any Philox key is drawn.
static bool tr_head_plan(const fd_score* m, int B, int* TS_out, int* kmax_out, size_t* lds_out) {
    const int T = m->d.max_len, C = m->d.n_channels, D = m->d.d_model;
    const int CD = C * D;
    // register slots per thread: C * D weight-gradient entries and 32 * C (row, channel) items over 256 threads
    const int slots = std::max((CD + 255) / 256, (32 * C + 255) / 256);
    const int kmax = slots <= 2 ? 2 : slots <= 4 ? 4 : slots <= 8 ? 8 : slots <= 16 ? 16 : 0;
    const size_t lds = sizeof(float) * ((size_t)C * (D + 1) + (size_t)32 * (D + 1) + (size_t)32 * C + 16);
    const size_t prow = (size_t)CD + C + 1;
    int TS = (T + 31) / 32;                            // one 32-token tile per workgroup whenever the partials fit
    while (TS > 1 && (size_t)B * TS * prow > kSkpFloats) --TS;
    if (!kmax || D > 80 || D < 8 || T > 4096 || lds > 64 * 1024 || (size_t)B * TS * prow > kSkpFloats) return false;
    *TS_out = TS; *kmax_out = kmax; *lds_out = lds;
    return true;
}
int fd_train_bf16_token_splits(const fd_score* m, int B, int* nblk) {
    if (nblk) *nblk = (int)(((long long)B * m->d.max_len + 31) / 32);
    return tr_TS(m, B);
}
void fd_train_bf16_forward_plan(const fd_score* m, int B, char* out, size_t n) {
    int nq = 0, spl = 0;
    const char* e = getenv("FDIFF_TR_PERSIST");
    const int mode = e ? atoi(e) : 1;
    const int nt = (mode != 0 && m->d.num_layers > 0 && !m->ctx->trp_disabled) ? fd_trp_tiles(m, B, &nq, &spl) : 0;
    if (nt) snprintf(out, n, "k_tr_fwd_layers NT=%d, %d x %d workgroups%s", nt, nq, std::min(B, spl), mode == 2 ? " per layer" : "");
    else snprintf(out, n, "2 kernels per layer%s", m->ctx->trp_disabled ? " (persistent form disabled after a timeout)" : "");
}
bool fd_score_train_dsm_bf16_supported(const fd_score* m, int B) {
    int TS, kmax;
    size_t lds;
    return fd_train_bf16_supported(m) && tr_head_plan(m, B, &TS, &kmax, &lds);
}

int fd_score_train_dsm_bf16(fd_score* m, const float* x, const float* t, const float* target, const float* stdv, int lw,
                            float grad_weight, int B, float p, uint64_t seed, uint64_t offset, float* loss_out, float* grads,
                            int accumulate, hipStream_t s) {
    fd_ctx* ctx = m->ctx;
    if (int rc = fd_train_async_check(ctx)) return rc;
    if (!fd_train_bf16_supported(m)) return fd_fail(ctx, FD_ERR_UNSUPPORTED, "bf16 training path unsupported for this model");
    const int T = m->d.max_len, C = m->d.n_channels, D = m->d.d_model;
    const int CD = C * D;
    const size_t prow = (size_t)CD + C + 1;
    int TS = 0, kmax = 0;
    size_t lds = 0;
    if (!tr_head_plan(m, B, &TS, &kmax, &lds))
        return fd_fail(ctx, FD_ERR_UNSUPPORTED, "fd_score_train_dsm: loss head not instantiated for C=%d, d_model=%d, B=%d", C, D, B);
    if (int rc = fd_score_forward_train_bf16(m, x, t, nullptr, B, p, seed, offset, s)) return rc;
    m->saved_bf16 = true; m->have_saved = false;       // consumed by the backward below
    m->saved_B = B; m->saved_p = p; m->saved_seed = seed; m->saved_offset = offset; m->saved_x = x; m->saved_t = t;
    m->saved_ws_gen = ctx->ws_gen; m->saved_ws = ctx->ws;
    TrBufs tb;
    tr_carve(m, B, (char*)ctx->ws, &tb, m->saved_mask_set);
    HeadArgs ha{};
    ha.hL = tb.hL; ha.Wu = m->params + m->un_w; ha.bu = m->params + m->un_b; ha.target = target; ha.stdv = stdv;
    ha.dh = tb.dh; ha.part = tb.skp; ha.T = T; ha.C = C; ha.D = D; ha.TS = TS; ha.lw = lw;
    ha.inv_cnt = 1.0f / ((float)((size_t)T * C) * (float)B); ha.gw = grad_weight;
    const dim3 grid(TS, B);
    switch (kmax) {
        case 2: hipLaunchKernelGGL(k_tr_head<2>, grid, dim3(256), lds, s, ha); break;
        case 4: hipLaunchKernelGGL(k_tr_head<4>, grid, dim3(256), lds, s, ha); break;
        case 8: hipLaunchKernelGGL(k_tr_head<8>, grid, dim3(256), lds, s, ha); break;
        default: hipLaunchKernelGGL(k_tr_head<16>, grid, dim3(256), lds, s, ha); break;
    }
    // (tr_backward_t skips its own clear of the non-layer gradients when the head has already added to them)
    if (!accumulate)
        FD_HIP(ctx, hipMemsetAsync(grads, 0, sizeof(float) * (size_t)(m->d.num_layers > 0 ? m->layers[0].in_w : m->nparams), s));
    hipLaunchKernelGGL(k_tr_head_final, dim3((unsigned)((prow + 15) / 16)), dim3(256), 0, s, (const float*)tb.skp, B * TS, CD, C,
                       grads + m->un_w, grads + m->un_b, loss_out);
    FD_LAUNCH_CHECK(ctx);
#define CALL_B(K, T_, O) tr_backward_t<K, T_, O>(m, nullptr, grads, accumulate, s, tb, true)
    FD_TR_DISPATCH(CALL_B);
#undef CALL_B
}
