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
// Byte offset of the 8-byte K row of (key tile kt, token tok of the tile, lane group g) in the K image.  A score MFMA's A fragment
// is one ds_read_b64 per lane, serviced in the lane groups {0-31}, {32-63} = the lane groups g in {0, 1} / {2, 3} of all 16 tokens:
// in the natural order [token][g] (32 bytes per token) such a group reads the first / second 16 bytes of 16 rows, i.e. a 32-byte
// stride, and tokens t and t + 8 share their banks -- a 2-way conflict on EVERY K read (63 % of the kernel's LDS-active cycles were
// conflict cycles, profiles/r05_attn_long_pmc_summary.txt).  [tile][g >> 1][token][g & 1] makes each group 256 contiguous bytes.
#ifndef FD_ATTN_KSWZ
#define FD_ATTN_KSWZ 1
#endif
__device__ __forceinline__ size_t krow(int kt, int tok, int g) {
#if FD_ATTN_KSWZ
    return (size_t)kt * 512 + (size_t)((g >> 1) * 256 + tok * 16 + (g & 1) * 8);
#else
    return ((size_t)(kt * 16 + tok) * 4 + g) * 8;
#endif
}
#ifndef FD_ATTN_LAG
#define FD_ATTN_LAG 2
#endif
#ifndef FD_ATTN_WAVES
#define FD_ATTN_WAVES 4      // waves per SIMD the register budget is set for (4 = two 8-wave workgroups per CU)
#endif
#ifdef FD_ATTN_ABL
constexpr int ABL = FD_ATTN_ABL;     // ablation builds (scripts/attn_abl.sh): 1 = staging only, 2 = one staged tile per wave, no rescue
#else
constexpr int ABL = 0;
#endif
#if defined(FD_ATTN_ABL) && FD_ATTN_ABL == 3
__device__ unsigned long long fd_attn_dbg[8 * 8];   // per wave of workgroup 0: staging, Q setup, key blocks, epilogue, units
#define ATTN_STAMP(slot, t_prev)                                                                   \
    do {                                                                                           \
        const unsigned long long now_ = __builtin_readcyclecounter();                              \
        if (blockIdx.x == 0 && lane == 0) fd_attn_dbg[wave * 8 + (slot)] += now_ - (t_prev); \
        (t_prev) = now_;                                                                           \
    } while (0)
#else
#define ATTN_STAMP(slot, t_prev) do { } while (0)
#endif

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
// KS1 > 0 : `in` = the layer input h (B*T, D) fp32, D % 4 == 0; Q, K, V of the pair are projected inside the kernel by MFMA
//           (one launch and the (B*T, 3D) round trip less per layer).
// out (B*T, D) fp32.  grid (query slices, head pairs, B), 512 threads.
//
// Phases of a workgroup (in-kernel time stamps at T = 1024, -DFD_ATTN_ABL=3: staging 27 %, per-unit Q set-up 14 %, key
// blocks 56 % before this form):
//   1. staging: K, V^T of the whole series and Q of this slice's query tiles -> LDS as bf16 fragments.  The x rows of the
//      wave's NEXT token tile are in flight (raw fp32 registers) while the current tile runs through the MFMAs.
//   2. units (2 query tiles x both heads): Q fragments come from LDS (no global round trip per unit); the key loop is ONE
//      software pipeline over the whole series -- stage A (QK^T MFMA16) of a 128-key block runs while stages B (exp2, pack)
//      and C (P V MFMA) finish the previous block's last tiles, and the next block's K / V fragments are read from LDS into
//      the registers that die during the current block.  scripts/ubench/attn_pattern.hip prices the steady state at
//      ~1.6 K cycles per block and SIMD (VALU: 128 exp2 x 8 + 64 cvt_pk x 4 = 1.28 K); draining and refilling the pipeline at
//      every block boundary cost 2.0 K.
//   ROWS (with KS1 > 0): the layer input also exists as bf16 rows `xrows` (B*T, 32 KS1 k-slots, 1.0 in slot D, zero padding --
//      written by the producer of the layer input: k_ffn_ln's epilogue / k_unembed_step_embed): a token tile's B fragment is
//      then ONE 16-byte load per k-step and lane, three instructions walking 16 x 64 contiguous bytes each, against six fp32
//      loads walking 32 half-used lines each plus twelve conversions and their selects.  The staging phase of this kernel is
//      bound by the CU's vector-memory address path (phase clocks at T = 1024: 20 K of a workgroup's 88 K cycles with fp32 rows,
//      profiles/r05_attn_long_phase_clocks_before.txt).
template <int KS1, int ROWS = 0>
__global__ __launch_bounds__(NTH, FD_ATTN_WAVES) void k_attention_bf16(const float* __restrict__ in, float* __restrict__ out, int T,
                                                           int H, int hd, int D, float qscale, int du_per_block, int exact_only,
                                                           fd_attn_w wimg, size_t pair_stride, int slices, int B, int out_bf16,
                                                           const __bf16* __restrict__ xrows, int KTP, int nblk1, int B1, int slices2,
                                                           int dpb2) {
    constexpr bool PROJ = KS1 > 0;
    constexpr int KSN = PROJ ? KS1 : 1;
    // Workgroup -> (series, head pair, query slice).  Hardware workgroup ids go round-robin over the 8 XCDs; all
    // NP * slices workgroups of a series are placed on ONE XCD (series s on XCD s % 8), so the series' x rows (and nothing
    // else) stream through that XCD's L2 once and are re-read from it by the other NP * slices - 1 workgroups.
    // Two classes of workgroups (host: fd_attention_bf16): the first nblk1 blocks cover the series [0, B1) with `slices` query slices each,
    // the rest cover [B1, B) with `slices2` = twice as many, half as long ones -- dispatched LAST, they fill every slot of the chip in the
    // final round (768 equal workgroups on 512 slots ran a second round with half of the slots empty and the VALU half used).
    int bidx = blockIdx.x, boff = 0, Bc = B1;
    if (bidx >= nblk1) {
        bidx -= nblk1; boff = B1; Bc = B - B1;
        slices = slices2; du_per_block = dpb2;
    }
    const int NPs = ((H + 1) >> 1) * slices;
    const int xcd = bidx & 7, slot = bidx >> 3;
    const int bl = (slot / NPs) * 8 + xcd, wsl = slot % NPs;
    if (bl >= Bc) return;
    const int b = boff + bl;
    const int pair = wsl / slices, slice = wsl % slices;
    const float* __restrict__ qkv = in;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, tok = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // KTP >= KT: key tiles in LDS.  A series whose last 128-key block is partial may be padded to whole blocks with all-zero key
    // tiles (host policy): the fast path then runs the block through the software pipeline instead of the rolled pair loop.
    const int KT = (T + 15) >> 4, NJ = (KTP + 1) >> 1, NTOK = KTP * 16;
    char* const kbf = smem;                       // [key tile][g >> 1][16 tokens][g & 1][8 B] (krow): lane group g = 2*hs + (d >> 2), element d & 3
    char* const vbf = smem + (size_t)NTOK * 32;   // [NJ][4 g][16 dim slots][16 B]: (half, r) -> token (2jj+half)*16 + 4g + r
    char* const qbf = vbf + (size_t)NJ * 1024;    // [slice tile][64 lanes][8 B]: Q^T C tile as bf16 (scaled by log2e/sqrt(hd))
    // [2] max_j |k_j|^2 per head (bits).  With hd < 7 dim slot 7 of V^T is padding: its row feeds only output row 7, which
    // nobody reads, so the two words live in that row of (group 0, lane group 0) and K + V^T + Q of T = 1024 are exactly
    // 80 KiB: two workgroups per CU.  hd == 7 (slot 7 = the ones row): 16 bytes behind Q.
    const bool kmax_in_v = hd < 7;
    unsigned* const kmax = reinterpret_cast<unsigned*>(kmax_in_v ? vbf + 7 * 16 : qbf + (size_t)du_per_block * NQ * 512);
    const float* base = qkv + (size_t)b * T * (PROJ ? 1 : 3) * D;
    const int DUS = (KT + NQ - 1) / NQ;
    const int du0 = slice * du_per_block, du1 = (ABL == 1) ? du0 : min(DUS, du0 + du_per_block);
    const int qt0 = du0 * NQ, qt1 = min(KT, (du0 + du_per_block) * NQ);   // this slice's query tiles
    unsigned long long tprev = __builtin_readcyclecounter();
    (void)tprev;

    // ---- stage K and V^T of this (series, pair) as bf16 fragments; every byte of both regions is written here
    if (threadIdx.x < 2) kmax[threadIdx.x] = 0u;
    __syncthreads();
    if (PROJ) {
        // K^T = W_k x^T (C tile rows = the pair's 16 dim slots) and V = x W_v^T (C tile rows = tokens: already the
        // V^T A-fragment layout; its ones row comes from the bias slot) per token tile, exactly as in fd_mega.hip
        auto wfrag = [&](const char* img, int ks) -> bf16x8 {
            return *reinterpret_cast<const bf16x8*>(img + (size_t)pair * pair_stride + ((size_t)ks * 64 + lane) * 16);
        };
        bf16x8 wkf[KSN], wvf[KSN], wqf[KSN];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            wkf[ks] = wfrag(wimg.wk, ks);
            wvf[ks] = wfrag(wimg.wv, ks);
            wqf[ks] = wfrag(wimg.wq, ks);
        }
        // raw x rows of a token tile: lane (tok, g) reads features 32ks + 8g .. +7 (clamped addresses, fixed up in xconv)
        auto xload = [&](int tile, float4 (&rlo)[KSN], float4 (&rhi)[KSN]) {
            const int t = min(tile * 16 + tok, T - 1);
            const float* row = base + (size_t)t * D;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int k0 = 32 * ks + 8 * g;
                rlo[ks] = *reinterpret_cast<const float4*>(row + min(k0, D - 4));
                rhi[ks] = *reinterpret_cast<const float4*>(row + min(k0 + 4, D - 4));
            }
        };
        // B fragment: features beyond D read 0, slot D the constant 1.0 (bias fold), rows beyond T all 0
        auto xconv = [&](int tile, float4 a, float4 c, int ks) -> bf16x8 {
            const bool tv = tile * 16 + tok < T;
            const int k0 = 32 * ks + 8 * g;
            const float4 one = {1.f, 0.f, 0.f, 0.f}, zero = {0.f, 0.f, 0.f, 0.f};
            if (!(tv && k0 + 4 <= D)) a = (tv && k0 == D) ? one : zero;
            if (!(tv && k0 + 8 <= D)) c = (tv && k0 + 4 == D) ? one : zero;
            const u32x4 pk = {cvt_pk_bf16(a.x, a.y), cvt_pk_bf16(a.z, a.w), cvt_pk_bf16(c.x, c.y), cvt_pk_bf16(c.z, c.w)};
            return __builtin_bit_cast(bf16x8, pk);
        };
        if constexpr (ROWS != 0) {
            constexpr int RBW = 32 * KSN;
            const __bf16* rbase = xrows + (size_t)b * T * RBW + 8 * g;
            // fragments of a tile straight from the bf16 rows (clamped row; rows beyond T are cleared at their use)
            auto rload = [&](int tile, u32x4 (&r)[KSN]) {
                const int t = min(tile * 16 + tok, T - 1);
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) r[ks] = *reinterpret_cast<const u32x4*>(rbase + (size_t)t * RBW + 32 * ks);
            };
            // PB tiles' rows in flight per wave (a ring of register slots: a tile's slot is reloaded with the tile PB steps ahead as
            // soon as its MFMAs have consumed it): with one tile ahead the staging phase was a chain of exposed L2 / Infinity-Cache
            // round trips, 14.4 K cycles for a wave's 8 tiles at T = 1024 (profiles/r05_attn_long_phase_clocks_xrows.txt)
            constexpr int PB = 4;
            u32x4 rr[PB][KSN];
#pragma unroll
            for (int p = 0; p < PB; ++p) rload(min(wave + p * NW, KT - 1), rr[p]);
            const int kt_end = (ABL == 2 ? min(KT, NW) : KT);
            for (int kt0 = wave; kt0 < kt_end; kt0 += NW * PB) {
#pragma unroll
                for (int p = 0; p < PB; ++p) {
                    const int kt = kt0 + p * NW;
                    if (kt >= kt_end) break;                       // wave-uniform
                    f32x4 a = f4zero(), c = f4zero(), qa = f4zero();
                    const bool isq = kt >= qt0 && kt < qt1;        // wave-uniform
                    const unsigned keep = (kt * 16 + tok < T) ? ~0u : 0u;
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        const bf16x8 xf = __builtin_bit_cast(bf16x8, u32x4{rr[p][ks][0] & keep, rr[p][ks][1] & keep, rr[p][ks][2] & keep, rr[p][ks][3] & keep});
                        a = MFMA(wkf[ks], xf, a);
                        c = MFMA(xf, wvf[ks], c);
                        if (isq) qa = MFMA(wqf[ks], xf, qa);
                    }
                    if (kt + NW * PB < kt_end) rload(kt + NW * PB, rr[p]);
                    *reinterpret_cast<u32x2*>(kbf + krow(kt, tok, g)) = u32x2{cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3])};
                    char* dst = vbf + ((size_t)((kt >> 1) * 4 + g) * 16 + tok) * 16;
                    if (!(kmax_in_v && kt == 0 && lane == 7))          // (that row's first 8 bytes hold kmax)
                        *reinterpret_cast<u32x2*>(dst + 8 * (kt & 1)) = u32x2{cvt_pk_bf16(c[0], c[1]), cvt_pk_bf16(c[2], c[3])};
                    if ((KT & 1) && kt == KT - 1 && KTP == KT) *reinterpret_cast<u32x2*>(dst + 8) = u32x2{0u, 0u};
                    if (isq)
                        *reinterpret_cast<u32x2*>(qbf + ((size_t)(kt - qt0) * 64 + lane) * 8) = u32x2{cvt_pk_bf16(qa[0], qa[1]), cvt_pk_bf16(qa[2], qa[3])};
                    float n2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
                    float ea, eb;
                    swap16(n2, ea, eb);
                    n2 = row_max16(ea + eb);
                    if (tok == 0 && (g & 1) == 0)
                        __hip_atomic_fetch_max(&kmax[g >> 1], __builtin_bit_cast(unsigned, n2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        } else {
        float4 cur_lo[KSN], cur_hi[KSN], nxt_lo[KSN], nxt_hi[KSN];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) nxt_lo[ks] = nxt_hi[ks] = float4{0.f, 0.f, 0.f, 0.f};
        if (wave < KT) xload(wave, cur_lo, cur_hi);
        for (int kt = wave; kt < (ABL == 2 ? min(KT, NW) : KT); kt += NW) {
            if (kt + NW < KT) xload(kt + NW, nxt_lo, nxt_hi);
            f32x4 a = f4zero(), c = f4zero(), qa = f4zero();
            const bool isq = kt >= qt0 && kt < qt1;        // wave-uniform
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const bf16x8 xf = xconv(kt, cur_lo[ks], cur_hi[ks], ks);
                a = MFMA(wkf[ks], xf, a);
                c = MFMA(xf, wvf[ks], c);
                if (isq) qa = MFMA(wqf[ks], xf, qa);
            }
            *reinterpret_cast<u32x2*>(kbf + krow(kt, tok, g)) = u32x2{cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3])};
            char* dst = vbf + ((size_t)((kt >> 1) * 4 + g) * 16 + tok) * 16;
            if (!(kmax_in_v && kt == 0 && lane == 7))          // (that row's first 8 bytes hold kmax)
                *reinterpret_cast<u32x2*>(dst + 8 * (kt & 1)) = u32x2{cvt_pk_bf16(c[0], c[1]), cvt_pk_bf16(c[2], c[3])};
            if ((KT & 1) && kt == KT - 1 && KTP == KT) *reinterpret_cast<u32x2*>(dst + 8) = u32x2{0u, 0u};
            if (isq)
                *reinterpret_cast<u32x2*>(qbf + ((size_t)(kt - qt0) * 64 + lane) * 8) = u32x2{cvt_pk_bf16(qa[0], qa[1]), cvt_pk_bf16(qa[2], qa[3])};
            float n2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
            float ea, eb;
            swap16(n2, ea, eb);
            n2 = row_max16(ea + eb);
            if (tok == 0 && (g & 1) == 0)
                __hip_atomic_fetch_max(&kmax[g >> 1], __builtin_bit_cast(unsigned, n2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                cur_lo[ks] = nxt_lo[ks];
                cur_hi[ks] = nxt_hi[ks];
            }
        }
        }
        // padding tiles: zero K rows and V^T halves (scores 0 -> exp2 = 1 against zero V^T / ones-row entries: no contribution)
        for (int kt = KT + wave; kt < KTP; kt += NW) {
            *reinterpret_cast<u32x2*>(kbf + krow(kt, tok, g)) = u32x2{0u, 0u};
            *reinterpret_cast<u32x2*>(vbf + ((size_t)((kt >> 1) * 4 + g) * 16 + tok) * 16 + 8 * (kt & 1)) = u32x2{0u, 0u};
        }
    } else {
    // K: one thread per (token, lane group gq): the 4 dims 4(gq&1)..+3 of head gq>>1 -> one 8-byte row; dim slot hd
    // carries the constant 1.0 the fast path's shift rides on (Q gets -bound there)
    for (int i = threadIdx.x; i < NTOK * 4; i += NTH) {
        const int t = i >> 2, gq = i & 3, head = 2 * pair + (gq >> 1);
        float kv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 4 * (gq & 1) + r;
            kv[r] = (t < T && head < H) ? (d < hd ? base[(size_t)t * 3 * D + D + head * hd + d] : (d == hd ? 1.0f : 0.f)) : 0.f;
        }
        *reinterpret_cast<u32x2*>(kbf + krow(t >> 4, t & 15, gq)) = u32x2{cvt_pk_bf16(kv[0], kv[1]), cvt_pk_bf16(kv[2], kv[3])};
        // |k_t|^2 of head gq>>1 (incl. the constant slot: the bound only grows): the two threads (gq even / odd) holding a
        // token's halves are lane neighbours
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
        if (kmax_in_v && i == 7)                             // (the row's first 8 bytes hold kmax)
            *reinterpret_cast<u32x2*>(vbf + (size_t)i * 16 + 8) = u32x2{cvt_pk_bf16(vv[4], vv[5]), cvt_pk_bf16(vv[6], vv[7])};
        else
            *reinterpret_cast<u32x4*>(vbf + (size_t)i * 16) = u32x4{cvt_pk_bf16(vv[0], vv[1]), cvt_pk_bf16(vv[2], vv[3]),
                                                                   cvt_pk_bf16(vv[4], vv[5]), cvt_pk_bf16(vv[6], vv[7])};
    }
    // Q of this slice's tiles: lane (tok, gq) of tile qt holds dims 4(gq&1)..+3 of head gq>>1, scaled by log2(e)/sqrt(hd)
    for (int i = threadIdx.x; i < (qt1 - qt0) * 64; i += NTH) {
        const int ln = i & 63, tl = i >> 6, t = (qt0 + tl) * 16 + (ln & 15), gq = ln >> 4, head = 2 * pair + (gq >> 1);
        float qv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 4 * (gq & 1) + r;
            qv[r] = (t < T && d < hd && head < H) ? base[(size_t)t * 3 * D + head * hd + d] * qscale : 0.f;
        }
        *reinterpret_cast<u32x2*>(qbf + (size_t)i * 8) = u32x2{cvt_pk_bf16(qv[0], qv[1]), cvt_pk_bf16(qv[2], qv[3])};
    }
    }
    ATTN_STAMP(0, tprev);
    __syncthreads();
    ATTN_STAMP(1, tprev);

    const bool lo_grp = (g >> 1) == 0;
    const int myhead = 2 * pair + (g >> 1);
    // keys beyond T in the ragged last tile.  Recomputed at its few uses (four compares) from a laundered copy of the lane
    // group: as a kernel-long f32x4 it was four of the 14 dwords hipcc spilled at the 128-VGPR budget (scripts/scratch_report.py
    // lists the kernels that touch scratch memory; without it this one went 106.1 -> 104.9 us per layer at T = 1024).
    auto cmask_now = [&]() {
        int gl = g;
        asm volatile("" : "+v"(gl));
        f32x4 m;
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = ((KT - 1) * 16 + 4 * gl + r >= T) ? kNegBig : 0.f;
        return m;
    };
#define cmask cmask_now()
    // The same reads for the code off the steady-state pipeline (partial last block, exact path): the lane's address terms
    // go through an opaque move, so that hipcc builds these addresses where they are used instead of keeping one VGPR per
    // (call site, loop start) alive across the whole unit loop -- those were the rest of the spilled dwords.
    auto kfrag_cold = [&](int kt) {
        int tl = tok, gl = g;
        asm volatile("" : "+v"(tl), "+v"(gl));
        return *reinterpret_cast<const s16x4*>(kbf + krow(kt, tl, gl));
    };
    auto vfrag_cold = [&](int jb) {
        int tl = tok, gl = g;
        asm volatile("" : "+v"(tl), "+v"(gl));
        return *reinterpret_cast<const bf16x8*>(vbf + ((size_t)(jb * 4 + gl) * 16 + tl) * 16);
    };
    bool prefer_exact = exact_only != 0;                              // wave-uniform
#ifdef FD_ATTN_PRIO_YOUNG        // experiment: the second wave of every SIMD (it loses every arbitration against the first) runs its units preferred
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(FD_ATTN_PRIO_YOUNG);
#endif
    for (int du = du0 + wave; du < du1; du += NW) {
        // the lane's fragment addresses are rebuilt per unit (a few VALU) from opaque copies of (tok, g): hoisted out of this
        // loop they were kept alive -- i.e. spilled -- across it, one VGPR per address stream of the key pipeline
        int tokd = tok, gd = g;
        asm volatile("" : "+v"(tokd), "+v"(gd));
        auto kfrag = [&](int kt) { return *reinterpret_cast<const s16x4*>(kbf + krow(kt, tokd, gd)); };
        auto vfrag = [&](int jb) { return *reinterpret_cast<const bf16x8*>(vbf + ((size_t)(jb * 4 + gd) * 16 + tokd) * 16); };
        int qt[NQ];
        bool qv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            qv[q] = du * NQ + q < KT;
            qt[q] = qv[q] ? du * NQ + q : du * NQ;
        }
        // Q^T B operands: lane (query tok, g) holds dims 4(g&1)..+3 of head g>>1 (other head's k-slots zero: the K fragment
        // then serves both heads unmodified)
        u32x2 qraw[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) qraw[q] = *reinterpret_cast<const u32x2*>(qbf + ((size_t)(qt[q] - qt0) * 64 + lane) * 8);
        auto qmasked = [&](int q, int hs) -> s16x4 {
            const bool mine = lo_grp == (hs == 0);
            const u32x2 w = {mine ? qraw[q][0] : 0u, mine ? qraw[q][1] : 0u};
            return __builtin_bit_cast(s16x4, w);
        };
        // Softmax shift: the bound |q| max_j |k_j| >= max_j q.k_j (2 % headroom for the bf16 rounding) instead of a
        // max pass; any shift cancels in P V / sum P.  A row sum below 2^-100 (bound > max + ~100) sends the unit
        // through the exact two-pass form (see fd_mega.hip, tests/test_gpu_baseline_shapes.py).
        float bq[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float q0 = __builtin_bit_cast(float, qraw[q][0] << 16), q1 = __builtin_bit_cast(float, qraw[q][0] & 0xffff0000u);
            const float q2 = __builtin_bit_cast(float, qraw[q][1] << 16), q3 = __builtin_bit_cast(float, qraw[q][1] & 0xffff0000u);
            float ea, eb;
            swap16(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3, ea, eb);
            const float k2 = __builtin_bit_cast(float, kmax[g >> 1]);
            swap32(__builtin_sqrtf((ea + eb) * k2) * 1.02f, bq[q][0], bq[q][1]);
        }
        ATTN_STAMP(2, tprev);
        float m2[NQ][2];
        f32x4 o2[NQ][2];

        // ------------------------------------------------------------------ fast path: shift = bound, one pipeline
        auto run_fast = [&]() {
            // Q with -bound in k-slot hd of its head: K carries a 1.0 there (bias row of the W_k image / staged constant),
            // so the shift comes out of the contraction and the MFMA's C operand is an inline 0
            s16x4 qs[NQ][2];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) {
                    u32x2 w = __builtin_bit_cast(u32x2, qmasked(q, hs));
                    const unsigned nb = cvt_pk_bf16(-bq[q][hs], 0.f) & 0xffffu;
                    const bool mine = g == 2 * hs + (hd >> 2);
                    const int dw = (hd & 3) >> 1;
                    const unsigned old = dw ? w[1] : w[0];
                    const unsigned patched = (hd & 1) ? ((old & 0x0000ffffu) | (nb << 16)) : ((old & 0xffff0000u) | nb);
                    if (dw) w[1] = mine ? patched : old;
                    else w[0] = mine ? patched : old;
                    qs[q][hs] = __builtin_bit_cast(s16x4, w);
                }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) o2[q][hs] = f4zero();
            // Tile order inside a 128-key block: k = ((jj * 2 + hs) * NQ + q) * 2 + jl, key tile 2jj + jl
            constexpr int NKT = 16 * NQ, LAG = FD_ATTN_LAG, HL = LAG / 2;
            static_assert(NQ == 2 && LAG % 2 == 0 && LAG >= 2 && LAG <= 4, "pipeline carry assumes pairs of tiles, last group only");
            bf16x8 vprev;
            f32x4 cpe[LAG];          // carried over the block boundary: scores of the block's last LAG tiles (before exp2),
            bf16x8 cpk[HL];          // its packed pairs not yet multiplied, and (vprev) its last V^T fragment
            const int NFULL = KTP >> 3;
            if (NFULL > 0) {
                // K / V^T fragments are read from LDS one 32-key group (8 slots, ~400 cycles) ahead of their use: two groups of K
                // and three of V^T are live at a time instead of a whole block's (the kernel must stay under 128 VGPRs: two
                // workgroups per CU, so that one stages while the other is in its key loop).  kg / vg = group 0 of the block
                // about to run.
                s16x4 kg[2] = {kfrag(0), kfrag(1)};
                bf16x8 vg = vfrag(0);
#pragma unroll
                for (int i = 0; i < LAG; ++i) cpe[i] = f32x4{kNegBig, kNegBig, kNegBig, kNegBig};   // exp2 -> 0: the first block's
#pragma unroll
                for (int i = 0; i < HL; ++i) cpk[i] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});   // "previous" tiles add 0
                vprev = vg;
                auto body = [&](int fb, int nb) {
                    s16x4 kk[5][2];          // kk[jj] / vv[jj]: group jj of this block; index 4 = group 0 of block nb
                    bf16x8 vv[5];
                    kk[0][0] = kg[0];
                    kk[0][1] = kg[1];
                    vv[0] = vg;
                    f32x4 pe[NKT + LAG];
                    bf16x8 pk[NKT / 2 + LAG];
#pragma unroll
                    for (int i = 0; i < LAG; ++i) pe[i] = cpe[i];
#pragma unroll
                    for (int i = 0; i < HL; ++i) pk[i] = cpk[i];
                    const bf16x8 vold = vprev;
#pragma unroll
                    for (int k = 0; k < NKT; ++k) {
                        if ((k & 7) == 0) {     // group start: issue the next group's reads
                            const int jn = (k >> 3) + 1;
                            const int kt = (jn < 4) ? fb * 8 + 2 * jn : nb * 8, jb = (jn < 4) ? fb * 4 + jn : nb * 4;
                            kk[jn][0] = kfrag(kt);
                            kk[jn][1] = kfrag(kt + 1);
                            vv[jn] = vfrag(jb);
                        }
                        {   // stage A: scores of tile k
                            const int jl = k & 1, q = (k >> 1) & 1, hs = (k >> 2) & 1, jj = k >> 3;
                            pe[LAG + k] = MFMA16(kk[jj][jl], qs[q][hs], f4zero());
                        }
                        {   // stage B: exp2 + pack of the tile LAG slots back (array slot k)
#pragma unroll
                            for (int r = 0; r < 4; ++r) pe[k][r] = __builtin_amdgcn_exp2f(pe[k][r]);
                            if (k & 1) pk[(k >> 1) + HL] = pack8(pe[k - 1], pe[k]);
                        }
                        if (k & 1) {   // stage C: P V of the pair 2 LAG slots back (array slot k >> 1)
                            const int a = k >> 1, p = a - LAG;            // p < 0: pair 16 + p of the previous block (jj = 3)
                            const int pp = p < 0 ? 16 + p : p;
                            const int q = pp & 1, hs = (pp >> 1) & 1, jj = pp >> 2;
                            o2[q][hs] = MFMA(p < 0 ? vold : vv[jj], pk[a], o2[q][hs]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    vprev = vv[3];
                    kg[0] = kk[4][0];
                    kg[1] = kk[4][1];
                    vg = vv[4];
#pragma unroll
                    for (int i = 0; i < LAG; ++i) cpe[i] = pe[NKT + i];
#pragma unroll
                    for (int i = 0; i < HL; ++i) cpk[i] = pk[NKT / 2 + i];
                };
                int fb = 0;
                for (; fb + 1 < NFULL; fb += 2) {       // (two blocks per trip: half the register moves at the back edge)
                    body(fb, fb + 1);
                    body(fb + 1, min(fb + 2, NFULL - 1));
                }
                if (fb < NFULL) body(fb, fb);
                // drain: the last block's final tiles
                {
                    f32x4 pe[LAG];
                    bf16x8 pk[LAG];
#pragma unroll
                    for (int i = 0; i < LAG; ++i) pe[i] = cpe[i];
#pragma unroll
                    for (int i = 0; i < HL; ++i) pk[i] = cpk[i];
#pragma unroll
                    for (int k = 0; k < LAG; ++k) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pe[k][r] = __builtin_amdgcn_exp2f(pe[k][r]);
                        if (k & 1) pk[(k >> 1) + HL] = pack8(pe[k - 1], pe[k]);
                    }
#pragma unroll
                    for (int a = 0; a < LAG; ++a) {
                        const int pp = 16 - LAG + a;
                        o2[pp & 1][(pp >> 1) & 1] = MFMA(vprev, pk[a], o2[pp & 1][(pp >> 1) & 1]);
                    }
                }
            }
            // The series' last, partial key block (1-7 tiles): a rolled loop over key-tile PAIRS with a static body.  Keys
            // beyond T need no mask here: their K rows, V rows and ones-row entries are all 0 (exp2(0) * 0).
            for (int jb = NFULL * 4; jb < NJ; ++jb) {
                const int ka = 2 * jb, kb2 = (2 * jb + 1 < KT) ? 2 * jb + 1 : ka;
                const s16x4 kfa = kfrag_cold(ka), kfb = kfrag_cold(kb2);
                const bf16x8 vfj = vfrag_cold(jb);      // (a missing odd tile: its V^T half was staged as zeros)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int hs = 0; hs < 2; ++hs) {
                        f32x4 pa = MFMA16(kfa, qs[q][hs], f4zero());
                        f32x4 pb = MFMA16(kfb, qs[q][hs], f4zero());
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pa[r] = __builtin_amdgcn_exp2f(pa[r]);
                            pb[r] = __builtin_amdgcn_exp2f(pb[r]);
                        }
                        o2[q][hs] = MFMA(vfj, pack8(pa, pb), o2[q][hs]);
                    }
            }
        };

        // ------------------------------------------------------------------ exact path: two passes per 128-key block
        auto run_exact = [&]() {
        s16x4 qb[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                qb[q][hs] = qmasked(q, hs);
                m2[q][hs] = kNegBig;
                o2[q][hs] = f4zero();
            }
        // One 128-key block.  FULL (8 key tiles) and LAST (the block holds the series' final, possibly ragged tile) are
        // compile-time: with run-time tile guards every MFMA sits in its own basic block and the hand-made software
        // pipeline below falls apart (measured 3x slower).
        auto key_block = [&](int kb, auto last_c) {
            constexpr bool LAST = decltype(last_c)::value;
            s16x4 kf[8];
            bf16x8 vf[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) kf[j] = kfrag_cold(kb + j);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) vf[jj] = vfrag_cold((kb >> 1) + jj);
            constexpr int NKT = 16 * NQ;          // score tiles per block: k = ((hs*4 + jj)*NQ + q)*2 + jl, key tile 2jj+jl
            // pass 1: row maxima of this key block (software-pipelined by hand, see fd_mega.hip)
            float bm[NQ][2];
#pragma unroll
            for (int q = 0; q < NQ; ++q) bm[q][0] = bm[q][1] = kNegBig;
            {
                constexpr int LAG = 3;
                f32x4 t4[NKT];
#pragma unroll
                for (int k = 0; k < NKT + LAG; ++k) {
                    if (k < NKT) {
                        const int jl = k & 1, q = (k >> 1) % NQ, jj = ((k >> 1) / NQ) & 3, hs = (k >> 1) / NQ >> 2;
                        const int j = 2 * jj + jl;
                        t4[k] = MFMA16(kf[j], qb[q][hs], (LAST && j == 7) ? cmask : f4zero());
                    }
                    if (k >= LAG) {
                        const int e = k - LAG;
                        const int q = (e >> 1) % NQ, hs = (e >> 1) / NQ >> 2;
                        const f32x4 v = t4[e];
                        float bb = bm[q][hs];
                        bb = fmaxf(fmaxf(bb, v[0]), v[1]);
                        bb = fmaxf(fmaxf(bb, v[2]), v[3]);
                        bm[q][hs] = bb;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            f32x4 negm[NQ][2], clast[NQ][2];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) {
                    const float mnew = fmaxf(m2[q][hs], group_max(bm[q][hs]));
                    const float alpha = __builtin_amdgcn_exp2f(m2[q][hs] - mnew);
                    o2[q][hs] = o2[q][hs] * alpha;
                    m2[q][hs] = mnew;
                    negm[q][hs] = f32x4{-mnew, -mnew, -mnew, -mnew};
                    clast[q][hs] = cmask - mnew;
                }
            // pass 2: P = exp2(S - max) (-max rides in the MFMA's C operand), packed to bf16 B fragments, then P V
            {
                constexpr int LAG = 2;
                f32x4 pe[NKT];
                bf16x8 pk[NKT / 2];
#pragma unroll
                for (int k = 0; k < NKT + 2 * LAG; ++k) {
                    if (k < NKT) {
                        const int jl = k & 1, q = (k >> 1) % NQ, jj = ((k >> 1) / NQ) & 3, hs = (k >> 1) / NQ >> 2;
                        const int j = 2 * jj + jl;
                        pe[k] = MFMA16(kf[j], qb[q][hs], (LAST && j == 7) ? clast[q][hs] : negm[q][hs]);
                    }
                    if (k >= LAG && k - LAG < NKT) {
                        const int e = k - LAG;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pe[e][r] = __builtin_amdgcn_exp2f(pe[e][r]);
                        if (e & 1) pk[e >> 1] = pack8(pe[e - 1], pe[e]);
                    }
                    if (k >= 2 * LAG && ((k - 2 * LAG) & 1)) {
                        const int e = k - 2 * LAG;
                        const int q = (e >> 1) % NQ, jj = ((e >> 1) / NQ) & 3, hs = (e >> 1) / NQ >> 2;
                        o2[q][hs] = MFMA(vf[jj], pk[e >> 1], o2[q][hs]);
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
            f32x4 negm[NQ][2];
            float bm[NQ][2];
#pragma unroll
            for (int q = 0; q < NQ; ++q) bm[q][0] = bm[q][1] = kNegBig;
            for (int kt = kb; kt < KT; ++kt) {
                const s16x4 kfa = kfrag_cold(kt);
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
                    negm[q][hs] = f32x4{-mnew, -mnew, -mnew, -mnew};
                }
            for (int jb = kb >> 1; jb < ((KT + 1) >> 1); ++jb) {          // (the series' own tiles only: no padding tiles here)
                const int ka = 2 * jb, kb2 = (2 * jb + 1 < KT) ? 2 * jb + 1 : ka;
                const s16x4 kfa = kfrag_cold(ka), kfb = kfrag_cold(kb2);
                const bf16x8 vfj = vfrag_cold(jb);
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
            for (; kb + 8 < KT; kb += 8) key_block(kb, std::false_type{});
            if (kb + 8 == KT) key_block(kb, std::true_type{});
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
        // A unit whose fast pass underflowed has paid for both forms (2.5 x).  Units of one wave belong to one (series, head
        // pair) and neighbouring query tiles, so after the first failure the wave goes straight to the exact form for its
        // remaining units (1.5 x): a sampler trajectory that leaves the data scale (random-init weights: |x| grows ~150 x
        // along the VP reverse SDE) made every layer-0 unit fail from the fifth step on, 186 us instead of 97 at T = 1024.
        // ... and a unit whose bound is beyond 2048 octaves does not try the fast pass at all: it succeeds only if EVERY row's true
        // maximum lies within ~100 octaves = 5 % of its bound, and such rows are softmaxes over scores of thousands (layer 0 of
        // the same trajectories: with two units per wave at T = 1024 the wave-sticky switch alone still paid 2.5 x + 1.5 x).
        bool huge = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q) huge |= (bq[q][0] > 2048.f) | (bq[q][1] > 2048.f);
        if (prefer_exact || (__builtin_amdgcn_ballot_w64(huge) != 0ull && ABL != 2)) {
            run_exact();
            (void)row_sums();
        } else {
            run_fast();
            if (__builtin_amdgcn_ballot_w64(row_sums()) != 0ull && ABL != 2) {
                run_exact();
                (void)row_sums();
                prefer_exact = true;
            }
        }
        ATTN_STAMP(3, tprev);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float o_sel[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o_sel[r] = lo_grp ? o2[q][0][r] : o2[q][1][r];
            const float inv = 1.0f / lrow[q];
            const int t = qt[q] * 16 + tok;
            if (out_bf16) {
                // bf16 rows (M, D) for k_ffn_ln's fused out-projection prologue, which rounds the attention output to bf16 MFMA
                // operands anyway: the same round-to-nearest-even here, half the bytes there (head_dim even: dword-aligned words)
                typedef unsigned u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
                if (qv[q] && t < T && myhead < H) {
                    __bf16* orow = reinterpret_cast<__bf16*>(out) + ((size_t)b * T + t) * D + myhead * hd + 4 * (g & 1);
                    const int nv = min(4, max(0, hd - 4 * (g & 1)));
                    const unsigned w0 = cvt_pk_bf16(o_sel[0] * inv, o_sel[1] * inv), w1 = cvt_pk_bf16(o_sel[2] * inv, o_sel[3] * inv);
                    if (nv == 4) *reinterpret_cast<u32x2_a4*>(orow) = u32x2_a4{w0, w1};
                    else if (nv == 2) *reinterpret_cast<unsigned*>(orow) = w0;
                }
            } else if (qv[q] && t < T && myhead < H) {
                // row form: the lane's dims as one 16- or 8-byte store at a dword-aligned address where head_dim allows
                typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
                typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
                float* orow = out + ((size_t)b * T + t) * D + myhead * hd + 4 * (g & 1);
                const int nv = min(4, max(0, hd - 4 * (g & 1)));
                if (nv == 4) *reinterpret_cast<f32x4_a4*>(orow) = f32x4_a4{o_sel[0] * inv, o_sel[1] * inv, o_sel[2] * inv, o_sel[3] * inv};
                else if (nv == 2) *reinterpret_cast<f32x2_a4*>(orow) = f32x2_a4{o_sel[0] * inv, o_sel[1] * inv};
                else {
                    if (nv > 0) orow[0] = o_sel[0] * inv;
                    if (nv > 1) orow[1] = o_sel[1] * inv;
                    if (nv > 2) orow[2] = o_sel[2] * inv;
                }
            }
        }
        ATTN_STAMP(4, tprev);
    }
}

}  // namespace

// bf16 MFMA attention.  wk/wv/wq == nullptr: `in` holds the packed projections (B*T, 3D); otherwise `in` is the layer
// input (B*T, D) and the images are the persistent kernel's per-layer W_k / W_v / W_q fragment blocks (ks1 blocks per
// head pair).  Returns FD_ERR_UNSUPPORTED when the shape does not fit (head_dim > 7, K/V^T of one series exceed the
// LDS, or no instantiation for ks1): the caller then runs the unfused / exact-f32 path.
// out_bf16: `out` receives bf16 rows (M, D) instead of fp32 ones (even head_dim only; the consumer is k_ffn_ln's fused prologue).
// in_rows (nullable; fused projections only): the layer input as bf16 rows (B*T, 32 ks1), see the kernel's ROWS form.
int fd_attention_bf16(fd_ctx* ctx, const float* in, float* out, int B, int T, int H, int hd, hipStream_t s, const char* wk,
                      const char* wv, const char* wq, int ks1, int out_bf16, const void* in_rows) {
    const int KT = (T + 15) / 16, D = H * hd;
    // A partial last 128-key block of >= FDIFF_ATTN_PAD_MIN (default 3) tiles is padded to a whole one with zero tiles when the LDS
    // allows: the fast path then has no rolled remainder loop (T = 365: 23 -> 24 tiles, 115.0 -> 112.2 us per layer at B = 512; profiles/r05_droughts_shape.txt).
    const int pad_min = getenv("FDIFF_ATTN_PAD_MIN") ? atoi(getenv("FDIFF_ATTN_PAD_MIN")) : 3;
    auto kv_bytes = [&](int ktp) { return (size_t)ktp * 16 * 32 + (size_t)((ktp + 1) / 2) * 1024 + (hd == 7 ? 16 : 0); };
    int KTP = KT;
    if ((KT & 7) >= pad_min && kv_bytes((KT + 7) & ~7) + NQ * 512 <= 160 * 1024) KTP = (KT + 7) & ~7;
    const size_t lds_kv = kv_bytes(KTP);
    if (hd > 7 || lds_kv + NQ * 512 > 160 * 1024) return FD_ERR_UNSUPPORTED;
    if (out_bf16 && (hd & 1)) return FD_ERR_UNSUPPORTED;
    const bool proj = wk != nullptr;
    if (proj && ((ks1 != 3 && ks1 != 2) || (D & 3))) return FD_ERR_UNSUPPORTED;   // (raw x rows are read as float4)
    const bool rows = proj && in_rows != nullptr;
    const void* kern = proj ? (ks1 == 3 ? (rows ? (const void*)k_attention_bf16<3, 1> : (const void*)k_attention_bf16<3>)
                                        : (rows ? (const void*)k_attention_bf16<2, 1> : (const void*)k_attention_bf16<2>))
                            : (const void*)k_attention_bf16<0>;
    static unsigned long long attr[5] = {};
    if (fd_first_on_device(attr[proj ? (ks1 == 3 ? 2 : 1) + (rows ? 2 : 0) : 0], ctx->device))
        FD_HIP(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int NP = (H + 1) / 2, DUS = (KT + NQ - 1) / NQ;
    // Query slices per (series, pair): every slice restages K/V (~16 % of a full slice's work) and keeps its own Q tiles in
    // LDS (512 B per tile); pick the count that minimises rounds x work per round among those that fit (two workgroups per
    // CU when one needs <= 80 KiB).
    const char* fs_env = getenv("FDIFF_ATTN_SLICES");        // (tests: force the query-slice count)
    const int force_slices = fs_env ? atoi(fs_env) : 0;
    int slices = 0;
    double best = 1e30;
    auto fits = [&](int sl) { return lds_kv + (size_t)((DUS + sl - 1) / sl) * NQ * 512 <= 160 * 1024; };
    for (int sl = 1; sl <= 16 && (sl == 1 || sl <= DUS); sl *= 2) {
        if (!fits(sl)) continue;
        if (slices && DUS / sl < NW) break;                  // at least one unit per wave, unless nothing coarser fits
        // time of a slice's workgroup ~ staging (the whole series: 0.3 of a full key loop, measured at T = 1024) + its share
        // of the key loop; workgroups that fit twice on a CU (<= 80 KiB, 128 VGPRs) overlap by ~12 % only, so the model
        // counts CU slots as one workgroup each
        const double rounds = (double)(((size_t)B * NP * sl + ctx->num_cu - 1) / ctx->num_cu);
        const double cost = rounds * (1.0 / sl + 0.3);
        if (cost < best - 1e-9) { best = cost; slices = sl; }
    }
    if (force_slices > 0 && force_slices <= DUS && fits(force_slices)) slices = force_slices;
    if (!slices) return FD_ERR_UNSUPPORTED;
    const int du_per_block = (DUS + slices - 1) / slices;
    const size_t lds = lds_kv + (size_t)du_per_block * NQ * 512;
    // Tail split.  B NP slices equal workgroups on `slots` resident ones run ceil(n / slots) rounds, and a last round that fills half of
    // the slots or less leaves the chip half empty for its whole length (T = 1024, B = 64: 768 workgroups, 512 slots).  The series
    // whose workgroups would form that round get twice as many, half as long slices instead, launched behind the others: the last
    // round then fills every slot with half-length workgroups.  MEASURED SLOWER and therefore OFF unless FDIFF_ATTN_TAIL_SPLIT=1
    // (same box, alternating: 1.347 -> 1.494 ms per diffusion step at T = 1024, B = 64; +-0 at the droughts shape; profiles/
    // r06_attn_long_tail_split_ab.txt): a workgroup that has its CU to itself runs its key loop almost twice as fast as two that
    // share one -- the half-empty round is NOT half-wasted -- and a third more workgroups restage K / V.
    int B1 = B, slices2 = slices, dpb2 = du_per_block;
    {
        static const bool tail_on = getenv("FDIFF_ATTN_TAIL_SPLIT") && atoi(getenv("FDIFF_ATTN_TAIL_SPLIT")) != 0;
        const long long slots = (long long)ctx->num_cu * (lds <= 80 * 1024 ? 2 : 1);
        const long long n = (long long)B * NP * slices, per_series = (long long)NP * slices;
        const long long full = (n / slots) * slots, rest = n - full;
        if (tail_on && full > 0 && rest > 0 && 2 * rest <= slots + per_series && DUS / (2 * slices) >= NW && fits(2 * slices)) {
            const int Bsplit = (int)((rest + per_series / 2) / per_series);
            if (Bsplit > 0 && Bsplit < B) {
                B1 = B - Bsplit; slices2 = 2 * slices; dpb2 = (DUS + slices2 - 1) / slices2;
            }
        }
    }
    const int nblk1 = ((B1 + 7) / 8) * 8 * NP * slices, nblk2 = B1 < B ? ((B - B1 + 7) / 8) * 8 * NP * slices2 : 0;
    const float qscale = 1.4426950408889634f / sqrtf((float)hd);
    const int exact = getenv("FDIFF_ATTN_EXACT") ? 1 : 0;
    const fd_attn_w w{wk, wv, wq};
    const size_t pair_stride = (size_t)ks1 * 1024;
    const dim3 grid((unsigned)(nblk1 + nblk2)), block(NTH);
    const __bf16* xr = reinterpret_cast<const __bf16*>(in_rows);
#define FD_ATT_GO(K, R) hipLaunchKernelGGL((k_attention_bf16<K, R>), grid, block, lds, s, in, out, T, H, hd, D, qscale, du_per_block, exact, w, pair_stride, slices, B, out_bf16, xr, KTP, nblk1, B1, slices2, dpb2)
    if (!proj) FD_ATT_GO(0, 0);
    else if (ks1 == 3) { if (rows) FD_ATT_GO(3, 1); else FD_ATT_GO(3, 0); }
    else { if (rows) FD_ATT_GO(2, 1); else FD_ATT_GO(2, 0); }
#undef FD_ATT_GO
    FD_LAUNCH_CHECK(ctx);
#if defined(FD_ATTN_ABL) && FD_ATTN_ABL == 3
    {
        static int calls = 0;
        if (++calls == 30) {
            unsigned long long h[64];
            hipStreamSynchronize(s);
            hipMemcpyFromSymbol(h, HIP_SYMBOL(fd_attn_dbg), sizeof(h));
            for (int w = 0; w < 8; ++w)
                fprintf(stderr, "[attn dbg] wave %d over %d launches: staging %llu, barrier %llu, q-setup %llu, key blocks %llu, epilogue %llu cycles (100 MHz ticks?)\n",
                        w, calls, h[w * 8 + 0] / calls, h[w * 8 + 1] / calls, h[w * 8 + 2] / calls, h[w * 8 + 3] / calls, h[w * 8 + 4] / calls);
        }
    }
#endif
    return FD_OK;
}
