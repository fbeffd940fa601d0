// fd_mega_kernel.h -- DEVICE code of the persistent series-resident kernel (k_mega and its helpers), kept in a header because it is
// compiled twice: ahead of time by hipcc (fd_mega.hip: the BASELINE shapes and the run-time-shape instantiations) and at run time
// by hiprtc (fd_mega_rtc.hip: one ShapeStatic instantiation per (model, series shape, workgroup plan) a user actually runs,
// cached on disk).  Nothing here may touch host-only headers: hiprtc sees this file, fd_mega_params.h and fd_philox.h only.
#pragma once
// the whole reverse-diffusion loop) for a group of series runs inside ONE workgroup with no inter-workgroup
// communication; activations never leave the CU.
//
// Why this shape (measured on MI355X, profiles/ r01): as separate launches the bf16 FFN kernel spends 40 % of
// its 44 us on prologue/epilogue HBM traffic and ~9 us on launch + cold weight stream, and a diffusion step
// would need >= 20 launches.  A series is tiny (T*72 fp32 = 29 KB at T=100), so a workgroup keeps
//   * the fp32 residual stream of its tokens in REGISTERS (each 16-token tile is owned by one wave),
//   * bf16 MFMA B-fragments of the current activations, K, V^T and the streamed weights in LDS (<=160 KiB),
// and loops over layers and diffusion steps; only x (B,T,C) and the L2-resident weight images are touched
// in global memory.  One workgroup = 8 waves = 2 per SIMD, S series (S*ceil16(T) <= 256 token slots).
//
// Operand convention (v_mfma_f32_16x16x32_bf16): token on lane&15, g = lane>>4 selects 8 k-slots; every
// GEMM is computed transposed (out^T = W . x^T) so C tiles are [feature = 4g+r][token = lane&15] and chain
// into the next GEMM's B operand with k-permuted weight images -- see fd_score_bf16.hip for the FFN case.
// Attention: units of (head pair) x (series) x (two query tiles); S^T = K Q^T by the K=16 MFMA with two heads sharing
// every K / V fragment (even head in k-slots / rows of lane groups 0-1, odd head in groups 2-3); the softmax shift
// (bound |q| max|k|, exact row maximum as fallback) rides in the MFMA's C operand, exp2 with the scale folded into W_q,
// the denominator comes out of the P V MFMAs through a row of ones in V^T; P is fed back as a B operand without
// leaving registers.  FFN: weights streamed L2 -> LDS through a 4-deep ring, hidden activations stay in registers.
// DESIGN.md sections 3.2 / 3.3 hold the measurements behind each of these choices.
//
// Reference arithmetic: src/fdiff/models/score_models.py:67-94 (+ torch TransformerEncoderLayer),
// src/fdiff/sampling/sampler.py:83-104, src/fdiff/schedulers/sde.py:129-165,215-246.
#include <type_traits>

#include "fd_mega_params.h"
#include "fd_philox.h"

// hipcc: internal linkage (one translation unit owns the instantiations).  hiprtc: a named namespace, so that the instantiation's
// name expression resolves to an external kernel symbol.
#ifdef __HIPCC_RTC__
#define FD_MEGA_NS_BEGIN namespace fdmega {
#define FD_MEGA_NS_END }
#else
#define FD_MEGA_NS_BEGIN namespace {
#define FD_MEGA_NS_END }
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#ifdef FD_ABL_NOEXP   // ablation build: a full-rate VALU op in place of the half-rate transcendental (wrong results)
#define FD_EXP2(x) ((x) * 1.0001f)
#else
#define FD_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

#ifndef FD_ABL_NODMA       // ablation build: FFN loop without its weight DMA (timing only, wrong results)
#define FD_DMA_ON true
#else
#define FD_DMA_ON false
#endif
#ifndef FD_LDS_DUMP        // debugging build: FDIFF_MEGA_DUMP=file dumps workgroup 0's LDS after layer 0's attention
#define FD_LDS_DUMP 0
#endif
#ifndef FD_NO_PROF
#define FD_NO_PROF 0
#endif
#ifndef FD_KV_TILE_MAJOR
#define FD_KV_TILE_MAJOR 1
#endif
#ifndef FD_STATIC_UNITS
#define FD_STATIC_UNITS 1
#endif
#ifndef FD_PROF_UNITS
#define FD_PROF_UNITS 0
#endif
#ifndef FD_ROLLED_ATTN
#define FD_ROLLED_ATTN 0
#endif
#ifndef FD_XF_REGS
#define FD_XF_REGS 1     // FFN activation fragments: 1 = registers for the whole phase, 0 = LDS read per use
#endif
#ifndef FD_DMA_LIGHT
#define FD_DMA_LIGHT 1   // FFN weight DMA issued by the waves with the fewest token tiles only (uneven splits)
#endif
#ifndef FD_FFN_PRIO
#define FD_FFN_PRIO 1    // the waves that carry one tile less (and issue the weight DMA) run the FFN loop at s_setprio 1
#endif
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define SG_VALU 0x2
#define SG_MFMA 0x8
#define SG_VMEM 0x10
#define SG_DSR 0x100

FD_MEGA_NS_BEGIN

constexpr float kNegBig = -1.0e30f;

__device__ __forceinline__ float relu_bits(float x) {
    int i = __builtin_bit_cast(int, x);
    i = i > 0 ? i : 0;
    return __builtin_bit_cast(float, i);
}
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    // vector fptrunc selects v_cvt_pk_bf16_f32 AND lets hipcc place the MFMA->VALU wait states itself
    // (an inline-asm cvt reading an MFMA result directly is not padded by the compiler: measured wrong data)
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
    u32x4 r = {cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3]), cvt_pk_bf16(b[0], b[1]), cvt_pk_bf16(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x8 relu_pack(f32x4 a, f32x4 b) {
    // convert first, then relu on the packed bf16 pairs: a negative bf16 is a negative int16, so one
    // v_pk_max_i16 against 0 clamps two values (8 VALU per 8 values instead of 12)
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_bit_cast(s16x8, pack8(a, b));
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(v, z));
}
// 32x32 C tile (hidden x tokens; this lane: token L % 32, rows 8 j + 4 (L / 32) + i in register 4 j + i) -> relu -> the two 16x16x32 B
// fragments of token tiles 0 / 1 of the pair.  lo = rows j in {0, 1}, hi = j in {2, 3}; v_permlane16_swap exchanges the odd 16-lane
// rows of `lo` with the even rows of `hi`: afterwards `lo` holds token tile 0 in all four lane rows and `hi` token tile 1, lane row q
// carrying hidden rows 16 (q & 1) + 8 (e >> 2) + 4 (q >> 1) + (e & 3) in k-slot e (the pair-form W2 image is k-permuted to match).
__device__ __forceinline__ void relu_split32(const f32x16& h, bf16x8& t0, bf16x8& t1) {
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    const s16x2 z = {0, 0};
    unsigned lo[4], hi[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        lo[d] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, cvt_pk_bf16(h[2 * d], h[2 * d + 1])), z));
        hi[d] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, cvt_pk_bf16(h[8 + 2 * d], h[8 + 2 * d + 1])), z));
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const u32x2 r = __builtin_amdgcn_permlane16_swap(lo[d], hi[d], false, false);
        lo[d] = r.x;
        hi[d] = r.y;
    }
    t0 = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
    t1 = __builtin_bit_cast(bf16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
}
__device__ __forceinline__ bf16x8 frag_zero() {
    u32x4 z = {0u, 0u, 0u, 0u};
    return __builtin_bit_cast(bf16x8, z);
}
__device__ __forceinline__ f32x4 f4zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
// Exchanges between the 4 lane groups (rows of 16 lanes) that hold one token: gfx950's row-swap instructions run at
// VALU latency; ds_bpermute (what __shfl_xor lowers to) is an LDS round trip of >100 cycles on the critical path
// of every softmax / LayerNorm.  swap16: a = value of the even row of each row pair, b = of the odd row.
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
// max over the 16 lanes of a row (DPP row rotations: VALU latency, no LDS)
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
__device__ __forceinline__ float group_sum(float v) {      // sum over the 4 lane groups holding one token
    float a, b;
    swap32(v, a, b);
    swap16(a + b, a, b);
    return a + b;
}
__device__ __forceinline__ float group_max(float v) {
    float a, b;
    swap32(v, a, b);
    swap16(fmaxf(a, b), a, b);
    return fmaxf(a, b);
}

// Shape policies: ShapeDyn reads every dimension from the parameter block (any supported model/shape);
// a ShapeStatic instantiation bakes the dimensions of one workload in, which removes the index arithmetic,
// the runtime loop guards (each guard splits a basic block and stops hipcc interleaving independent MFMA /
// VALU chains) and ~40 live scalars -- PMC showed 6.5 VALU + 3.2 SALU instructions per MFMA in the generic build.
// A dimension is compile-time when the policy gives it a non-zero value (every real dimension is >= 1).
struct ShapeDyn {
    static constexpr int T = 0, KT = 0, D = 0, C = 0, H = 0, hd = 0, S = 0, NPG = 0, KSE = 0, CT = 0, rot = 0, L = 0, F = 0;
    static constexpr int FFN32 = 0;
};
// FFN32_ = 1: the FFN's H GEMM runs on PAIRS of token tiles by v_mfma_f32_32x32x16_bf16 (K = D + 1 padded to 16 DT = 80 instead of
// 32 KS1 = 96: 5 x 32 cycles per 32 tokens and 32 hidden units instead of 12 x 16); its B fragments are gathered from the tiles' 16x16x32
// fragments in LDS by a per-lane address map once per FFN phase (the layout every other phase reads stays), W2 stays in the 16x16x32 form.
template <int T_, int D_, int C_, int H_, int S_, int NPG_, int ROT_, int L_, int F_, int FFN32_ = 0>
struct ShapeStatic {
    static constexpr int FFN32 = FFN32_;
    static constexpr int T = T_, KT = (T_ + 15) / 16, D = D_, C = C_, H = H_, hd = D_ / H_, S = S_, NPG = NPG_;
    static constexpr int KSE = (C_ + 1 + 31) / 32, CT = (C_ + 15) / 16, rot = ROT_, L = L_, F = F_;
};
// Model dimensions fixed, series shape (T, C) and the workgroup plan (S, NPG, rot) read at run time: every dataset run
// with one transformer configuration shares this instantiation (the per-feature loops of the LayerNorm / projection
// phases are then straight-line code; only the attention's key-tile loops keep their run-time guards).
template <int D_, int H_, int L_, int F_>
struct ShapeModel {
    static constexpr int T = 0, KT = 0, D = D_, C = 0, H = H_, hd = D_ / H_, S = 0, NPG = 0, KSE = 0, CT = 0, rot = 0, L = L_, F = F_;
    static constexpr int FFN32 = 0;
};
#define SHP(name) (SH::name != 0 ? SH::name : P.name)

// time embedding of one t by one wave: Gaussian Fourier features (LDS scratch `emb`, D floats) then the dense layer -> out[D]
__device__ __forceinline__ void time_embed_wave(float tv, const float* __restrict__ params, long long tW, long long td_w,
                                                long long td_b, float* emb, float* out, int D, int lane) {
    const int half = (D + 1) / 2;
    for (int j = lane; j < D; j += 64) {
        const int jj = (j < half) ? j : j - half;
        const float ph = ((tv * params[tW + jj]) * 2.0f) * 3.14159274101257324f;
        emb[j] = (j < half) ? sinf(ph) : cosf(ph);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < D; d += 64) {
        float a = params[td_b + d];
        const float* w = params + td_w + (size_t)d * D;
        for (int j = 0; j < D; ++j) a = fmaf(w[j], emb[j], a);
        out[d] = a;
    }
}

__global__ __launch_bounds__(64) void k_temb_table(const float* __restrict__ params, long long tW, long long td_w, long long td_b,
                                                   const fd_sde_step_coef* __restrict__ steps, float* __restrict__ table, int D) {
    extern __shared__ float emb_sh[];
    time_embed_wave(steps[blockIdx.x].t, params, tW, td_w, td_b, emb_sh, table + (size_t)blockIdx.x * D, D, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// NW = waves per workgroup (8: one workgroup per CU, up to 16 token tiles).  A 4-wave form (two co-resident
// workgroups per CU, one series each) was measured: the younger workgroup of each CU loses issue arbitration and
// finishes 25 % later than the older one, 0.65 vs 0.565 ms per diffusion step -- only NW = 8 is instantiated.
template <int KS1, int DT, int KSO, int MT, class SH, int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_mega(const fd_mega_params P) {
    constexpr int KSX = KS1;                     // x-fragment blocks per token tile (the host's LDS plan; the pair form needs less)
    // head_dim 8 (the d_model 64 / 8 heads class, KSO = 2): a head's dims fill its 8 k-slots, so there is no free slot for the softmax
    // shift / the row of ones -- its units always run the exact two-pass form (the shift rides in the score MFMA's C operand) and
    // take the softmax denominators from one more P V-shaped MFMA per key block whose A operand is all ones (the matrix pipe has
    // slack in the units, the VALU has none)
    constexpr bool HD8 = (KS1 == 3 && DT == 5 && KSO == 2) || (KS1 == 2 && DT == 3 && KSO == 1);   // (d_model 64 / 8 heads, 32 / 4 heads)
    constexpr bool F32 = SH::FFN32 != 0;         // pair form of the FFN (see ShapeStatic)
    constexpr int KS32 = DT;                     // pair form: k-steps of 16 (D + 1 <= 16 DT)
    static_assert(!F32 || (NW == 8 && DT == 2 * KS1 - 1 &&
                           ((MT == 4 && SH::S * SH::KT >= 12 && ((SH::S * SH::KT) & 1) == 0) || (MT == 3 && SH::S * SH::KT == 12))),
                  "pair-form FFN: 8 waves, 3 or 4 token tiles per wave, an even tile count, K = 16 DT");
    constexpr int NBF = F32 ? KS32 + DT : 2 * KS1 + DT;   // FFN blocks per (F-half, 32-wide chunk)
    constexpr int NBUF = 4;                      // FFN weight ring: 4 buffers of one 32-wide chunk per F-half
    constexpr int WB1 = 2 * NBF * 1024;          // bytes per ring buffer ([F-half][block])
    constexpr int NDMA = (2 * NBF + NW - 1) / NW;   // DMA instructions per wave per buffer (padded: uniform vmcnt)
    constexpr int MQ = NW / 2;                   // token-tile shares ("quarters" when NW = 8)
    constexpr int NTH = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // lane / tok / g are re-derived from an opaque (asm volatile) lane id at every phase boundary: otherwise
    // hipcc hoists every lane-dependent address of every phase out of the step/layer loops, keeps them live
    // across the FFN loop and spills them INSIDE it (a scratch reload forces s_waitcnt vmcnt(0), which also
    // drains the in-flight weight DMA and serialises stream and compute -- measured).
    int lane, tok, g;
    auto refresh_lane = [&]() {
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        lane = (int)l;
        tok = lane & 15;
        g = lane >> 4;
    };
    refresh_lane();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef FD_PRIO_YOUNG
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(FD_PRIO_YOUNG);   // experiment: static priority for the younger half
#endif
    const int T = SHP(T), KT = SHP(KT), D = SHP(D), C = SHP(C), H = SHP(H), hd = SHP(hd), S = SHP(S);
    const int NTILE = S * KT;                    // token tiles of this workgroup (16 slots each)
    const int NTOK = NTILE * 16;
    const int NP = (H + 1) >> 1;                 // head pairs
    const int NPG = SHP(NPG);                    // head pairs per attention group
    // W_o through LDS for the out-proj: it fits the (dead) W_k | W_v slots of the last attention group, and the FFN ring
    // can start one buffer later (buffers 1-2 = the first 3/4 of the ring must then lie in front of afr, which the
    // out-proj still reads; the host plan only guarantees the first half)
    const bool WO_LDS = (2 * NPG * KS1 >= DT * KSO) && (P.lds_afr - NTILE * KSX * 1024 >= 3 * WB1);
    const int rb = WO_LDS ? 1 : 0;               // FFN step s lives in ring buffer (s + rb) % NBUF
    const int NJ = (KT + 1) >> 1;                // 32-key blocks per series
    const int b0 = blockIdx.x * S;               // first series of this workgroup

    // ---- LDS map
    char* const xfr = smem;                                   // [NTILE][KSX][64][16 B]  activation B fragments
    char* const wsl = xfr + NTILE * KSX * 1024;               // [3][NPG][KS1][1 KiB]     W_k | W_v | W_q of the group
    char* const kbf = wsl + 3 * NPG * KS1 * 1024;             // [NPG][NTOK][4][8 B]      K (both heads per pair)
    char* const vbf = kbf + NPG * NTOK * 32;                  // [NPG][S][NJ][4][16][16 B] V^T
    char* const afr = smem + P.lds_afr;                       // [NTILE][KSO][64][16 B]  attention-output fragments
    char* const ring = wsl;                                   // FFN weight ring + exchange alias W/K/V(/afr)
    float* const temb = reinterpret_cast<float*>(smem + P.lds_temb);   // [S][D] + emb scratch [S][D]
    float* const lpar = temb + ((2 * S * D + 3) & ~3);                 // [6][D] bo, b2, g1, b1, g2, b2 of the layer (nlp KiB by DMA)
    unsigned* const kmax2 = reinterpret_cast<unsigned*>(lpar + P.nlp * 256); // [2 parities][NPG][S][2] max_j |k_j|^2 per head (bits)
    unsigned* const ucnt = kmax2 + 4 * NPG * S;                        // next attention unit of the group (dynamic hand-out)

    // ---- token-tile ownership (same split as the FFN: quarters mq, F-halves fh; fh waves rotated)
    const int fh = wave / MQ;
    // the two 4-wave workgroups sharing a CU (blocks i and i + num_cu under in-order dispatch) mirror their split
    const int wgpar = (NW == 4) ? (int)((blockIdx.x / (unsigned)P.num_cu) & 1u) : 0;
    const int mq = (wave + fh * SHP(rot) + wgpar) % MQ;
    const int tbase = NTILE / MQ, trem = NTILE % MQ;
    const int ntile = tbase + (mq < trem ? 1 : 0);
    const int tile0 = mq * tbase + (mq < trem ? mq : trem);
    // owned tiles (residual stream lives in this wave's registers): tt = fh, fh + 2
    // fp32 residual stream of the owned tiles (tt = fh, fh + 2).  It stays in registers except during the FFN
    // loop, where it IS the initial value of the owner's accumulator tiles (out = res + b2 + W2 relu(..)), so the
    // hot loop carries no extra live registers.
    f32x4 res[2][DT];
    int prof_cnt = 0;
    auto mark = [&](int phase, int step) {
#if FD_NO_PROF
        return;
#endif
        if (P.prof && blockIdx.x == 0 && wave == 0 && (step < 4 || phase == 0) && prof_cnt < 3990) {
            const unsigned long long tm = __builtin_readcyclecounter();
            if (lane == 0) {
                P.prof[2 * prof_cnt] = (unsigned long long)phase;
                P.prof[2 * prof_cnt + 1] = tm;
            }
            ++prof_cnt;
        }
    };

    // marks inside the attention units: only in -DFD_PROF_UNITS=1 builds (they cost registers and branches in the
    // most register-bound code of the kernel)
    auto umark = [&](int phase, int step) {
#if FD_PROF_UNITS
        mark(phase, step);
#endif
    };
    auto tile_token = [&](int tile, int& ser, int& t, bool& valid) {
        ser = tile / KT;
        t = (tile - ser * KT) * 16 + tok;
        valid = (t < T) && (b0 + ser < P.B);
    };
    auto layer_ptr = [&](int l) -> const char* { return P.img_layers + (size_t)l * P.layer_stride; };

    // LayerNorm of two C-layout tiles (DT row tiles each) over the D features of token lane&15, in place; both tiles in one
    // basic block so that their reductions and LDS reads interleave
    auto layer_norm2 = [&](f32x4 (&va)[DT], f32x4 (&vb)[DT], const float* __restrict__ gamma, const float* __restrict__ beta) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            if (16 * dt + 4 * g < D) {
                sa += (va[dt][0] + va[dt][1]) + (va[dt][2] + va[dt][3]);
                sb += (vb[dt][0] + vb[dt][1]) + (vb[dt][2] + vb[dt][3]);
            }
        const float invD = 1.0f / (float)D;
        const float ma = group_sum(sa) * invD, mb = group_sum(sb) * invD;
        float qa = 0.f, qb = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            if (16 * dt + 4 * g < D) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ca = va[dt][r] - ma, cb = vb[dt][r] - mb;
                    qa += ca * ca;
                    qb += cb * cb;
                }
            }
        const float ra = __builtin_amdgcn_rsqf(group_sum(qa) * invD + 1e-5f), rb2 = __builtin_amdgcn_rsqf(group_sum(qb) * invD + 1e-5f);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            if (d0 < D) {
                const float4 gm = *reinterpret_cast<const float4*>(gamma + d0);
                const float4 bt = *reinterpret_cast<const float4*>(beta + d0);
                va[dt][0] = (va[dt][0] - ma) * ra * gm.x + bt.x;
                va[dt][1] = (va[dt][1] - ma) * ra * gm.y + bt.y;
                va[dt][2] = (va[dt][2] - ma) * ra * gm.z + bt.z;
                va[dt][3] = (va[dt][3] - ma) * ra * gm.w + bt.w;
                vb[dt][0] = (vb[dt][0] - mb) * rb2 * gm.x + bt.x;
                vb[dt][1] = (vb[dt][1] - mb) * rb2 * gm.y + bt.y;
                vb[dt][2] = (vb[dt][2] - mb) * rb2 * gm.z + bt.z;
                vb[dt][3] = (vb[dt][3] - mb) * rb2 * gm.w + bt.w;
            } else {
                va[dt] = f4zero();
                vb[dt] = f4zero();
            }
        }
    };

    // residual (C layout) -> bf16 B fragments of `tile` in LDS; slot D carries the constant 1.0 (bias row)
    auto write_xfrags = [&](int tile, const f32x4 (&v)[DT]) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            u32x2 pk;
            if (d0 < D) {
                pk[0] = cvt_pk_bf16(v[dt][0], v[dt][1]);
                pk[1] = cvt_pk_bf16(v[dt][2], v[dt][3]);
            } else {
                pk[0] = (d0 == D) ? 0x00003F80u : 0u;      // bf16(1.0) in the low half
                pk[1] = 0u;
            }
            const int ks = dt >> 1, gd = 2 * (dt & 1) + (g >> 1);
            *reinterpret_cast<u32x2*>(xfr + ((tile * KSX + ks) * 64 + gd * 16 + tok) * 16 + 8 * (g & 1)) = pk;
        }
    };
    auto xfrag = [&](int tile, int ks) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(xfr + ((tile * KSX + ks) * 64 + lane) * 16);
    };
    auto gfrag = [&](const char* img, int blk) -> bf16x8 {   // fragment block straight from L2
        return *reinterpret_cast<const bf16x8*>(img + ((size_t)blk * 64 + lane) * 16);
    };
    auto dma_blocks = [&](const char* src, char* dst, int nblk) {   // nblk KiB, dealt round-robin to the 8 waves
        for (int b = wave; b < nblk; b += NW)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + ((size_t)b * 64 + lane) * 16), LDS_PTR(dst + b * 1024), 16,
                                             0, 0);
    };

    // ---- zero the fragment region once (k padding beyond the written slots must read as 0)
    for (int i = threadIdx.x; i < NTILE * KSX * 64; i += NTH) reinterpret_cast<u32x4*>(xfr)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    const int nsteps = (P.mode == FD_MEGA_SAMPLE) ? P.nsteps : 1;
    if (P.clk_out && blockIdx.x == 0 && wave == 0 && lane == 0) {     // (stored at once: nothing stays live across the kernel)
        P.clk_out[0] = __builtin_readcyclecounter();
        P.clk_out[1] = wall_clock64();
    }
    if (P.prof && wave == 0 && lane == 0 && blockIdx.x < 2048) {     // residency trace: start time + hardware id
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        P.prof[2 * (4100 + blockIdx.x)] = wall_clock64();
        P.prof[2 * (4100 + 2048 + blockIdx.x)] = ((unsigned long long)xcc << 32) | hwid;
    }
    // Layers whose softmax bound failed for this wave (bit l): their units go straight to the exact two-pass form until the
    // next retry step.  With random-init weights x leaves the data scale along the reverse SDE, layer 0 sees it un-normalised,
    // and from then on EVERY one of its units ran the fast pass, failed, and ran the exact passes (2.5 x; 2.6 % of the step).
    unsigned exact_layers = 0u;                                  // wave-uniform; layers >= 32 are not remembered (they retry the
                                                                 // fast pass every unit: slower, same results)
    for (int step = 0; step < nsteps; ++step) {
        if ((step & 15) == 0) exact_layers = 0u;                 // retry the fast pass every 16th step
        mark(0, step);
        refresh_lane();
        // ============================ time embedding (transformer.py:80-89)
        if (P.temb_table) {
            // sampler mode: every series has the step's t -- the embedding of all steps was computed before the launch
            // (fd_mega_temb_table): one load instead of a Fourier-feature + dense chain that two waves ran while six waited
            for (int i = threadIdx.x; i < S * D; i += NTH) temb[i] = P.temb_table[(size_t)step * D + (i % D)];
        } else {
            for (int sw = wave; sw < S; sw += NW) {       // one wave per series
                const int b = b0 + sw;
                float tv = 0.f;
                if (b < P.B) tv = (P.mode == FD_MEGA_SAMPLE) ? P.steps[step].t : P.tvec[b];
                time_embed_wave(tv, P.params, P.tW, P.td_w, P.td_b, temb + (S + sw) * D, temb + sw * D, D, lane);
            }
        }
        __syncthreads();

        // ============================ embed: h = x We^T + be + pe[t] + temb   (score_models.py:78-84)
#pragma unroll
        for (int oi = 0; oi < 2; ++oi) {
            const int tt = fh + 2 * oi;
            if (tt < ntile) {
                const int tile = tile0 + tt;
                int ser, t;
                bool valid;
                tile_token(tile, ser, t, valid);
                f32x4 acc[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) acc[dt] = f4zero();
                const float* xrow = P.x + ((size_t)(b0 + ser) * T + t) * C;
                for (int ks = 0; ks < SHP(KSE); ++ks) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = 32 * ks + 8 * g + e;
                        // agent-scope load: x was rewritten by other lanes in the previous step (bypass the CU's L1)
                        v[e] = (k < C) ? (valid ? __hip_atomic_load(xrow + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f)
                                       : (k == C ? 1.0f : 0.f);
                    }
                    u32x4 pk = {cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]),
                                cvt_pk_bf16(v[6], v[7])};
                    const bf16x8 xb = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) acc[dt] = MFMA(gfrag(P.img_emb, dt * SHP(KSE) + ks), xb, acc[dt]);
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int d0 = 16 * dt + 4 * g;
                    if (d0 < D) {
                        const int tc = (t < T) ? t : T - 1;
                        const float4 pe = *reinterpret_cast<const float4*>(P.params + P.pos + (size_t)tc * D + d0);
                        const float4 te = *reinterpret_cast<const float4*>(temb + ser * D + d0);
                        acc[dt][0] += pe.x + te.x;
                        acc[dt][1] += pe.y + te.y;
                        acc[dt][2] += pe.z + te.z;
                        acc[dt][3] += pe.w + te.w;
                    } else {
                        acc[dt] = f4zero();
                    }
                }
                write_xfrags(tile, acc);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) res[oi][dt] = acc[dt];
            }
        }
        __syncthreads();
        mark(1, step);
        refresh_lane();

        // ============================ encoder layers
        for (int l = 0; l < SHP(L); ++l) {
            const char* limg = layer_ptr(l);
            refresh_lane();
            // (the small fp32 vectors of the layer ride in the layer image and arrive with the first group's weight DMA below.
            //  One wave loading them -- offset table, then the vectors, then its share of the weight DMA: three dependent
            //  round trips in front of the layer's first barrier -- was most of a phase that took 8.6 K cycles per layer.)

            // -------- attention, one group of head pairs at a time
            for (int pg = 0; pg < NP; pg += NPG) {
                const int npg = min(NPG, NP - pg);
                // Weight stream of the attention groups.  Only the first group of a layer pays an exposed L2->LDS round
                // trip: W_k | W_v of group g+1 are fetched into their (dead) slots while group g's units run, and
                // become visible with the barrier that ends those units; W_q of group g+1 lands behind its K/V
                // projection.  The max|k|^2 table is double-buffered by group parity for the same reason.
                char* const wk = wsl;
                char* const wv = wsl + NPG * KS1 * 1024;
                char* const wq = wsl + 2 * NPG * KS1 * 1024;
                const int gpar = (pg / NPG) & 1;
                unsigned* const kmax = kmax2 + gpar * (NPG * S * 2);
                if (pg == 0) {
                    dma_blocks(limg + P.off_wk, wk, npg * KS1);
                    dma_blocks(limg + P.off_wv, wv, npg * KS1);
                    dma_blocks(limg + P.off_wq, wq, npg * KS1);
                    dma_blocks(limg + P.off_lpar, reinterpret_cast<char*>(lpar), P.nlp);
                    for (int i = threadIdx.x; i < NPG * S * 2; i += NTH) kmax[i] = 0u;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                } else {
                    dma_blocks(limg + P.off_wq + (size_t)pg * KS1 * 1024, wq, npg * KS1);
                }
                mark(2, step);
                // ---- K projection (K^T rows pair-major, 8 rows per head) and V projection (non-transposed, so the
                //      C tile is already the V^T A-fragment) for every token tile -> kbf / vbf
#if FD_KV_TILE_MAJOR
                // Tile-major: a wave takes whole token tiles and runs the group's pairs over ONE read of the tile's x fragments
                // (a (pair, tile) item per trip re-read them for every pair and had two MFMA chains to hide its LDS round trip
                // behind; here 2 x npg chains are in flight).  Same worst case per wave (2 tiles x 3 pairs at T = 100, S = 2).
                // (static group size: the group's W_k | W_v fragments stay in registers across the wave's tiles -- re-reading
                //  them per tile made the phase LDS-bandwidth bound: 14 tiles x 18 KiB per group)
                constexpr int NPGS = SH::NPG > 0 ? SH::NPG : 1;
                bf16x8 wkf[NPGS][KS1], wvf[NPGS][KS1];
                if (SH::NPG > 0) {
#pragma unroll
                    for (int pr = 0; pr < NPGS; ++pr)
#pragma unroll
                        for (int ks = 0; ks < KS1; ++ks) {
                            wkf[pr][ks] = *reinterpret_cast<const bf16x8*>(wk + ((pr * KS1 + ks) * 64 + lane) * 16);
                            wvf[pr][ks] = *reinterpret_cast<const bf16x8*>(wv + ((pr * KS1 + ks) * 64 + lane) * 16);
                        }
                }
                for (int tile = wave; tile < NTILE; tile += NW) {
                    bf16x8 xf[KS1];
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) xf[ks] = xfrag(tile, ks);
                    const int ser = tile / KT, kt = tile - ser * KT;
                    f32x4 a[SH::NPG > 0 ? SH::NPG : 1], b[SH::NPG > 0 ? SH::NPG : 1];
                    auto proj = [&](int pr, f32x4& ka, f32x4& vb) {
                        ka = f4zero();
                        vb = f4zero();
#pragma unroll
                        for (int ks = 0; ks < KS1; ++ks) {
                            if (SH::NPG > 0) {
                                ka = MFMA(wkf[pr][ks], xf[ks], ka);
                                vb = MFMA(xf[ks], wvf[pr][ks], vb);
                            } else {
                                ka = MFMA(*reinterpret_cast<const bf16x8*>(wk + ((pr * KS1 + ks) * 64 + lane) * 16), xf[ks], ka);
                                vb = MFMA(xf[ks], *reinterpret_cast<const bf16x8*>(wv + ((pr * KS1 + ks) * 64 + lane) * 16), vb);
                            }
                        }
                    };
                    auto store = [&](int pr, const f32x4& ka, const f32x4& vb) {
                        u32x2 pk = {cvt_pk_bf16(ka[0], ka[1]), cvt_pk_bf16(ka[2], ka[3])};
                        *reinterpret_cast<u32x2*>(kbf + ((size_t)(pr * NTOK + tile * 16 + tok) * 4 + g) * 8) = pk;
                        float n2 = ka[0] * ka[0] + ka[1] * ka[1] + ka[2] * ka[2] + ka[3] * ka[3];
                        float ea, eb;
                        swap16(n2, ea, eb);                           // the two lane groups of a head
                        n2 = row_max16(ea + eb);
                        if (tok == 0 && (g & 1) == 0)
                            __hip_atomic_fetch_max(&kmax[(pr * S + ser) * 2 + (g >> 1)], __builtin_bit_cast(unsigned, n2),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        u32x2 pv = {cvt_pk_bf16(vb[0], vb[1]), cvt_pk_bf16(vb[2], vb[3])};
                        char* dst = vbf + ((size_t)(((pr * S + ser) * NJ + (kt >> 1)) * 4 + g) * 16 + tok) * 16;
                        *reinterpret_cast<u32x2*>(dst + 8 * (kt & 1)) = pv;
                        if ((KT & 1) && kt == KT - 1) *reinterpret_cast<u32x2*>(dst + 8) = u32x2{0u, 0u};
                    };
                    if (SH::NPG > 0) {            // static group size: all pairs' chains issued before the first result is used
#pragma unroll
                        for (int pr = 0; pr < (SH::NPG > 0 ? SH::NPG : 1); ++pr)
                            if (pr < npg) proj(pr, a[pr], b[pr]);
#pragma unroll
                        for (int pr = 0; pr < (SH::NPG > 0 ? SH::NPG : 1); ++pr)
                            if (pr < npg) store(pr, a[pr], b[pr]);
                    } else {
                        for (int pr = 0; pr < npg; ++pr) {
                            proj(pr, a[0], b[0]);
                            store(pr, a[0], b[0]);
                        }
                    }
                }
#else
                for (int u = wave; u < npg * NTILE; u += NW) {
                    const int pr = u / NTILE, tile = u - pr * NTILE;
                    f32x4 a = f4zero(), b = f4zero();
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        const bf16x8 xf = xfrag(tile, ks);
                        a = MFMA(*reinterpret_cast<const bf16x8*>(wk + ((pr * KS1 + ks) * 64 + lane) * 16), xf, a);
                        b = MFMA(xf, *reinterpret_cast<const bf16x8*>(wv + ((pr * KS1 + ks) * 64 + lane) * 16), b);
                    }
                    u32x2 pk = {cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3])};
                    *reinterpret_cast<u32x2*>(kbf + ((size_t)(pr * NTOK + tile * 16 + tok) * 4 + g) * 8) = pk;
                    const int ser = tile / KT, kt = tile - ser * KT;
                    // max_j |k_j|^2 per (series, head) over the keys (softmax shift bound of the attention units):
                    // rows 4g+r are the dims (head g>>1), the 16 lanes of a row are this tile's tokens
                    {
                        float n2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
                        float ea, eb;
                        swap16(n2, ea, eb);                           // the two lane groups of a head
                        n2 = row_max16(ea + eb);
                        if (tok == 0 && (g & 1) == 0)
                            __hip_atomic_fetch_max(&kmax[(pr * S + ser) * 2 + (g >> 1)], __builtin_bit_cast(unsigned, n2),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    // V: lane (col = lane&15, g) holds 4 consecutive tokens (keys) 4g+r of this tile
                    u32x2 pv = {cvt_pk_bf16(b[0], b[1]), cvt_pk_bf16(b[2], b[3])};
                    char* dst = vbf + ((size_t)(((pr * S + ser) * NJ + (kt >> 1)) * 4 + g) * 16 + tok) * 16;
                    *reinterpret_cast<u32x2*>(dst + 8 * (kt & 1)) = pv;
                    if ((KT & 1) && kt == KT - 1) *reinterpret_cast<u32x2*>(dst + 8) = u32x2{0u, 0u};
                }
#endif
                if (threadIdx.x == 0) *ucnt = 0u;                     // (the previous group's units ended with a barrier)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // W_q of this group
                __syncthreads();
                mark(3, step);
                refresh_lane();
                if (pg + NPG >= NP && WO_LDS) {
                    // last group: its W_k | W_v slots are dead -> fetch W_o there for the out-proj (an L2 round trip and
                    // DT*KSO vector loads per wave off the critical path of the next phase)
                    dma_blocks(limg + P.off_wo, wk, DT * KSO);
                }
                if (pg + NPG < NP) {                                  // next group's W_k | W_v and its (zeroed) max table
                    const int npn = min(NPG, NP - pg - NPG);
                    dma_blocks(limg + P.off_wk + (size_t)(pg + NPG) * KS1 * 1024, wk, npn * KS1);
                    dma_blocks(limg + P.off_wv + (size_t)(pg + NPG) * KS1 * 1024, wv, npn * KS1);
                    for (int i = threadIdx.x; i < NPG * S * 2; i += NTH) kmax2[(gpar ^ 1) * (NPG * S * 2) + i] = 0u;
                }
                // ---- attention units: (head pair) x (series) x (NQ consecutive query tiles).  The NQ query tiles
                //      share every K / V fragment read and give each wave NQ independent dependency chains (one
                //      wave has only one partner on its SIMD to hide MFMA / exp / LDS latency behind).
                //      Units are handed out dynamically (LDS counter), two-tile units first, then the single-tile
                //      leftovers of an odd tile count: the older wave of each SIMD wins every issue arbitration and
                //      finishes a unit ~1.5x faster than its partner, so a static split leaves the SIMD to one
                //      (slow, alone) wave for the last third of the phase.
#if FD_PROF_UNITS
                const unsigned long long tw0 = P.prof ? __builtin_readcyclecounter() : 0ull;
#endif
                auto do_unit = [&](auto nqc, int pr, int ser, int qt0) {
                    constexpr int NQ = decltype(nqc)::value;
                    int qt[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) qt[q] = ser * KT + qt0 + q;
                    umark(9, step);
                    f32x4 qa[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) qa[q] = f4zero();
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wq + ((pr * KS1 + ks) * 64 + lane) * 16);
#pragma unroll
                        for (int q = 0; q < NQ; ++q) qa[q] = MFMA(wf, xfrag(qt[q], ks), qa[q]);
                    }
                    // The C tile holds both heads of the pair (even head in lane groups 0-1, odd in 2-3).  Masking
                    // Q once per unit (instead of every K fragment) selects the head: the K fragment then serves
                    // both heads unmodified because the other head's k-slots meet zeros.
                    // S^T tiles contract over 16 k-slots (8 dims x 2 heads): the K=16 MFMA takes the 8-byte K rows
                    // as they lie in LDS (no zero-padded upper half to materialise)
                    const bool lo_grp = (g >> 1) == 0;
                    u32x2 qraw[NQ];          // this lane's 4 dims of its own head (head g>>1), bf16
#pragma unroll
                    for (int q = 0; q < NQ; ++q) qraw[q] = u32x2{cvt_pk_bf16(qa[q][0], qa[q][1]), cvt_pk_bf16(qa[q][2], qa[q][3])};
                    auto qmasked = [&](int q, int hs) -> s16x4 {      // head hs's operand: the other head's lane groups zeroed
                        const bool mine = lo_grp == (hs == 0);
                        const u32x2 w = {mine ? qraw[q][0] : 0u, mine ? qraw[q][1] : 0u};
                        return __builtin_bit_cast(s16x4, w);
                    };
                    // keys beyond T in the ragged last tile: masked through the MFMA's C operand
                    f32x4 cmask;
#pragma unroll
                    for (int r = 0; r < 4; ++r) cmask[r] = ((KT - 1) * 16 + 4 * g + r >= T) ? kNegBig : 0.f;
                    // Softmax shift.  Fast path: the bound  |q| max_j |k_j|  >= max_j q.k_j  (2 % headroom for
                    // the bf16 rounding of q and k) replaces the row maximum -- any shift cancels in P V / sum P as long
                    // as nothing overflows (bound >= max) or flushes to zero; P keeps the fp32 exponent range in bf16.
                    // If a row sum comes out below 2^-100 (bound > max + ~100: not seen with real weights) the unit is
                    // redone with the exact two-pass maximum.
                    // Q with -bound in k-slot hd of its head: lane group 2hs + (hd >> 2), element hd & 3.  Every lane patches
                    // its OWN head's bound into its own words first (one convert + merge per query tile), the per-head
                    // operands are then two masked copies: 14 VALU instructions fewer per unit than masking first and
                    // patching each copy, and only the 4 raw words stay live for the exact path.
                    s16x4 qs[NQ][2];
                    {
                        const float k2 = reinterpret_cast<const float*>(kmax)[(pr * S + ser) * 2 + (g >> 1)];
                        const bool slot_here = (g & 1) == (hd >> 2);
                        const int dw = (hd & 3) >> 1;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            const float part = qa[q][0] * qa[q][0] + qa[q][1] * qa[q][1] + qa[q][2] * qa[q][2] + qa[q][3] * qa[q][3];
                            float ea, eb;
                            swap16(part, ea, eb);                     // the two lane groups of a head
                            const float bnd = __builtin_amdgcn_sqrtf((ea + eb) * k2) * 1.02f;   // (v_sqrt_f32, 1 ulp: the 2 % headroom covers it; the IEEE form is 15 instructions)
                            const unsigned nb = cvt_pk_bf16(-bnd, 0.f) & 0xffffu;
                            const unsigned old = dw ? qraw[q][1] : qraw[q][0];
                            const unsigned patched = (hd & 1) ? ((old & 0x0000ffffu) | (nb << 16)) : ((old & 0xffff0000u) | nb);
                            const unsigned neww = slot_here ? patched : old;
                            const unsigned w0 = dw ? qraw[q][0] : neww, w1 = dw ? neww : qraw[q][1];
                            const u32x2 qe = {lo_grp ? w0 : 0u, lo_grp ? w1 : 0u};
                            const u32x2 qo = {lo_grp ? 0u : w0, lo_grp ? 0u : w1};
                            qs[q][0] = __builtin_bit_cast(s16x4, qe);
                            qs[q][1] = __builtin_bit_cast(s16x4, qo);
                        }
                    }
                    float m2[NQ][2];
                    f32x4 o2[NQ][2];
                    f32x4 l2[HD8 ? NQ : 1][2];      // head_dim 8: row sums of P (every row of the tile carries the sum)
                    const bf16x8 ones8 = __builtin_bit_cast(bf16x8, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
                    auto run_unit = [&](auto exact_c) {
                    constexpr bool EXACT = decltype(exact_c)::value;
                    s16x4 qb[NQ][2];         // unpatched operands: exact path only
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int hs = 0; hs < 2; ++hs) {
                            if (EXACT) qb[q][hs] = qmasked(q, hs);
                            m2[q][hs] = kNegBig;
                            o2[q][hs] = f4zero();
                            if (HD8) l2[HD8 ? q : 0][hs] = f4zero();
                        }
                    if (SH::KT == 0 || FD_ROLLED_ATTN) {
                        // Run-time series length: the hand-unrolled pipeline below would need a guard around every MFMA
                        // (each guard = its own basic block; measured 2.6x slower than with a static tile count).  A
                        // rolled loop over PAIRS of key tiles (= one V^T block) keeps a static body instead: a missing
                        // odd tile re-reads the previous one (finite data) and is masked through the C operand.
                        const f32x4 allneg = {kNegBig, kNegBig, kNegBig, kNegBig};
                        const char* kbase = kbf + ((size_t)(pr * NTOK + ser * KT * 16 + tok) * 4 + g) * 8;
                        const char* vbase = vbf + ((size_t)((pr * S + ser) * NJ * 4 + g) * 16 + tok) * 16;
                        auto kfrag = [&](int kt) { return *reinterpret_cast<const s16x4*>(kbase + (size_t)kt * 512); };
                        auto vfrag = [&](int jb) { return *reinterpret_cast<const bf16x8*>(vbase + (size_t)jb * 1024); };
                        if (EXACT) {
                            // pass 1 over the whole series: exact row maxima
                            float bm[NQ][2];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) bm[q][0] = bm[q][1] = kNegBig;
                            for (int kt = 0; kt < KT; ++kt) {
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
                                for (int hs = 0; hs < 2; ++hs) m2[q][hs] = group_max(bm[q][hs]);
                        }
                        f32x4 negm[EXACT ? NQ : 1][2];
                        if (EXACT) {
#pragma unroll
                            for (int q = 0; q < NQ; ++q)
#pragma unroll
                                for (int hs = 0; hs < 2; ++hs) negm[q][hs] = f32x4{-m2[q][hs], -m2[q][hs], -m2[q][hs], -m2[q][hs]};
                        }
                        for (int jb = 0; jb < NJ; ++jb) {
                            const int ka = 2 * jb, kb2 = (2 * jb + 1 < KT) ? 2 * jb + 1 : ka;
                            const s16x4 kfa = kfrag(ka), kfb = kfrag(kb2);
                            const bf16x8 vfj = vfrag(jb);
                            // tile a is the series' last only when KT is odd; tile b either is the last or does not exist
                            const f32x4 ma = (ka == KT - 1) ? cmask : f4zero();
                            const f32x4 mb = (2 * jb + 1 >= KT) ? allneg : ((kb2 == KT - 1) ? cmask : f4zero());
#pragma unroll
                            for (int q = 0; q < NQ; ++q)
#pragma unroll
                                for (int hs = 0; hs < 2; ++hs) {
                                    f32x4 pa, pb;
                                    if (EXACT) {
                                        pa = MFMA16(kfa, qb[q][hs], ma + negm[q][hs]);
                                        pb = MFMA16(kfb, qb[q][hs], mb + negm[q][hs]);
                                    } else {
                                        pa = MFMA16(kfa, qs[q][hs], ma);
                                        pb = MFMA16(kfb, qs[q][hs], mb);
                                    }
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        pa[r] = FD_EXP2(pa[r]);
                                        pb[r] = FD_EXP2(pb[r]);
                                    }
                                    const bf16x8 pkj = pack8(pa, pb);
                                    o2[q][hs] = MFMA(vfj, pkj, o2[q][hs]);
                                    if (HD8) l2[HD8 ? q : 0][hs] = MFMA(ones8, pkj, l2[HD8 ? q : 0][hs]);
                                }
                        }
                    } else
                    for (int kb = 0; kb < KT; kb += 8) {
                        // K and V fragments of this 128-key block: one read serves both heads and all NQ query tiles
                        s16x4 kf[8];
                        bf16x8 vf[4];
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (kb + j < KT)
                                kf[j] = *reinterpret_cast<const s16x4*>(
                                    kbf + ((size_t)(pr * NTOK + (ser * KT + kb + j) * 16 + tok) * 4 + g) * 8);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            if ((kb >> 1) + jj < NJ)
                                vf[jj] = *reinterpret_cast<const bf16x8*>(
                                    vbf + ((size_t)(((pr * S + ser) * NJ + (kb >> 1) + jj) * 4 + g) * 16 + tok) * 16);
                        // Both passes are software-pipelined by hand, a few MFMAs ahead of their consumers, and
                        // fenced per stage: left alone hipcc issues MFMA -> s_nop 7 -> 4 exps strictly in sequence.
                        // Tile index k = ((hs * 4 + jj) * NQ + q) * 2 + jl with key tile j = 2 jj + jl.
                        const int nk = min(8, KT - kb);                 // key tiles in this block
                        constexpr int NKT = 16 * NQ;                    // score tiles per block
                        umark(10, step);
                        if (EXACT) {
                            // pass 1 (exact path only): row maxima; the scores are recomputed in pass 2 with -max
                            // riding in the C operand, which removes one v_sub per score
                            float bm[NQ][2];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) bm[q][0] = bm[q][1] = kNegBig;
                            constexpr int LAG = 3;
                            f32x4 t4[NKT];
#pragma unroll
                            for (int k = 0; k < NKT + LAG; ++k) {
                                if (k < NKT) {
                                    const int jl = k & 1, q = (k >> 1) % NQ, jj = ((k >> 1) / NQ) & 3, hs = (k >> 1) / NQ >> 2;
                                    const int j = 2 * jj + jl;
                                    if (j < nk) t4[k] = MFMA16(kf[j], qb[q][hs], (kb + j == KT - 1) ? cmask : f4zero());
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
#pragma unroll
                            for (int q = 0; q < NQ; ++q)
#pragma unroll
                                for (int hs = 0; hs < 2; ++hs) {
                                    const float mnew = fmaxf(m2[q][hs], group_max(bm[q][hs]));
                                    const float alpha = __builtin_amdgcn_exp2f(m2[q][hs] - mnew);
                                    o2[q][hs] = o2[q][hs] * alpha;
                                    if (HD8) l2[HD8 ? q : 0][hs] = l2[HD8 ? q : 0][hs] * alpha;
                                    m2[q][hs] = mnew;
                                }
                        }
                        umark(11, step);
                        // Exact path: -max rides in the C operand.  Fast path: the (per-unit constant) shift rides in
                        // the contraction itself -- K carries a 1.0 in the free dim slot hd (bias row of the W_k image)
                        // and Q gets -bound there -- so C is an inline 0 and no splat registers are live.
                        f32x4 negm[EXACT ? NQ : 1][2], clast[EXACT ? NQ : 1][2];
                        if (EXACT) {
#pragma unroll
                            for (int q = 0; q < NQ; ++q)
#pragma unroll
                                for (int hs = 0; hs < 2; ++hs) {
                                    const float mm = m2[q][hs];
                                    negm[q][hs] = f32x4{-mm, -mm, -mm, -mm};
                                    clast[q][hs] = cmask - mm;
                                }
                        }
                        umark(12, step);
                        // pass 2: P = exp2(S - shift) tile by tile, packed to bf16 B fragments, then P V.  The row sum of
                        // P comes out of the same MFMAs: V^T carries a row of ones (dim slot hd).
                        {
                            constexpr int LAG = 2;
                            f32x4 pe[NKT];
                            bf16x8 pk[NKT / 2];
#pragma unroll
                            for (int k = 0; k < NKT + 2 * LAG; ++k) {
                                if (k < NKT) {
                                    const int jl = k & 1, q = (k >> 1) % NQ, jj = ((k >> 1) / NQ) & 3, hs = (k >> 1) / NQ >> 2;
                                    const int j = 2 * jj + jl;
                                    if (j < nk) {
                                        if (EXACT) pe[k] = MFMA16(kf[j], qb[q][hs], (kb + j == KT - 1) ? clast[q][hs] : negm[q][hs]);
                                        else pe[k] = MFMA16(kf[j], qs[q][hs], (kb + j == KT - 1) ? cmask : f4zero());
                                    } else {
                                        pe[k] = f4zero();
                                    }
                                }
                                if (k >= LAG && k - LAG < NKT) {
                                    const int e = k - LAG;
                                    const int jl = e & 1, jj = ((e >> 1) / NQ) & 3;
                                    if (2 * jj + jl < nk) {
#pragma unroll
                                        for (int r = 0; r < 4; ++r) pe[e][r] = FD_EXP2(pe[e][r]);
                                    }
                                    if (jl) pk[e >> 1] = pack8(pe[e - 1], pe[e]);
                                }
                                if (k >= 2 * LAG && ((k - 2 * LAG) & 1)) {
                                    const int e = k - 2 * LAG;
                                    const int q = (e >> 1) % NQ, jj = ((e >> 1) / NQ) & 3, hs = (e >> 1) / NQ >> 2;
                                    if (2 * jj < nk) {
                                        o2[q][hs] = MFMA(vf[jj], pk[e >> 1], o2[q][hs]);
                                        if (HD8) l2[HD8 ? q : 0][hs] = MFMA(ones8, pk[e >> 1], l2[HD8 ? q : 0][hs]);
                                    }
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                    };
                    // row sums of P: O^T rows 8*hs + [0,8) live in lane groups 2hs, 2hs+1; row 8*hs + hd is the ones row.
                    // hd in [4,7]: register hd-4 of the odd lane group; hd < 4: register hd of the even one
                    float lrow[NQ];
                    auto row_sums = [&]() -> bool {
                        bool bad = false;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            if (HD8) {
                                lrow[q] = lo_grp ? l2[HD8 ? q : 0][0][0] : l2[HD8 ? q : 0][1][0];
                                bad |= !(lrow[q] > 7.8e-31f);
                                continue;
                            }
                            float cand = lo_grp ? o2[q][0][0] : o2[q][1][0];
#pragma unroll
                            for (int r = 1; r < 4; ++r) cand = ((hd & 3) == r) ? (lo_grp ? o2[q][0][r] : o2[q][1][r]) : cand;
                            float row_even, row_odd;
                            swap16(cand, row_even, row_odd);
                            lrow[q] = (hd >= 4) ? row_odd : row_even;
                            bad |= !(lrow[q] > 7.8e-31f);                 // 2^-100; also catches NaN
                        }
                        return bad;
                    };
                    // (dbg bit 64 suppresses the fallback: lets the tests prove that it is what rescues such rows; bit 32: exact only)
                    bool need_exact = HD8 || (P.dbg & 32) != 0 || (l < 32 && ((exact_layers >> l) & 1u) != 0u && !(P.dbg & 64));
                    if (!HD8 && !need_exact) {
                        run_unit(std::false_type{});
                        const bool bad = row_sums();
                        if (__builtin_amdgcn_ballot_w64(bad) != 0ull && !(P.dbg & 64)) {   // wave-uniform: redo with the exact maximum
                            need_exact = true;
                            if (l < 32) exact_layers |= 1u << l;
                        }
                    }
                    if (need_exact) {
                        run_unit(std::true_type{});
                        (void)row_sums();
                    }
                    umark(13, step);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        float o_sel[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) o_sel[r] = lo_grp ? o2[q][0][r] : o2[q][1][r];
                        const float inv = __builtin_amdgcn_rcpf(lrow[q]);    // (v_rcp_f32, 1 ulp, instead of the 10-instruction IEEE division: the result is rounded to bf16)
                        // head = 2*(pg+pr) + (g>>1); its 8 dims are one 16-B k-slot group of the out-proj B fragment
                        const int head = 2 * (pg + pr) + (g >> 1);
                        u32x2 pk = {cvt_pk_bf16(o_sel[0] * inv, o_sel[1] * inv), cvt_pk_bf16(o_sel[2] * inv, o_sel[3] * inv)};
                        if (head >= H) pk = u32x2{0u, 0u};
                        if (head < 4 * KSO)
                            *reinterpret_cast<u32x2*>(afr + ((qt[q] * KSO + (head >> 2)) * 64 + (head & 3) * 16 + tok) * 16 +
                                                      8 * (g & 1)) = pk;
                    }
                };
                {
                    const int DF = KT >> 1;                            // two-tile units per (pair, series)
                    const int ND = npg * S * DF, NU = ND + ((KT & 1) ? npg * S : 0);
                    // (s_setprio 1 for the younger wave of every SIMD, or for the waves whose last unit is a single tile, during the
                    //  units: -0.7 % / -0.9 % per diffusion step, same box -- unlike the FFN loop the units are VALU-issue bound)
#if FD_STATIC_UNITS
                    // Static hand-out: wave w runs units w, w + NW, ...  (The dynamic LDS counter balanced the per-wave times
                    // but never changed the phase time -- a wave left alone on its SIMD runs at nearly the throughput of two --
                    // and cost an LDS atomic round trip, ~12 VALU instructions and a wait for the weight prefetch per unit.)
                    for (int u = wave; u < NU; u += NW) {
#else
                    for (;;) {
                        int u = 0;
                        if (lane == 0) u = (int)__hip_atomic_fetch_add(ucnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        u = __builtin_amdgcn_readfirstlane(u);
                        if (u >= NU) break;
#endif
                        if (u < ND) {
                            const int pr = u / (S * DF), ur = u - pr * (S * DF);
                            const int ser = ur / DF, du = ur - ser * DF;
                            do_unit(std::integral_constant<int, 2>{}, pr, ser, 2 * du);
                        } else {
                            const int v = u - ND, pr = v / S, ser = v - pr * S;
                            do_unit(std::integral_constant<int, 1>{}, pr, ser, KT - 1);
                        }
                        refresh_lane();
                    }
                }
#if FD_PROF_UNITS
                if (P.prof && blockIdx.x == 0 && step == 1 && l == 1 && lane == 0) {   // per-wave unit-loop time
                    P.prof[2 * (4000 + 8 * (pg / NPG) + wave)] = 100 + wave;
                    P.prof[2 * (4000 + 8 * (pg / NPG) + wave) + 1] = __builtin_readcyclecounter() - tw0;
                }
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // prefetched W_k | W_v of the next group
                __syncthreads();
            }

            mark(4, step);
            refresh_lane();
#if FD_LDS_DUMP
            if (P.dbg_out && blockIdx.x == 0 && l == 0 && step == 0) {      // debugging aid: dump LDS
                for (int i = threadIdx.x; i < P.dbg_bytes / 4; i += NTH) P.dbg_out[i] = reinterpret_cast<unsigned*>(smem)[i];
                __syncthreads();
            }
#endif
            // -------- FFN weight stream: a ring of NBUF chunk buffers filled 3 steps ahead of their use; step s lives in
            //          buffer (s + rb) % NBUF.  Steps 0 and 1 are fetched during the out-proj into dead W / K / V space:
            //          rb = 1 -> buffers 1-2 (buffer 0 overlays the W_k | W_v slots that hold W_o during the out-proj),
            //          rb = 0 -> buffers 0-1.  The other two buffers may overlay afr, which the out-proj still reads, and
            //          are first filled after the barrier that ends it.
            const int NS = SHP(F) / 64;
            // Every CU streams the SAME weights; marching through them in lockstep makes all 32 CUs of an XCD hit
            // the same L2 channel at the same time (measured: the stream ran at ~25 GB/s per CU and bounded the FFN
            // loop).  The F chunks are summed, so each workgroup walks them in its own rotated order.
            const int st_rot = (blockIdx.x >> 3) % NS;
            auto issue_ffn = [&](int st_seq) {
                int st = st_seq + st_rot;
                st -= (st >= NS) ? NS : 0;
                // the image is chunk-major ([32-wide chunk][F-half][block]) and so is a ring buffer: one linear copy.
                // Every wave issues exactly NDMA instructions (the last ones repeat a block) so that
                // `s_waitcnt vmcnt(NDMA)` means "everything but the newest buffer has landed" for all waves.
                const char* src = limg + (F32 ? P.off_ffn32 : P.off_ffn) + (size_t)st * WB1 + lane * 16;
                char* dst = ring + ((st_seq + rb) % NBUF) * WB1;
#pragma unroll
                for (int i = 0; i < NDMA; ++i) {
                    int b = wave + i * NW;
                    b -= (b >= 2 * NBF) ? NW : 0;
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src + b * 1024), LDS_PTR(dst + b * 1024), 16, 0, 0);
                }
            };
            // in the FFN loop the two F-half wave sets take turns (even / odd steps) issuing a whole buffer: a DMA
            // instruction costs its wave 60-180 issue cycles, and with every wave issuing right after the barrier both
            // waves of each SIMD were away from the matrix pipe at the same time
            constexpr int NDH = (2 * NBF + MQ - 1) / MQ;
            auto issue_ffn_half = [&](int st_seq) {
                int st = st_seq + st_rot;
                st -= (st >= NS) ? NS : 0;
                const char* src = limg + (F32 ? P.off_ffn32 : P.off_ffn) + (size_t)st * WB1 + lane * 16;
                char* dst = ring + ((st_seq + rb) % NBUF) * WB1;
                const int w4 = wave % MQ;
#pragma unroll
                for (int i = 0; i < NDH; ++i) {
                    int b = w4 + i * MQ;
                    b -= (b >= 2 * NBF) ? MQ : 0;
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src + b * 1024), LDS_PTR(dst + b * 1024), 16, 0, 0);
                }
            };
            issue_ffn(0);
            if (NS > 1) issue_ffn(1);

            // -------- out-proj + residual + LayerNorm1 on the owned tiles (W_o fragments: one L2 round trip)
            bf16x8 wo[DT][KSO];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int ks = 0; ks < KSO; ++ks)
                    wo[dt][ks] = WO_LDS ? *reinterpret_cast<const bf16x8*>(wsl + ((dt * KSO + ks) * 64 + lane) * 16)
                                        : gfrag(limg + P.off_wo, dt * KSO + ks);
            {
                // Both owned tiles in ONE basic block (a wave that owns a single tile runs its first tile twice and skips the
                // second write): behind a `tt < ntile` branch each tile's out-proj -> residual -> LayerNorm -> fragment chain
                // ran strictly after the other's (2.9 K cycles per tile in the sub-phase marks, no latency hidden).
                bool ok[2];
                int tl[2];
#pragma unroll
                for (int oi = 0; oi < 2; ++oi) {
                    const int tt = fh + 2 * oi;
                    ok[oi] = tt < ntile;
                    tl[oi] = tile0 + (ok[oi] ? tt : 0);
                }
                f32x4 acc[2][DT];
#pragma unroll
                for (int oi = 0; oi < 2; ++oi)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) acc[oi][dt] = f4zero();
#pragma unroll
                for (int ks = 0; ks < KSO; ++ks)
#pragma unroll
                    for (int oi = 0; oi < 2; ++oi) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(afr + ((tl[oi] * KSO + ks) * 64 + lane) * 16);
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) acc[oi][dt] = MFMA(wo[dt][ks], af, acc[oi][dt]);
                    }
#pragma unroll
                for (int oi = 0; oi < 2; ++oi)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const int d0 = 16 * dt + 4 * g;
                        if (d0 < D) {
                            const float4 bo = *reinterpret_cast<const float4*>(lpar + 0 * D + d0);
                            res[oi][dt][0] += acc[oi][dt][0] + bo.x;
                            res[oi][dt][1] += acc[oi][dt][1] + bo.y;
                            res[oi][dt][2] += acc[oi][dt][2] + bo.z;
                            res[oi][dt][3] += acc[oi][dt][3] + bo.w;
                        }
                    }
                layer_norm2(res[0], res[1], lpar + 2 * D, lpar + 3 * D);
#pragma unroll
                for (int oi = 0; oi < 2; ++oi)
                    if (ok[oi]) write_xfrags(tl[oi], res[oi]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (NS > 2) issue_ffn(2);                // afr is dead now: the buffers overlaying it may fill

            mark(5, step);
            refresh_lane();
            // -------- FFN: hidden never leaves registers (see fd_score_bf16.hip)
            // The whole phase is instantiated per F-half (FH): which accumulator tiles a wave owns then is a
            // compile-time fact -- no per-element selects, and accumulators die as soon as they are exchanged
            // (with a runtime fh hipcc kept all 80 accumulator registers plus both candidates live and spilled).
            auto ffn_phase = [&](auto fhc) {
                constexpr int FH = decltype(fhc)::value;
                f32x4 acc[DT][MT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int tt = 0; tt < MT; ++tt)
                        acc[dt][tt] = ((tt & 1) == FH) ? res[(tt >> 1) & 1][dt] : f4zero();   // owner tiles start from the residual
                // The chunk body is instantiated per tile count so that it is ONE basic block: with a runtime
                // `tt < ntile` guard every tile becomes its own block and hipcc cannot interleave the tiles'
                // LDS reads / MFMAs / relu (measured: 45 % MFMA-pipe occupancy in this loop, fully serialised).
                auto ffn_loop = [&](auto ntc) {
                    constexpr int NTT = decltype(ntc)::value;
                    bf16x8 xf[FD_XF_REGS ? NTT : 1][KS1];           // activation fragments kept in registers (optional)
                    if (FD_XF_REGS) {
#pragma unroll
                        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
                            for (int ks = 0; ks < KS1; ++ks) xf[tt][ks] = xfrag(tile0 + tt, ks);
                    }
                    // One step = one 32-wide chunk per F-half x NTT tiles; item (s, i): H (2 K-chains of KS1 MFMAs:
                    // the two 16-wide hidden tiles), relu+pack (VALU), W2 (DT MFMAs into the tile's accumulators).
                    // The schedule is pinned by hand (sched_barrier between stages), one item ahead and straight
                    // across the step boundary:
                    //   H(next item) | relu(item) | W2(item)
                    // so the VALU of an item and the MFMA latency of its H hide behind the next H's MFMAs; left
                    // alone hipcc emits H, s_nop, relu, W2 strictly in sequence and waits for ALL weight fragments
                    // of a chunk (lgkmcnt(0)) before its first MFMA.  The W1 / W2 fragment registers are refilled for
                    // the NEXT step as soon as their last reader has issued: with the ring filled 3 steps ahead, the
                    // next step's buffer became visible one barrier ago, so no LDS read ever waits on the barrier
                    // it follows, and the barrier only keeps the waves within one step of each other (buffer reuse).
                    bf16x8 w1[2][KS1], w2[DT];
                    f32x4 h0, h1;                                     // hidden tiles of the item in flight
                    auto load_w1 = [&](int s) {
                        const char* wb = ring + ((s + rb) % NBUF) * WB1 + FH * NBF * 1024 + lane * 16;
#pragma unroll
                        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                            for (int ks = 0; ks < KS1; ++ks)
                                w1[ft][ks] = *reinterpret_cast<const bf16x8*>(wb + (ft * KS1 + ks) * 1024);
                    };
                    auto load_w2 = [&](int s) {
                        const char* wb = ring + ((s + rb) % NBUF) * WB1 + FH * NBF * 1024 + lane * 16;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt)
                            w2[dt] = *reinterpret_cast<const bf16x8*>(wb + (2 * KS1 + dt) * 1024);
                    };
                    auto do_h = [&](int tt) {
                        h0 = f4zero();
                        h1 = f4zero();
#pragma unroll
                        for (int ks = 0; ks < KS1; ++ks) {
                            const bf16x8 xv = FD_XF_REGS ? xf[FD_XF_REGS ? tt : 0][ks] : xfrag(tile0 + tt, ks);
                            h0 = MFMA(w1[0][ks], xv, h0);
                            h1 = MFMA(w1[1][ks], xv, h1);
                        }
                    };
                    load_w1(0);
                    load_w2(0);
                    do_h(0);
                    __builtin_amdgcn_sched_barrier(0);
#if FD_FFN_PRIO
                    // Two waves share a SIMD and a step ends with a barrier: the older wave wins every arbitration, finishes its items
                    // first and leaves the younger one to run the rest of the step alone, every stall exposed.  With the lighter wave
                    // (one item less + the DMA issue) preferred, the heavy wave fills its gaps (same-box A/B on the ecg shape: +2.1 %).
                    if (FD_DMA_LIGHT && (2 * trem == MQ) && (SHP(rot) == trem) && ntile < MT) __builtin_amdgcn_s_setprio(1);
#endif
                    for (int st = 0; st < NS; ++st) {
#if FD_DMA_LIGHT
                        // Uneven tile split (e.g. 14 tiles = 4,4,3,3 per quarter): the MQ waves that carry one tile less issue the
                        // whole buffer every step -- they have an item's worth of slack per step, and the waves on the critical
                        // path never leave the matrix pipe for the texture path.  Even splits alternate the F-half sets as before.
                        // (exactly MQ such waves with distinct wave % MQ -- issue_ffn_half's block split -- exist when half of the
                        //  quarters carry the extra tile and the F-half sets are rotated by that half: 14 tiles, rot 2)
                        const bool light_ok = (2 * trem == MQ) && (SHP(rot) == trem);
                        const bool my_turn = light_ok ? (ntile < MT) : ((st & 1) == FH);
#else
                        const bool my_turn = ((st & 1) == FH);
#endif
                        if (my_turn && st + 3 < NS && FD_DMA_ON) issue_ffn_half(st + 3);
                        // NTT == 1: H(0) of this step issued during the previous step, before this step's successor
                        // buffer was visible -- its W1 can only be fetched now
                        if (NTT == 1 && st + 1 < NS) load_w1(st + 1);
#pragma unroll
                        for (int i = 0; i < NTT; ++i) {
                            const f32x4 g0 = h0, g1 = h1;             // H(st, i): complete by the time the next H issued
                            if (i + 1 < NTT) {
                                do_h(i + 1);
                                if (i + 1 == NTT - 1 && st + 1 < NS) load_w1(st + 1);   // last reader of W1(st) issued
                            } else if (st + 1 < NS) {
                                do_h(0);                              // first item of the next step
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            const bf16x8 hb = relu_pack(g0, g1);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int dt = 0; dt < DT; ++dt) acc[dt][i] = MFMA(w2[dt], hb, acc[dt][i]);
                            if (i == NTT - 1 && st + 1 < NS) load_w2(st + 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        // buffer st+2 must have landed before anyone reads it in step st+1 (own DMA, then barrier)
                        // (this wave's share of buffer st+2 was issued one step ago if it was not its turn now; a wave
                        // whose turn it is has nothing older than the NDH instructions it just issued, except at st = 0)
                        if (my_turn && st + 3 < NS && FD_DMA_ON) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDH) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        // bare s_barrier: __syncthreads() carries a workgroup fence that hipcc lowers to
                        // s_waitcnt vmcnt(0) lgkmcnt(0) -- it would drain the DMA issued this very step (3 steps of
                        // slack thrown away, measured) and the fragment prefetch of the next step.  What must be
                        // ordered is ordered by hand: this wave's share of buffer st+2 has landed (vmcnt above); LDS
                        // reads of a buffer are consumed (waited on) a full step before that buffer is refilled.
                        __builtin_amdgcn_s_barrier();
                    }
#if FD_FFN_PRIO
                    if (FD_DMA_LIGHT && (2 * trem == MQ) && (SHP(rot) == trem) && ntile < MT) __builtin_amdgcn_s_setprio(0);
#endif
                };
                // ---- pair form (ShapeStatic<..., FFN32 = 1>): item = a PAIR of token tiles.  H^T (32 hidden x 32 tokens) = KS32
                // v_mfma_f32_32x32x16_bf16 (160 cycles against 2 x 96 in the 16x16x32 form), relu + pack + four v_permlane16_swap turn
                // the C tile into the pair's two 16x16x32 B fragments, W2 = 2 x DT 16x16x32 MFMAs.  A wave with three tiles runs its odd
                // one in the 16x16x32 form from the SAME weight image through a per-lane address map (row a of hidden tile ft <-> image
                // row 16 (a>>2 & 1) + 4 (a>>3) + (a&3) + 8 ft: pack8(h0, h1) then has the pair-form W2 image's k-slot order).
                // SHAPE 0: two pairs (tile0 even, 4 tiles); 1: pair then single (tile0 even, 3 tiles); 2: single then pair (tile0 odd).
                // Prototype and measurements: scripts/ubench/ffn32_proto.hip, profiles/r04_ffn_proto_matrix*.txt.
                auto ffn_loop32 = [&](auto shc) {
                    constexpr int SHAPE = decltype(shc)::value;
                    constexpr int NPAIR = SHAPE == 0 ? 2 : 1;
                    constexpr bool SINGLE = SHAPE != 0;
                    constexpr int acc_p0 = SHAPE == 2 ? 1 : 0;     // accumulator (tile) index of the first pair's first tile
                    constexpr int acc_s = SHAPE == 1 ? 2 : 0;      // ... of the single tile
                    // (the image stores hidden row h of a chunk in lane slot h with bits 3 and 4 swapped: ds_read_b128 serves 16 lanes
                    //  per cycle -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- and both this reader's rows and the pair reader's
                    //  then fall on 16 distinct 16-byte bank slots per lane group; natural order: 2-way conflicts on all six reads)
                    const int a16 = lane & 15;
#if FD_W1_SWAP34
                    const int hr = 8 * ((a16 >> 2) & 1) + 4 * (a16 >> 3) + (a16 & 3);   // row 16 (a>>2 & 1) + 4 (a>>3) + (a&3), bits 3 <-> 4
                    const int lperm = (lane & 39) | ((lane & 8) << 1) | ((lane & 16) >> 1);   // lane with bits 3 <-> 4 (row L & 31 of the pair form)
                    constexpr int FTB = 256;
#else
                    const int hr = 16 * ((a16 >> 2) & 1) + 4 * (a16 >> 3) + (a16 & 3);
                    const int lperm = lane;
                    constexpr int FTB = 128;
#endif
                    const int pair0 = (SHAPE == 2 ? tile0 + 1 : tile0) >> 1;
                    const int stile = SHAPE == 1 ? tile0 + 2 : tile0;
                    // 32x32x16 B fragments of a pair, gathered from the tiles' 16x16x32 fragments as they lie in LDS: lane L supplies token
                    // L % 32 (tile 2 pair + (L >> 4 & 1), row L & 15) and k-slots 16 ks + 8 (L >> 5) .. + 7 = k-step ks >> 1, lane group
                    // 2 (ks & 1) + (L >> 5) of that tile (once per FFN phase: every other reader of the fragments keeps its linear address)
                    bf16x8 xp[NPAIR][KS32];
                    {
                        const char* xl = xfr + (2 * pair0 + ((lane >> 4) & 1)) * (KSX * 1024) + (16 * (lane >> 5) + tok) * 16;
#pragma unroll
                        for (int pp = 0; pp < NPAIR; ++pp)
#pragma unroll
                            for (int ks = 0; ks < KS32; ++ks)
                                xp[pp][ks] = *reinterpret_cast<const bf16x8*>(xl + pp * (2 * KSX * 1024) + (ks >> 1) * 1024 + (ks & 1) * 512);
                    }
                    bf16x8 xs[SINGLE ? KS1 : 1];
                    if (SINGLE) {
#pragma unroll
                        for (int kk = 0; kk < KS1; ++kk) xs[kk] = xfrag(stile, kk);
                    }
                    // 16x16x32 A fragments of W1 inside the pair image: k-step kk < KS1 - 1 reads block 2 kk + (g >> 1), the last k-step
                    // block KS32 - 1 in every lane (its upper k-slot groups meet the zeros of xfrag)
                    const int o16a = (g >> 1) * 1024 + (32 * (g & 1) + hr) * 16, o16b = (KS32 - 1) * 1024 + (32 * (g & 1) + hr) * 16;
                    bf16x8 w1[KS32], w2[DT], w1s[SINGLE ? 2 : 1][SINGLE ? KS1 : 1];
                    f32x16 hp;                                       // pair item in flight
                    f32x4 h0, h1;                                    // single item in flight
                    auto ringbuf = [&](int s_) -> const char* { return ring + ((s_ + rb) % NBUF) * WB1 + FH * NBF * 1024; };
                    auto load_w1 = [&](int s_) {
                        const char* wb = ringbuf(s_) + lperm * 16;
#pragma unroll
                        for (int ks = 0; ks < KS32; ++ks) w1[ks] = *reinterpret_cast<const bf16x8*>(wb + ks * 1024);
                    };
                    auto load_w1s = [&](int s_) {
                        if constexpr (SINGLE) {
                            const char* wb = ringbuf(s_);
#pragma unroll
                            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                                for (int kk = 0; kk < KS1; ++kk)
                                    w1s[ft][kk] = *reinterpret_cast<const bf16x8*>(wb + (kk < KS1 - 1 ? o16a + kk * 2048 : o16b) + ft * FTB);
                        }
                    };
                    auto load_w2 = [&](int s_) {
                        const char* wb = ringbuf(s_) + lane * 16;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) w2[dt] = *reinterpret_cast<const bf16x8*>(wb + (KS32 + dt) * 1024);
                    };
                    auto h_pair = [&](int pp) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        hp = z;
#pragma unroll
                        for (int ks = 0; ks < KS32; ++ks) hp = MFMA32(w1[ks], xp[pp][ks], hp);
                    };
                    auto h_single = [&]() {
                        h0 = f4zero();
                        h1 = f4zero();
                        if constexpr (SINGLE) {
#pragma unroll
                            for (int kk = 0; kk < KS1; ++kk) {
                                h0 = MFMA(w1s[0][kk], xs[kk], h0);
                                h1 = MFMA(w1s[1][kk], xs[kk], h1);
                            }
                        }
                    };
                    auto w2_pair = [&](const bf16x8& t0, const bf16x8& t1, auto a0c) {
                        constexpr int a0 = decltype(a0c)::value;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            acc[dt][a0] = MFMA(w2[dt], t0, acc[dt][a0]);
                            acc[dt][a0 + 1] = MFMA(w2[dt], t1, acc[dt][a0 + 1]);
                        }
                    };
                    load_w1(0);
                    load_w1s(0);
                    load_w2(0);
                    h_pair(0);
                    __builtin_amdgcn_sched_barrier(0);
                    constexpr int TREM = (SH::S * SH::KT) % MQ;
                    constexpr bool LIGHT_OK = FD_DMA_LIGHT && (2 * TREM == MQ) && (SH::rot == TREM);
#if FD_FFN_PRIO
                    if (LIGHT_OK && SINGLE) __builtin_amdgcn_s_setprio(1);
#endif
                    // One step, instantiated per (issues the weight DMA of step st + 3, a next step exists): every condition inside a
                    // step is a compile-time fact, so a step is ONE basic block (a run-time `st + 1 < NS` / `my turn` test splits it
                    // and the sched_group_barrier pipelines below stop at the block boundaries: measured 1514 instead of 1390 cycles)
                    auto step32 = [&](int st, auto dmac, auto nextc) {
                        constexpr bool dma_now = decltype(dmac)::value && FD_DMA_ON;
                        constexpr bool NEXT = decltype(nextc)::value;
                        if constexpr (!SINGLE) {
                            {   // item 0 = pair 0 (H in flight); H of pair 1 shadows its relu, the next step's W1 reads trail its W2
                                const f32x16 gp = hp;
                                bf16x8 t0, t1;
                                h_pair(1);
                                relu_split32(gp, t0, t1);
#pragma unroll
                                for (int q = 0; q < KS32; ++q) {
                                    SGB(SG_MFMA, 1);
                                    SGB(SG_VALU, 4);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                if (NEXT) load_w1(st + 1);
                                w2_pair(t0, t1, std::integral_constant<int, 0>{});
#pragma unroll
                                for (int q = 0; q < KS32; ++q) {
                                    SGB(SG_MFMA, 1);
                                    SGB(SG_DSR, 1);
                                }
                                SGB(SG_MFMA, 2 * DT - KS32);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            {   // item 1 = pair 1; H of the next step's pair 0 shadows its relu
                                const f32x16 gp = hp;
                                bf16x8 t0, t1;
                                if (NEXT) h_pair(0);
                                relu_split32(gp, t0, t1);
#pragma unroll
                                for (int q = 0; q < KS32; ++q) {
                                    SGB(SG_MFMA, 1);
                                    SGB(SG_VALU, 4);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                if (dma_now) issue_ffn_half(st + 3);         // (even tile splits: the F-half wave sets take turns)
                                w2_pair(t0, t1, std::integral_constant<int, 2>{});
                                if (NEXT) load_w2(st + 1);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        } else {
                            {   // item 0 = the pair (H in flight); the single tile's H shadows its relu
                                const f32x16 gp = hp;
                                bf16x8 t0, t1;
                                h_single();
                                relu_split32(gp, t0, t1);
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    SGB(SG_MFMA, 1);
                                    SGB(SG_VALU, 3);
                                }
#pragma unroll
                                for (int q = 4; q < 2 * KS1; ++q) {
                                    SGB(SG_MFMA, 1);
                                    SGB(SG_VALU, 4);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                if (NEXT) {
                                    load_w1(st + 1);
                                    load_w1s(st + 1);
                                }
                                w2_pair(t0, t1, std::integral_constant<int, acc_p0>{});
#pragma unroll
                                for (int q = 0; q < 2 * DT - 2; ++q) {
                                    SGB(SG_MFMA, 1);
                                    SGB(SG_DSR, 1);
                                }
                                SGB(SG_MFMA, 1);
                                SGB(SG_DSR, 2);
                                SGB(SG_MFMA, 1);
                                SGB(SG_DSR, 1);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            {   // item 1 = the single tile; H of the next step's pair shadows its relu, the weight DMA trails its W2
                                const f32x4 g0 = h0, g1 = h1;
                                if (NEXT) h_pair(0);
                                const bf16x8 hb = relu_pack(g0, g1);
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    SGB(SG_MFMA, 1);
                                    SGB(SG_VALU, 2);
                                }
                                SGB(SG_MFMA, KS32 - 4);
                                __builtin_amdgcn_sched_barrier(0);
                                if (dma_now) {
                                    issue_ffn_half(st + 3);
#pragma unroll
                                    for (int dt = 0; dt < DT; ++dt) acc[dt][acc_s] = MFMA(w2[dt], hb, acc[dt][acc_s]);
#pragma unroll
                                    for (int q = 0; q < DT; ++q) {
                                        SGB(SG_MFMA, 1);
                                        SGB(SG_VMEM, 1);
                                    }
                                } else {
#pragma unroll
                                    for (int dt = 0; dt < DT; ++dt) acc[dt][acc_s] = MFMA(w2[dt], hb, acc[dt][acc_s]);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                if (NEXT) load_w2(st + 1);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        if (dma_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDH) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    };
                    {
                        using Tc = std::true_type;
                        using Fc = std::false_type;
                        int st = 0;
                        if constexpr (LIGHT_OK) {               // the three-tile waves issue every step's DMA
                            for (; st < NS - 3; ++st) step32(st, std::integral_constant<bool, SINGLE>{}, Tc{});
                        } else {                                // the F-half wave sets take turns: even steps FH 0, odd steps FH 1
                            for (; st + 1 < NS - 3; st += 2) {
                                step32(st, std::integral_constant<bool, FH == 0>{}, Tc{});
                                step32(st + 1, std::integral_constant<bool, FH == 1>{}, Tc{});
                            }
                            if (st < NS - 3) {
                                step32(st, std::integral_constant<bool, FH == 0>{}, Tc{});
                                ++st;
                            }
                        }
                        for (; st < NS - 1; ++st) step32(st, Fc{}, Tc{});
                        step32(st, Fc{}, Fc{});
                    }
#if FD_FFN_PRIO
                    if (LIGHT_OK && SINGLE) __builtin_amdgcn_s_setprio(0);
#endif
                };
                if constexpr (F32) {
                    if constexpr (MT == 3) {          // 12 tiles = 3,3,3,3 (the reference's ecg series, T = 187): pair + single in every wave
                        if ((tile0 & 1) == 0) ffn_loop32(std::integral_constant<int, 1>{});
                        else ffn_loop32(std::integral_constant<int, 2>{});
                    } else if (ntile == MT) ffn_loop32(std::integral_constant<int, 0>{});
                    else if constexpr (((SH::S * SH::KT) % MQ) != 0) {
                        if ((tile0 & 1) == 0) ffn_loop32(std::integral_constant<int, 1>{});
                        else ffn_loop32(std::integral_constant<int, 2>{});
                    }
                } else {
                if (ntile == MT) ffn_loop(std::integral_constant<int, MT>{});
                else if (MT > 1 && ntile == MT - 1) ffn_loop(std::integral_constant<int, (MT > 1 ? MT - 1 : 1)>{});
                else {   // fewer tiles only happens for tiny workgroups: run the full width (extra tiles are zeros)
                    ffn_loop(std::integral_constant<int, MT>{});
                }
                }
                mark(6, step);
                refresh_lane();
                // combine the two F halves: tile tt is finalised by its owner wave ((tt & 1) == FH)
                f32x4* xch = reinterpret_cast<f32x4*>(ring);       // [mq][tt][dt][lane]
#pragma unroll
                for (int tt = 0; tt < MT; ++tt)
                    if ((tt & 1) != FH && tt < ntile) {
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) xch[((mq * MT + tt) * DT + dt) * 64 + lane] = acc[dt][tt];
                    }
                __syncthreads();
                {
                    // both owned tiles in one basic block (see the out-proj phase): a missing second tile is computed from the
                    // first one's operands and zeroed afterwards
                    bool okc[2];
#pragma unroll
                    for (int oi = 0; oi < 2; ++oi) {
                        const int ttc = FH + 2 * oi;                 // owned tile: a constant after unrolling
                        okc[oi] = ttc < MT && ttc < ntile;
                        const int tts = okc[oi] ? ttc : FH;          // (tile FH of the quarter always exists when ntile > FH)
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            const int d0 = 16 * dt + 4 * g;
                            // every path below overwrites res[oi] completely: a partial / conditional update would
                            // keep the pre-FFN residual live across the whole loop (it cost 32 spilled registers)
                            const int dr = (d0 < D) ? d0 : 0;
                            const f32x4 mine = acc[dt][ttc < MT ? ttc : 0];          // already contains the residual
                            const f32x4 other = xch[((mq * MT + tts) * DT + dt) * 64 + lane];
                            const float4 b2 = *reinterpret_cast<const float4*>(lpar + 1 * D + dr);
                            res[oi][dt][0] = mine[0] + other[0] + b2.x;              // lanes with d0 >= D: zeroed by layer_norm
                            res[oi][dt][1] = mine[1] + other[1] + b2.y;
                            res[oi][dt][2] = mine[2] + other[2] + b2.z;
                            res[oi][dt][3] = mine[3] + other[3] + b2.w;
                        }
                    }
                    layer_norm2(res[0], res[1], lpar + 4 * D, lpar + 5 * D);
#pragma unroll
                    for (int oi = 0; oi < 2; ++oi) {
                        if (okc[oi]) write_xfrags(tile0 + FH + 2 * oi, res[oi]);
                        else {
#pragma unroll
                            for (int dt = 0; dt < DT; ++dt) res[oi][dt] = f4zero();
                        }
                    }
                }
                __syncthreads();
            };
            if (fh == 0) ffn_phase(std::integral_constant<int, 0>{});
            else ffn_phase(std::integral_constant<int, 1>{});
            mark(7, step);
            refresh_lane();
        }   // layers

        // ============================ unembed (score_models.py:90) + output / reverse SDE step
        const fd_sde_step_coef cf = (P.mode == FD_MEGA_SAMPLE) ? P.steps[step] : fd_sde_step_coef{0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int oi = 0; oi < 2; ++oi) {
            const int tt = fh + 2 * oi;
            if (tt < ntile) {
                const int tile = tile0 + tt;
                int ser, t;
                bool valid;
                tile_token(tile, ser, t, valid);
                for (int ct = 0; ct < SHP(CT); ++ct) {
                    f32x4 sc = f4zero();
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) sc = MFMA(gfrag(P.img_unemb, ct * KS1 + ks), xfrag(tile, ks), sc);
                    const int c0 = 16 * ct + 4 * g;
                    if (valid && c0 < C) {
                        const size_t e0 = ((size_t)(b0 + ser) * T + t) * C + c0;      // element index in (B,T,C)
                        if (P.mode == FD_MEGA_SAMPLE) {
                            // The noise stream is the standalone fd_sde_step's: Philox counter q yields the normals of
                            // elements 4q..4q+3 of the flattened (B,T,C) array.  C % 4 == 0: this lane's 4 channels are
                            // exactly one counter and one aligned float4; otherwise they straddle two counters.
                            const float Gk = P.G[t];
                            const float gk = cf.g * Gk;
                            if ((C & 3) == 0) {
                                const float4 xv = *reinterpret_cast<const float4*>(P.x + e0);
                                float z[4];
                                if (P.z_steps) {
                                    const float4 zz = *reinterpret_cast<const float4*>(P.z_steps + (size_t)step * P.n_elem + e0);
                                    z[0] = zz.x; z[1] = zz.y; z[2] = zz.z; z[3] = zz.w;
                                } else {
                                    fd_randn4(P.offset + (uint64_t)step * P.ctr_per_step + (e0 >> 2), P.seed, z);
                                }
                                float4 o;
                                o.x = xv.x - (-cf.a_x * xv.x - (gk * gk) * sc[0]) * cf.dt + cf.sqrt_dt * (gk * z[0]);
                                o.y = xv.y - (-cf.a_x * xv.y - (gk * gk) * sc[1]) * cf.dt + cf.sqrt_dt * (gk * z[1]);
                                o.z = xv.z - (-cf.a_x * xv.z - (gk * gk) * sc[2]) * cf.dt + cf.sqrt_dt * (gk * z[2]);
                                o.w = xv.w - (-cf.a_x * xv.w - (gk * gk) * sc[3]) * cf.dt + cf.sqrt_dt * (gk * z[3]);
                                *reinterpret_cast<float4*>(P.x + e0) = o;
                            } else {
                                float za[4], zb[4];
                                const int sh = (int)(e0 & 3);
                                if (!P.z_steps) {
                                    const uint64_t q0 = P.offset + (uint64_t)step * P.ctr_per_step + (e0 >> 2);
                                    fd_randn4(q0, P.seed, za);
                                    fd_randn4(q0 + 1, P.seed, zb);
                                }
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    if (c0 + r < C) {
                                        float z;
                                        if (P.z_steps) z = P.z_steps[(size_t)step * P.n_elem + e0 + r];
                                        else z = (sh + r < 4) ? za[(sh + r) & 3] : zb[(sh + r) & 3];
                                        const float xv = P.x[e0 + r];
                                        P.x[e0 + r] = xv - (-cf.a_x * xv - (gk * gk) * sc[r]) * cf.dt + cf.sqrt_dt * (gk * z);
                                    }
                                }
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (c0 + r < C) P.score_out[e0 + r] = sc[r];
                        }
                    }
                }
            }
        }
        mark(8, step);
        if (P.prof && step == nsteps - 1 && wave == 0 && lane == 0 && blockIdx.x < 2048)
            P.prof[2 * (4100 + blockIdx.x) + 1] = wall_clock64();
        if (P.clk_out && step == nsteps - 1 && blockIdx.x == 0 && wave == 0 && lane == 0) {
            P.clk_out[2] = __builtin_readcyclecounter();
            P.clk_out[3] = wall_clock64();
        }
        __syncthreads();   // x of this step is complete before the next step's embed reads it (same wave, but
                           // also fences the fragment region against the next embed's writes)
    }
}

FD_MEGA_NS_END
