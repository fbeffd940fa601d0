// fd_mega.hip -- the persistent "series-resident" kernel: the whole score network (and, in sampler mode,
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
// Attention: per (query tile, head pair) unit, S^T = K Q^T and O^T = V^T P^T with two heads sharing every
// K / V fragment (even head in k-slots / rows of lane groups 0-1, odd head in groups 2-3), fp32 online
// softmax in registers (exp2, scale folded into W_q), P fed back as a B operand without leaving registers.
//
// Reference arithmetic: src/fdiff/models/score_models.py:67-94 (+ torch TransformerEncoderLayer),
// src/fdiff/sampling/sampler.py:83-104, src/fdiff/schedulers/sde.py:129-165,215-246.
#include "fd_mega.h"
#include "fd_philox.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

namespace {

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
    u32x4 r = {cvt_pk_bf16(relu_bits(a[0]), relu_bits(a[1])), cvt_pk_bf16(relu_bits(a[2]), relu_bits(a[3])),
               cvt_pk_bf16(relu_bits(b[0]), relu_bits(b[1])), cvt_pk_bf16(relu_bits(b[2]), relu_bits(b[3]))};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x8 frag_zero() {
    u32x4 z = {0u, 0u, 0u, 0u};
    return __builtin_bit_cast(bf16x8, z);
}
__device__ __forceinline__ f32x4 f4zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ float group_sum(float v) {      // sum over the 4 lane groups holding one token
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// ---------------------------------------------------------------------------------------------------
template <int KS1, int DT, int KSO, int MT>
__global__ __launch_bounds__(512, 2) void k_mega(const fd_mega_params P) {
    constexpr int KSX = KS1;                     // x-fragment blocks per token tile
    constexpr int NBF = 2 * KS1 + DT;            // FFN blocks per (F-half, 32-wide chunk)
    constexpr int SUB = 2;                       // FFN chunks per barrier step
    constexpr int WBUF = 2 * SUB * NBF * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tok = lane & 15, g = lane >> 4;
    const int T = P.T, KT = P.KT, D = P.D, C = P.C, H = P.H, hd = P.hd, S = P.S;
    const int NTILE = S * KT;                    // token tiles of this workgroup (16 slots each)
    const int NTOK = NTILE * 16;
    const int NP = (H + 1) >> 1;                 // head pairs
    const int NPG = P.NPG;                       // head pairs per attention group
    const int NJ = (KT + 1) >> 1;                // 32-key blocks per series
    const int b0 = blockIdx.x * S;               // first series of this workgroup

    // ---- LDS map
    char* const xfr = smem;                                   // [NTILE][KSX][64][16 B]  activation B fragments
    char* const wsl = xfr + NTILE * KSX * 1024;               // [NPG][KS1][1 KiB]        W_k / W_v / W_q slot
    char* const kbf = wsl + NPG * KS1 * 1024;                 // [NPG][NTOK][4][8 B]      K (both heads per pair)
    char* const vbf = kbf + NPG * NTOK * 32;                  // [NPG][S][NJ][4][16][16 B] V^T
    char* const afr = smem + P.lds_afr;                       // [NTILE][KSO][64][16 B]  attention-output fragments
    char* const ring = wsl;                                   // FFN weight ring + exchange alias W/K/V(/afr)
    float* const temb = reinterpret_cast<float*>(smem + P.lds_temb);   // [S][D] + emb scratch [S][D]

    // ---- token-tile ownership (same split as the FFN: quarters mq, F-halves fh; fh waves rotated)
    const int fh = wave >> 2;
    const int mq = (wave + fh * P.rot) & 3;
    const int tbase = NTILE >> 2, trem = NTILE & 3;
    const int ntile = tbase + (mq < trem ? 1 : 0);
    const int tile0 = mq * tbase + (mq < trem ? mq : trem);
    // owned tiles (residual stream lives in this wave's registers): tt = fh, fh + 2
    f32x4 res[2][DT];

    auto tile_token = [&](int tile, int& ser, int& t, bool& valid) {
        ser = tile / KT;
        t = (tile - ser * KT) * 16 + tok;
        valid = (t < T) && (b0 + ser < P.B);
    };
    auto layer_ptr = [&](int l) -> const char* { return P.img_layers + (size_t)l * P.layer_stride; };

    // LayerNorm of v (C layout, DT tiles) over the D features of token lane&15, in place
    auto layer_norm = [&](f32x4 (&v)[DT], const float* __restrict__ gamma, const float* __restrict__ beta) {
        float s = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            if (16 * dt + 4 * g < D) s += (v[dt][0] + v[dt][1]) + (v[dt][2] + v[dt][3]);
        const float mean = group_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            if (16 * dt + 4 * g < D) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float c = v[dt][r] - mean;
                    q += c * c;
                }
            }
        const float rstd = rsqrtf(group_sum(q) / (float)D + 1e-5f);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            if (d0 < D) {
                const float4 gm = *reinterpret_cast<const float4*>(gamma + d0);
                const float4 bt = *reinterpret_cast<const float4*>(beta + d0);
                v[dt][0] = (v[dt][0] - mean) * rstd * gm.x + bt.x;
                v[dt][1] = (v[dt][1] - mean) * rstd * gm.y + bt.y;
                v[dt][2] = (v[dt][2] - mean) * rstd * gm.z + bt.z;
                v[dt][3] = (v[dt][3] - mean) * rstd * gm.w + bt.w;
            } else {
                v[dt] = f4zero();
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
        for (int b = wave; b < nblk; b += 8)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + ((size_t)b * 64 + lane) * 16), LDS_PTR(dst + b * 1024), 16,
                                             0, 0);
    };

    // ---- zero the fragment region once (k padding beyond the written slots must read as 0)
    for (int i = threadIdx.x; i < NTILE * KSX * 64; i += 512) reinterpret_cast<u32x4*>(xfr)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    const int nsteps = (P.mode == FD_MEGA_SAMPLE) ? P.nsteps : 1;
    for (int step = 0; step < nsteps; ++step) {
        // ============================ time embedding (transformer.py:80-89), one wave per series
        if (wave < S) {
            const int b = b0 + wave;
            float tv = 0.f;
            if (b < P.B) tv = (P.mode == FD_MEGA_SAMPLE) ? P.steps[step].t : P.tvec[b];
            const int half = (D + 1) / 2;
            float* emb = temb + (S + wave) * D;
            for (int j = lane; j < D; j += 64) {
                const int jj = (j < half) ? j : j - half;
                const float ph = ((tv * P.params[P.tW + jj]) * 2.0f) * 3.14159274101257324f;
                emb[j] = (j < half) ? sinf(ph) : cosf(ph);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (int d = lane; d < D; d += 64) {
                float a = P.params[P.td_b + d];
                const float* w = P.params + P.td_w + (size_t)d * D;
                for (int j = 0; j < D; ++j) a = fmaf(w[j], emb[j], a);
                temb[wave * D + d] = a;
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
                for (int ks = 0; ks < P.KSE; ++ks) {
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
                    for (int dt = 0; dt < DT; ++dt) acc[dt] = MFMA(gfrag(P.img_emb, dt * P.KSE + ks), xb, acc[dt]);
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
                    res[oi][dt] = acc[dt];
                }
                write_xfrags(tile, res[oi]);
            }
        }
        // first layer's W_k for group 0 can stream while the embed finishes
        __syncthreads();

        // ============================ encoder layers
        for (int l = 0; l < P.L; ++l) {
            const char* limg = layer_ptr(l);
            const fd_mega_layer_f32 lp = P.layers[l];

            // -------- attention, one group of head pairs at a time
            for (int pg = 0; pg < NP; pg += NPG) {
                const int npg = min(NPG, NP - pg);
                // ---- K projection: K^T rows (pair-major, 8 rows per head) x all token tiles -> kbf
                dma_blocks(limg + P.off_wk + (size_t)pg * KS1 * 1024, wsl, npg * KS1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                for (int u = wave; u < npg * NTILE; u += 8) {
                    const int pr = u / NTILE, tile = u - pr * NTILE;
                    f32x4 a = f4zero();
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks)
                        a = MFMA(*reinterpret_cast<const bf16x8*>(wsl + ((pr * KS1 + ks) * 64 + lane) * 16),
                                 xfrag(tile, ks), a);
                    u32x2 pk = {cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3])};
                    *reinterpret_cast<u32x2*>(kbf + ((size_t)(pr * NTOK + tile * 16 + tok) * 4 + g) * 8) = pk;
                }
                __syncthreads();
                // ---- V projection (non-transposed: C rows = tokens) -> vbf as V^T A-fragments
                dma_blocks(limg + P.off_wv + (size_t)pg * KS1 * 1024, wsl, npg * KS1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                for (int u = wave; u < npg * NTILE; u += 8) {
                    const int pr = u / NTILE, tile = u - pr * NTILE;
                    f32x4 a = f4zero();
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks)
                        a = MFMA(xfrag(tile, ks),
                                 *reinterpret_cast<const bf16x8*>(wsl + ((pr * KS1 + ks) * 64 + lane) * 16), a);
                    // lane (col = lane&15, g) holds 4 consecutive tokens (keys) 4g+r of this tile
                    const int ser = tile / KT, kt = tile - ser * KT;
                    u32x2 pk = {cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3])};
                    char* dst = vbf + ((size_t)(((pr * S + ser) * NJ + (kt >> 1)) * 4 + g) * 16 + tok) * 16;
                    *reinterpret_cast<u32x2*>(dst + 8 * (kt & 1)) = pk;
                    if ((KT & 1) && kt == KT - 1) *reinterpret_cast<u32x2*>(dst + 8) = u32x2{0u, 0u};
                }
                __syncthreads();
                // ---- W_q, then the attention units (query tile x head pair)
                dma_blocks(limg + P.off_wq + (size_t)pg * KS1 * 1024, wsl, npg * KS1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                for (int u = wave; u < npg * NTILE; u += 8) {
                    const int pr = u / NTILE, qt = u - pr * NTILE;
                    const int ser = qt / KT;
                    f32x4 qa = f4zero();
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks)
                        qa = MFMA(*reinterpret_cast<const bf16x8*>(wsl + ((pr * KS1 + ks) * 64 + lane) * 16),
                                  xfrag(qt, ks), qa);
                    u32x4 qpk = {cvt_pk_bf16(qa[0], qa[1]), cvt_pk_bf16(qa[2], qa[3]), 0u, 0u};
                    const bf16x8 qb = __builtin_bit_cast(bf16x8, qpk);   // both heads: even in g<2, odd in g>=2
                    float o_sel[4] = {0.f, 0.f, 0.f, 0.f};
                    float l_sel = 1.f;
#pragma unroll
                    for (int hs = 0; hs < 2; ++hs) {
                        const bool mine = ((g >> 1) == hs);         // lane groups carrying this head's k-slots
                        float m = kNegBig, lsum = 0.f;
                        f32x4 o = f4zero();
                        for (int kb = 0; kb < KT; kb += 8) {
                            f32x4 s[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int kt = kb + j;
                                if (kt < KT) {
                                    const u32x2 kr = *reinterpret_cast<const u32x2*>(
                                        kbf + ((size_t)(pr * NTOK + (ser * KT + kt) * 16 + tok) * 4 + g) * 8);
                                    u32x4 kk = {mine ? kr[0] : 0u, mine ? kr[1] : 0u, 0u, 0u};
                                    s[j] = MFMA(__builtin_bit_cast(bf16x8, kk), qb, f4zero());
                                    if (kt == KT - 1) {             // keys beyond T in the ragged last tile
#pragma unroll
                                        for (int r = 0; r < 4; ++r)
                                            if (kt * 16 + 4 * g + r >= T) s[j][r] = kNegBig;
                                    }
                                } else {
                                    s[j] = f32x4{kNegBig, kNegBig, kNegBig, kNegBig};
                                }
                            }
                            float bm = kNegBig;
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                bm = fmaxf(bm, fmaxf(fmaxf(s[j][0], s[j][1]), fmaxf(s[j][2], s[j][3])));
                            bm = group_max(bm);
                            const float mnew = fmaxf(m, bm);
                            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
                            lsum *= alpha;
                            o *= alpha;
                            m = mnew;
#pragma unroll
                            for (int j = 0; j < 8; ++j)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float p = __builtin_amdgcn_exp2f(s[j][r] - mnew);
                                    s[j][r] = p;
                                    lsum += p;
                                }
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const int jb = (kb >> 1) + jj;
                                if (jb < NJ) {
                                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(
                                        vbf + ((size_t)(((pr * S + ser) * NJ + jb) * 4 + g) * 16 + tok) * 16);
                                    o = MFMA(vf, pack8(s[2 * jj], s[2 * jj + 1]), o);
                                }
                            }
                        }
                        lsum = group_sum(lsum);
                        if (mine) {                                  // rows 8*hs + [0,8) of O^T live in these lanes
                            o_sel[0] = o[0]; o_sel[1] = o[1]; o_sel[2] = o[2]; o_sel[3] = o[3];
                            l_sel = lsum;
                        }
                    }
                    const float inv = 1.0f / l_sel;
                    // head = 2*(pg+pr) + (g>>1); its 8 dims are one 16-B k-slot group of the out-proj B fragment
                    const int head = 2 * (pg + pr) + (g >> 1);
                    u32x2 pk = {cvt_pk_bf16(o_sel[0] * inv, o_sel[1] * inv), cvt_pk_bf16(o_sel[2] * inv, o_sel[3] * inv)};
                    if (head >= H || (P.dbg & 1)) pk = u32x2{0u, 0u};
                    if (head < 4 * KSO)
                        *reinterpret_cast<u32x2*>(afr + ((qt * KSO + (head >> 2)) * 64 + (head & 3) * 16 + tok) * 16 +
                                                  8 * (g & 1)) = pk;
                }
                __syncthreads();
            }

            if (P.dbg_out && blockIdx.x == 0 && l == 0 && step == 0) {      // debugging aid: dump LDS
                for (int i = threadIdx.x; i < P.dbg_bytes / 4; i += 512) P.dbg_out[i] = reinterpret_cast<unsigned*>(smem)[i];
                __syncthreads();
            }
            // -------- FFN weight stream: buffer 0 overlays W/K/V only (afr is still read by the out-proj);
            //          buffer 1 (first used at step 0 of the FFN loop) may overlay afr
            const int NS = P.F / (64 * SUB);
            auto issue_ffn = [&](int st, int buf) {
                for (int b = wave; b < 2 * SUB * NBF; b += 8) {
                    const int h = b / (SUB * NBF);
                    const int j = b - h * (SUB * NBF);
                    const char* src = limg + P.off_ffn + ((((size_t)h * NS + st) * SUB * NBF + j) * 64 + lane) * 16;
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(ring + buf * WBUF + b * 1024), 16, 0, 0);
                }
            };
            issue_ffn(0, 0);

            // -------- out-proj + residual + LayerNorm1 on the owned tiles
#pragma unroll
            for (int oi = 0; oi < 2; ++oi) {
                const int tt = fh + 2 * oi;
                if (tt < ntile) {
                    const int tile = tile0 + tt;
                    f32x4 acc[DT];
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) acc[dt] = f4zero();
#pragma unroll
                    for (int ks = 0; ks < KSO; ++ks) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(afr + ((tile * KSO + ks) * 64 + lane) * 16);
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) acc[dt] = MFMA(gfrag(limg + P.off_wo, dt * KSO + ks), af, acc[dt]);
                    }
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const int d0 = 16 * dt + 4 * g;
                        if (d0 < D) {
                            const float4 bo = *reinterpret_cast<const float4*>(P.params + lp.out_b + d0);
                            res[oi][dt][0] += acc[dt][0] + bo.x;
                            res[oi][dt][1] += acc[dt][1] + bo.y;
                            res[oi][dt][2] += acc[dt][2] + bo.z;
                            res[oi][dt][3] += acc[dt][3] + bo.w;
                        }
                    }
                    layer_norm(res[oi], P.params + lp.n1_w, P.params + lp.n1_b);
                    write_xfrags(tile, res[oi]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            // -------- FFN: hidden never leaves registers (see fd_score_bf16.hip)
            {
                bf16x8 xf[MT][KS1];
#pragma unroll
                for (int tt = 0; tt < MT; ++tt)
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) xf[tt][ks] = (tt < ntile) ? xfrag(tile0 + tt, ks) : frag_zero();
                f32x4 acc[DT][MT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int tt = 0; tt < MT; ++tt) acc[dt][tt] = f4zero();
                int buf = 0;
                for (int st = 0; st < NS; ++st) {
                    if (st + 1 < NS) issue_ffn(st + 1, buf ^ 1);
                    if (P.dbg & 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); buf ^= 1; continue; }
#pragma unroll
                    for (int sub = 0; sub < SUB; ++sub) {
                        const char* wb = ring + buf * WBUF + (fh * SUB + sub) * NBF * 1024 + lane * 16;
                        bf16x8 w1[2][KS1], w2[DT];
#pragma unroll
                        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                            for (int ks = 0; ks < KS1; ++ks)
                                w1[ft][ks] = *reinterpret_cast<const bf16x8*>(wb + (ft * KS1 + ks) * 1024);
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt)
                            w2[dt] = *reinterpret_cast<const bf16x8*>(wb + (2 * KS1 + dt) * 1024);
#pragma unroll
                        for (int tt = 0; tt < MT; ++tt) {
                            if (tt < ntile) {
                                f32x4 h0 = f4zero(), h1 = f4zero();
#pragma unroll
                                for (int ks = 0; ks < KS1; ++ks) {
                                    h0 = MFMA(w1[0][ks], xf[tt][ks], h0);
                                    h1 = MFMA(w1[1][ks], xf[tt][ks], h1);
                                }
                                const bf16x8 hb = relu_pack(h0, h1);
#pragma unroll
                                for (int dt = 0; dt < DT; ++dt) acc[dt][tt] = MFMA(w2[dt], hb, acc[dt][tt]);
                            }
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    buf ^= 1;
                }
                // combine the two F halves: tile tt is finalised by its owner wave ((tt & 1) == fh)
                f32x4* xch = reinterpret_cast<f32x4*>(ring);       // [mq][tt][dt][lane]
#pragma unroll
                for (int tt = 0; tt < MT; ++tt)
                    if (tt < ntile && (tt & 1) != fh) {
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) xch[((mq * MT + tt) * DT + dt) * 64 + lane] = acc[dt][tt];
                    }
                __syncthreads();
#pragma unroll
                for (int oi = 0; oi < 2; ++oi) {
                    // owned tile index is a compile-time function of oi only through fh (runtime): select
                    const int tt = fh + 2 * oi;
                    if (tt < ntile) {
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            const int d0 = 16 * dt + 4 * g;
                            if (d0 < D) {
                                // acc index must be compile-time: tt is 2*oi or 2*oi+1
                                const f32x4 mine = (fh == 0) ? acc[dt][(2 * oi) < MT ? 2 * oi : 0]
                                                             : acc[dt][(2 * oi + 1) < MT ? 2 * oi + 1 : 0];
                                const f32x4 other = xch[((mq * MT + tt) * DT + dt) * 64 + lane];
                                const float4 b2 = *reinterpret_cast<const float4*>(P.params + lp.l2_b + d0);
                                res[oi][dt][0] += mine[0] + other[0] + b2.x;
                                res[oi][dt][1] += mine[1] + other[1] + b2.y;
                                res[oi][dt][2] += mine[2] + other[2] + b2.z;
                                res[oi][dt][3] += mine[3] + other[3] + b2.w;
                            }
                        }
                        layer_norm(res[oi], P.params + lp.n2_w, P.params + lp.n2_b);
                        write_xfrags(tile0 + tt, res[oi]);
                    }
                }
                __syncthreads();
            }
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
                for (int ct = 0; ct < P.CT; ++ct) {
                    f32x4 sc = f4zero();
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) sc = MFMA(gfrag(P.img_unemb, ct * KS1 + ks), xfrag(tile, ks), sc);
                    const int c0 = 16 * ct + 4 * g;
                    if (valid && c0 < C) {
                        const size_t e0 = ((size_t)(b0 + ser) * T + t) * C + c0;      // element index in (B,T,C)
                        if (P.mode == FD_MEGA_SAMPLE) {
                            // C % 4 == 0 (host-checked): one Philox counter = this lane's 4 channels
                            const float4 xv = *reinterpret_cast<const float4*>(P.x + e0);
                            float z[4];
                            if (P.z_steps) {
                                const float4 zz = *reinterpret_cast<const float4*>(P.z_steps + (size_t)step * P.n_elem + e0);
                                z[0] = zz.x; z[1] = zz.y; z[2] = zz.z; z[3] = zz.w;
                            } else {
                                fd_randn4(P.offset + (uint64_t)step * P.ctr_per_step + (e0 >> 2), P.seed, z);
                            }
                            const float Gk = P.G[t];
                            const float gk = cf.g * Gk;
                            float4 o;
                            o.x = xv.x - (-cf.a_x * xv.x - (gk * gk) * sc[0]) * cf.dt + cf.sqrt_dt * (gk * z[0]);
                            o.y = xv.y - (-cf.a_x * xv.y - (gk * gk) * sc[1]) * cf.dt + cf.sqrt_dt * (gk * z[1]);
                            o.z = xv.z - (-cf.a_x * xv.z - (gk * gk) * sc[2]) * cf.dt + cf.sqrt_dt * (gk * z[2]);
                            o.w = xv.w - (-cf.a_x * xv.w - (gk * gk) * sc[3]) * cf.dt + cf.sqrt_dt * (gk * z[3]);
                            *reinterpret_cast<float4*>(P.x + e0) = o;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (c0 + r < C) P.score_out[e0 + r] = sc[r];
                        }
                    }
                }
            }
        }
        __syncthreads();   // x of this step is complete before the next step's embed reads it (same wave, but
                           // also fences the fragment region against the next embed's writes)
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------ host side
template <int KS1, int DT, int KSO, int MT>
static int launch_mega_t(fd_ctx* ctx, const fd_mega_params& P, int grid, size_t lds, hipStream_t s) {
    auto kern = k_mega<KS1, DT, KSO, MT>;
    static bool attr = false;
    if (!attr) {
        FD_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, P);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

int fd_mega_launch(fd_ctx* ctx, const fd_mega_params& P, int ks1, int dt, int kso, int mt, int grid, size_t lds,
                   hipStream_t s) {
#define FD_MEGA_CASE(K, T_, O, M_)                                                                 \
    if (ks1 == K && dt == T_ && kso == O && mt == M_) return launch_mega_t<K, T_, O, M_>(ctx, P, grid, lds, s);
#define FD_MEGA_MT(K, T_, O) FD_MEGA_CASE(K, T_, O, 1) FD_MEGA_CASE(K, T_, O, 2) FD_MEGA_CASE(K, T_, O, 3) FD_MEGA_CASE(K, T_, O, 4)
    FD_MEGA_MT(3, 5, 3)   // d_model 72, 12 heads (hydra default)
    FD_MEGA_MT(2, 4, 3)   // d_model 60, 12 heads (class default)
    FD_MEGA_MT(1, 2, 1)   // d_model 24, 4 heads
    FD_MEGA_MT(1, 1, 1)   // d_model 8, 4 heads
#undef FD_MEGA_MT
#undef FD_MEGA_CASE
    return fd_fail(ctx, FD_ERR_UNSUPPORTED, "persistent kernel not instantiated for ks1=%d dt=%d kso=%d mt=%d", ks1, dt,
                   kso, mt);
}
