// Functional prototype + micro-benchmark of k_mega's FFN phase in two decompositions (round 4):
//   MODE 16: the shipped one -- per (16-token tile, 32-hidden chunk) item: H = 6 x v_mfma_f32_16x16x32_bf16 (K = 72+1 padded to 96),
//            relu/pack, W2 = 5 x 16x16x32 (72 output rows padded to 80): 11 MFMAs = 176 cycles of matrix pipe per 16 tokens.
//   MODE 32: H of a PAIR of token tiles by 5 x v_mfma_f32_32x32x16_bf16 (M = 32 hidden, N = 32 tokens, K = 72+1 padded to 80 --
//            the 32x32 shape contracts 16 at a time, so K pads to 80 instead of 96), relu/pack, 4 x v_permlane16_swap turn the
//            32x32 C tile into the two 16x16x32 B fragments (W2's k-slots permuted to match), W2 = 2 x 5 x 16x16x32:
//            160 + 160 = 320 cycles per 32 tokens (-9 %).  A wave with an odd tile count runs its last tile in the 16x16x32
//            form reading the SAME weight images through a per-lane address map.
// One workgroup = 8 waves = the kernel's split: 4 token quarters x 2 halves of F, second wave set rotated by 2; 14 token tiles
// (2 series x T=100) -> quarters of 4,4,3,3 tiles; weights stream L2 -> LDS through the 4-deep ring filled 3 steps ahead by the
// light waves; one s_barrier per step.  Both modes are checked against a CPU restatement (bf16 operands, fp32 accumulate),
// then timed: cycles per step of the slowest wave, 1 workgroup and 256 workgroups.
//   hipcc -O3 --offload-arch=gfx950 ffn32_proto.hip -o ffn32_proto && ./ffn32_proto
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

#ifndef ILV
#define ILV 0          // 1: VALU / LDS reads / DMA issue interleaved into the MFMA streams by sched_group_barrier pipelines
#endif
#ifndef NODMA
#define NODMA 0        // ablation: no weight DMA in the loop (timing only)
#endif
#ifndef NOLDSR
#define NOLDSR 0       // ablation: weight fragments read once (timing only)
#endif
#ifndef NOBAR
#define NOBAR 0        // ablation: no per-step barrier (timing only)
#endif
#ifndef NORELU
#define NORELU 0       // ablation: no relu / pack VALU (timing only)
#endif
#ifndef PRIO
#define PRIO 0         // 1: s_setprio 1 for the waves that carry 4 tiles, 2: for the waves that carry 3
#endif
#ifndef ROLES
#define ROLES 0        // 0: fh = wave / 4, quarters rotated by 2 in the second set (shipped); 1: the OLDER wave of every SIMD is a
#endif                 //    4-tile wave (fh = (wave & 3) >> 1, mq = (wave & 1) + 2 (wave >> 2)); 2: the older wave is the 3-tile one
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define SG_VALU 0x2
#define SG_MFMA 0x8
#define SG_VMEM 0x10
#define SG_DSR 0x100

constexpr int D = 72, F = 2048, NTILE = 14, NW = 8, MQ = 4, ROT = 2, DT = 5;
constexpr int NS = F / 64;   // steps: one 32-wide chunk per F-half per step

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ unsigned relu_pk(unsigned v) {   // two bf16: a negative bf16 is a negative int16
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}
__device__ __forceinline__ bf16x8 relu_pack(f32x4 a, f32x4 b) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    if (NORELU) return __builtin_bit_cast(bf16x8, f32x4{a[0], a[1], b[0], b[1]});
    u32x4 r = {cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3]), cvt_pk_bf16(b[0], b[1]), cvt_pk_bf16(b[2], b[3])};
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, r), z));
}
__device__ __forceinline__ f32x4 f4zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// 32x32 C tile (hidden x tokens, this lane: token L%32, rows 8j + 4(L/32) + i in register 4j + i) -> relu -> the two
// 16x16x32 B fragments of token tiles 0 / 1 of the pair.  lo = rows j in {0,1}, hi = j in {2,3}; permlane16_swap exchanges
// the odd 16-lane rows of `lo` with the even rows of `hi`: afterwards `lo` holds token tile 0 in all four lane rows and `hi`
// token tile 1, lane row q carrying hidden rows rho(q, e) = 8 (2 (q&1) + (e>>2)) + 4 (q>>1) + (e&3) in k-slot e (the W2 image
// is k-permuted accordingly).
__device__ __forceinline__ void relu_split32(const f32x16& h, bf16x8& t0, bf16x8& t1) {
    if (NORELU) {
        t0 = __builtin_bit_cast(bf16x8, f32x4{h[0], h[1], h[2], h[3]});
        t1 = __builtin_bit_cast(bf16x8, f32x4{h[4], h[5], h[6], h[7]});
        return;
    }
    unsigned lo[4], hi[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        lo[d] = relu_pk(cvt_pk_bf16(h[2 * d], h[2 * d + 1]));
        hi[d] = relu_pk(cvt_pk_bf16(h[8 + 2 * d], h[8 + 2 * d + 1]));
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

struct Args {
    const char* img;          // [NS][F-half][NBF][1 KiB] chunk-major fragment image of the mode
    const float* x;           // [NTILE*16][D] fp32 activations
    float* out;               // [NTILE*16][D] fp32  W2 relu(W1 x + b1)   (no b2 / residual: the prototype checks the GEMMs)
    unsigned long long* cyc;  // [0] max over waves of the loop cycles, [1] wave 0
    int reps;
};

template <int MODE>
__global__ __launch_bounds__(NW * 64, 2) void k_ffn(const Args A) {
    constexpr int KS1 = 3;                                  // MODE 16: k-steps of 32
    constexpr int KS32 = 5;                                 // MODE 32: k-steps of 16
    constexpr int NBF = (MODE == 16) ? 2 * KS1 + DT : KS32 + DT;   // blocks per (F-half, chunk): 11 / 10
    constexpr int NBUF = 4;
    constexpr int WB1 = 2 * NBF * 1024;
    constexpr int NDH = (2 * NBF + MQ - 1) / MQ;            // DMA instructions per light wave per step
    constexpr int XB = (MODE == 16) ? NTILE * KS1 * 1024 : (NTILE / 2) * KS32 * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xfr = smem;
    char* const ring = smem + XB;

    // lane / tok / g re-derived from an opaque lane id per loop instantiation (as k_mega does per phase): otherwise hipcc hoists
    // the lane-dependent addresses of all three loop shapes in front of the branch and spills them
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
    const int fh = ROLES == 0 ? wave / MQ : (wave & 3) >> 1;
    const int mq = ROLES == 0 ? (wave + fh * ROT) % MQ : (ROLES == 1 ? (wave & 1) + 2 * (wave >> 2) : (wave & 1) + 2 * (1 - (wave >> 2)));
    constexpr int tbase = NTILE / MQ, trem = NTILE % MQ;
    const int ntile = tbase + (mq < trem ? 1 : 0);
    const int tile0 = mq * tbase + (mq < trem ? mq : trem);
    const bool light = ntile < 4;
    if (PRIO == 1 && !light) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2 && light) __builtin_amdgcn_s_setprio(1);

    // ---- activation fragments into LDS (slot D carries 1.0: the bias row of W1)
    for (int i = threadIdx.x; i < XB / 16; i += NW * 64) reinterpret_cast<u32x4*>(xfr)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (fh == 0) {
        for (int tt = 0; tt < ntile; ++tt) {
            const int tile = tile0 + tt;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d0 = 16 * dt + 4 * g;             // this lane's 4 features of row tile dt (C layout of the producers)
                u32x2 pk;
                if (d0 < D) {
                    const float* xr = A.x + (size_t)(tile * 16 + tok) * D + d0;
                    pk[0] = cvt_pk_bf16(xr[0], xr[1]);
                    pk[1] = cvt_pk_bf16(xr[2], xr[3]);
                } else {
                    pk[0] = (d0 == D) ? 0x00003F80u : 0u;
                    pk[1] = 0u;
                }
                if (MODE == 16) {
                    const int ks = dt >> 1, gd = 2 * (dt & 1) + (g >> 1);
                    *reinterpret_cast<u32x2*>(xfr + ((tile * KS1 + ks) * 64 + gd * 16 + tok) * 16 + 8 * (g & 1)) = pk;
                } else {
                    // 32x32x16 B fragment: block (pair, ks = dt), lane L = 32 (g>>1) + 16 (tile&1) + tok, byte 8 (g&1)
                    *reinterpret_cast<u32x2*>(xfr + (((tile >> 1) * KS32 + dt) * 64 + 32 * (g >> 1) + 16 * (tile & 1) + tok) * 16 +
                                              8 * (g & 1)) = pk;
                }
            }
        }
    }
    __syncthreads();

    const int st_rot = (blockIdx.x >> 3) % NS;
    auto issue_ffn = [&](int st_seq) {                      // whole buffer by all waves (prologue)
        int st = st_seq + st_rot;
        st -= (st >= NS) ? NS : 0;
        const char* src = A.img + (size_t)st * WB1 + lane * 16;
        char* dst = ring + (st_seq % NBUF) * WB1;
        for (int b = wave; b < 2 * NBF; b += NW)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + b * 1024), LDS_PTR(dst + b * 1024), 16, 0, 0);
    };
    auto issue_ffn_light = [&](int st_seq) {                // whole buffer by the four light waves
        int st = st_seq + st_rot;
        st -= (st >= NS) ? NS : 0;
        const char* src = A.img + (size_t)st * WB1 + lane * 16;
        char* dst = ring + (st_seq % NBUF) * WB1;
        const int w4 = wave % MQ;
#pragma unroll
        for (int i = 0; i < NDH; ++i) {
            int b = w4 + i * MQ;
            b -= (b >= 2 * NBF) ? MQ : 0;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + b * 1024), LDS_PTR(dst + b * 1024), 16, 0, 0);
        }
    };

    unsigned long long t_loop = 0;
    unsigned long long t0 = 0;
    auto rep_begin = [&]() {
        issue_ffn(0);
        issue_ffn(1);
        issue_ffn(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        t0 = __builtin_readcyclecounter();
    };
    auto rep_end = [&]() {
        t_loop += __builtin_readcyclecounter() - t0;
        __syncthreads();
    };
    auto step_tail = [&](int st) {
        if (light && st + 3 < NS && !NODMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDH) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!NOBAR) __builtin_amdgcn_s_barrier();
    };
    // combine the F halves through LDS, write out[token][feature] (accumulated over reps: checked with reps = 1)
    auto finish = [&](auto ntc, f32x4 (&acc)[DT][decltype(ntc)::value]) {
        constexpr int NT = decltype(ntc)::value;
        refresh_lane();
        if (lane == 0) atomicMax(&A.cyc[0], t_loop);
        if (threadIdx.x == 0) A.cyc[1] = t_loop;
        f32x4* xch = reinterpret_cast<f32x4*>(ring);
        if (fh == 1) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) xch[((mq * 4 + tt) * DT + dt) * 64 + lane] = acc[dt][tt];
        }
        __syncthreads();
        if (fh == 0 && blockIdx.x == 0) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const f32x4 o = xch[((mq * 4 + tt) * DT + dt) * 64 + lane];
                    const int d0 = 16 * dt + 4 * g;
                    if (d0 < D) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) A.out[(size_t)((tile0 + tt) * 16 + tok) * D + d0 + r] = acc[dt][tt][r] + o[r];
                    }
                }
        }
    };
    {
        if constexpr (MODE == 16) {
            auto loop16 = [&](auto ntc) {
                constexpr int NTT = decltype(ntc)::value;
                refresh_lane();
                f32x4 acc[DT][NTT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int tt = 0; tt < NTT; ++tt) acc[dt][tt] = f4zero();
                bf16x8 xf[NTT][KS1];
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks)
                        xf[tt][ks] = *reinterpret_cast<const bf16x8*>(xfr + (((tile0 + tt) * KS1 + ks) * 64 + lane) * 16);
                bf16x8 w1[2][KS1], w2[DT];
                f32x4 h0, h1;
                auto load_w1 = [&](int s) {
                    const char* wb = ring + (s % NBUF) * WB1 + fh * NBF * 1024 + lane * 16;
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                        for (int ks = 0; ks < KS1; ++ks) w1[ft][ks] = *reinterpret_cast<const bf16x8*>(wb + (ft * KS1 + ks) * 1024);
                };
                auto load_w2 = [&](int s) {
                    const char* wb = ring + (s % NBUF) * WB1 + fh * NBF * 1024 + lane * 16;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) w2[dt] = *reinterpret_cast<const bf16x8*>(wb + (2 * KS1 + dt) * 1024);
                };
                auto do_h = [&](int tt) {
                    h0 = f4zero();
                    h1 = f4zero();
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        h0 = MFMA(w1[0][ks], xf[tt][ks], h0);
                        h1 = MFMA(w1[1][ks], xf[tt][ks], h1);
                    }
                };
                for (int rep = 0; rep < A.reps; ++rep) {
                rep_begin();
                load_w1(0);
                load_w2(0);
                do_h(0);
                __builtin_amdgcn_sched_barrier(0);
                constexpr bool LIGHT = NTT < 4;
                auto step16 = [&](int st, auto dmac) {
                    constexpr bool DMA = decltype(dmac)::value && LIGHT && !NODMA;
                    if (DMA && !ILV) issue_ffn_light(st + 3);
#pragma unroll
                    for (int i = 0; i < NTT; ++i) {
                        const f32x4 g0 = h0, g1 = h1;
                        if (i + 1 < NTT) {
                            do_h(i + 1);
                        } else if (st + 1 < NS) {
                            do_h(0);
                        }
                        if (!ILV) {
                            if (i + 1 < NTT && i + 1 == NTT - 1 && st + 1 < NS && !NOLDSR) load_w1(st + 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        const bf16x8 hb = relu_pack(g0, g1);
                        if (ILV) {      // H(next) MFMAs each shadow two of relu(cur)'s VALU
                            SGB(SG_MFMA, 1); SGB(SG_VALU, 2); SGB(SG_MFMA, 1); SGB(SG_VALU, 2); SGB(SG_MFMA, 1); SGB(SG_VALU, 2);
                            SGB(SG_MFMA, 1); SGB(SG_VALU, 2); SGB(SG_MFMA, 2);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (ILV && i + 1 < NTT && i + 1 == NTT - 1 && st + 1 < NS && !NOLDSR) load_w1(st + 1);
                        if (ILV && DMA && i == 0) issue_ffn_light(st + 3);
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) acc[dt][i] = MFMA(w2[dt], hb, acc[dt][i]);
                        if (i == NTT - 1 && st + 1 < NS && !NOLDSR) load_w2(st + 1);
                        if (ILV) {      // W2 MFMAs shadow the fragment reads of the next step / the DMA issue
                            if (DMA && i == 0) { SGB(SG_MFMA, 1); SGB(SG_VMEM, 2); SGB(SG_MFMA, 1); SGB(SG_VMEM, 2); SGB(SG_MFMA, 1); SGB(SG_VMEM, 2); SGB(SG_MFMA, 2); }
                            else { SGB(SG_MFMA, 1); SGB(SG_DSR, 2); SGB(SG_MFMA, 1); SGB(SG_DSR, 2); SGB(SG_MFMA, 1); SGB(SG_DSR, 2); SGB(SG_MFMA, 2); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    step_tail(st);
                };
                int st = 0;
                for (; st < NS - 3; ++st) step16(st, std::true_type{});
                for (; st < NS; ++st) step16(st, std::false_type{});
                rep_end();
                }
                finish(std::integral_constant<int, NTT>{}, acc);
            };
            if (ntile == 4) loop16(std::integral_constant<int, 4>{});
            else loop16(std::integral_constant<int, 3>{});
        } else {
            // Per-lane address of the 16x16x32 A fragment (row a = lane&15, k-quarter kq = lane>>4) of hidden tile ft inside
            // the 32x32x16 W1 image: k-step kk -> block 2 kk + (kq>>1) (clamped to the last block: its partner k-slots meet
            // zeros in x), lane 32 (kq&1) + hr with hr = 16 (a>>2 & 1) + 4 (a>>3) + (a&3) [+ 8 for the second hidden tile]:
            // the row permutation that makes pack8(h0, h1) the W2 image's k-slot order rho.
            // SHAPE: 0 = two pairs (tiles tile0..+3), 1 = pair then single (tile0 even, 3 tiles), 2 = single then pair (tile0 odd)
            auto loop32 = [&](auto shc) {
                constexpr int SHAPE = decltype(shc)::value;
                refresh_lane();
                const int a16 = lane & 15;
                const int hr = 16 * ((a16 >> 2) & 1) + 4 * (a16 >> 3) + (a16 & 3);
                auto w1_16_off = [&](int ft, int kk) -> int {
                    int blk = 2 * kk + (g >> 1);
                    blk = blk > KS32 - 1 ? KS32 - 1 : blk;
                    return blk * 1024 + (32 * (g & 1) + hr + 8 * ft) * 16;
                };
                // x fragment of a single tile for the 16x16x32 form, from the pair image: block 2 kk + (g>>1), lane
                // 32 (g&1) + 16 (tile&1) + tok; k-slots >= 80 read as zero
                auto xfrag16 = [&](int tile, int kk) -> bf16x8 {
                    const int blk = 2 * kk + (g >> 1);
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(
                        xfr + (((tile >> 1) * KS32 + (blk > KS32 - 1 ? KS32 - 1 : blk)) * 64 + 32 * (g & 1) + 16 * (tile & 1) + tok) * 16);
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    return blk > KS32 - 1 ? __builtin_bit_cast(bf16x8, z) : v;
                };
                constexpr int NPAIR = SHAPE == 0 ? 2 : 1;
                constexpr bool SINGLE = SHAPE != 0;
                constexpr int NT = SINGLE ? 3 : 4;
                f32x4 acc[DT][NT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) acc[dt][tt] = f4zero();
                const int pair0 = (SHAPE == 2 ? tile0 + 1 : tile0) >> 1;       // first pair tile
                const int stile = SHAPE == 1 ? tile0 + 2 : tile0;              // the single tile
                constexpr int acc_p0 = SHAPE == 2 ? 1 : 0;                     // acc index of the first pair's first tile
                constexpr int acc_s = SHAPE == 1 ? 2 : 0;
                bf16x8 xp[NPAIR][KS32];
#pragma unroll
                for (int p = 0; p < NPAIR; ++p)
#pragma unroll
                    for (int ks = 0; ks < KS32; ++ks)
                        xp[p][ks] = *reinterpret_cast<const bf16x8*>(xfr + (((pair0 + p) * KS32 + ks) * 64 + lane) * 16);
                bf16x8 xs[SINGLE ? 3 : 1];
                if (SINGLE) {
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) xs[kk] = xfrag16(stile, kk);
                }
                // two per-lane bases + immediates: k-steps 0, 1 read blocks (g>>1), 2 + (g>>1); k-step 2 reads block 4 in every lane
                const int o16a = (g >> 1) * 1024 + (32 * (g & 1) + hr) * 16, o16b = 4 * 1024 + (32 * (g & 1) + hr) * 16;
                bf16x8 w1[KS32], w2[DT], w1s[SINGLE ? 2 : 1][SINGLE ? 3 : 1];
                f32x16 hp;                                                     // pair item in flight
                f32x4 h0, h1;                                                  // single item in flight
                auto load_w1 = [&](int s) {
                    const char* wb = ring + (s % NBUF) * WB1 + fh * NBF * 1024;
#pragma unroll
                    for (int ks = 0; ks < KS32; ++ks) w1[ks] = *reinterpret_cast<const bf16x8*>(wb + ks * 1024 + lane * 16);
                };
                auto load_w1s = [&](int s) {
                    const char* wb = ring + (s % NBUF) * WB1 + fh * NBF * 1024;
                    if (SINGLE) {
#pragma unroll
                        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                            for (int kk = 0; kk < 3; ++kk)
                            w1s[ft][kk] = *reinterpret_cast<const bf16x8*>(wb + (kk < 2 ? o16a + kk * 2048 : o16b) + ft * 128);
                    }
                };
                auto load_w2 = [&](int s) {
                    const char* wb = ring + (s % NBUF) * WB1 + fh * NBF * 1024 + lane * 16;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) w2[dt] = *reinterpret_cast<const bf16x8*>(wb + (KS32 + dt) * 1024);
                };
                auto h_pair = [&](int p) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    hp = z;
#pragma unroll
                    for (int ks = 0; ks < KS32; ++ks) hp = MFMA32(w1[ks], xp[p][ks], hp);
                };
                auto h_single = [&]() {
                    h0 = f4zero();
                    h1 = f4zero();
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) {
                        h0 = MFMA(w1s[0][kk], xs[kk], h0);
                        h1 = MFMA(w1s[1][kk], xs[kk], h1);
                    }
                };
                for (int rep = 0; rep < A.reps; ++rep) {
                rep_begin();
                load_w1(0);
                load_w1s(0);
                load_w2(0);
                h_pair(0);
                __builtin_amdgcn_sched_barrier(0);
                auto w2_pair_mfma = [&](const bf16x8& t0, const bf16x8& t1, auto a0c) {
                    constexpr int a0 = decltype(a0c)::value;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        acc[dt][a0] = MFMA(w2[dt], t0, acc[dt][a0]);
                        acc[dt][a0 + 1] = MFMA(w2[dt], t1, acc[dt][a0 + 1]);
                    }
                };
                auto step32 = [&](int st, auto dmac) {
                    constexpr bool DMA = decltype(dmac)::value && SINGLE && !NODMA;
                    if (DMA && !ILV) issue_ffn_light(st + 3);
                    if constexpr (!SINGLE) {
                        // item 0 = pair 0 (in flight), item 1 = pair 1
                        {
                            const f32x16 gp = hp;
                            bf16x8 t0, t1;
                            h_pair(1);
                            if (!ILV) {
                                if (st + 1 < NS && !NOLDSR) load_w1(st + 1);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            relu_split32(gp, t0, t1);
                            if (ILV) { SGB(SG_MFMA, 1); SGB(SG_VALU, 4); SGB(SG_MFMA, 1); SGB(SG_VALU, 4); SGB(SG_MFMA, 1); SGB(SG_VALU, 4); SGB(SG_MFMA, 1); SGB(SG_VALU, 4); SGB(SG_MFMA, 1); SGB(SG_VALU, 4); }
                            __builtin_amdgcn_sched_barrier(0);
                            if (ILV && st + 1 < NS && !NOLDSR) load_w1(st + 1);
                            w2_pair_mfma(t0, t1, std::integral_constant<int, 0>{});
                            if (ILV) { SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 5); }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        {
                            const f32x16 gp = hp;
                            bf16x8 t0, t1;
                            if (st + 1 < NS) h_pair(0);
                            if (!ILV) __builtin_amdgcn_sched_barrier(0);
                            relu_split32(gp, t0, t1);
                            if (ILV) { SGB(SG_MFMA, 1); SGB(SG_VALU, 4); SGB(SG_MFMA, 1); SGB(SG_VALU, 4); SGB(SG_MFMA, 1); SGB(SG_VALU, 4); SGB(SG_MFMA, 1); SGB(SG_VALU, 4); SGB(SG_MFMA, 1); SGB(SG_VALU, 4); }
                            __builtin_amdgcn_sched_barrier(0);
                            w2_pair_mfma(t0, t1, std::integral_constant<int, 2>{});
                            if (st + 1 < NS && !NOLDSR) load_w2(st + 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                        // item 0 = the pair (in flight), item 1 = the single tile
                        {
                            const f32x16 gp = hp;
                            bf16x8 t0, t1;
                            h_single();
                            if (!ILV) {
                                if (st + 1 < NS && !NOLDSR) {
                                    load_w1(st + 1);
                                    load_w1s(st + 1);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            relu_split32(gp, t0, t1);
                            if (ILV) { SGB(SG_MFMA, 1); SGB(SG_VALU, 3); SGB(SG_MFMA, 1); SGB(SG_VALU, 3); SGB(SG_MFMA, 1); SGB(SG_VALU, 3); SGB(SG_MFMA, 1); SGB(SG_VALU, 3); SGB(SG_MFMA, 1); SGB(SG_VALU, 4); SGB(SG_MFMA, 1); SGB(SG_VALU, 4); }
                            __builtin_amdgcn_sched_barrier(0);
                            if (ILV && st + 1 < NS && !NOLDSR) {
                                load_w1(st + 1);
                                load_w1s(st + 1);
                            }
                            w2_pair_mfma(t0, t1, std::integral_constant<int, acc_p0>{});
                            if (ILV) { SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); SGB(SG_MFMA, 1); SGB(SG_DSR, 2); SGB(SG_MFMA, 1); SGB(SG_DSR, 1); }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        {
                            const f32x4 g0 = h0, g1 = h1;
                            if (st + 1 < NS) h_pair(0);
                            if (!ILV) __builtin_amdgcn_sched_barrier(0);
                            const bf16x8 hb = relu_pack(g0, g1);
                            if (ILV) { SGB(SG_MFMA, 1); SGB(SG_VALU, 2); SGB(SG_MFMA, 1); SGB(SG_VALU, 2); SGB(SG_MFMA, 1); SGB(SG_VALU, 2); SGB(SG_MFMA, 1); SGB(SG_VALU, 2); SGB(SG_MFMA, 1); }
                            __builtin_amdgcn_sched_barrier(0);
                            if (ILV && DMA) issue_ffn_light(st + 3);
#pragma unroll
                            for (int dt = 0; dt < DT; ++dt) acc[dt][acc_s] = MFMA(w2[dt], hb, acc[dt][acc_s]);
                            if (ILV && DMA) { SGB(SG_MFMA, 1); SGB(SG_VMEM, 1); SGB(SG_MFMA, 1); SGB(SG_VMEM, 1); SGB(SG_MFMA, 1); SGB(SG_VMEM, 1); SGB(SG_MFMA, 1); SGB(SG_VMEM, 1); SGB(SG_MFMA, 1); SGB(SG_VMEM, 1); }
                            __builtin_amdgcn_sched_barrier(0);
                            if (st + 1 < NS && !NOLDSR) load_w2(st + 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    step_tail(st);
                };
                int st = 0;
                for (; st < NS - 3; ++st) step32(st, std::true_type{});
                for (; st < NS; ++st) step32(st, std::false_type{});
                rep_end();
                }
                finish(std::integral_constant<int, NT>{}, acc);
            };
            if (ntile == 4) loop32(std::integral_constant<int, 0>{});
            else if ((tile0 & 1) == 0) loop32(std::integral_constant<int, 1>{});
            else loop32(std::integral_constant<int, 2>{});
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short b) {
    unsigned u = (unsigned)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

int main() {
    const int NTOK = NTILE * 16;
    std::vector<float> x((size_t)NTOK * D), W1((size_t)F * D), b1(F), W2((size_t)D * F);
    srand(7);
    for (auto& v : x) v = frand();
    for (auto& v : W1) v = frand() * 0.2f;
    for (auto& v : b1) v = frand() * 0.2f;
    for (auto& v : W2) v = frand() * 0.05f;
    // CPU restatement: bf16 operands, fp32 (double here) accumulate, hidden rounded to bf16 after relu
    std::vector<float> ref((size_t)NTOK * D);
    {
        std::vector<float> xb(x.size()), w1b(W1.size()), w2b(W2.size()), b1b(F);
        for (size_t i = 0; i < x.size(); ++i) xb[i] = bf2f(f2bf(x[i]));
        for (size_t i = 0; i < W1.size(); ++i) w1b[i] = bf2f(f2bf(W1[i]));
        for (size_t i = 0; i < W2.size(); ++i) w2b[i] = bf2f(f2bf(W2[i]));
        for (int i = 0; i < F; ++i) b1b[i] = bf2f(f2bf(b1[i]));
        std::vector<float> h(F);
        for (int t = 0; t < NTOK; ++t) {
            for (int f = 0; f < F; ++f) {
                double a = b1b[f];
                for (int d = 0; d < D; ++d) a += (double)w1b[(size_t)f * D + d] * xb[(size_t)t * D + d];
                h[f] = bf2f(f2bf(a > 0 ? (float)a : 0.f));
            }
            for (int d = 0; d < D; ++d) {
                double a = 0;
                for (int f = 0; f < F; ++f) a += (double)w2b[(size_t)d * F + f] * h[f];
                ref[(size_t)t * D + d] = (float)a;
            }
        }
    }
    auto w1k = [&](int f, int k) -> unsigned short { return k < D ? f2bf(W1[(size_t)f * D + k]) : (k == D ? f2bf(b1[f]) : 0); };
    auto w2k = [&](int d, int f) -> unsigned short { return d < D ? f2bf(W2[(size_t)d * F + f]) : 0; };
    // images: [st][fh][block][lane][8 bf16]; hidden base of (st, fh) = fh * F/2 + 32 st
    std::vector<unsigned short> img16((size_t)NS * 2 * 11 * 512), img32((size_t)NS * 2 * 10 * 512);
    for (int st = 0; st < NS; ++st)
        for (int fh = 0; fh < 2; ++fh) {
            const int hb = fh * (F / 2) + 32 * st;
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    // MODE 16
                    for (int ft = 0; ft < 2; ++ft)
                        for (int ks = 0; ks < 3; ++ks)
                            img16[((((size_t)st * 2 + fh) * 11 + ft * 3 + ks) * 64 + l) * 8 + e] = w1k(hb + 16 * ft + (l & 15), 32 * ks + 8 * (l >> 4) + e);
                    for (int dt = 0; dt < DT; ++dt) {
                        const int gg = l >> 4, hid = e < 4 ? 4 * gg + e : 16 + 4 * gg + (e - 4);
                        img16[((((size_t)st * 2 + fh) * 11 + 6 + dt) * 64 + l) * 8 + e] = w2k(16 * dt + (l & 15), hb + hid);
                    }
                    // MODE 32
                    for (int ks = 0; ks < 5; ++ks)
                        img32[((((size_t)st * 2 + fh) * 10 + ks) * 64 + l) * 8 + e] = w1k(hb + (l & 31), 16 * ks + 8 * (l >> 5) + e);
                    for (int dt = 0; dt < DT; ++dt) {
                        const int q = l >> 4, hid = 8 * (2 * (q & 1) + (e >> 2)) + 4 * (q >> 1) + (e & 3);
                        img32[((((size_t)st * 2 + fh) * 10 + 5 + dt) * 64 + l) * 8 + e] = w2k(16 * dt + (l & 15), hb + hid);
                    }
                }
        }
    char *d16, *d32;
    float *dx, *dout;
    unsigned long long* dc;
    hipMalloc(&d16, img16.size() * 2);
    hipMalloc(&d32, img32.size() * 2);
    hipMalloc(&dx, x.size() * 4);
    hipMalloc(&dout, x.size() * 4);
    hipMalloc(&dc, 64);
    hipMemcpy(d16, img16.data(), img16.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d32, img32.data(), img32.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_ffn<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_ffn<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int mode : {16, 32}) {
        const size_t lds = (mode == 16 ? NTILE * 3 * 1024 + 4 * 22 * 1024 : (NTILE / 2) * 5 * 1024 + 4 * 20 * 1024);
        Args a{mode == 16 ? d16 : d32, dx, dout, dc, 1};
        hipMemset(dout, 0, x.size() * 4);
        if (mode == 16) hipLaunchKernelGGL(k_ffn<16>, dim3(1), dim3(512), lds, 0, a);
        else hipLaunchKernelGGL(k_ffn<32>, dim3(1), dim3(512), lds, 0, a);
        std::vector<float> got(x.size());
        hipError_t e = hipMemcpy(got.data(), dout, x.size() * 4, hipMemcpyDeviceToHost);
        double me = 0, ms = 0;
        for (size_t i = 0; i < got.size(); ++i) {
            me = fmax(me, fabs((double)got[i] - ref[i]));
            ms = fmax(ms, fabs((double)ref[i]));
        }
        printf("MODE %d: %s, max |err| / max |ref| = %.3e (%s)\n", mode, hipGetErrorString(e), me / ms, me / ms < 2e-3 ? "OK" : "WRONG");
        for (int nwg : {1, 256}) {
            const int reps = 20;
            a.reps = reps;
            for (int it = 0; it < 2; ++it) {
                hipMemset(dc, 0, 64);
                if (mode == 16) hipLaunchKernelGGL(k_ffn<16>, dim3(nwg), dim3(512), lds, 0, a);
                else hipLaunchKernelGGL(k_ffn<32>, dim3(nwg), dim3(512), lds, 0, a);
                hipDeviceSynchronize();
            }
            unsigned long long h[2];
            hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
            printf("   %3d workgroup(s): %8.1f cycles per step (slowest wave), wave 0 %8.1f   [matrix pipe per SIMD and step: %d]\n", nwg,
                   (double)h[0] / (reps * NS), (double)h[1] / (reps * NS), mode == 16 ? 7 * 176 : 3 * 320 + 176);
        }
    }
    return 0;
}
