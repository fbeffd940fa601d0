// fd_train_dev.h -- device-side pieces shared by the bf16 training kernels (fd_train_bf16.hip: the five per-layer kernels;
// fd_train_persist.hip: the persistent forward).  Operand conventions, layouts and the F-split hand-over are described where
// they are defined below; everything lives in an anonymous namespace (each translation unit gets its own inlined copy).
#pragma once
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "fd_bf16_images.h"
#include "fd_philox.h"
#include "fd_score.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte access at a dword-aligned address
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#ifndef FD_TR_WG_NBUF
#define FD_TR_WG_NBUF 4          // LDS ring of k_tr_wgrad: 13-KiB stage records, NBUF - 1 blocks in flight (3 / 4 / 5: same time)
#endif
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
// ds_read_b64_tr_b16: the 16 lanes of a group hand in 16 8-byte-aligned addresses, together a [4][16] bf16 matrix (row j =
// the four runs of lanes 4j..4j+3); lane i of the group gets column i (element j = row j).
__device__ __forceinline__ s16x4 lds_read_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}


namespace {

constexpr float kNegBig = -1.0e30f;
constexpr int TW = 8;            // waves per token-parallel workgroup (one 16-token tile each)

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
__device__ __forceinline__ s16x4 pack4(f32x4 a) {
    u32x2 r = {cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3])};
    return __builtin_bit_cast(s16x4, r);
}
__device__ __forceinline__ f32x4 f4zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ bf16x8 frag_zero() {
    u32x4 z = {0u, 0u, 0u, 0u};
    return __builtin_bit_cast(bf16x8, z);
}
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
__device__ __forceinline__ float group_sum(float v) {      // over the 4 lane groups (same lane&15)
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
template <int N>
__device__ __forceinline__ float row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float v) {      // over the 16 lanes of a row (fixed order: deterministic)
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return v;
}

}  // namespace
// (global: the two translation units of the training path hand it to each other)
struct TrDims {
    int B, T, M, D, F, H, hd, NP, KT, NJ, NFT, RBW;   // NFT = 16*DT feature rows of a T-block, RBW = 32*KS1 slots of a row
    float p, keep_scale;
    unsigned thr16;
    unsigned long long seed;
    int xcd;                  // 1: workgroup ids are re-dealt so that neighbours in (x, y) order share an XCD (xcd_deal)
    int fsplit;               // FFN kernels: 1, or 2 = the hidden dimension of a 64-token block split over a PAIR of workgroups (small M)
    int norot;                // 1: every workgroup of the forward FFN kernels walks the F chunks in the natural order (FDIFF_TR_ROT=0:
                              // the bit-for-bit comparison of the persistent forward with the per-layer kernels needs one summation order)
};
// ---- the persistent training forward (fd_train_persist.hip: one launch for every encoder layer), driven by tr_forward_t
struct fd_trp_args {
    // buffers of layer 0; the same buffer of layer l lies l * lstride bytes further (tr_carve takes every layer's buffers in one loop)
    const float* x0; const __bf16* x0rb; __bf16* x0T;
    float* att; __bf16* attT; float* lse2; float* s1; float* s2; char* stage;
    unsigned char* active;
    const unsigned char* pmask; const unsigned char* hkeep; const unsigned char* rb1; const unsigned char* rb3;
    size_t lstride;
    float* hL;                               // output of the last layer
    const char* limg; size_t limg_stride;    // weight images of layer 0, bytes per layer
    size_t off_wk, off_wv, off_wq, off_wo, off_ffn;
    const float* P; long long pstride;       // fp32 parameters; floats per layer
    long long o_bo, o_g1, o_be1, o_b2, o_g2, o_be2;      // layer 0's out_proj.bias, norm1.weight / bias, linear2.bias, norm2.weight / bias
    int L, l0, l1;                           // layers of the model; layers [l0, l1) of this launch
    int b0;                                  // first series of this launch
    int Mpad;
    unsigned long long* xflag;               // [series][token tile]: epoch * 64 + (layers published)
    const unsigned long long* mflag;         // [layer]: epoch once the layer's dropout decisions are written (null: no dropout)
    unsigned long long epoch;
    unsigned* err; unsigned* err_gpu; unsigned long long timeout;      // the context's error word (host-mapped) and its device copy, bound of a wait in 100 MHz ticks
    int stall;                               // test hook: no tile flag is ever raised
};
size_t fd_trp_lds_bytes(int ks1, int dt, int kso, int NT, int T, int F, int NP);
int fd_trp_tiles(const fd_score* m, int B, int* nq_out, int* series_per_launch);
// `done` (may be null): an event completed by the LAST launch of the call (hipExtLaunchKernelGGL's stop event)
int fd_trp_forward(fd_score* m, const TrDims& d, fd_trp_args a, int NT, int nq, int series_per_launch, hipStream_t s, hipEvent_t done);
int fd_trp_set_flag(fd_ctx* ctx, unsigned long long* flag, unsigned long long value, hipStream_t s);
namespace {

// F-split of the FFN kernels (fsplit == 2).  A 64-token workgroup of k_tr_ffn_fwd / k_tr_ffn_bwd is a chain of F / 64 barrier
// steps whatever the token count: with M = 6400 tokens (T = 100, B = 64) 100 workgroups hold 100 of the 256 CUs for 40 us.
// Workgroups 2 i (producer) and 2 i + 1 (finisher) share token block i and take half of the chunk steps each; both run the
// prologue, the producer's owner waves hand their partial accumulators over through global memory (one flag per token tile,
// set to the launch's epoch behind an agent-scope release) and leave, the finisher adds them (own half + partner's half, a
// fixed order) and runs the epilogue.  The finisher has the HIGHER workgroup id, so its producer was dispatched before it, and
// the host enables the split only when 2 x blocks <= CUs (every workgroup finds a CU without another one of the grid retiring).
struct FSplit {
    float* ypart;             // [blocks][4 tiles][DT][64 lanes] f32x4
    unsigned* flag;           // [blocks][4 tiles]
    unsigned epoch;           // unique per launch
    unsigned* err;            // host-visible error word (pinned, mapped): 0, or 0x80000000 | (block << 2 | tile) of the first hand-over
                              // that timed out -- the host checks it at the entry of the next training call (fd_train_async_check)
    unsigned* err_gpu;        // device-resident copy of the word: the optimizer kernel skips its update when it is set (stream order)
    unsigned long long timeout;   // bound of the finisher's wait in ticks of the constant 100 MHz clock (s_memrealtime)
    int fence;                // 1: release / acquire fences instead of per-element coherent accesses (FDIFF_TR_FSPLIT_FENCE=1)
    int stall;                // test hook (FDIFF_TR_FSPLIT_TEST_STALL=1): the producer never raises its flags
};
// The hand-over moves 5 KiB per token tile between two workgroups that may sit on different XCDs (separate, mutually
// non-coherent L2s).  An agent-scope release / acquire FENCE would write back / invalidate the whole L2 of the XCD (measured:
// both FFN kernels at 1.4 x their unsplit time; that form stays selectable, `fence`); instead every element is itself an
// agent-scope relaxed atomic access -- a write-through store (sc1) / an L2-bypassing load -- and the flag is stored once the
// element stores have been acknowledged.
// Why this orders the data without a fence (it is outside the HIP memory model, which only speaks of fences and
// acquire / release; it rests on the gfx950 memory pipeline): (1) an agent-scope atomic store is written through to the
// memory-side coherence point shared by all XCDs and is counted in vmcnt until that write is ACKNOWLEDGED, so after
// `s_waitcnt vmcnt(0)` every element is visible to any agent-scope access from any XCD; (2) the flag store is issued only
// after that wait (the asm statement is a compiler barrier and the hardware issues in order); (3) the finisher's element
// loads are issued after the loop that saw the flag (control dependence on a loaded value + compiler barrier), and as
// agent-scope atomic loads they bypass its own XCD's non-coherent L2 lines.
// The finisher's wait is BOUNDED: a producer that is never scheduled (a CU-masked queue, a partition mode with fewer CUs than
// 2 x blocks, a co-tenant kernel that never retires) or that faulted would otherwise hang the device without a diagnostic.  After
// `timeout` ticks the finisher records (block, tile) in the error word and carries on with its own half -- the step's gradients
// are then wrong, and the next training call on the context fails with FD_ERR_STATE naming the block.
template <int DT>
__device__ __forceinline__ void fsplit_hand_over(const FSplit& fs, int blk, int tile, int lane, const f32x4 (&acc)[DT]) {
    float* yp = fs.ypart + (((size_t)(blk * 4 + tile) * DT) * 64 + lane) * 4;
    if (fs.fence) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(yp + dt * 256) = acc[dt];
        if (fs.stall) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(fs.flag + blk * 4 + tile, fs.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) __hip_atomic_store(yp + dt * 256 + r, acc[dt][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && !fs.stall) __hip_atomic_store(fs.flag + blk * 4 + tile, fs.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int DT>
__device__ __forceinline__ void fsplit_take_over(const FSplit& fs, int blk, int tile, int lane, f32x4 (&acc)[DT]) {
    if (__hip_atomic_load(fs.flag + blk * 4 + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fs.epoch) {
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0u;
        bool seen = false;
        for (;;) {
            __builtin_amdgcn_s_sleep(16);
            if (__hip_atomic_load(fs.flag + blk * 4 + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fs.epoch) { seen = true; break; }
            if ((++spins & 63u) == 0u && wall_clock64() - t0 > fs.timeout) break;
        }
        if (!seen) {          // (wave-uniform: the flag address and the clock are)
            if (lane == 0) {
                unsigned expected = 0u;
                __hip_atomic_compare_exchange_strong(fs.err, &expected, 0x80000000u | (unsigned)(blk * 4 + tile), __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(fs.err_gpu, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    }
    const float* yp = fs.ypart + (((size_t)(blk * 4 + tile) * DT) * 64 + lane) * 4;
    if (fs.fence) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] += *reinterpret_cast<const f32x4*>(yp + dt * 256);
        return;
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[dt][r] += __hip_atomic_load(yp + dt * 256 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own L2.  Workgroups that read the
// same rows -- the heads of one series in the attention kernels, the 20 role workgroups of one token split in k_tr_wgrad --
// are therefore spread over all eight L2s, and every L2 fetches every row from the Infinity Cache.  Re-dealing the ids (XCD k
// takes the k-th contiguous run of the virtual (x, y) order) keeps such a group on one XCD, or on two where a run ends inside it.
__device__ __forceinline__ void xcd_deal(const TrDims& d, int& bx, int& by) {
    bx = blockIdx.x; by = blockIdx.y;
    if (!d.xcd) return;
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int k = lin & 7, slot = lin >> 3, q = nwg >> 3, r = nwg & 7;
    const int v = k * q + (k < r ? k : r) + slot;
    by = v / (int)gridDim.x;
    bx = v - by * (int)gridDim.x;
}

// ------------------------------------------------------------------------------------------------ shared device pieces
// C-layout tile (features 16dt+4g+r of token `m`) <- fp32 rows.  Unconditional loads from clamped addresses (row 0 for an
// invalid token, D % 4 == 0) and a select afterwards: a load inside a divergent branch makes hipcc wait for it (vmcnt(0)) at
// the end of the branch -- one exposed L2 round trip per row tile, ~25 of them in the prologue of k_tr_ffn_bwd.
template <int DT>
__device__ __forceinline__ void load_ctile(const float* __restrict__ base, int m, bool valid, int D, int g, f32x4 (&v)[DT]) {
    const int mc = valid ? m : 0;
    float4 raw[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d0 = 16 * dt + 4 * g;
        raw[dt] = *reinterpret_cast<const float4*>(base + (size_t)mc * D + (d0 < D ? d0 : D - 4));
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const bool ok = valid && (16 * dt + 4 * g < D);
        v[dt] = f32x4{ok ? raw[dt].x : 0.f, ok ? raw[dt].y : 0.f, ok ? raw[dt].z : 0.f, ok ? raw[dt].w : 0.f};
    }
}
template <int DT>
__device__ __forceinline__ void store_ctile(float* __restrict__ base, int m, bool valid, int D, int g, const f32x4 (&v)[DT]) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d0 = 16 * dt + 4 * g;
        if (valid && d0 < D) *reinterpret_cast<float4*>(base + (size_t)m * D + d0) = float4{v[dt][0], v[dt][1], v[dt][2], v[dt][3]};
    }
}
// the same tile from a bf16 (M, D) tensor (the per-head partial tensors of d x, k_tr_attn_bwd OH form): 8-byte loads
template <int DT>
__device__ __forceinline__ void load_ctile_bf16(const __bf16* __restrict__ base, int m, bool valid, int D, int g, f32x4 (&v)[DT]) {
    const int mc = valid ? m : 0;
    u32x2 raw[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d0 = 16 * dt + 4 * g;
        raw[dt] = *reinterpret_cast<const u32x2*>(base + (size_t)mc * D + (d0 < D ? d0 : D - 4));
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const bool ok = valid && (16 * dt + 4 * g < D);
        const unsigned lo = ok ? raw[dt][0] : 0u, hi = ok ? raw[dt][1] : 0u;
        v[dt] = f32x4{__builtin_bit_cast(float, lo << 16), __builtin_bit_cast(float, lo & 0xffff0000u),
                      __builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    }
}
// T-block store: feature rows 16dt+4g+r, column m&31; `ones` puts 1.0 into row D (bias column of the weight gradients).
// Every row of the 16*DT block rows is written for this token (pads as 0), invalid tokens write zeros.
template <int DT>
__device__ __forceinline__ void store_T(__bf16* __restrict__ tb, int m, bool valid, int D, int g, const f32x4 (&v)[DT], bool ones) {
    const int NFT = 16 * DT;
    __bf16* col = tb + ((size_t)(m >> 5) * NFT) * 32 + (m & 31);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * dt + 4 * g + r;
            float x = 0.f;
            if (valid) x = (f < D) ? v[dt][r] : ((f == D && ones) ? 1.0f : 0.f);
            col[(size_t)f * 32] = (__bf16)x;
        }
}
// Row store (k-slot layout of the weight images): slots d < D, slot D = 1.0 when `ones`, zero padding up to 32*KS1.
template <int DT, int KS1>
__device__ __forceinline__ void store_rows(__bf16* __restrict__ rb, int m, bool valid, int D, int g, const f32x4 (&v)[DT], bool ones) {
    __bf16* row = rb + (size_t)m * (32 * KS1);
#pragma unroll
    for (int dt = 0; dt < 2 * KS1; ++dt) {
        const int d0 = 16 * dt + 4 * g;
        u32x2 pk = {0u, 0u};
        if (valid) {
            if (dt < DT && d0 < D) {
                pk[0] = cvt_pk_bf16(v[dt < DT ? dt : 0][0], v[dt < DT ? dt : 0][1]);
                pk[1] = cvt_pk_bf16(v[dt < DT ? dt : 0][2], v[dt < DT ? dt : 0][3]);
            } else if (d0 == D && ones) {
                pk[0] = 0x00003F80u;
            }
        }
        *reinterpret_cast<u32x2*>(row + d0) = pk;
    }
}
// "Stage" layout of the operands the weight-gradient kernel streams through LDS: per 32-token block ONE contiguous record
//   [x1 rows 32 x RBS][d f rows 32 x RBS]   (bf16, record padded to 1 KiB: 13 KiB at d_model 72)
// with the row stride padded off the LDS bank period (RBS = 32 KS1 + 8 -> 16-lane b128 reads hit every bank once; the
// natural stride of 192 B made every fragment read an 8-way conflict, 2.7 us per 32-token block).  global_load_lds copies
// a record verbatim.  The feature-major ("T-block") operands of the d W products are NOT stored: k_tr_wgrad reads them out
// of the same rows with ds_read_b64_tr_b16 (a 16-lane group reads a [4 tokens][16 features] block transposed), which
// halved the record, the staging DMA of k_tr_wgrad and the epilogue stores of k_tr_ffn_fwd / k_tr_ffn_bwd.
template <int KS1, int DT>
struct StageL {
    static constexpr int RBS = 32 * KS1 + 8, NFT = 16 * DT;
    static constexpr int off_xr = 0, off_dr = 32 * RBS * 2;
    static constexpr int bytes = (2 * 32 * RBS * 2 + 1023) & ~1023;
};
template <int DT, int KS1>
__device__ __forceinline__ void stage_rows(char* __restrict__ stage, int region_off, int m, bool valid, int D, int g,
                                           const f32x4 (&v)[DT], bool ones) {
    using SL = StageL<KS1, DT>;
    __bf16* row = reinterpret_cast<__bf16*>(stage + (size_t)(m >> 5) * SL::bytes + region_off) + (size_t)(m & 31) * SL::RBS;
#pragma unroll
    for (int dt = 0; dt < 2 * KS1; ++dt) {
        const int d0 = 16 * dt + 4 * g;
        u32x2 pk = {0u, 0u};
        if (valid) {
            if (dt < DT && d0 < D) {
                pk[0] = cvt_pk_bf16(v[dt < DT ? dt : 0][0], v[dt < DT ? dt : 0][1]);
                pk[1] = cvt_pk_bf16(v[dt < DT ? dt : 0][2], v[dt < DT ? dt : 0][3]);
            } else if (d0 == D && ones) {
                pk[0] = 0x00003F80u;
            }
        }
        *reinterpret_cast<u32x2*>(row + d0) = pk;
    }
}
// C-layout tile of 16 tokens -> T-layout rows through a wave-private LDS transpose: 4 lanes write one 32-byte run of a
// feature row (16 tokens x bf16) instead of 64 scattered 2-byte stores per instruction (20 store instructions per tile
// became 5).  trow0 = &T[row 0][first token of the tile], ts = row stride in elements, tscr = 32 * DT * 16 bytes of LDS.
template <int DT>
__device__ __forceinline__ void store_T16(char* tscr, __bf16* __restrict__ trow0, int ts, int lane, int D, const f32x4 (&v)[DT],
                                          bool ones, bool valid) {
    const int tok = lane & 15, g = lane >> 4;
    __bf16* l = reinterpret_cast<__bf16*>(tscr);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * dt + 4 * g + r;
            float x = 0.f;
            if (valid) x = (f < D) ? v[dt][r] : ((f == D && ones) ? 1.0f : 0.f);
            l[f * 16 + tok] = (__bf16)x;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < (16 * DT * 4 + 63) / 64; ++i) {
        const int idx = lane + 64 * i;
        if (idx < 16 * DT * 4) {
            const int f = idx >> 2, q = idx & 3;
            *reinterpret_cast<u32x2*>(trow0 + (size_t)f * ts + 4 * q) = *reinterpret_cast<const u32x2*>(l + f * 16 + 4 * q);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bf16x8 row_frag(const __bf16* __restrict__ rb, int m, bool valid, int RBW, int ks, int g) {
    if (!valid) return frag_zero();
    return *reinterpret_cast<const bf16x8*>(rb + (size_t)m * RBW + 32 * ks + 8 * g);
}
// C layout -> B fragments through a wave-private LDS scratch of KS1 KiB (same lanes write and read; LDS is in order per wave)
template <int DT, int KS1>
__device__ __forceinline__ void ctile_to_frags(char* scratch, int lane, int D, const f32x4 (&v)[DT], bool ones, bf16x8 (&xf)[KS1]) {
    const int tok = lane & 15, g = lane >> 4;
#pragma unroll
    for (int dt = 0; dt < 2 * KS1; ++dt) {
        const int d0 = 16 * dt + 4 * g;
        u32x2 pk = {0u, 0u};
        if (dt < DT && d0 < D) {
            pk[0] = cvt_pk_bf16(v[dt < DT ? dt : 0][0], v[dt < DT ? dt : 0][1]);
            pk[1] = cvt_pk_bf16(v[dt < DT ? dt : 0][2], v[dt < DT ? dt : 0][3]);
        } else if (d0 == D && ones) {
            pk[0] = 0x00003F80u;
        }
        const int ks = dt >> 1, gd = 2 * (dt & 1) + (g >> 1);
        *reinterpret_cast<u32x2*>(scratch + ((ks * 64 + gd * 16 + tok) * 16 + 8 * (g & 1))) = pk;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(scratch + (ks * 64 + lane) * 16);
    __builtin_amdgcn_wave_barrier();
}
// LayerNorm statistics of a C-layout tile over the D features of token lane&15
template <int DT>
__device__ __forceinline__ void ln_stats(const f32x4 (&v)[DT], int D, int g, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
        if (16 * dt + 4 * g < D) s += (v[dt][0] + v[dt][1]) + (v[dt][2] + v[dt][3]);
    const float invD = 1.0f / (float)D;      // (one reciprocal instead of two IEEE divisions on the serial path)
    mean = group_sum(s) * invD;
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
    rstd = __builtin_amdgcn_rsqf(group_sum(q) * invD + 1e-5f);
}
// dropout bits of a (token, D features) row in C layout: bytes (m, j, g) written by k_tr_masks, byte j covers the C tiles
// 2j (low nibble) and 2j+1 (high nibble)
template <int DT>
__device__ __forceinline__ void row_drop_bits(const TrDims& d, const unsigned char* __restrict__ rbits, int m, bool valid, int g,
                                              unsigned (&bits)[DT]) {      // (unconditional loads + select, see load_ctile)
    const int mc = valid ? m : 0;
    unsigned char raw[(DT + 1) / 2];
#pragma unroll
    for (int j = 0; j < (DT + 1) / 2; ++j) raw[j] = rbits[((size_t)mc * ((DT + 1) / 2) + j) * 4 + g];
#pragma unroll
    for (int j = 0; j < (DT + 1) / 2; ++j) {
        const unsigned b8 = (d.p > 0.f && valid) ? (unsigned)raw[j] : 0xffu;
        bits[2 * j] = b8 & 15u;
        if (2 * j + 1 < DT) bits[2 * j + 1] = b8 >> 4;
    }
}

}  // namespace
