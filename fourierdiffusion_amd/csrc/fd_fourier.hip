// fd_fourier.hip -- real DFT / inverse DFT of (B,T,C) series along the stride-C time axis, in the
// reference's packed real layout (src/fdiff/utils/fourier.py:8-87):
//     y[:, 0:T/2+1, :] = Re X_k ,  y[:, T/2+1:T, :] = Im X_k (k = 1..ceil(T/2)-1),  ortho norm.
//
// HBM-bound (8 B/element algorithmic: read x once, write y once).  One workgroup owns the
// (T, Cc) slab of one batch element (all channels when it fits in LDS): the slab is read with
// fully coalesced loads straight into an LDS complex image, transformed in LDS by a generic-radix
// Stockham autosort FFT (any T: radices are the prime factors of T with 2*2 merged to 4; a large
// prime factor simply becomes one direct-DFT stage), and written back coalesced.
// Two real channels ride in one complex transform (z = x_c + i x_{c+1}); the split/merge is done
// in the load/store passes.  Twiddles W_T^k are built per workgroup in double precision.
#include <cmath>
#include <vector>

#include "fd_common.h"
#include "fd_gemm_f32.h"

namespace {

constexpr int kMaxBlock = 1024;
constexpr int kMaxStages = 16;

struct FftPlan {
    int T;
    int nstages;
    int radix[kMaxStages];
    int vec_ok;        // C even, even chunk starts, 8-byte aligned x / y: channel pairs move as float2
    int stagger;       // phase stagger of the first resident round: sixteenths of a period in units of 1024 cycles (0 = off)
    int resident;      // workgroups resident at once
};

// n / d for 0 <= n < 2^22 through the float pipe (4 VALU instead of the ~40 of a 32-bit integer division; the index
// arithmetic, not the butterflies, dominated this kernel): (n + 0.5) * (1/d) is at least 0.5/d away from an integer
// and carries an absolute error below n * 2^-23.
// Global accessors of the load / store passes.  Nontemporal scalar accesses were measured and are NOT the default: (4096, 256, 28)
// 2.81 -> 2.40 TB/s, (512, 1024, 16) 1.67 -> 1.01 (-DFDIFF_FFT_NONTEMPORAL builds that form)
#ifdef FDIFF_FFT_NONTEMPORAL
__device__ __forceinline__ float ldg(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stg(float* p, float v) { __builtin_nontemporal_store(v, p); }
#else
__device__ __forceinline__ float ldg(const float* p) { return *p; }
__device__ __forceinline__ void stg(float* p, float v) { *p = v; }
#endif

__device__ __forceinline__ int fdiv(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }
// a * b for 0 <= a, b < 2^24 by v_mul_u32_u24 (full rate); the 32-bit v_mul_lo_u32 hipcc emits for plain `int * int` is
// quarter rate, and the index arithmetic of a stage -- not its butterflies -- is most of this kernel's VALU time (every index
// of the LDS-resident transform is below 2^22, every factor below 2^24)
__device__ __forceinline__ int m24(int a, int b) { return (int)__umul24((unsigned)a, (unsigned)b); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// One Stockham stage of a LARGE prime radix r (> 31), Ns = product of earlier radices.  Every thread produces the output
// pair (u, r-u) of one butterfly: W^{u(r-t)} = conj W^{ut}, so with P_t = a_t + a_{r-t}, M_t = a_t - a_{r-t}
//   X_u = a_0 + sum_{t=1..(r-1)/2} (P_t c_ut + i M_t s_ut),   X_{r-u} = a_0 + sum (P_t c_ut - i M_t s_ut)
// -- half the LDS reads and a quarter of the multiplies of one direct sum per output.  The inputs arrive pre-twiddled
// by W_T^(k ktw t) (applied on the fly: one complex multiply per input read).
__device__ __forceinline__ void stockham_stage(const float2* __restrict__ in, float2* __restrict__ out,
                                               const float2* __restrict__ tw, int T, int Cp, int r, int Ns) {
    const int NT = blockDim.x;
    const int tr = T / r;
    const int ktw = T / (Ns * r);
    const int h = (r - 1) >> 1;
    const int npair = h + 1;                        // u = 0 (alone) and the pairs u = 1..h
    const int total = tr * npair * Cp;
    const float inv_cp = 1.0f / (float)Cp, inv_ns = 1.0f / (float)Ns, inv_np = 1.0f / (float)npair;
    for (int id = threadIdx.x; id < total; id += NT) {
        const int q = fdiv(id, inv_cp), p = id - q * Cp;
        const int j = fdiv(q, inv_np), u = q - j * npair;           // butterfly j (0 .. T/r), output pair u
        const int blk = fdiv(j, inv_ns), k = j - blk * Ns;
        const int step = k * ktw;                                   // pre-twiddle exponent per t: step * t  (< T)
        const float2 a0 = in[j * Cp + p];
        float sre = 0.f, sim = 0.f, dre = 0.f, dim = 0.f;
        int e = 0, f1 = 0, f2 = 0;                                  // e = (u t mod r) * tr ; f1 = step*t ; f2 = step*(r-t)
        f2 = step * r;                                              // < T
        for (int t = 1; t <= h; ++t) {
            e += u * tr;
            e -= (e >= T) ? T : 0;
            f1 += step;
            f2 -= step;
            float2 x1 = in[(j + t * tr) * Cp + p], x2 = in[(j + (r - t) * tr) * Cp + p];
            if (k != 0) {
                x1 = cmul(x1, tw[f1]);
                x2 = cmul(x2, tw[f2]);
            }
            const float2 w = tw[e];
            const float px = x1.x + x2.x, py = x1.y + x2.y, mx = x1.x - x2.x, my = x1.y - x2.y;
            sre = fmaf(px, w.x, sre);
            sim = fmaf(py, w.x, sim);
            dre = fmaf(mx, w.y, dre);
            dim = fmaf(my, w.y, dim);
        }
        float2* o = out + ((blk * r) * Ns + k) * Cp + p;            // output v at o[v * Ns * Cp]
        const int os = Ns * Cp;
        o[u * os] = float2{a0.x + sre - dim, a0.y + sim + dre};
        if (u != 0) o[(r - u) * os] = float2{a0.x + sre + dim, a0.y + sim - dre};
    }
}

// One Stockham stage of a prime radix R in 11..31 with a whole butterfly per thread, in the paired form above: the R inputs
// are read (and pre-twiddled) once, the roots of unity live in registers, (R-1)^2 real multiplies per butterfly.
template <int R>
__device__ __forceinline__ void stockham_prime(const float2* __restrict__ in, float2* __restrict__ out,
                                               const float2* __restrict__ tw, int T, int Cp, int Ns) {
    constexpr int H = (R - 1) / 2;
    const int NT = blockDim.x;
    const int tr = T / R;
    const int ktw = T / (Ns * R);
    const int total = tr * Cp;
    const float inv_cp = 1.0f / (float)Cp, inv_ns = 1.0f / (float)Ns;
    float wc[H + 1], wsn[H + 1];                    // cos / sin of the roots m = 0..H (m > H by symmetry)
#pragma unroll
    for (int m = 0; m <= H; ++m) {
        const float2 w = tw[m * tr];
        wc[m] = w.x;
        wsn[m] = w.y;
    }
    for (int id = threadIdx.x; id < total; id += NT) {
        const int j = fdiv(id, inv_cp), p = id - j * Cp;
        const int blk = fdiv(j, inv_ns), k = j - blk * Ns;
        float2 a[R];
#pragma unroll
        for (int t = 0; t < R; ++t) a[t] = in[(j + t * tr) * Cp + p];
        if (k != 0) {
            const int step = k * ktw;
#pragma unroll
            for (int t = 1; t < R; ++t) a[t] = cmul(a[t], tw[step * t]);
        }
        // P_t -> a[t], M_t -> a[R - t]
#pragma unroll
        for (int t = 1; t <= H; ++t) {
            const float2 x1 = a[t], x2 = a[R - t];
            a[t] = float2{x1.x + x2.x, x1.y + x2.y};
            a[R - t] = float2{x1.x - x2.x, x1.y - x2.y};
        }
        float2* o = out + ((blk * R) * Ns + k) * Cp + p;
        const int os = Ns * Cp;
        {
            float2 acc = a[0];
#pragma unroll
            for (int t = 1; t <= H; ++t) { acc.x += a[t].x; acc.y += a[t].y; }
            o[0] = acc;
        }
#pragma unroll
        for (int u = 1; u <= H; ++u) {
            float sre = 0.f, sim = 0.f, dre = 0.f, dim = 0.f;
#pragma unroll
            for (int t = 1; t <= H; ++t) {
                const int m = (u * t) % R;                          // compile-time after unrolling
                const float c = wc[m <= H ? m : R - m];
                const float sn = (m <= H) ? wsn[m] : -wsn[R - m];
                sre = fmaf(a[t].x, c, sre);
                sim = fmaf(a[t].y, c, sim);
                dre = fmaf(a[R - t].x, sn, dre);
                dim = fmaf(a[R - t].y, sn, dim);
            }
            o[u * os] = float2{a[0].x + sre - dim, a[0].y + sim + dre};
            o[(R - u) * os] = float2{a[0].x + sre + dim, a[0].y + sim - dre};
        }
    }
}

// 16-point DFT in registers as 4 x 4 (two layers of radix-4 butterflies with the W_16^(q r) twiddles between them):
// x[r + 4 u] = sum_q ( (sum_s a[q + 4 s] W_4^(r s)) W_16^(q r) ) W_4^(q u).  w4 = W_4 (-i forward, +i inverse), w16[e] = W_16^e for
// e = 1, 2, 3, 6, 9 (index 0..4), all taken from the twiddle table so that they carry the transform's direction.
__device__ __forceinline__ void dft4(const float2 a0, const float2 a1, const float2 a2, const float2 a3, const float2 w4, float2& b0,
                                     float2& b1, float2& b2, float2& b3) {
    const float2 s02 = {a0.x + a2.x, a0.y + a2.y}, d02 = {a0.x - a2.x, a0.y - a2.y};
    const float2 s13 = {a1.x + a3.x, a1.y + a3.y}, d13 = {a1.x - a3.x, a1.y - a3.y};
    const float2 rot = cmul(d13, w4);
    b0 = float2{s02.x + s13.x, s02.y + s13.y};
    b1 = float2{d02.x + rot.x, d02.y + rot.y};
    b2 = float2{s02.x - s13.x, s02.y - s13.y};
    b3 = float2{d02.x - rot.x, d02.y - rot.y};
}
__device__ __forceinline__ void dft16(const float2 (&a)[16], float2 (&x)[16], const float2 w4, const float2 (&w16)[5]) {
    float2 b[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) dft4(a[q], a[q + 4], a[q + 8], a[q + 12], w4, b[q][0], b[q][1], b[q][2], b[q][3]);
    // W_16^(q r): q = 1: e = r (1, 2, 3); q = 2: e = 2 r (2, 4, 6); q = 3: e = 3 r (3, 6, 9); W_16^4 = W_4
    b[1][1] = cmul(b[1][1], w16[0]); b[1][2] = cmul(b[1][2], w16[1]); b[1][3] = cmul(b[1][3], w16[2]);
    b[2][1] = cmul(b[2][1], w16[1]); b[2][2] = cmul(b[2][2], w4);     b[2][3] = cmul(b[2][3], w16[3]);
    b[3][1] = cmul(b[3][1], w16[2]); b[3][2] = cmul(b[3][2], w16[3]); b[3][3] = cmul(b[3][3], w16[4]);
#pragma unroll
    for (int r = 0; r < 4; ++r) dft4(b[0][r], b[1][r], b[2][r], b[3][r], w4, x[r], x[r + 4], x[r + 8], x[r + 12]);
}

// One Stockham stage of small radix R with a whole butterfly per thread: R inputs are read once, pre-twiddled by
// W_T^(k*ktw*t) and combined by the R-point DFT (R = 2, 4: additions only; 3, 5, 7: the R roots of unity from the twiddle
// table).  Compared with one output per thread this does 1/R of the LDS reads, twiddle fetches and index arithmetic.
// tw already carries the direction (forward: W = exp(-2 pi i / T), inverse: conjugate), so -i / +i of the radix-4
// butterfly is tw[T/4].
template <int R>
__device__ __forceinline__ void stockham_butterflies(const float2* __restrict__ in, float2* __restrict__ out,
                                                     const float2* __restrict__ tw, int T, int Cp, int Ns) {
    const int NT = blockDim.x;
    const int tr = T / R;
    const int ktw = T / (Ns * R);
    const int total = tr * Cp;
    const float inv_cp = 1.0f / (float)Cp, inv_ns = 1.0f / (float)Ns;
    float2 wr[R];                                   // R-th roots of unity in the transform's direction
#pragma unroll
    for (int m = 0; m < R; ++m) wr[m] = tw[m * tr];
    float2 w16[5] = {};
    if (R == 16) {
        const int e16 = T / 16;
        w16[0] = tw[e16]; w16[1] = tw[2 * e16]; w16[2] = tw[3 * e16]; w16[3] = tw[6 * e16]; w16[4] = tw[9 * e16];
    }
    const int trC = tr * Cp, RNs = R * Ns;          // (wave-uniform products: scalar unit)
    for (int id = threadIdx.x; id < total; id += NT) {
        const int j = fdiv(id, inv_cp), p = id - m24(j, Cp);
        const int blk = fdiv(j, inv_ns), k = j - m24(blk, Ns);
        float2 a[R];
        const float2* i0 = in + (id - 0);           // in[j * Cp + p] == in[id]: the butterfly's inputs are id + t * tr * Cp
#pragma unroll
        for (int t = 0; t < R; ++t) a[t] = i0[t * trC];
        if (k != 0) {
            const int step = m24(k, ktw);           // < T / R
#pragma unroll
            for (int t = 1; t < R; ++t) a[t] = cmul(a[t], tw[step * t]);   // step * t < T (t is a literal: shifts / adds)
        }
        float2* o = out + m24(m24(blk, RNs) + k, Cp) + p;                    // output u at o[u * Ns * Cp]
        const int os = Ns * Cp;
        if (R == 16) {
            float2 x16[16];
            float2 a16[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) a16[t] = a[t % R];
            dft16(a16, x16, tw[T / 4], w16);
#pragma unroll
            for (int u = 0; u < 16; ++u) o[u * os] = x16[u];
        } else if (R == 2) {
            o[0] = float2{a[0].x + a[1].x, a[0].y + a[1].y};
            o[os] = float2{a[0].x - a[1].x, a[0].y - a[1].y};
        } else if (R == 4) {
            const float2 s02 = {a[0].x + a[2].x, a[0].y + a[2].y}, d02 = {a[0].x - a[2].x, a[0].y - a[2].y};
            const float2 s13 = {a[1].x + a[3].x, a[1].y + a[3].y}, d13 = {a[1].x - a[3].x, a[1].y - a[3].y};
            const float2 rot = cmul(d13, wr[1]);    // (a1 - a3) * W_4   (W_4 = -i forward, +i inverse)
            o[0] = float2{s02.x + s13.x, s02.y + s13.y};
            o[os] = float2{d02.x + rot.x, d02.y + rot.y};
            o[2 * os] = float2{s02.x - s13.x, s02.y - s13.y};
            o[3 * os] = float2{d02.x - rot.x, d02.y - rot.y};
        } else {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                float2 acc = a[0];
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    const float2 w = wr[(u * t) % R];
                    acc.x += a[t].x * w.x - a[t].y * w.y;
                    acc.y += a[t].x * w.y + a[t].y * w.x;
                }
                o[u * os] = acc;
            }
        }
    }
}

// A workgroup-uniform float that was produced by the VALU or read from LDS sits in a VGPR until it is moved to a scalar
// register by hand (the in-place instantiations were at their 128-VGPR budget with 2-5 spilled dwords; with the radix-16
// twiddles and the reciprocals in SGPRs they need 93-107 and no scratch memory).
__device__ __forceinline__ float uniformf(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ float2 uniformf2(float2 v) { return float2{uniformf(v.x), uniformf(v.y)}; }

// In-place decimation-in-frequency stage of small radix R on sub-transforms of length Lc (Gentleman-Sande): the butterfly of
// (block, j), j < m = Lc / R, reads X[base + j + t m], t < R, and writes  b_u = (sum_t a_t W_R^(u t)) W_Lc^(j u)  back to
// X[base + j + u m] -- the same R places, so ONE LDS image serves the whole transform (the Stockham form needs two) and four
// instead of two workgroups fit a CU at (T, C) = (256, 28).  After the last stage X_k sits at the digit-reversed place
// rev[k]; the store pass reads through that table.
template <int R>
__device__ __forceinline__ void dif_butterflies(float2* __restrict__ X, const float2* __restrict__ tw, int T, int Cp, int Lc) {
    const int NT = blockDim.x;
    const int m = Lc / R;
    const int tr = T / R;
    const int ktw = T / Lc;                         // W_Lc = W_T^ktw
    const int total = tr * Cp;
    const float inv_cp = uniformf(1.0f / (float)Cp), inv_m = uniformf(1.0f / (float)m);
    float2 wr[R];                                   // R-th roots of unity in the transform's direction
#pragma unroll
    for (int q = 0; q < R; ++q) wr[q] = tw[q * tr];
    float2 w16[5] = {};
    float2 w4 = {};
    if (R == 16) {
        const int e16 = T / 16;
        w16[0] = uniformf2(tw[e16]); w16[1] = uniformf2(tw[2 * e16]); w16[2] = uniformf2(tw[3 * e16]);
        w16[3] = uniformf2(tw[6 * e16]); w16[4] = uniformf2(tw[9 * e16]);
        w4 = uniformf2(tw[T / 4]);
    }
    const int ms = m * Cp;
    for (int id = threadIdx.x; id < total; id += NT) {
        const int bj = fdiv(id, inv_cp), p = id - m24(bj, Cp);
        const int blk = fdiv(bj, inv_m), j = bj - m24(blk, m);
        float2* x0 = X + m24(m24(blk, Lc) + j, Cp) + p;                      // input / output t at x0[t * m * Cp]
        float2 a[R], b[R];
#pragma unroll
        for (int t = 0; t < R; ++t) a[t] = x0[t * ms];
        if (R == 16) {
            float2 a16[16], x16[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) a16[t] = a[t % R];
            dft16(a16, x16, w4, w16);
#pragma unroll
            for (int u = 0; u < R; ++u) b[u] = x16[u % 16];
        } else if (R == 2) {
            b[0] = float2{a[0].x + a[1].x, a[0].y + a[1].y};
            b[1] = float2{a[0].x - a[1].x, a[0].y - a[1].y};
        } else if (R == 4) {
            const float2 s02 = {a[0].x + a[2].x, a[0].y + a[2].y}, d02 = {a[0].x - a[2].x, a[0].y - a[2].y};
            const float2 s13 = {a[1].x + a[3].x, a[1].y + a[3].y}, d13 = {a[1].x - a[3].x, a[1].y - a[3].y};
            const float2 rot = cmul(d13, wr[1]);    // (a1 - a3) * W_4   (W_4 = -i forward, +i inverse)
            b[0] = float2{s02.x + s13.x, s02.y + s13.y};
            b[1] = float2{d02.x + rot.x, d02.y + rot.y};
            b[2] = float2{s02.x - s13.x, s02.y - s13.y};
            b[3] = float2{d02.x - rot.x, d02.y - rot.y};
        } else {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                float2 acc = a[0];
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    const float2 w = wr[(u * t) % R];
                    acc.x += a[t].x * w.x - a[t].y * w.y;
                    acc.y += a[t].x * w.y + a[t].y * w.x;
                }
                b[u] = acc;
            }
        }
        if (j != 0 && m > 1) {
            const int step = m24(j, ktw);           // j u ktw < m R ktw = T
#pragma unroll
            for (int u = 1; u < R; ++u) b[u] = cmul(b[u], tw[step * u]);
        }
#pragma unroll
        for (int u = 0; u < R; ++u) x0[u * ms] = b[u];
    }
}

// INVERSE == false : x (time)  -> y (packed spectrum), optional (y - mean)/std
// INVERSE == true  : x (packed spectrum), optional x*std + mean  -> y (time)
// BATCHED (single-channel data, C == 1): the workgroup transforms Cc consecutive SERIES instead of Cc channels of one series
// -- "channel" j is series blockIdx.x * Cc + j, contiguous along time -- so that a (B, T, 1) set (the reference's ECG
// data, 87 554 x 187 x 1) still fills complex lanes and workgroups (one series per workgroup ran at 0.6 TB/s).
// BIGP: the instantiation that carries the register butterflies of the prime radices 17 / 19 / 23 (34-46 complex registers per
// thread: they spill at the 1024-thread budget, and a kernel with scratch pays for it in every stage --
// (4096, 256, 28) lost 9 % when they lived in the common instantiation); lengths without such a factor run the BIGP = false
// kernel.  (Built for 512 threads = 256 VGPRs the BIGP kernel has no scratch but half the waves per CU: (87 554, 187, 1)
// 58 -> 83 us, so it keeps its spills.)
template <bool INVERSE, bool BATCHED, bool BIGP, bool INPLACE = false>
__global__ __launch_bounds__(kMaxBlock) void k_fft(const float* __restrict__ x, float* __restrict__ y,
                                                    const float* __restrict__ mean, const float* __restrict__ stdv,
                                                    const float2* __restrict__ tw_fwd, int B, int C, int Cc, FftPlan plan) {
    // Phase stagger (plan.stagger > 0): every workgroup of the launch takes the same time, so without it the whole chip loads,
    // computes and stores in lockstep round after round -- HBM idle during the stages, the SIMDs idle during the transfers
    // (measured: kernel time = transfer time + stage time, also at four workgroups per CU).  The first resident round starts
    // spread over one period; later workgroups inherit the spread through the slots they take over.
    if (plan.stagger > 0) {
        const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x;
        if (lin < (unsigned)plan.resident) {
            const unsigned slots = ((lin * 2654435761u) >> 16) % 16u;          // 0 .. 15 sixteenths of the period
            for (unsigned i = 0; i < slots * (unsigned)plan.stagger; ++i) __builtin_amdgcn_s_sleep(16);   // 16 x 64 cycles
        }
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NT = blockDim.x;
    const int T = plan.T;
    const int b = BATCHED ? blockIdx.x * Cc : blockIdx.x;                 // first series of the group / the series
    const int c0 = BATCHED ? 0 : blockIdx.y * Cc;
    const int cc = BATCHED ? min(Cc, B - b) : min(Cc, C - c0);            // channels (series) in this chunk
    const int Cp = (cc + 1) >> 1;            // complex lanes (channel pairs)
    float2* tw = reinterpret_cast<float2*>(smem);
    float2* bufA = tw + T;
    float2* bufB = bufA + (size_t)T * ((Cc + 1) >> 1);          // Stockham ping-pong partner; INPLACE: the digit-reversal table
    unsigned short* rev = reinterpret_cast<unsigned short*>(bufB);
    if (INPLACE) {
        // rev[k] = place of X_k after the DIF stages: place = sum_s d_s m_s (m_s = T / (r_0 .. r_s)), k = sum_s d_s (r_0 .. r_{s-1})
        for (int k = threadIdx.x; k < T; k += blockDim.x) {
            int rem = k, place = 0, mcur = T;
            for (int st = 0; st < plan.nstages; ++st) {
                const int r = plan.radix[st];
                mcur /= r;
                const int dgt = rem % r;
                rem /= r;
                place += dgt * mcur;
            }
            rev[k] = (unsigned short)place;
        }
    }
    // W_T^k = exp(-2 pi i k / T), built once per T on the host in double precision (a per-workgroup sincospi in double
    // was a third of the kernel's time); the inverse transform conjugates
    for (int k = threadIdx.x; k < T; k += NT) {
        const float2 w = tw_fwd[k];
        tw[k] = float2{w.x, INVERSE ? -w.y : w.y};
    }
    const float* xb = x + (size_t)b * T * C;
    float* yb = y + (size_t)b * T * C;
    const int n_real = T / 2 + 1;
    const bool even = (T & 1) == 0;
    const float scale = uniformf(rsqrtf((float)T));
    // offset of (row, channel) in xb / yb, and in the (T, C) mean / std tables; BATCHED: channel = series, rows contiguous
    // (32-bit offsets inside one series / series group: T * C * Cc < 2^31 is checked on the host; factors < 2^24 for m24)
    const int cstep = BATCHED ? T : 1, rstep = BATCHED ? 1 : C, mstep = BATCHED ? 0 : 1;
    // channel pairs as 8-byte accesses: rows of an even number of channels, chunks start at even channels and the tensor base is
    // 8-byte aligned (the host passes `vec_ok`): half the memory instructions and address arithmetic of the load / store passes
    const bool vec2 = !BATCHED && plan.vec_ok != 0;
    // two channel pairs as one 16-byte access (vec_ok == 2: C and the chunk multiples of 4, 16-byte aligned tensors): half the
    // items, address arithmetic and memory instructions of the passes again
    const bool vec4 = vec2 && plan.vec_ok == 2 && !BIGP;
    const int Cq = Cp >> 1;
    const float inv_cq = uniformf(1.0f / (float)max(Cq, 1));
    auto f4 = [](const float* q) { return *reinterpret_cast<const float4*>(q); };

    // ---- load pass: build z[n][p]  (thread order: channel pairs fastest, BATCHED: rows fastest = contiguous in memory)
    const float inv_cp = uniformf(1.0f / (float)Cp), inv_t = uniformf(1.0f / (float)T), inv_nr = uniformf(1.0f / (float)n_real);
    if (!INVERSE && vec4) {
        for (int id = threadIdx.x; id < T * Cq; id += NT) {
            const int n = fdiv(id, inv_cq), q = id - m24(n, Cq);
            *reinterpret_cast<float4*>(bufA + m24(n, Cp) + 2 * q) = f4(xb + m24(n, C) + c0 + 4 * q);
        }
    } else if (INVERSE && vec4) {
        for (int id = threadIdx.x; id < n_real * Cq; id += NT) {
            const int kk = fdiv(id, inv_cq), q = id - m24(kk, Cq);
            const int ca = c0 + 4 * q;
            const bool has_im = (kk != 0) && !(even && kk == T / 2);
            const int ire = m24(kk, C) + ca, iim = m24(n_real + kk - 1, C) + ca;
            float4 re4 = f4(xb + ire);
            float4 im4 = has_im ? f4(xb + iim) : float4{0.f, 0.f, 0.f, 0.f};
            if (mean) {   // de-standardise in the frequency domain (cmd/sample.py:76-78)
                const float4 s4 = f4(stdv + ire), m4 = f4(mean + ire);
                re4 = float4{re4.x * s4.x + m4.x, re4.y * s4.y + m4.y, re4.z * s4.z + m4.z, re4.w * s4.w + m4.w};
                if (has_im) {
                    const float4 t4 = f4(stdv + iim), n4 = f4(mean + iim);
                    im4 = float4{im4.x * t4.x + n4.x, im4.y * t4.y + n4.y, im4.z * t4.z + n4.z, im4.w * t4.w + n4.w};
                }
            }
            // (re4, im4) = (Xa, Xb) of two channel pairs; Z = Xa + i Xb at k, both imaginary parts change sign at T - k
            *reinterpret_cast<float4*>(bufA + m24(kk, Cp) + 2 * q) = float4{re4.x - im4.y, im4.x + re4.y, re4.z - im4.w, im4.z + re4.w};
            if (has_im)
                *reinterpret_cast<float4*>(bufA + m24(T - kk, Cp) + 2 * q) = float4{re4.x + im4.y, re4.y - im4.x, re4.z + im4.w, re4.w - im4.z};
        }
    } else if (!INVERSE) {
        for (int id = threadIdx.x; id < T * Cp; id += NT) {
            int n, p;
            if (BATCHED) { p = fdiv(id, inv_t); n = id - m24(p, T); } else { n = fdiv(id, inv_cp); p = id - m24(n, Cp); }
            const int ca = c0 + 2 * p;
            const bool has_b = (2 * p + 1) < cc;
            float2 z;
            if (vec2) {           // both channels of the pair in one 8-byte access (C even: 8-byte aligned)
                z = *reinterpret_cast<const float2*>(xb + m24(n, rstep) + ca);
            } else {
                z.x = ldg(xb + m24(n, rstep) + m24(ca, cstep));
                z.y = has_b ? ldg(xb + m24(n, rstep) + m24(ca + 1, cstep)) : 0.f;
            }
            bufA[BATCHED ? m24(n, Cp) + p : id] = z;        // (non-batched: id == n * Cp + p)
        }
    } else if (BIGP && !BATCHED) {
        // (the instantiation with the radix 17 / 19 / 23 register butterflies lives at its VGPR budget with spills: there the
        // one-item-two-places form below measured 9-13 % slower -- (4096, 187, 12) 38.6 -> 42 us --, so it keeps a thread per
        // time step)
        for (int id = threadIdx.x; id < T * Cp; id += NT) {
            const int n = fdiv(id, inv_cp), p = id - m24(n, Cp);
            const int ca = c0 + 2 * p;
            const bool has_b = (2 * p + 1) < cc;
            float2 z;
            // Hermitian extension of the packed half spectrum (fourier.py:62-77): X[T-k] = conj X[k]
            const int kk = (n <= T / 2) ? n : T - n;
            const bool has_im = (kk != 0) && !(even && kk == T / 2);
            const float sg = (n <= T / 2) ? 1.0f : -1.0f;
            const int ire = m24(kk, rstep), iim = m24(n_real + kk - 1, rstep);
            const int oa = m24(ca, cstep), ob = m24(ca + 1, cstep);
            float are, aim, bre, bim;
            if (vec2) {
                const float2 re2 = *reinterpret_cast<const float2*>(xb + ire + oa);
                const float2 im2 = has_im ? *reinterpret_cast<const float2*>(xb + iim + oa) : float2{0.f, 0.f};
                are = re2.x; bre = re2.y; aim = im2.x; bim = im2.y;
            } else {
                are = ldg(xb + ire + oa); aim = has_im ? ldg(xb + iim + oa) : 0.f;
                bre = has_b ? ldg(xb + ire + ob) : 0.f; bim = (has_b && has_im) ? ldg(xb + iim + ob) : 0.f;
            }
            if (mean) {   // de-standardise in the frequency domain (cmd/sample.py:76-78)
                const int mre = m24(kk, C), mim = m24(n_real + kk - 1, C), ma = ca * mstep, mb = (ca + 1) * mstep;
                are = are * stdv[mre + ma] + mean[mre + ma];
                if (has_im) aim = aim * stdv[mim + ma] + mean[mim + ma];
                if (has_b) {
                    bre = bre * stdv[mre + mb] + mean[mre + mb];
                    if (has_im) bim = bim * stdv[mim + mb] + mean[mim + mb];
                }
            }
            aim *= sg;
            bim *= sg;
            // Z = Xa + i Xb
            z.x = are - bim;
            z.y = aim + bre;
            bufA[id] = z;
        }
    } else {
        // Hermitian extension of the packed half spectrum (fourier.py:62-77): X[T-k] = conj X[k].  One item = one coefficient
        // k <= T/2 of a channel pair: it is read once and written to both places k and T - k (a thread per time step n read
        // every coefficient twice: iRFFT fetched 124 MB for a 112 MB tensor and ran the index arithmetic of the pass twice).
        for (int id = threadIdx.x; id < n_real * Cp; id += NT) {
            int kk, p;
            if (BATCHED) { p = fdiv(id, inv_nr); kk = id - m24(p, n_real); } else { kk = fdiv(id, inv_cp); p = id - m24(kk, Cp); }
            const int ca = c0 + 2 * p;
            const bool has_b = (2 * p + 1) < cc;
            const bool has_im = (kk != 0) && !(even && kk == T / 2);
            const int ire = m24(kk, rstep), iim = m24(n_real + kk - 1, rstep);
            const int oa = m24(ca, cstep), ob = m24(ca + 1, cstep);
            float are, aim, bre, bim;
            if (vec2) {
                const float2 re2 = *reinterpret_cast<const float2*>(xb + ire + oa);
                const float2 im2 = has_im ? *reinterpret_cast<const float2*>(xb + iim + oa) : float2{0.f, 0.f};
                are = re2.x; bre = re2.y; aim = im2.x; bim = im2.y;
            } else {
                are = ldg(xb + ire + oa); aim = has_im ? ldg(xb + iim + oa) : 0.f;
                bre = has_b ? ldg(xb + ire + ob) : 0.f; bim = (has_b && has_im) ? ldg(xb + iim + ob) : 0.f;
            }
            if (mean) {   // de-standardise in the frequency domain (cmd/sample.py:76-78)
                const int mre = m24(kk, C), mim = m24(n_real + kk - 1, C), ma = ca * mstep, mb = (ca + 1) * mstep;
                are = are * stdv[mre + ma] + mean[mre + ma];
                if (has_im) aim = aim * stdv[mim + ma] + mean[mim + ma];
                if (has_b) {
                    bre = bre * stdv[mre + mb] + mean[mre + mb];
                    if (has_im) bim = bim * stdv[mim + mb] + mean[mim + mb];
                }
            }
            // Z = Xa + i Xb at k; at T - k both imaginary parts change sign
            bufA[m24(kk, Cp) + p] = float2{are - bim, aim + bre};
            if (has_im) bufA[m24(T - kk, Cp) + p] = float2{are + bim, bre - aim};
        }
    }
    __syncthreads();

    // ---- stages: Stockham ping-pong, or in place (INPLACE: smooth lengths only, radices 2 / 3 / 4 / 5 / 7)
    float2* src = bufA;
    float2* dst = bufB;
    if (INPLACE) {
        int Lc = T;
        for (int s = 0; s < plan.nstages; ++s) {
            const int r = plan.radix[s];
            switch (r) {
                case 2: dif_butterflies<2>(bufA, tw, T, Cp, Lc); break;
                case 3: dif_butterflies<3>(bufA, tw, T, Cp, Lc); break;
                case 4: dif_butterflies<4>(bufA, tw, T, Cp, Lc); break;
                case 5: dif_butterflies<5>(bufA, tw, T, Cp, Lc); break;
                case 16: dif_butterflies<16>(bufA, tw, T, Cp, Lc); break;
                default: dif_butterflies<7>(bufA, tw, T, Cp, Lc); break;
            }
            Lc /= r;
            __syncthreads();
        }
    } else {
    int Ns = 1;
#if defined(FD_FFT_ABL) && FD_FFT_ABL == 1
    for (int s = 0; s < 0; ++s) {
#else
    for (int s = 0; s < plan.nstages; ++s) {
#endif
        const int r = plan.radix[s];
        switch (r) {                                  // (uniform: one stage, one radix for the whole workgroup)
            case 2: stockham_butterflies<2>(src, dst, tw, T, Cp, Ns); break;
            case 3: stockham_butterflies<3>(src, dst, tw, T, Cp, Ns); break;
            case 4: stockham_butterflies<4>(src, dst, tw, T, Cp, Ns); break;
            case 5: stockham_butterflies<5>(src, dst, tw, T, Cp, Ns); break;
            case 7: stockham_butterflies<7>(src, dst, tw, T, Cp, Ns); break;
            case 16: stockham_butterflies<16>(src, dst, tw, T, Cp, Ns); break;
            case 11: stockham_prime<11>(src, dst, tw, T, Cp, Ns); break;
            case 13: stockham_prime<13>(src, dst, tw, T, Cp, Ns); break;
            case 17: if (BIGP) { stockham_prime<17>(src, dst, tw, T, Cp, Ns); break; }
            case 19: if (BIGP && r == 19) { stockham_prime<19>(src, dst, tw, T, Cp, Ns); break; }
            case 23: if (BIGP && r == 23) { stockham_prime<23>(src, dst, tw, T, Cp, Ns); break; }
            default: stockham_stage(src, dst, tw, T, Cp, r, Ns);   // larger prime factor: one paired direct-DFT stage
        }
        Ns *= r;
        __syncthreads();
        float2* tmp = src;
        src = dst;
        dst = tmp;
    }
    }
    // place of coefficient / sample k in `src` (identity for the autosort form)
    auto at = [&](int k) { return INPLACE ? (int)rev[k] : k; };

    // ---- store pass
    if (!INVERSE && vec4) {
        for (int id = threadIdx.x; id < n_real * Cq; id += NT) {
            const int k = fdiv(id, inv_cq), q = id - m24(k, Cq);
            const int ca = c0 + 4 * q;
            const float4 zk = *reinterpret_cast<const float4*>(src + m24(at(k), Cp) + 2 * q);
            const float4 zm = *reinterpret_cast<const float4*>(src + m24(at((k == 0) ? 0 : (T - k)), Cp) + 2 * q);
            const float h = 0.5f * scale;
            float4 re4 = {(zk.x + zm.x) * h, (zk.y + zm.y) * h, (zk.z + zm.z) * h, (zk.w + zm.w) * h};      // Xa.re, Xb.re of both pairs
            float4 im4 = {(zk.y - zm.y) * h, (zm.x - zk.x) * h, (zk.w - zm.w) * h, (zm.z - zk.z) * h};      // Xa.im, Xb.im
            const bool has_im = (k != 0) && !(even && k == T / 2);   // fourier.py:26-37 drop exact zeros
            const int ire = m24(k, C) + ca, iim = m24(n_real + k - 1, C) + ca;
            if (mean) {   // standardise (datamodules.py:61-62)
                const float4 m4 = f4(mean + ire), s4 = f4(stdv + ire);
                re4 = float4{(re4.x - m4.x) / s4.x, (re4.y - m4.y) / s4.y, (re4.z - m4.z) / s4.z, (re4.w - m4.w) / s4.w};
                if (has_im) {
                    const float4 n4 = f4(mean + iim), t4 = f4(stdv + iim);
                    im4 = float4{(im4.x - n4.x) / t4.x, (im4.y - n4.y) / t4.y, (im4.z - n4.z) / t4.z, (im4.w - n4.w) / t4.w};
                }
            }
            *reinterpret_cast<float4*>(yb + ire) = re4;
            if (has_im) *reinterpret_cast<float4*>(yb + iim) = im4;
        }
    } else if (INVERSE && vec4) {
        for (int id = threadIdx.x; id < T * Cq; id += NT) {
            const int n = fdiv(id, inv_cq), q = id - m24(n, Cq);
            const float4 z = *reinterpret_cast<const float4*>(src + m24(at(n), Cp) + 2 * q);
            *reinterpret_cast<float4*>(yb + m24(n, C) + c0 + 4 * q) = float4{z.x * scale, z.y * scale, z.z * scale, z.w * scale};
        }
    } else if (!INVERSE) {
        // X_a[k] = (Z[k] + conj Z[T-k]) / 2 ,  X_b[k] = (Z[k] - conj Z[T-k]) / (2i)
        for (int id = threadIdx.x; id < n_real * Cp; id += NT) {
            int k, p;
            if (BATCHED) { p = fdiv(id, inv_nr); k = id - m24(p, n_real); } else { k = fdiv(id, inv_cp); p = id - m24(k, Cp); }
            const int ca = c0 + 2 * p;
            const bool has_b = (2 * p + 1) < cc;
            const float2 zk = src[m24(at(k), Cp) + p];
            const float2 zm = src[m24(at((k == 0) ? 0 : (T - k)), Cp) + p];
            const float h = 0.5f * scale;
            float are = (zk.x + zm.x) * h, aim = (zk.y - zm.y) * h;
            float bre = (zk.y + zm.y) * h, bim = (zm.x - zk.x) * h;
            const bool has_im = (k != 0) && !(even && k == T / 2);   // fourier.py:26-37 drop exact zeros
            const int ire = m24(k, rstep), iim = m24(n_real + k - 1, rstep);
            const int oa = m24(ca, cstep), ob = m24(ca + 1, cstep);
            if (mean) {   // standardise (datamodules.py:61-62)
                const int mre = m24(k, C), mim = m24(n_real + k - 1, C), ma = ca * mstep, mb = (ca + 1) * mstep;
                are = (are - mean[mre + ma]) / stdv[mre + ma];
                if (has_im) aim = (aim - mean[mim + ma]) / stdv[mim + ma];
                if (has_b) {
                    bre = (bre - mean[mre + mb]) / stdv[mre + mb];
                    if (has_im) bim = (bim - mean[mim + mb]) / stdv[mim + mb];
                }
            }
            if (vec2) {
                *reinterpret_cast<float2*>(yb + ire + oa) = float2{are, bre};
                if (has_im) *reinterpret_cast<float2*>(yb + iim + oa) = float2{aim, bim};
            } else {
                stg(yb + ire + oa, are);
                if (has_b) stg(yb + ire + ob, bre);
                if (has_im) {
                    stg(yb + iim + oa, aim);
                    if (has_b) stg(yb + iim + ob, bim);
                }
            }
        }
    } else {
        for (int id = threadIdx.x; id < T * Cp; id += NT) {
            int n, p;
            if (BATCHED) { p = fdiv(id, inv_t); n = id - m24(p, T); } else { n = fdiv(id, inv_cp); p = id - m24(n, Cp); }
            const int ca = c0 + 2 * p;
            const float2 z = src[m24(at(n), Cp) + p];
            if (vec2) {
                *reinterpret_cast<float2*>(yb + m24(n, rstep) + ca) = float2{z.x * scale, z.y * scale};
            } else {
                stg(yb + m24(n, rstep) + m24(ca, cstep), z.x * scale);
                if ((2 * p + 1) < cc) stg(yb + m24(n, rstep) + m24(ca + 1, cstep), z.y * scale);
            }
        }
    }
}

bool make_plan(int T, FftPlan& plan) {
    plan.T = T;
    plan.nstages = 0;
    int n = T;
    // radix 16 (one 4 x 4 butterfly in registers): half the LDS round trips, barriers and index arithmetic of two radix-4
    // stages; FDIFF_FFT_RADIX16=0 plans radix-4 stages only (A/B runs, parity tests)
    static const bool r16 = !(getenv("FDIFF_FFT_RADIX16") && getenv("FDIFF_FFT_RADIX16")[0] == '0');
    while (r16 && n % 16 == 0) {
        if (plan.nstages >= kMaxStages) return false;
        plan.radix[plan.nstages++] = 16;
        n /= 16;
    }
    while (n % 4 == 0) {
        if (plan.nstages >= kMaxStages) return false;
        plan.radix[plan.nstages++] = 4;
        n /= 4;
    }
    for (int f = 2; n > 1; ) {
        if (n % f == 0) {
            if (plan.nstages >= kMaxStages) return false;
            plan.radix[plan.nstages++] = f;
            n /= f;
        } else {
            f += (f == 2) ? 1 : 2;
            if ((long long)f * f > n) f = n;
        }
    }
    return true;
}

}  // namespace

void fd_apply_row_matrix(fd_ctx* ctx, const float* in, const float* Mx, float* out, int B, int T, int C, hipStream_t stream);

namespace {

// The packed transform as a real (T, T) matrix, Mx[t][k] (forward: y_k = sum_t x_t Mx[t][k]) or Mx[k][t] (inverse:
// x_t = sum_k y_k Mx[k][t]), built on the host in double and cached per (T, direction) next to the twiddle tables (key
// -T / -T - 2^20 in ctx->fft_tw).  Row layout of the packed axis: fourier.py:30-40 / :62-77.
const float* dense_matrix(fd_ctx* ctx, int T, bool inverse) {
    const int key = inverse ? -T - (1 << 20) : -T;
    for (auto& e : ctx->fft_tw)
        if (e.first == key) return reinterpret_cast<const float*>(e.second);
    const int n_real = T / 2 + 1;
    const double sc = 1.0 / std::sqrt((double)T), w0 = 2.0 * 3.14159265358979323846 / (double)T;
    std::vector<float> h((size_t)T * T);
    for (int r = 0; r < T; ++r) {                            // packed row r: Re X_k (k = r) or Im X_k (k = r - n_real + 1)
        const bool is_im = r >= n_real;
        const int k = is_im ? r - n_real + 1 : r;
        const bool edge = (k == 0) || ((T % 2 == 0) && k == T / 2);
        for (int t = 0; t < T; ++t) {
            const double ang = w0 * (double)((long long)k * t % T);
            if (!inverse) {
                h[(size_t)t * T + r] = (float)((is_im ? -std::sin(ang) : std::cos(ang)) * sc);
            } else {                                         // Hermitian extension: every interior bin counts twice
                const double wgt = edge ? 1.0 : 2.0;
                h[(size_t)r * T + t] = (float)((is_im ? -std::sin(ang) : std::cos(ang)) * sc * wgt);
            }
        }
    }
    void* d = nullptr;
    if (hipMalloc(&d, sizeof(float) * h.size()) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return nullptr;
    }
    ctx->fft_tw.emplace_back(key, d);
    return reinterpret_cast<const float*>(d);
}

template <bool INVERSE>
int launch(fd_ctx* ctx, const float* x, float* y, const float* mean, const float* stdv, int B, int T, int C,
           void* stream, const char* who) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, x && y, "%s: null pointer", who);
    FD_REQUIRE(ctx, x != y, "%s: in-place transform is not supported", who);
    FD_REQUIRE(ctx, B > 0 && T > 0 && C > 0, "%s: bad shape B=%d T=%d C=%d", who, B, T, C);
    FD_REQUIRE(ctx, (mean == nullptr) == (stdv == nullptr), "%s: mean and std must come together", who);
    FD_REQUIRE(ctx, B <= 2147483647 / 2, "%s: batch too large", who);
    FD_REQUIRE(ctx, C < (1 << 24) && (long long)T * C < 2147483647LL, "%s: one series (T * C = %lld elements) exceeds the 32-bit "
               "in-series offsets of the kernel", who, (long long)T * C);
    FftPlan plan;
    FD_REQUIRE(ctx, make_plan(T, plan), "%s: T=%d has too many prime factors", who, T);
    // A length dominated by one large prime factor p runs that factor as a direct-DFT stage, O(T p) per transform; from
    // p ~ 0.4 T on the whole transform as ONE dense (T, T) real matrix product on the fp32 MFMA pipe is faster
    // ((4096, 251, 12): 529 -> 170 us).  FDIFF_DFT=fft|dense overrides (parity tests run both).
    {
        int pmax = 1;
        for (int i = 0; i < plan.nstages; ++i) pmax = std::max(pmax, plan.radix[i]);
        const char* e = getenv("FDIFF_DFT");
        bool dense = pmax > 16 && (double)pmax >= 0.4 * (double)T && T <= 2048;
        if (e && e[0] == 'f') dense = false;
        if (e && e[0] == 'd') dense = T <= 2048;
        if (dense && !mean && (long long)B * T * C < 2147483647LL) {
            const float* Mx = dense_matrix(ctx, T, INVERSE);
            if (Mx) {
                fd_apply_row_matrix(ctx, x, Mx, y, B, T, C, (hipStream_t)stream);
                FD_LAUNCH_CHECK(ctx);
                return FD_OK;
            }
        }
    }
    // In-place form (one LDS image + the digit-reversal table instead of two images) for smooth lengths: radices 2 / 3 / 4 / 5 / 7
    // only.  Measured (profiles/r03_fft_experiments.txt): with whole rows in one chunk either way it is 5-12 % SLOWER than the
    // autosort form although four instead of two workgroups fit a CU (the kernel is bound by its instruction count, 55 % VALU
    // busy, not by residency), but where the autosort form has to split the channels of a row over several workgroups
    // ((512, 1024, 16): two chunks of 8) the single image keeps rows whole: iRFFT 48.2 -> 41.6 us.  So: in place exactly when that
    // saves a channel split.  FDIFF_FFT_INPLACE=0 / 1 forces either form where both exist (the parity tests run both).
    bool smooth = T < 65536;
    for (int i = 0; i < plan.nstages; ++i) smooth &= (plan.radix[i] <= 5 || plan.radix[i] == 7 || plan.radix[i] == 16);
    static const int lds_target_kb0 = getenv("FDIFF_FFT_LDS_KB") ? atoi(getenv("FDIFF_FFT_LDS_KB")) : 72;
    const long long auto_pairs = ((long long)lds_target_kb0 * 1024 - (long long)T * 8) / ((long long)T * 16);
    bool inplace = smooth && C > 1 && auto_pairs >= 4 && (C + 1) / 2 > auto_pairs;
    // Round 3, after the in-place instantiations stopped spilling (93-107 VGPRs, no scratch memory) and with 16 elements per
    // thread (profiles/r03_fft_forms.txt, rocprofv3 kernel times): lengths whose plan starts with a radix-16 stage are faster
    // in place -- (4096, 256, 28) 68.9 / 71.6 -> 63.7 / 69.7 us with four 256-thread workgroups per CU, (512, 1024, 16)
    // 35.7 / 35.6 (two channel chunks) -> 27.9 / 26.1 us, (65 536, 256, 1) 43.2 / 45.2 -> 40.8 / 41.4 us; lengths of small odd
    // radices are not ((4096, 252, 6) 27.6 -> 32.7 us) and keep the autosort form.
    if (smooth && plan.nstages > 0 && plan.radix[0] == 16 && (C > 1 || (C == 1 && B >= 4))) inplace = true;
    if (const char* e = getenv("FDIFF_FFT_INPLACE")) inplace = smooth && e[0] != '0';
    // LDS: twiddles (8T) + complex images of T * ceil(Cc/2) float2: two (Stockham ping-pong) or one + 2T bytes (in place)
    const size_t lds_cap = 128 * 1024;
    const size_t fixed = (size_t)T * 8 + (inplace ? (size_t)T * 2 + 16 : 0), per_pair = (size_t)T * (inplace ? 8 : 16);
    FD_REQUIRE(ctx, (size_t)T * 8 + 2 * (size_t)T * 8 <= lds_cap, "%s: T=%d too long for the LDS-resident transform",
               who, T);   // (also keeps every index below 2^22, the range of fdiv)
    int max_pairs = (int)((lds_cap - fixed) / per_pair);
    // prefer <= 72 KiB per workgroup (two resident workgroups per CU overlap each other's load / store passes; the in-place
    // form then fits four) as long as a chunk keeps at least 8 channels = one 32-byte sector per time step
    static const int lds_target_kb = getenv("FDIFF_FFT_LDS_KB") ? atoi(getenv("FDIFF_FFT_LDS_KB")) : 72;
    const size_t target = (size_t)lds_target_kb * 1024;
    const int pairs_2wg = target > fixed ? (int)((target - fixed) / per_pair) : 0;
    // (in place: whole rows in one chunk whenever they fit at all -- that is what this form is selected for)
    if (pairs_2wg >= 4 && pairs_2wg < max_pairs && !(inplace && (C + 1) / 2 <= max_pairs)) max_pairs = pairs_2wg;
    if (const char* e = getenv("FDIFF_FFT_MAXPAIRS")) max_pairs = std::max(1, std::min(max_pairs, atoi(e)));      // experiments
    int Cc = C;
    if ((C + 1) / 2 > max_pairs) Cc = max_pairs * 2;
    // single-channel sets: Cc consecutive series per workgroup (BATCHED), as many as keep >= 2 workgroups per CU busy,
    // at most 32 (16 complex lanes)
    const bool batched = (C == 1 && B >= 4);
    if (batched) {
        Cc = std::min(32, std::max(2, B / (2 * ctx->num_cu)));
        Cc = std::min(Cc & ~1, max_pairs * 2);
        Cc = std::max(Cc, 2);
    }
    const int nchunks = batched ? 1 : (C + Cc - 1) / Cc;
    const int ngroups = batched ? (B + Cc - 1) / Cc : B;
    const size_t lds = fixed + (inplace ? 1 : 2) * (size_t)T * ((Cc + 1) / 2) * 8;
    bool bigp = false;
    for (int i = 0; i < plan.nstages; ++i) bigp |= (plan.radix[i] == 17 || plan.radix[i] == 19 || plan.radix[i] == 23);
    void (*kern)(const float*, float*, const float*, const float*, const float2*, int, int, int, FftPlan);
    if (inplace) kern = batched ? k_fft<INVERSE, true, false, true> : k_fft<INVERSE, false, false, true>;
    else kern = bigp ? (batched ? k_fft<INVERSE, true, true> : k_fft<INVERSE, false, true>)
                     : (batched ? k_fft<INVERSE, true, false> : k_fft<INVERSE, false, false>);
    static unsigned long long attr_set[2][2][2][2] = {};
    if (fd_first_on_device(attr_set[INVERSE ? 1 : 0][batched ? 1 : 0][bigp ? 1 : 0][inplace ? 1 : 0], ctx->device))
        FD_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // twiddle table of this T (cached on the context; a handful of distinct T per process)
    const float2* tw_dev = nullptr;
    for (auto& e : ctx->fft_tw)
        if (e.first == T) tw_dev = reinterpret_cast<const float2*>(e.second);
    if (!tw_dev) {
        std::vector<float2> h(T);
        for (int k = 0; k < T; ++k) {
            const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)T;
            h[k] = float2{(float)std::cos(a), (float)std::sin(a)};
        }
        void* d = nullptr;
        FD_HIP(ctx, hipMalloc(&d, sizeof(float2) * (size_t)T));
        FD_HIP(ctx, hipMemcpy(d, h.data(), sizeof(float2) * (size_t)T, hipMemcpyHostToDevice));
        ctx->fft_tw.emplace_back(T, d);
        tw_dev = reinterpret_cast<const float2*>(d);
    }
    // threads per workgroup: about 4 complex elements per thread and stage, whole waves, 128..1024
    const int elems = T * ((Cc + 1) / 2);
    int block = 128;
    static const int ept_env = getenv("FDIFF_FFT_EPT") ? atoi(getenv("FDIFF_FFT_EPT")) : 0;
    const int ept = ept_env > 0 ? ept_env : (inplace ? 16 : 8);
    while (block < kMaxBlock && block * ept < elems) block *= 2;
    {
        const int wg_per_cu = std::max(1, std::min((int)((size_t)160 * 1024 / (lds + 256)), 2048 / block));
        plan.resident = ctx->num_cu * wg_per_cu;
        const long long nwg = (long long)ngroups * nchunks;
        plan.vec_ok = (!batched && (C & 1) == 0 && (Cc & 1) == 0 && ((uintptr_t)x & 7) == 0 && ((uintptr_t)y & 7) == 0) ? 1 : 0;
        const uintptr_t al16 = (uintptr_t)x | (uintptr_t)y | (uintptr_t)mean | (uintptr_t)stdv;
        static const bool no_vec4 = getenv("FDIFF_FFT_VEC4") && getenv("FDIFF_FFT_VEC4")[0] == '0';
        // (even T only: the row buffer follows T float2 twiddles in LDS, so an odd T leaves it 8-byte aligned and every
        //  float4 LDS access of the vec4 passes would be a misaligned ds_*_b128)
        if (plan.vec_ok && (C & 3) == 0 && (Cc & 3) == 0 && (al16 & 15) == 0 && (T & 1) == 0 && !no_vec4) plan.vec_ok = 2;
        const char* e = getenv("FDIFF_FFT_STAGGER");
        plan.stagger = e ? atoi(e) : 0;
        if (nwg < 3LL * plan.resident) plan.stagger = 0;           // short launches: the spread would cost more than it hides
    }
    hipLaunchKernelGGL(kern, dim3(ngroups, nchunks), dim3(block), lds, (hipStream_t)stream, x, y, mean, stdv, tw_dev, B, C,
                       Cc, plan);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

}  // namespace

extern "C" int fd_rfft_pack(fd_ctx* ctx, const float* x, float* y, int B, int T, int C, void* stream) {
    return launch<false>(ctx, x, y, nullptr, nullptr, B, T, C, stream, "fd_rfft_pack");
}
extern "C" int fd_irfft_unpack(fd_ctx* ctx, const float* x, float* y, int B, int T, int C, void* stream) {
    return launch<true>(ctx, x, y, nullptr, nullptr, B, T, C, stream, "fd_irfft_unpack");
}
extern "C" int fd_rfft_pack_standardize(fd_ctx* ctx, const float* x, const float* mean, const float* std, float* y,
                                        int B, int T, int C, void* stream) {
    if (ctx && !(mean && std)) return fd_fail(ctx, FD_ERR_ARG, "fd_rfft_pack_standardize: null mean/std");
    return launch<false>(ctx, x, y, mean, std, B, T, C, stream, "fd_rfft_pack_standardize");
}
extern "C" int fd_destandardize_irfft(fd_ctx* ctx, const float* x, const float* mean, const float* std, float* y,
                                      int B, int T, int C, void* stream) {
    if (ctx && !(mean && std)) return fd_fail(ctx, FD_ERR_ARG, "fd_destandardize_irfft: null mean/std");
    return launch<true>(ctx, x, y, mean, std, B, T, C, stream, "fd_destandardize_irfft");
}

// ------------------------------------------------------------------------------------------------------------
// Spectral utilities of the reference's fourier.py on the packed representation (dataset front-end).
// All three work on the (B, T, C) real spectral layout rfft_pack writes: rows [0, n_real) = Re X_k, rows
// [n_real, T) = Im X_k for k = 1..; Im X_0 (and Im X_{T/2}, T even) are zero by construction.
namespace {

__device__ __forceinline__ float packed_im(const float* __restrict__ xt_b, int k, int T, int n_real, int C, int c) {
    const bool has_im = (k != 0) && !((T & 1) == 0 && k == T / 2);
    return has_im ? xt_b[(size_t)(n_real + k - 1) * C + c] : 0.f;
}

// dens[b, k, c] = Re X_k^2 + Im X_k^2, k = 0..n_real-1                       (fourier.py:90-124)
__global__ __launch_bounds__(256) void k_spectral_density(const float* __restrict__ xt, float* __restrict__ dens, int B,
                                                           int T, int C, int n_real) {
    const size_t per_b = (size_t)n_real * C;
    const size_t id = blockIdx.x * (size_t)256 + threadIdx.x;
    if (id >= (size_t)B * per_b) return;
    const int b = (int)(id / per_b);
    const int r = (int)(id - (size_t)b * per_b);
    const int k = r / C, c = r - k * C;
    const float* xb = xt + (size_t)b * T * C;
    const float re = xb[(size_t)k * C + c];
    const float im = packed_im(xb, k, T, n_real, C, c);
    dens[id] = re * re + im * im;
}

// Delocalisation of each series in time and in frequency (fourier.py:127-175): with e_t the normalised energy per time
// step (resp. per frequency bin of the two-sided spectrum) and d(t, s) = min(|t - s|, T - |t - s|),
//   loc = min_s sum_t e_t d(t, s)^2.          One workgroup per series; e in LDS.
__global__ __launch_bounds__(256) void k_localization(const float* __restrict__ x, const float* __restrict__ xt,
                                                       float* __restrict__ loc, float* __restrict__ spec_loc, int T, int C,
                                                       int n_real) {
    extern __shared__ float sh[];          // [T] time energy | [T] two-sided spectral energy | [8] reduction | [2T] distances
    float* et = sh;
    float* es = sh + T;
    float* red = sh + 2 * T;
    float* dd = sh + 2 * T + 8;            // dd[T + k] = min(|k|, T - |k|)^2 for k = -T+1 .. T-1: the centre search below is
                                           // the circular correlation of the energies with this table
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (size_t)b * T * C;
    const float* sb = xt + (size_t)b * T * C;
    for (int i = tid; i < 2 * T; i += 256) {
        const int ad = abs(i - T);
        const float d = (float)min(ad, T - ad);
        dd[i] = d * d;
    }
    for (int t = tid; t < T; t += 256) {
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(xb[(size_t)t * C + c], xb[(size_t)t * C + c], a);
        et[t] = a;
        // two-sided spectrum: bins 0..n_real-1, then the mirror of bins 1..K (K = n_real-1 for odd T, n_real-2 for even T)
        const int k = (t < n_real) ? t : T - t;
        float d = 0.f;
        for (int c = 0; c < C; ++c) {
            const float re = sb[(size_t)k * C + c], im = packed_im(sb, k, T, n_real, C, c);
            d += re * re + im * im;
        }
        es[t] = d;
    }
    __syncthreads();
    // totals (fixed order: strided partials, wave shuffle, then the 4 wave sums)
    float tot[2] = {0.f, 0.f};
    for (int t = tid; t < T; t += 256) { tot[0] += et[t]; tot[1] += es[t]; }
    for (int o = 32; o > 0; o >>= 1) { tot[0] += __shfl_down(tot[0], o); tot[1] += __shfl_down(tot[1], o); }
    if ((tid & 63) == 0) { red[(tid >> 6) * 2] = tot[0]; red[(tid >> 6) * 2 + 1] = tot[1]; }
    __syncthreads();
    const float tot_t = (red[0] + red[2]) + (red[4] + red[6]);
    const float tot_s = (red[1] + red[3]) + (red[5] + red[7]);
    __syncthreads();
    float best_t = INFINITY, best_s = INFINITY;
    for (int s = tid; s < T; s += 256) {
        float at = 0.f, as = 0.f;
        const float* dds = dd + T - s;                         // dds[t] = d(t, s)^2
#pragma unroll 4
        for (int t = 0; t < T; ++t) {
            const float d2 = dds[t];
            at = fmaf(et[t], d2, at);
            as = fmaf(es[t], d2, as);
        }
        best_t = fminf(best_t, at);
        best_s = fminf(best_s, as);
    }
    for (int o = 32; o > 0; o >>= 1) {
        best_t = fminf(best_t, __shfl_down(best_t, o));
        best_s = fminf(best_s, __shfl_down(best_s, o));
    }
    if ((tid & 63) == 0) { red[(tid >> 6) * 2] = best_t; red[(tid >> 6) * 2 + 1] = best_s; }
    __syncthreads();
    if (tid == 0) {
        loc[b] = fminf(fminf(red[0], red[2]), fminf(red[4], red[6])) / tot_t;
        spec_loc[b] = fminf(fminf(red[1], red[3]), fminf(red[5], red[7])) / tot_s;
    }
}

// Gaussian mixing matrix over the packed rows (fourier.py:189-200): row i of the representation carries frequency
// f_i = i (i < n_real) or i - n_real + 1; G[t][s] = exp(-((f_t - f_s) / sigma)^2 / 2) / sum_t' exp(...).  One block per s.
__global__ __launch_bounds__(256) void k_gauss_matrix(float* __restrict__ G, int T, int n_real, float sigma) {
    __shared__ float red[4];
    const int s = blockIdx.x, tid = threadIdx.x;
    const float fs = (float)(s < n_real ? s : s - n_real + 1);
    float part = 0.f;
    for (int t = tid; t < T; t += 256) {
        const float ft = (float)(t < n_real ? t : t - n_real + 1);
        const float u = (ft - fs) / sigma;
        part += expf(-(u * u) / 2.0f);
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int t = tid; t < T; t += 256) {
        const float ft = (float)(t < n_real ? t : t - n_real + 1);
        const float u = (ft - fs) / sigma;
        G[(size_t)t * T + s] = expf(-(u * u) / 2.0f) / tot;
    }
}

// out[b, s, c] = sum_t xt[b, t, c] G[t, s]                                     (fourier.py:203, einsum "btc,ts->bsc")
__global__ __launch_bounds__(256) void k_frequency_mix(const float* __restrict__ xt, const float* __restrict__ G,
                                                        float* __restrict__ out, int T, int C) {
    const int b = blockIdx.y;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= T * C) return;
    const int s = id / C, c = id - s * C;
    const float* xb = xt + (size_t)b * T * C;
    float a = 0.f;
    for (int t = 0; t < T; ++t) a = fmaf(xb[(size_t)t * C + c], G[(size_t)t * T + s], a);
    out[(size_t)b * T * C + id] = a;
}

// out[b, c, r] = in[b, r, c]  for nb matrices of (rows, cols): 32 x 32 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void k_transpose_batched(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                            int cols) {
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[base + (size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) out[base + (size_t)c * rows + r] = tile[tx][i];
    }
}

}  // namespace

// out[b, s, c] = sum_t in[b, t, c] Mx[t, s]  for a (T, T) device matrix: the Gaussian mixing of smooth_frequency and the
// dense form of the transform at lengths with a dominating prime factor.  Single channel: ONE fp32-MFMA GEMM
// (B, T) . (T, T).  Several channels: per chunk of series, (b, t, c) -> (b, c, t) [into `out`], one GEMM (nb*C, T) . (T, T) into
// the context's GEMM scratch, and back to (b, t, c) (3.8 ms -> 0.5 ms at (4096, 255, 28)); the per-element VALU kernel remains
// for shapes whose single series does not fit the scratch.  `in` and `out` must not alias.
void fd_apply_row_matrix(fd_ctx* ctx, const float* in, const float* Mx, float* out, int B, int T, int C, hipStream_t stream) {
    if (C == 1 && (long long)B * T < 2147483647LL) {
        fdgemm::Args g{in, Mx, out, nullptr, B, T, T, (long long)T, 1, (long long)T, 1, (long long)T, 1.0f, 0, 0};
        fdgemm::launch(g, stream);
        return;
    }
    size_t nscr = 0;
    float* scr = fd_gemm_scratch(ctx, &nscr);
    const size_t per_b = (size_t)T * C;
    const int chunk = scr ? (int)std::min<size_t>(nscr / per_b, 65535) : 0;
    if (chunk < 1 || (C + 31) / 32 > 65535) {
        for (int b0 = 0; b0 < B; b0 += 65535) {
            const int nb = std::min(65535, B - b0);
            hipLaunchKernelGGL(k_frequency_mix, dim3((T * C + 255) / 256, nb), dim3(256), 0, stream, in + (size_t)b0 * per_b, Mx,
                               out + (size_t)b0 * per_b, T, C);
        }
        return;
    }
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = std::min(chunk, B - b0);
        const float* xin = in + (size_t)b0 * per_b;
        float* o = out + (size_t)b0 * per_b;
        hipLaunchKernelGGL(k_transpose_batched, dim3((C + 31) / 32, (T + 31) / 32, nb), dim3(256), 0, stream, xin, o, T, C);
        fdgemm::Args g{o, Mx, scr, nullptr, nb * C, T, T, (long long)T, 1, (long long)T, 1, (long long)T, 1.0f, 0, 0};
        fdgemm::launch(g, stream);
        hipLaunchKernelGGL(k_transpose_batched, dim3((T + 31) / 32, (C + 31) / 32, nb), dim3(256), 0, stream, scr, o, C, T);
    }
}

extern "C" int fd_spectral_density(fd_ctx* ctx, const float* xt, float* dens, int B, int T, int C, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, xt && dens, "fd_spectral_density: null pointer");
    FD_REQUIRE(ctx, B > 0 && T > 0 && C > 0, "fd_spectral_density: bad shape B=%d T=%d C=%d", B, T, C);
    const int n_real = T / 2 + 1;
    const size_t n = (size_t)B * n_real * C;
    hipLaunchKernelGGL(k_spectral_density, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xt, dens, B, T,
                       C, n_real);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_localization_metrics(fd_ctx* ctx, const float* x, const float* xt, float* loc, float* spec_loc, int B,
                                       int T, int C, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, x && xt && loc && spec_loc, "fd_localization_metrics: null pointer");
    FD_REQUIRE(ctx, B > 0 && T > 0 && C > 0, "fd_localization_metrics: bad shape B=%d T=%d C=%d", B, T, C);
    FD_REQUIRE(ctx, (size_t)T * 16 + 32 <= 64 * 1024, "fd_localization_metrics: T=%d too long", T);
    hipLaunchKernelGGL(k_localization, dim3(B), dim3(256), (size_t)(4 * T + 8) * sizeof(float), (hipStream_t)stream, x, xt, loc,
                       spec_loc, T, C, T / 2 + 1);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_frequency_smooth(fd_ctx* ctx, const float* xt, float sigma, float* gauss_scratch, float* out, int B,
                                   int T, int C, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, xt && gauss_scratch && out, "fd_frequency_smooth: null pointer");
    FD_REQUIRE(ctx, xt != out, "fd_frequency_smooth: in-place mixing is not supported");
    FD_REQUIRE(ctx, B > 0 && T > 0 && C > 0, "fd_frequency_smooth: bad shape B=%d T=%d C=%d", B, T, C);
    FD_REQUIRE(ctx, B <= 65535, "fd_frequency_smooth: batch too large for one launch (%d > 65535)", B);
    FD_REQUIRE(ctx, sigma > 0.f, "fd_frequency_smooth: sigma=%f", (double)sigma);
    // the reference builds its frequency vector with T - 1 entries when T is even and then fails in the einsum
    // (fourier.py:192-203): only odd lengths are defined
    FD_REQUIRE(ctx, (T & 1) == 1, "fd_frequency_smooth: max_len=%d must be odd (fourier.py:192-203)", T);
    hipLaunchKernelGGL(k_gauss_matrix, dim3(T), dim3(256), 0, (hipStream_t)stream, gauss_scratch, T, T / 2 + 1, sigma);
    fd_apply_row_matrix(ctx, xt, gauss_scratch, out, B, T, C, (hipStream_t)stream);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}
