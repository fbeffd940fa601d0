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

namespace {

constexpr int kMaxBlock = 1024;
constexpr int kMaxStages = 16;

struct FftPlan {
    int T;
    int nstages;
    int radix[kMaxStages];
};

// n / d for 0 <= n < 2^22 through the float pipe (4 VALU instead of the ~40 of a 32-bit integer division; the index
// arithmetic, not the butterflies, dominated this kernel): (n + 0.5) * (1/d) is at least 0.5/d away from an integer
// and carries an absolute error below n * 2^-23.
__device__ __forceinline__ int fdiv(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// One Stockham stage, radix r, Ns = product of earlier radices.  Every thread produces one output.
__device__ __forceinline__ void stockham_stage(const float2* __restrict__ in, float2* __restrict__ out,
                                               const float2* __restrict__ tw, int T, int Cp, int r, int Ns) {
    const int NT = blockDim.x;
    const int tr = T / r;
    const int ktw = T / (Ns * r);
    const int total = T * Cp;
    const float inv_cp = 1.0f / (float)Cp, inv_nsr = 1.0f / (float)(Ns * r), inv_ns = 1.0f / (float)Ns;
    for (int id = threadIdx.x; id < total; id += NT) {
        const int q = fdiv(id, inv_cp), p = id - q * Cp;
        const int blk = fdiv(q, inv_nsr), rem = q - blk * (Ns * r);
        const int u = fdiv(rem, inv_ns), k = rem - u * Ns;
        const int j = blk * Ns + k;
        int step = k * ktw + u * tr;
        step -= (step >= T) ? T : 0;           // k*ktw < T/r, u*tr < T  ->  < 2T
        float2 acc = in[j * Cp + p];
        int e = 0;
        for (int t = 1; t < r; ++t) {
            e += step;
            e -= (e >= T) ? T : 0;
            const float2 v = in[(j + t * tr) * Cp + p];
            const float2 w = tw[e];
            acc.x += v.x * w.x - v.y * w.y;
            acc.y += v.x * w.y + v.y * w.x;
        }
        out[q * Cp + p] = acc;
    }
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
    for (int id = threadIdx.x; id < total; id += NT) {
        const int j = fdiv(id, inv_cp), p = id - j * Cp;
        const int blk = fdiv(j, inv_ns), k = j - blk * Ns;
        float2 a[R];
#pragma unroll
        for (int t = 0; t < R; ++t) a[t] = in[(j + t * tr) * Cp + p];
        if (k != 0) {
            const int step = k * ktw;               // < T / R
#pragma unroll
            for (int t = 1; t < R; ++t) a[t] = cmul(a[t], tw[step * t]);   // step * t < T
        }
        float2* o = out + ((blk * R) * Ns + k) * Cp + p;                     // output u at o[u * Ns * Cp]
        const int os = Ns * Cp;
        if (R == 2) {
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

// INVERSE == false : x (time)  -> y (packed spectrum), optional (y - mean)/std
// INVERSE == true  : x (packed spectrum), optional x*std + mean  -> y (time)
template <bool INVERSE>
__global__ __launch_bounds__(kMaxBlock) void k_fft(const float* __restrict__ x, float* __restrict__ y,
                                                    const float* __restrict__ mean, const float* __restrict__ stdv,
                                                    const float2* __restrict__ tw_fwd, int B, int C, int Cc, FftPlan plan) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NT = blockDim.x;
    const int T = plan.T;
    const int b = blockIdx.x;
    const int c0 = blockIdx.y * Cc;
    const int cc = min(Cc, C - c0);          // channels in this chunk
    const int Cp = (cc + 1) >> 1;            // complex lanes (channel pairs)
    float2* tw = reinterpret_cast<float2*>(smem);
    float2* bufA = tw + T;
    float2* bufB = bufA + (size_t)T * ((Cc + 1) >> 1);
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
    const float scale = rsqrtf((float)T);

    // ---- load pass: build z[n][p]
    const float inv_cp = 1.0f / (float)Cp;
    for (int id = threadIdx.x; id < T * Cp; id += NT) {
        const int n = fdiv(id, inv_cp), p = id - n * Cp;
        const int ca = c0 + 2 * p;
        const bool has_b = (2 * p + 1) < cc;
        float2 z;
        if (!INVERSE) {
            z.x = xb[(size_t)n * C + ca];
            z.y = has_b ? xb[(size_t)n * C + ca + 1] : 0.f;
        } else {
            // Hermitian extension of the packed half spectrum (fourier.py:62-77): X[T-k] = conj X[k]
            const int kk = (n <= T / 2) ? n : T - n;
            const bool has_im = (kk != 0) && !(even && kk == T / 2);
            const float sg = (n <= T / 2) ? 1.0f : -1.0f;
            const size_t ire = (size_t)kk * C, iim = (size_t)(n_real + kk - 1) * C;
            float are = xb[ire + ca], aim = has_im ? xb[iim + ca] : 0.f;
            float bre = has_b ? xb[ire + ca + 1] : 0.f, bim = (has_b && has_im) ? xb[iim + ca + 1] : 0.f;
            if (mean) {   // de-standardise in the frequency domain (cmd/sample.py:76-78)
                are = are * stdv[ire + ca] + mean[ire + ca];
                if (has_im) aim = aim * stdv[iim + ca] + mean[iim + ca];
                if (has_b) {
                    bre = bre * stdv[ire + ca + 1] + mean[ire + ca + 1];
                    if (has_im) bim = bim * stdv[iim + ca + 1] + mean[iim + ca + 1];
                }
            }
            aim *= sg;
            bim *= sg;
            // Z = Xa + i Xb
            z.x = are - bim;
            z.y = aim + bre;
        }
        bufA[id] = z;
    }
    __syncthreads();

    // ---- Stockham stages (ping-pong)
    float2* src = bufA;
    float2* dst = bufB;
    int Ns = 1;
    for (int s = 0; s < plan.nstages; ++s) {
        const int r = plan.radix[s];
        switch (r) {                                  // (uniform: one stage, one radix for the whole workgroup)
            case 2: stockham_butterflies<2>(src, dst, tw, T, Cp, Ns); break;
            case 3: stockham_butterflies<3>(src, dst, tw, T, Cp, Ns); break;
            case 4: stockham_butterflies<4>(src, dst, tw, T, Cp, Ns); break;
            case 5: stockham_butterflies<5>(src, dst, tw, T, Cp, Ns); break;
            case 7: stockham_butterflies<7>(src, dst, tw, T, Cp, Ns); break;
            default: stockham_stage(src, dst, tw, T, Cp, r, Ns);   // large prime factor: one direct-DFT stage
        }
        Ns *= r;
        __syncthreads();
        float2* tmp = src;
        src = dst;
        dst = tmp;
    }

    // ---- store pass
    if (!INVERSE) {
        // X_a[k] = (Z[k] + conj Z[T-k]) / 2 ,  X_b[k] = (Z[k] - conj Z[T-k]) / (2i)
        for (int id = threadIdx.x; id < n_real * Cp; id += NT) {
            const int k = fdiv(id, inv_cp), p = id - k * Cp;
            const int ca = c0 + 2 * p;
            const bool has_b = (2 * p + 1) < cc;
            const float2 zk = src[k * Cp + p];
            const float2 zm = src[((k == 0) ? 0 : (T - k)) * Cp + p];
            const float h = 0.5f * scale;
            float are = (zk.x + zm.x) * h, aim = (zk.y - zm.y) * h;
            float bre = (zk.y + zm.y) * h, bim = (zm.x - zk.x) * h;
            const bool has_im = (k != 0) && !(even && k == T / 2);   // fourier.py:26-37 drop exact zeros
            const size_t ire = (size_t)k * C, iim = (size_t)(n_real + k - 1) * C;
            if (mean) {   // standardise (datamodules.py:61-62)
                are = (are - mean[ire + ca]) / stdv[ire + ca];
                if (has_im) aim = (aim - mean[iim + ca]) / stdv[iim + ca];
                if (has_b) {
                    bre = (bre - mean[ire + ca + 1]) / stdv[ire + ca + 1];
                    if (has_im) bim = (bim - mean[iim + ca + 1]) / stdv[iim + ca + 1];
                }
            }
            yb[ire + ca] = are;
            if (has_b) yb[ire + ca + 1] = bre;
            if (has_im) {
                yb[iim + ca] = aim;
                if (has_b) yb[iim + ca + 1] = bim;
            }
        }
    } else {
        for (int id = threadIdx.x; id < T * Cp; id += NT) {
            const int n = fdiv(id, inv_cp), p = id - n * Cp;
            const int ca = c0 + 2 * p;
            const float2 z = src[id];
            yb[(size_t)n * C + ca] = z.x * scale;
            if ((2 * p + 1) < cc) yb[(size_t)n * C + ca + 1] = z.y * scale;
        }
    }
}

bool make_plan(int T, FftPlan& plan) {
    plan.T = T;
    plan.nstages = 0;
    int n = T;
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

template <bool INVERSE>
int launch(fd_ctx* ctx, const float* x, float* y, const float* mean, const float* stdv, int B, int T, int C,
           void* stream, const char* who) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, x && y, "%s: null pointer", who);
    FD_REQUIRE(ctx, x != y, "%s: in-place transform is not supported", who);
    FD_REQUIRE(ctx, B > 0 && T > 0 && C > 0, "%s: bad shape B=%d T=%d C=%d", who, B, T, C);
    FD_REQUIRE(ctx, (mean == nullptr) == (stdv == nullptr), "%s: mean and std must come together", who);
    FD_REQUIRE(ctx, B <= 2147483647 / 2, "%s: batch too large", who);
    FftPlan plan;
    FD_REQUIRE(ctx, make_plan(T, plan), "%s: T=%d has too many prime factors", who, T);
    // LDS: twiddles (8T) + two complex images of T * ceil(Cc/2) float2
    const size_t lds_cap = 128 * 1024;
    FD_REQUIRE(ctx, (size_t)T * 8 + 2 * (size_t)T * 8 <= lds_cap, "%s: T=%d too long for the LDS-resident transform",
               who, T);   // (also keeps every index below 2^22, the range of fdiv)
    int max_pairs = (int)((lds_cap - (size_t)T * 8) / ((size_t)T * 16));
    // prefer <= 72 KiB per workgroup (two resident workgroups per CU overlap each other's load / store passes) as long
    // as a chunk keeps at least 8 channels = one 32-byte sector per time step
    const int pairs_2wg = (int)(((size_t)72 * 1024 - (size_t)T * 8) / ((size_t)T * 16));
    if (pairs_2wg >= 4 && pairs_2wg < max_pairs) max_pairs = pairs_2wg;
    int Cc = C;
    if ((C + 1) / 2 > max_pairs) Cc = max_pairs * 2;
    const int nchunks = (C + Cc - 1) / Cc;
    const size_t lds = (size_t)T * 8 + 2 * (size_t)T * ((Cc + 1) / 2) * 8;
    auto kern = k_fft<INVERSE>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[INVERSE ? 1 : 0]) {
        FD_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[INVERSE ? 1 : 0] = true;
    }
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
    while (block < kMaxBlock && block * 8 < elems) block *= 2;
    hipLaunchKernelGGL(kern, dim3(B, nchunks), dim3(block), lds, (hipStream_t)stream, x, y, mean, stdv, tw_dev, B, C, Cc,
                       plan);
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
