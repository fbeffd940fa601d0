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
#include "fd_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxStages = 16;

struct FftPlan {
    int T;
    int nstages;
    int radix[kMaxStages];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// One Stockham stage, radix r, Ns = product of earlier radices.  Every thread produces one output.
__device__ __forceinline__ void stockham_stage(const float2* __restrict__ in, float2* __restrict__ out,
                                               const float2* __restrict__ tw, int T, int Cp, int r, int Ns) {
    const int tr = T / r;
    const int ktw = T / (Ns * r);
    const int total = T * Cp;
    for (int id = threadIdx.x; id < total; id += kBlock) {
        const int q = id / Cp, p = id - q * Cp;
        const int blk = q / (Ns * r), rem = q - blk * (Ns * r);
        const int u = rem / Ns, k = rem - u * Ns;
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

// INVERSE == false : x (time)  -> y (packed spectrum), optional (y - mean)/std
// INVERSE == true  : x (packed spectrum), optional x*std + mean  -> y (time)
template <bool INVERSE>
__global__ __launch_bounds__(kBlock) void k_fft(const float* __restrict__ x, float* __restrict__ y,
                                                 const float* __restrict__ mean, const float* __restrict__ stdv,
                                                 int B, int C, int Cc, FftPlan plan) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = plan.T;
    const int b = blockIdx.x;
    const int c0 = blockIdx.y * Cc;
    const int cc = min(Cc, C - c0);          // channels in this chunk
    const int Cp = (cc + 1) >> 1;            // complex lanes (channel pairs)
    float2* tw = reinterpret_cast<float2*>(smem);
    float2* bufA = tw + T;
    float2* bufB = bufA + (size_t)T * ((Cc + 1) >> 1);
    const float sgn = INVERSE ? 1.0f : -1.0f;
    for (int k = threadIdx.x; k < T; k += kBlock) {
        double s, c;
        sincospi(2.0 * (double)k / (double)T, &s, &c);
        tw[k] = float2{(float)c, sgn * (float)s};
    }
    const float* xb = x + (size_t)b * T * C;
    float* yb = y + (size_t)b * T * C;
    const int n_real = T / 2 + 1;
    const bool even = (T & 1) == 0;
    const float scale = rsqrtf((float)T);

    // ---- load pass: build z[n][p]
    for (int id = threadIdx.x; id < T * Cp; id += kBlock) {
        const int n = id / Cp, p = id - n * Cp;
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
        stockham_stage(src, dst, tw, T, Cp, r, Ns);
        Ns *= r;
        __syncthreads();
        float2* tmp = src;
        src = dst;
        dst = tmp;
    }

    // ---- store pass
    if (!INVERSE) {
        // X_a[k] = (Z[k] + conj Z[T-k]) / 2 ,  X_b[k] = (Z[k] - conj Z[T-k]) / (2i)
        for (int id = threadIdx.x; id < n_real * Cp; id += kBlock) {
            const int k = id / Cp, p = id - k * Cp;
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
        for (int id = threadIdx.x; id < T * Cp; id += kBlock) {
            const int n = id / Cp, p = id - n * Cp;
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
               who, T);
    int max_pairs = (int)((lds_cap - (size_t)T * 8) / ((size_t)T * 16));
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
    hipLaunchKernelGGL(kern, dim3(B, nchunks), dim3(kBlock), lds, (hipStream_t)stream, x, y, mean, stdv, B, C, Cc,
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
