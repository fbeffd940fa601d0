// fd_philox.h -- Philox4x32-10 counter RNG + Box-Muller, device side.
// One counter value yields 4 standard normals; element e of a tensor uses counter
// (offset + e/4) and lane e%4, so any kernel shape reproduces the same stream.
#pragma once
#ifdef __HIPCC_RTC__      // run-time compilation (fd_mega_rtc.hip): the HIP device runtime is implicit, host headers do not exist
typedef unsigned int uint32_t;
typedef unsigned long uint64_t;
#else
#include <hip/hip_runtime.h>

#include <cstdint>
#endif

struct fd_u4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ fd_u4 fd_philox4x32_10(uint64_t counter, uint64_t seed) {
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // full 32x32->64 products: ONE v_mad_u64_u32 each instead of a v_mul_hi_u32 + v_mul_lo_u32 pair (all three are
        // quarter-rate; the multiplies are ~2/3 of a Philox evaluation)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return fd_u4{c0, c1, c2, c3};
}

// uniform in (0,1): 24 random bits, centred
__device__ __forceinline__ float fd_u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void fd_box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u1 = fd_u01(a), u2 = fd_u01(b);
    // v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32 (1 ulp; the sine unit takes its argument in revolutions, i.e. computes
    // sin(2 pi u2) from u2 directly): ~12 instructions per pair of normals instead of ~90 with the range-reducing sincospif and
    // the IEEE square root -- the draws are N(0,1) to 1e-6, which is what the sampler needs (the reference's torch.randn is
    // not bit-matched anyway: SURVEY 7.2)
    const float r = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1));
    n0 = r * __builtin_amdgcn_cosf(u2);
    n1 = r * __builtin_amdgcn_sinf(u2);
}

__device__ __forceinline__ void fd_randn4(uint64_t counter, uint64_t seed, float (&n)[4]) {
    const fd_u4 r = fd_philox4x32_10(counter, seed);
    fd_box_muller(r.x, r.y, n[0], n[1]);
    fd_box_muller(r.z, r.w, n[2], n[3]);
}

// 16 dropout decisions from one Philox4x32-10 evaluation: decision e looks at the 16-bit window at byte offset e of the
// 128-bit output (wrapping), bit e = (window >= thr16), i.e. kept with probability 1 - thr16 / 65536 exactly as with disjoint
// 16-bit fields (thr16 = round(p * 65536): p = 0.1 -> 0.100006; the rescale uses the exact keep probability).  Every byte is
// the HIGH byte of exactly one window, so a decision is settled by its own byte unless that byte equals thr16 >> 8 (1 case in
// 256), where the neighbouring byte breaks the tie: marginals exact to 2^-16, dependence between neighbours only through
// those ties.  Half the Philox evaluations of the 8-per-call form (they are the cost of dropout: ~560 issue cycles each,
// 10 M of them per layer at 16 K tokens).
__device__ __forceinline__ unsigned fd_drop16(uint64_t ctr, uint64_t seed, unsigned thr16) {
    const fd_u4 r = fd_philox4x32_10(ctr, seed);
    const unsigned w[5] = {r.x, r.y, r.z, r.w, r.x};
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned v0 = w[i] & 0xffffu, v1 = (w[i] >> 8) & 0xffffu, v2 = w[i] >> 16;
        const unsigned v3 = __builtin_amdgcn_alignbit(w[i + 1], w[i], 24) & 0xffffu;
        m |= (v0 >= thr16 ? 1u : 0u) << (4 * i);
        m |= (v1 >= thr16 ? 1u : 0u) << (4 * i + 1);
        m |= (v2 >= thr16 ? 1u : 0u) << (4 * i + 2);
        m |= (v3 >= thr16 ? 1u : 0u) << (4 * i + 3);
    }
    return m;
}
