// Micro-benchmark: issue cost (cycles per wave instruction) of the VALU / transcendental / MFMA instructions the
// attention phase of k_mega is built from, alone and interleaved, with 1 or 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
#define TIMED(NAME, BODY)                                                                                   \
    __global__ void NAME(unsigned long long* out, float seed) {                                            \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, \
              a7 = seed + 7;                                                                                \
        typedef float f4 __attribute__((ext_vector_type(4)));                                              \
        typedef short s4 __attribute__((ext_vector_type(4)));                                              \
        typedef __bf16 b8 __attribute__((ext_vector_type(8)));                                             \
        f4 c0 = {seed, seed, seed, seed}, c1 = c0, c2 = c0, c3 = c0;                                        \
        s4 ka = {1, 2, 3, 4};                                                                               \
        b8 kb8 = {};                                                                                        \
        typedef float f2 __attribute__((ext_vector_type(2)));                                              \
        f2 p0 = {seed, seed}, p1 = p0, p2 = p0, p3 = p0;                                                    \
        __syncthreads();                                                                                    \
        unsigned long long t0 = __builtin_readcyclecounter();                                               \
        for (int it = 0; it < 16; ++it) {                                                                   \
            asm volatile(".rept %c[rep]\n" BODY "\n.endr"                                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(c0), \
                           "+v"(c1), "+v"(c2), "+v"(c3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)                \
                         : "v"(ka), "v"(kb8), [rep] "i"(REP));                                              \
        }                                                                                                   \
        unsigned long long t1 = __builtin_readcyclecounter();                                               \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                          \
        if ((threadIdx.x & 63) == 0) atomicMax(&out[2], t1 - t0);                                          \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[0] + c2[0] + c3[0] + p0[0] + p1[0] + p2[0] + p3[0] == 12345.f) out[1] = 1;    \
    }

// each body = 4 independent instructions
TIMED(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3")
TIMED(k_fma, "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3")
TIMED(k_max3, "v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1")
TIMED(k_cvt, "v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %6, %6, %7")
TIMED(k_pkfma, "v_pk_fma_f32 %12, %12, %12, %12\n v_pk_fma_f32 %13, %13, %13, %13\n v_pk_fma_f32 %14, %14, %14, %14\n v_pk_fma_f32 %15, %15, %15, %15")
TIMED(k_exp_fma, "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3")
TIMED(k_exp_fma2, "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %3, %3")
TIMED(k_mfma16, "v_mfma_f32_16x16x16_bf16 %8, %16, %16, %8\n v_mfma_f32_16x16x16_bf16 %9, %16, %16, %9\n v_mfma_f32_16x16x16_bf16 %10, %16, %16, %10\n v_mfma_f32_16x16x16_bf16 %11, %16, %16, %11")
TIMED(k_mfma32, "v_mfma_f32_16x16x32_bf16 %8, %17, %17, %8\n v_mfma_f32_16x16x32_bf16 %9, %17, %17, %9\n v_mfma_f32_16x16x32_bf16 %10, %17, %17, %10\n v_mfma_f32_16x16x32_bf16 %11, %17, %17, %11")
TIMED(k_mfma16_exp, "v_mfma_f32_16x16x16_bf16 %8, %16, %16, %8\n v_exp_f32 %0, %0\n v_mfma_f32_16x16x16_bf16 %9, %16, %16, %9\n v_exp_f32 %1, %1")
TIMED(k_mfma32_d1, "v_mfma_f32_16x16x32_bf16 %8, %17, %17, %8\n v_mfma_f32_16x16x32_bf16 %8, %17, %17, %8\n v_mfma_f32_16x16x32_bf16 %8, %17, %17, %8\n v_mfma_f32_16x16x32_bf16 %8, %17, %17, %8")
TIMED(k_mfma32_d2, "v_mfma_f32_16x16x32_bf16 %8, %17, %17, %8\n v_mfma_f32_16x16x32_bf16 %9, %17, %17, %9\n v_mfma_f32_16x16x32_bf16 %8, %17, %17, %8\n v_mfma_f32_16x16x32_bf16 %9, %17, %17, %9")
TIMED(k_mfma32_d2v, "v_mfma_f32_16x16x32_bf16 %8, %17, %17, %8\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_mfma_f32_16x16x32_bf16 %9, %17, %17, %9\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3")
TIMED(k_mfma32_d4v, "v_mfma_f32_16x16x32_bf16 %8, %17, %17, %8\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_mfma_f32_16x16x32_bf16 %9, %17, %17, %9\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_mfma_f32_16x16x32_bf16 %10, %17, %17, %10\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_mfma_f32_16x16x32_bf16 %11, %17, %17, %11\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7")
TIMED(k_ldexp, "v_ldexp_f32 %0, %0, %1\n v_ldexp_f32 %2, %2, %3\n v_ldexp_f32 %4, %4, %5\n v_ldexp_f32 %6, %6, %7")
TIMED(k_cvti, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3")
TIMED(k_pkmul, "v_pk_mul_f32 %12, %12, %12\n v_pk_mul_f32 %13, %13, %13\n v_pk_mul_f32 %14, %14, %14\n v_pk_mul_f32 %15, %15, %15")
TIMED(k_exp16, "v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3")
TIMED(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3")

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    struct { const char* n; void (*k)(unsigned long long*, float); } ks[] = {
        {"v_exp_f32", k_exp}, {"v_fma_f32", k_fma}, {"v_max3_f32", k_max3}, {"v_cvt_pk_bf16_f32", k_cvt},
        {"v_pk_fma_f32", k_pkfma}, {"v_pk_mul_f32", k_pkmul}, {"1 exp + 3 fma", k_exp_fma}, {"2 exp + 2 fma", k_exp_fma2},
        {"mfma 16x16x16 bf16", k_mfma16}, {"mfma 16x16x32 bf16", k_mfma32}, {"2 mfma16 + 2 exp", k_mfma16_exp},
        {"mfma32 dependent d=1 (x4)", k_mfma32_d1}, {"mfma32 dependent d=2 (x4)", k_mfma32_d2}, {"2 mfma32 d=2 + 4 fma", k_mfma32_d2v}, {"4 mfma32 d=4 + 8 fma", k_mfma32_d4v},
        {"v_ldexp_f32", k_ldexp}, {"v_cvt_i32_f32", k_cvti}, {"v_exp_f16", k_exp16}, {"v_rcp_f32", k_rcp}};
    for (int threads : {256, 512}) {
        printf("--- %d threads per CU (%d wave(s) per SIMD), 1 workgroup\n", threads, threads / 256);
        for (auto& e : ks) {
            hipMemset(d, 0, 64);
            hipLaunchKernelGGL(e.k, dim3(1), dim3(threads), 0, 0, d, 0.5f);
            hipLaunchKernelGGL(e.k, dim3(1), dim3(threads), 0, 0, d, 0.5f);
            unsigned long long h[3];
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            // s_memtime ticks at a constant 100 MHz on gfx9: convert through the shader clock measured by a reference
            printf("%-28s wave0 %8.2f  slowest wave %8.2f cycles per group\n", e.n, (double)h[0] / (16.0 * REP), (double)h[2] / (16.0 * REP));
        }
    }
    return 0;
}
