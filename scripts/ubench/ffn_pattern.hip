// Micro-benchmark of the FFN item schedule of k_mega: does the relu VALU work overlap the MFMAs when two waves share
// a SIMD?  Variants: A = H(next) | relu(cur) block | W2(cur);  B = relu pieces interleaved between the H MFMAs;
// C = MFMAs only (no VALU);  D = A with the VALU block made independent of the MFMA results.
//   hipcc -O3 --offload-arch=gfx950 ffn_pattern.hip -o ffn_pattern && ./ffn_pattern
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define MF "v_mfma_f32_16x16x32_bf16 "
// fixed registers: hA0 v[100:103] hA1 v[104:107] hB0 v[108:111] hB1 v[112:115] hb v[116:119] acc v[120:139]
//                  w v[140:143] x v[144:147] independent scratch v[148:151]
#define HA0 "v[100:103]"
#define HA1 "v[104:107]"
#define HB0 "v[108:111]"
#define HB1 "v[112:115]"
#define WW "v[140:143]"
#define XX "v[144:147]"
#define H_INTO(a, b) MF a ", " WW ", " XX ", 0\n" MF b ", " WW ", " XX ", 0\n" MF a ", " WW ", " XX ", " a "\n" MF b ", " WW ", " XX ", " b "\n" MF a ", " WW ", " XX ", " a "\n" MF b ", " WW ", " XX ", " b "\n"
#define PIECE(lo, hi, r) "v_cvt_pk_bf16_f32 " r ", " lo ", " hi "\n v_pk_max_i16 " r ", " r ", 0\n"
#define RELU_A PIECE("v100", "v101", "v116") PIECE("v102", "v103", "v117") PIECE("v104", "v105", "v118") PIECE("v106", "v107", "v119")
#define RELU_B PIECE("v108", "v109", "v116") PIECE("v110", "v111", "v117") PIECE("v112", "v113", "v118") PIECE("v114", "v115", "v119")
#define RELU_I PIECE("v148", "v149", "v116") PIECE("v150", "v151", "v117") PIECE("v148", "v149", "v118") PIECE("v150", "v151", "v119")
#define W2 MF "v[120:123], " WW ", v[116:119], v[120:123]\n" MF "v[124:127], " WW ", v[116:119], v[124:127]\n" MF "v[128:131], " WW ", v[116:119], v[128:131]\n" MF "v[132:135], " WW ", v[116:119], v[132:135]\n" MF "v[136:139], " WW ", v[116:119], v[136:139]\n"

#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", \
             "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139", \
             "v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151"

#define KERNEL(NAME, BODY)                                                                                              \
    __global__ void NAME(unsigned long long* out, float seed) {                                                        \
        asm volatile(".irp r,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151\n v_mov_b32 v\\r, 0\n.endr" ::: CLOB); \
        __syncthreads();                                                                                                \
        unsigned long long t0 = __builtin_readcyclecounter();                                                           \
        for (int it = 0; it < 64; ++it) {                                                                               \
            asm volatile(".rept 8\n" BODY "\n.endr" ::: CLOB);                                                          \
        }                                                                                                               \
        unsigned long long t1 = __builtin_readcyclecounter();                                                           \
        if ((threadIdx.x & 63) == 0) atomicMax(&out[0], t1 - t0);                                                       \
        if (threadIdx.x == 0) out[1] = t1 - t0;                                                                         \
    }

// one .rept body = 2 items (even item writes hA and consumes hB, odd item the reverse)
KERNEL(k_A, H_INTO(HA0, HA1) RELU_B "s_nop 1\n" W2 H_INTO(HB0, HB1) RELU_A "s_nop 1\n" W2)
KERNEL(k_C, H_INTO(HA0, HA1) W2 H_INTO(HB0, HB1) W2)
KERNEL(k_D, H_INTO(HA0, HA1) RELU_I "s_nop 1\n" W2 H_INTO(HB0, HB1) RELU_I "s_nop 1\n" W2)
#define H_RELU(a, b, p0, p1, p2, p3) MF a ", " WW ", " XX ", 0\n" p0 MF b ", " WW ", " XX ", 0\n" p1 MF a ", " WW ", " XX ", " a "\n" p2 MF b ", " WW ", " XX ", " b "\n" p3 MF a ", " WW ", " XX ", " a "\n" MF b ", " WW ", " XX ", " b "\n"
KERNEL(k_B, H_RELU(HA0, HA1, PIECE("v108", "v109", "v116"), PIECE("v110", "v111", "v117"), PIECE("v112", "v113", "v118"), PIECE("v114", "v115", "v119")) W2
            H_RELU(HB0, HB1, PIECE("v100", "v101", "v116"), PIECE("v102", "v103", "v117"), PIECE("v104", "v105", "v118"), PIECE("v106", "v107", "v119")) W2)

// ---- step-structured variants: one step = 4 items (pattern A) [+ barrier] [+ 11 ds_read_b128 prefetched mid-step]
//      [+ 6 LDS-DMA instructions by alternating wave sets], like the FFN loop of k_mega
#define ITEM_A H_INTO(HA0, HA1) RELU_B "s_nop 1\n" W2
#define ITEM_B H_INTO(HB0, HB1) RELU_A "s_nop 1\n" W2
#define DSREADS "ds_read_b128 v[152:155], %0\n ds_read_b128 v[156:159], %0 offset:1024\n ds_read_b128 v[160:163], %0 offset:2048\n ds_read_b128 v[164:167], %0 offset:3072\n" \
                "ds_read_b128 v[168:171], %0 offset:4096\n ds_read_b128 v[172:175], %0 offset:5120\n ds_read_b128 v[176:179], %0 offset:6144\n ds_read_b128 v[180:183], %0 offset:7168\n" \
                "ds_read_b128 v[184:187], %0 offset:8192\n ds_read_b128 v[188:191], %0 offset:9216\n ds_read_b128 v[192:195], %0 offset:10240\n"
#define CLOB2 CLOB, "v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171", \
              "v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195"

template <int BAR, int LDSR, int DMA>
__global__ __launch_bounds__(512) void k_step(unsigned long long* out, const char* gsrc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile(".irp r,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151\n v_mov_b32 v\\r, 0\n.endr" ::: CLOB);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lds_addr = (unsigned)(lane * 16 + (wave >> 2) * 11264);
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int st = 0; st < 256; ++st) {
        if (DMA == 1 && ((st & 1) == (wave >> 2))) {
            const char* src = gsrc + (size_t)(st & 7) * 22528 + lane * 16;
            char* dst = smem + 32768 + (st & 3) * 22528;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int b = (wave & 3) + 4 * i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (b % 22) * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + (b % 22) * 1024), 16, 0, 0);
            }
        }
        if (DMA == 2) {   // plain loads into registers now, ds_write_b128 of the previous batch (the other wave set's turn)
            if ((st & 1) == (wave >> 2)) {
                const char* src = gsrc + (size_t)(st & 7) * 22528 + lane * 16 + (wave & 3) * 1024;
#define GLD(r, i) asm volatile("global_load_dwordx4 v[" #r ":" #r "+3], %0, off" ::"v"(src + (i) * 4096) : "memory")
                GLD(200, 0); GLD(204, 1); GLD(208, 2); GLD(212, 3); GLD(216, 4); GLD(220, 5);
            } else {
                const unsigned dst = 32768 + (st & 3) * 22528 + lane * 16 + (wave & 3) * 1024;
#define DSW(r, i) asm volatile("ds_write_b128 %0, v[" #r ":" #r "+3]" ::"v"(dst + (i) * 4096) : "memory")
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                DSW(200, 0); DSW(204, 1); DSW(208, 2); DSW(212, 3); DSW(216, 4); DSW(220, 5);
            }
        }
        asm volatile(ITEM_A ITEM_B ::: CLOB);
        if (LDSR) asm volatile(DSREADS ::"v"(lds_addr) : CLOB2);
        asm volatile(ITEM_A ITEM_B ::: CLOB);
        if (LDSR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (DMA == 1) {
            if ((st & 1) == (wave >> 2)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) atomicMax(&out[0], t1 - t0);
    if (threadIdx.x == 0) out[1] = t1 - t0;
}

// ---- mixed pairing: does an MFMA-heavy wave (FFN items) overlap a VALU-heavy wave (attention pass-2 stages) on the
//      same SIMD?  mode 0: all 8 waves FFN; 1: all 8 waves attention; 2: waves 0-3 FFN, waves 4-7 attention (one of each per SIMD)
#define ATT_STAGE "v_mfma_f32_16x16x16_bf16 v[152:155], v[140:141], v[144:145], v[156:159]\n v_exp_f32 v160, v164\n v_exp_f32 v161, v165\n v_exp_f32 v162, v166\n v_exp_f32 v163, v167\n" \
                  "v_cvt_pk_bf16_f32 v168, v160, v161\n v_cvt_pk_bf16_f32 v169, v162, v163\n" \
                  "v_mfma_f32_16x16x16_bf16 v[164:167], v[140:141], v[144:145], v[156:159]\n v_exp_f32 v160, v152\n v_exp_f32 v161, v153\n v_exp_f32 v162, v154\n v_exp_f32 v163, v155\n" \
                  "v_cvt_pk_bf16_f32 v170, v160, v161\n v_cvt_pk_bf16_f32 v171, v162, v163\n" MF "v[172:175], " WW ", v[168:171], v[172:175]\n"
template <int MODE>
__global__ __launch_bounds__(512) void k_mix(unsigned long long* out, int n_ffn, int n_att) {
    asm volatile(".irp r,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175\n v_mov_b32 v\\r, 0\n.endr" ::: CLOB2);
    const int wave = threadIdx.x >> 6;
    const bool ffn = (MODE == 0) || (MODE == 2 && wave < 4);
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (ffn) {
        for (int it = 0; it < n_ffn; ++it) asm volatile(ITEM_A ITEM_B ::: CLOB);
    } else {
        for (int it = 0; it < n_att; ++it) asm volatile(ATT_STAGE ATT_STAGE ::: CLOB2);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) atomicMax(&out[0], t1 - t0);
    if (threadIdx.x == 0) out[1] = t1 - t0;
    if (threadIdx.x == 256) out[2] = t1 - t0;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    struct { const char* n; void (*k)(unsigned long long*, float); } ks[] = {
        {"A: H | relu block | W2", k_A}, {"B: relu interleaved in H", k_B}, {"C: MFMAs only", k_C},
        {"D: A, relu independent of MFMA", k_D}};
    for (int threads : {256, 512}) {
        printf("--- %d wave(s) per SIMD\n", threads / 256);
        for (auto& e : ks) {
            hipMemset(d, 0, 64);
            hipLaunchKernelGGL(e.k, dim3(1), dim3(threads), 0, 0, d, 0.5f);
            hipMemset(d, 0, 64);
            hipLaunchKernelGGL(e.k, dim3(1), dim3(threads), 0, 0, d, 0.5f);
            unsigned long long h[3];
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            const double items = 64.0 * 8 * 2;
            printf("%-34s wave0 %7.1f  slowest wave %7.1f cycles per item (11 MFMA = 176 of pipe time)\n", e.n, h[1] / items, h[0] / items);
        }
    }
    {
        printf("--- mixed pairing on each SIMD (cycles for the whole job; FFN job = 2000 items per wave, attention job = 2000 double stages per wave)\n");
        struct { const char* n; void (*k)(unsigned long long*, int, int); int nf, na; } ms[] = {
            {"8 waves FFN, full job each", k_mix<0>, 1000, 0}, {"8 waves attention, full job each", k_mix<1>, 0, 1000},
            {"4 waves FFN + 4 waves attention, full job each", k_mix<2>, 1000, 1000},
            {"8 waves FFN, half job each", k_mix<0>, 500, 0}, {"8 waves attention, half job each", k_mix<1>, 0, 500}};
        for (auto& e : ms) {
            hipMemset(d, 0, 64);
            hipLaunchKernelGGL(e.k, dim3(1), dim3(512), 0, 0, d, e.nf, e.na);
            hipMemset(d, 0, 64);
            hipLaunchKernelGGL(e.k, dim3(1), dim3(512), 0, 0, d, e.nf, e.na);
            unsigned long long h[3];
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("%-48s wave0 %9llu  wave4 %9llu  slowest %9llu cycles\n", e.n, h[1], h[2], h[0]);
        }
    }
    {
        char* g;
        hipMalloc(&g, 1 << 20);
        hipMemset(g, 0, 1 << 20);
        struct { const char* n; void (*k)(unsigned long long*, const char*); } ss[] = {
            {"step: 4 items A, no barrier", k_step<0, 0, 0>}, {"step: + barrier", k_step<1, 0, 0>},
            {"step: + barrier + 11 ds_read_b128", k_step<1, 1, 0>}, {"step: + barrier + ds_read + DMA", k_step<1, 1, 1>},
            {"step: ds_read + DMA, no barrier", k_step<0, 1, 1>},
            {"step: + barrier + ds_read + load/ds_write", k_step<1, 1, 2>}};
        for (int nwg : {1, 256}) {
            printf("--- step-structured, 8 waves per workgroup, %d workgroup(s)\n", nwg);
            for (auto& e : ss) {
                hipFuncSetAttribute((const void*)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipMemset(d, 0, 64);
                hipLaunchKernelGGL(e.k, dim3(nwg), dim3(512), 140 * 1024, 0, d, g);
                hipMemset(d, 0, 64);
                hipLaunchKernelGGL(e.k, dim3(nwg), dim3(512), 140 * 1024, 0, d, g);
                unsigned long long h[3];
                hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
                printf("%-38s wave0 %7.1f  slowest wave %7.1f cycles per step (2 waves x 4 items x 176 = 1408 of pipe time)\n", e.n,
                       h[1] / 256.0, h[0] / 256.0);
            }
        }
    }
    return 0;
}
