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
// Attention: units of (head pair) x (series) x (two query tiles); S^T = K Q^T by the K=16 MFMA with two heads sharing
// every K / V fragment (even head in k-slots / rows of lane groups 0-1, odd head in groups 2-3); the softmax shift
// (bound |q| max|k|, exact row maximum as fallback) rides in the MFMA's C operand, exp2 with the scale folded into W_q,
// the denominator comes out of the P V MFMAs through a row of ones in V^T; P is fed back as a B operand without
// leaving registers.  FFN: weights streamed L2 -> LDS through a 4-deep ring, hidden activations stay in registers.
// DESIGN.md sections 3.2 / 3.3 hold the measurements behind each of these choices.
//
// Reference arithmetic: src/fdiff/models/score_models.py:67-94 (+ torch TransformerEncoderLayer),
// src/fdiff/sampling/sampler.py:83-104, src/fdiff/schedulers/sde.py:129-165,215-246.
#include "fd_mega.h"
#include "fd_mega_kernel.h"
#include "fd_mega_rtc.h"

// ------------------------------------------------------------------------------------------ host side
void fd_mega_temb_table(const fd_mega_params& P, float* table, hipStream_t s) {
    hipLaunchKernelGGL(k_temb_table, dim3(P.nsteps), dim3(64), (size_t)P.D * sizeof(float), s, P.params, P.tW, P.td_w, P.td_b, P.steps,
                       table, P.D);
}

template <int KS1, int DT, int KSO, int MT, class SH, int NW = 8>
static int launch_mega_t(fd_ctx* ctx, const fd_mega_params& P, int grid, size_t lds, hipStream_t s) {
    auto kern = k_mega<KS1, DT, KSO, MT, SH, NW>;
    static unsigned long long attr = 0;
    if (fd_first_on_device(attr, ctx->device))
        FD_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (NW == 8 ? 160 : 80) * 1024));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, P);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

// Static-shape instantiations for the BASELINE.json workloads (default transformer: D=72, H=12, L=10, F=2048):
// configs[1] ecg (T=100, C=12): 2 series per workgroup, 2 head groups of 3 pairs;
// configs[2] nasdaq (T=252, C=6) and configs[3] mimiciii (T=256, C=28): 1 series (16 token tiles), 3 groups of 2 pairs.
#ifndef FD_MEGA_FFN32
#define FD_MEGA_FFN32 1      // pair-form FFN (32x32x16 H) in the static-shape instantiations; 0 = the 16x16x32 form (A/B builds)
#endif
using ShapeEcg = ShapeStatic<100, 72, 12, 12, 2, 3, 2, 10, 2048, FD_MEGA_FFN32>;
using ShapeNasdaq = ShapeStatic<252, 72, 6, 12, 1, 2, 1, 10, 2048, FD_MEGA_FFN32>;
using ShapeMimic = ShapeStatic<256, 72, 28, 12, 1, 2, 1, 10, 2048, FD_MEGA_FFN32>;

template <class SH>
static bool shape_matches(const fd_mega_params& P, int ks1, int dt, int kso, int mt) {
    return ks1 == 3 && dt == 5 && kso == 3 && mt == 4 && P.T == SH::T && P.D == SH::D && P.C == SH::C && P.H == SH::H &&
           P.S == SH::S && P.NPG == SH::NPG && P.rot == SH::rot && P.L == SH::L && P.F == SH::F &&
           (SH::FFN32 == 0 || P.off_ffn32 != 0);
}

// `describe` (>= 192 bytes, nullable): when given, the instantiation that WOULD run is written there and nothing is
// launched (fd_score_plan: the parity tests assert which kernel they exercised).
#define FD_MEGA_NOTE ""
#define FD_MEGA_GO(K, T_, O, M_, SH, NAME)                                                             \
    do {                                                                                               \
        if (describe) {                                                                                \
            snprintf(describe, 192, "k_mega<%d,%d,%d,%d,%s> S=%d NPG=%d rot=%d grid=%d lds=%zu%s%s", K, T_, O, M_, NAME, P.S, \
                     P.NPG, P.rot, grid, lds, SH::FFN32 ? " ffn32 (H by 32x32x16 MFMAs on token-tile pairs)" : "", FD_MEGA_NOTE); \
            return FD_OK;                                                                              \
        }                                                                                              \
        return launch_mega_t<K, T_, O, M_, SH>(ctx, P, grid, lds, s);                                  \
    } while (0)

int fd_mega_launch(fd_ctx* ctx, const fd_mega_params& P, int ks1, int dt, int kso, int mt, int nw, int grid, size_t lds,
                   hipStream_t s, char* describe) {
    if (nw != 8) return fd_fail(ctx, FD_ERR_UNSUPPORTED, "persistent kernel: only 8-wave workgroups are instantiated");
    const bool generic_only = getenv("FDIFF_MEGA_GENERIC") != nullptr;
    const int jit = generic_only ? FD_MEGA_RTC_OFF : fd_mega_rtc_mode();
    if (!generic_only && jit != FD_MEGA_RTC_FORCE) {
        if (shape_matches<ShapeEcg>(P, ks1, dt, kso, mt)) FD_MEGA_GO(3, 5, 3, 4, ShapeEcg, "ShapeStatic<100,72,12,12,2,3,2,10,2048>");
        if (shape_matches<ShapeNasdaq>(P, ks1, dt, kso, mt)) FD_MEGA_GO(3, 5, 3, 4, ShapeNasdaq, "ShapeStatic<252,72,6,12,1,2,1,10,2048>");
        if (shape_matches<ShapeMimic>(P, ks1, dt, kso, mt)) FD_MEGA_GO(3, 5, 3, 4, ShapeMimic, "ShapeStatic<256,72,28,12,1,2,1,10,2048>");
    }
    // Run-time specialisation (fd_mega_rtc.hip): a ShapeStatic instantiation for THIS (model class, series shape, plan), compiled by
    // hiprtc on first use and cached on disk.  AUTO: sampler loops of >= FD_MEGA_RTC_MIN_STEPS diffusion steps; a failure (no hiprtc,
    // compile error) falls through to the library's run-time-shape instantiations below -- the same HIP path, slower.
    char rtc_note[160] = "";
    if (jit != FD_MEGA_RTC_OFF) {
        fd_mega_rtc_key key{};
        key.ks1 = ks1; key.dt = dt; key.kso = kso; key.mt = mt;
        key.T = P.T; key.D = P.D; key.C = P.C; key.H = P.H; key.S = P.S; key.NPG = P.NPG; key.rot = P.rot; key.L = P.L; key.F = P.F;
        const int ntile = P.S * P.KT;
        key.ffn32 = (FD_MEGA_FFN32 && dt == 2 * ks1 - 1 && P.off_ffn32 != 0 &&
                     ((mt == 4 && ntile >= 12 && (ntile & 1) == 0) || (mt == 3 && ntile == 12))) ? 1 : 0;
        const bool wanted = jit >= FD_MEGA_RTC_ALWAYS || (P.mode == FD_MEGA_SAMPLE && P.nsteps >= FD_MEGA_RTC_MIN_STEPS);
        if (wanted) {
            void* fn = nullptr;
            std::string why;
            if (fd_mega_rtc_get(ctx, key, &fn, &why, /*compile=*/describe == nullptr)) {      // (a description compiles nothing)
                if (describe) {
                    snprintf(describe, 192, "k_mega<%d,%d,%d,%d,ShapeStatic<%d,%d,%d,%d,%d,%d,%d,%d,%d> (hiprtc)> S=%d NPG=%d rot=%d grid=%d lds=%zu%s", ks1, dt,
                             kso, mt, P.T, P.D, P.C, P.H, P.S, P.NPG, P.rot, P.L, P.F, P.S, P.NPG, P.rot, grid, lds, key.ffn32 ? " ffn32" : "");
                    return FD_OK;
                }
                return fd_mega_rtc_launch(ctx, fn, P, grid, lds, s);
            }
            snprintf(rtc_note, sizeof rtc_note, " [no run-time specialisation: %.100s]", why.c_str());
        } else if (describe) {
            snprintf(rtc_note, sizeof rtc_note, " [sampler runs of >= %d steps: hiprtc ShapeStatic]", FD_MEGA_RTC_MIN_STEPS);
        }
    }
    (void)rtc_note;
#undef FD_MEGA_NOTE
#define FD_MEGA_NOTE rtc_note
#ifdef FD_MEGA_EXTRA_SHAPE
    // Ahead-of-time specialisation for one more workload (scripts/specialize.sh builds a library variant with
    //   -DFD_MEGA_EXTRA_SHAPE=T,D,C,H,S,NPG,rot,L,F -DFD_MEGA_EXTRA_TILES=KS1,DT,KSO,MT
    // and FDIFF_LIB selects it): a static instantiation is ~1.6x faster than the run-time-shape kernel.
    {
        using ShapeExtra = ShapeStatic<FD_MEGA_EXTRA_SHAPE>;
        constexpr int xt[4] = {FD_MEGA_EXTRA_TILES};
        if (!getenv("FDIFF_MEGA_GENERIC") && ks1 == xt[0] && dt == xt[1] && kso == xt[2] && mt == xt[3] && P.T == ShapeExtra::T &&
            P.D == ShapeExtra::D && P.C == ShapeExtra::C && P.H == ShapeExtra::H && P.S == ShapeExtra::S &&
            P.NPG == ShapeExtra::NPG && P.rot == ShapeExtra::rot && P.L == ShapeExtra::L && P.F == ShapeExtra::F)
            FD_MEGA_GO(xt[0], xt[1], xt[2], xt[3], ShapeExtra, "ShapeStatic<extra>");
    }
#endif
    // the hydra default transformer (d_model 72, 12 heads, 10 layers, ff 2048) at any other series shape
    using ShapeDefaultModel = ShapeModel<72, 12, 10, 2048>;
    if (!getenv("FDIFF_MEGA_GENERIC") && ks1 == 3 && dt == 5 && kso == 3 && P.D == 72 && P.H == 12 && P.L == 10 && P.F == 2048) {
        switch (mt) {
            case 1: FD_MEGA_GO(3, 5, 3, 1, ShapeDefaultModel, "ShapeModel<72,12,10,2048>");
            case 2: FD_MEGA_GO(3, 5, 3, 2, ShapeDefaultModel, "ShapeModel<72,12,10,2048>");
            case 3: FD_MEGA_GO(3, 5, 3, 3, ShapeDefaultModel, "ShapeModel<72,12,10,2048>");
            case 4: FD_MEGA_GO(3, 5, 3, 4, ShapeDefaultModel, "ShapeModel<72,12,10,2048>");
            default: break;
        }
    }
#define FD_MEGA_CASE(K, T_, O, M_)                                                                 \
    if (ks1 == K && dt == T_ && kso == O && mt == M_) FD_MEGA_GO(K, T_, O, M_, ShapeDyn, "ShapeDyn");
#define FD_MEGA_MT(K, T_, O) FD_MEGA_CASE(K, T_, O, 1) FD_MEGA_CASE(K, T_, O, 2) FD_MEGA_CASE(K, T_, O, 3) FD_MEGA_CASE(K, T_, O, 4)
    FD_MEGA_MT(3, 5, 3)   // d_model 72, 12 heads (hydra default)
    FD_MEGA_MT(3, 5, 2)   // d_model 64, 8 heads: head_dim 8 (exact two-pass softmax, denominators from an all-ones MFMA)
    FD_MEGA_MT(2, 3, 1)   // d_model 32, 4 heads: head_dim 8
    FD_MEGA_MT(2, 4, 3)   // d_model 60, 12 heads (class default)
    FD_MEGA_MT(1, 2, 1)   // d_model 24, 4 heads
    FD_MEGA_MT(1, 1, 1)   // d_model 8, 4 heads
#undef FD_MEGA_MT
#undef FD_MEGA_CASE
    return fd_fail(ctx, FD_ERR_UNSUPPORTED, "persistent kernel not instantiated for ks1=%d dt=%d kso=%d mt=%d", ks1, dt,
                   kso, mt);
}
