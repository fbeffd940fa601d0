// fd_sampler.hip -- the reverse-diffusion loop of DiffusionSampler.sample
// (src/fdiff/sampling/sampler.py:83-104): n_steps x { score = model(x, t_i); x = sde.step(score, t_i, x) }.
// The reference syncs the device every step (`timesteps[0].item()`, sampler.py:37) and draws the prior on
// the CPU (sde.py:85); here the whole loop is enqueued on one stream with per-step coefficients computed on
// the host up front, and the noise comes from the on-device Philox stream unless injected.
#include "fd_philox.h"
#include "fd_score.h"
#include "fd_sde.h"

namespace {
__global__ __launch_bounds__(256) void k_fill(float* __restrict__ p, int n, float v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
// t vectors of ALL steps in one launch: p[i*B + b] = ts[i]  (one k_fill launch per diffusion step was 4.5 us of a 1.5 ms step
// at T = 1024, and one more launch boundary between the SDE step and the next time embedding)
__global__ __launch_bounds__(256) void k_fill_steps(float* __restrict__ p, const float* __restrict__ ts, int B, size_t n) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < n) p[i] = ts[i / B];
}
constexpr size_t kMaxStepTable = (size_t)16 << 20;     // floats: beyond this (n_steps * B) the t vector is refilled every step

// reserves `fwd + own` (+ the step table) and returns the t vector of step 0 and its stride between steps (0: refill per step)
int step_table(fd_ctx* ctx, size_t fwd, size_t own, const float* timesteps, int n_steps, int B, hipStream_t s, float** tvec,
               size_t* stride) {
    const size_t nt = (size_t)n_steps * B;
    const bool table = nt <= kMaxStepTable && !getenv("FDIFF_SAMPLER_FILL_PER_STEP");
    const size_t extra = table ? fd_ws::padded(nt * sizeof(float)) + fd_ws::padded((size_t)n_steps * sizeof(float))
                               : fd_ws::padded((size_t)B * sizeof(float));
    if (int rc = fd_ws_reserve(ctx, fwd + own + extra)) return rc;
    *tvec = (float*)((char*)ctx->ws + fwd + own);
    *stride = table ? (size_t)B : 0;
    if (table) {
        float* ts = (float*)((char*)*tvec + fd_ws::padded(nt * sizeof(float)));
        // pageable source: the runtime stages the copy before returning
        FD_HIP(ctx, hipMemcpyAsync(ts, timesteps, (size_t)n_steps * sizeof(float), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_fill_steps, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, *tvec, ts, B, nt);
    }
    return FD_OK;
}
}  // namespace

int fd_sampler_run_mega(fd_score* m, const fd_sde_params* sde, const float* G, const float* timesteps, int n_steps,
                        float dt, float* x, const float* z_steps, uint64_t seed, uint64_t offset, int B,
                        hipStream_t s);
int fd_sampler_run_layers(fd_score* m, const fd_sde_params* sde, const float* G, const float* timesteps, int n_steps, float dt,
                          float* x, const float* z_steps, uint64_t seed, uint64_t offset, int B, hipStream_t s);

extern "C" int fd_sampler_run(fd_score* m, const fd_sde_params* sde, const float* G, const float* timesteps,
                              int n_steps, float dt, float* x, const float* z_steps, uint64_t seed, uint64_t offset,
                              int B, int mode, void* stream) {
    if (!m) return FD_ERR_ARG;
    fd_ctx* ctx = m->ctx;
    FD_REQUIRE(ctx, sde && G && timesteps && x, "fd_sampler_run: null pointer");
    FD_REQUIRE(ctx, sde->kind == 0 || sde->kind == 1, "fd_sampler_run: unknown SDE kind %d", sde->kind);
    FD_REQUIRE(ctx, n_steps > 0 && B > 0, "fd_sampler_run: n_steps=%d B=%d", n_steps, B);
    FD_REQUIRE(ctx, dt > 0.f, "fd_sampler_run: step size must be > 0 (sde.py:158)");
    if (!m->prepared) return fd_fail(ctx, FD_ERR_STATE, "fd_sampler_run: call fd_score_prepare first");
    hipStream_t s = (hipStream_t)stream;
    FD_REQUIRE(ctx, mode == FD_MODE_F32 || mode == FD_MODE_BF16, "fd_sampler_run: unknown mode %d", mode);
    if (mode == FD_MODE_BF16 && m->backbone == FD_BACKBONE_TRANSFORMER && !getenv("FDIFF_SAMPLER_STEPWISE")) {   // (switch: per-step launches; tests compare the two)
        const int rc = fd_sampler_run_mega(m, sde, G, timesteps, n_steps, dt, x, z_steps, seed, offset, B, s);
        if (rc != FD_ERR_UNSUPPORTED) return rc;     // ran (or failed loudly); else: step-by-step fallback below
    }
    if (mode == FD_MODE_BF16 && m->backbone == FD_BACKBONE_TRANSFORMER && !getenv("FDIFF_SAMPLER_STEPWISE")) {
        // same model family beyond the persistent kernel's length limit: layer launches + one fused launch per step
        const int rc = fd_sampler_run_layers(m, sde, G, timesteps, n_steps, dt, x, z_steps, seed, offset, B, s);
        if (rc != FD_ERR_UNSUPPORTED) return rc;
    }

    const int T = m->d.max_len, C = m->d.n_channels;
    const size_t n = (size_t)B * T * C;
    const size_t fwd = (m->backbone != FD_BACKBONE_TRANSFORMER) ? fd_bb_workspace(m, B, false) : fd_score_f32_workspace(m, B, false);
    const size_t own = fd_ws::padded(n * sizeof(float));
    float* tvec0 = nullptr;
    size_t tstride = 0;
    if (int rc = step_table(ctx, fwd, own, timesteps, n_steps, B, s, &tvec0, &tstride)) return rc;
    float* score = (float*)((char*)ctx->ws + fwd);
    const uint64_t per_step = (uint64_t)((n + 3) / 4);
    for (int i = 0; i < n_steps; ++i) {
        float* tvec = tvec0 + (size_t)i * tstride;
        if (!tstride) hipLaunchKernelGGL(k_fill, dim3((B + 255) / 256), dim3(256), 0, s, tvec, B, timesteps[i]);
        const int rc_f = fd_score_forward_any(m, x, tvec, score, B, mode, s);
        if (rc_f) return rc_f;
        const float* z = z_steps ? z_steps + (size_t)i * n : nullptr;
        if (int rc = fd_sde_step(ctx, sde, G, x, score, z, seed, offset + (uint64_t)i * per_step,
                                 (double)timesteps[i], dt, x, B, T, C, stream))
            return rc;
    }
    return FD_OK;
}

// Predictor-corrector variant (not in the reference; BASELINE.json configs[3] says "PC sampler"): n_corr Langevin corrector
// steps (fd_langevin_step, signal-to-noise ratio snr) before every predictor step.  zc_steps (nullable): injected corrector
// noise (n_steps, n_corr, B, T, C).  Step-by-step launches (the persistent kernel is predictor-only).
extern "C" int fd_sampler_run_pc(fd_score* m, const fd_sde_params* sde, const float* G, const float* timesteps, int n_steps,
                                 float dt, float* x, const float* z_steps, const float* zc_steps, int n_corr, float snr,
                                 uint64_t seed, uint64_t offset, int B, int mode, void* stream) {
    if (!m) return FD_ERR_ARG;
    fd_ctx* ctx = m->ctx;
    FD_REQUIRE(ctx, sde && G && timesteps && x, "fd_sampler_run_pc: null pointer");
    FD_REQUIRE(ctx, sde->kind == 0 || sde->kind == 1, "fd_sampler_run_pc: unknown SDE kind %d", sde->kind);
    FD_REQUIRE(ctx, n_steps > 0 && B > 0 && n_corr >= 0, "fd_sampler_run_pc: n_steps=%d B=%d n_corr=%d", n_steps, B, n_corr);
    FD_REQUIRE(ctx, dt > 0.f && (n_corr == 0 || snr > 0.f), "fd_sampler_run_pc: dt=%f snr=%f", dt, snr);
    FD_REQUIRE(ctx, mode == FD_MODE_F32 || mode == FD_MODE_BF16, "fd_sampler_run_pc: unknown mode %d", mode);
    if (!m->prepared) return fd_fail(ctx, FD_ERR_STATE, "fd_sampler_run_pc: call fd_score_prepare first");
    hipStream_t s = (hipStream_t)stream;
    const int T = m->d.max_len, C = m->d.n_channels;
    const size_t n = (size_t)B * T * C;
    const size_t fwd = (m->backbone != FD_BACKBONE_TRANSFORMER) ? fd_bb_workspace(m, B, false) : fd_score_f32_workspace(m, B, false);
    const size_t own = fd_ws::padded(n * sizeof(float));
    float* tvec0 = nullptr;
    size_t tstride = 0;
    if (int rc = step_table(ctx, fwd, own, timesteps, n_steps, B, s, &tvec0, &tstride)) return rc;
    float* score = (float*)((char*)ctx->ws + fwd);
    const uint64_t per_step = (uint64_t)((n + 3) / 4);
    // Philox counters: predictor noise of step i at offset + i*per_step (as fd_sampler_run); corrector noise behind them
    const uint64_t corr_base = offset + (uint64_t)n_steps * per_step;
    for (int i = 0; i < n_steps; ++i) {
        float* tvec = tvec0 + (size_t)i * tstride;
        if (!tstride) hipLaunchKernelGGL(k_fill, dim3((B + 255) / 256), dim3(256), 0, s, tvec, B, timesteps[i]);
        for (int k = 0; k < n_corr; ++k) {
            if (int rc = fd_score_forward_any(m, x, tvec, score, B, mode, s)) return rc;
            // alpha_t of Song et al.: 1 - beta(t) dt for the VP-SDE, 1 for the VE-SDE
            float alpha = 1.0f;
            if (sde->kind == 0) alpha = 1.0f - (sde->p0 + timesteps[i] * (sde->p1 - sde->p0)) * dt;
            if (alpha <= 0.f) alpha = 1e-6f;
            const float* zc = zc_steps ? zc_steps + ((size_t)i * n_corr + k) * n : nullptr;
            if (int rc = fd_langevin_step(ctx, G, x, score, zc, seed, corr_base + ((uint64_t)i * n_corr + k) * per_step, snr, alpha, x,
                                          B, T, C, stream))
                return rc;
        }
        if (int rc = fd_score_forward_any(m, x, tvec, score, B, mode, s)) return rc;
        const float* z = z_steps ? z_steps + (size_t)i * n : nullptr;
        if (int rc = fd_sde_step(ctx, sde, G, x, score, z, seed, offset + (uint64_t)i * per_step, (double)timesteps[i], dt, x, B, T,
                                 C, stream))
            return rc;
    }
    return FD_OK;
}
