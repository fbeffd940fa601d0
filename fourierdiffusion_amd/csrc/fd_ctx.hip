// fd_ctx.hip -- context, error string, workspace arena.
#include <algorithm>

#include "fd_common.h"

extern "C" int fd_version(void) { return 100; }

extern "C" int fd_ctx_create(int device, fd_ctx** out) {
    if (!out) return FD_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return FD_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return FD_ERR_HIP;
    fd_ctx* c = new fd_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    *out = c;
    return FD_OK;
}

extern "C" int fd_comm_destroy(fd_ctx* ctx);

extern "C" int fd_ctx_destroy(fd_ctx* ctx) {
    if (!ctx) return FD_ERR_ARG;
    fd_comm_destroy(ctx);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->gemm_scratch) (void)hipFree(ctx->gemm_scratch);
    if (ctx->red_scratch) (void)hipFree(ctx->red_scratch);
    if (ctx->tr_ypart) (void)hipFree(ctx->tr_ypart);
    if (ctx->tr_yflag) (void)hipFree(ctx->tr_yflag);
    if (ctx->tr_err_host) (void)hipHostFree(ctx->tr_err_host);
    if (ctx->tr_err_gpu) (void)hipFree(ctx->tr_err_gpu);
    if (ctx->trp_flags) (void)hipFree(ctx->trp_flags);
    if (ctx->prof_clk) (void)hipFree(ctx->prof_clk);
    for (hipEvent_t e : ctx->side_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->tr_readers_event)
        if (e) (void)hipEventDestroy(e);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->mask_stream) (void)hipStreamDestroy(ctx->mask_stream);
    if (ctx->side_stream2) (void)hipStreamDestroy(ctx->side_stream2);
    for (auto& e : ctx->fft_tw) (void)hipFree(e.second);
    delete ctx;
    return FD_OK;
}

int fd_train_async_check(fd_ctx* ctx) {
    if (!ctx || !ctx->tr_err_host) return FD_OK;
    const unsigned e = __atomic_load_n(ctx->tr_err_host, __ATOMIC_RELAXED);
    if (!e) return FD_OK;
    __atomic_store_n(ctx->tr_err_host, 0u, __ATOMIC_RELAXED);
    if (ctx->tr_err_gpu) (void)hipMemsetAsync(ctx->tr_err_gpu, 0, sizeof(unsigned), nullptr);      // (reported: the optimizer may update again)
    if (e & 0x40000000u) ctx->trp_disabled = true;      // (whatever kept a cluster from becoming resident may still be there)
    if (e & 0x40000000u)
        return fd_fail(ctx, FD_ERR_STATE,
                       "an earlier training step's persistent forward (k_tr_fwd_layers) gave up waiting in layer %u, series %u, for %s %u: a "
                       "workgroup of the series' cluster never published its rows (not scheduled -- CU-masked queue / partitioned device / a "
                       "co-tenant kernel holding the CUs -- or faulted).  That step's gradients are invalid (the optimizer skipped it).  This context "
                       "uses the per-layer kernels from now on (FDIFF_TR_PERSIST=0 selects them from the start); FDIFF_TR_TIMEOUT_MS (default "
                       "2000) sets the bound.",
                       (e >> 20) & 0x3ffu, (e >> 4) & 0xffffu, (e & 0x80000u) ? "the dropout decisions, lane" : "token tile", e & 15u);
    const unsigned id = e & 0x3fffffffu;
    return fd_fail(ctx, FD_ERR_STATE,
                   "an earlier training step's F-split hand-over timed out at token block %u, tile %u: the finisher workgroup never saw "
                   "its producer's partial sums (producer not scheduled -- CU-masked queue / partitioned device / a co-tenant kernel "
                   "holding the CUs -- or faulted).  That step's gradients are invalid.  FDIFF_TR_FSPLIT=0 disables the split, "
                   "FDIFF_TR_FSPLIT_TIMEOUT_MS (default 2000) sets the bound.", id >> 2, id & 3u);
}

extern "C" int fd_prof_shader_clock_mhz(fd_ctx* ctx, double* mhz) {
    if (!ctx || !mhz) return FD_ERR_ARG;
    *mhz = ctx->prof_clock_mhz;
    return FD_OK;
}

extern "C" int fd_ctx_check(fd_ctx* ctx) {
    if (!ctx) return FD_ERR_ARG;
    return fd_train_async_check(ctx);
}

extern "C" int fd_ctx_rearm(fd_ctx* ctx) {
    if (!ctx) return FD_ERR_ARG;
    ctx->trp_disabled = false;
    return FD_OK;
}

extern "C" const char* fd_last_error(fd_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" size_t fd_ctx_workspace_bytes(fd_ctx* ctx) { return ctx ? ctx->ws_bytes : 0; }

int fd_ws_reserve(fd_ctx* ctx, size_t bytes) {
    ++ctx->ws_gen;                      // whoever reserves is about to use the arena
    if (bytes <= ctx->ws_bytes) return FD_OK;
    FD_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->ws) {
        FD_HIP(ctx, hipDeviceSynchronize());
        FD_HIP(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    size_t want = bytes + (bytes >> 3);   // 12.5% slack so near-equal sizes do not thrash
    FD_HIP(ctx, hipMalloc(&ctx->ws, want));
    ctx->ws_bytes = want;
    return FD_OK;
}

float* fd_red_scratch(fd_ctx* ctx, size_t n_floats) {
    if (ctx->red_scratch_floats < n_floats) {
        if (ctx->red_scratch) {
            (void)hipDeviceSynchronize();
            (void)hipFree(ctx->red_scratch);
            ctx->red_scratch = nullptr;
            ctx->red_scratch_floats = 0;
        }
        const size_t want = std::max(n_floats, (size_t)1 << 20);
        if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void**)&ctx->red_scratch, want * sizeof(float)) != hipSuccess) {
            ctx->red_scratch = nullptr;
            return nullptr;
        }
        ctx->red_scratch_floats = want;
    }
    return ctx->red_scratch;
}

// 64 MiB of split-K scratch for the exact-f32 GEMMs (skinny outputs with long reductions would otherwise run on ~100
// workgroups); nullptr result = allocation failed, the GEMMs then simply do not split.
float* fd_gemm_scratch(fd_ctx* ctx, size_t* n_floats) {
    if (!ctx->gemm_scratch) {
        const size_t n = (size_t)16 << 20;
        if (hipSetDevice(ctx->device) == hipSuccess && hipMalloc((void**)&ctx->gemm_scratch, n * sizeof(float)) == hipSuccess)
            ctx->gemm_scratch_floats = n;
        else
            ctx->gemm_scratch = nullptr;
    }
    *n_floats = ctx->gemm_scratch_floats;
    return ctx->gemm_scratch;
}

extern "C" int fd_prof_begin(fd_ctx* ctx) {
    if (!ctx) return FD_ERR_ARG;
    for (auto& e : ctx->prof_events) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    ctx->prof_events.clear();
    ctx->prof_kernels.clear();
    ctx->prof_stride = 1;
    ctx->prof_on = true;
    return FD_OK;
}

extern "C" int fd_prof_stride(fd_ctx* ctx, int every) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, every >= 1, "fd_prof_stride: every=%d", every);
    ctx->prof_stride = every;
    return FD_OK;
}

extern "C" int fd_prof_end(fd_ctx* ctx, char* name_out, double* avg_us, int* launches, double* flops_per_launch) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, name_out && avg_us && launches && flops_per_launch, "fd_prof_end: null output");
    ctx->prof_on = false;
    std::vector<double> total_ms(ctx->prof_kernels.size(), 0.0);
    std::vector<int> n(ctx->prof_kernels.size(), 0);
    for (auto& e : ctx->prof_events) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            total_ms[e.kernel] += ms;
            ++n[e.kernel];
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    ctx->prof_events.clear();
    ctx->prof_clock_mhz = 0.0;
    if (ctx->prof_clk) {        // (the events above are synchronised: the launch that wrote the stamps has completed)
        unsigned long long c[4] = {0, 0, 0, 0};
        if (hipMemcpy(c, ctx->prof_clk, sizeof c, hipMemcpyDeviceToHost) == hipSuccess && c[3] > c[1] && c[2] > c[0])
            ctx->prof_clock_mhz = (double)(c[2] - c[0]) / ((double)(c[3] - c[1]) / 100.0);      // wall clock: 100 MHz
    }
    if (getenv("FDIFF_PROF_VERBOSE"))     // every bracketed kernel of the window, not only the dominant one
        for (size_t k = 0; k < total_ms.size(); ++k)
            if (n[k] > 0)
                fprintf(stderr, "[fdiff prof] %-28.28s %4d brackets, avg %8.1f us, %8.2f TFLOP/s\n", ctx->prof_kernels[k].name.c_str(), n[k],
                        1e3 * total_ms[k] / n[k], ctx->prof_kernels[k].flops / (1e-3 * total_ms[k] / n[k]) / 1e12);
    int best = -1;                        // the bracketed kernel with the largest total time in the window
    for (size_t k = 0; k < total_ms.size(); ++k)
        if (n[k] > 0 && (best < 0 || total_ms[k] > total_ms[best])) best = (int)k;
    if (best < 0) {
        name_out[0] = 0; *launches = 0; *avg_us = 0.0; *flops_per_launch = 0.0;
        return FD_OK;
    }
    snprintf(name_out, 128, "%s", ctx->prof_kernels[best].name.c_str());
    *launches = n[best];
    *avg_us = 1e3 * total_ms[best] / n[best];
    *flops_per_launch = ctx->prof_kernels[best].flops;
    return FD_OK;
}
