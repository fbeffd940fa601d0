// fd_optim.hip -- fused optimiser over the flat fp32 parameter buffer.
// Reference: ScoreModule.configure_optimizers (src/fdiff/models/score_models.py:122-130) =
// torch.optim.AdamW(lr_max, betas (0.9,0.999), eps 1e-8, weight_decay 1e-2); Lightning's
// gradient_clip_val=1.0 (cmd/conf/trainer/default.yaml:4) = clip_grad_norm_ by global L2 norm.
// HBM-bound: 16 B/param read (p, g, m, v) + 12 B/param written (p, m, v).
#include "fd_common.h"

namespace {

// stage 1 of the squared gradient norm: one double partial per block (fixed strand order inside the block); stage 2
// (k_sqnorm_final) adds the blocks in ascending order -- the clipping coefficient, hence the parameter update, is then
// bit-reproducible from run to run (a float atomic per block was not)
__global__ __launch_bounds__(256) void k_sqnorm(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = (double)g[i];
        acc += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void k_sqnorm_final(const double* __restrict__ part, int nblk, float* __restrict__ out) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) v += part[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (threadIdx.x == 0) *out = (float)v;
}

__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g,
                                                float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                float beta1, float beta2, float eps, float wd, float bc1,
                                                float bc2_sqrt, const float* __restrict__ sqnorm, float max_norm,
                                                float grad_scale, int64_t fz0, int64_t fz1, const unsigned* __restrict__ err) {
    // The device copy of the error word of the training kernels' bounded waits (set together with the host-visible word when a
    // hand-over of THIS step timed out): its gradients are invalid, so the update is skipped IN STREAM ORDER -- parameters and
    // moments stay what they were -- and the host check at the next call entry (fd_train_async_check) only has to report it.
    // (Device memory: the host-mapped word itself cost one PCIe read per wave, 0.7 ms per optimizer step.)
    if (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    float coef = grad_scale;
    if (sqnorm) {
        // clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
        const float tn = sqrtf(*sqnorm) * grad_scale;
        coef *= fminf(1.0f, max_norm / (tn + 1e-6f));
    }
    const float step_size = lr / bc1;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (i >= fz0 && i < fz1) continue;   // requires_grad=False range (time_encoder.W)
        const float gi = g[i] * coef;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

inline unsigned grid_for(fd_ctx* ctx, int64_t n) {
    int64_t b = (n + 255) / 256;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int fd_grad_sqnorm(fd_ctx* ctx, const float* grads, int64_t n, float* sqnorm_out, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, grads && sqnorm_out && n > 0, "fd_grad_sqnorm: bad arguments");
    const unsigned nblk = grid_for(ctx, n);
    double* part = reinterpret_cast<double*>(fd_red_scratch(ctx, 2 * (size_t)nblk));
    if (!part) return fd_fail(ctx, FD_ERR_HIP, "fd_grad_sqnorm: reduction scratch allocation failed");
    hipLaunchKernelGGL(k_sqnorm, dim3(nblk), dim3(256), 0, (hipStream_t)stream, grads, n, part);
    hipLaunchKernelGGL(k_sqnorm_final, dim3(1), dim3(64), 0, (hipStream_t)stream, part, (int)nblk, sqnorm_out);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}

extern "C" int fd_adamw_step(fd_ctx* ctx, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                             int64_t n, int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                             const float* sqnorm, float max_norm, float grad_scale, int64_t frozen_begin,
                             int64_t frozen_end, void* stream) {
    if (!ctx) return FD_ERR_ARG;
    FD_REQUIRE(ctx, params && grads && exp_avg && exp_avg_sq && n > 0, "fd_adamw_step: null pointer or n <= 0");
    // (an EARLIER call's timed-out hand-over; one of the step whose gradients these are may not have happened yet when this host
    // check runs -- the kernel below re-reads the error word on the device, in stream order, and skips the update then)
    if (int rc = fd_train_async_check(ctx)) return rc;
    FD_REQUIRE(ctx, step >= 1, "fd_adamw_step: step is 1-based, got %d", step);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(k_adamw, dim3(grid_for(ctx, n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), sqnorm,
                       max_norm, grad_scale, frozen_begin, frozen_end, (const unsigned*)ctx->tr_err_gpu);
    FD_LAUNCH_CHECK(ctx);
    return FD_OK;
}
