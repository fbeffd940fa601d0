// fd_score_bf16.hip -- bf16 MFMA inference path (placeholder until the fused kernels land).
#include "fd_score.h"

struct fd_bf16_images {
    int dummy;
};

int fd_bf16_create(fd_score* m) {
    m->bf16 = nullptr;
    return FD_OK;
}
void fd_bf16_destroy(fd_score* m) { (void)m; }
int fd_bf16_prepare(fd_score* m, hipStream_t s) {
    (void)m;
    (void)s;
    return FD_OK;
}
int fd_score_forward_bf16(fd_score* m, const float* x, const float* t, float* out, int B, hipStream_t s) {
    (void)x; (void)t; (void)out; (void)B; (void)s;
    return fd_fail(m->ctx, FD_ERR_UNSUPPORTED, "bf16 path not built yet");
}
int fd_sampler_run_bf16(fd_score* m, const fd_sde_params*, const float*, const float*, int, float, float*,
                        const float*, uint64_t, uint64_t, int, hipStream_t) {
    return fd_fail(m->ctx, FD_ERR_UNSUPPORTED, "bf16 path not built yet");
}
