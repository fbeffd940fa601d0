// fd_score_bwd.hip -- backward pass of the score network (placeholder).
#include "fd_score.h"

extern "C" int fd_score_backward(fd_score* m, const float* dout, float* grads, int accumulate, void* stream) {
    (void)dout; (void)grads; (void)accumulate; (void)stream;
    if (!m) return FD_ERR_ARG;
    return fd_fail(m->ctx, FD_ERR_UNSUPPORTED, "backward not built yet");
}
