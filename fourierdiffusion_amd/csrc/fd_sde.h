// fd_sde.h -- per-step SDE coefficients shared by the standalone step kernel and the fused
// sampler kernels.  Host side computes them in double exactly where the reference uses Python
// floats (sde.py:143-147, 212-213, 229).
#pragma once
#include <cmath>

#include "fd_common.h"

struct SdeCoef {
    float a_x;      // coefficient of x inside the drift:  VP 0.5*beta, VE 0
    float g;        // scalar diffusion at t (before G_k):  VP sqrt(beta), VE sigma_min*sqrt(2 ln r)*r^t
    float dt;       // step size
    float sqrt_dt;  // sqrt(step size)  (torch.sqrt(self.step_size), sde.py:162/243)
};

inline SdeCoef fd_sde_coef(const fd_sde_params& p, double t, float dt) {
    SdeCoef c;
    if (p.kind == 0) {
        const double beta = (double)p.p0 + t * ((double)p.p1 - (double)p.p0);   // sde.py:212-213
        c.a_x = (float)(0.5 * beta);
        c.g = (float)std::sqrt(beta);                                             // sde.py:229
    } else {
        const double r = (double)p.p1 / (double)p.p0;
        c.a_x = 0.f;
        c.g = (float)((double)p.p0 * std::sqrt(2.0 * std::log(r)) * std::pow(r, t));   // sde.py:143-147
    }
    c.dt = dt;
    c.sqrt_dt = sqrtf(dt);
    return c;
}

// x' = x - drift*dt + sqrt(dt)*g_k*z with drift = -a_x*x - g_k^2*score  (sde.py:152-163, 232-244)
__device__ __forceinline__ float fd_sde_apply(float x, float s, float z, float Gk, const SdeCoef& cf) {
    const float gk = cf.g * Gk;
    const float drift = -cf.a_x * x - (gk * gk) * s;
    return x - drift * cf.dt + cf.sqrt_dt * (gk * z);
}
