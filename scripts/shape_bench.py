"""Time the hot path at the other BASELINE.json shapes (not bench lines: DESIGN.md section 4 quotes them).
usage: python scripts/shape_bench.py [sample|train] NAME B [diffusion_steps]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {"ecg": (100, 12), "nasdaq": (252, 6), "mimic": (256, 28), "long": (1024, 16)}


def flops_fwd(T, Cn, D=72, L=10, F=2048):
    return T * (L * (2 * D * 3 * D + 2 * D * D + 4 * D * F + 4 * T * D) + 4 * Cn * D) + 2 * D * D


def main():
    what, name, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    T, Cn = SHAPES[name]
    from fourierdiffusion_amd import _C, _rng
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.schedulers.sde import VPScheduler
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True)
    sch.set_noise_scaling(T)
    m = ScoreModule(n_channels=Cn, max_len=T, noise_scheduler=sch, fourier_noise_scaling=True, d_model=72, num_layers=10,
                    n_head=12).to(dev)
    m.precision = os.environ.get("FDIFF_PRECISION", "bf16")
    if what == "sample":
        m.eval()
        sch.set_timesteps(N)
        ctx, h = m._engine()
        lib = _C.lib()
        ts = (C.c_float * N)(*sch.timesteps.tolist())
        prm = sch._c_params()
        G = sch.G_on(dev)
        X = torch.randn(B, T, Cn, device=dev)
        mode = _C.FD_MODE_BF16 if m.precision == "bf16" else _C.FD_MODE_F32
        st = torch.cuda.current_stream(dev).cuda_stream

        def run():
            key, off = _rng.stream()
            _C.check(lib.fd_sampler_run(h, C.byref(prm), G.data_ptr(), ts, N, float(sch.step_size), X.data_ptr(), None, key, off,
                                        B, mode, st), ctx)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
        print(f"{name} sample B={B} T={T} C={Cn} {m.precision}: {1e3 * dt:.3f} ms per diffusion step, "
              f"{B * flops_fwd(T, Cn) / dt / 1e12:.1f} TFLOP/s algorithmic, {B / (dt * 1000):.1f} series/s at N=1000")
    else:
        from fourierdiffusion_amd.optim import FusedAdamW
        m.train()
        opt = FusedAdamW(m, lr=1e-3)
        X = torch.randn(B, T, Cn, device=dev)

        def step():
            m.zero_grad()
            loss = m.training_step(DiffusableBatch(X=X), 0)
            opt.step()
            return loss
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name} train B={B} T={T} C={Cn}: {1e3 * dt:.2f} ms per optimizer step (fwd+bwd+AdamW), "
              f"{3 * B * flops_fwd(T, Cn) / dt / 1e12:.2f} TFLOP/s algorithmic, {B / dt:.0f} series/s")


main()
