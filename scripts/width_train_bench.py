"""Optimizer-step time at a width outside the persistent kernel's family, fused bf16 MFMA training kernels against the exact-f32 path.
usage: python scripts/width_train_bench.py D H [T C B]   (default 64 8 100 12 64; 10 layers, dropout 0.1, AdamW + clip)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    D, H = int(sys.argv[1]), int(sys.argv[2])
    T, C, B = (int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else (100, 12, 64)
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.optim import FusedAdamW
    from fourierdiffusion_amd.schedulers.sde import VPScheduler
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    dev = torch.device("cuda", 0)
    for prec in ("fp32", "bf16"):
        torch.manual_seed(0)
        sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True)
        sch.set_noise_scaling(T)
        m = ScoreModule(n_channels=C, max_len=T, noise_scheduler=sch, fourier_noise_scaling=True, d_model=D, num_layers=10, n_head=H).to(dev)
        m.train_precision = prec
        m.train()
        opt = FusedAdamW(m, lr=1e-3, max_grad_norm=1.0)
        X = torch.randn(B, T, C, device=dev)

        def step(i):
            m.zero_grad()
            loss = m.training_step(DiffusableBatch(X=X), i)
            opt.step()
            return loss
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        for i in range(n):
            loss = step(i)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        print(f"d_model {D} / {H} heads, T={T} C={C} B={B}, train_precision={prec} (effective {m.train_mode_effective}): {ms:.3f} ms per optimizer step, "
              f"loss {float(loss):.4f}")


if __name__ == "__main__":
    main()
