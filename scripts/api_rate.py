"""PCIe-inclusive rate of the Python boundary: DiffusionSampler.sample() returns a CPU tensor like the
reference (sampler.py:45-122), i.e. prior + N reverse steps on the GPU, then one device-to-host copy; idft of the result as
cmd/sample.py does.  usage: python scripts/api_rate.py [B N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    from fourierdiffusion_amd.schedulers.sde import VPScheduler
    from fourierdiffusion_amd.utils.fourier import idft
    torch.manual_seed(0)
    sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True)
    sch.set_noise_scaling(100)                                # (set lazily by training in the reference, sde.py:192)
    m = ScoreModule(n_channels=12, max_len=100, noise_scheduler=sch, fourier_noise_scaling=True, d_model=72, num_layers=10,
                    n_head=12).to("cuda")
    m.precision = "bf16"
    m.eval()
    s = DiffusionSampler(score_model=m, sample_batch_size=B)
    s.sample(num_samples=B, num_diffusion_steps=8)
    torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.perf_counter()
        X = s.sample(num_samples=B, num_diffusion_steps=N)
        t1 = time.perf_counter()
        Xt = idft(X)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    assert X.device.type == "cpu" and X.shape == (B, 100, 12)
    print(f"DiffusionSampler.sample({B}, {N}) -> CPU tensor: {(t1 - t0) * 1e3:.1f} ms = {B / (t1 - t0):.1f} series/s; "
          f"+ idft of the CPU result (H2D, kernel, D2H): {(t2 - t1) * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
