"""Score-network forward time at a given width (bf16 mode, whatever path fd_score_plan picks): python scripts/width_bench.py D H [T C B L]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fourierdiffusion_amd.models.score_models import ScoreModule
from fourierdiffusion_amd.schedulers.sde import VPScheduler
from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
a = [int(v) for v in sys.argv[1:]]
D, H = a[0], a[1]
T, C, B, L = (a[2:] + [100, 12, 512, 10][len(a) - 2:])[:4]
torch.manual_seed(0)
sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True); sch.set_noise_scaling(T)
m = ScoreModule(n_channels=C, max_len=T, noise_scheduler=sch, d_model=D, num_layers=L, n_head=H).to("cuda")
m.eval()
X = torch.randn(B, T, C, device="cuda"); t = torch.rand(B, device="cuda")
for prec in ("bf16", "fp32"):
    m.precision = prec
    for _ in range(3): m(DiffusableBatch(X=X, timesteps=t))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20 if prec == "bf16" else 5
    for _ in range(n): m(DiffusableBatch(X=X, timesteps=t))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    fl = T * (L * (2 * D * 3 * D + 2 * D * D + 4 * D * 2048 + 4 * T * D) + 4 * C * D) + 2 * D * D
    print(f"D={D} H={H} T={T} C={C} B={B} L={L} asked {prec} ran {m.precision_effective}: {1e3*dt:.3f} ms per forward, {fl*B/dt/1e12:.1f} TFLOP/s  [{m.plan(B)[0][:110]}]")
