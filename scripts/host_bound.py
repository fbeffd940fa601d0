import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fourierdiffusion_amd.models.score_models import ScoreModule
from fourierdiffusion_amd.schedulers.sde import VPScheduler
from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
from fourierdiffusion_amd.optim import FusedAdamW
T, Cn, B = 100, 12, 64
dev = torch.device("cuda", 0)
sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True); sch.set_noise_scaling(T)
m = ScoreModule(n_channels=Cn, max_len=T, noise_scheduler=sch, fourier_noise_scaling=True, d_model=72, num_layers=10, n_head=12).to(dev)
m.train(); opt = FusedAdamW(m, lr=1e-3); X = torch.randn(B, T, Cn, device=dev)
def step():
    m.zero_grad(); loss = m.training_step(DiffusableBatch(X=X), 0); opt.step(); return loss
for _ in range(5): step()
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
for _ in range(50): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue per step {1e3*(t1-t0)/50:.3f} ms; total per step {1e3*(t2-t0)/50:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
