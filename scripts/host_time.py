#!/usr/bin/env python
"""Host-side cost of one optimizer step: how long the Python loop needs to ENQUEUE `steps` steps (no synchronisation inside
the loop) against how long the GPU needs to run them.  host < gpu means the launches run ahead of the kernels (the step is
GPU-bound); host ~ gpu means the step is launch-bound.  Run on the GPU box:  python scripts/host_time.py [nasdaq|ecg]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fourierdiffusion_amd.models.score_models import ScoreModule
from fourierdiffusion_amd.optim import FusedAdamW
from fourierdiffusion_amd.schedulers.sde import VPScheduler
from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch

name = sys.argv[1] if len(sys.argv) > 1 else "nasdaq"
T, CH, B = {"nasdaq": (252, 6, 64), "ecg": (100, 12, 64)}[name]
dev = torch.device("cuda:0")
sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True)
sch.set_noise_scaling(T)
model = ScoreModule(n_channels=CH, max_len=T, noise_scheduler=sch, fourier_noise_scaling=True, d_model=72, num_layers=10, n_head=12).to(dev)
model.train_precision = "bf16"
model.train()
if os.environ.get("FDIFF_DROPOUT"):          # (0 = no dropout decisions at all: what the mask kernels on the side stream cost)
    model.dropout = float(os.environ["FDIFF_DROPOUT"])
opt = FusedAdamW(model, lr=1e-3, max_grad_norm=1.0)
X = torch.randn(B, T, CH).to(dev)

def one(i):
    model.zero_grad()
    loss = model.training_step(DiffusableBatch(X=X), i)
    opt.step()
    return loss

for i in range(5):
    one(i)
torch.cuda.synchronize()
steps = 100
stamps = []
t0 = time.perf_counter()
for i in range(steps):
    one(i)
    stamps.append(time.perf_counter())
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
d = [1e3 * (b - a) for a, b in zip(stamps[:-1], stamps[1:])]
d.sort()
print(f"{name}: host enqueue {1e3 * t_host / steps:.3f} ms/step (median {d[len(d) // 2]:.3f}, min {d[0]:.3f}, max {d[-1]:.3f}); "
      f"with GPU drain {1e3 * t_all / steps:.3f} ms/step")
