import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fourierdiffusion_amd.schedulers.sde import VPScheduler
for (B,T,C) in [(64,1024,16),(512,100,12),(512,1024,16)]:
    x = torch.randn(B,T,C,device="cuda"); s = torch.randn_like(x); z = torch.randn_like(x)
    sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True); sch.set_noise_scaling(T); sch.set_timesteps(1000)
    for i in range(10): sch.step(s, 0.37, x)
    torch.cuda.synchronize()
    for i in range(10): sch.step(s, 0.37, x, noise=z)
    torch.cuda.synchronize()
