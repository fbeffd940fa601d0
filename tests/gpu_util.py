"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch

from oracle import fdiff_oracle as O
from oracle import weights as W

DEV = "cuda"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(DEV)


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def make_model(cfg, kind="vp", p=(0.1, 20.0), scaling=True, seed=1234, precision="fp32"):
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.schedulers.sde import VEScheduler, VPScheduler
    if kind == "vp":
        sch = VPScheduler(beta_min=p[0], beta_max=p[1], fourier_noise_scaling=scaling)
    else:
        sch = VEScheduler(sigma_min=p[0], sigma_max=p[1], fourier_noise_scaling=scaling)
    sch.set_noise_scaling(cfg["T"])
    m = ScoreModule(n_channels=cfg["C"], max_len=cfg["T"], noise_scheduler=sch, fourier_noise_scaling=scaling,
                    d_model=cfg["D"], num_layers=cfg["L"], n_head=cfg["H"])
    sd = W.make_state_dict(cfg["C"], cfg["T"], cfg["D"], cfg["L"], seed=seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.to(DEV)
    m.precision = precision
    m.train_precision = precision        # fp32 models train with the exact-f32 kernels (parity anchor)
    return m, sch, sd


def oracle_sde(kind, p, scaling, T):
    return O.SDEParams(kind, p[0], p[1], O.noise_scaling(T, scaling))


def report_err(tag, got, ref):
    """(max error / scale of ref, relative rms); printed and appended to gpurun_out/parity_errors.log so that the margins
    of the bf16 tolerances are on record (profiles/r02_parity_errors.txt is a copy of one such run)."""
    import os
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    err = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
    rms = float(np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-30))
    line = f"[parity] {tag}: max err / scale = {err:.3e}, relative rms = {rms:.3e}"
    print(line)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_errors.log"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    return err, rms
