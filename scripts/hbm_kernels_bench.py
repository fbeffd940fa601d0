"""Achieved HBM bandwidth of the bandwidth-bound kernels of the path (algorithmic bytes / time), against ~8 TB/s:
dft / idft (8 B per element), sde_step (12 B per element), AdamW (28 B per parameter)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fourierdiffusion_amd.utils.fourier import dft, idft
from fourierdiffusion_amd.schedulers.sde import VPScheduler


def timed(fn, n=20):
    """Median over n calls of the event-bracketed device time (a mean over host wall time picks up the caching allocator's
    occasional hipMalloc: one 70-90 ms stall in 20 calls, seen at a different row in every run)."""
    for _ in range(3):          # (the first call of a kernel variant loads its code object: several ms)
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:           # back to back (no host sync in between): the events bracket the kernel on the device time line
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = [e0.elapsed_time(e1) * 1e-3 for e0, e1 in ev]
    ts.sort()
    return ts[n // 2]


ALL = [(512, 100, 12), (4096, 256, 28), (512, 1024, 16), (4096, 252, 6), (87554, 187, 1), (65536, 256, 1), (4096, 187, 12),
       (4096, 365, 8), (4096, 143, 12), (4096, 253, 8)]
as_json = "--json" in sys.argv          # (bench.py's `secondary.hbm_kernels`: the BASELINE shapes only, one JSON line)
shapes = [(4096, 256, 28), (512, 1024, 16), (512, 100, 12)] if as_json else ALL
n_calls = 5 if "--quick" in sys.argv else 20
rows = {}
for (B, T, C) in shapes:
    x = torch.randn(B, T, C, device="cuda")
    n = x.numel()
    td = timed(lambda: dft(x), n_calls); ti = timed(lambda: idft(x), n_calls)
    sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True)
    sch.set_noise_scaling(T); sch.set_timesteps(1000)
    s = torch.randn_like(x)
    ts = timed(lambda: sch.step(s, 0.37, x), n_calls)
    rows[f"B{B}_T{T}_C{C}"] = {"MB": n * 4 / 1e6, "dft_us": td * 1e6, "dft_TBps": 8 * n / td / 1e12, "idft_us": ti * 1e6,
                               "idft_TBps": 8 * n / ti / 1e12, "sde_step_us": ts * 1e6, "sde_step_TBps": 12 * n / ts / 1e12}
    if not as_json:
        print(f"(B={B},T={T},C={C}) {n*4/1e6:7.1f} MB: dft {td*1e6:7.1f} us = {8*n/td/1e12:5.2f} TB/s | idft {ti*1e6:7.1f} us = {8*n/ti/1e12:5.2f} TB/s | "
              f"sde_step {ts*1e6:7.1f} us = {12*n/ts/1e12:5.2f} TB/s")
if as_json:
    import json
    print(json.dumps({"what": "algorithmic bytes (8 B/element dft, idft; 12 B/element sde_step) / median event-bracketed time of "
                              f"{n_calls} back-to-back calls; HBM peak ~8 TB/s", "shapes": rows}))
