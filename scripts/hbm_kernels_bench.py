"""Achieved HBM bandwidth of the bandwidth-bound kernels of the path (algorithmic bytes / time), against ~8 TB/s:
dft / idft (8 B per element), sde_step (12 B per element).  The kernels are called through the C ABI on preallocated buffers,
N calls back to back between ONE pair of events (the Python wrappers allocate the result and cost ~25 us of host time per call:
bracketing single wrapper calls measured the host, not the kernel, below ~30 us)."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fourierdiffusion_amd import _C
from fourierdiffusion_amd.schedulers.sde import VPScheduler


def timed(fn, n=20):
    """Seconds per call: median over 5 batches of n back-to-back calls, each batch bracketed by one event pair."""
    for _ in range(3):          # (the first call of a kernel variant loads its code object: several ms)
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / n)
    ts.sort()
    return ts[2]


ALL = [(512, 100, 12), (4096, 256, 28), (512, 1024, 16), (4096, 252, 6), (87554, 187, 1), (65536, 256, 1), (4096, 187, 12),
       (4096, 365, 8), (4096, 143, 12), (4096, 253, 8)]
as_json = "--json" in sys.argv          # (bench.py's `secondary.hbm_kernels`: the BASELINE shapes only, one JSON line)
shapes = [(4096, 256, 28), (512, 1024, 16), (512, 100, 12)] if as_json else ALL
n_calls = 5 if "--quick" in sys.argv else 20
rows = {}
lib = _C.lib()
dev = torch.device("cuda", 0)
ctx = _C.ctx(dev)
for (B, T, Cn) in shapes:
    x = torch.randn(B, T, Cn, device=dev)
    y = torch.empty_like(x)
    s = torch.randn_like(x)
    n = x.numel()
    st = torch.cuda.current_stream(dev).cuda_stream
    sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True)
    sch.set_noise_scaling(T); sch.set_timesteps(1000)
    prm = sch._c_params()
    G = sch.G_on(dev)

    def f_dft():
        _C.check(lib.fd_rfft_pack(ctx, x.data_ptr(), y.data_ptr(), B, T, Cn, st), ctx)

    def f_idft():
        _C.check(lib.fd_irfft_unpack(ctx, x.data_ptr(), y.data_ptr(), B, T, Cn, st), ctx)

    def f_step():
        _C.check(lib.fd_sde_step(ctx, C.byref(prm), G.data_ptr(), x.data_ptr(), s.data_ptr(), None, 1234, 0, 0.37,
                                 float(sch.step_size), y.data_ptr(), B, T, Cn, st), ctx)

    td, ti, ts = timed(f_dft, n_calls), timed(f_idft, n_calls), timed(f_step, n_calls)
    rows[f"B{B}_T{T}_C{Cn}"] = {"MB": n * 4 / 1e6, "dft_us": td * 1e6, "dft_TBps": 8 * n / td / 1e12, "idft_us": ti * 1e6,
                                "idft_TBps": 8 * n / ti / 1e12, "sde_step_us": ts * 1e6, "sde_step_TBps": 12 * n / ts / 1e12}
    if not as_json:
        print(f"(B={B},T={T},C={Cn}) {n*4/1e6:7.1f} MB: dft {td*1e6:7.1f} us = {8*n/td/1e12:5.2f} TB/s | idft {ti*1e6:7.1f} us = {8*n/ti/1e12:5.2f} TB/s | "
              f"sde_step {ts*1e6:7.1f} us = {12*n/ts/1e12:5.2f} TB/s")
if as_json:
    print(json.dumps({"what": "algorithmic bytes (8 B/element dft, idft; 12 B/element sde_step) / time per call (C-ABI calls on "
                              f"preallocated buffers, {n_calls} back to back per event pair, median of 5 batches); HBM peak ~8 TB/s",
                      "shapes": rows}))
