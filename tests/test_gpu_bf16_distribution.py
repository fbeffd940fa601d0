"""the bf16 MFMA sampler (the product default and the benched path) must produce the reference's SAMPLES, not
just one good forward.  Full 1000-step reverse diffusion at the BASELINE configs[1] workload (T=100, C=12, default
transformer, VP-SDE), 2048 series per arithmetic mode with the SAME Philox keys (torch.manual_seed), i.e. the exact-f32
engine path (pinned to the reference's trajectories by test_gpu_sampler.py) against the bf16 persistent kernel:

  * per-(t, c) mean: |mean_bf16 - mean_f32| <= 0.02 * std_f32(t, c)   (measured 3.5e-3; the sampling error of a mean over
                                                                       2048 draws is 0.022 std)
  * per-(t, c) std:  |std_bf16 / std_f32 - 1| <= 0.01                  (measured 6.4e-4; sampling error 0.016)
  * sliced Wasserstein-2 over 256 random directions (the engine's own metric kernels, standardised per direction):
    mean SW2(f32, bf16) <= 0.1 x the sampling-noise floor SW2(f32 first half, f32 second half)
    (measured 1.2e-3 against a floor of 7.8e-2: the arithmetic mode moves the sample set 60x less than redrawing it);
  * pairwise (same noise): max |x_bf16 - x_f32| <= 1e-2 of the sample scale (measured 1.6e-3) -- reported even though an
    untrained network does not contract the reverse SDE (the drift expands by e^5 over the run).
The measured values are printed ([parity] lines) and recorded in DESIGN.md / profiles/r02_parity_errors.txt.
"""
import numpy as np
import pytest
import torch

from .gpu_util import DEV, make_model

pytestmark = pytest.mark.gpu

ECG = dict(T=100, C=12, D=72, L=10, H=12)


def _sample(precision, n, steps, seed):
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    m, _, _ = make_model(ECG, precision=precision)
    smp = DiffusionSampler(score_model=m, sample_batch_size=512)
    torch.manual_seed(seed)
    return smp.sample(num_samples=n, num_diffusion_steps=steps)


def test_bf16_sampler_reproduces_the_f32_sample_distribution():
    from fourierdiffusion_amd.utils.wasserstein import WassersteinDistances
    n, steps = 2048, 1000
    a = _sample("fp32", n, steps, seed=2024)
    b = _sample("bf16", n, steps, seed=2024)
    assert a.shape == b.shape == (n, ECG["T"], ECG["C"]) and torch.isfinite(a).all() and torch.isfinite(b).all()
    ma, mb = a.mean(0), b.mean(0)
    sa, sb = a.std(0), b.std(0)
    dmean = float(((mb - ma).abs() / sa).max())
    dstd = float((sb / sa - 1).abs().max())
    print(f"[parity] 1000-step samples f32 vs bf16 (n={n}): max |dmean|/std = {dmean:.3e}, max |std ratio - 1| = {dstd:.3e}, "
          f"sample scale {float(a.abs().max()):.3g}")
    flat = lambda x: x.reshape(x.shape[0], -1).to(DEV)           # noqa: E731
    K = 256
    sw_modes = WassersteinDistances(flat(a), flat(b), normalisation="standardise", seed=0).sliced_distances(K)
    sw_floor = WassersteinDistances(flat(a[: n // 2]), flat(a[n // 2:]), normalisation="standardise", seed=0).sliced_distances(K)
    # pairwise (coupled) difference, reported only: same noise, different arithmetic
    pair = float((a - b).abs().max() / a.abs().max())
    print(f"[parity] sliced W2 over {K} directions (standardised): f32 vs bf16 mean {sw_modes.mean():.4e} max {sw_modes.max():.4e}; "
          f"f32 half vs half (sampling floor) mean {sw_floor.mean():.4e}; coupled max |a-b| / scale = {pair:.3e}")
    try:
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with open(os.path.join(root, "gpurun_out", "parity_errors.log"), "a") as f:
            f.write(f"[parity] 1000-step f32 vs bf16 n={n}: dmean/std {dmean:.3e}, dstd {dstd:.3e}, SW2 modes {sw_modes.mean():.4e} "
                    f"(max {sw_modes.max():.4e}), SW2 floor {sw_floor.mean():.4e}, coupled {pair:.3e}\n")
    except OSError:
        pass
    assert dmean <= 0.02 and dstd <= 0.01, (dmean, dstd)
    assert sw_modes.mean() <= 0.1 * sw_floor.mean(), (sw_modes.mean(), sw_floor.mean())
    assert pair <= 1e-2, pair
