"""Sampler-mode parity of the persistent kernel's OTHER static instantiations and of the step-by-step bf16 loop
(VERDICT r2 "what's weak" 1 / "next" 1).

test_gpu_bench_instantiation.py holds the ecg instantiation (S=2, VP).  Here:

  * ShapeStatic<252,72,6,...> (nasdaq, configs[2] shape) under VP(0.1, 20) and ShapeStatic<256,72,28,...> (mimiciii,
    configs[3] shape: seven Philox groups per token) under **VE(0.01, 2)** -- the coefficients of sde.py:129-165 inside
    the in-register Euler-Maruyama step -- run 20 reverse-diffusion steps with injected normals through
    fd_sampler_run(FD_MODE_BF16) at B = #CU (every CU holds a workgroup); rows of the first, a middle and the last
    workgroup are compared with the float64 oracle's loop (sampler.py:83-104), and the whole batch with the per-step
    launches (FDIFF_SAMPLER_STEPWISE=1).  Tolerances as for the ecg instantiation: <= 1e-2 of the trajectory scale
    (max), <= 5e-3 relative rms; persistent vs per-step <= 2e-3 of scale.
  * the on-device Philox stream at C=28 and C=6 (lane -> counter map of the static kernels) against the standalone
    fd_sde_step's stream: persistent loop vs per-step launches without injected noise.
  * the T=1024, C=16 (configs[4]) bf16 sampler -- per-layer kernels + fd_sde_step, the only path T > 256 has -- 8 steps,
    B=2, VP and VE, against the oracle.
  * configs[3]'s real per-GPU shard, 512 series x 2000 VE predictor steps, through size-independent properties:
    finite, bit-reproducible for one Philox key, different keys give different samples, the 2000-step persistent loop
    agrees with 2000 per-step launches on the whole shard, and the iDFT of the samples round-trips.  (A series' noise
    counters depend on the batch size by design -- the stream is the standalone fd_sde_step's, whose offset advances by
    B*T*C/4 per step -- so "prefix of the batch re-run alone" is not a property of the sampler; row independence of the
    network itself is checked in test_gpu_baseline_shapes.py.)
"""
import os

import numpy as np
import pytest
import torch

from oracle import fdiff_oracle as O
from oracle import weights as W

from .gpu_util import DEV, dev, make_model, oracle_sde, report_err

pytestmark = pytest.mark.gpu

NASDAQ = dict(T=252, C=6, D=72, L=10, H=12)
MIMIC = dict(T=256, C=28, D=72, L=10, H=12)
LONG = dict(T=1024, C=16, D=72, L=10, H=12)
STATIC = {
    "nasdaq_vp": (NASDAQ, "vp", (0.1, 20.0), "ShapeStatic<252,72,6,12,1,2,1,10,2048>"),
    "mimic_ve": (MIMIC, "ve", (0.01, 2.0), "ShapeStatic<256,72,28,12,1,2,1,10,2048>"),
    "mimic_vp": (MIMIC, "vp", (0.1, 20.0), "ShapeStatic<256,72,28,12,1,2,1,10,2048>"),
    "nasdaq_ve": (NASDAQ, "ve", (0.01, 2.0), "ShapeStatic<252,72,6,12,1,2,1,10,2048>"),
}


def _cu_count():
    return torch.cuda.get_device_properties(0).multi_processor_count


class _env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _sample(m, B, N, stepwise=False, zp=None, zs=None, seed=None):
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    smp = DiffusionSampler(score_model=m, sample_batch_size=B)
    with _env(FDIFF_SAMPLER_STEPWISE="1" if stepwise else None):
        if seed is not None:
            torch.manual_seed(seed)
        kw = {} if zp is None else dict(prior_noise=[zp], step_noise=[zs])
        return smp.sample(num_samples=B, num_diffusion_steps=N, **kw).numpy()


@pytest.mark.parametrize("case", sorted(STATIC))
def test_trajectory_bf16_static_instantiations_vs_oracle(case):
    cfg, kind, p, static = STATIC[case]
    B, N = _cu_count(), 20
    m, sch, sd = make_model(cfg, kind=kind, p=p, precision="bf16")
    desc, S = m.plan(B)
    assert static in desc and S == 1, desc
    g = torch.Generator(device="cpu").manual_seed(29)
    shape = (B, cfg["T"], cfg["C"])
    zp = torch.randn(shape, generator=g)
    zs = torch.randn((N,) + shape, generator=g)
    zp_d, zs_d = zp.to(DEV), zs.to(DEV)
    got = _sample(m, B, N, zp=zp_d, zs=zs_d).astype(np.float64)
    rows = [0, B // 2, B - 1]
    ref, _ = O.sample_trajectory(sd, oracle_sde(kind, p, True, cfg["T"]), zp.numpy()[rows].astype(np.float64),
                                 [z[rows].astype(np.float64) for z in zs.numpy()], cfg["H"])
    err, rms = report_err(f"20-step trajectory bf16 {case} {desc.split(' S=')[0]} S={S} B={B} rows={rows}", got[rows], ref)
    assert err <= 1e-2 and rms <= 5e-3, (err, rms)
    for i, r in enumerate(rows):
        e = np.abs(got[r] - ref[i]).max() / np.abs(ref).max()
        assert e <= 1e-2, (r, e)
    assert np.isfinite(got).all()
    sw = _sample(m, B, N, stepwise=True, zp=zp_d, zs=zs_d).astype(np.float64)
    e2 = np.abs(sw - got).max() / np.abs(got).max()
    print(f"[parity] {case}: persistent loop vs per-step launches (all {B} series), max diff / scale = {e2:.3e}")
    assert e2 <= 2e-3, e2


@pytest.mark.parametrize("case", ["mimic_ve", "nasdaq_vp"])
def test_philox_stream_static_instantiations_equals_standalone_step(case):
    """C=28: seven Philox counters per token, C=6: counters straddle tokens.  A wrong lane -> counter map gives
    independent normals, i.e. an O(1) difference."""
    cfg, kind, p, static = STATIC[case]
    B = _cu_count()
    outs = []
    for stepwise in (False, True):
        m, _, _ = make_model(cfg, kind=kind, p=p, precision="bf16")
        assert static in m.plan(B)[0]
        outs.append(_sample(m, B, 8, stepwise=stepwise, seed=321))
    scale = np.abs(outs[1]).max()
    d = np.abs(outs[0] - outs[1]).max() / scale
    print(f"[parity] {case} Philox: persistent vs per-step, max diff / scale = {d:.3e}")
    assert np.isfinite(outs[0]).all() and d <= 2e-3, d


@pytest.mark.parametrize("kind,p", [("vp", (0.1, 20.0)), ("ve", (0.01, 2.0))])
def test_trajectory_bf16_long_stepwise_vs_oracle(kind, p):
    """configs[4]: T=1024 > 256 runs the per-layer kernels step by step (k_attention_bf16 + k_ffn_ln + fd_sde_step)."""
    cfg = LONG
    B, N = 2, 8
    m, sch, sd = make_model(cfg, kind=kind, p=p, precision="bf16")
    desc, _ = m.plan(B)
    assert "k_mega" not in desc, desc
    zp = W.randn(f"sp_zp_long_{kind}", (B, cfg["T"], cfg["C"]), 3)
    zs = np.stack([W.randn(f"sp_zs_long_{kind}_{i}", (B, cfg["T"], cfg["C"]), 3) for i in range(N)])
    got = _sample(m, B, N, zp=dev(zp), zs=dev(zs)).astype(np.float64)
    ref, _ = O.sample_trajectory(sd, oracle_sde(kind, p, True, cfg["T"]), zp, list(zs), cfg["H"])
    err, rms = report_err(f"8-step trajectory bf16 long T=1024 {kind} ({desc.split(' S=')[0]}) B={B}", got, ref)
    assert err <= 1e-2 and rms <= 5e-3, (err, rms)


def test_mimic_shard_2000_steps_ve_properties():
    """configs[3] per-GPU shard: 512 series x 2000 predictor steps at (T=256, C=28), VE(0.01, 2), one persistent launch."""
    from fourierdiffusion_amd.utils.fourier import dft, idft
    cfg, kind, p, static = STATIC["mimic_ve"]
    B, N = 512, 2000
    m, _, _ = make_model(cfg, kind=kind, p=p, precision="bf16")
    assert static in m.plan(B)[0]
    a = _sample(m, B, N, seed=7)
    assert a.shape == (B, cfg["T"], cfg["C"]) and np.isfinite(a).all()
    b = _sample(m, B, N, seed=7)
    assert np.array_equal(a, b), "same Philox key must reproduce the samples bit for bit"
    c = _sample(m, B, N, seed=8)
    assert np.abs(a - c).max() > 1e-2 * np.abs(a).max(), "a different key must give different samples"
    # the prior is N(0, sigma_max^2 G^2): after 2000 reverse steps under a random-weight score net the samples must still
    # be O(sigma_max) and differ between series (no broadcast / stuck lanes)
    assert np.abs(a).max() < 1e3 and a.std(axis=0).min() > 0
    # the 2000-step persistent loop against 2000 x (forward launch + fd_sde_step) on the whole shard: same Philox counters
    # (offset advances by B*T*C/4 per step in both), so the two trajectories differ by bf16 rounding noise only -- a wrong
    # counter range late in the loop, a stale time-embedding table row or a mis-indexed timestep gives O(1)
    sw = _sample(m, B, N, stepwise=True, seed=7)
    d2 = np.abs(sw - a).max() / np.abs(a).max()
    rms2 = np.sqrt(((sw - a) ** 2).mean()) / np.sqrt((a ** 2).mean())
    print(f"[parity] mimic shard 512 x 2000 VE: persistent vs per-step launches, max diff / scale = {d2:.3e}, rel rms = {rms2:.3e}")
    assert rms2 <= 5e-3 and d2 <= 5e-2, (d2, rms2)
    # back to the time domain (cmd/sample.py:82) and forth
    xt = idft(torch.from_numpy(a))
    assert xt.shape == a.shape and torch.isfinite(xt).all()
    back = dft(xt.to(DEV)).cpu().numpy()
    assert np.abs(back - a).max() <= 1e-4 * max(1.0, np.abs(a).max())
