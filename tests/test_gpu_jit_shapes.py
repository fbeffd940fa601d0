"""GPU parity of the RUN-TIME SPECIALISED persistent kernel (csrc/fd_mega_rtc.hip): ShapeStatic instantiations of k_mega compiled by
hiprtc for the series shapes the reference actually ships -- the library carries static instantiations for the three BASELINE
stand-in shapes only, everything else ran the 25-35 % slower run-time-shape kernel (VERDICT r4 "missing" 2):

  ecg       (187, 1)   /root/reference/src/fdiff/dataloaders/datamodules.py:194-201
  nasdaq    (252, 5)   :404-410
  mimiciii  (24, 40)   /root/reference/cmd/conf/datamodule/mimiciii.yaml:7 (n_feats 40; eight series per workgroup)
  nasa      (251, 4) charge / (134, 5) discharge   datamodules.py:471-476
  ecg stand-in (100, 12) at the reference's default sample_batch_size = 200 (cmd/conf/sampler/default.yaml: one series per workgroup)

For each, with FDIFF_MEGA_JIT=1 and the plan asserted to name the hiprtc instantiation: the forward against the float64 oracle on
every series of the first / a middle / the last workgroup (<= 1e-2 of scale, rms <= 7e-3: test_gpu_bench_instantiation.py's bounds),
a 20-step injected-noise trajectory against the oracle's loop (<= 1e-2, rms <= 5e-3), and the on-device Philox stream against the
per-step launches.  Plus: the AUTO policy (a single forward never compiles, a >= 100-step sampler run does, and the code object
lands in FDIFF_CACHE_DIR), and force-compiling the ecg BASELINE shape reproduces the ahead-of-time instantiation to rounding noise.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import fdiff_oracle as O
from oracle import weights as W

from .gpu_util import DEV, dev, host, make_model, oracle_sde, report_err
from .test_gpu_bench_instantiation import _cu_count, _env, _fwd, _rows, _run_sampler

pytestmark = pytest.mark.gpu

# name: (T, C, series per workgroup wanted, B as a multiple of the CU count (B = mult * #CU - 1: a ragged last workgroup))
SHAPES = {
    "ecg187": (187, 1, 1, 2),
    "nasdaq5": (252, 5, 1, 2),
    "mimic24_s8": (24, 40, 8, 8),
    "nasa_charge": (251, 4, 1, 2),
    "nasa_discharge": (134, 5, 1, 2),
    "ecg_b200": (100, 12, 1, None),          # B = 200: the reference's sample_batch_size
    "mimic24_s2": (24, 40, 2, 2),
}


def _case(name):
    T, C, S_want, mult = SHAPES[name]
    cfg = dict(T=T, C=C, D=72, L=10, H=12)
    B = 200 if mult is None else mult * _cu_count() - 1
    return cfg, S_want, B


@pytest.fixture(autouse=True)
def _cache_dir(tmp_path_factory, monkeypatch):
    # one cache for the whole module run: the trajectory tests reuse what the forward tests compiled
    d = os.environ.get("FDIFF_TEST_JIT_CACHE") or str(tmp_path_factory.getbasetemp() / "fdiff_jit_cache")
    monkeypatch.setenv("FDIFF_CACHE_DIR", d)
    yield d


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_forward_specialised_kernel_vs_oracle(name, monkeypatch):
    cfg, S_want, B = _case(name)
    monkeypatch.setenv("FDIFF_MEGA_JIT", "1")
    m, _, sd = make_model(cfg, precision="bf16")
    desc, S = m.plan(B)
    assert S == S_want and "(hiprtc)" in desc and f"ShapeStatic<{cfg['T']},72,{cfg['C']},12,{S}," in desc, desc
    X = W.randn(f"jit_x_{name}", (B, cfg["T"], cfg["C"]), 2)
    t = W.uniform(f"jit_t_{name}", (B,), 2, 1e-5, 1.0)
    out = _fwd(m, X, t)
    rows = _rows(B, S)[:12]
    ref = O.score_forward(sd, X[rows], t[rows], cfg["H"])
    err, rms = report_err(f"forward bf16 hiprtc {desc.split(' S=')[0]} B={B}", out[rows], ref)
    assert err <= 1e-2 and rms <= 7e-3, (err, rms)
    for i, r in enumerate(rows):
        e = np.abs(out[r] - ref[i]).max() / np.abs(ref[i]).max()
        assert e <= 1.2e-2, (r, e)
    assert np.isfinite(out).all()
    # and against the library's run-time-shape instantiation of the same launch (same arithmetic up to the FFN's summation form)
    monkeypatch.setenv("FDIFF_MEGA_JIT", "0")
    desc0, _ = m.plan(B)
    assert "(hiprtc)" not in desc0, desc0
    gen = _fwd(m, X, t)
    d = np.abs(gen - out).max() / np.abs(out).max()
    print(f"[parity] {name}: specialised vs run-time-shape kernel, max diff / scale = {d:.3e}")
    assert d <= 6e-3, d


@pytest.mark.parametrize("name", ["ecg187", "nasdaq5", "mimic24_s8", "nasa_discharge"])
def test_trajectory_specialised_kernel_vs_oracle(name, monkeypatch):
    cfg, S_want, B = _case(name)
    N = 20
    kind, p = "vp", (0.1, 20.0)
    monkeypatch.setenv("FDIFF_MEGA_JIT", "1")
    m, sch, sd = make_model(cfg, kind=kind, p=p, precision="bf16")
    desc, S = m.plan(B)
    assert S == S_want and "(hiprtc)" in desc, desc
    g = torch.Generator(device="cpu").manual_seed(13)
    shape = (B, cfg["T"], cfg["C"])
    zp = torch.randn(shape, generator=g)
    zs = torch.randn((N,) + shape, generator=g)
    zp_d, zs_d = zp.to(DEV), zs.to(DEV)
    got = _run_sampler(m, B, N, zp_d, zs_d, stepwise=False)
    rows = _rows(B, S)[:6]
    ref, _ = O.sample_trajectory(sd, oracle_sde(kind, p, True, cfg["T"]), zp.numpy()[rows].astype(np.float64),
                                 [z[rows].astype(np.float64) for z in zs.numpy()], cfg["H"])
    err, rms = report_err(f"20-step trajectory bf16 hiprtc {name} {desc.split(' S=')[0]} B={B}", got[rows], ref)
    assert err <= 1e-2 and rms <= 5e-3, (err, rms)
    assert np.isfinite(got).all()
    # on-device noise: the specialised kernel's lane -> Philox counter map (C = 1 / 5: lanes straddling counters) against fd_sde_step
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    outs = []
    for stepwise in (False, True):
        smp = DiffusionSampler(score_model=m, sample_batch_size=B)
        with _env(FDIFF_SAMPLER_STEPWISE="1" if stepwise else None):
            torch.manual_seed(321)
            outs.append(smp.sample(num_samples=B, num_diffusion_steps=6).numpy())
    d = np.abs(outs[0] - outs[1]).max() / np.abs(outs[1]).max()
    print(f"[parity] {name} hiprtc Philox: persistent vs per-step, max diff / scale = {d:.3e}")
    assert np.isfinite(outs[0]).all() and d <= 2e-3, d


def test_auto_policy_compiles_for_long_sampler_runs_only(_cache_dir, monkeypatch, tmp_path):
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    monkeypatch.setenv("FDIFF_CACHE_DIR", str(tmp_path))
    monkeypatch.delenv("FDIFF_MEGA_JIT", raising=False)
    cfg, B = dict(T=44, C=3, D=72, L=2, H=12), 5
    m, _, _ = make_model(cfg, precision="bf16")
    desc, _ = m.plan(B)
    assert "(hiprtc)" not in desc and "sampler runs of >= 100 steps" in desc, desc
    _fwd(m, W.randn("jit_auto_x", (B, cfg["T"], cfg["C"]), 2), W.uniform("jit_auto_t", (B,), 2, 1e-5, 1.0))
    smp = DiffusionSampler(score_model=m, sample_batch_size=B)
    torch.manual_seed(5)
    a20 = smp.sample(num_samples=B, num_diffusion_steps=20)
    assert glob.glob(str(tmp_path / "*.fdco")) == [], "a forward / a 20-step run compiled a specialised kernel"
    torch.manual_seed(5)
    a100 = smp.sample(num_samples=B, num_diffusion_steps=100)
    files = glob.glob(str(tmp_path / "*.fdco"))
    assert len(files) == 1 and os.path.getsize(files[0]) > 10000, files
    assert torch.isfinite(a20).all() and torch.isfinite(a100).all()
    # switched off, the same run gives the run-time-shape kernel's samples: close, and no new code object
    monkeypatch.setenv("FDIFF_MEGA_JIT", "0")
    torch.manual_seed(5)
    b100 = smp.sample(num_samples=B, num_diffusion_steps=100)
    assert len(glob.glob(str(tmp_path / "*.fdco"))) == 1
    d = (a100 - b100).abs().max().item() / b100.abs().max().item()
    print(f"[parity] 100-step samples, specialised vs run-time-shape kernel: max diff / scale = {d:.3e}")
    assert d <= 2e-2, d


def test_force_compiled_ecg_equals_the_ahead_of_time_instantiation(monkeypatch):
    """Same kernel text through hipcc (library; ROCm 7.2's clang) and hiprtc (FDIFF_MEGA_JIT=force; in a torch process the hiprtc
    image torch ships, ROCm 7.0's): same arithmetic, but the two compilers contract multiply-adds and select transcendental
    sequences differently (measured: 3e-3 of scale after 10 steps), so the samples agree to bf16 rounding noise, not bit for bit --
    a wrong instantiation (another shape's tile plan, a different Philox map) gives O(1)."""
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    cfg = dict(T=100, C=12, D=72, L=10, H=12)
    B = 2 * _cu_count()
    outs = []
    for mode in (None, "force"):
        if mode:
            monkeypatch.setenv("FDIFF_MEGA_JIT", mode)
        m, _, _ = make_model(cfg, precision="bf16")
        desc, S = m.plan(B)
        assert S == 2 and ("(hiprtc)" in desc) == bool(mode), desc
        smp = DiffusionSampler(score_model=m, sample_batch_size=B)
        torch.manual_seed(77)
        outs.append(smp.sample(num_samples=B, num_diffusion_steps=10))
    d = (outs[0] - outs[1]).abs().max().item() / outs[0].abs().max().item()
    print(f"[parity] ecg S=2: hiprtc-compiled vs ahead-of-time instantiation after 10 steps, max diff / scale = {d:.3e}")
    assert d <= 6e-3, d
