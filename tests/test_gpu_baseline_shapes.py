"""GPU parity at the series shapes of BASELINE.json configs[2..4] (nasdaq T=252 C=6, mimiciii T=256 C=28, long-horizon
T=1024 C=16; default transformer d_model=72, 10 layers, 12 heads), against the float64 oracle at batch sizes it
finishes in seconds, plus size-independent properties at the full per-GPU batch.

Tolerances as in test_gpu_score.py: fp32 mode 5e-6 abs... at T=1024 the fp32 attention sums 1024 terms, 2e-5;
bf16 MFMA mode <= 2e-2 of the output scale (max) and <= 1e-2 rms."""
import os

import numpy as np
import pytest
import torch

from oracle import fdiff_oracle as O
from oracle import weights as W

from .gpu_util import DEV, dev, host, make_model, oracle_sde, report_err

pytestmark = pytest.mark.gpu

SHAPES = {
    "nasdaq": dict(T=252, C=6, D=72, L=10, H=12),
    "mimic": dict(T=256, C=28, D=72, L=10, H=12),
    "long": dict(T=1024, C=16, D=72, L=10, H=12),
}


def run(model, X, t):
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    model.eval()
    return host(model(DiffusableBatch(X=dev(X), y=None, timesteps=dev(t))))


@pytest.mark.parametrize("name,B", [("nasdaq", 2), ("mimic", 2), ("long", 1)])
def test_forward_f32_vs_oracle(name, B):
    cfg = SHAPES[name]
    m, _, sd = make_model(cfg, precision="fp32")
    X = W.randn(f"bs_x_{name}", (B, cfg["T"], cfg["C"]), 2)
    t = W.uniform(f"bs_t_{name}", (B,), 2, 1e-5, 1.0)
    out = run(m, X, t)
    ref = O.score_forward(sd, X, t, cfg["H"])
    np.testing.assert_allclose(out, ref, atol=2e-5 if cfg["T"] > 256 else 5e-6, rtol=0)


@pytest.mark.parametrize("name,B", [("nasdaq", 3), ("mimic", 3), ("long", 2)])
def test_forward_bf16_vs_oracle(name, B):
    cfg = SHAPES[name]
    m, _, sd = make_model(cfg, precision="bf16")
    X = W.randn(f"bs_x_{name}", (B, cfg["T"], cfg["C"]), 2)
    t = W.uniform(f"bs_t_{name}", (B,), 2, 1e-5, 1.0)
    out = run(m, X, t)
    ref = O.score_forward(sd, X, t, cfg["H"])
    err, rms = report_err(f"forward bf16 {name} B={B} ({m.plan(B)[0].split(' S=')[0]})", out, ref)
    assert err <= 2e-2 and rms <= 1e-2, (err, rms)


@pytest.mark.parametrize("name,kind,p", [("nasdaq", "vp", (0.1, 20.0)), ("mimic", "ve", (0.01, 2.0))])
def test_short_trajectory_f32_vs_oracle(name, kind, p):
    """8-step reverse diffusion with injected normals (prior + per step), fp32 engine vs the oracle's sampler loop."""
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    cfg = SHAPES[name]
    B, N = 2, 8
    m, sch, sd = make_model(cfg, kind=kind, p=p, precision="fp32")
    zp = W.randn(f"bs_zp_{name}", (B, cfg["T"], cfg["C"]), 3)
    zs = np.stack([W.randn(f"bs_zs_{name}_{i}", (B, cfg["T"], cfg["C"]), 3) for i in range(N)])
    ref, _ = O.sample_trajectory(sd, oracle_sde(kind, p, True, cfg["T"]), zp, list(zs), cfg["H"])
    smp = DiffusionSampler(score_model=m, sample_batch_size=B)
    out = smp.sample(num_samples=B, num_diffusion_steps=N, prior_noise=[dev(zp)], step_noise=[dev(zs)]).numpy()
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(out - ref).max() <= 1e-4 * scale, (np.abs(out - ref).max(), scale)


ODD_SHAPES = {
    # reference dataset-like and corner shapes: odd T beyond one workgroup, single channel, class-default model
    # (d_model 60 -> head_dim 5), head_dim > 7 (no ones-row slot: exact-f32 attention inside the bf16 path), T = 2048
    "drought": (dict(T=365, C=1, D=72, L=3, H=12), 3),
    "class_default": (dict(T=128, C=5, D=60, L=3, H=12), 4),
    "class_default_long": (dict(T=400, C=7, D=60, L=3, H=12), 3),
    "wide_head": (dict(T=300, C=4, D=24, L=2, H=2), 3),
    "t2048": (dict(T=2048, C=4, D=72, L=1, H=12), 1),
}


@pytest.mark.parametrize("name", sorted(ODD_SHAPES))
def test_forward_bf16_odd_shapes_vs_oracle(name):
    cfg, B = ODD_SHAPES[name]
    m, _, sd = make_model(cfg, precision="bf16")
    X = W.randn(f"os_x_{name}", (B, cfg["T"], cfg["C"]), 2)
    t = W.uniform(f"os_t_{name}", (B,), 2, 1e-5, 1.0)
    out = run(m, X, t)
    ref = O.score_forward(sd, X, t, cfg["H"])
    err, rms = report_err(f"forward bf16 {name} B={B} ({m.plan(B)[0].split(' S=')[0]})", out, ref)
    assert err <= 2e-2 and rms <= 1e-2, (err, rms)


def test_mimic_full_batch_properties_bf16():
    """512 series per GPU at T=256, C=28 (configs[3] per-GPU shard): finite, deterministic, rows independent."""
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    cfg = SHAPES["mimic"]
    m, _, _ = make_model(cfg, precision="bf16")
    m.eval()
    g = torch.Generator(device="cpu").manual_seed(5)
    X = torch.randn(512, cfg["T"], cfg["C"], generator=g).to(DEV)
    t = torch.rand(512, generator=g).to(DEV)
    a = m(DiffusableBatch(X=X, timesteps=t))
    b = m(DiffusableBatch(X=X, timesteps=t))
    assert torch.isfinite(a).all() and torch.equal(a, b)
    idx = torch.tensor([0, 129, 255, 511], device=DEV)
    sub = m(DiffusableBatch(X=X[idx].contiguous(), timesteps=t[idx].contiguous()))
    # a different batch changes the workgroup shape and the fp32 summation order; in bf16 mode a 1e-7 change can flip
    # an activation rounding (2^-9 relative), so rows agree to bf16 noise, not bitwise (fp32 mode: test_gpu_score.py)
    assert (sub - a[idx]).abs().max() <= 2e-2 * a.abs().max()


def test_softmax_shift_fallback_layer_kernel():
    """Same property for the standalone attention kernel of the step-by-step path (FDIFF_NO_MEGA routes there;
    FDIFF_ATTN_EXACT forces its exact two-pass form)."""
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    from oracle.make_golden import CFG_DEFAULT
    cfg = dict(CFG_DEFAULT, L=2)
    saved = {k: os.environ.get(k) for k in ("FDIFF_NO_MEGA", "FDIFF_ATTN_EXACT")}
    try:
        os.environ["FDIFF_NO_MEGA"] = "1"
        for scale in (1.0, 40.0):
            m, _, sd = make_model(cfg, precision="bf16")
            st = m.state_dict()
            for k in list(st):
                if k.endswith("self_attn.in_proj_weight"):
                    st[k] = st[k] * scale
            m.load_state_dict(st)
            m.eval()
            X = dev(W.randn("fb_x", (6, cfg["T"], cfg["C"]), 2))
            t = dev(W.uniform("fb_t", (6,), 2, 1e-5, 1.0))
            os.environ.pop("FDIFF_ATTN_EXACT", None)
            fast = host(m(DiffusableBatch(X=X, timesteps=t)))
            os.environ["FDIFF_ATTN_EXACT"] = "1"
            exact = host(m(DiffusableBatch(X=X, timesteps=t)))
            assert np.isfinite(fast).all() and np.isfinite(exact).all()
            np.testing.assert_allclose(fast, exact, atol=2e-3 * np.abs(exact).max(), rtol=0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_softmax_shift_fallback_equals_exact_path():
    """The persistent kernel shifts the softmax by the bound |q| max|k| and redoes a unit with the exact row maximum
    when a row sum underflows.  With the attention input projection scaled up (logits of several hundred) the bound
    overshoots by more than 2^100, so the fallback must trigger; its result must equal the always-exact path
    (FDIFF_MEGA_DBG=32) -- both are bf16, the only difference allowed is the rounding of the shift."""
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    from oracle.make_golden import CFG_DEFAULT
    cfg = dict(CFG_DEFAULT, L=2)
    outs = {}
    for scale in (1.0, 40.0):
        m, _, sd = make_model(cfg, precision="bf16")
        st = m.state_dict()
        for k in list(st):
            if k.endswith("self_attn.in_proj_weight"):
                st[k] = st[k] * scale
        m.load_state_dict(st)
        m.eval()
        X = dev(W.randn("fb_x", (6, cfg["T"], cfg["C"]), 2))
        t = dev(W.uniform("fb_t", (6,), 2, 1e-5, 1.0))
        old = os.environ.get("FDIFF_MEGA_DBG")
        try:
            os.environ.pop("FDIFF_MEGA_DBG", None)
            fast = host(m(DiffusableBatch(X=X, timesteps=t)))
            os.environ["FDIFF_MEGA_DBG"] = "32"
            exact = host(m(DiffusableBatch(X=X, timesteps=t)))
            os.environ["FDIFF_MEGA_DBG"] = "64"                      # bound only, fallback suppressed
            nofb = host(m(DiffusableBatch(X=X, timesteps=t)))
        finally:
            if old is None:
                os.environ.pop("FDIFF_MEGA_DBG", None)
            else:
                os.environ["FDIFF_MEGA_DBG"] = old
        assert np.isfinite(fast).all() and np.isfinite(exact).all()
        if scale > 1.0:   # the fixture really drives rows into underflow: without the fallback the result is wrong
            assert (not np.isfinite(nofb).all()) or np.abs(nofb - exact).max() > 0.1 * np.abs(exact).max()
        else:
            np.testing.assert_allclose(nofb, exact, atol=2e-3 * np.abs(exact).max(), rtol=0)
        outs[scale] = (fast, exact)
        np.testing.assert_allclose(fast, exact, atol=2e-3 * np.abs(exact).max(), rtol=0)
    assert np.abs(outs[40.0][1] - outs[1.0][1]).max() > 1e-2, "scaled projection must change the output"


def test_softmax_shift_failure_memory_in_the_sampler_loop():
    """Sampler mode: a wave that saw a layer's bound fail sends that layer's units straight to the exact form until the next
    retry step (every 16th).  20 steps with the scaled projection (every step fails) must agree with the always-exact loop
    (FDIFF_MEGA_DBG=32) to the rounding of the shift, across the retry boundary, and stay seed-reproducible."""
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    from oracle.make_golden import CFG_DEFAULT
    cfg = dict(CFG_DEFAULT, L=2)
    old = os.environ.get("FDIFF_MEGA_DBG")
    outs = []
    try:
        for dbg in (None, None, "32"):
            m, _, sd = make_model(cfg, precision="bf16")
            st = m.state_dict()
            for k in list(st):
                if k.endswith("self_attn.in_proj_weight"):
                    st[k] = st[k] * 40.0
            m.load_state_dict(st)
            m.eval()
            if dbg is None:
                os.environ.pop("FDIFF_MEGA_DBG", None)
            else:
                os.environ["FDIFF_MEGA_DBG"] = dbg
            sampler = DiffusionSampler(score_model=m, sample_batch_size=6)
            torch.manual_seed(3)
            outs.append(sampler.sample(num_samples=6, num_diffusion_steps=20).numpy())
    finally:
        if old is None:
            os.environ.pop("FDIFF_MEGA_DBG", None)
        else:
            os.environ["FDIFF_MEGA_DBG"] = old
    assert np.isfinite(outs[0]).all() and np.isfinite(outs[2]).all()
    assert np.array_equal(outs[0], outs[1])
    np.testing.assert_allclose(outs[0], outs[2], atol=2e-3 * max(1.0, np.abs(outs[2]).max()), rtol=0)


ATTN_SHAPES = {
    # long-series attention kernel (fd_attn_bf16.hip): full 128-key blocks + ragged tail (T=365: 23 key tiles), odd tile count,
    # head_dim 5 and 7 (7: the ones row sits in V^T's last slot, max|k|^2 words move out of it), d_model 56 = two k-steps
    "drought": dict(T=365, C=1, D=72, L=2, H=12),
    "hd5": dict(T=400, C=7, D=60, L=2, H=12),
    "hd7_ks2": dict(T=300, C=3, D=56, L=2, H=8),
    "blocks_only": dict(T=512, C=2, D=72, L=1, H=12),
}


@pytest.mark.parametrize("env", [{}, {"FDIFF_ATTN_UNFUSED": "1"}, {"FDIFF_ATTN_SLICES": "4"}, {"FDIFF_ATTN_SLICES": "1"},
                                 {"FDIFF_ATTN_EXACT": "1"}, {"FDIFF_ATTN_UNFUSED": "1", "FDIFF_ATTN_SLICES": "2"},
                                 {"FDIFF_ATTN_PAD_MIN": "1"}, {"FDIFF_ATTN_PAD_MIN": "9"},
                                 {"FDIFF_ATTN_PAD_MIN": "1", "FDIFF_ATTN_UNFUSED": "1"}],
                         ids=lambda e: ",".join(f"{k[6:]}={v}" for k, v in e.items()) or "default")
@pytest.mark.parametrize("name", sorted(ATTN_SHAPES))
def test_layer_attention_kernel_variants(name, env):
    """Every instantiation / launch shape of k_attention_bf16 against the oracle: projections fused (the persistent kernel's
    weight images) or packed q|k|v input (FDIFF_ATTN_UNFUSED), 1 / 2 / 4 query slices per (series, head pair), the exact two-pass
    softmax, the bound-shifted fast path (default), and the last partial key block padded to a whole one with zero tiles (default
    from three tiles on; FDIFF_ATTN_PAD_MIN=1 always, =9 never)."""
    cfg = ATTN_SHAPES[name]
    B = 2
    saved = {k: os.environ.get(k) for k in ("FDIFF_ATTN_UNFUSED", "FDIFF_ATTN_SLICES", "FDIFF_ATTN_EXACT", "FDIFF_ATTN_PAD_MIN")}
    try:
        for k in saved:
            os.environ.pop(k, None)
        os.environ.update(env)
        m, _, sd = make_model(cfg, precision="bf16")
        assert "per-layer" in m.plan(B)[0], m.plan(B)
        X = W.randn(f"at_x_{name}", (B, cfg["T"], cfg["C"]), 2)
        t = W.uniform(f"at_t_{name}", (B,), 2, 1e-5, 1.0)
        out = run(m, X, t)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ref = O.score_forward(sd, X, t, cfg["H"])
    err, rms = report_err(f"forward bf16 layer-attention {name} {env}", out, ref)
    assert err <= 2e-2 and rms <= 1e-2, (err, rms)
