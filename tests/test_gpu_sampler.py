"""GPU parity: fd_sampler_run (through DiffusionSampler) vs the reference's 20-step trajectories with injected
noise (golden) and the oracle.  Tolerance: 1e-4 of the trajectory scale."""
import numpy as np
import pytest
import torch

from oracle import weights as W
from oracle.make_golden import CFG_DEFAULT, CFG_TINY, SDE_CASES

from .gpu_util import DEV, dev, host, make_model

pytestmark = pytest.mark.gpu
CFGS = {"default": CFG_DEFAULT, "tiny": CFG_TINY}


@pytest.mark.parametrize("name,B", [("tiny", 6), ("default", 2)])
def test_trajectory_f32_vs_golden(golden, name, B):
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    g = golden("sampler")
    cfg = CFGS[name]
    shape = (B, cfg["T"], cfg["C"])
    zp = W.randn(f"samp_prior_{name}", shape, 4)
    zs = np.stack([W.randn(f"samp_z_{name}_{i}", shape, 4) for i in range(20)])
    for ci, (kind, p) in enumerate(SDE_CASES[:2]):
        tag = f"{name}_{kind}{ci}"
        for nsteps, key in ((1, "step1"), (5, "step5"), (20, "final")):
            m, sch, _ = make_model(cfg, kind, p, precision="fp32")
            sampler = DiffusionSampler(score_model=m, sample_batch_size=B)
            if nsteps == 20:
                X = sampler.sample(num_samples=B, num_diffusion_steps=20, prior_noise=[dev(zp)],
                                   step_noise=[dev(zs)])
                assert X.device.type == "cpu"
                got = X.numpy()
            else:
                # intermediate states: run the first k steps of the same 20-step grid through the C ABI
                import ctypes as C
                from fourierdiffusion_amd import _C
                sch.set_timesteps(20)
                Xd = sampler.sample_prior(B, noise=dev(zp))
                ctx, h = m._engine()
                ts = (C.c_float * nsteps)(*sch.timesteps[:nsteps].tolist())
                pz = dev(zs[:nsteps])
                prm = sch._c_params()
                _C.check(_C.lib().fd_sampler_run(h, C.byref(prm), sch.G_on(Xd.device).data_ptr(), ts, nsteps,
                                                 float(sch.step_size), Xd.data_ptr(), pz.data_ptr(), 0, 0, B,
                                                 _C.FD_MODE_F32, None), ctx)
                got = host(Xd)
            ref = g[f"{key}_{tag}"]
            scale = max(1.0, np.abs(ref).max())
            assert np.abs(got - ref).max() <= 1e-4 * scale, (tag, key, np.abs(got - ref).max(), scale)


@pytest.mark.parametrize("kind", ["vp", "ve"])
def test_reference_sampler_test(kind):
    """tests/test_sampling.py:21-40 of the reference: default-ctor ScoreModule (d=60, L=3, H=12), 48 samples in
    batches of 12, 10 steps -> (48, 50, 3); plus the batching rule incl. the dropped remainder (sampler.py:63)."""
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    from fourierdiffusion_amd.schedulers.sde import VEScheduler, VPScheduler
    sch = VPScheduler() if kind == "vp" else VEScheduler()
    m = ScoreModule(n_channels=3, max_len=50, noise_scheduler=sch).to(DEV)
    m.precision = "fp32"
    sch.set_noise_scaling(max_len=50)
    sampler = DiffusionSampler(score_model=m, sample_batch_size=12)
    s = sampler.sample(num_samples=48, num_diffusion_steps=10)
    assert s.shape == (48, 50, 3) and s.device.type == "cpu" and torch.isfinite(s).all()
    assert sampler.sample(num_samples=50, num_diffusion_steps=2).shape[0] == 48     # remainder dropped
    assert sampler.sample(num_samples=5, num_diffusion_steps=2).shape[0] == 5       # fewer than one batch


def test_sampling_is_seed_reproducible_and_seed_sensitive():
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    m, sch, _ = make_model(CFG_TINY, precision="fp32")
    sampler = DiffusionSampler(score_model=m, sample_batch_size=8)
    torch.manual_seed(3)
    a = sampler.sample(num_samples=16, num_diffusion_steps=5)
    torch.manual_seed(3)
    b = sampler.sample(num_samples=16, num_diffusion_steps=5)
    torch.manual_seed(4)
    c = sampler.sample(num_samples=16, num_diffusion_steps=5)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert not torch.equal(a[:8], a[8:])          # the two batches draw disjoint Philox ranges


def test_stepwise_api_equals_fused_loop():
    """reverse_diffusion_step x N (the reference's structure) == one fd_sampler_run call."""
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    cfg = CFG_TINY
    B, N = 4, 6
    shape = (B, cfg["T"], cfg["C"])
    zp = dev(W.randn("sw_p", shape, 5))
    zs = dev(np.stack([W.randn(f"sw_{i}", shape, 5) for i in range(N)]))
    m, sch, _ = make_model(cfg, precision="fp32")
    sampler = DiffusionSampler(score_model=m, sample_batch_size=B)
    fused = sampler.sample(num_samples=B, num_diffusion_steps=N, prior_noise=[zp], step_noise=[zs])
    sch.set_timesteps(N)
    X = sampler.sample_prior(B, noise=zp)
    for i, t in enumerate(sch.timesteps):
        tb = torch.full((B,), float(t), device=DEV)
        X = sampler.reverse_diffusion_step(DiffusableBatch(X=X, timesteps=tb), noise=zs[i])
    assert torch.allclose(X.cpu(), fused, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("C", [3, 6, 12])
def test_persistent_sampler_equals_stepwise_launches_bf16(C):
    """bf16 mode: the one-launch reverse-diffusion loop (Euler-Maruyama step + Philox noise inside the persistent kernel)
    against one score launch + fd_sde_step per step (FDIFF_SAMPLER_STEPWISE), same seed.  C = 3 and 6 exercise lanes whose
    4 channels straddle two Philox counters; the noise stream must be the standalone kernel's in every case."""
    import os
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    cfg = dict(T=40, C=C, D=24, L=2, H=4)
    outs = []
    for stepwise in (False, True):
        m, _, _ = make_model(cfg, precision="bf16")
        sampler = DiffusionSampler(score_model=m, sample_batch_size=5)
        old = os.environ.get("FDIFF_SAMPLER_STEPWISE")
        try:
            if stepwise:
                os.environ["FDIFF_SAMPLER_STEPWISE"] = "1"
            else:
                os.environ.pop("FDIFF_SAMPLER_STEPWISE", None)
            torch.manual_seed(77)
            outs.append(sampler.sample(num_samples=5, num_diffusion_steps=12).numpy())
        finally:
            if old is None:
                os.environ.pop("FDIFF_SAMPLER_STEPWISE", None)
            else:
                os.environ["FDIFF_SAMPLER_STEPWISE"] = old
    bad = ~np.isfinite(outs[0])
    assert not bad.any(), f"{int(bad.sum())} non-finite samples of {bad.size} at {np.argwhere(bad)[:8].tolist()}; stepwise non-finite: {int((~np.isfinite(outs[1])).sum())}"
    # identical network kernel and noise; only the fusion of the SDE step differs (fp32 rounding of the update).  In bf16
    # mode a 1e-7 change of x can flip an activation's bf16 rounding (2^-9 relative) in the next forward, and this toy model's
    # trajectories grow to |x| ~ 300, so the two loops agree to bf16 noise, not to fp32 rounding: 2e-3 of the scale as in
    # test_gpu_bench_instantiation.py (measured 1.8e-4; a wrong lane -> Philox-counter map gives O(1): independent normals)
    np.testing.assert_allclose(outs[0], outs[1], atol=2e-3 * max(1.0, np.abs(outs[1]).max()), rtol=0)


@pytest.mark.parametrize("cfg", [dict(T=300, C=12, D=72, L=2, H=12), dict(T=272, C=3, D=72, L=2, H=12), dict(T=260, C=6, D=24, L=2, H=4),
                                 dict(T=290, C=28, D=72, L=1, H=12)],
                         ids=lambda c: f"T{c['T']}_C{c['C']}_D{c['D']}")
def test_long_series_fused_step_equals_separate_launches_bf16(cfg):
    """T > 256 (the persistent kernel does not fit): per diffusion step the loop runs the layer launches and ONE launch for
    unembed + reverse-SDE step + the next step's embedding (k_unembed_step_embed, the persistent kernel's arithmetic) against
    the separate unembedding GEMM / fd_sde_step / time embedding / embedding launches (FDIFF_SAMPLER_UNFUSED_STEP), same seed.
    C = 3 and 6: lanes whose four channels straddle two Philox counters; C = 28: two channel tiles; ragged last token tile.
    sampler.py:83-104, sde.py:129-165, score_models.py:78-90."""
    import os
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    outs = []
    old = os.environ.get("FDIFF_SAMPLER_UNFUSED_STEP")
    try:
        for unfused in (False, True):
            m, _, _ = make_model(cfg, precision="bf16")
            sampler = DiffusionSampler(score_model=m, sample_batch_size=3)
            if unfused:
                os.environ["FDIFF_SAMPLER_UNFUSED_STEP"] = "1"
            else:
                os.environ.pop("FDIFF_SAMPLER_UNFUSED_STEP", None)
            torch.manual_seed(31)
            outs.append(sampler.sample(num_samples=3, num_diffusion_steps=8).numpy())
    finally:
        if old is None:
            os.environ.pop("FDIFF_SAMPLER_UNFUSED_STEP", None)
        else:
            os.environ["FDIFF_SAMPLER_UNFUSED_STEP"] = old
    assert np.isfinite(outs[0]).all() and np.isfinite(outs[1]).all()
    # same layer kernels and noise stream; the fused launch embeds / unembeds with bf16 MFMA operands where the separate launches
    # use fp32 GEMMs: bf16 noise (2^-9 relative per operand), not fp32 rounding -- a wrong lane -> counter map gives O(1)
    scale = max(1.0, np.abs(outs[1]).max())
    err = np.abs(outs[0] - outs[1]).max() / scale
    print(f"[parity] fused long-series step vs separate launches T={cfg['T']} C={cfg['C']} D={cfg['D']}: max err / scale = {err:.3e}")
    assert err <= 5e-3, err


def test_long_series_fused_step_at_a_production_step_count_bf16():
    """the fused launch rounds x to bf16 operands at EVERY diffusion step (embedding / unembedding by MFMA) where the
    separate launches and the reference (score_models.py:78-90) embed in fp32; the 8-step test above cannot show a drift that
    accumulates.  250 reverse-SDE steps (VP, the hydra default runs 1000: `cmd/conf/sample.yaml`), 64 series, same Philox key:
    the two paths share noise and layer kernels, so they are compared path by path AND on the statistics a user evaluates
    (per-channel mean / standard deviation over batch and time).  Bounds: statistics within 5e-3 of the sample scale, paths
    within 1e-2 (measured 1.2e-3, logged; the bf16 layer kernels themselves carry ~5e-3 per forward)."""
    import os
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    from .gpu_util import report_err
    cfg, B, N = dict(T=272, C=3, D=72, L=2, H=12), 64, 250
    outs = []
    old = os.environ.get("FDIFF_SAMPLER_UNFUSED_STEP")
    try:
        for unfused in (False, True):
            m, _, _ = make_model(cfg, precision="bf16")
            sampler = DiffusionSampler(score_model=m, sample_batch_size=B)
            if unfused:
                os.environ["FDIFF_SAMPLER_UNFUSED_STEP"] = "1"
            else:
                os.environ.pop("FDIFF_SAMPLER_UNFUSED_STEP", None)
            torch.manual_seed(47)
            outs.append(sampler.sample(num_samples=B, num_diffusion_steps=N).numpy().astype(np.float64))
    finally:
        if old is None:
            os.environ.pop("FDIFF_SAMPLER_UNFUSED_STEP", None)
        else:
            os.environ["FDIFF_SAMPLER_UNFUSED_STEP"] = old
    fused, sep = outs
    assert np.isfinite(fused).all() and np.isfinite(sep).all()
    scale = max(1.0, float(np.abs(sep).std()))
    err, rms = report_err(f"fused long-series step vs separate fp32 launches after {N} steps (paths)", fused, sep)
    dm = np.abs(fused.mean(axis=(0, 1)) - sep.mean(axis=(0, 1))).max() / scale
    ds = np.abs(fused.std(axis=(0, 1)) - sep.std(axis=(0, 1))).max() / scale
    print(f"[parity] fused long-series step after {N} steps: per-channel mean differs by {dm:.3e}, std by {ds:.3e} of the sample scale")
    assert dm <= 5e-3 and ds <= 5e-3, (dm, ds)
    assert err <= 1e-2, err        # (measured 1.2e-3)


def test_sampler_merges_the_reference_batches_into_device_sized_launches():
    """sample_batch_size is the reference's memory knob (200 by default, cmd/conf/sampler/default.yaml); the same num_batches x
    batch_size series go to the engine in launches sized for the device (DiffusionSampler._launch_sizes).  Same count and shape as
    the reference's rule (sampler.py:63: the remainder of num_samples is dropped), reproducible under a seed, and the same
    distribution as the unmerged launches (per-channel mean / std of 1000 series within 8 % of the sample scale: the two runs use
    different Philox keys, so this is a two-sample comparison, not a path comparison)."""
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    cfg = dict(T=100, C=12, D=72, L=2, H=12)
    m, _, _ = make_model(cfg, precision="bf16")
    merged = DiffusionSampler(score_model=m, sample_batch_size=200)
    plain = DiffusionSampler(score_model=m, sample_batch_size=200, merge_batches=False)
    assert merged.merge_batches and not plain.merge_batches
    sizes = merged._launch_sizes(1000, 1)
    assert sum(sizes) == 1000 and len(sizes) < 5, sizes
    assert merged._launch_sizes(5 * 4096 + 3, 1)[-1] >= 512 or len(merged._launch_sizes(5 * 4096 + 3, 1)) == 5      # no sliver launch
    torch.manual_seed(9)
    a = merged.sample(num_samples=1090, num_diffusion_steps=12)        # 5 batches of 200: 1000 series, 90 dropped
    torch.manual_seed(9)
    a2 = merged.sample(num_samples=1090, num_diffusion_steps=12)
    torch.manual_seed(9)
    b = plain.sample(num_samples=1090, num_diffusion_steps=12)
    assert tuple(a.shape) == tuple(b.shape) == (1000, cfg["T"], cfg["C"])
    assert torch.equal(a, a2) and torch.isfinite(a).all() and torch.isfinite(b).all()
    scale = float(b.std())
    dm = float((a.mean(dim=(0, 1)) - b.mean(dim=(0, 1))).abs().max()) / scale
    ds = float((a.std(dim=(0, 1)) - b.std(dim=(0, 1))).abs().max()) / scale
    print(f"[parity] merged vs per-batch sampler launches (1000 series, 12 steps): per-channel mean differs by {dm:.3e}, std by {ds:.3e} of the scale")
    assert dm <= 8e-2 and ds <= 8e-2, (dm, ds)
    # a single batch, injected noise: the caller's launches are kept
    assert DiffusionSampler(score_model=m, sample_batch_size=64).sample(num_samples=64, num_diffusion_steps=3).shape[0] == 64


def test_precomputed_time_embedding_table_is_bit_identical():
    """Sampler mode reads the time embedding of every step from a table filled before the launch (fd_mega_temb_table: all
    series share the step's t); FDIFF_MEGA_NO_TEMB_TABLE computes it inside the kernel every step as forward mode does.
    Same arithmetic, so the samples must be bit-identical -- on a two-series-per-workgroup shape and a dynamic one."""
    import os
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    for cfg, B in ((dict(T=100, C=12, D=72, L=2, H=12), 512), (dict(T=40, C=3, D=24, L=2, H=4), 7)):
        outs = []
        for no_table in (False, True):
            m, _, _ = make_model(cfg, precision="bf16")
            sampler = DiffusionSampler(score_model=m, sample_batch_size=B)
            old = os.environ.get("FDIFF_MEGA_NO_TEMB_TABLE")
            try:
                if no_table:
                    os.environ["FDIFF_MEGA_NO_TEMB_TABLE"] = "1"
                else:
                    os.environ.pop("FDIFF_MEGA_NO_TEMB_TABLE", None)
                torch.manual_seed(5)
                outs.append(sampler.sample(num_samples=B, num_diffusion_steps=6).numpy())
            finally:
                if old is None:
                    os.environ.pop("FDIFF_MEGA_NO_TEMB_TABLE", None)
                else:
                    os.environ["FDIFF_MEGA_NO_TEMB_TABLE"] = old
        assert np.isfinite(outs[0]).all()
        assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_stepwise_loop_step_table_and_register_embed_are_bit_identical(precision):
    """The step-by-step loop fills the t vectors of all steps in one launch (FDIFF_SAMPLER_FILL_PER_STEP: one k_fill per step as
    before) and embeds with the weights-in-registers kernel when C % 4 == 0 (FDIFF_EMBED_LDS: the LDS form).  Same values, same
    fma order: the samples must be bit-identical."""
    import os
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    cfg = dict(T=48, C=8, D=24, L=2, H=4)
    outs = []
    keys = ("FDIFF_SAMPLER_STEPWISE", "FDIFF_SAMPLER_FILL_PER_STEP", "FDIFF_EMBED_LDS")
    old = {k: os.environ.get(k) for k in keys}
    try:
        for legacy in (False, True):
            os.environ["FDIFF_SAMPLER_STEPWISE"] = "1"
            for k in keys[1:]:
                if legacy:
                    os.environ[k] = "1"
                else:
                    os.environ.pop(k, None)
            m, _, _ = make_model(cfg, precision=precision)
            sampler = DiffusionSampler(score_model=m, sample_batch_size=5)
            torch.manual_seed(11)
            outs.append(sampler.sample(num_samples=5, num_diffusion_steps=9).numpy())
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert np.isfinite(outs[0]).all()
    assert np.array_equal(outs[0], outs[1])
