#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (this container only).

TEST INFRASTRUCTURE ONLY.  Imports /root/reference/src/fdiff *by path* (nothing is copied
into this repo) after registering stand-in modules for three third-party packages that
are absent from the image and are NOT on the arithmetic path (pytorch_lightning,
diffusers, torchvision -- SURVEY 8c / Appendix B).  Inputs and weights come from the
counter-based recipe in oracle/weights.py, so the fixtures hold expected OUTPUTS only.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz
"""
from __future__ import annotations

import os
import sys
import types
from contextlib import contextmanager

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import weights as W  # noqa: E402

REF = "/root/reference/src"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present; golden fixtures can only be regenerated in the build container")

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

    stub("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=object,
         Callback=object, Trainer=object)
    stub("pytorch_lightning.utilities")
    stub("pytorch_lightning.utilities.types", OptimizerLRScheduler=object)
    stub("diffusers")
    stub("diffusers.optimization", get_cosine_schedule_with_warmup=lambda **k: None)
    class StandInMLP(nn.Sequential):
        """torchvision.ops.MLP is an unpinned dependency of the reference and absent from this image.  Its published
        structure (torchvision/ops/misc.py: per hidden width Linear -> activation -> Dropout, then Linear -> Dropout) is
        restated here ONLY so that the reference's MLPScoreModule class can be constructed and run for the fixtures of
        SURVEY 8(f)4 -- parity of the MLP backbone is therefore pinned to the reference's module code over this stand-in,
        and UNPINNED against torchvision itself (said so in oracle/fdiff_oracle.py and DESIGN.md)."""

        def __init__(self, in_channels, hidden_channels, dropout=0.0, **unused):
            layers, d = [], in_channels
            for hdim in hidden_channels[:-1]:
                layers += [nn.Linear(d, hdim), nn.ReLU(), nn.Dropout(dropout)]
                d = hdim
            layers += [nn.Linear(d, hidden_channels[-1]), nn.Dropout(dropout)]
            super().__init__(*layers)

    stub("torchvision")
    stub("torchvision.ops", MLP=StandInMLP)
    # the reference's `fdiff` is a namespace package (no __init__.py): the repo's own `fdiff` alias package would shadow
    # it from ANY position on sys.path, so the repo root leaves the path while the reference is imported
    saved_path = list(sys.path)
    sys.path[:] = [REF] + [p for p in sys.path if os.path.abspath(p or os.getcwd()) != ROOT]
    assert "fdiff" not in sys.modules
    import fdiff.models.score_models as sm
    import fdiff.sampling.sampler as sp
    import fdiff.schedulers.sde as sde
    import fdiff.utils.dataclasses as dc
    import fdiff.utils.fourier as fourier
    import fdiff.utils.losses as losses
    assert fourier.__file__.startswith(REF), fourier.__file__
    sys.path[:] = saved_path
    return types.SimpleNamespace(sm=sm, sp=sp, sde=sde, dc=dc, fourier=fourier, losses=losses)


@contextmanager
def replay_noise(randn_like_seq=None, randn_seq=None):
    """Replace torch.randn_like / torch.randn by replaying stored tensors."""
    orig_like, orig_randn = torch.randn_like, torch.randn
    like_it = iter(randn_like_seq or [])
    randn_it = iter(randn_seq or [])
    if randn_like_seq is not None:
        torch.randn_like = lambda x, *a, **k: next(like_it).to(x.dtype)
    if randn_seq is not None:
        torch.randn = lambda *a, **k: next(randn_it)
    try:
        yield
    finally:
        torch.randn_like, torch.randn = orig_like, orig_randn


def t_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def build_ref_model(R, cfg, sde_kind, sde_p, scaling, seed):
    if sde_kind == "vp":
        sch = R.sde.VPScheduler(beta_min=sde_p[0], beta_max=sde_p[1], fourier_noise_scaling=scaling)
    else:
        sch = R.sde.VEScheduler(sigma_min=sde_p[0], sigma_max=sde_p[1], fourier_noise_scaling=scaling)
    sch.set_noise_scaling(cfg["T"])
    m = R.sm.ScoreModule(n_channels=cfg["C"], max_len=cfg["T"], noise_scheduler=sch,
                         fourier_noise_scaling=scaling, d_model=cfg["D"], num_layers=cfg["L"],
                         n_head=cfg["H"])
    sd = W.make_state_dict(cfg["C"], cfg["T"], cfg["D"], cfg["L"], seed=seed)
    m.load_state_dict({k: t_(v) for k, v in sd.items()}, strict=True)
    return m, sch, sd


# Shared with tests/: model configurations used by the fixtures.
CFG_DEFAULT = dict(T=100, C=12, D=72, L=10, H=12)      # hydra default score model, ecg-synth shape
CFG_TINY = dict(T=20, C=3, D=8, L=2, H=4)             # the reference tests' config (tests/test_score_models.py:13-19)
CFG_ODD = dict(T=37, C=5, D=24, L=2, H=4)             # odd T, head_dim 6
DFT_T = [16, 24, 100, 101, 187, 251, 252, 256, 365, 1024]
DFT_C = [1, 12, 28]
DFT_B = 2


def gen_dft(R):
    out = {}
    for T in DFT_T:
        for C in DFT_C:
            x = W.randn(f"dft_x_{T}_{C}", (DFT_B, T, C), 0)
            out[f"dft_{T}_{C}"] = R.fourier.dft(t_(x)).numpy()
            xt = W.randn(f"idft_x_{T}_{C}", (DFT_B, T, C), 0)
            out[f"idft_{T}_{C}"] = R.fourier.idft(t_(xt)).numpy()
    # known-answer pairs: impulse and a pure cosine
    for T in (16, 15):
        imp = np.zeros((1, T, 1), np.float32)
        imp[0, 3, 0] = 1.0
        out[f"dft_impulse_{T}"] = R.fourier.dft(t_(imp)).numpy()
        n = np.arange(T, dtype=np.float64)
        cosw = np.cos(2 * np.pi * 2 * n / T).astype(np.float32).reshape(1, T, 1)
        out[f"dft_cos2_{T}"] = R.fourier.dft(t_(cosw)).numpy()
    np.savez_compressed(os.path.join(OUT, "dft.npz"), **out)


def gen_spectral(R):
    """spectral_density / localization_metrics / smooth_frequency of the reference (fourier.py:90-209)."""
    out = {}
    for T in (16, 100, 101, 187):
        for C in (1, 12):
            x = W.randn(f"spec_x_{T}_{C}", (3, T, C), 0)
            # a localised bump on top of the noise so that the two metrics differ between series
            x[1, T // 3: T // 3 + 4] += 3.0
            out[f"dens_{T}_{C}"] = R.fourier.spectral_density(t_(x)).numpy()
            out[f"dens_nodft_{T}_{C}"] = R.fourier.spectral_density(t_(x), apply_dft=False).numpy()
            loc, sloc = R.fourier.localization_metrics(t_(x))
            out[f"loc_{T}_{C}"] = np.stack([loc.numpy(), sloc.numpy()])
            if T % 2 == 1:
                for sigma in (1.0, 4.5):
                    out[f"smooth_{T}_{C}_{sigma}"] = R.fourier.smooth_frequency(t_(x), sigma).numpy()
    # the reference fails for even lengths (kernel of T-1 frequencies): record that it does
    try:
        R.fourier.smooth_frequency(t_(W.randn("spec_even", (1, 16, 1), 0)), 1.0)
        out["smooth_even_raises"] = np.array(0)
    except RuntimeError:
        out["smooth_even_raises"] = np.array(1)
    np.savez_compressed(os.path.join(OUT, "spectral.npz"), **out)


SDE_CASES = [("vp", (0.1, 20.0)), ("ve", (0.01, 2.0)), ("ve", (0.01, 50.0))]


def make_ref_sde(R, kind, p, scaling, T):
    s = (R.sde.VPScheduler(beta_min=p[0], beta_max=p[1], fourier_noise_scaling=scaling) if kind == "vp"
         else R.sde.VEScheduler(sigma_min=p[0], sigma_max=p[1], fourier_noise_scaling=scaling))
    s.set_noise_scaling(T)
    return s


def gen_sde(R):
    out = {}
    for T in (100, 101):
        for scaling in (False, True):
            s = make_ref_sde(R, "vp", (0.1, 20.0), scaling, T)
            out[f"G_{T}_{int(scaling)}"] = s.G.numpy()
    for N in (10, 1000, 2000):
        s = make_ref_sde(R, "vp", (0.1, 20.0), True, 100)
        s.set_timesteps(N)
        out[f"timesteps_{N}"] = s.timesteps.numpy()
        out[f"step_size_{N}"] = s.step_size.numpy()
    B, T, C = 4, 20, 3
    x = W.randn("sde_x", (B, T, C), 1)
    score = W.randn("sde_score", (B, T, C), 1)
    z = W.randn("sde_z", (B, T, C), 1)
    tvals = np.array([1e-5, 0.1, 0.5, 1.0], np.float32)
    for ci, (kind, p) in enumerate(SDE_CASES):
        for scaling in (False, True):
            tag = f"{kind}{ci}_{int(scaling)}"
            s = make_ref_sde(R, kind, p, scaling, T)
            mean, std = s.marginal_prob(t_(x), t_(tvals))
            out[f"mean_{tag}"] = mean.numpy()
            out[f"std_{tag}"] = std.numpy()
            s.set_timesteps(1000)
            for ti, tv in enumerate((0.37, 1e-5, 1.0)):
                with replay_noise(randn_like_seq=[t_(z)]):
                    o = s.step(t_(score), tv, t_(x)).prev_sample
                out[f"step_{tag}_{ti}"] = o.numpy()
            with replay_noise(randn_seq=[t_(z)]):
                out[f"prior_{tag}"] = s.prior_sampling((B, T, C)).numpy()
    np.savez_compressed(os.path.join(OUT, "sde.npz"), **out)


def gen_score(R):
    out = {}
    for name, cfg, B in (("default", CFG_DEFAULT, 4), ("tiny", CFG_TINY, 3), ("odd", CFG_ODD, 3)):
        m, sch, _ = build_ref_model(R, cfg, "vp", (0.1, 20.0), True, seed=1234)
        m.eval()
        X = W.randn(f"score_x_{name}", (B, cfg["T"], cfg["C"]), 2)
        t = W.uniform(f"score_t_{name}", (B,), 2, 1e-5, 1.0)
        batch = R.dc.DiffusableBatch(X=t_(X), y=None, timesteps=t_(t))
        with torch.no_grad():
            fast = m(batch).numpy()
        torch.backends.mha.set_fastpath_enabled(False)
        with torch.no_grad():
            slow = m(batch).numpy()
        torch.backends.mha.set_fastpath_enabled(True)
        out[f"fast_{name}"] = fast
        out[f"slow_{name}"] = slow
        print(f"score {name}: fast-vs-slow max abs {np.abs(fast - slow).max():.3e}, |out| max {np.abs(fast).max():.3f}")
    np.savez_compressed(os.path.join(OUT, "score_forward.npz"), **out)


def gen_ckpt(R):
    """A checkpoint in the reference's Lightning layout: the reference model's own state_dict and the
    hyper-parameters Lightning's save_hyperparameters() records for ScoreModule.__init__ (score_models.py:25-43) with the
    reference's scheduler OBJECT pickled inside (class path fdiff.schedulers.sde.VPScheduler).  Same weights / inputs as the
    "tiny" case of score_forward.npz, so loading it on the engine must reproduce that golden output."""
    m, sch, _ = build_ref_model(R, CFG_TINY, "vp", (0.1, 20.0), True, seed=1234)
    hp = dict(n_channels=CFG_TINY["C"], max_len=CFG_TINY["T"], noise_scheduler=sch, fourier_noise_scaling=True,
              d_model=CFG_TINY["D"], num_layers=CFG_TINY["L"], n_head=CFG_TINY["H"], num_training_steps=1000, lr_max=1e-3,
              likelihood_weighting=False)
    ckpt = {"epoch": 3, "global_step": 40, "pytorch-lightning_version": "2.1.0", "state_dict": m.state_dict(),
            "hparams_name": "kwargs", "hyper_parameters": hp}
    torch.save(ckpt, os.path.join(OUT, "reference_tiny.ckpt"))


def zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, nn.MultiheadAttention):
            mod.dropout = 0.0


def gen_loss(R):
    out = {}
    for name, cfg, B in (("tiny", CFG_TINY, 5), ("odd", CFG_ODD, 3)):
        X = W.randn(f"loss_x_{name}", (B, cfg["T"], cfg["C"]), 3)
        z = W.randn(f"loss_z_{name}", (B, cfg["T"], cfg["C"]), 3)
        t = W.uniform(f"loss_t_{name}", (B,), 3, 0.05, 1.0)
        for ci, (kind, p) in enumerate(SDE_CASES[:2]):
            for lw in (False, True):
                m, sch, _ = build_ref_model(R, cfg, kind, p, True, seed=1234)
                zero_dropout(m)
                batch = R.dc.DiffusableBatch(X=t_(X), y=None, timesteps=t_(t))
                # eval-mode value
                fn_eval = R.losses.get_sde_loss_fn(sch, train=False, likelihood_weighting=lw)
                with replay_noise(randn_like_seq=[t_(z)]):
                    with torch.no_grad():
                        lv = fn_eval(m, batch)
                tag = f"{name}_{kind}{ci}_{int(lw)}"
                out[f"loss_{tag}"] = np.array(lv.item(), np.float64)
                # train-mode value + gradients with dropout p forced to 0
                fn_tr = R.losses.get_sde_loss_fn(sch, train=True, likelihood_weighting=lw)
                m.zero_grad()
                with replay_noise(randn_like_seq=[t_(z)]):
                    lt = fn_tr(m, batch)
                lt.backward()
                out[f"loss_train_{tag}"] = np.array(lt.item(), np.float64)
                if not lw:
                    for k, prm in m.named_parameters():
                        if prm.grad is not None:
                            out[f"grad_{tag}/{k}"] = prm.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)


CFG_BB = dict(T=20, C=3, D=24, L=2)                   # MLP / LSTM backbone fixtures (d_mlp 64 for the MLP)


def gen_backbones(R):
    """MLPScoreModule / LSTMScoreModule of the reference: eval forward, training loss and autograd gradients
    (dropout forced to 0, injected t and z) at a small configuration and, forward only, at the hydra configs' width
    (d_model 72; MLP d_mlp 1024 / 10 layers are cut to 3 layers to keep the fixture small)."""
    out = {}
    for kind in ("mlp", "lstm"):
        for name, cfg, B in (("small", CFG_BB, 4), ("wide", dict(T=50, C=4, D=72, L=3), 3)):
            d_mlp = 64 if name == "small" else 1024
            sch = R.sde.VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True)
            sch.set_noise_scaling(cfg["T"])
            if kind == "mlp":
                m = R.sm.MLPScoreModule(n_channels=cfg["C"], max_len=cfg["T"], noise_scheduler=sch, d_model=cfg["D"], d_mlp=d_mlp,
                                        num_layers=cfg["L"])
            else:
                m = R.sm.LSTMScoreModule(n_channels=cfg["C"], max_len=cfg["T"], noise_scheduler=sch, d_model=cfg["D"],
                                         num_layers=cfg["L"])
            sd = W.make_state_dict_backbone(kind, cfg["C"], cfg["T"], cfg["D"], cfg["L"], d_mlp=d_mlp, seed=4321)
            missing = m.load_state_dict({k: t_(v) for k, v in sd.items()}, strict=False)
            # the parent constructor's transformer pieces that the subclasses overwrite leave no keys; pos_encoder is None
            assert not missing.unexpected_keys and all(k.startswith("pos_encoder") for k in missing.missing_keys), missing
            X = W.randn(f"bb_x_{kind}_{name}", (B, cfg["T"], cfg["C"]), 5)
            t = W.uniform(f"bb_t_{kind}_{name}", (B,), 5, 0.05, 1.0)
            z = W.randn(f"bb_z_{kind}_{name}", (B, cfg["T"], cfg["C"]), 5)
            m.eval()
            with torch.no_grad():
                out[f"fwd_{kind}_{name}"] = m(R.dc.DiffusableBatch(X=t_(X), y=None, timesteps=t_(t))).numpy()
            if name != "small":
                continue
            zero_dropout(m)
            fn_tr = R.losses.get_sde_loss_fn(sch, train=True, likelihood_weighting=False)
            m.zero_grad()
            with replay_noise(randn_like_seq=[t_(z)]):
                lt = fn_tr(m, R.dc.DiffusableBatch(X=t_(X), y=None, timesteps=t_(t)))
            lt.backward()
            out[f"loss_{kind}_{name}"] = np.array(lt.item(), np.float64)
            for k, prm in m.named_parameters():
                if prm.grad is not None:
                    out[f"grad_{kind}_{name}/{k}"] = prm.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "backbones.npz"), **out)


GRAD_STRIDE = 97      # the default-model gradient fixture keeps every 97th element of each tensor + its norms


def gen_grad_default(R):
    """Reference autograd gradients of the ONLY model the hydra configs name (d_model 72, 10 layers, ff 2048) at the ecg
    shape and at the BASELINE training shape (nasdaq T=252, C=6): per tensor the l2 norm, max |g| and a strided subset
    (every GRAD_STRIDE-th element), so that 3.2 M gradients cost 0.3 MB of fixture.  Dropout forced to 0, injected t, z."""
    out = {}
    for name, cfg, B in (("default", CFG_DEFAULT, 4), ("nasdaq", dict(T=252, C=6, D=72, L=10, H=12), 2)):
        X = W.randn(f"gd_x_{name}", (B, cfg["T"], cfg["C"]), 3)
        z = W.randn(f"gd_z_{name}", (B, cfg["T"], cfg["C"]), 3)
        t = W.uniform(f"gd_t_{name}", (B,), 3, 0.05, 1.0)
        m, sch, _ = build_ref_model(R, cfg, "vp", (0.1, 20.0), True, seed=1234)
        zero_dropout(m)
        batch = R.dc.DiffusableBatch(X=t_(X), y=None, timesteps=t_(t))
        fn_tr = R.losses.get_sde_loss_fn(sch, train=True, likelihood_weighting=False)
        m.zero_grad()
        with replay_noise(randn_like_seq=[t_(z)]):
            lt = fn_tr(m, batch)
        lt.backward()
        out[f"loss_{name}"] = np.array(lt.item(), np.float64)
        for k, prm in m.named_parameters():
            if prm.grad is None:
                continue
            g = prm.grad.numpy().astype(np.float64).ravel()
            out[f"norm_{name}/{k}"] = np.array([np.linalg.norm(g), np.abs(g).max()], np.float64)
            out[f"sub_{name}/{k}"] = g[::GRAD_STRIDE].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "grad_default.npz"), **out)


def gen_sampler(R):
    out = {}
    N = 20
    for name, cfg, B in (("tiny", CFG_TINY, 6), ("default", CFG_DEFAULT, 2)):
        for ci, (kind, p) in enumerate(SDE_CASES[:2]):
            m, sch, _ = build_ref_model(R, cfg, kind, p, True, seed=1234)
            shape = (B, cfg["T"], cfg["C"])
            zp = W.randn(f"samp_prior_{name}", shape, 4)
            zs = [W.randn(f"samp_z_{name}_{i}", shape, 4) for i in range(N)]
            sampler = R.sp.DiffusionSampler(score_model=m, sample_batch_size=B)
            rec = {}
            # record intermediate X by wrapping reverse_diffusion_step
            orig = sampler.reverse_diffusion_step
            cnt = {"i": 0}

            def wrapped(batch, orig=orig, cnt=cnt, rec=rec):
                Xn = orig(batch)
                cnt["i"] += 1
                if cnt["i"] in (1, 5, 20):
                    rec[cnt["i"]] = Xn.numpy().copy()
                return Xn

            sampler.reverse_diffusion_step = wrapped
            with replay_noise(randn_like_seq=[t_(z) for z in zs], randn_seq=[t_(zp)]):
                Xf = sampler.sample(num_samples=B, num_diffusion_steps=N)
            tag = f"{name}_{kind}{ci}"
            out[f"final_{tag}"] = Xf.numpy()
            for k, v in rec.items():
                out[f"step{k}_{tag}"] = v
            print(f"sampler {tag}: |X| max {np.abs(Xf.numpy()).max():.3f}")
    # batching rule (sampler.py:63,74-77): shapes only
    m, sch, _ = build_ref_model(R, CFG_TINY, "vp", (0.1, 20.0), True, seed=1234)
    shapes = []
    for ns, bs in ((48, 12), (50, 12), (5, 12), (12, 12)):
        s = R.sp.DiffusionSampler(score_model=m, sample_batch_size=bs)
        shapes.append([ns, bs, s.sample(num_samples=ns, num_diffusion_steps=2).shape[0]])
    out["batching"] = np.array(shapes, np.int64)
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **out)


def gen_dataset(R):
    """DiffusionDataset statistics (datamodules.py:42-62) restated with the reference's dft + torch mean/std
    (fdiff.dataloaders.datamodules itself needs pytorch_lightning/pandas-kaggle and is not importable)."""
    out = {}
    X = W.randn("ds_x", (16, 24, 3), 5)
    Xt = R.fourier.dft(t_(X))
    out["mean"] = Xt.mean(dim=0).numpy()
    out["std"] = Xt.std(dim=0).numpy()
    out["item3"] = ((Xt[3] - Xt.mean(dim=0)) / Xt.std(dim=0)).numpy()
    np.savez_compressed(os.path.join(OUT, "dataset.npz"), **out)


def gen_optim(R):
    """AdamW + clip from this image's torch 2.10 (the reference delegates to torch, score_models.py:123;
    Lightning's gradient_clip_val=1.0 is torch.nn.utils.clip_grad_norm_)."""
    out = {}
    p0 = W.randn("opt_p", (257,), 6)
    prm = nn.Parameter(t_(p0.copy()))
    opt = torch.optim.AdamW([prm], lr=1e-3)
    for it in range(3):
        g = W.randn(f"opt_g{it}", (257,), 6) * np.float32(3.0)
        prm.grad = t_(g.copy())
        tn = torch.nn.utils.clip_grad_norm_([prm], 1.0)
        out[f"total_norm_{it}"] = np.array(tn.item())
        for grp in opt.param_groups:
            grp["lr"] = 1e-3 * (it + 1) / 3
        opt.step()
        out[f"param_{it}"] = prm.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "optim.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = import_reference()
    gens = dict(dft=gen_dft, spectral=gen_spectral, sde=gen_sde, score=gen_score, ckpt=gen_ckpt, loss=gen_loss, sampler=gen_sampler,
                dataset=gen_dataset, optim=gen_optim, grad_default=gen_grad_default, backbones=gen_backbones)
    for name in (sys.argv[1:] or list(gens)):                  # `make_golden.py spectral` regenerates one file
        gens[name](R)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
