"""The two launch shapes bench.py times that no oracle comparison ran before (VERDICT r3 "what's weak" 1):

  (a) training at B = 64 -- T=252 C=6 (16 128 tokens = 504 blocks) and T=100 C=12 (6 400 tokens = 200 blocks): k_tr_wgrad runs
      ts = 20 token splits there (every other gradient test runs ts <= 16), i.e. the 20-way fixed-order reduce of k_tr_reduce
      and the per-split block ranges blk0..blk1 of 504 / 200 blocks.  Default model (d_model 72, 10 layers), dropout off, same
      injected t and z: bf16 gradients against the exact-f32 engine per tensor (the f32 engine is anchored to the REFERENCE's
      autograd fixture at B = 2 / 4 in test_gpu_train_bf16.py), at that file's bf16 tolerances; the plan hook asserts ts == 20;
      and the step is bit-reproducible at that shape (utils/losses.py:39-125 of the reference is the loss being differentiated).
  (b) T=1024 C=16 at B = 64 (configs[4]'s bench row): series -> XCD placement, 12 workgroups of a series sharing an L2, two
      workgroups per CU -- none of which exists at the B <= 2 the oracle comparisons ran.  Forward: rows {0, 31, 63} against the
      float64 oracle, all 64 rows bf16 against the exact-f32 engine; then a 4-step injected-noise trajectory of the whole
      batch, the same rows against the oracle's loop (sampler.py:83-104).  Tolerances: those of test_gpu_baseline_shapes.py
      (forward <= 2e-2 of scale, rms <= 1e-2) and test_gpu_sampler_parity_shapes.py (trajectory <= 1e-2, rms <= 5e-3).
"""
import numpy as np
import pytest
import torch

from oracle import fdiff_oracle as O
from oracle import weights as W

from .gpu_util import DEV, dev, host, make_model, oracle_sde, report_err
from .test_gpu_train_bf16 import _compare_grads, _grads_of, _log, batch_of

pytestmark = pytest.mark.gpu

TRAIN_SHAPES = {"nasdaq": dict(T=252, C=6, D=72, L=10, H=12), "ecg": dict(T=100, C=12, D=72, L=10, H=12)}
LONG = dict(T=1024, C=16, D=72, L=10, H=12)


@pytest.mark.parametrize("name", sorted(TRAIN_SHAPES))
def test_training_gradients_at_the_benched_batch_vs_exact_f32_engine_reference_anchored_at_B_le_6(name):
    """TRANSITIVE parity: bf16 kernels against the exact-f32 ENGINE at B = 64; the exact-f32 engine itself is held to the
    reference's autograd fixtures at B = 2 ... 6 only (test_gpu_train_bf16.py / test_gpu_score_bwd.py)."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfg, B = TRAIN_SHAPES[name], 64
    X = W.randn(f"bs_x_{name}", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn(f"bs_z_{name}", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform(f"bs_t_{name}", (B,), 3, 0.05, 1.0)
    res = {}
    for prec in ("fp32", "bf16"):
        m, sch, _ = make_model(cfg, precision=prec)
        m.dropout = 0.0
        fn = get_sde_loss_fn(sch, train=True)
        runs = []
        for rep in range(2 if prec == "bf16" else 1):
            m.zero_grad()
            loss = fn(m, batch_of(X, t), noise=dev(z)).item()
            runs.append((loss, m.grads.clone()))
        assert m.train_mode_effective == prec
        desc, ts = m.train_plan(B)
        if prec == "bf16":
            assert ts == 20 and "TS=20" in desc, desc            # the shape bench.py --mode train runs
            # ... with the forward of all layers as ONE persistent launch, 256 workgroups = one per CU at both shapes (round 6)
            assert ("k_tr_fwd_layers NT=4, 4 x 64 workgroups" if name == "nasdaq" else "k_tr_fwd_layers NT=2, 4 x 64 workgroups") in desc, desc
            assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]), "bf16 step not bit-reproducible at ts = 20"
            _log(f"[parity] benched training shape {name} B={B}: {desc}; two runs bit-identical")
        else:
            assert ts == 0
        res[prec] = (runs[0][0], _grads_of(m))
    lf, lb = res["fp32"][0], res["bf16"][0]
    assert abs(lb - lf) <= 1e-2 * abs(lf), (lb, lf)
    _compare_grads(f"benched shape {name} B={B} ts=20 (bf16 vs exact-f32 engine, dropout off)", res["bf16"][1], res["fp32"][1])


def _fwd(m, X, t):
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    m.eval()
    return host(m(DiffusableBatch(X=dev(X), timesteps=dev(t))))


def test_long_forward_at_the_benched_batch():
    cfg, B, rows = LONG, 64, [0, 31, 63]
    X = W.randn("bs_long_x", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform("bs_long_t", (B,), 3, 0.05, 1.0)
    mb, _, sd = make_model(cfg, precision="bf16")
    desc, _ = mb.plan(B)
    assert "k_mega" not in desc and "k_attention_bf16" in desc, desc
    got = _fwd(mb, X, t)
    ref = O.score_forward(sd, X[rows], t[rows], cfg["H"])
    err, rms = report_err(f"forward bf16 long T=1024 C=16 B={B} rows={rows} vs oracle ({desc[:60]})", got[rows], ref)
    assert err <= 2e-2 and rms <= 1e-2, (err, rms)
    for i, r in enumerate(rows):
        e = np.abs(got[r] - ref[i]).max() / np.abs(ref).max()
        assert e <= 2e-2, (r, e)
    mf, _, _ = make_model(cfg, precision="fp32")
    f32 = _fwd(mf, X, t)
    e32, _ = report_err(f"forward fp32 engine long B={B} rows={rows} vs oracle", f32[rows], ref)
    assert np.abs(f32[rows] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), e32        # (test_gpu_baseline_shapes.py: 2e-5 at T=1024)
    err_all, rms_all = report_err(f"forward bf16 vs exact-f32 engine, long, all {B} rows", got, f32)
    assert err_all <= 2e-2 and rms_all <= 1e-2, (err_all, rms_all)
    per_row = np.abs(got - f32).reshape(B, -1).max(axis=1) / np.abs(f32).max()
    assert per_row.max() <= 2e-2, (int(per_row.argmax()), float(per_row.max()))


def test_long_trajectory_at_the_benched_batch():
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    cfg, B, N, rows = LONG, 64, 4, [0, 31, 63]
    kind, p = "vp", (0.1, 20.0)
    m, sch, sd = make_model(cfg, kind=kind, p=p, precision="bf16")
    g = torch.Generator(device="cpu").manual_seed(41)
    shape = (B, cfg["T"], cfg["C"])
    zp = torch.randn(shape, generator=g)
    zs = torch.randn((N,) + shape, generator=g)
    smp = DiffusionSampler(score_model=m, sample_batch_size=B)
    got = smp.sample(num_samples=B, num_diffusion_steps=N, prior_noise=[zp.to(DEV)], step_noise=[zs.to(DEV)]).numpy().astype(np.float64)
    ref, _ = O.sample_trajectory(sd, oracle_sde(kind, p, True, cfg["T"]), zp.numpy()[rows].astype(np.float64),
                                 [z[rows].astype(np.float64) for z in zs.numpy()], cfg["H"])
    err, rms = report_err(f"{N}-step trajectory bf16 long T=1024 C=16 B={B} rows={rows} vs oracle", got[rows], ref)
    assert err <= 1e-2 and rms <= 5e-3, (err, rms)
    assert np.isfinite(got).all()
