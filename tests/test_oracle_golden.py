"""Pin the oracle (oracle/fdiff_oracle.py) against golden vectors produced by the reference itself
(oracle/make_golden.py -> tests/golden/*.npz).  CPU only."""
import math

import numpy as np
import pytest

from oracle import fdiff_oracle as O
from oracle import weights as W
from oracle.make_golden import (CFG_DEFAULT, CFG_ODD, CFG_TINY, DFT_B, DFT_C, DFT_T, SDE_CASES)

CFGS = {"default": CFG_DEFAULT, "tiny": CFG_TINY, "odd": CFG_ODD}


def make_sde(kind, p, scaling, T):
    return O.SDEParams(kind, p[0], p[1], O.noise_scaling(T, scaling))


@pytest.mark.parametrize("T", DFT_T)
@pytest.mark.parametrize("C", DFT_C)
def test_dft_idft_vs_reference(golden, T, C):
    g = golden("dft")
    x = W.randn(f"dft_x_{T}_{C}", (DFT_B, T, C), 0)
    # reference's own tolerance for this transform is 1e-5 (tests/test_utils.py:44-51)
    np.testing.assert_allclose(O.dft(x), g[f"dft_{T}_{C}"], atol=1e-5, rtol=0)
    xt = W.randn(f"idft_x_{T}_{C}", (DFT_B, T, C), 0)
    np.testing.assert_allclose(O.idft(xt), g[f"idft_{T}_{C}"], atol=1e-5, rtol=0)


def test_dft_known_answers(golden):
    g = golden("dft")
    for T in (16, 15):
        imp = np.zeros((1, T, 1))
        imp[0, 3, 0] = 1.0
        np.testing.assert_allclose(O.dft(imp), g[f"dft_impulse_{T}"], atol=2e-7)
        np.testing.assert_allclose(O.dft_by_definition(imp), g[f"dft_impulse_{T}"], atol=2e-7)
        n = np.arange(T)
        cosw = np.cos(2 * np.pi * 2 * n / T).reshape(1, T, 1)
        np.testing.assert_allclose(O.dft(cosw), g[f"dft_cos2_{T}"], atol=1e-6)
        # analytic: a pure cosine at bin 2 has Re X_2 = sqrt(T)/2 and nothing else
        expect = np.zeros((1, T, 1))
        expect[0, 2, 0] = math.sqrt(T) / 2
        np.testing.assert_allclose(O.dft(cosw), expect, atol=1e-12)


@pytest.mark.parametrize("T", [16, 24, 101, 187])
def test_dft_definition_matches_fft(T):
    x = W.randn("defn", (2, T, 3), 9)
    np.testing.assert_allclose(O.dft_by_definition(x), O.dft(x), atol=1e-10)
    np.testing.assert_allclose(O.idft(O.dft(x)), x, atol=1e-10)


def test_noise_scaling_and_timesteps(golden):
    g = golden("sde")
    for T in (100, 101):
        for s in (False, True):
            np.testing.assert_array_equal(O.noise_scaling(T, s), g[f"G_{T}_{int(s)}"])
    for N in (10, 1000, 2000):
        ts, dt = O.timesteps(N)
        # ATen's vectorised linspace rounds differently per SIMD width (AVX2/AVX-512 blocks restart
        # the recurrence), so the reference grid is only reproducible to 1 ulp across hosts.
        np.testing.assert_allclose(ts, g[f"timesteps_{N}"], rtol=2.5e-7, atol=1e-12)
        np.testing.assert_allclose(dt, g[f"step_size_{N}"], rtol=2e-4)


def test_marginal_step_prior(golden):
    g = golden("sde")
    B, T, C = 4, 20, 3
    x = W.randn("sde_x", (B, T, C), 1)
    score = W.randn("sde_score", (B, T, C), 1)
    z = W.randn("sde_z", (B, T, C), 1)
    tvals = np.array([1e-5, 0.1, 0.5, 1.0], np.float32)
    _, dt = O.timesteps(1000)
    for ci, (kind, p) in enumerate(SDE_CASES):
        for scaling in (False, True):
            tag = f"{kind}{ci}_{int(scaling)}"
            sde = make_sde(kind, p, scaling, T)
            mean, std = O.marginal_prob(sde, x, tvals)
            np.testing.assert_allclose(mean, g[f"mean_{tag}"], rtol=2e-6, atol=1e-7)
            # std at t=1e-5 (VP) is sqrt(1-exp(-1e-6..)) computed in fp32 by the reference: loose rtol there
            np.testing.assert_allclose(std[1:], g[f"std_{tag}"][1:], rtol=2e-6, atol=1e-9)
            np.testing.assert_allclose(std[0], g[f"std_{tag}"][0], rtol=5e-2 if kind == "vp" else 2e-6)
            for ti, tv in enumerate((0.37, 1e-5, 1.0)):
                o = O.sde_step(sde, score, tv, x, z, float(dt))
                np.testing.assert_allclose(o, g[f"step_{tag}_{ti}"], rtol=2e-6, atol=2e-6)
            np.testing.assert_allclose(O.prior_sampling(sde, z), g[f"prior_{tag}"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name,B", [("default", 4), ("tiny", 3), ("odd", 3)])
def test_score_forward_vs_reference(golden, name, B):
    g = golden("score_forward")
    cfg = CFGS[name]
    sd = W.make_state_dict(cfg["C"], cfg["T"], cfg["D"], cfg["L"], seed=1234)
    X = W.randn(f"score_x_{name}", (B, cfg["T"], cfg["C"]), 2)
    t = W.uniform(f"score_t_{name}", (B,), 2, 1e-5, 1.0)
    out = O.score_forward(sd, X, t, cfg["H"])
    # 5e-6 abs at O(1) outputs; torch's own two paths differ by ~1e-6
    np.testing.assert_allclose(out, g[f"fast_{name}"], atol=5e-6, rtol=0)
    np.testing.assert_allclose(out, g[f"slow_{name}"], atol=5e-6, rtol=0)


@pytest.mark.parametrize("name,B", [("tiny", 5), ("odd", 3)])
def test_loss_vs_reference(golden, name, B):
    g = golden("loss")
    cfg = CFGS[name]
    sd = W.make_state_dict(cfg["C"], cfg["T"], cfg["D"], cfg["L"], seed=1234)
    X = W.randn(f"loss_x_{name}", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn(f"loss_z_{name}", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform(f"loss_t_{name}", (B,), 3, 0.05, 1.0)
    for ci, (kind, p) in enumerate(SDE_CASES[:2]):
        sde = make_sde(kind, p, True, cfg["T"])
        for lw in (False, True):
            tag = f"{name}_{kind}{ci}_{int(lw)}"
            val = O.loss_fn(sd, sde, X, t, z, cfg["H"], likelihood_weighting=lw)
            np.testing.assert_allclose(val, g[f"loss_{tag}"], rtol=2e-5)
            np.testing.assert_allclose(val, g[f"loss_train_{tag}"], rtol=2e-5)   # dropout forced to 0


@pytest.mark.parametrize("name,B", [("tiny", 6), ("default", 2)])
def test_sampler_trajectory_vs_reference(golden, name, B):
    g = golden("sampler")
    cfg = CFGS[name]
    sd = W.make_state_dict(cfg["C"], cfg["T"], cfg["D"], cfg["L"], seed=1234)
    shape = (B, cfg["T"], cfg["C"])
    zp = W.randn(f"samp_prior_{name}", shape, 4)
    zs = [W.randn(f"samp_z_{name}_{i}", shape, 4) for i in range(20)]
    for ci, (kind, p) in enumerate(SDE_CASES[:2]):
        sde = make_sde(kind, p, True, cfg["T"])
        Xf, rec = O.sample_trajectory(sd, sde, zp, zs, cfg["H"], record=(1, 5, 20))
        tag = f"{name}_{kind}{ci}"
        for k in (1, 5, 20):
            ref = g[f"step{k}_{tag}"]
            scale = np.abs(ref).max()
            # 1e-4 abs at O(1); trajectories here reach |X|~400 with 20 coarse steps -> relative
            assert np.abs(rec[k] - ref).max() <= 1e-4 * max(1.0, scale), (tag, k)
        np.testing.assert_allclose(Xf, g[f"final_{tag}"], atol=1e-4 * max(1.0, np.abs(g[f'final_{tag}']).max()))


def test_sampler_batching_rule(golden):
    for ns, bs, n_out in golden("sampler")["batching"]:
        nb, per = O.num_sample_batches(int(ns), int(bs))
        assert nb * per == n_out


def test_dataset_statistics(golden):
    g = golden("dataset")
    X = W.randn("ds_x", (16, 24, 3), 5)
    Xs, mean, std = O.dataset_standardize(X, fourier_transform=True)
    np.testing.assert_allclose(mean, g["mean"], atol=1e-6)
    np.testing.assert_allclose(std, g["std"], rtol=1e-5)
    np.testing.assert_allclose(Xs[3], g["item3"], atol=1e-5)


def test_adamw_and_clip(golden):
    g = golden("optim")
    p = W.randn("opt_p", (257,), 6).astype(np.float64)
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for it in range(3):
        grad = (W.randn(f"opt_g{it}", (257,), 6) * np.float32(3.0)).astype(np.float64)
        tn = float(np.sqrt((grad**2).sum()))
        np.testing.assert_allclose(tn, g[f"total_norm_{it}"], rtol=1e-6)
        grad = grad * O.clip_grad_norm_scale(tn, 1.0)
        p, m, v = O.adamw_step(p, grad, m, v, it + 1, 1e-3 * (it + 1) / 3)
        np.testing.assert_allclose(p, g[f"param_{it}"], rtol=1e-6, atol=1e-7)


def test_cosine_warmup_formula():
    # formula-pinned (diffusers absent): SURVEY A.6
    W_, S = 10, 100
    assert O.cosine_warmup_factor(0, W_, S) == 0.0
    assert O.cosine_warmup_factor(5, W_, S) == 0.5
    assert O.cosine_warmup_factor(10, W_, S) == 1.0
    assert abs(O.cosine_warmup_factor(55, W_, S) - 0.5) < 1e-12
    assert O.cosine_warmup_factor(100, W_, S) < 1e-12


# ---------------------------------------------------------------- spectral utilities (fourier.py:90-209)
SPEC_T, SPEC_C = (16, 100, 101, 187), (1, 12)


def spectral_input(T, C):
    x = W.randn(f"spec_x_{T}_{C}", (3, T, C), 0)
    x[1, T // 3: T // 3 + 4] += 3.0
    return x


@pytest.mark.parametrize("T", SPEC_T)
@pytest.mark.parametrize("C", SPEC_C)
def test_spectral_utilities_vs_reference(golden, T, C):
    g = golden("spectral")
    x = spectral_input(T, C)
    np.testing.assert_allclose(O.spectral_density(x), g[f"dens_{T}_{C}"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(O.spectral_density(x, apply_dft=False), g[f"dens_nodft_{T}_{C}"], rtol=1e-5, atol=1e-6)
    loc, sloc = O.localization_metrics(x)
    np.testing.assert_allclose(np.stack([loc, sloc]), g[f"loc_{T}_{C}"], rtol=1e-4)
    assert loc[1] < loc[0]                                   # the series with the bump is more localised in time
    if T % 2 == 1:
        for sigma in (1.0, 4.5):
            np.testing.assert_allclose(O.smooth_frequency(x, sigma), g[f"smooth_{T}_{C}_{sigma}"], atol=2e-5)
    else:
        assert int(g["smooth_even_raises"]) == 1             # the reference itself fails for even lengths
        with pytest.raises(RuntimeError):
            O.smooth_frequency(x, 1.0)


def test_spectral_density_is_parseval():
    """sum of the two-sided density = energy of the series (ortho norm)."""
    x = W.randn("pars", (2, 101, 3), 4)
    d = O.spectral_density(x)
    two_sided = d.sum(axis=1) + d[:, 1:].sum(axis=1)
    np.testing.assert_allclose(two_sided, (x.astype(np.float64) ** 2).sum(axis=1), rtol=1e-10)


# ---------------------------------------------------------------- Wasserstein metrics (wasserstein.py, metrics.py)
# POT is absent here and unpinned in the reference: the 1-D transport restatement is pinned against the transport LP.
@pytest.mark.parametrize("n,m", [(3, 3), (4, 2), (5, 3), (1, 6), (6, 4), (7, 7)])
def test_emd2_1d_is_the_optimal_transport_cost(n, m):
    from scipy.optimize import linprog
    rng = np.random.default_rng(n * 10 + m)
    a, b = rng.normal(size=n), rng.normal(size=m) + 0.3
    cost = (a[:, None] - b[None, :]) ** 2
    rows, rhs = [], []
    for i in range(n):
        r = np.zeros((n, m)); r[i] = 1; rows.append(r.ravel()); rhs.append(1 / n)
    for j in range(m):
        r = np.zeros((n, m)); r[:, j] = 1; rows.append(r.ravel()); rhs.append(1 / m)
    lp = linprog(cost.ravel(), A_eq=np.array(rows), b_eq=np.array(rhs), bounds=(0, None))
    assert lp.status == 0
    np.testing.assert_allclose(O.emd2_1d(a, b), lp.fun, rtol=1e-9, atol=1e-12)


def test_emd2_1d_closed_forms():
    rng = np.random.default_rng(0)
    a = rng.normal(size=50)
    assert O.emd2_1d(a, a) == 0.0
    np.testing.assert_allclose(O.emd2_1d(a, a + 2.0), 4.0, rtol=1e-12)                 # a shift moves every quantile by 2
    b = rng.normal(size=50)
    np.testing.assert_allclose(O.emd2_1d(a, b), np.mean((np.sort(a) - np.sort(b)) ** 2), rtol=1e-12)   # equal sizes
    np.testing.assert_allclose(O.emd2_1d(a, [0.7]), np.mean((a - 0.7) ** 2), rtol=1e-12)              # a point mass


def test_wasserstein_metric_dicts():
    X = W.randn("wm_x", (40, 6, 2), 1)
    Y = W.randn("wm_y", (25, 6, 2), 2) * 1.5 + 0.2
    dirs = O.random_directions(7, 12, 5)
    np.testing.assert_allclose(np.linalg.norm(dirs, axis=1), 1.0, rtol=1e-12)
    np.testing.assert_array_equal(dirs, O.random_directions(7, 12, 5))                   # same seed, same directions
    s = O.sliced_wasserstein_metric(X, Y, seed=7, num_directions=5)
    assert sorted(s) == ["sliced_wasserstein_all", "sliced_wasserstein_max", "sliced_wasserstein_max_dummy",
                         "sliced_wasserstein_max_self", "sliced_wasserstein_mean", "sliced_wasserstein_mean_dummy",
                         "sliced_wasserstein_mean_self"]
    assert len(s["sliced_wasserstein_all"]) == 5 and s["sliced_wasserstein_max"] >= s["sliced_wasserstein_mean"] > 0
    mg = O.marginal_wasserstein_metric(X, Y)
    assert len(mg["marginal_wasserstein_all"]) == 12
    # marginal distance of feature f = sliced distance along e_f
    f = 3
    e = np.zeros(12); e[f] = 1
    Xf, Yf = O.check_flat_array(X), O.check_flat_array(Y)
    np.testing.assert_allclose(mg["marginal_wasserstein_all"][f], math.sqrt(O.emd2_1d(Xf @ e, Yf @ e)), rtol=1e-12)


def test_backbone_oracles_match_the_reference_modules(golden):
    """the numpy restatements of MLPScoreModule / LSTMScoreModule.forward against outputs of the reference's
    own classes (tests/golden/backbones.npz; the MLP blocks over the documented stand-in for the absent torchvision)."""
    from oracle import fdiff_oracle as O
    from oracle import weights as W
    from oracle.make_golden import CFG_BB
    g = golden("backbones")
    for kind, f in (("mlp", O.mlp_score_forward), ("lstm", O.lstm_score_forward)):
        for name, cfg, B in (("small", CFG_BB, 4), ("wide", dict(T=50, C=4, D=72, L=3), 3)):
            d_mlp = 64 if name == "small" else 1024
            sd = W.make_state_dict_backbone(kind, cfg["C"], cfg["T"], cfg["D"], cfg["L"], d_mlp=d_mlp, seed=4321)
            X = W.randn(f"bb_x_{kind}_{name}", (B, cfg["T"], cfg["C"]), 5)
            t = W.uniform(f"bb_t_{kind}_{name}", (B,), 5, 0.05, 1.0)
            np.testing.assert_allclose(f(sd, X, t), g[f"fwd_{kind}_{name}"], atol=5e-6, rtol=0)


def test_philox4x32_10_known_answers():
    """The oracle's restatement of the generator against the published known-answer vectors of Philox4x32-10 (Random123
    kat_vectors: counter, key -> output); the GPU test then holds the engine's device code to this restatement."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = O.philox4x32_10(np.array([ctr], dtype=np.uint64), np.array([key], dtype=np.uint64))[0]
        assert tuple(int(v) for v in got) == want, (ctr, key, [hex(int(v)) for v in got])


def test_dropout_decision_rule_marginals_and_ties():
    """fd_drop16's rule on the oracle side: 16 decisions per evaluation, keep probability 1 - thr16 / 65536 (p = 0.1: 0.899994),
    a byte other than thr16 >> 8 settles its decision alone, and neighbouring decisions are uncorrelated to 1e-3."""
    words = O.engine_philox_words(seed=0x1234567890ABCDEF, offset=7, n=1 << 16)
    dec = O.dropout_decisions16(words, 0.1)
    bits = ((dec[:, None].astype(np.uint32) >> np.arange(16)) & 1).astype(np.float64)          # (n, 16)
    n = bits.size
    keep = bits.mean()
    assert abs(keep - (1 - 6554 / 65536)) < 4 * np.sqrt(0.09 / n), keep
    raw = np.ascontiguousarray(words.astype("<u4")).view(np.uint8).reshape(-1, 16)
    high = np.roll(raw, -1, axis=1)                                         # window e = byte e | byte e+1 << 8: its high byte
    clear = high != (6554 >> 8)
    assert np.array_equal(bits[clear] > 0, high[clear] > (6554 >> 8))       # the high byte alone decides away from the tie value
    a, b = bits[:, :-1].ravel(), bits[:, 1:].ravel()
    rho = np.corrcoef(a, b)[0, 1]
    assert abs(rho) < 5e-3, rho

