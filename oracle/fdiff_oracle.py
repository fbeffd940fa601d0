"""numpy restatement of the reference's score-matching hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Every function cites
the reference file:line (relative to /root/reference) whose arithmetic it
restates.  Default working precision is float64 so that the oracle is the
"truth" both the reference's fp32 CPU path and the HIP fp32 path are compared
against; places where the reference's fp32 rounding is *semantically visible*
(timestep grid, time-embedding phase) are reproduced in float32 explicitly.

Parity: pinned by tests/golden/*.npz (generated from the reference by
oracle/make_golden.py, checked by tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

Array = np.ndarray


# --------------------------------------------------------------------------
# a1/a2  spectral representation          src/fdiff/utils/fourier.py:8-87
# --------------------------------------------------------------------------
def dft(x: Array) -> Array:
    """fourier.py:8-45 -- ortho rFFT along axis 1, packed [Re X_0..X_{T//2} ; Im X_1..X_K]."""
    x = np.asarray(x)
    T = x.shape[1]
    X = np.fft.rfft(x.astype(np.float64), axis=1, norm="ortho")
    re = X.real
    im = X.imag[:, 1:]                       # fourier.py:30 (DC imaginary dropped)
    if T % 2 == 0:
        im = im[:, :-1]                      # fourier.py:33-37 (Nyquist imaginary dropped)
    out = np.concatenate([re, im], axis=1)   # fourier.py:40
    assert out.shape == x.shape
    return out


def idft(x: Array) -> Array:
    """fourier.py:48-87 -- unpack to a half spectrum (zero Im at DC / Nyquist), ortho irfft(n=T)."""
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[1]
    n_real = math.ceil((T + 1) / 2)          # fourier.py:59
    re = x[:, :n_real]
    im = x[:, n_real:]
    zero = np.zeros((x.shape[0], 1, x.shape[2]))
    im = np.concatenate([zero, im], axis=1)  # fourier.py:66-67
    if T % 2 == 0:
        im = np.concatenate([im, zero], axis=1)  # fourier.py:70-71
    assert im.shape == re.shape
    return np.fft.irfft(re + 1j * im, n=T, axis=1, norm="ortho")  # fourier.py:80


def dft_by_definition(x: Array) -> Array:
    """O(T^2) evaluation of SURVEY A.1, independent of any FFT library (small cases only)."""
    x = np.asarray(x, dtype=np.float64)
    B, T, C = x.shape
    n = np.arange(T)
    k = np.arange(T // 2 + 1)
    ang = -2.0 * np.pi * np.outer(k, n) / T
    re = np.einsum("kn,bnc->bkc", np.cos(ang), x) / math.sqrt(T)
    im = np.einsum("kn,bnc->bkc", np.sin(ang), x) / math.sqrt(T)
    K = T // 2 - 1 if T % 2 == 0 else (T - 1) // 2
    return np.concatenate([re, im[:, 1 : 1 + K]], axis=1)


# --------------------------------------------------------------------------
# 8(f)2  spectral utilities of the dataset front-end   src/fdiff/utils/fourier.py:90-209
# --------------------------------------------------------------------------
def spectral_density(x: Array, apply_dft: bool = True) -> Array:
    """fourier.py:90-124 -- Re^2 + Im^2 per bin k = 0..T//2 of the packed representation."""
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[1]
    xt = dft(x) if apply_dft else x
    n_real = math.ceil((T + 1) / 2)          # fourier.py:104
    re = xt[:, :n_real]
    zero = np.zeros((x.shape[0], 1, x.shape[2]))
    im = np.concatenate([zero, xt[:, n_real:]], axis=1)      # fourier.py:109-110
    if T % 2 == 0:
        im = np.concatenate([im, zero], axis=1)              # fourier.py:113-114
    assert im.shape == re.shape
    return re**2 + im**2                                     # fourier.py:121


def cyclic_distance_sq(T: int) -> Array:
    """fourier.py:160-163 -- min(|t - s|, T - |t - s|)^2."""
    t = np.arange(T, dtype=np.float64)
    d = np.abs(t[:, None] - t[None, :])
    return np.minimum(d, T - d) ** 2


def localization_metrics(X: Array):
    """fourier.py:127-175 -- (delocalisation in time, delocalisation in frequency), one value per series."""
    X = np.asarray(X, dtype=np.float64)
    T = X.shape[1]
    e_t = (X**2).sum(axis=2) / (X**2).sum(axis=(1, 2))[:, None]              # fourier.py:141-144
    spec = spectral_density(X)                                               # fourier.py:147
    mirror = spec[:, 1:][:, ::-1] if T % 2 != 0 else spec[:, 1:-1][:, ::-1]  # fourier.py:148-152
    spec = np.concatenate([spec, mirror], axis=1)
    assert spec.shape[1] == T
    e_s = spec.sum(axis=2) / spec.sum(axis=(1, 2))[:, None]                  # fourier.py:154-157
    d2 = cyclic_distance_sq(T)
    return (e_t @ d2).min(axis=1), (e_s @ d2).min(axis=1)                    # fourier.py:166-173


def gaussian_frequency_kernel(T: int, sigma: float) -> Array:
    """fourier.py:189-200 -- column-normalised Gaussian over the frequencies of the packed rows (odd T only: for even
    T the reference's vector has T - 1 entries and its einsum raises)."""
    if T % 2 == 0:
        raise RuntimeError("smooth_frequency is only defined for odd max_len (fourier.py:192-203)")
    nyq = T / 2
    k = np.concatenate([np.arange(0, nyq), np.arange(1, nyq)]).astype(np.float64)
    g = np.exp(-(((k[:, None] - k[None, :]) / sigma) ** 2) / 2)
    return g / g.sum(axis=0, keepdims=True)


def smooth_frequency(X: Array, sigma: float) -> Array:
    """fourier.py:178-209 -- idft(einsum("btc,ts->bsc", dft(X), gaussian kernel))."""
    X = np.asarray(X, dtype=np.float64)
    g = gaussian_frequency_kernel(X.shape[1], sigma)
    return idft(np.einsum("btc,ts->bsc", dft(X), g))


# --------------------------------------------------------------------------
# 8(f)3  Wasserstein evaluation metrics   src/fdiff/utils/wasserstein.py:12-199, src/fdiff/sampling/metrics.py:100-217
#
# The 1-D transport itself lives in POT (`ot.emd2_1d`; `pot` is unpinned in the reference's pyproject.toml and absent from
# this image): PARITY UNPINNED against POT.  Restated from its published algorithm (sort both samples, move mass greedily
# along the two quantile functions; uniform weights, squared Euclidean cost) and pinned instead against the transport LP
# solved exactly by scipy on small cases and against closed forms (tests/test_oracle_golden.py).
# --------------------------------------------------------------------------
def emd2_1d(a: Array, b: Array) -> float:
    """W2^2 between the empirical measures of a (n values) and b (m values), uniform weights."""
    a = np.sort(np.asarray(a, dtype=np.float64).ravel())
    b = np.sort(np.asarray(b, dtype=np.float64).ravel())
    n, m = len(a), len(b)
    i = j = 0
    wa, wb = 1.0 / n, 1.0 / m           # mass left on a[i], b[j]
    cost = 0.0
    while i < n and j < m:
        mv = min(wa, wb)
        cost += mv * (a[i] - b[j]) ** 2
        wa -= mv
        wb -= mv
        if wa <= 1e-15:
            i += 1
            wa = 1.0 / n
        if wb <= 1e-15:
            j += 1
            wb = 1.0 / m
    return cost


def check_flat_array(x) -> Array:
    """utils/tensors.py:5-23 -- (n, ...) -> (n, d)."""
    x = np.asarray(x)
    return x.reshape(x.shape[0], -1) if x.ndim > 2 else x


def random_directions(seed, dim: int, num_directions: int) -> Array:
    """wasserstein.py:38-74 -- unit vectors from ONE numpy Generator, drawn direction after direction."""
    rng = np.random.default_rng(seed)
    out = np.empty((num_directions, dim))
    for k in range(num_directions):
        v = rng.normal(size=dim)
        out[k] = v / np.linalg.norm(v)
    return out


def sliced_distances(original: Array, other: Array, seed, num_directions: int) -> Array:
    """wasserstein.py:163-181 with directional_distance :117-143 (normalisation 'none')."""
    original, other = check_flat_array(original), check_flat_array(other)
    dirs = random_directions(seed, original.shape[1], num_directions)
    return np.array([math.sqrt(emd2_1d(original @ d, other @ d)) for d in dirs])


def marginal_distances(original: Array, other: Array) -> Array:
    """wasserstein.py:183-199 with feature_distance :91-115."""
    original, other = check_flat_array(original), check_flat_array(other)
    return np.array([math.sqrt(emd2_1d(original[:, f], other[:, f])) for f in range(original.shape[1])])


def _wasserstein_metric(kind: str, original: Array, other: Array, seed, num_directions, baselines: bool) -> dict:
    """metrics.py:100-217 -- the dict a SlicedWasserstein / MarginalWasserstein metric contributes (with its baselines)."""
    original, other = check_flat_array(original), check_flat_array(other)
    dist = (lambda a, b: sliced_distances(a, b, seed, num_directions)) if kind == "sliced" else marginal_distances
    d = dist(original, other)
    out = {f"{kind}_wasserstein_mean": float(d.mean()), f"{kind}_wasserstein_max": float(d.max()),
           f"{kind}_wasserstein_all": d.tolist()}
    if baselines:
        n = original.shape[0]
        ds = dist(original[: n // 2], original[n // 2:])                 # metrics.py:131-137 / 189-195
        dd = dist(original, original.mean(axis=0, keepdims=True))         # metrics.py:140-146 / 198-204
        out.update({f"{kind}_wasserstein_mean_self": float(ds.mean()), f"{kind}_wasserstein_max_self": float(ds.max()),
                    f"{kind}_wasserstein_mean_dummy": float(dd.mean()), f"{kind}_wasserstein_max_dummy": float(dd.max())})
    return out


def sliced_wasserstein_metric(original, other, seed, num_directions, baselines=True) -> dict:
    return _wasserstein_metric("sliced", original, other, seed, num_directions, baselines)


def marginal_wasserstein_metric(original, other, baselines=True) -> dict:
    return _wasserstein_metric("marginal", original, other, None, None, baselines)


# --------------------------------------------------------------------------
# a3/a4  noise scaling and timestep grid  src/fdiff/schedulers/sde.py:42-64
# --------------------------------------------------------------------------
def noise_scaling(max_len: int, fourier_noise_scaling: bool) -> Array:
    """sde.py:42-60 -- G (float32, as the reference stores it)."""
    G = np.ones(max_len, dtype=np.float32)
    if fourier_noise_scaling:
        G = (np.float32(1.0 / math.sqrt(2.0)) * G).astype(np.float32)
        G[0] *= np.float32(math.sqrt(2.0))
        if max_len % 2 == 0:
            G[max_len // 2] *= np.float32(math.sqrt(2.0))
    return G


def linspace_f32(start: float, end: float, steps: int) -> Array:
    """torch.linspace(start, end, steps) in float32 (sde.py:63).

    ATen computes step=(end-start)/(steps-1) in float32 and fills symmetrically:
    i < steps//2 -> start + step*i, else end - step*(steps-1-i).
    """
    start32, end32 = np.float32(start), np.float32(end)
    if steps == 1:
        return np.array([start32], dtype=np.float32)
    step = np.float32((end32 - start32) / np.float32(steps - 1))
    i = np.arange(steps)
    lo = (start32 + step * i.astype(np.float32)).astype(np.float32)
    hi = (end32 - step * (steps - 1 - i).astype(np.float32)).astype(np.float32)
    return np.where(i < steps // 2, lo, hi).astype(np.float32)


def timesteps(num_diffusion_steps: int, eps: float = 1e-5) -> Tuple[Array, np.float32]:
    """sde.py:62-64 -- (timesteps float32, step_size float32 = t[0]-t[1])."""
    ts = linspace_f32(1.0, eps, num_diffusion_steps)
    return ts, np.float32(ts[0] - ts[1])


# --------------------------------------------------------------------------
# a5..a8  SDE                             src/fdiff/schedulers/sde.py:66-246
# --------------------------------------------------------------------------
class SDEParams:
    """kind: 'vp' (beta_min, beta_max) or 've' (sigma_min, sigma_max)."""

    def __init__(self, kind: str, p0: float, p1: float, G: Array, eps: float = 1e-5):
        assert kind in ("vp", "ve")
        self.kind, self.p0, self.p1, self.eps = kind, float(p0), float(p1), eps
        self.G = np.asarray(G, dtype=np.float64)


def marginal_prob(sde: SDEParams, x: Array, t: Array) -> Tuple[Array, Array]:
    """VP sde.py:187-210 ; VE sde.py:108-123.  Returns mean (B,T,C), std (B,T)."""
    x = np.asarray(x, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    if sde.kind == "vp":
        lmc = -0.25 * t**2 * (sde.p1 - sde.p0) - 0.5 * t * sde.p0      # sde.py:195-197
        mean = np.exp(lmc)[:, None, None] * x                           # sde.py:199-201
        std = np.sqrt(1.0 - np.exp(2.0 * lmc))[:, None] * sde.G[None]   # sde.py:203-207
    else:
        std = (sde.p0 * (sde.p1 / sde.p0) ** t)[:, None] * sde.G[None]  # sde.py:117-119
        mean = x
    return mean, std


def prior_sampling(sde: SDEParams, z: Array) -> Array:
    """sde.py:79-87 (G_matrix @ z), VE override sde.py:125-127 (x sigma_max)."""
    out = sde.G[None, :, None] * np.asarray(z, dtype=np.float64)
    if sde.kind == "ve":
        out = sde.p1 * out
    return out


def sde_step(sde: SDEParams, score: Array, t: float, x: Array, z: Array, step_size: float) -> Array:
    """Euler-Maruyama reverse step.  VP sde.py:215-246 ; VE sde.py:129-165.

    diag_embed + (T,T)@(B,T,C) matmuls of the reference are row scalings by g_k.
    """
    score = np.asarray(score, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64)
    dt = float(step_size)
    if sde.kind == "vp":
        beta = sde.p0 + t * (sde.p1 - sde.p0)                 # sde.py:212-213
        g = (math.sqrt(beta) * sde.G)[None, :, None]          # sde.py:229
        drift = -0.5 * beta * x - (g * g) * score             # sde.py:232-234
    else:
        sd = sde.p0 * math.sqrt(2.0 * math.log(sde.p1 / sde.p0)) * (sde.p1 / sde.p0) ** t  # sde.py:143-147
        g = (sd * sde.G)[None, :, None]                       # sde.py:149
        drift = -((g * g) * score)                            # sde.py:152-154
    return x - drift * dt + math.sqrt(dt) * g * z             # sde.py:159-163 / 240-244


# --------------------------------------------------------------------------
# a9  score network (eval)    src/fdiff/models/score_models.py:67-94,
#                             src/fdiff/models/transformer.py:17-29,77-91,
#                             torch nn.TransformerEncoderLayer (post-LN, relu)
# --------------------------------------------------------------------------
def renorm_rows(P: Array, max_norm: float) -> Array:
    """torch embedding_renorm_ (nn.Embedding(max_norm), transformer.py:13-15): rows with
    ||row||_2 > max_norm are scaled by max_norm / (norm + 1e-7)."""
    P = np.asarray(P, dtype=np.float64)
    n = np.linalg.norm(P, axis=1)
    scale = np.where(n > max_norm, max_norm / (n + 1e-7), 1.0)
    return P * scale[:, None]


def layer_norm(x: Array, g: Array, b: Array, eps: float = 1e-5) -> Array:
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def gfp_embedding(t: Array, W: Array, d_model: int) -> Array:
    """transformer.py:80-84.  The phase t*W*2*pi is formed in float32 in the reference
    (values up to ~600 rad, so the float32 rounding of the phase is visible at 1e-5);
    reproduce that rounding, then evaluate sin/cos exactly."""
    t32 = np.asarray(t, dtype=np.float32)
    W32 = np.asarray(W, dtype=np.float32)
    proj = ((t32[:, None] * W32[None, :]) * np.float32(2.0)) * np.float32(np.pi)
    proj = proj.astype(np.float64)
    emb = np.concatenate([np.sin(proj), np.cos(proj)], axis=-1)
    return emb[:, :d_model]


def _f64(p: Dict[str, Array], k: str) -> Array:
    return np.asarray(p[k], dtype=np.float64)


def score_forward(p: Dict[str, Array], X: Array, t: Array, n_head: int,
                  return_hidden: bool = False):
    """ScoreModule.forward (score_models.py:67-94) in eval mode; SURVEY A.3.

    p uses the reference's state_dict key names.
    """
    X = np.asarray(X, dtype=np.float64)
    B, T, C = X.shape
    We, be = _f64(p, "embedder.weight"), _f64(p, "embedder.bias")
    D = We.shape[0]
    hd = D // n_head
    h = X @ We.T + be                                              # score_models.py:78
    pe = renorm_rows(_f64(p, "pos_encoder.embedding.weight"), math.sqrt(D))
    h = h + pe[None, :T]                                           # transformer.py:26-28
    emb = gfp_embedding(t, p["time_encoder.W"], D)
    h = h + (emb @ _f64(p, "time_encoder.dense.weight").T + _f64(p, "time_encoder.dense.bias"))[:, None, :]
    hidden = [h.copy()]
    L = 0
    while f"backbone.layers.{L}.linear1.weight" in p:
        L += 1
    for i in range(L):
        pre = f"backbone.layers.{i}."
        qkv = h @ _f64(p, pre + "self_attn.in_proj_weight").T + _f64(p, pre + "self_attn.in_proj_bias")
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        q = q.reshape(B, T, n_head, hd).transpose(0, 2, 1, 3)
        k = k.reshape(B, T, n_head, hd).transpose(0, 2, 1, 3)
        v = v.reshape(B, T, n_head, hd).transpose(0, 2, 1, 3)
        s = (q @ k.transpose(0, 1, 3, 2)) / math.sqrt(hd)
        s = s - s.max(axis=-1, keepdims=True)
        e = np.exp(s)
        a = (e / e.sum(axis=-1, keepdims=True)) @ v                # (B,H,T,hd)
        a = a.transpose(0, 2, 1, 3).reshape(B, T, D)
        a = a @ _f64(p, pre + "self_attn.out_proj.weight").T + _f64(p, pre + "self_attn.out_proj.bias")
        h = layer_norm(h + a, _f64(p, pre + "norm1.weight"), _f64(p, pre + "norm1.bias"))
        f = np.maximum(h @ _f64(p, pre + "linear1.weight").T + _f64(p, pre + "linear1.bias"), 0.0)
        f = f @ _f64(p, pre + "linear2.weight").T + _f64(p, pre + "linear2.bias")
        h = layer_norm(h + f, _f64(p, pre + "norm2.weight"), _f64(p, pre + "norm2.bias"))
        hidden.append(h.copy())
    out = h @ _f64(p, "unembedder.weight").T + _f64(p, "unembedder.bias")   # score_models.py:90
    if return_hidden:
        return out, hidden
    return out


def langevin_step(sde: SDEParams, score: Array, x: Array, z: Array, snr: float, alpha: float) -> Array:
    """Corrector step of a predictor-corrector sampler.  NOT in the reference (its sampler is predictor-only,
    sampler.py:24-43): PARITY UNPINNED -- restates Song et al. 2021 (Alg. 4/5) in the coordinates whitened by G:
    per series eps = 2 alpha (snr |z| / |G score|)^2 ;  x + eps G^2 score + sqrt(2 eps) G z."""
    x, score, z = (np.asarray(a, dtype=np.float64) for a in (x, score, z))
    G = np.asarray(sde.G, dtype=np.float64)[None, :, None]
    gn = np.sqrt(((G * score) ** 2).sum(axis=(1, 2)))
    zn = np.sqrt((z ** 2).sum(axis=(1, 2)))
    eps = (2.0 * alpha * (snr * zn / gn) ** 2)[:, None, None]
    return x + eps * G * G * score + np.sqrt(2.0 * eps) * G * z


# --------------------------------------------------------------------------
# (f)4  the other two score backbones     src/fdiff/models/score_models.py:169-317
# --------------------------------------------------------------------------
def mlp_score_forward(p: Dict[str, Array], X: Array, t: Array) -> Array:
    """MLPScoreModule.forward (score_models.py:220-246), eval mode.  The blocks are torchvision.ops.MLP(in=d_model,
    hidden=[d_mlp, d_model], dropout=0.1) = Linear -> ReLU -> Dropout -> Linear -> Dropout (torchvision is an unpinned
    dependency of the reference and absent from the image: PARITY UNPINNED against torchvision itself -- the fixture comes
    from the reference's class built over a stand-in with that published structure, oracle/make_golden.py)."""
    X = np.asarray(X, dtype=np.float64)
    B, T, C = X.shape
    h = X.reshape(B, T * C) @ _f64(p, "embedder.weight").T + _f64(p, "embedder.bias")
    D = h.shape[1]
    emb = gfp_embedding(t, p["time_encoder.W"], D)
    h = h + (emb @ _f64(p, "time_encoder.dense.weight").T + _f64(p, "time_encoder.dense.bias"))       # use_time_axis=False
    i = 0
    while f"backbone.{i}.0.weight" in p:
        a = np.maximum(h @ _f64(p, f"backbone.{i}.0.weight").T + _f64(p, f"backbone.{i}.0.bias"), 0.0)
        h = h + (a @ _f64(p, f"backbone.{i}.3.weight").T + _f64(p, f"backbone.{i}.3.bias"))
        i += 1
    out = h @ _f64(p, "unembedder.weight").T + _f64(p, "unembedder.bias")
    return out.reshape(B, T, C)


def lstm_score_forward(p: Dict[str, Array], X: Array, t: Array) -> Array:
    """LSTMScoreModule.forward (score_models.py:291-317), eval mode: Linear embed, + time embedding on the time axis,
    h += nn.LSTM(D, D, batch_first)(h)[0] per layer (torch gate order i, f, g, o; zero initial state), Linear unembed."""
    X = np.asarray(X, dtype=np.float64)
    B, T, C = X.shape
    h = X @ _f64(p, "embedder.weight").T + _f64(p, "embedder.bias")
    D = h.shape[-1]
    emb = gfp_embedding(t, p["time_encoder.W"], D)
    h = h + (emb @ _f64(p, "time_encoder.dense.weight").T + _f64(p, "time_encoder.dense.bias"))[:, None, :]
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))      # noqa: E731
    i = 0
    while f"backbone.{i}.weight_ih_l0" in p:
        Wih, Whh = _f64(p, f"backbone.{i}.weight_ih_l0"), _f64(p, f"backbone.{i}.weight_hh_l0")
        bias = _f64(p, f"backbone.{i}.bias_ih_l0") + _f64(p, f"backbone.{i}.bias_hh_l0")
        hs = np.zeros((B, D))
        cs = np.zeros((B, D))
        ys = np.empty_like(h)
        for tt in range(T):
            g = h[:, tt] @ Wih.T + hs @ Whh.T + bias
            ig, fg, gg, og = sig(g[:, :D]), sig(g[:, D:2 * D]), np.tanh(g[:, 2 * D:3 * D]), sig(g[:, 3 * D:])
            cs = fg * cs + ig * gg
            hs = og * np.tanh(cs)
            ys[:, tt] = hs
        h = h + ys
        i += 1
    return h @ _f64(p, "unembedder.weight").T + _f64(p, "unembedder.bias")


# --------------------------------------------------------------------------
# a10  denoising score-matching loss      src/fdiff/utils/losses.py:39-125
# --------------------------------------------------------------------------
def perturb(sde: SDEParams, X: Array, t: Array, z: Array) -> Tuple[Array, Array, Array]:
    """losses.py:66-85 -- returns (X_noisy, target_noise = z/std, std (B,T))."""
    mean, std = marginal_prob(sde, X, t)
    z = np.asarray(z, dtype=np.float64)
    noise = std[:, :, None] * z                   # losses.py:75  (diag(std) @ z)
    target = z / std[:, :, None]                  # losses.py:78-80
    return mean + noise, target, std              # losses.py:83-85, sde.py:66-77


def dsm_loss(score: Array, target: Array, std: Array, likelihood_weighting: bool) -> float:
    """losses.py:92-124 with reduce_mean=True."""
    score = np.asarray(score, dtype=np.float64)
    if not likelihood_weighting:
        w = 1.0 / np.sum(1.0 / std**2, axis=1)                       # losses.py:96
        losses = w[:, None, None] * (score + target) ** 2            # losses.py:100-102
    else:
        losses = (std[:, :, None] * (score + target)) ** 2           # losses.py:115-121
    losses = losses.reshape(losses.shape[0], -1).mean(axis=-1)       # losses.py:109 / 122
    return float(losses.mean())                                      # losses.py:124


def loss_fn(p: Dict[str, Array], sde: SDEParams, X: Array, t: Array, z: Array, n_head: int,
            likelihood_weighting: bool = False) -> float:
    Xn, target, std = perturb(sde, X, t, z)
    score = score_forward(p, Xn, t, n_head)
    return dsm_loss(score, target, std, likelihood_weighting)


# --------------------------------------------------------------------------
# a12  sampler                            src/fdiff/sampling/sampler.py:45-122
# --------------------------------------------------------------------------
def sample_trajectory(p: Dict[str, Array], sde: SDEParams, z_prior: Array, z_steps: Sequence[Array],
                      n_head: int, eps: float = 1e-5, record: Optional[Sequence[int]] = None):
    """sampler.py:45-109 for ONE batch with an injected noise sequence.

    z_prior (B,T,C) replaces torch.randn in sde.py:85, z_steps[i] replaces randn_like in
    sde.py:157/238.  Returns final X and {step_index(1-based): X after that step}.
    """
    N = len(z_steps)
    ts, dt = timesteps(N, eps)
    X = prior_sampling(sde, z_prior)                                  # sampler.py:80
    rec = {}
    B = X.shape[0]
    for i, t in enumerate(ts):                                        # sampler.py:83
        tb = np.full((B,), t, dtype=np.float32)                       # sampler.py:91-99
        score = score_forward(p, X, tb, n_head)                       # sampler.py:34
        X = sde_step(sde, score, float(t), X, z_steps[i], float(dt))  # sampler.py:36-38
        if record is not None and (i + 1) in record:
            rec[i + 1] = X.copy()
    return X, rec


def num_sample_batches(num_samples: int, sample_batch_size: int) -> Tuple[int, int]:
    """sampler.py:63,74-77 -- (num_batches, per-batch size); the remainder is silently dropped."""
    nb = max(1, num_samples // sample_batch_size)
    return nb, min(num_samples, sample_batch_size)


# --------------------------------------------------------------------------
# a11  optimiser / LR schedule   score_models.py:122-130, diffusers formula
# --------------------------------------------------------------------------
def cosine_warmup_factor(step: int, num_warmup_steps: int, num_training_steps: int,
                         num_cycles: float = 0.5) -> float:
    """diffusers.optimization.get_cosine_schedule_with_warmup lr_lambda (diffusers is absent
    from /root/reference and this image: formula-pinned)."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def adamw_step(param: Array, grad: Array, m: Array, v: Array, step: int, lr: float,
               beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
               weight_decay: float = 1e-2) -> Tuple[Array, Array, Array]:
    """torch.optim.AdamW defaults (score_models.py:123); step is 1-based."""
    param = param * (1.0 - lr * weight_decay)
    m = beta1 * m + (1.0 - beta1) * grad
    v = beta2 * v + (1.0 - beta2) * grad * grad
    bc1 = 1.0 - beta1**step
    bc2 = 1.0 - beta2**step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    param = param - (lr / bc1) * (m / denom)
    return param, m, v


def clip_grad_norm_scale(total_norm: float, max_norm: float = 1.0) -> float:
    """torch.nn.utils.clip_grad_norm_ coefficient (Lightning gradient_clip_val, trainer/default.yaml:4)."""
    return min(1.0, max_norm / (total_norm + 1e-6))


# --------------------------------------------------------------------------
# DiffusionDataset statistics      src/fdiff/dataloaders/datamodules.py:42-62
# --------------------------------------------------------------------------
def dataset_standardize(X: Array, fourier_transform: bool, X_ref: Optional[Array] = None):
    """Returns (standardised X, feature_mean (T,C), feature_std (T,C)); std is unbiased (torch default)."""
    X = np.asarray(X, dtype=np.float64)
    if fourier_transform:
        X = dft(X)                                           # datamodules.py:42-43
    if X_ref is None:
        ref = X
    else:
        ref = dft(X_ref) if fourier_transform else np.asarray(X_ref, dtype=np.float64)  # :49-50
    mean = ref.mean(axis=0)                                  # :52
    std = ref.std(axis=0, ddof=1)                            # :53
    return (X - mean) / std, mean, std                       # :61-62


# ---------------------------------------------------------------------------------------------------------------------
# The engine's counter-based generator (csrc/fd_philox.h) restated for the tests: Philox4x32-10 of Salmon, Moraes, Dror and
# Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11), constants and round function as published (Random123 v1.14
# philox.h; the known-answer vectors of its kat_vectors file pin this restatement in tests/test_oracle_golden.py), and the
# 16-decisions-per-evaluation dropout rule of the bf16 training path (fd_drop16).
def philox4x32_10(counter: Array, key: Array) -> Array:
    """counter (..., 4) uint32, key (..., 2) uint32 -> (..., 4) uint32."""
    c = np.asarray(counter, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    k = np.asarray(key, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2                       # 32 x 32 -> 64 bit products (no overflow in uint64)
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def engine_philox_words(seed: int, offset: int, n: int) -> Array:
    """What fd_philox_words writes: counter i = (lo, hi, 0, 0) of the 64-bit value offset + i, key = (lo, hi) of seed."""
    ctr = (np.uint64(offset) + np.arange(n, dtype=np.uint64))
    counter = np.stack([ctr & np.uint64(0xFFFFFFFF), ctr >> np.uint64(32), np.zeros(n, np.uint64), np.zeros(n, np.uint64)], axis=-1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64), (n, 2))
    return philox4x32_10(counter, key)


def dropout_decisions16(words: Array, p: float) -> Array:
    """fd_drop16: (n, 4) uint32 generator output -> (n,) uint16, bit e = keep decision e: the 16-bit little-endian window at
    byte offset e of the 128-bit output (wrapping) >= thr16 = round(p * 65536)."""
    thr16 = int(p * 65536.0 + 0.5)
    if p > 0 and thr16 == 0:
        thr16 = 1
    b = np.ascontiguousarray(words.astype("<u4")).view(np.uint8).reshape(-1, 16).astype(np.uint32)
    out = np.zeros(b.shape[0], dtype=np.uint32)
    for e in range(16):
        window = b[:, e] | (b[:, (e + 1) % 16] << 8)
        out |= (window >= thr16).astype(np.uint32) << e
    return out.astype(np.uint16)

