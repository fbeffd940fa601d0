"""Counter-based, library-independent weight recipe for parity fixtures.

TEST INFRASTRUCTURE ONLY.  The golden fixtures need the *same* score-network
weights inside the reference (oracle/make_golden.py, this container) and
inside the HIP engine (GPU box) without committing 12.8 MB of state: every
tensor is a pure function of (seed, tensor name, flat index) through a
splitmix64 hash, so it is reproducible with nothing but numpy integer ops.

Key names follow the reference's state_dict.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform(name: str, shape, seed: int, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """U[lo, hi) float32 tensor, a pure function of (seed, name, index)."""
    n = int(np.prod(shape))
    key = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF) << np.uint64(32)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) + key) ^ _splitmix64(np.array([seed], dtype=np.uint64))
    bits = _splitmix64(ctr) >> np.uint64(11)                  # 53 random bits
    u = bits.astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(name: str, shape, seed: int) -> np.ndarray:
    """N(0,1) float32 via Box-Muller on two uniform streams."""
    u1 = uniform(name + "#u1", shape, seed, 0.0, 1.0).astype(np.float64)
    u2 = uniform(name + "#u2", shape, seed, 0.0, 1.0).astype(np.float64)
    u1 = np.maximum(u1, 1e-12)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def make_state_dict(n_channels: int, max_len: int, d_model: int, num_layers: int,
                    dim_feedforward: int = 2048, seed: int = 1234) -> Dict[str, np.ndarray]:
    """Score-network weights with the reference's key names and shapes.

    Scales mimic a trained-ish network: Linear ~ U(+-1/sqrt(fan_in)) like torch's default
    init, non-trivial biases and LayerNorm affine, N(0,1) positional table so that about
    half the rows exceed max_norm=sqrt(d_model) and exercise the renorm, and a per-layer
    distinct draw (torch's deep-copied init would make all layers identical).
    """
    D, C, F = d_model, n_channels, dim_feedforward
    sd: Dict[str, np.ndarray] = {}

    def lin(name, out_f, in_f, wscale=1.0):
        b = wscale / math.sqrt(in_f)
        sd[name + ".weight"] = uniform(name + ".weight", (out_f, in_f), seed, -b, b)
        sd[name + ".bias"] = uniform(name + ".bias", (out_f,), seed, -b, b)

    sd["pos_encoder.embedding.weight"] = normal("pos_encoder.embedding.weight", (max_len, D), seed)
    sd["time_encoder.W"] = (normal("time_encoder.W", ((D + 1) // 2,), seed) * np.float32(30.0)).astype(np.float32)
    lin("time_encoder.dense", D, D)
    lin("embedder", D, C)
    lin("unembedder", C, D)
    for i in range(num_layers):
        pre = f"backbone.layers.{i}."
        b = math.sqrt(6.0 / (D + 3 * D))          # xavier-uniform like nn.MultiheadAttention
        sd[pre + "self_attn.in_proj_weight"] = uniform(pre + "in_proj_weight", (3 * D, D), seed, -b, b)
        sd[pre + "self_attn.in_proj_bias"] = uniform(pre + "in_proj_bias", (3 * D,), seed, -0.05, 0.05)
        lin(pre + "self_attn.out_proj", D, D)
        lin(pre + "linear1", F, D)
        lin(pre + "linear2", D, F)
        for nm in ("norm1", "norm2"):
            sd[pre + nm + ".weight"] = uniform(pre + nm + ".weight", (D,), seed, 0.9, 1.1)
            sd[pre + nm + ".bias"] = uniform(pre + nm + ".bias", (D,), seed, -0.1, 0.1)
    return sd


def make_state_dict_backbone(kind: str, n_channels: int, max_len: int, d_model: int, num_layers: int, d_mlp: int = 512,
                             seed: int = 1234) -> Dict[str, np.ndarray]:
    """Weights of the reference's MLPScoreModule / LSTMScoreModule (score_models.py:169-317) with its key names:
    MLP  : embedder (D, T*C), unembedder (T*C, D), backbone.{i}.0 / .3 (torchvision.ops.MLP = Sequential(Linear, ReLU,
           Dropout, Linear, Dropout)); LSTM: backbone.{i}.weight_ih_l0 | weight_hh_l0 | bias_ih_l0 | bias_hh_l0."""
    D, C, T = d_model, n_channels, max_len
    sd: Dict[str, np.ndarray] = {}

    def lin(name, out_f, in_f):
        b = 1.0 / math.sqrt(in_f)
        sd[name + ".weight"] = uniform(name + ".weight", (out_f, in_f), seed, -b, b)
        sd[name + ".bias"] = uniform(name + ".bias", (out_f,), seed, -b, b)

    sd["time_encoder.W"] = (normal("time_encoder.W", ((D + 1) // 2,), seed) * np.float32(30.0)).astype(np.float32)
    lin("time_encoder.dense", D, D)
    cin = T * C if kind == "mlp" else C
    lin("embedder", D, cin)
    lin("unembedder", cin, D)
    for i in range(num_layers):
        if kind == "mlp":
            lin(f"backbone.{i}.0", d_mlp, D)
            lin(f"backbone.{i}.3", D, d_mlp)
        else:
            k = 1.0 / math.sqrt(D)
            for nm, shape in (("weight_ih_l0", (4 * D, D)), ("weight_hh_l0", (4 * D, D)), ("bias_ih_l0", (4 * D,)),
                              ("bias_hh_l0", (4 * D,))):
                sd[f"backbone.{i}.{nm}"] = uniform(f"backbone.{i}.{nm}", shape, seed, -k, k)
    return sd


def randn(name: str, shape, seed: int) -> np.ndarray:
    """Deterministic N(0,1) input tensors for tests (same hash family)."""
    return normal("input:" + name, shape, seed)
