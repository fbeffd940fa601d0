"""GPU: stand-alone PositionalEncoding / GaussianFourierProjection -- the reference's tests/test_transformer.py:18-82
re-expressed against the engine (shape, max_norm bound, each position / batch element gets the right vector)."""
import numpy as np
import pytest
import torch

from .gpu_util import DEV

pytestmark = pytest.mark.gpu
max_len, batch_size, d_model, EPS = 20, 16, 5, 1e-5


def test_positional_encoding():
    from fourierdiffusion_amd.models.transformer import PositionalEncoding
    torch.manual_seed(42)
    enc = PositionalEncoding(d_model=d_model, max_len=max_len)
    raw = enc.embedding.weight.clone()
    X = torch.randn((batch_size, max_len, d_model), device=DEV)
    out = enc(X)
    assert out.shape == X.shape
    assert torch.max(torch.sum((out - X) ** 2, dim=-1)) <= d_model + EPS          # max_norm = sqrt(d_model)
    W = enc.embedding.weight                                                      # renormed in place, like torch
    for l in range(max_len):
        assert torch.allclose((out - X)[:, l, :], W[l, :].expand(batch_size, -1), atol=EPS)
    n = raw.norm(dim=1)
    want = torch.where((n > np.sqrt(d_model))[:, None], raw * (np.sqrt(d_model) / (n + 1e-7))[:, None], raw)
    assert torch.allclose(W.cpu(), want, atol=1e-6)


def test_gaussian_fourier_projection():
    from fourierdiffusion_amd.models.transformer import GaussianFourierProjection
    torch.manual_seed(42)
    enc = GaussianFourierProjection(d_model=d_model)
    X = torch.randn((batch_size, max_len, d_model), device=DEV)
    timesteps = torch.randint(low=0, high=max_len, size=(batch_size,)).float() / max_len
    out = enc(X, timesteps.to(DEV))
    assert out.shape == X.shape
    Wc, dw, db = enc.W.cpu(), enc.dense.weight.cpu(), enc.dense.bias.cpu()
    proj = timesteps[:, None] * Wc[None, :] * 2 * np.pi
    emb = torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)[:, :d_model]
    truth = emb @ dw.T + db
    for l in range(max_len):
        assert torch.allclose((out - X)[:, l, :].cpu(), truth, atol=EPS)
    flat = enc(X[:, 0, :].contiguous(), timesteps.to(DEV), use_time_axis=False)
    assert torch.allclose((flat - X[:, 0, :]).cpu(), truth, atol=EPS)
