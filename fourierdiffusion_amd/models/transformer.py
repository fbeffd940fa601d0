"""PositionalEncoding and GaussianFourierProjection on the HIP engine -- same surface as fdiff.models.transformer
(reference: src/fdiff/models/transformer.py:8-29, 61-91).  Inside ScoreModule these encoders are fused into the
score network's embed stage; the stand-alone classes exist for API parity (and the reference's own unit tests).
`TimeEncoding` (:32-58) is dead code on the SDE path (score_models.py:159-166) and is not provided."""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch

from .. import _C


class PositionalEncoding:
    """Learned table (max_len, d_model) with max_norm = sqrt(d_model), added over the time axis."""

    def __init__(self, d_model: int, max_len: int):
        self.d_model, self.max_len = d_model, max_len
        self.embedding = SimpleNamespace(weight=torch.nn.init.normal_(torch.empty(max_len, d_model)),
                                         max_norm=math.sqrt(d_model))

    def to(self, device) -> "PositionalEncoding":
        self.embedding.weight = self.embedding.weight.to(device)
        return self

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        xd = _C.dev_f32(x, "x")
        if self.embedding.weight.device != xd.device:
            self.to(xd.device)
        B, T, D = xd.shape
        assert D == self.d_model and T <= self.max_len
        out = torch.empty_like(xd)
        h = _C.ctx(xd.device)
        # like nn.Embedding(max_norm), the looked-up rows are renormed IN PLACE
        _C.check(_C.lib().fd_positional_add(h, xd.data_ptr(), self.embedding.weight.data_ptr(), out.data_ptr(), B, T, D,
                                            float(self.embedding.max_norm), _C.stream_of(xd)), h)
        return out

    __call__ = forward


class GaussianFourierProjection:
    """Fixed random Fourier features of the diffusion time -> Linear(d_model, d_model), broadcast-added."""

    def __init__(self, d_model: int, scale: float = 30.0):
        self.d_model = d_model
        self.W = torch.randn((d_model + 1) // 2) * scale           # requires_grad=False in the reference
        lin = torch.nn.Linear(d_model, d_model)                     # host-side init only (same initialiser)
        self.dense = SimpleNamespace(weight=lin.weight.detach().clone(), bias=lin.bias.detach().clone())
        self.dense.__call__ = None

    def to(self, device) -> "GaussianFourierProjection":
        self.W = self.W.to(device)
        self.dense.weight = self.dense.weight.to(device)
        self.dense.bias = self.dense.bias.to(device)
        return self

    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, use_time_axis: bool = True) -> torch.Tensor:
        xd = _C.dev_f32(x, "x")
        if self.W.device != xd.device:
            self.to(xd.device)
        td = _C.dev_f32(timesteps.to(xd.device), "timesteps")
        if use_time_axis:
            B, T, D = xd.shape
        else:
            (B, D), T = xd.shape, 0
        assert D == self.d_model and td.shape[0] == B
        out = torch.empty_like(xd)
        h = _C.ctx(xd.device)
        _C.check(_C.lib().fd_time_embed_add(h, xd.data_ptr(), td.data_ptr(), self.W.data_ptr(),
                                            self.dense.weight.data_ptr(), self.dense.bias.data_ptr(), out.data_ptr(),
                                            B, T, D, _C.stream_of(xd)), h)
        return out

    __call__ = forward
