"""VE / VP SDE schedulers on the HIP engine -- same surface as fdiff.schedulers.sde
(reference: src/fdiff/schedulers/sde.py:13-246).

Attributes kept from the reference: ``noise_scaling, eps, G, G_matrix, timesteps, step_size, T``.
What differs underneath: the reference spells every row scaling as ``diag_embed`` + a
(T,T)@(B,T,C) matmul and draws noise with ``torch.randn*``; here each method is one fused HIP
kernel and the noise comes from the engine's Philox stream unless injected via ``noise=``.
"""
from __future__ import annotations

import abc
import ctypes as C
import math
from collections import namedtuple
from typing import Optional, Tuple

import torch

from .. import _C, _rng

SamplingOutput = namedtuple("SamplingOutput", ["prev_sample"])


class SDE(abc.ABC):
    """Forward SDE dx = f dt + G dw with a diagonal, per-frequency diffusion matrix G."""

    kind: int = -1

    def __init__(self, fourier_noise_scaling: bool = False, eps: float = 1e-5):
        super().__init__()
        self.noise_scaling = fourier_noise_scaling
        self.eps = eps
        self.G: Optional[torch.Tensor] = None
        self._G_dev: dict = {}

    # ------------------------------------------------------------ plumbing
    @property
    def T(self) -> float:
        return 1.0

    @abc.abstractmethod
    def _params(self) -> Tuple[float, float]:
        ...

    def _c_params(self) -> _C.SdeParams:
        p0, p1 = self._params()
        return _C.SdeParams(self.kind, p0, p1)

    def G_on(self, device: torch.device) -> torch.Tensor:
        """G as a float32 device vector (cached per device)."""
        assert self.G is not None
        key = str(device)
        g = self._G_dev.get(key)
        if g is None or g.shape != self.G.shape:
            g = self.G.to(device=device, dtype=torch.float32).contiguous()
            self._G_dev[key] = g
        return g

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_G_dev"] = {}
        return st

    def __setstate__(self, st):
        # also the entry point for scheduler objects pickled by the REFERENCE (Lightning checkpoints keep the scheduler
        # instance in hyper_parameters, class path fdiff.schedulers.sde.*): same attribute names, no device cache
        self.__dict__.update(st)
        self._G_dev = {}

    # ------------------------------------------------------------ sde.py:42-64
    def set_noise_scaling(self, max_len: int) -> None:
        """G_k = 1, or with Fourier scaling 1/sqrt(2) except G_0 (and G_{T/2}, T even) = 1
        ("mirrored Brownian motion": Re/Im halves of a real signal's spectrum carry half the variance)."""
        G = torch.ones(max_len)
        if self.noise_scaling:
            G = 1 / (math.sqrt(2)) * G
            G[0] *= math.sqrt(2)
            if max_len % 2 == 0:
                G[max_len // 2] *= math.sqrt(2)
        self.G = G
        self.G_matrix = torch.diag(G)     # kept for API compatibility; the engine never forms it
        self._G_dev = {}

    def set_timesteps(self, num_diffusion_steps: int) -> None:
        self.timesteps = torch.linspace(1.0, self.eps, num_diffusion_steps)
        self.step_size = self.timesteps[0] - self.timesteps[1]

    # ------------------------------------------------------------ sde.py:66-87
    def marginal_prob(self, x: torch.Tensor, t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Perturbation-kernel parameters: mean (B,T,C), std (B,T) = s(t) * G."""
        if self.G is None:
            self.set_noise_scaling(x.shape[1])
        xd = _C.dev_f32(x, "x")
        td = _C.dev_f32(t.to(xd.device), "t")
        B, T, Cn = xd.shape
        mean = torch.empty_like(xd)
        std = torch.empty((B, T), device=xd.device, dtype=torch.float32)
        zeros = torch.zeros_like(xd)
        h = _C.ctx(xd.device)
        p = self._c_params()
        rc = _C.lib().fd_perturb(h, C.byref(p), self.G_on(xd.device).data_ptr(), xd.data_ptr(), td.data_ptr(),
                                 zeros.data_ptr(), 0, 0, mean.data_ptr(), None, std.data_ptr(), B, T, Cn,
                                 _C.stream_of(xd))
        _C.check(rc, h)
        return mean, std

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """mean(x0, t) + noise -- the noise is already scaled by the caller (sde.py:66-77)."""
        mean, _ = self.marginal_prob(original_samples, timesteps)
        return mean + noise.to(mean.device)

    def perturb(self, x: torch.Tensor, t: torch.Tensor, noise: Optional[torch.Tensor] = None):
        """Fused forward perturbation used by the loss (losses.py:66-85):
        returns (x_noisy, target = z/std, std)."""
        if self.G is None:
            self.set_noise_scaling(x.shape[1])
        xd = _C.dev_f32(x, "x")
        td = _C.dev_f32(t.to(xd.device), "t")
        B, T, Cn = xd.shape
        xn = torch.empty_like(xd)
        target = torch.empty_like(xd)
        std = torch.empty((B, T), device=xd.device, dtype=torch.float32)
        z = None if noise is None else _C.dev_f32(noise.to(xd.device), "noise")
        key, off = (0, 0) if z is not None else _rng.stream()
        h = _C.ctx(xd.device)
        p = self._c_params()
        rc = _C.lib().fd_perturb(h, C.byref(p), self.G_on(xd.device).data_ptr(), xd.data_ptr(), td.data_ptr(),
                                 _C.ptr(z), key, off, xn.data_ptr(), target.data_ptr(), std.data_ptr(),
                                 B, T, Cn, _C.stream_of(xd))
        _C.check(rc, h)
        return xn, target, std

    def prior_sampling(self, shape: Tuple[int, ...], noise: Optional[torch.Tensor] = None,
                       device: Optional[torch.device] = None) -> torch.Tensor:
        """G * z (VE: * sigma_max), z ~ N(0, I).  Needs ``set_noise_scaling`` first, like the reference
        (AttributeError on G_matrix otherwise, sde.py:81)."""
        _ = self.G_matrix
        B, T, Cn = shape
        if noise is not None:
            dev = noise.device if noise.device.type == "cuda" else torch.device("cuda")
        else:
            dev = torch.device(device) if device is not None else torch.device("cuda")
        out = torch.empty(tuple(shape), device=dev, dtype=torch.float32)
        z = None if noise is None else _C.dev_f32(noise.to(dev), "noise")
        key, off = (0, 0) if z is not None else _rng.stream()
        h = _C.ctx(dev)
        p = self._c_params()
        rc = _C.lib().fd_prior_sample(h, C.byref(p), self.G_on(dev).data_ptr(), _C.ptr(z), key, off,
                                      out.data_ptr(), B, T, Cn, _C.stream_of(out))
        _C.check(rc, h)
        return out

    # ------------------------------------------------------------ sde.py:129-165, 215-246
    def step(self, model_output: torch.Tensor, timestep: float, sample: torch.Tensor,
             noise: Optional[torch.Tensor] = None) -> SamplingOutput:
        """One Euler-Maruyama step of the reverse SDE, fused (score-add + noise-inject in one pass)."""
        assert self.G is not None
        assert self.step_size > 0
        xd = _C.dev_f32(sample, "sample")
        sd = _C.dev_f32(model_output.to(xd.device), "model_output")
        B, T, Cn = xd.shape
        out = torch.empty_like(xd)
        z = None if noise is None else _C.dev_f32(noise.to(xd.device), "noise")
        key, off = (0, 0) if z is not None else _rng.stream()
        h = _C.ctx(xd.device)
        p = self._c_params()
        rc = _C.lib().fd_sde_step(h, C.byref(p), self.G_on(xd.device).data_ptr(), xd.data_ptr(), sd.data_ptr(),
                                  _C.ptr(z), key, off, float(timestep), float(self.step_size),
                                  out.data_ptr(), B, T, Cn, _C.stream_of(xd))
        _C.check(rc, h)
        return SamplingOutput(prev_sample=out)


class VEScheduler(SDE):
    """Variance-exploding SDE: std(t) = sigma_min (sigma_max/sigma_min)^t * G (sde.py:90-165)."""

    kind = 1

    def __init__(self, sigma_min: float = 0.01, sigma_max: float = 50.0, fourier_noise_scaling: bool = False,
                 eps: float = 1e-5):
        super().__init__(fourier_noise_scaling=fourier_noise_scaling, eps=eps)
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max

    def _params(self):
        return float(self.sigma_min), float(self.sigma_max)


class VPScheduler(SDE):
    """Variance-preserving SDE with beta(t) = beta_0 + t (beta_1 - beta_0) (sde.py:168-246)."""

    kind = 0

    def __init__(self, beta_min: float = 0.1, beta_max: float = 20.0, fourier_noise_scaling: bool = False,
                 eps: float = 1e-5):
        super().__init__(fourier_noise_scaling=fourier_noise_scaling, eps=eps)
        self.beta_0 = beta_min
        self.beta_1 = beta_max

    def _params(self):
        return float(self.beta_0), float(self.beta_1)

    def get_beta(self, timestep: float) -> float:
        return self.beta_0 + timestep * (self.beta_1 - self.beta_0)
