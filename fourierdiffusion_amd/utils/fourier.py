"""dft / idft on the HIP engine -- same surface as fdiff.utils.fourier
(reference: src/fdiff/utils/fourier.py:8-87).

(B,T,C) real series <-> same-shape real spectral representation:
rows [0, T//2] hold Re X_k, rows [T//2+1, T) hold Im X_k (k = 1..), ortho norm.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _C


def _to_engine(x: torch.Tensor, name: str) -> torch.Tensor:
    if x.dim() != 3:
        raise AssertionError(f"{name} must have shape (batch_size, max_len, n_channels), got {tuple(x.shape)}")
    if x.device.type != "cuda":
        if not torch.cuda.is_available():
            raise _C.FdError("dft/idft run on the HIP engine and no GPU is visible (no CPU fallback)")
        x = x.to("cuda")
    return _C.dev_f32(x.detach(), name)


def _run(fn_name: str, x: torch.Tensor, mean: Optional[torch.Tensor] = None,
         std: Optional[torch.Tensor] = None) -> torch.Tensor:
    src_device = x.device
    xd = _to_engine(x, "x")
    y = torch.empty_like(xd)
    B, T, Cn = xd.shape
    h = _C.ctx(xd.device)
    L = _C.lib()
    if mean is None:
        rc = getattr(L, fn_name)(h, xd.data_ptr(), y.data_ptr(), B, T, Cn, _C.stream_of(xd))
    else:
        md = _C.dev_f32(mean.to(xd.device), "mean")
        sd = _C.dev_f32(std.to(xd.device), "std")
        if md.shape != (T, Cn) or sd.shape != (T, Cn):
            raise AssertionError("feature mean/std must have shape (max_len, n_channels)")
        rc = getattr(L, fn_name)(h, xd.data_ptr(), md.data_ptr(), sd.data_ptr(), y.data_ptr(), B, T, Cn,
                                 _C.stream_of(xd))
    _C.check(rc, h)
    return y if src_device.type == "cuda" else y.to(src_device)


def dft(x: torch.Tensor) -> torch.Tensor:
    """fourier.py:8-45 -- returns a detached tensor of the same shape and device."""
    return _run("fd_rfft_pack", x)


def idft(x: torch.Tensor) -> torch.Tensor:
    """fourier.py:48-87 -- inverse of :func:`dft` (the reference version only works on CPU tensors,
    fourier.py:66; this one accepts either and returns on the input's device)."""
    return _run("fd_irfft_unpack", x)


def dft_standardize(x: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """(dft(x) - mean) / std in one pass (DiffusionDataset semantics, datamodules.py:42-43,61-62)."""
    return _run("fd_rfft_pack_standardize", x, mean, std)


def destandardize_idft(x: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """idft(x * std + mean) in one pass (cmd/sample.py:76-82)."""
    return _run("fd_destandardize_irfft", x, mean, std)


def spectral_density(x: torch.Tensor, apply_dft: bool = True) -> torch.Tensor:
    """fourier.py:90-124 -- |X_k|^2 per frequency bin, shape (batch_size, max_len // 2 + 1, n_channels)."""
    src_device = x.device
    xt = _to_engine(dft(x) if apply_dft else x, "x")
    B, T, Cn = xt.shape
    out = torch.empty((B, T // 2 + 1, Cn), dtype=torch.float32, device=xt.device)
    h = _C.ctx(xt.device)
    _C.check(_C.lib().fd_spectral_density(h, xt.data_ptr(), out.data_ptr(), B, T, Cn, _C.stream_of(xt)), h)
    return out if src_device.type == "cuda" else out.to(src_device)


def localization_metrics(X: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """fourier.py:127-175 -- delocalisation of every series in the time and in the frequency domain, two (batch_size,)
    tensors."""
    src_device = X.device
    xd = _to_engine(X, "X")
    xt = dft(xd)
    B, T, Cn = xd.shape
    loc = torch.empty((B,), dtype=torch.float32, device=xd.device)
    spec_loc = torch.empty_like(loc)
    h = _C.ctx(xd.device)
    _C.check(_C.lib().fd_localization_metrics(h, xd.data_ptr(), xt.data_ptr(), loc.data_ptr(), spec_loc.data_ptr(), B, T, Cn,
                                               _C.stream_of(xd)), h)
    if src_device.type != "cuda":
        loc, spec_loc = loc.to(src_device), spec_loc.to(src_device)
    return loc, spec_loc


def smooth_frequency(X: torch.Tensor, sigma: float) -> torch.Tensor:
    """fourier.py:178-209 -- idft(Gaussian mixing over frequencies of dft(X)).  Like the reference this is defined for odd
    max_len only (its frequency vector has max_len - 1 entries otherwise and the einsum raises)."""
    src_device = X.device
    xd = _to_engine(X, "X")
    B, T, Cn = xd.shape
    if T % 2 == 0:
        raise RuntimeError(f"smooth_frequency: max_len={T} must be odd (the reference's Gaussian kernel is "
                           f"({T - 1}, {T - 1}) for even lengths and its einsum fails, fourier.py:192-203)")
    xt = dft(xd)
    mixed = torch.empty_like(xt)
    gauss = torch.empty((T, T), dtype=torch.float32, device=xd.device)
    h = _C.ctx(xd.device)
    L = _C.lib()
    for b0 in range(0, B, 65535):                                  # (grid.y limit of one launch)
        nb = min(65535, B - b0)
        _C.check(L.fd_frequency_smooth(h, xt[b0:b0 + nb].data_ptr(), float(sigma), gauss.data_ptr(),
                                       mixed[b0:b0 + nb].data_ptr(), nb, T, Cn, _C.stream_of(xd)), h)
    out = idft(mixed)
    return out if src_device.type == "cuda" else out.to(src_device)
