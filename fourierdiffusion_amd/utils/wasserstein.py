"""Wasserstein-2 distances between two sample sets on the HIP engine -- same surface as
fdiff.utils.wasserstein.WassersteinDistances (reference: src/fdiff/utils/wasserstein.py:12-199).

The directions come from the same numpy Generator calls as the reference (host logic); projections, per-direction sorts and
the exact 1-D transport run on the GPU through the C ABI (fd_project_rows, fd_transpose_rows, fd_sort_rows,
fd_w2_sorted_rows).  All directions of a call are processed together instead of one POT call per direction."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from .. import _C
from .tensors import check_flat_array


def sort_rows(P: torch.Tensor) -> torch.Tensor:
    """Every row of the (K, n) device tensor sorted ascending."""
    K, n = P.shape
    h, L = _C.ctx(P.device), _C.lib()
    need = C.c_size_t(0)
    _C.check(L.fd_sort_rows_temp_bytes(h, K, n, C.byref(need)), h)
    temp = torch.empty((need.value,), dtype=torch.uint8, device=P.device)
    out = torch.empty_like(P)
    _C.check(L.fd_sort_rows(h, P.data_ptr(), out.data_ptr(), K, n, temp.data_ptr(), need.value, _C.stream_of(P)), h)
    return out


def project_rows(data: torch.Tensor, directions: np.ndarray) -> torch.Tensor:
    """(K, n) = directions (K, d) . data (n, d)^T   (wasserstein.py:150-153 for all directions at once)."""
    n, d = data.shape
    dirs = _C.dev_f32(torch.from_numpy(np.ascontiguousarray(directions, dtype=np.float32)).to(data.device), "directions")
    K = dirs.shape[0]
    assert dirs.shape == (K, d), f"directions must have shape (K, {d}), got {tuple(dirs.shape)}"
    out = torch.empty((K, n), dtype=torch.float32, device=data.device)
    h = _C.ctx(data.device)
    _C.check(_C.lib().fd_project_rows(h, data.data_ptr(), dirs.data_ptr(), out.data_ptr(), n, d, K, _C.stream_of(data)), h)
    return out


def transpose_rows(data: torch.Tensor) -> torch.Tensor:
    """(d, n): feature f of every sample in row f (the marginal directions are the standard basis, wasserstein.py:77-89)."""
    n, d = data.shape
    out = torch.empty((d, n), dtype=torch.float32, device=data.device)
    h = _C.ctx(data.device)
    _C.check(_C.lib().fd_transpose_rows(h, data.data_ptr(), out.data_ptr(), n, d, _C.stream_of(data)), h)
    return out


def w2_sorted_rows(a: torch.Tensor, b: torch.Tensor) -> np.ndarray:
    """W2 between sorted row k of a (K, n) and of b (K, m) for every k -- exact 1-D transport with uniform weights."""
    K, n = a.shape
    assert b.shape[0] == K
    out = torch.empty((K,), dtype=torch.float32, device=a.device)
    h = _C.ctx(a.device)
    _C.check(_C.lib().fd_w2_sorted_rows(h, a.data_ptr(), b.data_ptr(), out.data_ptr(), K, n, b.shape[1], _C.stream_of(a)), h)
    return out.cpu().numpy().astype(np.float64)


class WassersteinDistances:
    """wasserstein.py:12-36: original_data / other_data are (n, d) and (m, d) sample sets (numpy or torch, any device)."""

    def __init__(self, original_data, other_data, normalisation: Optional[str] = "none", seed: Optional[int] = None) -> None:
        self.original_data = check_flat_array(original_data)
        self.other_data = check_flat_array(other_data)
        assert self.original_data.shape[1] == self.other_data.shape[1], "both sample sets must have the same number of features"
        self.normalisation = normalisation
        self.rng = np.random.default_rng(seed)

    # ---- directions (host logic, identical draws to the reference)
    def random_direction(self, dim: int) -> np.ndarray:
        vector = self.rng.normal(size=dim)                        # wasserstein.py:55-58
        return vector / np.linalg.norm(vector)

    def get_random_directions(self, n_directions: int) -> list[np.ndarray]:
        dimension = self.original_data.shape[1]
        return [self.random_direction(dimension) for _ in range(n_directions)]

    def get_marginal_directions(self) -> list[np.ndarray]:
        dimension = self.original_data.shape[1]
        return [np.identity(dimension)[i] for i in range(dimension)]

    # ---- distances
    def _distances(self, orig_rows: torch.Tensor, other_rows: torch.Tensor) -> np.ndarray:
        if self.normalisation == "none":
            scale = None
        elif self.normalisation == "standardise":                 # wasserstein.py:155-161: both divided by std(original)
            scale = orig_rows.std(dim=1, unbiased=False).cpu().numpy().astype(np.float64)
        else:
            raise ValueError(f"Unrecognised normalisation type: {self.normalisation}")
        d = w2_sorted_rows(sort_rows(orig_rows), sort_rows(other_rows))
        return d if scale is None else d / scale

    def directional_distances(self, directions) -> np.ndarray:
        """All of `directions` (K, d) in one pass: one projection GEMM per sample set, one segmented sort, one W2 launch."""
        directions = np.asarray(directions, dtype=np.float64).reshape(-1, self.original_data.shape[1])
        return self._distances(project_rows(self.original_data, directions), project_rows(self.other_data, directions))

    def directional_distance(self, direction: np.ndarray) -> float:
        return float(self.directional_distances(np.asarray(direction)[None])[0])          # wasserstein.py:117-143

    def feature_distance(self, feature: int) -> float:
        o = self.original_data[:, feature:feature + 1].contiguous()                        # wasserstein.py:91-115
        x = self.other_data[:, feature:feature + 1].contiguous()
        return float(self._distances(transpose_rows(o), transpose_rows(x))[0])

    def sliced_distances(self, num_directions: int) -> np.ndarray:
        return self.directional_distances(np.stack(self.get_random_directions(num_directions)))   # wasserstein.py:163-181

    def marginal_distances(self) -> np.ndarray:
        return self._distances(transpose_rows(self.original_data), transpose_rows(self.other_data))   # wasserstein.py:183-199
