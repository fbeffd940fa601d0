"""Same surface as fdiff.utils.tensors (reference: src/fdiff/utils/tensors.py:5-23)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _C


def check_flat_array(x: torch.Tensor | np.ndarray) -> torch.Tensor:
    """(n, ...) samples -> contiguous (n, d) float32 tensor ON THE GPU (the reference returns a numpy array for POT;
    the metrics here run on the engine, so the flat array stays device-resident)."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    assert isinstance(x, torch.Tensor), f"x must be a numpy array or a torch tensor. Got {type(x)}"
    x = x.detach()
    if x.dim() > 2:
        x = x.reshape(x.shape[0], -1)
    assert x.dim() == 2, f"x must be a 2d array. Got {x.dim()}d array."
    if x.device.type != "cuda":
        if not torch.cuda.is_available():
            raise _C.FdError("the Wasserstein metrics run on the HIP engine and no GPU is visible (no CPU fallback)")
        x = x.to("cuda")
    return _C.dev_f32(x, "samples")
