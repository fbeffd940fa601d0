"""Denoising score-matching loss on the HIP engine -- same surface as fdiff.utils.losses.get_sde_loss_fn
(reference: src/fdiff/utils/losses.py:12-127).

loss_fn(model, batch): t ~ U[eps, T], z ~ N(0, I), x_t = mean + diag(std) z, target = -z/std,
weight 1/tr(Sigma^-1) (default) or the Mahalanobis form (likelihood weighting).  The reference
materialises two (B,T,T) diagonal matrices and four matmuls for what are row scalings; here it is
fd_perturb -> score network -> fd_dsm_loss, and in training mode the same call also runs the backward
pass (there is no autograd graph): the parameter gradients are ACCUMULATED into ``model.grads``.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import torch

from .. import _C
from ..schedulers.sde import SDE
from .dataclasses import DiffusableBatch


def get_sde_loss_fn(
    scheduler: SDE,
    train: bool,
    reduce_mean: bool = True,
    likelihood_weighting: bool = False,
) -> Callable[..., torch.Tensor]:
    if not reduce_mean:
        raise NotImplementedError("reduce_mean=False (0.5 * sum) is never used by the reference's callers")

    def loss_fn(model, batch: DiffusableBatch, noise: Optional[torch.Tensor] = None,
                backward: Optional[bool] = None, grad_weight: float = 1.0) -> torch.Tensor:
        """Scalar loss tensor (device).  noise= injects z (parity tests); backward=None means "iff train";
        grad_weight scales this batch's gradient contribution (data-parallel ranks holding unequal slices of a global
        batch weight theirs by n_local * world / n_global; 1.0 in every evenly divided batch)."""
        if train:
            model.train()
        else:
            model.eval()
        do_bwd = train if backward is None else bool(backward)
        dev = model.device
        X = _C.dev_f32(batch.X.to(dev), "batch.X")
        timesteps = batch.timesteps
        if timesteps is None:
            # t ~ U[eps, T] per sample, drawn on X's device like the reference (losses.py:59-63: torch.rand(..., device=X.device)).
            # A host-side draw + .to(device) is a blocking copy from pageable memory: it stalled the host behind the previous
            # optimizer step in every iteration, so the GPU idled through the next step's first ~35 launches (~0.2 ms per step).
            # (uniform_(eps, T) is rand * (T - eps) + eps in one kernel instead of three)
            timesteps = torch.empty(X.shape[0], device=dev).uniform_(scheduler.eps, scheduler.T)
        timesteps = _C.dev_f32(timesteps.to(dev), "timesteps")
        x_noisy, target, std = scheduler.perturb(X, timesteps, noise=noise)
        if train and do_bwd and hasattr(model, "train_dsm"):
            # forward + loss + backward as one engine call where the model has it (bf16 transformer training path)
            fused = model.train_dsm(x_noisy, timesteps, target, std, likelihood_weighting=likelihood_weighting,
                                    grad_weight=grad_weight)
            if fused is not None:
                return fused
        if train:
            score = model(DiffusableBatch(X=x_noisy, y=batch.y, timesteps=timesteps))
        else:
            # The validation loss drives ModelCheckpoint's best-model choice and the two-decimal checkpoint name: it is
            # evaluated with the exact-f32 kernels whatever `model.precision` says (bf16 MFMA is for sampling).
            saved = getattr(model, "precision", None)
            if saved is not None:
                model.precision = getattr(model, "eval_loss_precision", "fp32")
            try:
                score = model(DiffusableBatch(X=x_noisy, y=batch.y, timesteps=timesteps))
            finally:
                if saved is not None:
                    model.precision = saved
        B, T, Cn = X.shape
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        dscore = torch.empty_like(score) if do_bwd else None
        h = _C.ctx(dev)
        rc = _C.lib().fd_dsm_loss(h, score.data_ptr(), target.data_ptr(), std.data_ptr(),
                                  1 if likelihood_weighting else 0, loss.data_ptr(), _C.ptr(dscore), B, T, Cn,
                                  _C.stream_of(score))
        _C.check(rc, h)
        if do_bwd:
            if grad_weight != 1.0:
                dscore.mul_(float(grad_weight))      # only on unevenly divided (last) batches: off the steady-state path
            model.backward(dscore, accumulate=True)
        return loss[0]

    return loss_fn
