"""Fused optimiser + LR schedule of the score model (reference: ScoreModule.configure_optimizers,
src/fdiff/models/score_models.py:122-130: AdamW(lr_max, torch defaults) + diffusers'
get_cosine_schedule_with_warmup stepped every batch; Lightning clips the global grad norm at 1.0,
cmd/conf/trainer/default.yaml:4).  One pass over the flat parameter buffer (fd_adamw_step)."""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch

from . import _C


def cosine_schedule_with_warmup(num_warmup_steps: int, num_training_steps: int,
                                num_cycles: float = 0.5) -> Callable[[int], float]:
    """LR multiplier lambda(step) of diffusers.optimization.get_cosine_schedule_with_warmup."""

    def lr_lambda(current_step: int) -> float:
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1, num_warmup_steps))
        progress = float(current_step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))

    return lr_lambda


class FusedAdamW:
    """torch.optim.AdamW semantics (betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2, decoupled decay) on the
    model's flat fp32 parameter buffer; parameters with requires_grad=False in the reference (time_encoder.W)
    are skipped; optional global-norm clipping is fused (norm computed on device, never synced to the host)."""

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: Optional[float] = None):
        self.model = model
        self.lr = lr
        self.base_lr = lr
        self.betas = betas
        self.eps = eps
        self.weight_decay = weight_decay
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        flat = model.flat_parameters
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self._sqnorm = torch.zeros(1, device=flat.device, dtype=torch.float32)
        frozen = [(off, off + numel) for name, off, numel, _, tr in model._layout if not tr]
        assert len(frozen) <= 1, "the engine's AdamW skips a single frozen range"
        self._frozen = frozen[0] if frozen else (0, 0)

    def zero_grad(self) -> None:
        self.model.zero_grad()

    @property
    def grad_sqnorm(self) -> torch.Tensor:
        """device scalar: squared global gradient norm of the last step (before clipping)."""
        return self._sqnorm

    def step(self, grad_scale: float = 1.0) -> None:
        m = self.model
        flat, grads = m.flat_parameters, m.grads
        if grads is None:
            raise _C.FdError("FusedAdamW.step(): no gradients (run a training-mode loss first)")
        if self.exp_avg.device != flat.device:
            self.exp_avg = self.exp_avg.to(flat.device)
            self.exp_avg_sq = self.exp_avg_sq.to(flat.device)
            self._sqnorm = self._sqnorm.to(flat.device)
        self.step_count += 1
        h = _C.ctx(flat.device)
        L = _C.lib()
        stream = _C.stream_of(flat)
        n = flat.numel()
        sq = None
        if self.max_grad_norm is not None:
            _C.check(L.fd_grad_sqnorm(h, grads.data_ptr(), n, self._sqnorm.data_ptr(), stream), h)
            sq = self._sqnorm.data_ptr()
        _C.check(L.fd_adamw_step(h, flat.data_ptr(), grads.data_ptr(), self.exp_avg.data_ptr(),
                                 self.exp_avg_sq.data_ptr(), n, self.step_count, float(self.lr), self.betas[0],
                                 self.betas[1], self.eps, self.weight_decay, sq,
                                 float(self.max_grad_norm or 0.0), float(grad_scale), self._frozen[0],
                                 self._frozen[1], stream), h)
        m.mark_parameters_changed()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(),
                "lr": self.lr}

    def load_state_dict(self, sd) -> None:
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr = float(sd.get("lr", self.lr))
