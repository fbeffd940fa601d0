"""SamplingCallback -- same surface as fdiff.utils.callbacks.SamplingCallback (reference:
src/fdiff/utils/callbacks.py:12-89): every N epochs draw samples with the model being trained, de-standardise,
idft, score with the metrics.  The Wasserstein metrics of the reference need POT (absent, out of scope,
SURVEY.md 2 #11): `metrics` may be any callables(X) -> dict; with none configured the callback still samples and
records simple moment statistics, so the sampling path is exercised during training as in the reference."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional

import torch

from ..sampling.sampler import DiffusionSampler
from ..trainer import Callback
from .fourier import destandardize_idft, idft


class SamplingCallback(Callback):
    def __init__(self, every_n_epochs: int, sample_batch_size: int, num_samples: int, num_diffusion_steps: int,
                 metrics: Optional[List[Callable[[torch.Tensor], Dict[str, float]]]] = None) -> None:
        self.every_n_epochs = every_n_epochs
        self.sample_batch_size = sample_batch_size
        self.num_samples = num_samples
        self.num_diffusion_steps = num_diffusion_steps
        self.metrics = [m for m in (metrics or []) if callable(m)]
        self.datamodule_initialized = False

    def setup_datamodule(self, datamodule) -> None:
        self.standardize = datamodule.standardize
        self.fourier_transform = datamodule.fourier_transform
        self.feature_mean, self.feature_std = datamodule.feature_mean_and_std
        self.datamodule_initialized = True

    def on_train_start(self, trainer, model) -> None:
        self.sampler = DiffusionSampler(score_model=model, sample_batch_size=self.sample_batch_size)

    def on_train_epoch_end(self, trainer, model) -> None:
        if trainer.current_epoch % self.every_n_epochs == 0 or trainer.current_epoch + 1 == trainer.max_epochs:
            was_training = model.training
            X = self.sample()
            results: Dict[str, Any] = {"sample_mean": float(X.mean()), "sample_std": float(X.std())}
            for metric in self.metrics:
                results.update(metric(X))
            trainer.logged.update({f"metrics/{k}": v for k, v in results.items()})
            model.train(was_training)

    def sample(self) -> torch.Tensor:
        assert self.datamodule_initialized, (
            "The datamodule has not been initialized. Please call `setup_datamodule` before sampling.")
        X = self.sampler.sample(num_samples=self.num_samples, num_diffusion_steps=self.num_diffusion_steps)
        if self.standardize and self.fourier_transform:
            return destandardize_idft(X, self.feature_mean, self.feature_std)      # one fused kernel
        if self.standardize:
            X = X * self.feature_std.cpu() + self.feature_mean.cpu()
        if self.fourier_transform:
            X = idft(X)
        return X
