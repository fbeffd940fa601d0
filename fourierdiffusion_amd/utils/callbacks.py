"""SamplingCallback -- same surface as fdiff.utils.callbacks.SamplingCallback (reference:
src/fdiff/utils/callbacks.py:12-89): every N epochs draw samples with the model being trained, de-standardise,
idft, score with the metrics.  `metrics` is the reference's list of partially instantiated metrics
(fdiff.sampling.metrics.SlicedWasserstein / MarginalWasserstein, bound to the training set in `setup_datamodule` through a
MetricCollection without baselines, callbacks.py:33-37); plain callables(X) -> dict are accepted as well.  Simple moment
statistics of the samples are always recorded."""
from __future__ import annotations

from functools import partial
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
        self.metric_collection = None
        self.datamodule_initialized = False

    def setup_datamodule(self, datamodule) -> None:
        self.standardize = datamodule.standardize
        self.fourier_transform = datamodule.fourier_transform
        self.feature_mean, self.feature_std = datamodule.feature_mean_and_std
        partials = [m for m in self.metrics if isinstance(m, partial)]
        if partials:                                          # callbacks.py:33-37
            from ..sampling.metrics import MetricCollection
            self.metric_collection = MetricCollection(metrics=partials, original_samples=datamodule.X_train,
                                                      include_baselines=False)
        self.datamodule_initialized = True

    def on_train_start(self, trainer, model) -> None:
        self.sampler = DiffusionSampler(score_model=model, sample_batch_size=self.sample_batch_size)

    def on_train_epoch_end(self, trainer, model) -> None:
        if trainer.current_epoch % self.every_n_epochs == 0 or trainer.current_epoch + 1 == trainer.max_epochs:
            was_training = model.training
            X = self.sample()
            results: Dict[str, Any] = {"sample_mean": float(X.mean()), "sample_std": float(X.std())}
            if self.metric_collection is not None:
                results.update(self.metric_collection(X))
            for metric in self.metrics:
                if not isinstance(metric, partial):
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
