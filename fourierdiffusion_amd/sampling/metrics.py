"""Evaluation metrics -- same surface as fdiff.sampling.metrics (reference: src/fdiff/sampling/metrics.py:13-217):
`Metric`, `MetricCollection`, `SlicedWasserstein`, `MarginalWasserstein`, same constructor arguments, same result keys.
The distances run on the HIP engine (utils/wasserstein.py); results are plain Python floats / lists as in the reference."""
from __future__ import annotations

from abc import ABC, abstractmethod
from functools import partial
from typing import Any, Optional

import numpy as np
import torch

from ..utils.fourier import dft, spectral_density
from ..utils.tensors import check_flat_array
from ..utils.wasserstein import WassersteinDistances


class Metric(ABC):
    def __init__(self, original_samples: np.ndarray | torch.Tensor) -> None:
        self.original_samples = check_flat_array(original_samples)

    @abstractmethod
    def __call__(self, other_samples: np.ndarray | torch.Tensor) -> dict[str, Any]: ...

    @property
    @abstractmethod
    def name(self) -> str: ...

    @property
    def baseline_metrics(self) -> dict[str, float]:
        return {}


def _as_tensor(x) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x


class MetricCollection:
    """metrics.py:28-99: every metric is evaluated on the samples (time_*) and on their dft (freq_*), optionally a marginal
    Wasserstein on the spectral densities (spectral_*), plus the metrics' baselines."""

    def __init__(self, metrics: list, original_samples: Optional[np.ndarray | torch.Tensor] = None,
                 include_baselines: bool = True, include_spectral_density: bool = False) -> None:
        metrics_time: list[Metric] = []
        metrics_freq: list[Metric] = []
        if original_samples is not None:
            original_samples = _as_tensor(original_samples)
        original_samples_freq = dft(original_samples) if original_samples is not None else None
        for metric in metrics:
            if isinstance(metric, partial):                   # partially instantiated: bind the original samples
                assert original_samples is not None, "Original samples must be provided for the metrics to be instantiated."
                metrics_time.append(metric(original_samples=original_samples))
                metrics_freq.append(metric(original_samples=original_samples_freq))
        self.metrics_time = metrics_time
        self.metrics_freq = metrics_freq
        self.include_baselines = include_baselines
        self.metric_spectral = (MarginalWasserstein(original_samples=spectral_density(original_samples), random_seed=42,
                                                    save_all_distances=True) if include_spectral_density else None)

    def __call__(self, other_samples: np.ndarray | torch.Tensor) -> dict[str, Any]:
        other_samples = _as_tensor(other_samples)
        metric_dict: dict[str, Any] = {}
        other_samples_freq = dft(other_samples)
        for metric_time, metric_freq in zip(self.metrics_time, self.metrics_freq):
            metric_dict.update({f"time_{k}": v for k, v in metric_time(other_samples).items()})
            metric_dict.update({f"freq_{k}": v for k, v in metric_freq(other_samples_freq).items()})
        if self.include_baselines:
            metric_dict.update(self.baseline_metrics)
        if self.metric_spectral is not None:
            metric_dict.update({f"spectral_{k}": v for k, v in self.metric_spectral(spectral_density(other_samples)).items()})
        return dict(sorted(metric_dict.items(), key=lambda item: item[0]))

    @property
    def baseline_metrics(self) -> dict[str, float]:
        metric_dict: dict[str, float] = {}
        for metric_time, metric_freq in zip(self.metrics_time, self.metrics_freq):
            metric_dict.update({f"time_{k}": v for k, v in metric_time.baseline_metrics.items()})
            metric_dict.update({f"freq_{k}": v for k, v in metric_freq.baseline_metrics.items()})
        return metric_dict


class _WassersteinMetric(Metric):
    kind = ""

    def _distances(self, original: torch.Tensor, other: torch.Tensor) -> np.ndarray:
        raise NotImplementedError

    def __call__(self, other_samples: np.ndarray | torch.Tensor) -> dict[str, Any]:
        distances = self._distances(self.original_samples, check_flat_array(other_samples))
        metrics: dict[str, Any] = {f"{self.kind}_wasserstein_mean": float(np.mean(distances)),
                                   f"{self.kind}_wasserstein_max": float(np.max(distances))}
        if self.save_all_distances:
            metrics[f"{self.kind}_wasserstein_all"] = distances.tolist()
        return metrics

    @property
    def baseline_metrics(self) -> dict[str, float]:
        n_samples = self.original_samples.shape[0]
        # two folds of the original samples against each other (metrics.py:129-137)
        distances_self = self._distances(self.original_samples[: n_samples // 2].contiguous(),
                                         self.original_samples[n_samples // 2:].contiguous())
        # a generator that only outputs the average sample (metrics.py:139-147); the mean is a host-side reduction
        avg_sample = self.original_samples.mean(dim=0, keepdim=True)
        distances_dummy = self._distances(self.original_samples, avg_sample)
        return {f"{self.kind}_wasserstein_mean_self": float(np.mean(distances_self)),
                f"{self.kind}_wasserstein_max_self": float(np.max(distances_self)),
                f"{self.kind}_wasserstein_mean_dummy": float(np.mean(distances_dummy)),
                f"{self.kind}_wasserstein_max_dummy": float(np.max(distances_dummy))}

    @property
    def name(self) -> str:
        return f"{self.kind}_wasserstein"


class SlicedWasserstein(_WassersteinMetric):
    """metrics.py:100-160."""
    kind = "sliced"

    def __init__(self, original_samples: np.ndarray | torch.Tensor, random_seed: int, num_directions: int,
                 save_all_distances: bool = False) -> None:
        super().__init__(original_samples=original_samples)
        self.random_seed = random_seed
        self.num_directions = num_directions
        self.save_all_distances = save_all_distances

    def _distances(self, original, other):
        return WassersteinDistances(original_data=original, other_data=other, seed=self.random_seed).sliced_distances(
            self.num_directions)


class MarginalWasserstein(_WassersteinMetric):
    """metrics.py:163-217."""
    kind = "marginal"

    def __init__(self, original_samples: np.ndarray | torch.Tensor, random_seed: int, save_all_distances: bool = False) -> None:
        super().__init__(original_samples=original_samples)
        self.random_seed = random_seed
        self.save_all_distances = save_all_distances

    def _distances(self, original, other):
        return WassersteinDistances(original_data=original, other_data=other, seed=self.random_seed).marginal_distances()
