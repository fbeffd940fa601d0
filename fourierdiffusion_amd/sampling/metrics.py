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


class _View:
    """One representation of the samples a collection evaluates in: a key prefix, the map from time-domain samples into the
    representation, and the metrics bound to the ORIGINAL samples in that representation."""

    def __init__(self, prefix: str, transform, metrics: list) -> None:
        self.prefix, self.transform, self.metrics = prefix, transform, metrics

    def _prefixed(self, results) -> dict[str, Any]:
        return {f"{self.prefix}_{key}": val for res in results for key, val in res.items()}

    def evaluate(self, samples: torch.Tensor) -> dict[str, Any]:
        if not self.metrics:
            return {}
        mapped = self.transform(samples)
        return self._prefixed(m(mapped) for m in self.metrics)

    def baselines(self) -> dict[str, Any]:
        return self._prefixed(m.baseline_metrics for m in self.metrics)


class MetricCollection:
    """Same surface and result keys as the reference's collection (metrics.py:28-99): every partially instantiated metric is bound
    to the original samples once per view -- `time` (the samples as they are) and `freq` (their dft) -- and, on request, a marginal
    Wasserstein on the spectral densities forms a third view (`spectral`, seed 42, all distances kept, no baselines).  A call
    returns the union of the views' results (+ the time / freq baselines) sorted by key."""

    def __init__(self, metrics: list, original_samples: Optional[np.ndarray | torch.Tensor] = None,
                 include_baselines: bool = True, include_spectral_density: bool = False) -> None:
        factories = [m for m in metrics if isinstance(m, partial)]      # (like the reference, only partials are taken up)
        if factories and original_samples is None:
            raise AssertionError("Original samples must be provided for the metrics to be instantiated.")
        original = _as_tensor(original_samples) if original_samples is not None else None
        self._views = [_View("time", lambda x: x, []), _View("freq", dft, [])]
        for view in self._views:
            if factories:
                bound_to = view.transform(original)
                view.metrics = [make(original_samples=bound_to) for make in factories]
        self.include_baselines = include_baselines
        self.metric_spectral = None
        self._spectral_view = None
        if include_spectral_density:
            self.metric_spectral = MarginalWasserstein(original_samples=spectral_density(original), random_seed=42,
                                                       save_all_distances=True)
            self._spectral_view = _View("spectral", spectral_density, [self.metric_spectral])

    # the reference's attribute names for the two bound lists
    @property
    def metrics_time(self) -> list:
        return self._views[0].metrics

    @property
    def metrics_freq(self) -> list:
        return self._views[1].metrics

    def __call__(self, other_samples: np.ndarray | torch.Tensor) -> dict[str, Any]:
        samples = _as_tensor(other_samples)
        merged: dict[str, Any] = {}
        for view in self._views:
            merged.update(view.evaluate(samples))
        if self.include_baselines:
            merged.update(self.baseline_metrics)
        if self._spectral_view is not None:
            merged.update(self._spectral_view.evaluate(samples))
        return {key: merged[key] for key in sorted(merged)}

    @property
    def baseline_metrics(self) -> dict[str, float]:
        merged: dict[str, float] = {}
        for view in self._views:
            merged.update(view.baselines())
        return merged


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
