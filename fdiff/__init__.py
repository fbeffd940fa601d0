"""`fdiff` -- alias of `fourierdiffusion_amd` so that the reference's dotted paths keep working
(hydra `_target_: fdiff.models.score_models.ScoreModule`, `fdiff.schedulers.sde.VPScheduler`,
`fdiff.sampling.sampler.DiffusionSampler`, ... )."""
import importlib
import sys

import fourierdiffusion_amd as _pkg

_SUBMODULES = [
    "utils", "utils.dataclasses", "utils.fourier", "utils.losses", "utils.extraction", "utils.callbacks", "utils.tensors",
    "utils.wasserstein",
    "schedulers", "schedulers.sde",
    "models", "models.score_models", "models.transformer",
    "sampling", "sampling.sampler", "sampling.metrics",
    "dataloaders", "dataloaders.datamodules",
]
for _name in _SUBMODULES:
    try:
        _mod = importlib.import_module(f"fourierdiffusion_amd.{_name}")
    except ModuleNotFoundError:      # optional pieces
        continue
    sys.modules[f"fdiff.{_name}"] = _mod
    _parent, _, _leaf = _name.rpartition(".")
    if not _parent:
        globals()[_leaf] = _mod

__version__ = _pkg.__version__

# Objects pickled while the alias is active name the reference's module paths (Lightning checkpoints keep the noise-scheduler
# INSTANCE in hyper_parameters): a checkpoint written here then unpickles in the reference, and the reference's unpickle here.
for _cls_name in ("SDE", "VPScheduler", "VEScheduler"):
    _cls = getattr(sys.modules["fdiff.schedulers.sde"], _cls_name, None)
    if _cls is not None:
        _cls.__module__ = "fdiff.schedulers.sde"
