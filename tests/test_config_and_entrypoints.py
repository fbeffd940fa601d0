"""CPU: the hydra-compatible composer / instantiator and the host logic of the entry points
(reference surface: cmd/conf/**, tests/test_hydra_configs.py:22-51, tests/test_utils.py:12-33)."""
import os
from functools import partial
from pathlib import Path

import pytest
import torch

from fourierdiffusion_amd.config import compose, instantiate, load_yaml, save_yaml, to_container

ROOT = Path(__file__).resolve().parent.parent
CONF = ROOT / "cmd" / "conf"


def test_compose_train_defaults_and_interpolation():
    cfg = compose(CONF, "train", [], cwd="/work")
    assert cfg.random_seed == 42 and cfg.fourier_transform is False and cfg.standardize is True
    sm = cfg.score_model
    assert (sm.d_model, sm.num_layers, sm.n_head) == (72, 10, 12) and sm._partial_ is True
    assert sm._target_ == "fdiff.models.score_models.ScoreModule"
    assert sm.fourier_noise_scaling is False                       # ${fourier_transform}
    ns = sm.noise_scheduler
    assert ns._target_ == "fdiff.schedulers.sde.VPScheduler" and ns.beta_max == 20 and ns.eps == pytest.approx(1e-5)
    assert isinstance(ns.eps, float)                                # hydra reads 1e-5 as a float
    assert ns.fourier_noise_scaling is False                        # ${score_model.fourier_noise_scaling}
    assert cfg.datamodule.data_dir == "/work/data"                  # ${hydra:runtime.cwd}
    assert cfg.trainer.max_epochs == 200 and cfg.trainer.gradient_clip_val == 1.0
    cbs = cfg.trainer.callbacks
    assert isinstance(cbs, list) and len(cbs) == 3                  # a YAML *list* group
    assert cbs[2].sample_batch_size == cfg.datamodule.batch_size    # ${datamodule.batch_size}


def test_overrides_like_the_readme():
    cfg = compose(CONF, "train", ["fourier_transform=true", "score_model/noise_scheduler=vesde",
                                  "datamodule=synthetic", "trainer.max_epochs=3", "score_model.d_model=24",
                                  "+trainer.limit_train_batches=2"])
    assert cfg.score_model.noise_scheduler._target_.endswith("VEScheduler")
    assert cfg.score_model.noise_scheduler.sigma_max == 2
    assert cfg.score_model.fourier_noise_scaling is True and cfg.score_model.noise_scheduler.fourier_noise_scaling is True
    assert cfg.trainer.max_epochs == 3 and cfg.score_model.d_model == 24 and cfg.trainer.limit_train_batches == 2


def test_instantiate_train_and_sample_configs(tmp_path):
    """tests/test_hydra_configs.py of the reference: every top-level config composes and instantiates."""
    cfg = compose(CONF, "train", ["fourier_transform=true"], cwd=str(tmp_path))
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.schedulers.sde import VPScheduler
    from fourierdiffusion_amd.trainer import LearningRateMonitor, ModelCheckpoint, Trainer
    from fourierdiffusion_amd.utils.callbacks import SamplingCallback
    sm = instantiate(cfg.score_model)
    assert isinstance(sm, partial) and sm.func is ScoreModule
    sch = sm.keywords["noise_scheduler"]
    assert isinstance(sch, VPScheduler) and sch.noise_scaling is True and sch.beta_1 == 20
    model = sm(n_channels=2, max_len=16, num_training_steps=100)
    assert model.n_channels == 2 and model.d_model == 72 and model.num_warmup_steps == 10
    tr = instantiate(cfg.trainer)
    assert isinstance(tr, Trainer) and tr.max_epochs == 200 and tr.gradient_clip_val == 1.0
    kinds = [type(c) for c in tr.callbacks]
    assert kinds == [LearningRateMonitor, ModelCheckpoint, SamplingCallback]
    from fourierdiffusion_amd.sampling.metrics import MarginalWasserstein, MetricCollection, SlicedWasserstein
    cb_metrics = tr.callbacks[2].metrics                            # partially instantiated, bound to X_train later
    assert [m.func for m in cb_metrics] == [SlicedWasserstein, MarginalWasserstein]
    assert cb_metrics[0].keywords == {"random_seed": 42, "num_directions": 200}
    dm = instantiate(cfg.datamodule)
    assert dm.fourier_transform is True and dm.batch_size == 64 and dm.dataset_name == "synthetic"
    scfg = compose(CONF, "sample", ["model_id=abc", f"model_path={tmp_path}"])
    assert scfg.num_samples == 10000 and scfg.num_diffusion_steps == 1000 and scfg.model_id == "abc"
    sampler_partial = instantiate(scfg.sampler)
    assert isinstance(sampler_partial, partial) and sampler_partial.keywords["sample_batch_size"] == 200
    mc = instantiate(scfg.metrics)
    assert isinstance(mc, partial) and mc.func is MetricCollection and mc.keywords["include_spectral_density"] is True
    assert [m.func for m in mc.keywords["metrics"]] == [SlicedWasserstein, MarginalWasserstein]
    assert mc.keywords["metrics"][0].keywords["num_directions"] == 1000


def test_yaml_roundtrip(tmp_path):
    cfg = compose(CONF, "train", ["random_seed=7"])
    save_yaml(cfg, tmp_path / "train_config.yaml")
    again = load_yaml(tmp_path / "train_config.yaml")
    assert to_container(again) == to_container(cfg) and again.random_seed == 7


def test_flatten_config_matches_reference_example():
    """tests/test_utils.py:12-33 of the reference."""
    from fourierdiffusion_amd.utils.extraction import flatten_config
    cfg = {"Option1": "Value1", "Option2": {"_target_": "Value2", "Option3": "Value3",
                                           "Option4": {"_target_": "Value4", "Option5": [
                                               {"_target_": "Value5_0"}, {"_target_": "Value5_1"}]}},
           "Option6": "Value6"}
    assert flatten_config(cfg) == {"Option1": "Value1", "Option2": "Value2", "Option3": "Value3", "Option4": "Value4",
                                   "Option5": ["Value5_0", "Value5_1"], "Option6": "Value6"}
    # a genuine user key that is spelled like a list item's event survives (the reference keeps it: its recursion never invents keys)
    assert flatten_config({"layers[0]": "x", "a": [{"_target_": "T0", "b": 1}]}) == {"layers[0]": "x", "b": 1, "a": ["T0"]}


def test_best_checkpoint_and_model_type(tmp_path):
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.utils.extraction import get_best_checkpoint, get_model_type
    for name in ("epoch=3-val_loss=0.52.ckpt", "epoch=7-val_loss=0.31.ckpt", "epoch=9-val_loss=0.40.ckpt", "last.ckpt"):
        (tmp_path / name).write_bytes(b"")
    assert get_best_checkpoint(tmp_path).name == "epoch=7-val_loss=0.31.ckpt"
    assert get_model_type({"score_model": {"_target_": "fdiff.models.score_models.ScoreModule"}}) is ScoreModule
    with pytest.raises(NotImplementedError):
        get_model_type({"score_model": {"_target_": "fdiff.models.score_models.SomethingElse"}})


def test_fdiff_alias_package_resolves_reference_dotted_paths():
    import importlib
    for path in ("fdiff.models.score_models.ScoreModule", "fdiff.schedulers.sde.VPScheduler",
                 "fdiff.schedulers.sde.VEScheduler", "fdiff.sampling.sampler.DiffusionSampler",
                 "fdiff.utils.fourier.dft", "fdiff.utils.fourier.idft", "fdiff.utils.dataclasses.DiffusableBatch",
                 "fdiff.utils.losses.get_sde_loss_fn", "fdiff.dataloaders.datamodules.SyntheticDatamodule",
                 "fdiff.utils.callbacks.SamplingCallback", "fdiff.utils.extraction.get_best_checkpoint"):
        mod, _, attr = path.rpartition(".")
        assert hasattr(importlib.import_module(mod), attr), path


def test_datamodule_statistics_and_sharding(golden):
    """Time-domain dataset logic on the CPU (the DFT itself needs the GPU): train statistics are reused for
    validation (datamodules.py:128-142), loaders yield DiffusableBatch, rank shards are disjoint and complete."""
    from fourierdiffusion_amd.dataloaders.datamodules import BatchLoader, DiffusionDataset, TensorDatamodule
    if torch.cuda.is_available():
        pytest.skip("CPU-side check")
    g = torch.Generator().manual_seed(0)
    Xtr, Xte = torch.randn(50, 20, 3, generator=g) * 2 + 1, torch.randn(10, 20, 3, generator=g)
    dm = TensorDatamodule(Xtr, Xte, batch_size=16, standardize=True)
    mean, std = dm.feature_mean_and_std
    assert torch.allclose(mean, Xtr.mean(0)) and torch.allclose(std, Xtr.std(0))
    assert dm.dataset_parameters == {"n_channels": 3, "max_len": 20, "num_training_steps": 4}
    vb = next(iter(dm.val_dataloader()))
    assert torch.allclose(vb.X, (Xte - mean) / std, atol=1e-6)
    ds = DiffusionDataset(Xtr, standardize=True)
    torch.manual_seed(5)
    whole = torch.cat([b.X for b in BatchLoader(ds, 16, shuffle=True)])
    torch.manual_seed(5)
    r0 = torch.cat([b.X for b in BatchLoader(ds, 16, shuffle=True, rank=0, world=2)])
    torch.manual_seed(5)
    r1 = torch.cat([b.X for b in BatchLoader(ds, 16, shuffle=True, rank=1, world=2)])
    assert r0.shape[0] + r1.shape[0] == 50
    merged = torch.cat([r0, r1]).sort(dim=0).values
    assert torch.allclose(merged, whole.sort(dim=0).values)


def test_mlp_and_lstm_score_model_configs_compose():
    """cmd/conf/score_model/{mlp,lstm}.yaml (reference: cmd/conf/score_model/mlp.yaml, lstm.yaml): same keys and values."""
    from fourierdiffusion_amd.utils.extraction import get_model_type
    for name, target, extra in (("mlp", "fdiff.models.score_models.MLPScoreModule", {"d_mlp": 1024, "lr_max": 1.0e-4}),
                                ("lstm", "fdiff.models.score_models.LSTMScoreModule", {"lr_max": 1.0e-3})):
        cfg = compose(CONF, "train", [f"score_model={name}", "fourier_transform=true"])
        sm = cfg.score_model
        assert sm._target_ == target and sm.d_model == 72 and sm.num_layers == 10 and sm.fourier_noise_scaling is True
        for k, v in extra.items():
            assert sm[k] == v
        assert get_model_type(cfg).__name__ == target.rsplit(".", 1)[1]
