#!/usr/bin/env python3
"""python cmd/sample.py model_id=XYZ ... -- sample from a trained score model (reference: cmd/sample.py:18-100).

Loads lightning_logs/<model_id>/train_config.yaml + the best checkpoint, draws num_samples series with
num_diffusion_steps reverse-diffusion steps on the MI355X engine, de-standardises, maps back to the time domain
(one fused kernel) and writes samples.pt + results.yaml next to the checkpoint.  With several processes
(torch.distributed.run) the sample batches are sharded over the ranks (no collective on the data path)."""
from __future__ import annotations

import logging
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import yaml  # noqa: E402

from fourierdiffusion_amd import _rng  # noqa: E402
from fourierdiffusion_amd.config import compose, instantiate, load_yaml, save_yaml  # noqa: E402
from fourierdiffusion_amd.parallel import bind_device, env, init_process_group, shard_range  # noqa: E402
from fourierdiffusion_amd.utils.extraction import dict_to_str, get_best_checkpoint, get_model_type  # noqa: E402
from fourierdiffusion_amd.utils.fourier import destandardize_idft, idft  # noqa: E402


class SamplingRunner:
    def __init__(self, cfg) -> None:
        self.random_seed: int = cfg.random_seed
        torch.manual_seed(self.random_seed)
        logging.info(f"Welcome in the sampling script! You are using the following config:\n{dict_to_str(cfg)}")
        self.dist = init_process_group()
        _rng.set_rank(self.dist.rank)
        self.dev_index = bind_device()
        self.model_path = Path(cfg.model_path)
        self.model_id = cfg.model_id
        if self.model_id == "latest":
            runs = sorted(p for p in self.model_path.iterdir() if (p / "train_config.yaml").exists())
            self.model_id = runs[-1].name
        self.save_dir = self.model_path / self.model_id
        if self.dist.is_main:
            save_yaml(cfg, self.save_dir / "sample_config.yaml")
        train_cfg = load_yaml(self.save_dir / "train_config.yaml")
        self.datamodule = instantiate(train_cfg.datamodule)
        self.fourier_transform: bool = self.datamodule.fourier_transform
        self.datamodule.prepare_data()
        self.datamodule.setup()
        self.num_samples: int = cfg.num_samples
        self.num_diffusion_steps: int = cfg.num_diffusion_steps
        best_checkpoint_path = get_best_checkpoint(self.save_dir / "checkpoints")
        model_type = get_model_type(train_cfg)
        self.score_model = model_type.load_from_checkpoint(checkpoint_path=best_checkpoint_path)
        self.score_model.to(device=torch.device("cuda", self.dev_index))
        self.sampler = instantiate(cfg.sampler)(score_model=self.score_model)
        # metrics against the training set, on the main rank only (reference cmd/sample.py:62-65)
        self.metrics = None
        if "metrics" in cfg and cfg.metrics is not None and self.dist.is_main:
            self.metrics = instantiate(cfg.metrics)(original_samples=self.datamodule.X_train)

    def sample(self) -> None:
        bs = self.sampler.sample_batch_size
        num_batches = max(1, self.num_samples // bs)
        lo, hi = shard_range(num_batches, self.dist.rank, self.dist.world)       # independent units: no exchange
        n_local = (hi - lo) * min(bs, self.num_samples)
        X = self.sampler.sample(num_samples=n_local, num_diffusion_steps=self.num_diffusion_steps) if n_local else None
        if X is not None:
            if self.datamodule.standardize:
                feature_mean, feature_std = self.datamodule.feature_mean_and_std
                if self.fourier_transform:
                    X = destandardize_idft(X, feature_mean, feature_std)            # cmd/sample.py:76-82 fused
                else:
                    X = X * feature_std.cpu() + feature_mean.cpu()
            elif self.fourier_transform:
                X = idft(X)
        if self.dist.world > 1:
            import torch.distributed as dist
            parts = [None] * self.dist.world
            dist.all_gather_object(parts, X)                                        # host-side gather of the results
            X = torch.cat([p for p in parts if p is not None], dim=0)
        if self.dist.is_main:
            results = {"num_samples": int(X.shape[0]), "sample_mean": float(X.mean()), "sample_std": float(X.std())}
            if self.metrics is not None:
                results.update(self.metrics(X))                                     # reference cmd/sample.py:84-86
            logging.info(f"Saving samples ands metrics to {self.save_dir}.\n{dict_to_str(results)}")
            yaml.dump(data=results, stream=open(self.save_dir / "results.yaml", "w"))
            torch.save(X, self.save_dir / "samples.pt")


def main(argv=None) -> None:
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s] %(message)s")
    cfg = compose(Path(__file__).parent / "conf", "sample", overrides=list(sys.argv[1:] if argv is None else argv))
    SamplingRunner(cfg).sample()


if __name__ == "__main__":
    main()
