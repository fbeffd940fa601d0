#!/usr/bin/env python3
"""python cmd/train.py key=value ... -- train a score model (same command line as the reference's cmd/train.py:66-69).

Differences from the reference runner (cmd/train.py:19-63): hydra/Lightning/wandb are replaced by the bundled
composer and trainer; the run id is a timestamp instead of a wandb id; multi-GPU = one process per GPU
(`python -m torch.distributed.run --nproc-per-node N cmd/train.py ...`) with a flat RCCL gradient all-reduce."""
from __future__ import annotations

import logging
import os
import sys
import time
from functools import partial
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from fourierdiffusion_amd.config import compose, instantiate, save_yaml  # noqa: E402
from fourierdiffusion_amd.parallel import bind_device  # noqa: E402
from fourierdiffusion_amd.utils.callbacks import SamplingCallback  # noqa: E402
from fourierdiffusion_amd.utils.extraction import dict_to_str, get_training_params  # noqa: E402


class TrainingRunner:
    def __init__(self, cfg) -> None:
        torch.manual_seed(cfg.random_seed)
        bind_device()         # this rank's GPU, before the datamodule / callbacks place anything on "cuda"
        logging.info(f"Welcome in the training script! You are using the following config:\n{dict_to_str(cfg)}")
        run_id = cfg.get("run_id") or time.strftime("run-%Y%m%d-%H%M%S")
        self.score_model = instantiate(cfg.score_model)
        self.save_dir = Path.cwd() / "lightning_logs" / run_id
        self.trainer = instantiate(cfg.trainer, default_root_dir=str(self.save_dir))
        self.datamodule = instantiate(cfg.datamodule)
        if int(os.environ.get("RANK", "0")) == 0:
            os.makedirs(self.save_dir, exist_ok=True)
            logging.info(f"Saving the config into {self.save_dir}.")
            save_yaml(cfg, self.save_dir / "train_config.yaml")
        self.datamodule.prepare_data()
        self.datamodule.setup("fit")
        if isinstance(self.score_model, partial):
            self.score_model = self.score_model(**get_training_params(self.datamodule, self.trainer))
        for callback in self.trainer.callbacks:
            if isinstance(callback, SamplingCallback):
                callback.setup_datamodule(datamodule=self.datamodule)

    def train(self) -> None:
        assert not (self.score_model.scale_noise and not self.datamodule.fourier_transform), (
            "You cannot use noise scaling without the Fourier transform.")
        self.trainer.fit(model=self.score_model, datamodule=self.datamodule)


def main(argv=None) -> None:
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s] %(message)s")
    cfg = compose(Path(__file__).parent / "conf", "train", overrides=list(sys.argv[1:] if argv is None else argv))
    runner = TrainingRunner(cfg)
    runner.train()
    logging.info(f"run directory: {runner.save_dir}")


if __name__ == "__main__":
    main()
