"""Training loop standing in for `pl.Trainer` (absent from the image) -- the subset the reference configures
(cmd/conf/trainer/default.yaml: max_epochs, gradient_clip_val, callbacks; Lightning defaults otherwise):
per batch  zero_grad -> training_step (engine forward + backward) -> [RCCL all-reduce] -> fused clip+AdamW -> LR step;
per epoch  validation loss with the training statistics, callbacks (checkpoint on val/loss, LR monitor, sampling)."""
from __future__ import annotations

import logging
import math
import os
from pathlib import Path
from typing import Any, Dict, List, Optional

import torch

from . import _rng
from .optim import FusedAdamW
from .parallel import DistEnv, GradExchange, bind_device, init_process_group


class Callback:
    def on_train_start(self, trainer, model) -> None: ...
    def on_train_epoch_end(self, trainer, model) -> None: ...
    def on_validation_end(self, trainer, model) -> None: ...


class LearningRateMonitor(Callback):
    def on_train_epoch_end(self, trainer, model) -> None:
        trainer.logged["lr-AdamW"] = trainer.optimizer.lr


class ModelCheckpoint(Callback):
    """Keeps the best checkpoint by `monitor` under <default_root_dir>/checkpoints, named like Lightning's
    `epoch={epoch}-val_loss={val/loss:.2f}.ckpt` (cmd/conf/trainer/callbacks/default.yaml:2-5), the name
    `get_best_checkpoint` parses."""

    def __init__(self, monitor: str = "val/loss", filename: str = "epoch={epoch}-val_loss={val/loss:.2f}",
                 auto_insert_metric_name: bool = False, dirpath: Optional[str] = None, save_top_k: int = 1) -> None:
        self.monitor, self.filename, self.dirpath = monitor, filename, dirpath
        self.best_score = math.inf
        self.best_model_path: Optional[str] = None

    def on_validation_end(self, trainer, model) -> None:
        if not trainer.dist.is_main or self.monitor not in trainer.logged:
            return
        score = float(trainer.logged[self.monitor])
        if score >= self.best_score:
            return
        d = Path(self.dirpath) if self.dirpath else Path(trainer.default_root_dir) / "checkpoints"
        os.makedirs(d, exist_ok=True)
        name = self.filename.replace("{epoch}", str(trainer.current_epoch))
        name = name.replace("{val/loss:.2f}", f"{score:.2f}")
        path = d / f"{name}.ckpt"
        model.save_checkpoint(path, epoch=trainer.current_epoch, global_step=trainer.global_step,
                              optimizer_state=trainer.optimizer.state_dict())
        if self.best_model_path and os.path.exists(self.best_model_path) and str(path) != self.best_model_path:
            os.remove(self.best_model_path)
        self.best_score, self.best_model_path = score, str(path)


class Trainer:
    def __init__(self, accelerator: str = "auto", max_epochs: int = 200, gradient_clip_val: Optional[float] = None,
                 enable_progress_bar: bool = True, logger: Any = None, callbacks: Optional[List[Callback]] = None,
                 accumulate_grad_batches: int = 1, default_root_dir: Optional[str] = None,
                 grad_exchange: str = "rccl", log_every_n_steps: int = 50, limit_train_batches: Optional[int] = None,
                 **unused: Any) -> None:
        self.max_epochs = max_epochs
        self.gradient_clip_val = gradient_clip_val
        self.accumulate_grad_batches = accumulate_grad_batches
        self.callbacks: List[Callback] = list(callbacks or [])
        self.logger = logger
        self.default_root_dir = default_root_dir or os.getcwd()
        self.grad_exchange_backend = grad_exchange
        self.log_every_n_steps = log_every_n_steps
        self.limit_train_batches = limit_train_batches
        self.enable_progress_bar = enable_progress_bar
        self.current_epoch = 0
        self.global_step = 0
        self.logged: Dict[str, float] = {}
        self.history: List[Dict[str, float]] = []
        self.dist = DistEnv()
        self.optimizer: Optional[FusedAdamW] = None
        self._keys_per_step = 2                 # perturbation + dropout key; re-measured on every non-empty step

    # ------------------------------------------------------------------
    def fit(self, model, datamodule) -> None:
        self.dist = init_process_group()
        _rng.set_rank(self.dist.rank)
        if torch.cuda.is_available():
            model.to(torch.device("cuda", bind_device()))
        datamodule.set_shard(self.dist.rank, self.dist.world)
        exchange = GradExchange(self.dist, backend=self.grad_exchange_backend)
        opt_cfg = model.configure_optimizers()
        self.optimizer = opt_cfg["optimizer"]
        self.optimizer.max_grad_norm = self.gradient_clip_val
        lr_lambda = opt_cfg["lr_scheduler"]["scheduler"]
        for cb in self.callbacks:
            cb.on_train_start(self, model)
        sched_step = 0
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            # ---- train
            losses = []
            model.zero_grad()
            pending = 0                                            # micro-batches accumulated since the last optimizer step

            def optimizer_step() -> None:
                nonlocal sched_step, pending
                exchange.all_reduce_mean(model.grads)
                self.optimizer.lr = self.optimizer.base_lr * lr_lambda(sched_step)
                self.optimizer.step(grad_scale=1.0 / self.accumulate_grad_batches)
                sched_step += 1                                    # LambdaLR stepped per optimizer step (interval: step)
                self.global_step += 1
                model.zero_grad()
                pending = 0

            for bi, batch in enumerate(datamodule.train_dataloader()):
                if self.limit_train_batches is not None and bi >= self.limit_train_batches:
                    break
                # Every rank sees every global batch (possibly an EMPTY slice of a small last batch) and joins every
                # all-reduce; a rank's contribution is weighted by its share n_local * world / n_global, so the exchange
                # (sum / world) yields the mean over the global batch whatever the slice sizes.
                n_local = len(batch)
                n_global = int(getattr(batch, "global_size", n_local * self.dist.world))
                weight = n_local * self.dist.world / max(1, n_global)
                if n_local > 0:
                    k0 = _rng.keys_drawn()
                    loss = model.training_step(batch, bi, grad_weight=weight)   # forward + backward inside the engine
                    self._keys_per_step = _rng.keys_drawn() - k0
                    losses.append(loss * weight)
                else:
                    # the other ranks draw Philox keys (perturbation noise, dropout) from torch's global CPU generator in
                    # this step: draw and discard as many, or the generators -- seeded alike on every rank -- drift apart
                    # and everything derived from them later (sampling callbacks, a re-seeded shuffle) differs per rank
                    _rng.discard_keys(self._keys_per_step)
                    if model.grads is None:
                        model.grads = torch.zeros_like(model.flat_parameters)
                    losses.append(torch.zeros((), device=model.device))
                pending += 1
                if pending == self.accumulate_grad_batches:
                    optimizer_step()
            if pending:                                            # Lightning steps on the last batch of an epoch even when
                optimizer_step()                                   # the accumulation window is not full
            train_loss = float(torch.stack(losses).mean().item()) if losses else float("nan")
            self.logged["train/loss"] = exchange.all_reduce_scalar_mean(train_loss)
            for cb in self.callbacks:
                cb.on_train_epoch_end(self, model)
            # ---- validate (every rank evaluates the full validation set: no exchange needed)
            vals, weights = [], []
            for bi, batch in enumerate(datamodule.val_dataloader()):
                vals.append(model.validation_step(batch, bi))
                weights.append(len(batch))
            if vals:
                w = torch.tensor(weights, dtype=torch.float64)
                v = torch.stack([x.double().cpu() for x in vals])
                self.logged["val/loss"] = float((v * w).sum() / w.sum())
            for cb in self.callbacks:
                cb.on_validation_end(self, model)
            self.history.append(dict(self.logged, epoch=epoch))
            if self.dist.is_main and self.enable_progress_bar:
                logging.info("epoch %d  train/loss %.5f  val/loss %.5f  lr %.2e", epoch, self.logged["train/loss"],
                             self.logged.get("val/loss", float("nan")), self.optimizer.lr)
        model.eval()
