"""Entry-point helpers -- same behaviour as fdiff.utils.extraction (reference: src/fdiff/utils/extraction.py:12-121)."""
from __future__ import annotations

import re
from pathlib import Path
from typing import Any, Dict


def get_training_params(datamodule, trainer) -> Dict[str, Any]:
    """n_channels, max_len and num_training_steps = len(train_loader) * max_epochs / accumulate (a float, :12-17)."""
    params = datamodule.dataset_parameters
    params["num_training_steps"] *= trainer.max_epochs
    params["num_training_steps"] /= trainer.accumulate_grad_batches
    return params


def flatten_config(cfg) -> Dict[str, Any]:
    """Nested config -> flat dict for logging; `_target_` of a sub-dict becomes the value of its key (:20-55)."""
    out: Dict[str, Any] = {}
    for k, v in dict(cfg).items():
        if isinstance(v, dict):
            if "_target_" in v:
                out[k] = v["_target_"]
            out.update(**flatten_config(v))
        elif isinstance(v, list):
            names = []
            for item in v:
                if isinstance(item, dict):
                    if "_target_" in item:
                        names.append(item["_target_"])
                    out.update(**flatten_config(item))
            out[k] = names
        elif k not in {"_target_", "_partial_"}:
            out[k] = v
    return out


def get_model_type(cfg):
    """Model class named by cfg.score_model._target_ (:58-76)."""
    from ..models.score_models import LSTMScoreModule, MLPScoreModule, ScoreModule
    model_class = cfg["score_model"]["_target_"]
    if model_class == "fdiff.models.score_models.ScoreModule":
        return ScoreModule
    if model_class == "fdiff.models.score_models.MLPScoreModule":
        return MLPScoreModule
    if model_class == "fdiff.models.score_models.LSTMScoreModule":
        return LSTMScoreModule
    raise NotImplementedError(f"Model class {model_class} not implemented yet.")


def get_best_checkpoint(checkpoint_path: Path) -> Path:
    """Lowest val_loss among `epoch=E-val_loss=L.ckpt` files, by the 2-decimal loss in the FILE NAME (:79-98)."""
    pattern = r"(.+?)epoch=(\d+)-val_loss=(\d+\.\d+).ckpt"
    best_loss, best = float("inf"), None
    for ckpt in Path(checkpoint_path).glob("*.ckpt"):
        mt = re.match(pattern, str(ckpt))
        if mt is not None and float(mt.group(3)) < best_loss:
            best_loss, best = float(mt.group(3)), ckpt
    if best is None:
        raise FileNotFoundError(f"no checkpoint named epoch=E-val_loss=L.ckpt in {checkpoint_path}")
    return best


def dict_to_str(d) -> str:
    d = flatten_config(d) if any(isinstance(v, (dict, list)) for v in dict(d).values()) else dict(d)
    width = max(len(k) for k in d)
    lines = []
    for k, v in d.items():
        if isinstance(v, list) and len(v) > 3:
            v = v[:3] + ["..."]
        lines.append(f"\t {k: <{width + 5}} : \t  {v} \t \n")
    return "".join(lines)
