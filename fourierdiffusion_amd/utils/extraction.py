"""Entry-point helpers with the public names of fdiff.utils.extraction (reference: src/fdiff/utils/extraction.py:12-121): what
cmd/train.py / cmd/sample.py import from there.  Own implementation: one iterative walker over the config tree serves both the
flat logging view and the pretty-printer."""
from __future__ import annotations

import re
from pathlib import Path
from typing import Any, Dict, Iterator, Tuple

_META_KEYS = ("_target_", "_partial_")
_CKPT_NAME = re.compile(r"(.+?)epoch=(\d+)-val_loss=(\d+\.\d+).ckpt")      # the file-name contract of the checkpoint callback
_MODEL_TARGETS = {"fdiff.models.score_models.ScoreModule": "ScoreModule",
                  "fdiff.models.score_models.MLPScoreModule": "MLPScoreModule",
                  "fdiff.models.score_models.LSTMScoreModule": "LSTMScoreModule"}


def get_training_params(datamodule, trainer) -> Dict[str, Any]:
    """Dataset parameters with num_training_steps scaled to optimizer steps of the whole run: batches per epoch x epochs /
    accumulation (true division, a float, like the reference :12-17)."""
    params = datamodule.dataset_parameters
    params["num_training_steps"] = params["num_training_steps"] * trainer.max_epochs / trainer.accumulate_grad_batches
    return params


def _plain(node):
    """Mapping-like config nodes (this package's Config, an OmegaConf node, a dict) -> plain dict; everything else unchanged."""
    if isinstance(node, dict):
        return node
    if hasattr(node, "items") and hasattr(node, "keys"):
        return dict(node.items())
    return node


class _ItemKey(str):
    """Key of a list ITEM's event (`name[i]`, made up by _walk): marked by type, not by spelling, so that a genuine user key that
    happens to look like one (`layers[0]: x`) is never mistaken for it."""


def _walk(cfg) -> Iterator[Tuple[str, Any]]:
    """Depth-first (key, value) events of a config tree in document order.  A mapping that names a `_target_` reports that name
    under its own key before its children; a list reports the `_target_` names of its mapping items (a list) after their children;
    scalars report themselves; the hydra meta keys never report."""
    stack = [iter(_plain(cfg).items())]
    pending = []                                   # (depth, key, names) of the lists being walked
    while stack:
        try:
            key, val = next(stack[-1])
        except StopIteration:
            stack.pop()
            while pending and pending[-1][0] > len(stack):
                _, k, names = pending.pop()
                yield k, names
            continue
        val = _plain(val)
        if isinstance(val, dict):
            if "_target_" in val:
                yield key, val["_target_"]
            stack.append(iter(val.items()))
        elif isinstance(val, (list, tuple)):
            items = [_plain(v) for v in val]
            maps = [v for v in items if isinstance(v, dict)]
            names = [v["_target_"] for v in maps if "_target_" in v]
            # children of every mapping item first, then the list's own entry
            stack.append(iter([(_ItemKey(f"{key}[{i}]"), v) for i, v in enumerate(maps)]))
            pending.append((len(stack), key, names))
        elif key not in _META_KEYS:
            yield key, val


def flatten_config(cfg) -> Dict[str, Any]:
    """Flat {key: value} view of a nested config for logging: later keys win, `_target_` names stand for their sub-trees
    (same result as the reference's recursive version :20-55, e.g. its tests/test_utils.py example)."""
    flat: Dict[str, Any] = {}
    for key, val in _walk(cfg):
        if isinstance(key, _ItemKey):
            continue                               # the `_target_` of a list item is reported through the list's own entry
        flat[key] = val
    return flat


def get_model_type(cfg):
    """The score-model class cfg.score_model._target_ names (reference paths, :58-76)."""
    from ..models import score_models
    target = cfg["score_model"]["_target_"]
    if target not in _MODEL_TARGETS:
        raise NotImplementedError(f"Model class {target} not implemented yet.")
    return getattr(score_models, _MODEL_TARGETS[target])


def get_best_checkpoint(checkpoint_path: Path) -> Path:
    """The `epoch=E-val_loss=L.ckpt` file with the lowest L, read from the FILE NAME (2 decimals; :79-98)."""
    scored = []
    for ckpt in Path(checkpoint_path).glob("*.ckpt"):
        hit = _CKPT_NAME.match(str(ckpt))
        if hit:
            scored.append((float(hit.group(3)), len(scored), ckpt))
    if not scored:
        raise FileNotFoundError(f"no checkpoint named epoch=E-val_loss=L.ckpt in {checkpoint_path}")
    return min(scored)[2]


def dict_to_str(d) -> str:
    """One `key : value` line per entry of the flat view, keys padded to a common width, long lists cut to three items."""
    top = _plain(d)
    # a config tree is shown through its flat view; a flat result dict (metrics: scalar lists are data) is shown as it is
    flat = flatten_config(top) if any(isinstance(_plain(v), dict) for v in top.values()) else dict(top)
    if not flat:
        return ""
    pad = max(map(len, flat)) + 5

    def show(v):
        return v if not (isinstance(v, list) and len(v) > 3) else list(v[:3]) + ["..."]

    return "".join("\t {} : \t  {} \t \n".format(k.ljust(pad), show(v)) for k, v in flat.items())
