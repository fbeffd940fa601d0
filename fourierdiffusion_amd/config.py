"""Minimal hydra-compatible config composer + instantiator (hydra / omegaconf are absent from the image).

Supports what the reference's config tree uses (cmd/conf/**, SURVEY.md 5 "config / flags"):
  * root config with a `defaults:` list (`_self_`, `group: option`), nested defaults inside group files
    (score_model/default.yaml -> noise_scheduler: vpsde; trainer/default.yaml -> callbacks: default, a YAML list);
  * CLI overrides `a.b=value`, group overrides `datamodule=synthetic`, `score_model/noise_scheduler=vesde`,
    `+new.key=value`;
  * `${a.b}` interpolation from the root and `${hydra:runtime.cwd}`;
  * `instantiate`: `_target_` dotted paths, `_partial_: true`, recursion into dicts and lists.
Third-party `_target_`s the reference names (pytorch_lightning.*) are mapped onto this package's stand-ins.
"""
from __future__ import annotations

import copy
import functools
import importlib
import os
import re
from pathlib import Path
from typing import Any, Dict, List, Optional

import yaml

TARGET_ALIASES = {
    "pytorch_lightning.Trainer": "fourierdiffusion_amd.trainer.Trainer",
    "pytorch_lightning.callbacks.LearningRateMonitor": "fourierdiffusion_amd.trainer.LearningRateMonitor",
    "pytorch_lightning.callbacks.ModelCheckpoint": "fourierdiffusion_amd.trainer.ModelCheckpoint",
    "pytorch_lightning.loggers.WandbLogger": "fourierdiffusion_amd.config.NullLogger",
}
# `_target_` prefixes that instantiate to None (nothing is skipped any more: the Wasserstein metrics run on the engine)
SKIPPED_TARGET_PREFIXES: tuple = ()


class NullLogger:
    """Stand-in for pytorch_lightning.loggers.WandbLogger (SaaS logger, no network)."""

    def __init__(self, *a: Any, **k: Any) -> None:
        pass


class Config(dict):
    """dict with attribute access (cfg.score_model.d_model)."""

    def __getattr__(self, k: str) -> Any:
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k: str, v: Any) -> None:
        self[k] = v


def _wrap(x: Any) -> Any:
    if isinstance(x, dict):
        return Config({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def to_container(cfg: Any) -> Any:
    if isinstance(cfg, dict):
        return {k: to_container(v) for k, v in cfg.items()}
    if isinstance(cfg, list):
        return [to_container(v) for v in cfg]
    return cfg


_SCI = re.compile(r"^[+-]?\d+(\.\d*)?[eE][+-]?\d+$")


def _coerce(x: Any) -> Any:
    """PyYAML reads `1e-5` (no dot) as a string; OmegaConf / hydra read it as a float -- follow hydra."""
    if isinstance(x, dict):
        return {k: _coerce(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_coerce(v) for v in x]
    if isinstance(x, str) and _SCI.match(x):
        return float(x)
    return x


def _load_yaml(path: Path) -> Any:
    with open(path) as f:
        return _coerce(yaml.safe_load(f) or {})


def _merge(dst: Dict[str, Any], src: Dict[str, Any]) -> None:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


def _load_group(conf_dir: Path, group: str, option: str, choices: Dict[str, str]) -> Any:
    """Load <conf_dir>/<group>/<option>.yaml and resolve its own (relative) defaults list."""
    option = choices.get(group, option)
    node = _load_yaml(conf_dir / group / f"{option}.yaml")
    if isinstance(node, dict) and "defaults" in node:
        defaults = node.pop("defaults")
        body = node
        node = {}
        for d in defaults:
            if d == "_self_":
                _merge(node, body)
                body = None
            else:
                (sub, opt), = d.items()
                node[sub] = _load_group(conf_dir, f"{group}/{sub}", opt, choices)
        if body:
            _merge(node, body)
    return node


def _set_path(cfg: Dict[str, Any], dotted: str, value: Any) -> None:
    keys = dotted.split(".")
    cur: Any = cfg
    for k in keys[:-1]:
        cur = cur[int(k)] if isinstance(cur, list) else cur.setdefault(k, {})     # list index: trainer.callbacks.2.x
    if isinstance(cur, list):
        cur[int(keys[-1])] = value
    else:
        cur[keys[-1]] = value


def _get_path(cfg: Any, dotted: str) -> Any:
    cur = cfg
    for k in dotted.split("."):
        cur = cur[int(k)] if isinstance(cur, list) else cur[k]
    return cur


_INTERP = re.compile(r"\$\{([^}]+)\}")


def _resolve(node: Any, root: Any, cwd: str, depth: int = 0) -> Any:
    if depth > 20:
        raise ValueError("interpolation cycle")
    if isinstance(node, dict):
        return {k: _resolve(v, root, cwd, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, cwd, depth) for v in node]
    if isinstance(node, str) and "${" in node:
        def lookup(expr: str) -> Any:
            if expr == "hydra:runtime.cwd":
                return cwd
            return _resolve(_get_path(root, expr), root, cwd, depth + 1)
        whole = _INTERP.fullmatch(node)
        if whole:
            return lookup(whole.group(1))
        return _INTERP.sub(lambda m: str(lookup(m.group(1))), node)
    return node


def compose(config_dir: str | Path, config_name: str, overrides: Optional[List[str]] = None,
            cwd: Optional[str] = None) -> Config:
    conf_dir = Path(config_dir)
    overrides = list(overrides or [])
    choices: Dict[str, str] = {}
    values: List[tuple] = []
    for ov in overrides:
        key, _, val = ov.partition("=")
        key = key.lstrip("+")
        if (conf_dir / key).is_dir():
            choices[key] = val
        else:
            values.append((key, _coerce(yaml.safe_load(val))))
    root_node = _load_yaml(conf_dir / f"{config_name}.yaml")
    defaults = root_node.pop("defaults", ["_self_"])
    cfg: Dict[str, Any] = {}
    body: Optional[Dict[str, Any]] = root_node
    for d in defaults:
        if d == "_self_":
            _merge(cfg, body or {})
            body = None
        else:
            (group, opt), = d.items()
            cfg[group] = _load_group(conf_dir, group, opt, choices)
    if body:
        _merge(cfg, body)
    for key, val in values:
        _set_path(cfg, key, val)
    return _wrap(_resolve(cfg, cfg, cwd or os.getcwd()))


def save_yaml(cfg: Any, path: str | Path) -> None:
    with open(path, "w") as f:
        yaml.safe_dump(to_container(cfg), f, sort_keys=False)


def load_yaml(path: str | Path) -> Config:
    return _wrap(_load_yaml(Path(path)))


def _locate(target: str) -> Any:
    target = TARGET_ALIASES.get(target, target)
    module, _, attr = target.rpartition(".")
    return getattr(importlib.import_module(module), attr)


def instantiate(node: Any, **extra: Any) -> Any:
    if isinstance(node, list):
        out = [instantiate(v) for v in node]
        return [v for v in out if v is not None]
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return _wrap({k: instantiate(v) for k, v in node.items()})
    target = node["_target_"]
    if SKIPPED_TARGET_PREFIXES and target.startswith(SKIPPED_TARGET_PREFIXES):
        return None
    kwargs = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_partial_")}
    kwargs.update(extra)
    fn = _locate(target)
    if node.get("_partial_", False):
        return functools.partial(fn, **kwargs)
    return fn(**kwargs)
