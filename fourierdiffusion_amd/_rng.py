"""Philox stream bookkeeping for the engine's on-device noise.

The reference draws from torch's global generator (seeded once by ``torch.manual_seed(cfg.random_seed)``,
cmd/train.py:22, cmd/sample.py:22).  Bit-matching torch's CPU/GPU normal generators on the device is
pointless, so the engine owns a counter-based Philox4x32-10 stream -- but it stays slaved to
torch's generator: every engine call that needs noise draws ONE 62-bit key from torch's global CPU
generator (host-side plumbing) and uses counters [rank_base, rank_base + n/4) under that key.
``torch.manual_seed(s)`` therefore reproduces a run exactly as it does for the reference, and
data-parallel ranks (same seed) get disjoint counter ranges through ``set_rank``.
"""
from __future__ import annotations

import torch

_state = {"rank_base": 0, "keys_drawn": 0}


def set_rank(rank: int) -> None:
    """Ranks share the seed; rank r draws counters starting at r << 56."""
    _state["rank_base"] = (int(rank) & 0xFF) << 56


def next_key() -> int:
    """A fresh Philox key from torch's global CPU generator."""
    _state["keys_drawn"] += 1
    return int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item())


def base_offset() -> int:
    return _state["rank_base"]


def stream():
    """(key, offset) for one engine call."""
    return next_key(), _state["rank_base"]


def keys_drawn() -> int:
    """Keys drawn so far in this process (the Trainer measures how many a training step takes, so that a rank whose slice of
    a global batch is empty can discard as many and keep torch's generator in step with the other ranks)."""
    return _state["keys_drawn"]


def discard_keys(n: int) -> None:
    for _ in range(int(n)):
        next_key()
