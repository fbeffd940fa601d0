"""Batch container of the hot path -- same surface as fdiff.utils.dataclasses
(reference: src/fdiff/utils/dataclasses.py:7-31)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch


@dataclass
class DiffusableBatch:
    X: torch.Tensor
    y: Optional[torch.Tensor] = None
    timesteps: Optional[torch.Tensor] = None

    def __len__(self) -> int:
        return self.X.shape[0]

    @property
    def device(self) -> torch.device:
        return self.X.device

    def to(self, device) -> "DiffusableBatch":
        mv = lambda t: None if t is None else t.to(device)
        return DiffusableBatch(X=self.X.to(device), y=mv(self.y), timesteps=mv(self.timesteps))


def collate_batch(data: List[Dict[str, torch.Tensor]]) -> DiffusableBatch:
    """DataLoader collate: dicts with 'X' (+ optional 'y', 'timestep') -> DiffusableBatch."""
    if "X" not in data[0]:
        raise AssertionError("The construction of a batch requires a 'X' key.")
    stack = lambda key: torch.stack([d[key] for d in data]) if key in data[0] else None
    return DiffusableBatch(X=stack("X"), y=stack("y"), timesteps=stack("timestep"))
