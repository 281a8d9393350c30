"""``LAB/utils/modifiers/modifier_base.py``: base of class modifiers."""
from __future__ import annotations

from abc import ABC, abstractmethod
from collections.abc import Sequence

import torch


class ModifierBase(ABC):
    def __init__(self, cfg, data_dim: tuple[int, ...], device: str) -> None:
        self._cfg = cfg
        self._data_dim = tuple(data_dim)
        self._device = device

    @abstractmethod
    def reset(self, env_ids: Sequence[int] | None = None):
        raise NotImplementedError

    @abstractmethod
    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError
