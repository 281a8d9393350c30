"""Data containers shared by assets and actuators (``LAB/utils/types.py:15-39``)."""
from __future__ import annotations

from collections.abc import Sequence
from dataclasses import dataclass

import torch


@dataclass
class ArticulationActions:
    """Joint-level targets handed to ``ActuatorBase.compute`` and returned by it."""

    joint_positions: torch.Tensor | None = None
    joint_velocities: torch.Tensor | None = None
    joint_efforts: torch.Tensor | None = None
    joint_indices: torch.Tensor | Sequence[int] | slice | None = None
