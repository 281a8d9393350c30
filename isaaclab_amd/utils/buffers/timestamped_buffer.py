"""``LAB/utils/buffers/timestamped_buffer.py:11-28``."""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class TimestampedBuffer:
    data: torch.Tensor = None  # type: ignore
    timestamp: float = -1.0
