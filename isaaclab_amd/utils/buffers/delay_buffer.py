"""Per-env delayed read of a command stream (``LAB/utils/buffers/delay_buffer.py:15-177``)."""
from __future__ import annotations

from collections.abc import Sequence

import torch

from .circular_buffer import CircularBuffer


class DelayBuffer:
    def __init__(self, history_length: int, batch_size: int, device: str):
        self._history_length = max(0, history_length)
        self._circular_buffer = CircularBuffer(self._history_length + 1, batch_size, device)
        self._min_time_lag = 0
        self._max_time_lag = 0
        self._time_lags = torch.zeros(batch_size, dtype=torch.int, device=device)

    @property
    def batch_size(self) -> int:
        return self._circular_buffer.batch_size

    @property
    def device(self) -> str:
        return self._circular_buffer.device

    @property
    def history_length(self) -> int:
        return self._history_length

    @property
    def min_time_lag(self) -> int:
        return self._min_time_lag

    @property
    def max_time_lag(self) -> int:
        return self._max_time_lag

    @property
    def time_lags(self) -> torch.Tensor:
        return self._time_lags

    def set_time_lag(self, time_lag: int | torch.Tensor, batch_ids: Sequence[int] | None = None):
        if batch_ids is None:
            batch_ids = slice(None)
        if isinstance(time_lag, int):
            self._time_lags[batch_ids] = time_lag
        elif isinstance(time_lag, torch.Tensor):
            if time_lag.dtype not in [torch.int, torch.long]:
                raise TypeError(f"Invalid dtype for time_lag: {time_lag.dtype}. Expected torch.int or torch.long.")
            self._time_lags[batch_ids] = time_lag.to(device=self.device, dtype=torch.int)
        else:
            raise TypeError(f"Invalid type for time_lag: {type(time_lag)}. Expected int or integer tensor.")
        # reset-time validation, same host round trip as the reference (delay_buffer.py:141-149)
        self._min_time_lag = int(torch.min(self._time_lags).item())
        self._max_time_lag = int(torch.max(self._time_lags).item())
        if self._min_time_lag < 0:
            raise ValueError(f"The minimum time lag cannot be negative. Received: {self._min_time_lag}")
        if self._max_time_lag > self._history_length:
            raise ValueError(
                f"The maximum time lag cannot be larger than the history length. Received: {self._max_time_lag}"
            )

    def reset(self, batch_ids: Sequence[int] | None = None):
        self._circular_buffer.reset(batch_ids)

    def compute(self, data: torch.Tensor) -> torch.Tensor:
        return self._circular_buffer.append_and_read(data, self._time_lags)
