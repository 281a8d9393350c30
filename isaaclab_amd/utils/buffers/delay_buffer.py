"""Per-env delayed read of a stream of tensors -- the interface of ``LAB/utils/buffers/delay_buffer.py:15-177`` over a
device-resident design.

A :class:`CircularBuffer` of ``history_length + 1`` slots holds the stream; ``compute(data)`` pushes one sample and
returns, for every env, the sample pushed ``time_lags[env]`` calls ago (one launch: ``append_and_read``).  The lags live
on the device and so does their range check: ``set_time_lag`` with a tensor records "some lag left [0, history_length]"
in a device flag and stores the lags clamped into the ring, so nothing on the step path waits for the host (the
reference reads ``min`` / ``max`` back with two ``.item()`` calls on every reset of a DelayedPDActuator).  The
``ValueError`` the reference raises is raised by the next host-visible look at the lags: ``min_time_lag`` /
``max_time_lag`` / ``validate()``.  Integer arguments are checked on the host at once (no device work needed).
"""
from __future__ import annotations

from collections.abc import Sequence

import torch

from .circular_buffer import CircularBuffer


class DelayBuffer:
    def __init__(self, history_length: int, batch_size: int, device: str):
        self._history_length = max(0, int(history_length))
        self._ring = CircularBuffer(self._history_length + 1, batch_size, device)
        self._lags = torch.zeros(batch_size, dtype=torch.int32, device=device)
        # int32 {smallest requested lag, largest requested lag} of the tensor calls so far (device-side range check)
        self._requested = torch.zeros(2, dtype=torch.int32, device=device)

    # -- read-only views -----------------------------------------------------------------------------
    batch_size = property(lambda self: self._ring.batch_size)
    device = property(lambda self: self._ring.device)
    history_length = property(lambda self: self._history_length)
    time_lags = property(lambda self: self._lags)

    @property
    def min_time_lag(self) -> int:
        """Smallest lag in use (reads the device: one synchronisation, not on the step path)."""
        self.validate()
        return int(self._lags.min())

    @property
    def max_time_lag(self) -> int:
        self.validate()
        return int(self._lags.max())

    def validate(self) -> None:
        """Raises the reference's errors (delay_buffer.py:141-149) if a tensor handed to ``set_time_lag`` held a lag
        outside ``[0, history_length]``.  Host round trip; call it where a synchronisation is acceptable."""
        lo, hi = (int(v) for v in self._requested.tolist())
        if lo < 0:
            raise ValueError(f"The minimum time lag cannot be negative. Received: {lo}")
        if hi > self._history_length:
            raise ValueError(f"The maximum time lag cannot be larger than the history length. Received: {hi}")

    # -- configuration -------------------------------------------------------------------------------
    def set_time_lag(self, time_lag: int | torch.Tensor, batch_ids: Sequence[int] | None = None):
        rows = slice(None) if batch_ids is None else batch_ids
        if isinstance(time_lag, bool) or not isinstance(time_lag, (int, torch.Tensor)):
            raise TypeError(f"Invalid type for time_lag: {type(time_lag)}. Expected int or integer tensor.")
        if isinstance(time_lag, int):
            if time_lag < 0:
                raise ValueError(f"The minimum time lag cannot be negative. Received: {time_lag}")
            if time_lag > self._history_length:
                raise ValueError(f"The maximum time lag cannot be larger than the history length. Received: {time_lag}")
            self._lags[rows] = time_lag
            return
        if time_lag.dtype not in (torch.int32, torch.int64):
            raise TypeError(f"Invalid dtype for time_lag: {time_lag.dtype}. Expected torch.int or torch.long.")
        lag = time_lag.to(device=self.device, dtype=torch.int32)
        if lag.numel() > 0:
            # range check without leaving the device: widen the recorded {min, max} of everything requested so far
            torch.minimum(self._requested[0], lag.min(), out=self._requested[0])
            torch.maximum(self._requested[1], lag.max(), out=self._requested[1])
        self._lags[rows] = lag.clamp(0, self._history_length)

    def reset(self, batch_ids: Sequence[int] | None = None):
        self._ring.reset(batch_ids)

    # -- the step --------------------------------------------------------------------------------------
    def compute(self, data: torch.Tensor) -> torch.Tensor:
        """Push ``data`` (batch_size, ...) and return the delayed samples (same shape)."""
        return self._ring.append_and_read(data, self._lags)
