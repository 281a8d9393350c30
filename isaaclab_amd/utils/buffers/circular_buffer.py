"""Batched ring buffer on the GPU (``LAB/utils/buffers/circular_buffer.py:10-166``).

Storage is (max_len, batch, D) like the reference.  Unlike the reference, ``append`` does not sync
with the host: rows whose push counter is zero are back-filled inside the kernel
(``mdp_circular_buffer_append``) instead of through ``.tolist()`` (circular_buffer.py:130-133).
"""
from __future__ import annotations

from collections.abc import Sequence

import torch

from ... import _lib as L


def _mask_from_ids(ids, n: int, device) -> torch.Tensor | None:
    """env-id selector (None / slice(None) / list / tensor) -> uint8 mask (None == all)."""
    if ids is None or (isinstance(ids, slice) and ids == slice(None)):
        return None
    mask = torch.zeros(n, dtype=torch.bool, device=device)
    mask[ids] = True
    return mask


class CircularBuffer:
    def __init__(self, max_len: int, batch_size: int, device: str):
        if max_len < 1:
            raise ValueError(f"The buffer size should be greater than zero. However, it is set to {max_len}!")
        self._batch_size = batch_size
        self._device = device
        self._max_length = int(max_len)
        self._ALL_INDICES = torch.arange(batch_size, device=device)
        self._max_len = torch.full((batch_size,), max_len, dtype=torch.int, device=device)
        self._num_pushes = torch.zeros(batch_size, dtype=torch.long, device=device)
        self._pointer: int = -1
        self._buffer: torch.Tensor = None  # type: ignore
        self._data_shape = None

    @property
    def batch_size(self) -> int:
        return self._batch_size

    @property
    def device(self) -> str:
        return self._device

    @property
    def max_length(self) -> int:
        return self._max_length

    @property
    def current_length(self) -> torch.Tensor:
        return torch.minimum(self._num_pushes, self._max_len)

    @property
    def buffer(self) -> torch.Tensor:
        """(batch, max_len, ...) oldest -> newest (circular_buffer.py:78-86)."""
        d = self._flat_dim
        out = torch.empty((self._batch_size, self._max_length, d), dtype=torch.float32, device=self._device)
        L.check(L.lib.mdp_circular_buffer_read(L.ptr(self._buffer), self._max_length, self._batch_size, d,
                                               self._pointer, L.ptr(out), L.stream()))
        return out.reshape(self._batch_size, self._max_length, *self._data_shape)

    @property
    def _flat_dim(self) -> int:
        d = 1
        for s in self._data_shape:
            d *= s
        return d

    def reset(self, batch_ids: Sequence[int] | None = None):
        mask = _mask_from_ids(batch_ids, self._batch_size, self._device)
        d = self._flat_dim if self._buffer is not None else 1
        L.check(L.lib.mdp_circular_buffer_reset(L.ptr(self._buffer) if self._buffer is not None else None,
                                                L.ptr(self._num_pushes), self._max_length, self._batch_size, d,
                                                L.ptr(mask) if mask is not None else None, L.stream()))

    def _append(self, data: torch.Tensor, lags: torch.Tensor | None, out: torch.Tensor | None):
        if data.shape[0] != self.batch_size:
            raise ValueError(f"The input data has {data.shape[0]} environments while expecting {self.batch_size}")
        data = L.f32(data, "data")
        if self._buffer is None:
            self._pointer = -1
            self._data_shape = tuple(data.shape[1:])
            self._buffer = torch.empty((self._max_length, *data.shape), dtype=torch.float32, device=self._device)
        self._pointer = (self._pointer + 1) % self._max_length
        d2, ld = L.rows(data.reshape(self._batch_size, -1), "data")
        L.check(L.lib.mdp_circular_buffer_append(L.ptr(self._buffer), L.ptr(self._num_pushes), self._max_length,
                                                 self._batch_size, self._flat_dim, self._pointer, L.ptr(d2), ld,
                                                 L.ptr(lags) if lags is not None else None,
                                                 L.ptr(out) if out is not None else None, L.stream()))

    def append(self, data: torch.Tensor):
        self._append(data, None, None)

    def append_and_read(self, data: torch.Tensor, lags: torch.Tensor) -> torch.Tensor:
        """Fused ``append`` + ``self[lags]`` used by :class:`DelayBuffer` (one launch)."""
        out = torch.empty((self._batch_size, *data.shape[1:]), dtype=torch.float32, device=self._device)
        self._append(data, lags, out)
        return out

    def __getitem__(self, key: torch.Tensor) -> torch.Tensor:
        if len(key) != self.batch_size:
            raise ValueError(f"The argument 'key' has length {key.shape[0]}, while expecting {self.batch_size}")
        if self._buffer is None or bool(torch.any(self._num_pushes == 0)):
            raise RuntimeError("Attempting to retrieve data on an empty circular buffer. Please append data first.")
        # index arithmetic only (LIFO lookup, circular_buffer.py:161-166): device-side gather of stored rows
        valid_keys = torch.minimum(key.to(self._num_pushes.dtype), self._num_pushes - 1)
        index_in_buffer = torch.remainder(self._pointer - valid_keys, self._max_length)
        return self._buffer[index_in_buffer, self._ALL_INDICES]
