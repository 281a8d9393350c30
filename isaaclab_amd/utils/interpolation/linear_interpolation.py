"""1-D LUT with zero-order hold outside the samples (``LAB/utils/interpolation/linear_interpolation.py:22-85``)."""
from __future__ import annotations

import torch

from ... import _lib as L


class LinearInterpolation:
    def __init__(self, x: torch.Tensor, y: torch.Tensor, device: str):
        self._x = x.view(-1).clone().to(device=device, dtype=torch.float32)
        self._y = y.view(-1).clone().to(device=device, dtype=torch.float32)
        if self._x.numel() == 0:
            raise ValueError("Input tensor x is empty!")
        if self._x.numel() != self._y.numel():
            raise ValueError(f"Input tensors x and y have different sizes: {self._x.numel()} != {self._y.numel()}")
        if torch.any(self._x[1:] < self._x[:-1]):
            raise ValueError("Input tensor x is not sorted in ascending order!")

    def compute(self, q: torch.Tensor, clamp_inplace: torch.Tensor | None = None) -> torch.Tensor:
        """f(q); with ``clamp_inplace`` also clamps that tensor to [-f(q), f(q)] (actuator_pd.py:357-362)."""
        qc = L.f32(q, "q").contiguous()
        fq = torch.empty_like(qc)
        if clamp_inplace is not None and not clamp_inplace.is_contiguous():
            raise ValueError("clamp_inplace must be contiguous")
        L.check(L.lib.mdp_linear_interp(L.ptr(self._x), L.ptr(self._y), self._x.numel(), L.ptr(qc), qc.numel(),
                                        L.ptr(fq), L.ptr(clamp_inplace) if clamp_inplace is not None else None,
                                        L.stream()))
        return fq.view(q.shape)
