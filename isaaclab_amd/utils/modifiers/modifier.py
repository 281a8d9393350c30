"""Observation modifiers (``LAB/utils/modifiers/modifier.py:22-259``), HIP-backed."""
from __future__ import annotations

import math
from collections.abc import Sequence

import torch

from ... import _lib as L
from .._post import NOISE_CONSTANT, postprocess_inplace
from ..buffers.circular_buffer import _mask_from_ids
from .modifier_base import ModifierBase


def _fresh(data: torch.Tensor) -> torch.Tensor:
    return L.f32(data, "data").clone().contiguous()


def scale(data: torch.Tensor, multiplier: float) -> torch.Tensor:
    t = L.MdpTerm()
    t.has_scale, t.scale = 1, float(multiplier)
    return postprocess_inplace(_fresh(data), t)


def clip(data: torch.Tensor, bounds: tuple[float | None, float | None]) -> torch.Tensor:
    t = L.MdpTerm()
    t.has_clip = 1
    t.clip_lo = -math.inf if bounds[0] is None else float(bounds[0])
    t.clip_hi = math.inf if bounds[1] is None else float(bounds[1])
    return postprocess_inplace(_fresh(data), t)


def bias(data: torch.Tensor, value: float) -> torch.Tensor:
    t = L.MdpTerm()
    t.noise_kind, t.noise_op, t.n0 = NOISE_CONSTANT, 0, float(value)
    return postprocess_inplace(_fresh(data), t)


class DigitalFilter(ModifierBase):
    def __init__(self, cfg, data_dim: tuple[int, ...], device: str) -> None:
        if cfg.A is None or cfg.B is None:
            raise ValueError("Digital filter coefficients A and B must not be None. Please provide valid coefficients.")
        super().__init__(cfg, data_dim, device)
        self.A = torch.tensor(self._cfg.A, device=self._device, dtype=torch.float32).unsqueeze(1)
        self.B = torch.tensor(self._cfg.B, device=self._device, dtype=torch.float32).unsqueeze(1)
        self.x_n = torch.zeros(self._data_dim + (self.B.shape[0],), device=self._device)
        self.y_n = torch.zeros(self._data_dim + (self.A.shape[0],), device=self._device)

    def reset(self, env_ids: Sequence[int] | None = None):
        mask = _mask_from_ids(env_ids, self._data_dim[0], self._device)
        for buf in (self.x_n, self.y_n):
            L.check(L.lib.mdp_rows_zero(L.ptr(buf), self._data_dim[0], buf.numel() // max(self._data_dim[0], 1),
                                        L.ptr(mask) if mask is not None else None, L.stream()))

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        d = L.f32(data, "data").contiguous()
        out = torch.empty_like(d)
        L.check(L.lib.mdp_modifier_digital_filter(d.numel(), L.ptr(d), L.ptr(self.x_n), self.B.shape[0],
                                                  L.ptr(self.y_n), self.A.shape[0], L.ptr(self.B), L.ptr(self.A),
                                                  L.ptr(out), L.stream()))
        return out


class Integrator(ModifierBase):
    def __init__(self, cfg, data_dim: tuple[int, ...], device: str):
        super().__init__(cfg, data_dim, device)
        self.integral = torch.zeros(self._data_dim, device=self._device)
        self.y_prev = torch.zeros(self._data_dim, device=self._device)

    def reset(self, env_ids: Sequence[int] | None = None):
        mask = _mask_from_ids(env_ids, self._data_dim[0], self._device)
        for buf in (self.integral, self.y_prev):
            L.check(L.lib.mdp_rows_zero(L.ptr(buf), self._data_dim[0], buf.numel() // max(self._data_dim[0], 1),
                                        L.ptr(mask) if mask is not None else None, L.stream()))

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        d = L.f32(data, "data").contiguous()
        out = torch.empty_like(d)
        L.check(L.lib.mdp_modifier_integrator(d.numel(), L.ptr(d), L.ptr(self.integral), L.ptr(self.y_prev),
                                              float(self._cfg.dt), L.ptr(out), L.stream()))
        return out
