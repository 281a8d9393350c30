"""Observation corruption (``LAB/utils/noise/noise_model.py:20-182``).

The three noise functions launch the post-processing kernel with a counter-based Philox stream
(RNG parity with ``torch.rand_like`` is neither possible nor required: SURVEY.md section 7).
"""
from __future__ import annotations

from collections.abc import Sequence
from typing import TYPE_CHECKING

import torch

from ... import _lib as L
from .._post import fill_noise, postprocess_inplace

if TYPE_CHECKING:
    from . import noise_cfg


def _apply(data: torch.Tensor, cfg) -> torch.Tensor:
    out = L.f32(data, "data").clone().contiguous()
    term = L.MdpTerm()
    fill_noise(term, cfg)
    return postprocess_inplace(out, term)


def constant_noise(data: torch.Tensor, cfg: "noise_cfg.ConstantNoiseCfg") -> torch.Tensor:
    return _apply(data, cfg)


def uniform_noise(data: torch.Tensor, cfg: "noise_cfg.UniformNoiseCfg") -> torch.Tensor:
    return _apply(data, cfg)


def gaussian_noise(data: torch.Tensor, cfg: "noise_cfg.GaussianNoiseCfg") -> torch.Tensor:
    return _apply(data, cfg)


class NoiseModel:
    def __init__(self, noise_model_cfg, num_envs: int, device: str):
        self._noise_model_cfg = noise_model_cfg
        self._num_envs = num_envs
        self._device = device

    def reset(self, env_ids: Sequence[int] | None = None):
        pass

    def reset_masked(self, mask: torch.Tensor | None) -> None:
        """``reset`` for the envs of a device bool mask (None = all): no id list, no host synchronisation."""
        pass

    def apply(self, data: torch.Tensor) -> torch.Tensor:
        return self._noise_model_cfg.noise_cfg.func(data, self._noise_model_cfg.noise_cfg)


class NoiseModelWithAdditiveBias(NoiseModel):
    def __init__(self, noise_model_cfg, num_envs: int, device: str):
        super().__init__(noise_model_cfg, num_envs, device)
        self._bias_noise_cfg = noise_model_cfg.bias_noise_cfg
        self._bias = torch.zeros((num_envs, 1), device=self._device)

    def reset(self, env_ids: Sequence[int] | None = None):
        if env_ids is None:
            env_ids = slice(None)
        self._bias[env_ids] = self._bias_noise_cfg.func(self._bias[env_ids], self._bias_noise_cfg)

    def reset_masked(self, mask: torch.Tensor | None) -> None:
        # the bias of every env is drawn and kept only where the mask is set: same distribution as the id-list form,
        # the selection stays on the device
        fresh = self._bias_noise_cfg.func(self._bias, self._bias_noise_cfg)
        if mask is None:
            self._bias.copy_(fresh)
        else:
            torch.where(mask.view(-1, 1), fresh, self._bias, out=self._bias)

    def apply(self, data: torch.Tensor) -> torch.Tensor:
        from ..._terms import make_term, run_obs_single

        noisy = super().apply(data)
        flat = noisy.reshape(self._num_envs, -1)
        term = make_term("MDP_OBS_ADD_ROW_BIAS", dim=flat.shape[1], p0=flat, i0=flat.shape[1], p1=self._bias, i2=1)
        return run_obs_single(term, self._num_envs, flat.device).reshape(noisy.shape)
