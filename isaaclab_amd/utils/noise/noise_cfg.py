"""Noise configuration objects (``LAB/utils/noise/noise_cfg.py:18-92``)."""
from __future__ import annotations

from collections.abc import Callable
from dataclasses import MISSING
from typing import Literal

import torch

from ..configclass import configclass
from . import noise_model


@configclass
class NoiseCfg:
    func: Callable = MISSING
    operation: str = "add"


@configclass
class ConstantNoiseCfg(NoiseCfg):
    func = noise_model.constant_noise
    bias: torch.Tensor | float = 0.0


@configclass
class UniformNoiseCfg(NoiseCfg):
    func = noise_model.uniform_noise
    n_min: torch.Tensor | float = -1.0
    n_max: torch.Tensor | float = 1.0


@configclass
class GaussianNoiseCfg(NoiseCfg):
    func = noise_model.gaussian_noise
    mean: torch.Tensor | float = 0.0
    std: torch.Tensor | float = 1.0


@configclass
class NoiseModelCfg:
    class_type: type = noise_model.NoiseModel
    noise_cfg: NoiseCfg = MISSING


@configclass
class NoiseModelWithAdditiveBiasCfg(NoiseModelCfg):
    class_type: type = noise_model.NoiseModelWithAdditiveBias
    bias_noise_cfg: NoiseCfg = MISSING
