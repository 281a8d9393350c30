"""``LAB/utils/modifiers/modifier_cfg.py``."""
from __future__ import annotations

from collections.abc import Callable
from dataclasses import MISSING
from typing import Any

from ..configclass import configclass
from . import modifier


@configclass
class ModifierCfg:
    func: Callable = MISSING
    params: dict[str, Any] = dict()


@configclass
class DigitalFilterCfg(ModifierCfg):
    func: type = modifier.DigitalFilter
    A: list = MISSING
    B: list = MISSING


@configclass
class IntegratorCfg(ModifierCfg):
    func: type = modifier.Integrator
    dt: float = MISSING
