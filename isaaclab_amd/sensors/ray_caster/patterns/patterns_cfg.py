"""Pattern cfgs (``LAB/sensors/ray_caster/patterns/patterns_cfg.py``)."""
from __future__ import annotations

from collections.abc import Callable
from dataclasses import MISSING
from typing import Literal

from ....utils.configclass import configclass
from . import patterns


@configclass
class PatternBaseCfg:
    func: Callable = MISSING


@configclass
class GridPatternCfg(PatternBaseCfg):
    func: Callable = patterns.grid_pattern
    resolution: float = MISSING
    size: tuple = MISSING
    direction: tuple = (0.0, 0.0, -1.0)
    ordering: Literal["xy", "yx"] = "xy"
