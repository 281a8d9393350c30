"""Term configuration dataclasses (``LAB/managers/manager_term_cfg.py:28-344``)."""
from __future__ import annotations

from collections.abc import Callable
from dataclasses import MISSING
from typing import Any

import torch

from ..utils.configclass import configclass
from ..utils.modifiers import ModifierCfg
from ..utils.noise import NoiseCfg
from .scene_entity_cfg import SceneEntityCfg


@configclass
class ManagerTermBaseCfg:
    func: Callable | Any = MISSING
    params: dict[str, Any | SceneEntityCfg] = dict()


@configclass
class ActionTermCfg:
    class_type: type = MISSING
    asset_name: str = MISSING
    debug_vis: bool = False
    clip: dict[str, tuple] | None = None


@configclass
class CommandTermCfg:
    class_type: type = MISSING
    resampling_time_range: tuple[float, float] = MISSING
    debug_vis: bool = False


@configclass
class CurriculumTermCfg(ManagerTermBaseCfg):
    func: Callable = MISSING


@configclass
class EventTermCfg(ManagerTermBaseCfg):
    func: Callable = MISSING
    mode: str = MISSING
    interval_range_s: tuple[float, float] | None = None
    is_global_time: bool = False
    min_step_count_between_reset: int = 0


@configclass
class ObservationTermCfg(ManagerTermBaseCfg):
    func: Callable = MISSING
    modifiers: list[ModifierCfg] | None = None
    noise: NoiseCfg | None = None
    clip: tuple[float, float] | None = None
    scale: tuple[float, ...] | float | None = None
    history_length: int = 0
    flatten_history_dim: bool = True


@configclass
class ObservationGroupCfg:
    concatenate_terms: bool = True
    enable_corruption: bool = False
    history_length: int | None = None
    flatten_history_dim: bool = True


@configclass
class RewardTermCfg(ManagerTermBaseCfg):
    func: Callable = MISSING
    weight: float = MISSING


@configclass
class TerminationTermCfg(ManagerTermBaseCfg):
    func: Callable = MISSING
    time_out: bool = False
