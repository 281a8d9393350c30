"""Command cfgs (``LAB/envs/mdp/commands/commands_cfg.py:33-112``)."""
from __future__ import annotations

import math
from dataclasses import MISSING

from ....managers.manager_term_cfg import CommandTermCfg
from ....utils.configclass import configclass
from .velocity_command import UniformVelocityCommand


@configclass
class UniformVelocityCommandCfg(CommandTermCfg):
    class_type: type = UniformVelocityCommand
    asset_name: str = MISSING
    heading_command: bool = False
    heading_control_stiffness: float = 1.0
    rel_standing_envs: float = 0.0
    rel_heading_envs: float = 1.0

    @configclass
    class Ranges:
        lin_vel_x: tuple = MISSING
        lin_vel_y: tuple = MISSING
        ang_vel_z: tuple = MISSING
        heading: tuple | None = None

    ranges: Ranges = MISSING
    seed: int | None = None
