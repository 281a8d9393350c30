"""Actuator configuration classes (``LAB/actuators/actuator_cfg.py:16-188``)."""
from __future__ import annotations

from collections.abc import Iterable
from dataclasses import MISSING
from typing import Literal

from ..utils.configclass import configclass
from . import actuator_net, actuator_pd


@configclass
class ActuatorBaseCfg:
    class_type: type = MISSING
    joint_names_expr: list[str] = MISSING
    effort_limit: dict[str, float] | float | None = None
    velocity_limit: dict[str, float] | float | None = None
    stiffness: dict[str, float] | float | None = MISSING
    damping: dict[str, float] | float | None = MISSING
    armature: dict[str, float] | float | None = None
    friction: dict[str, float] | float | None = None


@configclass
class ImplicitActuatorCfg(ActuatorBaseCfg):
    class_type: type = actuator_pd.ImplicitActuator


@configclass
class IdealPDActuatorCfg(ActuatorBaseCfg):
    class_type: type = actuator_pd.IdealPDActuator


@configclass
class DCMotorCfg(IdealPDActuatorCfg):
    class_type: type = actuator_pd.DCMotor
    saturation_effort: float = MISSING


@configclass
class ActuatorNetLSTMCfg(DCMotorCfg):
    class_type: type = actuator_net.ActuatorNetLSTM
    stiffness = None
    damping = None
    network_file: str = MISSING


@configclass
class ActuatorNetMLPCfg(DCMotorCfg):
    class_type: type = actuator_net.ActuatorNetMLP
    stiffness = None
    damping = None
    network_file: str = MISSING
    pos_scale: float = MISSING
    vel_scale: float = MISSING
    torque_scale: float = MISSING
    input_order: Literal["pos_vel", "vel_pos"] = MISSING
    input_idx: Iterable[int] = MISSING


@configclass
class DelayedPDActuatorCfg(IdealPDActuatorCfg):
    class_type: type = actuator_pd.DelayedPDActuator
    min_delay: int = 0
    max_delay: int = 0


@configclass
class RemotizedPDActuatorCfg(DelayedPDActuatorCfg):
    class_type: type = actuator_pd.RemotizedPDActuator
    joint_parameter_lookup: list[list[float]] = MISSING
