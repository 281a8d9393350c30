"""PD actuator family (``LAB/actuators/actuator_pd.py:32-363``); each ``compute`` is one HIP launch."""
from __future__ import annotations

from collections.abc import Sequence

import torch

from ..utils import DelayBuffer, LinearInterpolation
from ..utils.types import ArticulationActions
from .actuator_base import ActuatorBase


class ImplicitActuator(ActuatorBase):
    """actuator_pd.py:32-87: returns the action unchanged, stores the approximate torques."""

    carries_scene_update = True
    folds_joint_action = True
    supports_gate = True

    def reset(self, *args, **kwargs):
        pass

    def compute(self, control_action, joint_pos, joint_vel):
        self._pd_launch("MDP_ACT_IMPLICIT", control_action, joint_pos, joint_vel)
        return control_action


class IdealPDActuator(ActuatorBase):
    """actuator_pd.py:95-145"""

    carries_scene_update = True
    folds_joint_action = True
    supports_gate = True

    def reset(self, env_ids: Sequence[int]):
        pass

    def compute(self, control_action, joint_pos, joint_vel):
        self._pd_launch("MDP_ACT_IDEAL_PD", control_action, joint_pos, joint_vel)
        control_action.joint_efforts = self.applied_effort
        control_action.joint_positions = None
        control_action.joint_velocities = None
        return control_action


class DCMotor(IdealPDActuator):
    """actuator_pd.py:148-233: velocity-dependent torque saturation."""

    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        if self.cfg.saturation_effort is not None:
            self._saturation_effort = self.cfg.saturation_effort
        else:
            self._saturation_effort = torch.inf
        if self.cfg.velocity_limit is None:
            raise ValueError("The velocity limit must be provided for the DC motor actuator model.")

    def compute(self, control_action, joint_pos, joint_vel):
        self._pd_launch("MDP_ACT_DC_MOTOR", control_action, joint_pos, joint_vel, self._saturation_effort)
        control_action.joint_efforts = self.applied_effort
        control_action.joint_positions = None
        control_action.joint_velocities = None
        return control_action


class DelayedPDActuator(IdealPDActuator):
    """actuator_pd.py:236-293: per-env integer command lag (three delay buffers)."""

    folds_joint_action = False   # the delay buffers read the processed targets before the PD launch
    supports_gate = False        # ... and advance with every compute

    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        self.positions_delay_buffer = DelayBuffer(cfg.max_delay, self._num_envs, device=self._device)
        self.velocities_delay_buffer = DelayBuffer(cfg.max_delay, self._num_envs, device=self._device)
        self.efforts_delay_buffer = DelayBuffer(cfg.max_delay, self._num_envs, device=self._device)
        self._ALL_INDICES = torch.arange(self._num_envs, dtype=torch.long, device=self._device)

    def reset(self, env_ids: Sequence[int]):
        super().reset(env_ids)
        if env_ids is None or (isinstance(env_ids, slice) and env_ids == slice(None)):
            num_envs = self._num_envs
        else:
            num_envs = len(env_ids)
        time_lags = torch.randint(low=self.cfg.min_delay, high=self.cfg.max_delay + 1, size=(num_envs,),
                                  dtype=torch.int, device=self._device)
        for buf in (self.positions_delay_buffer, self.velocities_delay_buffer, self.efforts_delay_buffer):
            buf.set_time_lag(time_lags, env_ids)
        for buf in (self.positions_delay_buffer, self.velocities_delay_buffer, self.efforts_delay_buffer):
            buf.reset(env_ids)

    def compute(self, control_action, joint_pos, joint_vel):
        control_action.joint_positions = self.positions_delay_buffer.compute(control_action.joint_positions)
        control_action.joint_velocities = self.velocities_delay_buffer.compute(control_action.joint_velocities)
        control_action.joint_efforts = self.efforts_delay_buffer.compute(control_action.joint_efforts)
        return super().compute(control_action, joint_pos, joint_vel)


class RemotizedPDActuator(DelayedPDActuator):
    """actuator_pd.py:296-363: angle-dependent torque limit from a lookup table."""

    def __init__(self, cfg, joint_names, joint_ids, num_envs, device, stiffness=0.0, damping=0.0, armature=0.0,
                 friction=0.0, effort_limit=torch.inf, velocity_limit=torch.inf):
        cfg.effort_limit = torch.inf
        cfg.velocity_limit = torch.inf
        super().__init__(cfg, joint_names, joint_ids, num_envs, device, stiffness, damping, armature, friction,
                         torch.inf, torch.inf)
        self._joint_parameter_lookup = torch.tensor(cfg.joint_parameter_lookup, device=device)
        self._torque_limit = LinearInterpolation(self.angle_samples, self.max_torque_samples, device=device)

    @property
    def angle_samples(self) -> torch.Tensor:
        return self._joint_parameter_lookup[:, 0]

    @property
    def transmission_ratio_samples(self) -> torch.Tensor:
        return self._joint_parameter_lookup[:, 1]

    @property
    def max_torque_samples(self) -> torch.Tensor:
        return self._joint_parameter_lookup[:, 2]

    def compute(self, control_action, joint_pos, joint_vel):
        control_action = super().compute(control_action, joint_pos, joint_vel)
        # LUT lookup + clamp of applied_effort in one launch (actuator_pd.py:357-362)
        self._torque_limit.compute(joint_pos, clamp_inplace=self.applied_effort)
        control_action.joint_efforts = self.applied_effort
        return control_action
