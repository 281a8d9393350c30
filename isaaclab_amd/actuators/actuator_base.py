"""``ActuatorBase`` (``LAB/actuators/actuator_base.py:20-261``): parameter parsing into (N, Jg) buffers
and the ``compute(control_action, joint_pos, joint_vel)`` contract."""
from __future__ import annotations

from abc import ABC, abstractmethod
from collections.abc import Sequence

import ctypes as C

import torch

from .. import _lib as L
from ..utils import string as string_utils
from ..utils.types import ArticulationActions


class ActuatorBase(ABC):
    effort_limit: torch.Tensor
    velocity_limit: torch.Tensor
    stiffness: torch.Tensor
    damping: torch.Tensor
    armature: torch.Tensor
    friction: torch.Tensor

    def __init__(self, cfg, joint_names: list[str], joint_ids: slice | Sequence[int], num_envs: int, device: str,
                 stiffness: torch.Tensor | float = 0.0, damping: torch.Tensor | float = 0.0,
                 armature: torch.Tensor | float = 0.0, friction: torch.Tensor | float = 0.0,
                 effort_limit: torch.Tensor | float = torch.inf, velocity_limit: torch.Tensor | float = torch.inf):
        self.cfg = cfg
        self._num_envs = num_envs
        self._device = device
        self._joint_names = joint_names
        self._joint_indices = joint_ids
        self.stiffness = self._parse_joint_parameter(self.cfg.stiffness, stiffness)
        self.damping = self._parse_joint_parameter(self.cfg.damping, damping)
        self.armature = self._parse_joint_parameter(self.cfg.armature, armature)
        self.friction = self._parse_joint_parameter(self.cfg.friction, friction)
        self.effort_limit = self._parse_joint_parameter(self.cfg.effort_limit, effort_limit)
        self.velocity_limit = self._parse_joint_parameter(self.cfg.velocity_limit, velocity_limit)
        self.computed_effort = torch.zeros(self._num_envs, self.num_joints, device=self._device)
        self.applied_effort = torch.zeros_like(self.computed_effort)

    # A group that ``Articulation`` evaluates as part of a merged launch (``Articulation._merge_actuator_groups``) has
    # no buffers of its own any more: its efforts are the group's columns of the merged (N, J) outputs.
    _merged_into = None

    @property
    def computed_effort(self) -> torch.Tensor:
        if self._merged_into is not None:
            return self._merged_into.computed_effort[:, self._joint_indices]
        return self._computed_effort

    @computed_effort.setter
    def computed_effort(self, value: torch.Tensor):
        self._computed_effort = value

    @property
    def applied_effort(self) -> torch.Tensor:
        if self._merged_into is not None:
            return self._merged_into.applied_effort[:, self._joint_indices]
        return self._applied_effort

    @applied_effort.setter
    def applied_effort(self, value: torch.Tensor):
        self._applied_effort = value

    def __str__(self) -> str:
        joint_indices = self.joint_indices
        if joint_indices == slice(None):
            joint_indices = list(range(self.num_joints))
        return (
            f"<class {self.__class__.__name__}> object:\n"
            f"\tNumber of joints      : {self.num_joints}\n"
            f"\tJoint names expression: {self.cfg.joint_names_expr}\n"
            f"\tJoint names           : {self.joint_names}\n"
            f"\tJoint indices         : {joint_indices}\n"
        )

    @property
    def num_joints(self) -> int:
        return len(self._joint_names)

    @property
    def joint_names(self) -> list[str]:
        return self._joint_names

    @property
    def joint_indices(self) -> slice | Sequence[int]:
        return self._joint_indices

    @abstractmethod
    def reset(self, env_ids: Sequence[int]):
        raise NotImplementedError

    @abstractmethod
    def compute(self, control_action: ArticulationActions, joint_pos: torch.Tensor,
                joint_vel: torch.Tensor) -> ArticulationActions:
        raise NotImplementedError

    def _parse_joint_parameter(self, cfg_value, default_value) -> torch.Tensor:
        param = torch.zeros(self._num_envs, self.num_joints, device=self._device)
        if cfg_value is not None:
            if isinstance(cfg_value, (float, int)):
                param[:] = float(cfg_value)
            elif isinstance(cfg_value, dict):
                indices, _, values = string_utils.resolve_matching_names_values(cfg_value, self.joint_names)
                param[:, indices] = torch.tensor(values, dtype=torch.float, device=self._device)
            else:
                raise TypeError(
                    f"Invalid type for parameter value: {type(cfg_value)} for "
                    + f"actuator on joints {self.joint_names}. Expected float or dict."
                )
        elif default_value is not None:
            if isinstance(default_value, (float, int)):
                param[:] = float(default_value)
            elif isinstance(default_value, torch.Tensor):
                if default_value.shape == (self._num_envs, self.num_joints):
                    param = default_value.float().contiguous()
                else:
                    raise ValueError(
                        "Invalid default value tensor shape.\n"
                        f"Got: {default_value.shape}\n"
                        f"Expected: {(self._num_envs, self.num_joints)}"
                    )
            else:
                raise TypeError(
                    f"Invalid type for default value: {type(default_value)} for "
                    + f"actuator on joints {self.joint_names}. Expected float or Tensor."
                )
        else:
            raise ValueError("The parameter value is None and no default value is provided.")
        return param

    # -- shared launcher ---------------------------------------------------------------------------
    #: whether ``compute`` issues exactly one launch that can carry a pending scene update along
    #: (``mdp_actuator_*_compute_after_update``); set by ``Articulation`` right before ``compute``
    carries_scene_update: bool = False
    pending_scene_update = None
    #: whether ``compute`` can process the step's joint-position action in its own launch
    #: (``mdp_actuator_*_compute_with_action``); ``Articulation`` sets ``pending_joint_action`` right before ``compute``
    folds_joint_action: bool = False
    pending_joint_action = None
    #: device float the next ``compute`` is conditional on (``mdp_actuator_*_compute_gated``: the launch does nothing
    #: unless it is > 0); ``None`` for actuators that cannot take one.  Set by ``Articulation`` right before ``compute``
    supports_gate: bool = False
    pending_gate = None
    #: (cast arguments, reset mask) of a flagged height-scan recast the next GATED ``compute`` carries in its launch
    #: (``mdp_actuator_lstm_compute_gated_with_recast``); ``supports_recast()`` says whether this instance can
    pending_recast = None

    def supports_recast(self) -> bool:
        return False

    def _take_joint_action(self, q_des_ptr, ld_q_des):
        """The pending joint action if this launch can process it (its processed buffer IS the position target the
        launch reads); otherwise processes it with its own launch and returns None."""
        act, self.pending_joint_action = self.pending_joint_action, None
        if act is None:
            return None
        if self.pending_scene_update is None and q_des_ptr == act.processed and ld_q_des == self.num_joints:
            return act
        L.check(L.lib.mdp_joint_action_process(
            self._num_envs, self.num_joints, act.actions_in, act.ld_in, act.action, act.prev_action, act.processed,
            act.scale_vec, act.scale, act.offset, act.clip, L.stream()))
        return None

    def _pd_launch(self, kind: str, control_action: ArticulationActions, joint_pos: torch.Tensor,
                   joint_vel: torch.Tensor, saturation: float = torch.inf):
        """One fused kernel: PD law + saturation model, writes ``computed_effort`` / ``applied_effort``
        in place (the tensors keep their identity, so fused manager programs can stay bound to them)."""
        q, ld_q = L.rows(joint_pos, "joint_pos")
        qd, ld_qd = L.rows(joint_vel, "joint_vel")
        if q.shape != (self._num_envs, self.num_joints):
            raise ValueError(f"joint_pos shape {tuple(q.shape)} != {(self._num_envs, self.num_joints)}")
        args = []
        for name in ("joint_positions", "joint_velocities", "joint_efforts"):
            t = getattr(control_action, name)
            if t is None:
                args += [None, 0]
            else:
                t2, ld = L.rows(t, name)
                if t2.shape != q.shape:
                    raise ValueError(f"{name} shape {tuple(t2.shape)} != {tuple(q.shape)}")
                args += [t2.data_ptr(), ld]
        for name in ("stiffness", "damping", "effort_limit", "velocity_limit"):
            p = getattr(self, name)
            if not p.is_contiguous():
                setattr(self, name, p.contiguous())
        # a pending scene update (scene.update of the frame being read) rides along in the same launch
        gate, self.pending_gate = self.pending_gate, None
        if gate is not None:
            L.check(L.lib.mdp_actuator_pd_compute_gated(
                gate.data_ptr(), L.ACT_KINDS[kind], self._num_envs, self.num_joints, *args, q.data_ptr(), ld_q,
                qd.data_ptr(), ld_qd, self.stiffness.data_ptr(), self.damping.data_ptr(), self.effort_limit.data_ptr(),
                self.velocity_limit.data_ptr(), float(saturation), self.computed_effort.data_ptr(),
                self.applied_effort.data_ptr(), L.stream(q.device)))
            return
        act = self._take_joint_action(args[0], args[1])
        if act is not None:
            L.check(L.lib.mdp_actuator_pd_compute_with_action(
                C.byref(act), L.ACT_KINDS[kind], self._num_envs, self.num_joints, *args[2:], q.data_ptr(), ld_q,
                qd.data_ptr(), ld_qd, self.stiffness.data_ptr(), self.damping.data_ptr(), self.effort_limit.data_ptr(),
                self.velocity_limit.data_ptr(), float(saturation), self.computed_effort.data_ptr(),
                self.applied_effort.data_ptr(), L.stream(q.device)))
            return
        upd, self.pending_scene_update = self.pending_scene_update, None
        L.check(L.lib.mdp_actuator_pd_compute_after_update(
            C.byref(upd[0]) if upd is not None else None, float(upd[1]) if upd is not None else 0.0,
            L.ACT_KINDS[kind], self._num_envs, self.num_joints, *args, q.data_ptr(), ld_q, qd.data_ptr(), ld_qd,
            self.stiffness.data_ptr(), self.damping.data_ptr(), self.effort_limit.data_ptr(),
            self.velocity_limit.data_ptr(), float(saturation), self.computed_effort.data_ptr(),
            self.applied_effort.data_ptr(), L.stream(q.device)))
