"""``Articulation`` (``LAB/assets/articulation/articulation.py``) reduced to what the hot path needs:
name lookup, joint targets, actuator groups and ``_apply_actuator_model`` (:1399-1436)."""
from __future__ import annotations

from collections.abc import Sequence
from dataclasses import MISSING

import torch

from ..actuators import ActuatorBase, DCMotor, IdealPDActuator, ImplicitActuator
from ..utils import string as string_utils
from ..utils.configclass import configclass
from ..utils.types import ArticulationActions
from .articulation_data import ArticulationData


@configclass
class ArticulationCfg:
    """Subset of ``LAB/assets/articulation/articulation_cfg.py`` consumed by the MDP path."""

    joint_names: list = MISSING
    body_names: list = MISSING
    default_joint_pos: dict = {".*": 0.0}
    default_joint_vel: dict = {".*": 0.0}
    soft_joint_pos_limit_factor: float = 1.0
    joint_pos_limits: tuple | dict = (-3.14, 3.14)   # synthetic stand-in for the USD joint limits
    joint_vel_limits: float = 100.0
    joint_effort_limits: float = 1000.0
    actuators: dict = MISSING


class Articulation:
    #: several stateless PD groups of one class that together cover every joint are evaluated by ONE launch over the
    #: full (N, J) buffers (see ``_merge_actuator_groups``); ``False`` restores one launch + gather / scatter per group
    merge_actuator_groups: bool = True

    def __init__(self, cfg: ArticulationCfg, root_physx_view, device: str, merge_actuator_groups: bool | None = None):
        if merge_actuator_groups is not None:
            self.merge_actuator_groups = merge_actuator_groups
        self.cfg = cfg
        self.device = device
        self.root_physx_view = root_physx_view
        self._joint_names = list(cfg.joint_names)
        self._body_names = list(cfg.body_names)
        self.num_instances = root_physx_view.count
        self._data = ArticulationData(root_physx_view, device, len(self._body_names), self._body_names, self._joint_names)
        self._ALL_INDICES = torch.arange(self.num_instances, dtype=torch.long, device=device)
        n, j = self.num_instances, self.num_joints
        self._joint_pos_target_sim = torch.zeros((n, j), device=device)
        self._joint_vel_target_sim = torch.zeros((n, j), device=device)
        self._joint_effort_target_sim = torch.zeros((n, j), device=device)
        self._process_cfg()
        self._process_actuators_cfg()

    # -- properties -----------------------------------------------------------------------------------
    @property
    def data(self) -> ArticulationData: return self._data
    @property
    def num_joints(self) -> int: return len(self._joint_names)
    @property
    def num_bodies(self) -> int: return len(self._body_names)
    @property
    def num_fixed_tendons(self) -> int: return 0
    @property
    def joint_names(self) -> list[str]: return self._joint_names
    @property
    def body_names(self) -> list[str]: return self._body_names
    @property
    def fixed_tendon_names(self) -> list[str]: return []

    def find_bodies(self, name_keys, preserve_order: bool = False):
        return string_utils.resolve_matching_names(name_keys, self.body_names, preserve_order)

    def find_joints(self, name_keys, joint_subset: list[str] | None = None, preserve_order: bool = False):
        if joint_subset is None:
            joint_subset = self.joint_names
        return string_utils.resolve_matching_names(name_keys, joint_subset, preserve_order)

    def find_fixed_tendons(self, name_keys, tendon_subsets=None, preserve_order: bool = False):
        return string_utils.resolve_matching_names(name_keys, [], preserve_order)

    # -- operations -----------------------------------------------------------------------------------
    def reset(self, env_ids: Sequence[int] | None = None):
        if env_ids is None:
            env_ids = slice(None)
        for actuator in self.actuators.values():
            actuator.reset(env_ids)

    def write_data_to_sim(self, carry_update=None, joint_action=None, gate=None, recast=None):
        """articulation.py:172-199 (the physics-side setters are out of scope).  ``carry_update = (desc, dt)`` lets the
        actuator launch run the scene update of the frame it reads in the same kernel (``can_carry_update``);
        ``joint_action`` (``L.MdpJointAction``) lets it process the action of the step (``can_fold_joint_action``).
        ``gate`` (device float): the whole call is conditional on ``gate > 0``, decided on the device -- the
        ``write_data_to_sim`` of manager_based_rl_env.py:221, which only runs in steps that reset an env."""
        self._apply_actuator_model(carry_update, joint_action, gate, recast)

    @property
    def can_fold_joint_action(self) -> bool:
        if not getattr(self, "_single_full_group", False):
            return False
        return bool(getattr(self._step_actuators()[0], "folds_joint_action", False))

    @property
    def can_carry_update(self) -> bool:
        if not getattr(self, "_single_full_group", False):
            return False
        return bool(self._step_actuators()[0].carries_scene_update)

    def update(self, dt: float):
        self._data.update(dt)

    def _set_target(self, buf: torch.Tensor, target: torch.Tensor, joint_ids, env_ids):
        """articulation.py:807-880"""
        if env_ids is None and (joint_ids is None or joint_ids == slice(None)):
            if target.data_ptr() != buf.data_ptr():
                buf.copy_(target)
            return
        if env_ids is None:
            env_ids = slice(None)
        if joint_ids is None:
            joint_ids = slice(None)
        if env_ids != slice(None) and joint_ids != slice(None):
            env_ids = env_ids[:, None]
        buf[env_ids, joint_ids] = target

    def set_joint_position_target(self, target, joint_ids=None, env_ids=None):
        self._set_target(self._data.joint_pos_target, target, joint_ids, env_ids)

    def set_joint_velocity_target(self, target, joint_ids=None, env_ids=None):
        self._set_target(self._data.joint_vel_target, target, joint_ids, env_ids)

    def set_joint_effort_target(self, target, joint_ids=None, env_ids=None):
        self._set_target(self._data.joint_effort_target, target, joint_ids, env_ids)

    # -- internals ------------------------------------------------------------------------------------
    def _process_cfg(self):
        n, j = self.num_instances, self.num_joints
        d = self._data
        idx, _, vals = string_utils.resolve_matching_names_values(self.cfg.default_joint_pos, self.joint_names)
        d.default_joint_pos[:, idx] = torch.tensor(vals, dtype=torch.float32, device=self.device)
        idx, _, vals = string_utils.resolve_matching_names_values(self.cfg.default_joint_vel, self.joint_names)
        d.default_joint_vel[:, idx] = torch.tensor(vals, dtype=torch.float32, device=self.device)
        lim = self.cfg.joint_pos_limits
        if isinstance(lim, dict):
            idx, _, vals = string_utils.resolve_matching_names_values(lim, self.joint_names)
            d.joint_limits[:, idx] = torch.tensor(vals, dtype=torch.float32, device=self.device)
        else:
            d.joint_limits[..., 0] = float(lim[0])
            d.joint_limits[..., 1] = float(lim[1])
        d.default_joint_limits.copy_(d.joint_limits)
        d.joint_velocity_limits[:] = float(self.cfg.joint_vel_limits)
        d.joint_effort_limits[:] = float(self.cfg.joint_effort_limits)
        # soft limits = mean +- 0.5 * range * factor (articulation.py:1231-1237); init-time host math
        mean = (d.joint_limits[..., 0] + d.joint_limits[..., 1]) / 2
        rng = d.joint_limits[..., 1] - d.joint_limits[..., 0]
        f = self.cfg.soft_joint_pos_limit_factor
        d.soft_joint_pos_limits[..., 0] = mean - 0.5 * rng * f
        d.soft_joint_pos_limits[..., 1] = mean + 0.5 * rng * f
        d.joint_pos_target.copy_(d.default_joint_pos)

    def _process_actuators_cfg(self):
        """articulation.py:1289-1368"""
        self.actuators: dict[str, ActuatorBase] = dict()
        self._has_implicit_actuators = False
        d = self._data
        for actuator_name, actuator_cfg in self.cfg.actuators.items():
            joint_ids, joint_names = self.find_joints(actuator_cfg.joint_names_expr)
            if len(joint_names) == 0:
                raise ValueError(
                    f"No joints found for actuator group: {actuator_name} with joint name expression:"
                    f" {actuator_cfg.joint_names_expr}."
                )
            actuator = actuator_cfg.class_type(
                cfg=actuator_cfg, joint_names=joint_names,
                joint_ids=(slice(None) if len(joint_names) == self.num_joints
                           else torch.tensor(joint_ids, device=self.device)),
                num_envs=self.num_instances, device=self.device,
                stiffness=d.default_joint_stiffness[:, joint_ids], damping=d.default_joint_damping[:, joint_ids],
                armature=d.default_joint_armature[:, joint_ids], friction=d.default_joint_friction[:, joint_ids],
                effort_limit=d.joint_effort_limits.clone()[:, joint_ids],
                velocity_limit=d.joint_velocity_limits.clone()[:, joint_ids],
            )
            self.actuators[actuator_name] = actuator
            if isinstance(actuator, ImplicitActuator):
                self._has_implicit_actuators = True
            d.default_joint_stiffness[:, actuator.joint_indices] = actuator.stiffness
            d.default_joint_damping[:, actuator.joint_indices] = actuator.damping
        # single group over all joints: bind the data buffers to the actuator's outputs (no scatter copies)
        if len(self.actuators) == 1:
            act = next(iter(self.actuators.values()))
            if act.joint_indices == slice(None):
                d.computed_torque = act.computed_effort
                d.applied_torque = act.applied_effort
                d.soft_joint_vel_limits = act.velocity_limit
                self._single_full_group = True
                return
        self._single_full_group = False
        self._merged_actuator = None
        if self.merge_actuator_groups:
            self._merge_actuator_groups()

    def _merge_actuator_groups(self):
        """Groups of ONE stateless PD class (ImplicitActuator / IdealPDActuator / DCMotor with one saturation effort) that
        partition the joints -- H1's legs / feet / arms (ASSETS/robots/unitree.py H1_MINIMAL_CFG) -- do the same arithmetic
        per (env, joint) element with per-element parameters, so ``_apply_actuator_model`` (articulation.py:1399-1436)
        runs them as one launch over the full (N, J) buffers writing ``computed_torque`` / ``applied_torque`` in place,
        instead of a gather of five inputs, a launch and a scatter of six outputs per group.  The groups stay in
        ``self.actuators`` with their own parameters (edits of a group's gains or limits are picked up by the next step,
        see ``_step_actuators``); a group's ``computed_effort`` / ``applied_effort`` are READ-ONLY gathers of its columns
        of the merged outputs -- a fresh tensor per access, so writes to them and pointers taken from them do not reach
        the articulation: use ``data.computed_torque`` / ``data.applied_torque`` for that."""
        acts = list(self.actuators.values())
        if len(acts) < 2:
            return
        cls = type(acts[0])
        if cls not in (ImplicitActuator, IdealPDActuator, DCMotor) or any(type(a) is not cls for a in acts):
            return
        if cls is DCMotor and len({float(a._saturation_effort) for a in acts}) != 1:
            return
        ids = []
        for a in acts:
            ji = a.joint_indices
            ids += list(range(self.num_joints)) if ji == slice(None) else [int(i) for i in ji.tolist()]
        if sorted(ids) != list(range(self.num_joints)):
            return
        n, j = self.num_instances, self.num_joints
        merged = object.__new__(cls)
        merged.cfg = acts[0].cfg
        merged._num_envs, merged._device = n, self.device
        merged._joint_names, merged._joint_indices = list(self.joint_names), slice(None)
        for name in ("stiffness", "damping", "armature", "friction", "effort_limit", "velocity_limit"):
            setattr(merged, name, torch.zeros((n, j), device=self.device))
        merged.computed_effort = torch.zeros((n, j), device=self.device)
        merged.applied_effort = torch.zeros((n, j), device=self.device)
        if cls is DCMotor:
            merged._saturation_effort = acts[0]._saturation_effort
        self._merged_actuator = merged
        self.refresh_merged_actuator_parameters()
        d = self._data
        d.computed_torque, d.applied_torque = merged.computed_effort, merged.applied_effort
        d.soft_joint_vel_limits = merged.velocity_limit
        for a in acts:
            a._merged_into = merged
        self._single_full_group = True

    _MERGED_PARAMS = ("stiffness", "damping", "armature", "friction", "effort_limit", "velocity_limit")

    def _group_param_versions(self) -> tuple:
        """Identity and in-place version counter of every group parameter tensor: changes when a caller edits a group's
        gains or limits in place (``actuator.stiffness[env_ids] = ...``, the reference's randomize_actuator_gains
        pattern) or rebinds the attribute.  Host-only bookkeeping, no launch."""
        return tuple((id(t), t._version) for a in self.actuators.values()
                     for t in (getattr(a, name) for name in self._MERGED_PARAMS))

    def refresh_merged_actuator_parameters(self):
        """Copies the groups' parameters into the merged (N, J) buffers.  ``_step_actuators`` calls it by itself when a
        group's parameter tensor was modified since the last copy."""
        m = self._merged_actuator
        if m is None:
            return
        for a in self.actuators.values():
            for name in self._MERGED_PARAMS:
                getattr(m, name)[:, a.joint_indices] = getattr(a, name)
        self._merged_versions = self._group_param_versions()

    def _step_actuators(self) -> list:
        """The actuator objects ``_apply_actuator_model`` evaluates: the merged one, else every group."""
        if getattr(self, "_merged_actuator", None) is not None:
            if self._group_param_versions() != getattr(self, "_merged_versions", None):
                self.refresh_merged_actuator_parameters()      # a group's gains / limits were edited: pick them up
            return [self._merged_actuator]
        return list(self.actuators.values())

    def can_carry_recast(self) -> bool:
        """True when the gated actuator step is ONE launch that can carry a flagged height-scan recast."""
        acts = self._step_actuators()
        return len(acts) == 1 and acts[0].supports_gate and acts[0].supports_recast()

    def _apply_actuator_model(self, carry_update=None, joint_action=None, gate=None, recast=None):
        """articulation.py:1399-1436"""
        d = self._data
        if carry_update is not None:
            # the launch that refreshes d.joint_pos / d.joint_vel is this very launch: read the physics buffers
            joint_pos_all = self.root_physx_view.get_dof_positions()
            joint_vel_all = self.root_physx_view.get_dof_velocities()
        else:
            joint_pos_all, joint_vel_all = d.joint_pos, d.joint_vel
        for actuator in self._step_actuators():
            ji = actuator.joint_indices
            actuator.pending_scene_update = carry_update
            actuator.pending_joint_action = joint_action
            if gate is not None:
                if not actuator.supports_gate or carry_update is not None or joint_action is not None:
                    raise NotImplementedError(f"device-conditional compute of {type(actuator).__name__}")
                actuator.pending_gate = gate
                if recast is not None:
                    if not actuator.supports_recast():
                        raise NotImplementedError(f"{type(actuator).__name__} cannot carry the flagged recast")
                    actuator.pending_recast = recast
            control_action = ArticulationActions(
                joint_positions=d.joint_pos_target[:, ji], joint_velocities=d.joint_vel_target[:, ji],
                joint_efforts=d.joint_effort_target[:, ji], joint_indices=ji,
            )
            control_action = actuator.compute(control_action, joint_pos=joint_pos_all[:, ji], joint_vel=joint_vel_all[:, ji])
            if self._single_full_group:
                # outputs already live in the data buffers; targets handed to physics are aliases
                if control_action.joint_efforts is not None:
                    self._joint_effort_target_sim = control_action.joint_efforts
                if control_action.joint_positions is not None:
                    self._joint_pos_target_sim = control_action.joint_positions
                if control_action.joint_velocities is not None:
                    self._joint_vel_target_sim = control_action.joint_velocities
                continue
            if control_action.joint_positions is not None:
                self._joint_pos_target_sim[:, ji] = control_action.joint_positions
            if control_action.joint_velocities is not None:
                self._joint_vel_target_sim[:, ji] = control_action.joint_velocities
            if control_action.joint_efforts is not None:
                self._joint_effort_target_sim[:, ji] = control_action.joint_efforts
            d.computed_torque[:, ji] = actuator.computed_effort
            d.applied_torque[:, ji] = actuator.applied_effort
            d.soft_joint_vel_limits[:, ji] = actuator.velocity_limit
            if hasattr(actuator, "gear_ratio"):
                d.gear_ratio[:, ji] = actuator.gear_ratio
