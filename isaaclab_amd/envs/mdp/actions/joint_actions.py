"""Joint action terms (``LAB/envs/mdp/actions/joint_actions.py:25-225``): ``processed = raw*scale + offset``."""
from __future__ import annotations

from collections.abc import Sequence

import torch

from .... import _lib as L
from ....managers.action_manager import ActionTerm
from ....utils import string as string_utils


class JointAction(ActionTerm):
    _scale: torch.Tensor | float
    _offset: torch.Tensor | float
    _clip: torch.Tensor

    def __init__(self, cfg, env) -> None:
        super().__init__(cfg, env)
        self._joint_ids, self._joint_names = self._asset.find_joints(
            self.cfg.joint_names, preserve_order=self.cfg.preserve_order
        )
        self._num_joints = len(self._joint_ids)
        if self._num_joints == self._asset.num_joints and not self.cfg.preserve_order:
            self._joint_ids = slice(None)
        self._raw_actions = torch.zeros(self.num_envs, self.action_dim, device=self.device)
        self._processed_actions = torch.zeros_like(self.raw_actions)
        if isinstance(cfg.scale, (float, int)):
            self._scale = float(cfg.scale)
        elif isinstance(cfg.scale, dict):
            self._scale = torch.ones(self.num_envs, self.action_dim, device=self.device)
            index_list, _, value_list = string_utils.resolve_matching_names_values(self.cfg.scale, self._joint_names)
            self._scale[:, index_list] = torch.tensor(value_list, device=self.device)
        else:
            raise ValueError(f"Unsupported scale type: {type(cfg.scale)}. Supported types are float and dict.")
        if isinstance(cfg.offset, (float, int)):
            self._offset = float(cfg.offset)
        elif isinstance(cfg.offset, dict):
            self._offset = torch.zeros_like(self._raw_actions)
            index_list, _, value_list = string_utils.resolve_matching_names_values(self.cfg.offset, self._joint_names)
            self._offset[:, index_list] = torch.tensor(value_list, device=self.device)
        else:
            raise ValueError(f"Unsupported offset type: {type(cfg.offset)}. Supported types are float and dict.")
        self._clip = None
        if self.cfg.clip is not None:
            if isinstance(cfg.clip, dict):
                self._clip = torch.tensor([[-float("inf"), float("inf")]], device=self.device).repeat(
                    self.num_envs, self.action_dim, 1
                )
                index_list, _, value_list = string_utils.resolve_matching_names_values(self.cfg.clip, self._joint_names)
                self._clip[:, index_list] = torch.tensor(value_list, device=self.device)
            else:
                raise ValueError(f"Unsupported clip type: {type(cfg.clip)}. Supported types are dict.")

    @property
    def action_dim(self) -> int:
        return self._num_joints

    @property
    def raw_actions(self) -> torch.Tensor:
        return self._raw_actions

    @property
    def processed_actions(self) -> torch.Tensor:
        return self._processed_actions

    def _offset_tensor(self) -> torch.Tensor | None:
        if isinstance(self._offset, torch.Tensor):
            return self._offset
        if self._offset == 0.0:
            return None
        self._offset = torch.full_like(self._raw_actions, self._offset)
        return self._offset

    def _launch(self, actions: torch.Tensor, action_buf: torch.Tensor, prev_buf: torch.Tensor | None):
        a, ld, scale_vec, scale, off = self._action_args(actions, action_buf, prev_buf)
        L.check(L.lib.mdp_joint_action_process(self.num_envs, self.action_dim, a.data_ptr(), ld,
                                               action_buf.data_ptr(), L.ptr(prev_buf) if prev_buf is not None else None,
                                               self._processed_actions.data_ptr(),
                                               L.ptr(scale_vec) if scale_vec is not None else None, float(scale),
                                               L.ptr(off) if off is not None else None,
                                               L.ptr(self._clip) if self._clip is not None else None, L.stream()))

    def _action_args(self, actions: torch.Tensor, action_buf: torch.Tensor, prev_buf: torch.Tensor | None):
        a, ld = L.rows(actions.to(self.device), "actions")
        scale_vec = None
        scale = 1.0
        if isinstance(self._scale, torch.Tensor):
            # per-env scale tensors are constant over envs in the reference cfgs: the kernel takes the first row
            scale_vec = self._scale[0].contiguous()
            self._scale_vec_keep = scale_vec
        else:
            scale = self._scale
        off = self._offset_tensor()
        return a, ld, scale_vec, scale, off

    def can_fold_into_actuator(self) -> bool:
        """Whether the first actuator launch of the step can process this term's action itself
        (``mdp_actuator_*_compute_with_action``): a position target for every joint of an articulation whose single
        actuator group accepts it."""
        return False

    def fill_joint_action(self, actions: torch.Tensor, action_buf: torch.Tensor, prev_buf: torch.Tensor) -> "L.MdpJointAction":
        """The arguments of ``process_actions_fused`` as the record the actuator launch takes."""
        a, ld, scale_vec, scale, off = self._action_args(actions, action_buf, prev_buf)
        self._raw_actions = action_buf
        d = L.MdpJointAction()
        d.actions_in, d.ld_in = a.data_ptr(), ld
        d.action, d.prev_action = action_buf.data_ptr(), prev_buf.data_ptr()
        d.processed = self._processed_actions.data_ptr()
        d.scale_vec = scale_vec.data_ptr() if scale_vec is not None else None
        d.scale = float(scale)
        d.offset = off.data_ptr() if off is not None else None
        d.clip = self._clip.data_ptr() if self._clip is not None else None
        self._fold_keep = (a, off, scale_vec)
        return d

    def process_actions(self, actions: torch.Tensor):
        """joint_actions.py:130-139"""
        self._launch(actions, self._raw_actions, None)

    def process_actions_fused(self, actions: torch.Tensor, action_buf: torch.Tensor, prev_buf: torch.Tensor):
        """ActionManager bookkeeping (action_manager.py:331-332) + process_actions in one launch; the raw
        action buffer of the term aliases the manager's action buffer."""
        self._raw_actions = action_buf
        self._launch(actions, action_buf, prev_buf)

    def reset(self, env_ids: Sequence[int] | None = None) -> None:
        self._raw_actions[env_ids if env_ids is not None else slice(None)] = 0.0


class JointPositionAction(JointAction):
    def __init__(self, cfg, env):
        super().__init__(cfg, env)
        if cfg.use_default_offset:
            self._offset = self._asset.data.default_joint_pos[:, self._joint_ids].clone()
        if self._joint_ids == slice(None):
            # all joints: write the processed action straight into the articulation's target buffer, so
            # apply_actions() (called every physics sub-step) has nothing left to copy
            self._processed_actions = self._asset.data.joint_pos_target

    def can_fold_into_actuator(self) -> bool:
        return (self._joint_ids == slice(None) and self.action_dim == self._asset.num_joints
                and self._processed_actions.data_ptr() == self._asset.data.joint_pos_target.data_ptr()
                and getattr(self._asset, "can_fold_joint_action", False))

    def apply_actions(self):
        self._asset.set_joint_position_target(self.processed_actions, joint_ids=self._joint_ids)


class JointVelocityAction(JointAction):
    def __init__(self, cfg, env):
        super().__init__(cfg, env)
        if cfg.use_default_offset:
            self._offset = self._asset.data.default_joint_vel[:, self._joint_ids].clone()

    def apply_actions(self):
        self._asset.set_joint_velocity_target(self.processed_actions, joint_ids=self._joint_ids)


class JointEffortAction(JointAction):
    def apply_actions(self):
        self._asset.set_joint_effort_target(self.processed_actions, joint_ids=self._joint_ids)
