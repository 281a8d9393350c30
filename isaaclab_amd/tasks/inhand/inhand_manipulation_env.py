"""``InHandManipulationEnv`` (``TASKS/direct/inhand_manipulation/inhand_manipulation_env.py:29-433``) over a synthetic
physics feed, stepped in the ``DirectRLEnv.step`` order (``LAB/envs/direct_rl_env.py:293-380``).

The method names, buffers and return conventions are the reference's; the arithmetic of every method is one HIP
kernel behind the C ABI (``mdp_inhand_*``).  ``consecutive_successes`` is the one cross-env reduction on the path
(:424-431): its two partial sums are produced in fp64 per rank, optionally all-reduced (RCCL / gloo) and folded in
by ``mdp_inhand_finish_rewards`` so the logged value is independent of how envs are sharded.
Reset randomisation (``_reset_idx`` / ``_reset_target_pose``, :201-260) consumes the global torch generator in the
reference through ``nonzero()`` id lists (a host sync per step); here both are masked launches over the device generator
(``mdp_inhand_reset_idx``, the goal reset inside ``mdp_inhand_dones_rewards_step``), so the whole step incl. resets is
hipGraph-capturable.  ``inject_uniforms`` hands them recorded draws for parity tests.  The id-list forms
(``_reset_idx(env_ids)``, ``_reset_target_pose(env_ids)``) are kept with the reference's signatures on the same kernels.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from ... import _lib as L
from ...assets import SyntheticPhysicsView
from ...assets.articulation_data import ArticulationData
from ...assets.rigid_object import RigidObject
from .shadow_hand_env_cfg import ShadowHandEnvCfg


class _ObjectView:
    """Free-body physics view: ``get_transforms`` (N,7) xyzw + ``get_root_velocities`` (N,6)."""

    def __init__(self, n: int, device: str):
        self.count = n
        self.transforms = torch.zeros(n, 7, device=device)
        self.transforms[:, 6] = 1.0
        self.velocities = torch.zeros(n, 6, device=device)

    def get_transforms(self): return self.transforms
    def get_root_velocities(self): return self.velocities


class _Hand:
    def __init__(self, cfg: ShadowHandEnvCfg, view: SyntheticPhysicsView, device: str):
        self.joint_names = list(cfg.joint_names)
        self.body_names = list(cfg.body_names)
        self.num_joints = len(self.joint_names)
        self.num_bodies = len(self.body_names)
        self.root_physx_view = view
        self.data = ArticulationData(view, device, body_names=self.body_names, joint_names=self.joint_names)
        self.joint_pos_target = torch.zeros(view.count, self.num_joints, device=device)

    def set_joint_position_target(self, target: torch.Tensor, joint_ids=None, env_ids=None):
        """articulation.py:807-830 (``joint_ids`` as a device index tensor keeps the call free of host-to-device
        copies, i.e. capturable in a hipGraph)"""
        if env_ids is None and isinstance(joint_ids, torch.Tensor):
            self.joint_pos_target.index_copy_(1, joint_ids, target)
            return
        rows = slice(None) if env_ids is None else env_ids
        cols = slice(None) if joint_ids is None else joint_ids
        self.joint_pos_target[rows, cols] = target


class InHandManipulationEnv:
    cfg: ShadowHandEnvCfg

    def __init__(self, cfg: ShadowHandEnvCfg, device: str = "cuda:0", env_origins: torch.Tensor | None = None,
                 joint_pos_limits: torch.Tensor | None = None, group=None):
        self.cfg = cfg
        self.device = device
        self.num_envs = n = cfg.num_envs
        self.group = group              # torch.distributed process group for the sharded run (None = single rank)
        self.extras: dict = {}
        nj, nb = len(cfg.joint_names), len(cfg.body_names)
        self.hand_view = SyntheticPhysicsView(n, nj, nb, device, num_contact_bodies=0,
                                              with_joint_forces=bool(cfg.asymmetric_obs))
        self.object_view = _ObjectView(n, device)
        self.hand = _Hand(cfg, self.hand_view, device)
        self.object = RigidObject(self.object_view, device)
        self.object.data.default_root_state[:, :3] = torch.tensor(cfg.object_init_pos, device=device)
        self.scene = type("Scene", (), {})()
        self.scene.env_origins = (torch.zeros(n, 3, device=device) if env_origins is None
                                  else env_origins.to(device=device, dtype=torch.float32).contiguous())
        # :32-77
        self.num_hand_dofs = self.hand.num_joints
        f = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)
        self.hand_dof_targets, self.prev_targets, self.cur_targets = f(n, nj), f(n, nj), f(n, nj)
        self.actuated_dof_indices = sorted(self.hand.joint_names.index(j) for j in cfg.actuated_joint_names)
        self.finger_bodies = sorted(self.hand.body_names.index(b) for b in cfg.fingertip_body_names)
        self.num_fingertips = len(self.finger_bodies)
        if joint_pos_limits is None:
            joint_pos_limits = torch.tensor([-1.0, 1.0], device=device).repeat(n, nj, 1)
        joint_pos_limits = joint_pos_limits.to(device=device, dtype=torch.float32)
        self.hand_dof_lower_limits = joint_pos_limits[..., 0].contiguous()
        self.hand_dof_upper_limits = joint_pos_limits[..., 1].contiguous()
        self.reset_goal_buf = torch.zeros(n, dtype=torch.bool, device=device)
        self.in_hand_pos = self.object.data.default_root_state[:, 0:3].clone()
        self.in_hand_pos[:, 2] -= 0.04
        self.goal_rot = f(n, 4)
        self.goal_rot[:, 0] = 1.0
        self.goal_pos = torch.tensor([-0.2, -0.45, 0.68], device=device).repeat(n, 1)
        self.successes = f(n)
        self.consecutive_successes = f(1)
        self.x_unit_tensor = torch.tensor([1.0, 0, 0], device=device).repeat(n, 1)
        self.y_unit_tensor = torch.tensor([0, 1.0, 0], device=device).repeat(n, 1)
        self.z_unit_tensor = torch.tensor([0, 0, 1.0], device=device).repeat(n, 1)
        # DirectRLEnv buffers (direct_rl_env.py:142-150)
        self.actions = f(n, cfg.action_space)
        self.episode_length_buf = torch.zeros(n, dtype=torch.long, device=device)
        self.reset_terminated = torch.zeros(n, dtype=torch.bool, device=device)
        self.reset_time_outs = torch.zeros(n, dtype=torch.bool, device=device)
        self.reset_buf = torch.zeros(n, dtype=torch.bool, device=device)
        self.reward_buf = f(n)
        self.common_step_counter = 0
        self._sim_step_counter = 0
        # kernel-side constants
        self._act_idx = torch.tensor(self.actuated_dof_indices, dtype=torch.int32, device=device)
        self._fingers = torch.tensor(self.finger_bodies, dtype=torch.int32, device=device)
        self._partials = torch.zeros(2, dtype=torch.float64, device=device)
        self._obs = f(n, self.num_full_obs)
        self._goal_resets = torch.zeros(n, dtype=torch.bool, device=device)
        self._ccfg = self._make_cfg()
        self._critic = f(n, self.num_full_state) if cfg.asymmetric_obs else None
        if cfg.obs_type == "openai":
            self._obs = f(n, self.num_reduced_obs)
        # reset path: per-env generator epochs ({seed, epoch of env 0, ...}), injected uniforms for parity tests
        self._reset_rng = torch.zeros(1 + n, dtype=torch.int64, device=device)
        self._reset_rng[0] = int(getattr(cfg, "seed", 0) or 0) * 2654435761 % (1 << 62) + 12345
        self._injected_u = None
        # "masked": done envs are reset inside step() by masked launches (no host sync); "ids": the reference's
        # nonzero() + _reset_idx(env_ids) (host sync); "off": the caller resets
        self.auto_reset = "masked"
        self.reset_goals = True            # :172-175 (the reference always does it)
        # DirectRLEnv noise models (direct_rl_env.py:104-112)
        self._action_noise_model = (cfg.action_noise_model.class_type(cfg.action_noise_model, n, device)
                                    if getattr(cfg, "action_noise_model", None) else None)
        self._observation_noise_model = (cfg.observation_noise_model.class_type(cfg.observation_noise_model, n, device)
                                         if getattr(cfg, "observation_noise_model", None) else None)
        # DirectRLEnv.step's host statements around the five methods folded into their launches (mdp_inhand_*_to / _step /
        # _from_view, include/mdp_hip.h): same buffers, same values; False = one launch per reference statement
        self.fused_step = True

    @property
    def fused_step(self) -> bool:
        return self._fused_step

    @fused_step.setter
    def fused_step(self, value: bool) -> None:
        # the step form expects clean partial sums on entry (it leaves them clean itself); the per-method form leaves
        # the last step's sums behind -- switching forms starts from zero
        if getattr(self, "_pending_update", None) is not None:
            self._flush_hand_update()
        self._fused_step = bool(value)
        self._partials.zero_()
        self._finish_pending = False

    # -- properties (direct_rl_env.py:228-248) ----------------------------------------------------------
    @property
    def physics_dt(self) -> float: return self.cfg.sim_dt
    @property
    def step_dt(self) -> float: return self.cfg.sim_dt * self.cfg.decimation
    @property
    def max_episode_length_s(self) -> float: return self.cfg.episode_length_s
    @property
    def max_episode_length(self) -> int: return math.ceil(self.max_episode_length_s / self.step_dt)
    @property
    def num_full_obs(self) -> int:
        return 2 * self.num_hand_dofs + 13 + 11 + 13 * self.num_fingertips + self.cfg.action_space

    @property
    def num_reduced_obs(self) -> int:
        return 3 * self.num_fingertips + 3 + 4 + self.cfg.action_space
    @property
    def num_full_state(self) -> int:
        return self.num_full_obs + 6 * self.num_fingertips

    def inject_uniforms(self, u: torch.Tensor | None) -> None:
        """Parity tests: ``u`` (N, >= 7 + 2 * num_dofs) replaces the device generator of the reset path -- per env the
        draws in the reference's order (goal 2 | position 3 | rotation 2 | dof positions D | dof velocities D); the goal
        reset inside ``_get_rewards`` reads columns [0, 2)."""
        self._injected_u = None if u is None else L.f32(u, "u").contiguous()

    def _goal_reset_args(self, mask: torch.Tensor | None) -> L.MdpInhandGoalReset:
        g = L.MdpInhandGoalReset()
        g.env_mask = L.ptr(mask)
        u = self._injected_u
        g.injected_u, g.ld_injected = (L.ptr(u), u.stride(0)) if u is not None else (None, 0)
        g.rng_state = L.ptr(self._reset_rng)
        g.x_unit, g.y_unit = L.ptr(self.x_unit_tensor), L.ptr(self.y_unit_tensor)
        g.goal_rot, g.reset_goal_buf = L.ptr(self.goal_rot), L.ptr(self.reset_goal_buf)
        return g

    def _make_cfg(self) -> L.MdpInhandCfg:
        c, k = L.MdpInhandCfg(), self.cfg
        c.num_envs, c.num_dofs, c.num_actuated = self.num_envs, self.num_hand_dofs, len(self.actuated_dof_indices)
        c.num_fingertips, c.num_bodies = self.num_fingertips, self.hand.num_bodies
        c.act_moving_average, c.vel_obs_scale = k.act_moving_average, k.vel_obs_scale
        c.dist_reward_scale, c.rot_reward_scale, c.rot_eps = k.dist_reward_scale, k.rot_reward_scale, k.rot_eps
        c.action_penalty_scale, c.success_tolerance = k.action_penalty_scale, k.success_tolerance
        c.reach_goal_bonus, c.fall_dist, c.fall_penalty, c.av_factor = k.reach_goal_bonus, k.fall_dist, k.fall_penalty, k.av_factor
        c.max_consecutive_success, c.max_episode_length = k.max_consecutive_success, self.max_episode_length
        return c

    def refresh_cfg(self):
        """Call after editing ``self.cfg`` scalars (the kernels take the constants by value)."""
        self._ccfg = self._make_cfg()

    # -- DirectRLEnv hooks --------------------------------------------------------------------------------
    def _pre_physics_step(self, actions: torch.Tensor) -> None:
        """:94-95"""
        if self.fused_step and actions.dtype == torch.float32 and actions.is_contiguous() and actions.shape == self.actions.shape:
            self._actions_in = actions       # the first _apply_action launch of the step writes the copy
            return
        self._actions_in = None
        self.actions.copy_(actions)

    def _apply_action(self) -> None:
        """:97-117: scale -> moving average -> saturate on the actuated joints, targets handed to the articulation."""
        if self.fused_step:
            tgt = self.hand.joint_pos_target
            src = getattr(self, "_actions_in", None)
            self._actions_in = None
            pend, self._pending_update = getattr(self, "_pending_update", None), None
            args = (C.byref(self._ccfg), L.ptr(src if src is not None else self.actions),
                    L.ptr(self.actions) if src is not None else None, L.ptr(self._act_idx),
                    L.ptr(self.hand_dof_lower_limits), L.ptr(self.hand_dof_upper_limits),
                    L.ptr(self.cur_targets), L.ptr(self.prev_targets), L.ptr(tgt), tgt.stride(0))
            if pend is not None:     # the previous sub-step's hand.data.update(dt) rides in this launch
                L.check(L.lib.mdp_inhand_apply_action_to_with_update(*args, C.byref(pend[0]), float(pend[1]), L.stream()))
            else:
                L.check(L.lib.mdp_inhand_apply_action_to(*args, L.stream()))
            return
        L.check(L.lib.mdp_inhand_apply_action(C.byref(self._ccfg), L.ptr(self.actions), L.ptr(self._act_idx),
                                              L.ptr(self.hand_dof_lower_limits), L.ptr(self.hand_dof_upper_limits),
                                              L.ptr(self.cur_targets), L.ptr(self.prev_targets), L.stream()))
        idx = self._act_idx_long()
        self.hand.set_joint_position_target(self.cur_targets.index_select(1, idx), joint_ids=idx)

    def _act_idx_long(self) -> torch.Tensor:
        """``actuated_dof_indices`` as a device int64 tensor (rebuilt when ``_act_idx`` is replaced)."""
        cached = getattr(self, "_act_idx_cache", None)
        if cached is None or cached[0] is not self._act_idx:
            cached = (self._act_idx, self._act_idx.to(torch.long))
            self._act_idx_cache = cached
        return cached[1]

    def _get_dones(self) -> tuple[torch.Tensor, torch.Tensor]:
        """:179-199 (``_compute_intermediate_values`` is folded into the kernels: they read the raw state)."""
        out_of_reach = torch.empty(self.num_envs, dtype=torch.bool, device=self.device)
        time_out = torch.empty(self.num_envs, dtype=torch.bool, device=self.device)
        L.check(L.lib.mdp_inhand_dones(C.byref(self._ccfg), L.ptr(self.object.data.root_state_w),
                                       L.ptr(self.scene.env_origins), L.ptr(self.in_hand_pos), L.ptr(self.goal_rot),
                                       L.ptr(self.successes), L.ptr(self.episode_length_buf), L.ptr(out_of_reach),
                                       L.ptr(time_out), L.stream()))
        return out_of_reach, time_out

    def _dones_step(self) -> None:
        """direct_rl_env.py:343-346 as one launch: ``episode_length_buf += 1``, ``_get_dones`` written straight into
        ``reset_terminated`` / ``reset_time_outs``, ``reset_buf = reset_terminated | reset_time_outs``."""
        state, xf, vel = self._object_state_args()
        L.check(L.lib.mdp_inhand_dones_step(C.byref(self._ccfg), L.ptr(state),
                                            L.ptr(self.scene.env_origins), L.ptr(self.in_hand_pos), L.ptr(self.goal_rot),
                                            L.ptr(self.successes), L.ptr(self.episode_length_buf),
                                            L.ptr(self.reset_terminated), L.ptr(self.reset_time_outs),
                                            L.ptr(self.reset_buf), L.ptr(xf), L.ptr(vel), L.stream()))

    def _object_state_args(self):
        """(root_state buffer, view transforms, view velocities): with a stale ``root_state_w`` the launch assembles it
        from the view (what the property's pack launch would do) and the launches behind it read the buffer it leaves."""
        od = self.object.data
        if od._root_state_w.timestamp < od._sim_timestamp:
            od._root_state_w.timestamp = od._sim_timestamp
            return od._root_state_w.data, od._root_physx_view.get_transforms(), od._root_physx_view.get_root_velocities()
        return od.root_state_w, None, None

    def _dones_rewards_step(self) -> None:
        """direct_rl_env.py:343-349 as ONE launch (``mdp_inhand_dones_rewards_step``): ``_dones_step`` and the fused
        ``_get_rewards`` per env in the same thread, with the last sub-step's deferred ``hand.data.update(dt)`` on blocks
        of the same launch; the ``consecutive_successes`` average follows in the observation launch."""
        state, xf, vel = self._object_state_args()
        pend, self._pending_update = getattr(self, "_pending_update", None), None
        L.check(L.lib.mdp_inhand_dones_rewards_step(
            C.byref(self._ccfg), L.ptr(state), L.ptr(self.scene.env_origins), L.ptr(self.in_hand_pos), L.ptr(self.goal_rot),
            L.ptr(self.successes), L.ptr(self.episode_length_buf), L.ptr(self.reset_terminated),
            L.ptr(self.reset_time_outs), L.ptr(self.reset_buf), L.ptr(xf), L.ptr(vel), L.ptr(self.actions),
            L.ptr(self.reset_goal_buf), L.ptr(self.reward_buf), L.ptr(self._partials),
            C.byref(pend[0]) if pend is not None else None, float(pend[1]) if pend is not None else 0.0,
            C.byref(self._goal_reset_args(None)) if self.reset_goals else None, L.stream()))
        if self.group is not None:
            import torch.distributed as dist

            dist.all_reduce(self._partials, group=self.group)
        self._finish_pending = True
        if "log" not in self.extras:
            self.extras["log"] = dict()
        # A VIEW of the live one-element buffer, not a value like the reference's ``.mean()`` (:431): the observation
        # launch of this same step writes the moving average into it, so it is final once ``step()`` returns and it
        # changes again with the next step -- a logger that keeps it across steps must ``.clone()`` / ``float()`` it.
        self.extras["log"]["consecutive_successes"] = self.consecutive_successes[0]

    def _get_rewards(self, defer_finish: bool = False) -> torch.Tensor:
        """:140-177 / compute_rewards :377-433.  ``defer_finish`` (only ``step()`` passes it): the moving average of
        ``consecutive_successes`` is folded into the observation launch that follows in the same step."""
        if self.fused_step:
            L.check(L.lib.mdp_inhand_rewards_step(C.byref(self._ccfg), L.ptr(self.object.data.root_state_w),
                                                  L.ptr(self.scene.env_origins), L.ptr(self.in_hand_pos),
                                                  L.ptr(self.goal_rot), L.ptr(self.actions), L.ptr(self.reset_buf),
                                                  L.ptr(self.reset_goal_buf), L.ptr(self.successes),
                                                  L.ptr(self.reward_buf), L.ptr(self._partials), L.stream()))
            if self.group is not None:
                import torch.distributed as dist

                dist.all_reduce(self._partials, group=self.group)
            if defer_finish:
                self._finish_pending = True
            else:
                L.check(L.lib.mdp_inhand_finish_rewards_step(C.byref(self._ccfg), L.ptr(self._partials),
                                                             L.ptr(self.consecutive_successes), L.stream()))
            if "log" not in self.extras:
                self.extras["log"] = dict()
            # the mean of the one-element buffer (:431) is that element: a view, no reduction kernel
            self.extras["log"]["consecutive_successes"] = self.consecutive_successes[0]
            return self.reward_buf
        L.check(L.lib.mdp_inhand_rewards(C.byref(self._ccfg), L.ptr(self.object.data.root_state_w),
                                         L.ptr(self.scene.env_origins), L.ptr(self.in_hand_pos), L.ptr(self.goal_rot),
                                         L.ptr(self.actions), L.ptr(self.reset_buf), L.ptr(self.reset_goal_buf),
                                         L.ptr(self.successes), L.ptr(self.reward_buf), L.ptr(self._goal_resets),
                                         L.ptr(self._partials), L.stream()))
        if self.group is not None:
            import torch.distributed as dist

            dist.all_reduce(self._partials, group=self.group)   # [sum resets, sum successes*resets], exact in fp64
        L.check(L.lib.mdp_inhand_finish_rewards(C.byref(self._ccfg), L.ptr(self._partials),
                                                L.ptr(self.consecutive_successes), L.stream()))
        self.reset_goal_buf.copy_(self._goal_resets)
        if "log" not in self.extras:
            self.extras["log"] = dict()
        self.extras["log"]["consecutive_successes"] = self.consecutive_successes.mean()
        return self.reward_buf

    _OBS_KIND = {"full": 0, "openai": 1}      # MDP_INHAND_OBS_FULL / _REDUCED; the critic is MDP_INHAND_OBS_FULL_STATE (2)

    def _observations_launch(self, policy_kind: int, policy: torch.Tensor | None, critic: torch.Tensor | None,
                             reset_mask: torch.Tensor | None = None) -> None:
        """``mdp_inhand_observations``: the policy layout and (asymmetric) ``compute_full_state`` in one launch; with
        ``reset_mask`` the blocks first run ``_reset_idx`` for their envs of the mask (direct_rl_env.py:355-372)."""
        a = L.MdpInhandObsArgs()
        rst = None
        if reset_mask is not None:
            rst = self._reset_args(reset_mask)
            a.reset = C.addressof(rst)
        view = self.hand.root_physx_view
        a.joint_pos, a.joint_vel = L.ptr(self.hand.data.joint_pos), L.ptr(self.hand.data.joint_vel)
        a.lower, a.upper = L.ptr(self.hand_dof_lower_limits), L.ptr(self.hand_dof_upper_limits)
        a.object_root_state, a.env_origins = L.ptr(self.object.data.root_state_w), L.ptr(self.scene.env_origins)
        a.in_hand_pos, a.goal_rot = L.ptr(self.in_hand_pos), L.ptr(self.goal_rot)
        if self.fused_step:
            a.link_transforms, a.link_velocities = L.ptr(view.get_link_transforms()), L.ptr(view.get_link_velocities())
        else:
            a.body_state_w = L.ptr(self.hand.data.body_state_w)
        a.finger_bodies, a.actions, a.action_dim = L.ptr(self._fingers), L.ptr(self.actions), self.cfg.action_space
        a.force_torque_obs_scale = float(self.cfg.force_torque_obs_scale)
        a.policy_kind, a.critic_kind = (policy_kind if policy is not None else -1), (2 if critic is not None else -1)
        if policy is not None:
            a.policy, a.ld_policy = L.ptr(policy), policy.stride(0)
        if critic is not None:
            # :121-123 fingertip_force_sensors = get_link_incoming_joint_force()[:, finger_bodies], gathered by the kernel
            self.fingertip_force_sensors = view.get_link_incoming_joint_force()[:, self.finger_bodies]
            a.link_incoming_joint_force = L.ptr(view.get_link_incoming_joint_force())
            a.critic, a.ld_critic = L.ptr(critic), critic.stride(0)
        finish, self._finish_pending = getattr(self, "_finish_pending", False), False
        if finish:
            a.partials, a.consecutive_successes = L.ptr(self._partials), L.ptr(self.consecutive_successes)
        L.check(L.lib.mdp_inhand_observations(C.byref(self._ccfg), C.byref(a), L.stream()))
        if rst is not None:
            self._reset_noise_models(reset_mask)

    def _policy_buffer(self, width: int) -> torch.Tensor:
        if self._obs.shape[1] != width:
            self._obs = torch.zeros(self.num_envs, width, dtype=torch.float32, device=self.device)
        return self._obs

    def compute_full_observations(self) -> torch.Tensor:
        """:298-322"""
        out = self._policy_buffer(self.num_full_obs)
        self._observations_launch(0, out, None)
        return out

    def compute_reduced_observations(self) -> torch.Tensor:
        """:281-296 (fingertip positions, object position, relative target orientation, actions)"""
        out = self._policy_buffer(self.num_reduced_obs)
        self._observations_launch(1, out, None)
        return out

    def compute_full_state(self) -> torch.Tensor:
        """:324-350"""
        if self._critic is None:
            self._critic = torch.zeros(self.num_envs, self.num_full_state, dtype=torch.float32, device=self.device)
        self._observations_launch(0, None, self._critic)
        return self._critic

    def _get_observations(self, reset_mask: torch.Tensor | None = None) -> dict:
        """:119-138: policy layout by ``obs_type``; with ``asymmetric_obs`` the critic's ``compute_full_state`` comes out of
        the same launch.  ``reset_mask`` (only ``step()`` passes it): the masked ``_reset_idx`` rides in that launch."""
        if self.cfg.obs_type not in self._OBS_KIND:
            raise ValueError(f"Unknown observations type: {self.cfg.obs_type!r}")    # the reference prints and fails later
        kind = self._OBS_KIND[self.cfg.obs_type]
        out = self._policy_buffer(self.num_full_obs if kind == 0 else self.num_reduced_obs)
        if not self.cfg.asymmetric_obs:
            self._observations_launch(kind, out, None, reset_mask)
            return {"policy": out}
        if self._critic is None:
            self._critic = torch.zeros(self.num_envs, self.num_full_state, dtype=torch.float32, device=self.device)
        self._observations_launch(kind, out, self._critic, reset_mask)
        return {"policy": out, "critic": self._critic}

    # -- reset path ----------------------------------------------------------------------------------------
    def _ids_to_mask(self, env_ids) -> torch.Tensor | None:
        if env_ids is None:
            return None
        mask = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        mask[env_ids] = True
        return mask

    def _reset_target_pose(self, env_ids):
        """:248-260 for an id list (the reference's signature); the launch is the masked one."""
        self._reset_target_pose_masked(self._ids_to_mask(env_ids))

    def _reset_target_pose_masked(self, mask: torch.Tensor | None) -> None:
        L.check(L.lib.mdp_inhand_reset_target_pose(C.byref(self._ccfg), C.byref(self._goal_reset_args(mask)), L.stream()))

    def _reset_idx(self, env_ids):
        """:201-246 + direct_rl_env.py:560-581 for an id list (the reference's signature)."""
        self._reset_idx_masked(self._ids_to_mask(env_ids))

    def _reset_idx_masked(self, mask: torch.Tensor | None) -> None:
        """The same for the envs of a bool mask (None = all): one launch, no host synchronisation."""
        a = self._reset_args(mask)
        L.check(L.lib.mdp_inhand_reset_idx(C.byref(self._ccfg), C.byref(a), L.stream()))
        self._reset_noise_models(mask)

    def _reset_noise_models(self, mask: torch.Tensor | None) -> None:
        """direct_rl_env.py:574-578"""
        for model in (self._action_noise_model, self._observation_noise_model):
            if model is not None:
                model.reset_masked(mask)

    def _reset_args(self, mask: torch.Tensor | None) -> L.MdpInhandReset:
        """``mdp_inhand_reset_t`` over the env's buffers (the caller keeps the struct alive until its launch is queued)."""
        if getattr(self, "_pending_update", None) is not None:
            self._flush_hand_update()                       # the reset writes the joint buffers the update refreshes
        hd, od, hv, ov = self.hand.data, self.object.data, self.hand.root_physx_view, self.object_view
        a = L.MdpInhandReset()
        a.goal = self._goal_reset_args(mask)
        k = self.cfg
        a.reset_position_noise, a.reset_dof_pos_noise = float(k.reset_position_noise), float(k.reset_dof_pos_noise)
        a.reset_dof_vel_noise = float(k.reset_dof_vel_noise)
        a.object_default_root_state, a.env_origins = L.ptr(od.default_root_state), L.ptr(self.scene.env_origins)
        # write_root_pose_to_sim / write_root_velocity_to_sim: the data buffer (kept fresh) and the view's buffers
        od._root_state_w.timestamp = od._sim_timestamp
        a.object_root_state_w = L.ptr(od._root_state_w.data)
        a.sim_object_transforms, a.sim_object_velocities = L.ptr(ov.get_transforms()), L.ptr(ov.get_root_velocities())
        a.default_joint_pos, a.default_joint_vel = L.ptr(hd.default_joint_pos), L.ptr(hd.default_joint_vel)
        a.lower, a.upper = L.ptr(self.hand_dof_lower_limits), L.ptr(self.hand_dof_upper_limits)
        a.prev_targets, a.cur_targets = L.ptr(self.prev_targets), L.ptr(self.cur_targets)
        a.hand_dof_targets = L.ptr(self.hand_dof_targets)
        a.joint_pos_target, a.ld_target = L.ptr(self.hand.joint_pos_target), self.hand.joint_pos_target.stride(0)
        q_sim, qd_sim = hv.get_dof_positions(), hv.get_dof_velocities()
        a.joint_pos, a.joint_vel = L.ptr(hd.joint_pos), L.ptr(hd.joint_vel)
        a.prev_joint_vel, a.joint_acc = L.ptr(hd._previous_joint_vel), L.ptr(hd.joint_acc)
        a.sim_dof_positions, a.sim_dof_velocities, a.ld_sim = L.ptr(q_sim), L.ptr(qd_sim), q_sim.stride(0)
        a.successes, a.episode_length_buf = L.ptr(self.successes), L.ptr(self.episode_length_buf)
        return a

    def _defer_hand_update(self, dt: float) -> None:
        """``self.hand.data.update(dt)`` described now (the view's buffers of THIS sub-step) and executed by the next
        launch of the step (the next sub-step's ``_apply_action`` or ``_dones_rewards_step``): nothing in between reads
        the hand's buffers."""
        self._flush_hand_update()
        d = L.MdpSceneUpdate()
        self.hand.data.fill_scene_update(d, dt)
        self.hand.data.commit_scene_update(dt)
        self._pending_update = (d, dt)

    def _flush_hand_update(self) -> None:
        """Runs a deferred hand update on its own (a caller that reads the hand outside ``step()``)."""
        pend, self._pending_update = getattr(self, "_pending_update", None), None
        if pend is not None:
            L.check(L.lib.mdp_scene_update(C.byref(pend[0]), float(pend[1]), L.stream()))

    def _update_hand_fused(self, dt: float) -> None:
        """``self.hand.data.update(dt)`` (articulation_data.py:77-82: timestamp, joint state refresh, joint_acc finite
        difference) as the one launch InteractiveScene.update uses (``mdp_scene_update``) instead of two copies and a
        kernel."""
        d = L.MdpSceneUpdate()
        self.hand.data.fill_scene_update(d, dt)
        L.check(L.lib.mdp_scene_update(C.byref(d), float(dt), L.stream()))
        self.hand.data.commit_scene_update(dt)

    # -- stepping (direct_rl_env.py:293-380) ------------------------------------------------------------
    def step(self, action: torch.Tensor, reset_goals: bool | None = None):
        if reset_goals is not None:
            self.reset_goals = bool(reset_goals)
        action = action.to(self.device)
        if self._action_noise_model is not None:           # direct_rl_env.py:319-320
            action = self._action_noise_model.apply(action)
        self._pre_physics_step(action)
        for _ in range(self.cfg.decimation):
            self._sim_step_counter += 1
            self._apply_action()
            self.hand_view.advance()                      # sim.step(): synthetic / replayed
            if self.fused_step:
                self._defer_hand_update(self.physics_dt)
            else:
                self.hand.data.update(self.physics_dt)
            self.object.update(self.physics_dt)
        self.common_step_counter += 1
        if self.fused_step:
            self._dones_rewards_step()                     # incl. the goal reset of _get_rewards (:172-175)
        else:
            self.episode_length_buf += 1
            terminated, time_outs = self._get_dones()
            self.reset_terminated[:] = terminated
            self.reset_time_outs[:] = time_outs
            torch.logical_or(self.reset_terminated, self.reset_time_outs, out=self.reset_buf)
            self.reward_buf = self._get_rewards()
            if self.reset_goals:                           # :172-175, masked: no nonzero()
                self._reset_target_pose_masked(self.reset_goal_buf)
        # direct_rl_env.py:355-364
        reset_mask = None
        if self.auto_reset == "masked":
            if self.fused_step:
                reset_mask = self.reset_buf                # rides in the observation launch
            else:
                self._reset_idx_masked(self.reset_buf)
        elif self.auto_reset == "ids":
            reset_env_ids = self.reset_buf.nonzero(as_tuple=False).squeeze(-1)
            if len(reset_env_ids) > 0:
                self._reset_idx(reset_env_ids)
        self.obs_buf = self._get_observations(reset_mask)
        if self._observation_noise_model is not None:      # direct_rl_env.py:376-377 (policy only, never the critic)
            self.obs_buf["policy"] = self._observation_noise_model.apply(self.obs_buf["policy"])
        return self.obs_buf, self.reward_buf, self.reset_terminated, self.reset_time_outs, self.extras
