"""``UniformVelocityCommand`` (``LAB/envs/mdp/commands/velocity_command.py:27-160``).

``compute`` = metrics update + time_left decrement + resample where elapsed + heading control +
standing envs, all in ONE kernel (``mdp_velocity_command_compute``) with a device-side counter-based
generator: no ``nonzero()`` / ``len()`` host syncs (command_manager.py:162-163, velocity_command.py:149,159).
"""
from __future__ import annotations

import ctypes as C
from collections.abc import Sequence

import torch

from .... import _lib as L
from ....managers.command_manager import CommandTerm
from ....utils.buffers.circular_buffer import _mask_from_ids


class UniformVelocityCommand(CommandTerm):
    def __init__(self, cfg, env):
        super().__init__(cfg, env)
        if self.cfg.heading_command and self.cfg.ranges.heading is None:
            raise ValueError(
                "The velocity command has heading commands active (heading_command=True) but the `ranges.heading`"
                " parameter is set to None."
            )
        self.robot = env.scene[cfg.asset_name]
        self.vel_command_b = torch.zeros(self.num_envs, 3, device=self.device)
        self.heading_target = torch.zeros(self.num_envs, device=self.device)
        self.is_heading_env = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        self.is_standing_env = torch.zeros_like(self.is_heading_env)
        self.metrics["error_vel_xy"] = torch.zeros(self.num_envs, device=self.device)
        self.metrics["error_vel_yaw"] = torch.zeros(self.num_envs, device=self.device)
        seed = int(getattr(cfg, "seed", None) or (torch.initial_seed() & 0x7FFFFFFFFFFFFFFF))
        # {seed, epoch, ticket, pad, one epoch per manager block of mdp_post_physics_step (<= num_envs) ...}: advanced by the kernels
        # themselves (hipGraph-replay safe)
        self._rng_state = torch.zeros(4 + self.num_envs, dtype=torch.int64, device=self.device)
        self._rng_state[0] = seed
        self._ccfg = self._make_cfg()
        self.injected_uniforms: torch.Tensor | None = None  # (N,7) test hook: replaces the generator

    def _make_cfg(self) -> L.MdpVelocityCommandCfg:
        c, r = L.MdpVelocityCommandCfg(), self.cfg.ranges
        c.resampling_time_lo, c.resampling_time_hi = self.cfg.resampling_time_range
        c.lin_vel_x_lo, c.lin_vel_x_hi = r.lin_vel_x
        c.lin_vel_y_lo, c.lin_vel_y_hi = r.lin_vel_y
        c.ang_vel_z_lo, c.ang_vel_z_hi = r.ang_vel_z
        if r.heading is not None:
            c.heading_lo, c.heading_hi = r.heading
        c.rel_standing_envs = self.cfg.rel_standing_envs
        c.rel_heading_envs = self.cfg.rel_heading_envs
        c.heading_control_stiffness = self.cfg.heading_control_stiffness
        c.heading_command = 1 if self.cfg.heading_command else 0
        c.max_command_step = self.cfg.resampling_time_range[1] / self._env.step_dt
        return c

    def __str__(self) -> str:
        msg = "UniformVelocityCommand:\n"
        msg += f"\tCommand dimension: {tuple(self.command.shape[1:])}\n"
        msg += f"\tResampling time range: {self.cfg.resampling_time_range}\n"
        msg += f"\tHeading command: {self.cfg.heading_command}\n"
        if self.cfg.heading_command:
            msg += f"\tHeading probability: {self.cfg.rel_heading_envs}\n"
        msg += f"\tStanding probability: {self.cfg.rel_standing_envs}"
        return msg

    @property
    def command(self) -> torch.Tensor:
        return self.vel_command_b

    def _launch(self, dt: float, force_mask: torch.Tensor | None):
        st = self.robot.data.root_state_w
        inj = self.injected_uniforms
        L.check(L.lib.mdp_velocity_command_compute(
            C.byref(self._ccfg), self.num_envs, st.data_ptr(), st.stride(0), self.vel_command_b.data_ptr(),
            self.heading_target.data_ptr(), self.is_heading_env.data_ptr(), self.is_standing_env.data_ptr(),
            self.time_left.data_ptr(), self.command_counter.data_ptr(), self.metrics["error_vel_xy"].data_ptr(),
            self.metrics["error_vel_yaw"].data_ptr(), float(dt), self._rng_state.data_ptr(),
            L.ptr(inj) if inj is not None else None, L.ptr(force_mask) if force_mask is not None else None,
            L.stream(self.device)))

    def fill_post_physics(self, dt: float) -> "L.MdpVelocityCommandArgs":
        """The arguments of :meth:`compute` as the command member of ``mdp_post_physics_step``."""
        a = L.MdpVelocityCommandArgs()
        st = self.robot.data.root_state_w
        a.cfg = self._ccfg
        a.ld_state = st.stride(0)
        a.root_state = st.data_ptr()
        a.vel_command_b = self.vel_command_b.data_ptr()
        a.heading_target = self.heading_target.data_ptr()
        a.is_heading_env = self.is_heading_env.data_ptr()
        a.is_standing_env = self.is_standing_env.data_ptr()
        a.time_left = self.time_left.data_ptr()
        a.command_counter = self.command_counter.data_ptr()
        a.metric_error_vel_xy = self.metrics["error_vel_xy"].data_ptr()
        a.metric_error_vel_yaw = self.metrics["error_vel_yaw"].data_ptr()
        a.dt = float(dt)
        a.rng_state = self._rng_state.data_ptr()
        inj = self.injected_uniforms
        a.injected_u = inj.data_ptr() if inj is not None else None
        self._post_keep = (st, inj)
        return a

    def compute(self, dt: float):
        self._launch(dt, None)

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, float]:
        """command_manager.py:120-149: log + zero metrics, counter = 0, resample (then counter += 1)."""
        sel = slice(None) if env_ids is None else env_ids
        extras = {name: torch.mean(value[sel]).item() for name, value in self.metrics.items()}
        mask = _mask_from_ids(env_ids, self.num_envs, self.device)
        if mask is None:
            mask = torch.ones(self.num_envs, dtype=torch.bool, device=self.device)
        self._launch(0.0, mask)
        return extras

    # generic hooks of the base class are not used by the fused path
    def _update_metrics(self):
        raise NotImplementedError

    def _resample_command(self, env_ids):
        raise NotImplementedError

    def _update_command(self):
        raise NotImplementedError
