"""``Imu`` (``LAB/sensors/imu/imu.py:30-215``): a sensor that is pure math over one rigid body's state.
Same cfg fields, ``update(dt)`` / ``reset(env_ids)`` / lazily evaluated ``data``; the buffer fill is one launch
(``mdp_imu_update``) that only touches the outdated envs and clears their flags itself -- no ``nonzero()``."""
from __future__ import annotations

import ctypes as C
from collections.abc import Sequence
from dataclasses import dataclass

import torch

from ... import _lib as L
from ...utils.configclass import configclass
from ..sensor_base import SensorBase, SensorBaseCfg


@dataclass
class ImuData:
    """imu_data.py:13-48"""

    pos_w: torch.Tensor | None = None
    quat_w: torch.Tensor | None = None
    lin_vel_b: torch.Tensor | None = None
    ang_vel_b: torch.Tensor | None = None
    lin_acc_b: torch.Tensor | None = None
    ang_acc_b: torch.Tensor | None = None


class Imu(SensorBase):
    def __init__(self, cfg: "ImuCfg", body_view, num_envs: int, device: str, sim_physics_dt: float):
        """``body_view``: the rigid-body view of the link the sensor sits on (``get_transforms`` (N,7) xyzw,
        ``get_velocities`` (N,6), ``get_coms`` (N,7))."""
        super().__init__(cfg, num_envs, device, sim_physics_dt)
        self._view = body_view
        z3 = lambda: torch.zeros(num_envs, 3, device=device)
        self._data = ImuData(pos_w=z3(), quat_w=torch.zeros(num_envs, 4, device=device), lin_vel_b=z3(), ang_vel_b=z3(),
                             lin_acc_b=z3(), ang_acc_b=z3())
        self._data.quat_w[:, 0] = 1.0
        self._prev_lin_vel_w, self._prev_ang_vel_w = z3(), z3()
        self._offset_pos_b = torch.tensor(list(cfg.offset.pos), dtype=torch.float32, device=device).repeat(num_envs, 1)
        self._offset_quat_b = torch.tensor(list(cfg.offset.rot), dtype=torch.float32, device=device).repeat(num_envs, 1)
        self._gravity_bias_w = torch.tensor(list(cfg.gravity_bias), dtype=torch.float32, device=device).repeat(num_envs, 1)

    @property
    def data(self) -> ImuData:
        self._update_outdated_buffers()
        return self._data

    def reset(self, env_ids: Sequence[int] | None = None):
        """imu.py:90-102"""
        super().reset(env_ids)
        if env_ids is None:
            env_ids = slice(None)
        for buf in (self._data.quat_w, self._data.lin_vel_b, self._data.ang_vel_b, self._data.lin_acc_b, self._data.ang_acc_b):
            buf[env_ids] = 0.0

    def update(self, dt: float, force_recompute: bool = False):
        """imu.py:104-108 + sensor_base.py:197-205 (timestamps advance on the device)."""
        self._dt = dt
        L.check(L.lib.mdp_sensor_tick(self._num_envs, L.ptr(self._timestamp), L.ptr(self._timestamp_last_update),
                                      L.ptr(self._is_outdated), float(dt), float(self.cfg.update_period), L.stream()))
        if force_recompute or self._is_visualizing or (self.cfg.history_length > 0):
            self._update_outdated_buffers()

    def _update_outdated_buffers(self):
        if not hasattr(self, "_dt"):
            raise RuntimeError(
                "The update function must be called before the data buffers are accessed the first time."
            )
        a = L.MdpImuArgs()
        t, v, coms = self._view.get_transforms(), self._view.get_velocities(), self._view.get_coms()
        for name, x, w in (("transforms", t, 7), ("velocities", v, 6)):
            if x.shape != (self._num_envs, w) or not x.is_contiguous() or x.dtype != torch.float32:
                raise ValueError(f"body view {name} must be a contiguous float32 ({self._num_envs}, {w}) tensor")
        if coms.dim() != 2 or coms.shape[0] != self._num_envs or coms.stride(1) != 1 or coms.shape[1] < 3:
            raise ValueError("body view coms must be (num_envs, >=3) with unit column stride")
        a.is_outdated, a.timestamp = self._is_outdated.data_ptr(), self._timestamp.data_ptr()
        a.timestamp_last_update = self._timestamp_last_update.data_ptr()
        a.transforms, a.velocities, a.coms, a.ld_coms = t.data_ptr(), v.data_ptr(), coms.data_ptr(), coms.stride(0)
        a.dt = float(self._dt)
        a.offset_pos_b, a.offset_quat_b = self._offset_pos_b.data_ptr(), self._offset_quat_b.data_ptr()
        a.gravity_bias_w = self._gravity_bias_w.data_ptr()
        a.prev_lin_vel_w, a.prev_ang_vel_w = self._prev_lin_vel_w.data_ptr(), self._prev_ang_vel_w.data_ptr()
        d = self._data
        a.pos_w, a.quat_w, a.lin_vel_b, a.ang_vel_b = d.pos_w.data_ptr(), d.quat_w.data_ptr(), d.lin_vel_b.data_ptr(), d.ang_vel_b.data_ptr()
        a.lin_acc_b, a.ang_acc_b = d.lin_acc_b.data_ptr(), d.ang_acc_b.data_ptr()
        L.check(L.lib.mdp_imu_update(self._num_envs, C.byref(a), L.stream(self._device)))


@configclass
class ImuCfg(SensorBaseCfg):
    """imu_cfg.py:18-50"""

    class_type: type = Imu

    @configclass
    class OffsetCfg:
        pos: tuple = (0.0, 0.0, 0.0)
        rot: tuple = (1.0, 0.0, 0.0, 0.0)

    offset: OffsetCfg = OffsetCfg()
    gravity_bias: tuple = (0.0, 0.0, 9.81)
