"""``ContactSensor`` (``LAB/sensors/contact_sensor/contact_sensor.py:136-373``): net-force history and the
air / contact time state machine, one fused kernel per ``update`` (``mdp_contact_sensor_update``)."""
from __future__ import annotations

from collections.abc import Sequence
from dataclasses import dataclass

import torch

from ... import _lib as L
from ...utils import string as string_utils
from ...utils.buffers.circular_buffer import _mask_from_ids
from ...utils.configclass import configclass
from ..sensor_base import SensorBase, SensorBaseCfg


@dataclass
class ContactSensorData:
    pos_w: torch.Tensor | None = None
    quat_w: torch.Tensor | None = None
    net_forces_w: torch.Tensor | None = None
    net_forces_w_history: torch.Tensor | None = None
    force_matrix_w: torch.Tensor | None = None
    last_air_time: torch.Tensor | None = None
    current_air_time: torch.Tensor | None = None
    last_contact_time: torch.Tensor | None = None
    current_contact_time: torch.Tensor | None = None


class ContactSensor(SensorBase):
    def __init__(self, cfg: "ContactSensorCfg", contact_physx_view, body_names: list[str], num_envs: int,
                 device: str, sim_physics_dt: float):
        super().__init__(cfg, num_envs, device, sim_physics_dt)
        self._contact_physx_view = contact_physx_view
        self._body_names = list(body_names)
        self._num_bodies = len(self._body_names)
        self._data = ContactSensorData()
        n, b = num_envs, self._num_bodies
        self._data.net_forces_w = torch.zeros(n, b, 3, device=device)
        if cfg.history_length > 0:
            self._data.net_forces_w_history = torch.zeros(n, cfg.history_length, b, 3, device=device)
        else:
            self._data.net_forces_w_history = self._data.net_forces_w.unsqueeze(1)
        if cfg.track_air_time:
            self._data.last_air_time = torch.zeros(n, b, device=device)
            self._data.current_air_time = torch.zeros(n, b, device=device)
            self._data.last_contact_time = torch.zeros(n, b, device=device)
            self._data.current_contact_time = torch.zeros(n, b, device=device)

    @property
    def data(self) -> ContactSensorData:
        self._update_outdated_buffers()
        return self._data

    @property
    def num_bodies(self) -> int: return self._num_bodies
    @property
    def body_names(self) -> list[str]: return self._body_names
    @property
    def contact_physx_view(self): return self._contact_physx_view

    def find_bodies(self, name_keys, preserve_order: bool = False):
        return string_utils.resolve_matching_names(name_keys, self.body_names, preserve_order)

    def _args(self):
        d = self._data
        hist = d.net_forces_w_history if self.cfg.history_length > 0 else None
        return [L.ptr(d.net_forces_w), L.ptr(hist), L.ptr(d.last_air_time), L.ptr(d.current_air_time),
                L.ptr(d.last_contact_time), L.ptr(d.current_contact_time), L.ptr(self._timestamp),
                L.ptr(self._timestamp_last_update), L.ptr(self._is_outdated)]

    def reset(self, env_ids: Sequence[int] | None = None):
        mask = _mask_from_ids(env_ids, self._num_envs, self._device)
        L.check(L.lib.mdp_contact_sensor_reset(self._num_envs, self._num_bodies, self.cfg.history_length,
                                               *self._args(), L.ptr(mask) if mask is not None else None, L.stream()))

    def _launch(self, dt: float, eager: bool):
        forces = self._contact_physx_view.get_net_contact_forces(dt=self._sim_physics_dt)
        L.check(L.lib.mdp_contact_sensor_update(self._num_envs, self._num_bodies, self.cfg.history_length,
                                                L.ptr(forces), *self._args(), float(dt),
                                                float(self.cfg.update_period), float(self.cfg.force_threshold),
                                                1 if eager else 0, L.stream()))

    def fill_scene_update(self, d: "L.MdpSceneUpdate", force_recompute: bool = False) -> bool:
        """This sensor's share of the fused post-physics launch (``InteractiveScene.update``)."""
        if self._num_bodies > 256 or self._num_bodies == 0:
            return False
        dd = self._data
        d.num_contact_bodies, d.history_len = self._num_bodies, self.cfg.history_length
        d.net_contact_forces = self._contact_physx_view.get_net_contact_forces(dt=self._sim_physics_dt).data_ptr()
        d.net_forces_w = dd.net_forces_w.data_ptr()
        d.net_forces_w_history = dd.net_forces_w_history.data_ptr() if self.cfg.history_length > 0 else None
        if self.cfg.track_air_time:
            d.last_air_time, d.current_air_time = dd.last_air_time.data_ptr(), dd.current_air_time.data_ptr()
            d.last_contact_time = dd.last_contact_time.data_ptr()
            d.current_contact_time = dd.current_contact_time.data_ptr()
        d.contact_timestamp = self._timestamp.data_ptr()
        d.contact_timestamp_last_update = self._timestamp_last_update.data_ptr()
        d.contact_is_outdated = self._is_outdated.data_ptr()
        d.contact_update_period, d.force_threshold = float(self.cfg.update_period), float(self.cfg.force_threshold)
        d.contact_eager = 1 if (force_recompute or self._is_visualizing or self.cfg.history_length > 0) else 0
        return True

    def update(self, dt: float, force_recompute: bool = False):
        """sensor_base.py:197-205 (timestamps, outdated flags) fused with the buffer update."""
        eager = force_recompute or self._is_visualizing or (self.cfg.history_length > 0)
        self._launch(dt, eager)

    def _update_outdated_buffers(self):
        if self.cfg.history_length > 0:
            return  # eager sensors are never left outdated by update()
        self._launch(0.0, True)

    def compute_first_contact(self, dt: float, abs_tol: float = 1.0e-8) -> torch.Tensor:
        """contact_sensor.py:169-203 (exposed through the fused ``feet_air_time`` term; stand-alone use
        evaluates the same expression as a termination-style mask kernel)."""
        if not self.cfg.track_air_time:
            raise RuntimeError(
                "The contact sensor is not configured to track contact time."
                "Please enable the 'track_air_time' in the sensor configuration."
            )
        return self._first(self.data.current_contact_time, dt, abs_tol)

    def compute_first_air(self, dt: float, abs_tol: float = 1.0e-8) -> torch.Tensor:
        if not self.cfg.track_air_time:
            raise RuntimeError(
                "The contact sensor is not configured to track contact time."
                "Please enable the 'track_air_time' in the sensor configuration."
            )
        return self._first(self.data.current_air_time, dt, abs_tol)

    def _first(self, timer: torch.Tensor, dt: float, abs_tol: float) -> torch.Tensor:
        from ..._terms import make_term, run_obs_single
        import numpy as np

        # (t > 0) * (t < dt + tol): clip to {0,1} via the observation post-processing:  obs kernel COPY then
        # comparison is done with two clamps is not exact -> use the dedicated termination op per column
        from ..._terms import TermSpec, TERMINATION, evaluate

        n, b = timer.shape
        flat = timer.reshape(n * b, 1)
        # manual-limit op: "out of [lo, hi]" == not first contact; first = (t > 0) & (t < thr)
        thr = float(np.float32(dt + abs_tol))
        lo = float(np.nextafter(np.float32(0.0), np.float32(1.0)))          # t >= lo  <=> t > 0
        hi = float(np.nextafter(np.float32(thr), np.float32(-np.inf)))      # t <= hi  <=> t < thr
        spec = TermSpec(TERMINATION, "MDP_TERM_JOINT_POS_OUT_OF_MANUAL_LIMIT", p={0: flat}, i={0: 1, 1: 1},
                        f={0: lo, 1: hi})
        outside = evaluate(spec, n * b, timer.device)
        return (~outside).view(n, b)


@configclass
class ContactSensorCfg(SensorBaseCfg):
    class_type: type = ContactSensor
    track_pose: bool = False
    track_air_time: bool = False
    force_threshold: float = 1.0
    filter_prim_paths_expr: list = list()
