"""``RayCaster`` (``LAB/sensors/ray_caster/ray_caster.py:41-289``) against ONE static mesh.

``update`` only ticks the timestamps (lazy sensor, sensor_base.py:197-205); the cast runs for the
outdated envs when ``.data`` is read (``mdp_raycaster_update``) or inside the fused observation
kernel (``MDP_OBS_HEIGHT_SCAN_CAST``).  Ray generation (yaw-only attach), the cast, and the
bookkeeping happen in one launch; no ``(N, R, 3)`` start/direction tensors are materialised.
"""
from __future__ import annotations

from collections.abc import Sequence
from dataclasses import MISSING, dataclass

import torch

from ... import _lib as L
from ...utils.configclass import configclass
from ...utils.warp import Mesh
from ..sensor_base import SensorBase, SensorBaseCfg
from .patterns import PatternBaseCfg


@dataclass
class RayCasterData:
    pos_w: torch.Tensor = None
    quat_w: torch.Tensor = None
    ray_hits_w: torch.Tensor = None


class RayCaster(SensorBase):
    def __init__(self, cfg: "RayCasterCfg", view, mesh: Mesh, num_envs: int, device: str, sim_physics_dt: float):
        """``view`` exposes ``get_root_transforms()`` -> (N,7) [pos, quat xyzw] (ray_caster.py:226)."""
        super().__init__(cfg, num_envs, device, sim_physics_dt)
        self._view = view
        self._data = RayCasterData()
        self.meshes = {cfg.mesh_prim_paths[0] if cfg.mesh_prim_paths else "mesh": mesh}
        self._mesh = mesh
        # rays (ray_caster.py:201-218): pattern + offset; stored once as (R,3), identical for every env
        starts, dirs = cfg.pattern_cfg.func(cfg.pattern_cfg, device)
        self.num_rays = len(dirs)
        offset_pos = torch.tensor(list(cfg.offset.pos), device=device)
        offset_quat = torch.tensor(list(cfg.offset.rot), device=device)
        if tuple(cfg.offset.rot) != (1.0, 0.0, 0.0, 0.0):
            from ...utils.math import quat_apply

            dirs = quat_apply(offset_quat.repeat(len(dirs), 1), dirs)
        starts = starts + offset_pos
        self._ray_starts_local = starts.contiguous()
        self._ray_dirs_local = dirs.contiguous()
        # height-scan patterns point straight down: lets mdp_raycaster_update take its vertical-ray kernel
        d_host = self._ray_dirs_local.cpu()
        self._pattern_vertical_down = bool(((d_host[:, 0] == 0) & (d_host[:, 1] == 0) & (d_host[:, 2] < 0)).all())
        # per-env views with the reference's shapes (expanded, no storage)
        self.ray_starts = self._ray_starts_local.unsqueeze(0).expand(num_envs, -1, -1)
        self.ray_directions = self._ray_dirs_local.unsqueeze(0).expand(num_envs, -1, -1)
        self.drift = torch.zeros(num_envs, 3, device=device)
        self._data.pos_w = torch.zeros(num_envs, 3, device=device)
        self._data.quat_w = torch.zeros(num_envs, 4, device=device)
        self._data.ray_hits_w = torch.zeros(num_envs, self.num_rays, 3, device=device)
        # pose source in wxyz for the kernels: refreshed view of the physics transforms
        self._pose_wxyz = torch.zeros(num_envs, 13, device=device)
        self._pose_source = None
        # the separate vertical-ray kernel + lean observation kernel beat the fused cast (measured: 61 vs 86 us)
        self.fuse_into_observations = False
        # the cast kernel may write the post-processed height_scan observation columns itself (one launch less on
        # the critical path, the hits are not read back)
        self.scan_in_cast = True

    def bind_pose(self, root_state_fn):
        """Use an (N,13) wxyz root-state provider (ArticulationData.root_state_w) instead of converting the
        physics transforms again."""
        self._pose_source = root_state_fn

    @property
    def can_fuse(self) -> bool:
        """Whether an ObservationManager may cast this sensor's rays inside its own kernel."""
        return self.fuse_into_observations

    @property
    def data(self) -> RayCasterData:
        self._update_outdated_buffers()
        return self._data

    def reset(self, env_ids: Sequence[int] | None = None):
        super().reset(env_ids)
        if env_ids is None:
            env_ids = slice(None)
        lo, hi = self.cfg.drift_range
        if lo != 0.0 or hi != 0.0:
            self.drift[env_ids] = self.drift[env_ids].uniform_(lo, hi)

    def fill_scene_tick(self, d: "L.MdpSceneUpdate", force_recompute: bool = False) -> bool:
        """Registers the timestamp / outdated bookkeeping of this lazily evaluated sensor with the fused
        post-physics launch (sensor_base.py:197-205); the cast itself stays lazy."""
        if force_recompute or self._is_visualizing or self.cfg.history_length > 0 or d.num_ticks >= 4:
            return False
        k = d.num_ticks
        d.tick_timestamp[k] = self._timestamp.data_ptr()
        d.tick_timestamp_last_update[k] = self._timestamp_last_update.data_ptr()
        d.tick_is_outdated[k] = self._is_outdated.data_ptr()
        d.tick_update_period[k] = float(self.cfg.update_period)
        d.num_ticks = k + 1
        return True

    def update(self, dt: float, force_recompute: bool = False):
        L.check(L.lib.mdp_sensor_tick(self._num_envs, L.ptr(self._timestamp), L.ptr(self._timestamp_last_update),
                                      L.ptr(self._is_outdated), float(dt), float(self.cfg.update_period), L.stream()))
        if force_recompute or self._is_visualizing or (self.cfg.history_length > 0):
            self._update_outdated_buffers()

    def _pose(self) -> torch.Tensor:
        if self._pose_source is not None:
            return self._pose_source()
        vel = torch.zeros((self._num_envs, 6), device=self._device) if not hasattr(self, "_zero_vel") else self._zero_vel
        self._zero_vel = vel
        L.check(L.lib.mdp_state_pack(self._num_envs, L.ptr(self._view.get_root_transforms()), L.ptr(vel),
                                     L.ptr(self._pose_wxyz), L.stream()))
        return self._pose_wxyz

    def binding_desc(self) -> L.MdpRaycasterDesc:
        """Descriptor consumed by ``mdp_program_create`` for the fused observation term."""
        pose = self._pose()
        d = L.MdpRaycasterDesc()
        d.mesh = self._mesh.handle
        d.num_rays = self.num_rays
        d.attach_yaw_only = 1 if self.cfg.attach_yaw_only else 0
        d.max_dist = float(self.cfg.max_distance)
        d.ld_pose = pose.stride(0)
        d.root_pose = pose.data_ptr()
        d.drift = self.drift.data_ptr()
        d.ray_starts_local = self._ray_starts_local.data_ptr()
        d.ray_dirs_local = self._ray_dirs_local.data_ptr()
        d.pos_w = self._data.pos_w.data_ptr()
        d.quat_w = self._data.quat_w.data_ptr()
        d.ray_hits_w = self._data.ray_hits_w.data_ptr()
        d.timestamp = self._timestamp.data_ptr()
        d.timestamp_last_update = self._timestamp_last_update.data_ptr()
        d.is_outdated = self._is_outdated.data_ptr()
        return d

    def refresh_pose(self):
        """Hook run before a fused program launch: makes sure the wxyz pose buffer is current."""
        self._pose()

    def _update_outdated_buffers(self, scan=None, tick_dt: float = 0.0, from_physics: bool = False):
        """Casts the outdated envs.  ``scan`` (``L.MdpHeightScanObs``) additionally makes the kernel write the
        post-processed ``height_scan`` observation columns of every env (``ObservationManager`` binds it).
        ``tick_dt`` > 0 folds this physics step's ``update(dt)`` bookkeeping into the launch and ``from_physics`` reads
        the sensor pose from the physics view's transforms in place (ray_caster.py:226-232) instead of a packed
        root-state row: together they make the cast independent of the launch that updates the rest of the scene."""
        import ctypes as C

        if from_physics:
            pose = self._view.get_root_transforms()
            pose_ptr, ld, xyzw = pose.data_ptr(), pose.stride(0), 1
        else:
            pose = self._pose()
            pose_ptr, ld, xyzw = pose.data_ptr(), pose.stride(0), 0
        L.check(L.lib.mdp_raycaster_update(
            self._mesh.handle, self._num_envs, self.num_rays, pose_ptr, ld, self.drift.data_ptr(),
            self._ray_starts_local.data_ptr(), self._ray_dirs_local.data_ptr(), 1 if self.cfg.attach_yaw_only else 0,
            1 if self._pattern_vertical_down else 0, float(self.cfg.max_distance),
            self._data.pos_w.data_ptr(), self._data.quat_w.data_ptr(),
            self._data.ray_hits_w.data_ptr(), self._timestamp.data_ptr(), self._timestamp_last_update.data_ptr(),
            self._is_outdated.data_ptr(), C.byref(scan) if scan is not None else None, xyzw, float(tick_dt),
            float(self.cfg.update_period), L.stream()))

    def fill_post_physics(self, scan, tick_dt: float) -> "L.MdpRaycasterUpdateArgs":
        """The arguments of ``_update_outdated_buffers(scan, tick_dt, from_physics=True)`` as the cast member of
        ``mdp_post_physics_step``."""
        import ctypes as C

        a = L.MdpRaycasterUpdateArgs()
        pose = self._view.get_root_transforms()
        a.mesh = self._mesh.handle
        a.num_rays, a.ld_pose, a.pose_xyzw = self.num_rays, pose.stride(0), 1
        a.attach_yaw_only = 1 if self.cfg.attach_yaw_only else 0
        a.pattern_vertical_down = 1 if self._pattern_vertical_down else 0
        a.root_pose = pose.data_ptr()
        a.drift = self.drift.data_ptr()
        a.ray_starts_local = self._ray_starts_local.data_ptr()
        a.ray_dirs_local = self._ray_dirs_local.data_ptr()
        a.max_dist, a.tick_dt, a.update_period = float(self.cfg.max_distance), float(tick_dt), float(self.cfg.update_period)
        a.pos_w, a.quat_w = self._data.pos_w.data_ptr(), self._data.quat_w.data_ptr()
        a.ray_hits_w = self._data.ray_hits_w.data_ptr()
        a.timestamp, a.timestamp_last_update = self._timestamp.data_ptr(), self._timestamp_last_update.data_ptr()
        a.is_outdated = self._is_outdated.data_ptr()
        a.scan = C.cast(C.pointer(scan), C.c_void_p) if scan is not None else None
        if getattr(self, "_pose_ws", None) is None:
            import torch

            self._pose_ws = torch.zeros((self._num_envs_ws(), 8), dtype=torch.float32, device=pose.device)
        a.pose_ws = self._pose_ws.data_ptr()
        self._post_keep = (pose, scan)
        return a

    def _num_envs_ws(self) -> int:
        return int(self._data.pos_w.shape[0])


@configclass
class RayCasterCfg(SensorBaseCfg):
    @configclass
    class OffsetCfg:
        pos: tuple = (0.0, 0.0, 0.0)
        rot: tuple = (1.0, 0.0, 0.0, 0.0)

    class_type: type = RayCaster
    mesh_prim_paths: list = MISSING
    offset: OffsetCfg = OffsetCfg()
    attach_yaw_only: bool = MISSING
    pattern_cfg: PatternBaseCfg = MISSING
    max_distance: float = 1e6
    drift_range: tuple = (0.0, 0.0)
