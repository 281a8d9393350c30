"""``RayCasterCamera`` (``LAB/sensors/ray_caster/ray_caster_camera.py:26-430``) against ONE static mesh.

Same surface as the reference class: ``data`` (``CameraData``: ``pos_w``, ``quat_w_world``, ``intrinsic_matrices``,
``image_shape``, ``output[...]``), ``image_shape``, ``frame``, ``set_intrinsic_matrices``, ``set_world_poses`` (``opengl`` /
``ros`` / ``world`` conventions), ``set_world_poses_from_view``, ``reset``, ``update``.  The per-step work of
``_update_buffers_impl`` (:257-318) -- camera pose from the view pose and the offset, rays of the pinhole pattern rotated by
the full camera orientation, the cast, ``distance_to_image_plane`` / ``distance_to_camera`` with the three
``depth_clipping_behavior``s, normals, frame counter and the lazy-sensor bookkeeping -- is ONE C entry point,
``mdp_raycaster_camera_update``, gated per env by the device-side outdated flags: no ``nonzero()``, no host sync.
Configuration calls (offset convention, ``set_world_poses*``) are host-side (``utils/camera_math.py``)."""
from __future__ import annotations

import ctypes as C
from collections.abc import Sequence
from dataclasses import MISSING, dataclass, field

import numpy as np
import torch

from ... import _lib as L
from ...utils import camera_math as cm
from ...utils.configclass import configclass
from ...utils.warp import Mesh
from ..sensor_base import SensorBase
from .patterns import PinholeCameraPatternCfg
from .ray_caster import RayCasterCfg


@dataclass
class CameraData:
    """``LAB/sensors/camera/camera_data.py:17-90`` (the members the ray-caster camera fills)."""

    pos_w: torch.Tensor = None
    quat_w_world: torch.Tensor = None
    image_shape: tuple = None
    intrinsic_matrices: torch.Tensor = None
    output: dict = field(default_factory=dict)
    info: list = field(default_factory=list)

    @property
    def quat_w_ros(self) -> torch.Tensor:
        q = cm.convert_camera_frame_orientation_convention(self.quat_w_world.cpu().numpy(), origin="world", target="ros")
        return torch.from_numpy(q).to(self.quat_w_world.device)

    @property
    def quat_w_opengl(self) -> torch.Tensor:
        q = cm.convert_camera_frame_orientation_convention(self.quat_w_world.cpu().numpy(), origin="world", target="opengl")
        return torch.from_numpy(q).to(self.quat_w_world.device)


_CLIP = {"none": 0, "max": 1, "zero": 2}


class RayCasterCamera(SensorBase):
    UNSUPPORTED_TYPES = {
        "rgb", "instance_id_segmentation", "instance_id_segmentation_fast", "instance_segmentation",
        "instance_segmentation_fast", "semantic_segmentation", "skeleton_data", "motion_vectors", "bounding_box_2d_tight",
        "bounding_box_2d_tight_fast", "bounding_box_2d_loose", "bounding_box_2d_loose_fast", "bounding_box_3d",
        "bounding_box_3d_fast",
    }

    def __init__(self, cfg: "RayCasterCameraCfg", view, mesh: Mesh, num_envs: int, device: str, sim_physics_dt: float,
                 stage_up_axis: str = "Z"):
        """``view`` exposes ``get_root_transforms()`` (articulation) or ``get_transforms()`` (rigid body) -> (N, 7)
        [pos, quat xyzw] (ray_caster_camera.py:394-413); ``stage_up_axis`` is what ``get_stage_up_axis()`` reports."""
        self._check_supported_data_types(cfg)
        super().__init__(cfg, num_envs, device, sim_physics_dt)
        if cfg.depth_clipping_behavior not in _CLIP:
            raise ValueError(f"Unknown depth_clipping_behavior: {cfg.depth_clipping_behavior}.")
        self._view = view
        self._mesh = mesh
        self.meshes = {cfg.mesh_prim_paths[0] if cfg.mesh_prim_paths else "mesh": mesh}
        self._stage_up_axis = stage_up_axis
        self._data = CameraData()
        # ---- _initialize_rays_impl (:234-255)
        n = num_envs
        self._ALL_INDICES = torch.arange(n, device=device, dtype=torch.long)
        self._frame = torch.zeros(n, device=device, dtype=torch.long)
        self._create_buffers()
        self._compute_intrinsic_matrices()
        self.ray_starts, self.ray_directions = cfg.pattern_cfg.func(cfg.pattern_cfg, self._data.intrinsic_matrices, device)
        self.num_rays = self.ray_directions.shape[1]
        self.ray_hits_w = torch.zeros(n, self.num_rays, 3, device=device)
        quat_w = cm.convert_camera_frame_orientation_convention(np.array([cfg.offset.rot], np.float32),
                                                                 origin=cfg.offset.convention, target="world")
        self._offset_quat = torch.from_numpy(quat_w).to(device).repeat(n, 1).contiguous()
        self._offset_pos = torch.tensor(list(cfg.offset.pos), dtype=torch.float32, device=device).repeat(n, 1).contiguous()
        # outputs of the entry point, (N, R[, 3]); data.output[...] are (N, H, W, C) views of the same storage
        self._store = {}
        for name in cfg.data_types:
            c = 3 if name == "normals" else 1
            self._store[name] = torch.zeros(n, self.num_rays, c, device=device)
            self._data.output[name] = self._store[name].view(n, *self.image_shape, c)

    def __str__(self) -> str:
        return (
            f"Ray-Caster-Camera @ '{self.cfg.prim_path}': \n"
            f"\tview type            : {self._view.__class__}\n"
            f"\tupdate period (s)    : {self.cfg.update_period}\n"
            f"\tnumber of meshes     : {len(self.meshes)}\n"
            f"\tnumber of sensors    : {self._num_envs}\n"
            f"\tnumber of rays/sensor: {self.num_rays}\n"
            f"\ttotal number of rays : {self.num_rays * self._num_envs}\n"
            f"\timage shape          : {self.image_shape}"
        )

    # ---- properties
    @property
    def data(self) -> CameraData:
        self._update_outdated_buffers()
        return self._data

    @property
    def image_shape(self) -> tuple[int, int]:
        return (self.cfg.pattern_cfg.height, self.cfg.pattern_cfg.width)

    @property
    def frame(self) -> torch.Tensor:
        return self._frame

    # ---- operations
    def set_intrinsic_matrices(self, matrices: torch.Tensor, focal_length: float = 1.0, env_ids: Sequence[int] | None = None):
        """:121-140: new intrinsics for ``env_ids`` and the rays that go with them."""
        if env_ids is None:
            env_ids = slice(None)
        self._data.intrinsic_matrices[env_ids] = matrices.to(self._device)
        self._focal_length = focal_length
        starts, dirs = self.cfg.pattern_cfg.func(self.cfg.pattern_cfg, self._data.intrinsic_matrices[env_ids], self._device)
        self.ray_starts[env_ids], self.ray_directions[env_ids] = starts, dirs

    def reset(self, env_ids: Sequence[int] | None = None):
        super().reset(env_ids)
        if env_ids is None:
            env_ids = slice(None)
        pos_w, quat_w = self._compute_camera_world_poses(env_ids)
        self._data.pos_w[env_ids] = pos_w
        self._data.quat_w_world[env_ids] = quat_w
        self._frame[env_ids] = 0

    def set_world_poses(self, positions: torch.Tensor | None = None, orientations: torch.Tensor | None = None,
                        env_ids: Sequence[int] | None = None, convention: str = "ros"):
        """:156-206: the offsets that put the camera at the given world pose (a configuration call)."""
        from ...utils import math as math_utils

        if env_ids is None:
            env_ids = self._ALL_INDICES
        pos_w, quat_w = self._compute_view_world_poses(env_ids)
        if positions is not None:
            offset_world = (positions.to(self._device) - pos_w).contiguous()
            self._offset_pos[env_ids] = math_utils.quat_apply(math_utils.quat_inv(quat_w), offset_world)
        if orientations is not None:
            quat_w_set = cm.convert_camera_frame_orientation_convention(orientations.detach().cpu().numpy(), origin=convention,
                                                                         target="world")
            quat_w_set = torch.from_numpy(quat_w_set).to(self._device).contiguous()
            self._offset_quat[env_ids] = math_utils.quat_mul(math_utils.quat_inv(quat_w), quat_w_set)
        pos_w, quat_w = self._compute_camera_world_poses(env_ids)
        self._data.pos_w[env_ids] = pos_w
        self._data.quat_w_world[env_ids] = quat_w

    def set_world_poses_from_view(self, eyes: torch.Tensor, targets: torch.Tensor, env_ids: Sequence[int] | None = None):
        """:208-228."""
        from ...utils import math as math_utils

        rot = cm.create_rotation_matrix_from_view(eyes.detach().cpu().numpy(), targets.detach().cpu().numpy(), up_axis=self._stage_up_axis)
        orientations = math_utils.quat_from_matrix(torch.from_numpy(rot).to(self._device).contiguous())
        self.set_world_poses(eyes, orientations, env_ids, convention="opengl")

    def update(self, dt: float, force_recompute: bool = False):
        L.check(L.lib.mdp_sensor_tick(self._num_envs, L.ptr(self._timestamp), L.ptr(self._timestamp_last_update),
                                      L.ptr(self._is_outdated), float(dt), float(self.cfg.update_period), L.stream()))
        if force_recompute or self._is_visualizing or (self.cfg.history_length > 0):
            self._update_outdated_buffers()

    # ---- implementation
    def _update_outdated_buffers(self):
        """``_update_buffers_impl`` for the envs whose outdated flag is set (the flags stay on the device)."""
        pose = self._view_pose()
        a = L.MdpRaycasterCameraArgs()
        a.mesh = self._mesh.handle
        a.is_outdated, a.timestamp = self._is_outdated.data_ptr(), self._timestamp.data_ptr()
        a.timestamp_last_update = self._timestamp_last_update.data_ptr()
        a.view_pose, a.ld_pose, a.pose_xyzw = pose.data_ptr(), pose.stride(0), 1
        a.offset_pos, a.offset_quat, a.num_rays = self._offset_pos.data_ptr(), self._offset_quat.data_ptr(), self.num_rays
        a.depth_clipping, a.max_distance = _CLIP[self.cfg.depth_clipping_behavior], float(self.cfg.max_distance)
        a.ray_starts, a.ray_directions = self.ray_starts.data_ptr(), self.ray_directions.data_ptr()
        a.pos_w, a.quat_w_world = self._data.pos_w.data_ptr(), self._data.quat_w_world.data_ptr()
        a.ray_hits_w = self.ray_hits_w.data_ptr()
        for name, member in (("distance_to_image_plane", "distance_to_image_plane"), ("distance_to_camera", "distance_to_camera"),
                             ("normals", "normals")):
            setattr(a, member, self._store[name].data_ptr() if name in self._store else None)
        a.frame = self._frame.data_ptr()
        self._keep = (pose, a)
        L.check(L.lib.mdp_raycaster_camera_update(self._num_envs, C.byref(a), L.stream()))

    def _view_pose(self) -> torch.Tensor:
        """(N, 7) [pos, quat xyzw] of the prim the camera is attached to (:394-413)."""
        if hasattr(self._view, "get_root_transforms"):
            return self._view.get_root_transforms()
        if hasattr(self._view, "get_transforms"):
            return self._view.get_transforms()
        raise RuntimeError(f"Unsupported view type: {type(self._view)}")

    def _check_supported_data_types(self, cfg):
        common_elements = set(cfg.data_types) & RayCasterCamera.UNSUPPORTED_TYPES
        if common_elements:
            raise ValueError(
                f"RayCasterCamera class does not support the following sensor types: {common_elements}."
                "\n\tThis is because these sensor types cannot be obtained in a fast way using ''warp''."
                "\n\tHint: If you need to work with these sensor types, we recommend using the USD camera"
                " interface from the isaaclab.sensors.camera module."
            )

    def _create_buffers(self):
        n, dev = self._num_envs, self._device
        self.drift = torch.zeros(n, 3, device=dev)
        self._data.pos_w = torch.zeros((n, 3), device=dev)
        self._data.quat_w_world = torch.zeros((n, 4), device=dev)
        self._data.intrinsic_matrices = torch.zeros((n, 3, 3), device=dev)
        self._data.intrinsic_matrices[:, 2, 2] = 1.0
        self._data.image_shape = self.image_shape
        self._data.output = {}
        self._data.info = [{name: None for name in self.cfg.data_types}] * n
        for name in self.cfg.data_types:
            if name not in ("distance_to_image_plane", "distance_to_camera", "normals"):
                raise ValueError(f"Received unknown data type: {name}. Please check the configuration.")

    def _compute_intrinsic_matrices(self):
        """:370-392."""
        p = self.cfg.pattern_cfg
        if p.vertical_aperture is None:
            p.vertical_aperture = p.horizontal_aperture * p.height / p.width
        f_x = p.width * p.focal_length / p.horizontal_aperture
        f_y = p.height * p.focal_length / p.vertical_aperture
        c_x = p.horizontal_aperture_offset * f_x + p.width / 2
        c_y = p.vertical_aperture_offset * f_y + p.height / 2
        k = self._data.intrinsic_matrices
        k[:, 0, 0], k[:, 0, 2], k[:, 1, 1], k[:, 1, 2] = f_x, c_x, f_y, c_y
        self._focal_length = p.focal_length

    def _compute_view_world_poses(self, env_ids) -> tuple[torch.Tensor, torch.Tensor]:
        from ...utils import math as math_utils

        pose = self._view_pose()[env_ids]
        pos_w, quat_w = pose[:, :3].clone(), math_utils.convert_quat(pose[:, 3:7].contiguous(), to="wxyz")
        return pos_w, quat_w.clone()

    def _compute_camera_world_poses(self, env_ids) -> tuple[torch.Tensor, torch.Tensor]:
        """:415-430 (configuration calls only; the per-step poses come out of the update kernel)."""
        from ...utils import math as math_utils

        pos_w, quat_w = self._compute_view_world_poses(env_ids)
        pos_w = pos_w + math_utils.quat_apply(quat_w, self._offset_pos[env_ids].contiguous())
        quat_w = math_utils.quat_mul(quat_w, self._offset_quat[env_ids].contiguous())
        return pos_w, quat_w


@configclass
class RayCasterCameraCfg(RayCasterCfg):
    """``ray_caster_camera_cfg.py:19-63``."""

    @configclass
    class OffsetCfg:
        pos: tuple = (0.0, 0.0, 0.0)
        rot: tuple = (1.0, 0.0, 0.0, 0.0)
        convention: str = "ros"

    class_type: type = RayCasterCamera
    offset: OffsetCfg = OffsetCfg()
    data_types: list = ["distance_to_image_plane"]
    depth_clipping_behavior: str = "none"
    pattern_cfg: PinholeCameraPatternCfg = MISSING

    def __post_init__(self):
        self.attach_yaw_only = False      # for cameras, this quantity should be False always
