"""Synthetic stand-in for the PhysX tensor views the hot path reads.

PhysX / Isaac Sim are out of scope: this object exposes the getters that
``ArticulationData`` / ``ContactSensor`` / ``RayCaster`` call on ``omni.physics.tensors`` views
(articulation_data.py:273-275,340-342,406,432,441; contact_sensor.py:323; ray_caster.py:226) and is
fed either by direct writes or by replaying a ring of pre-generated state frames.

All tensors are views into ONE flat device buffer: ``live`` for direct writes, or -- in replay mode -- the current
frame of a ring resident in HBM.  "Stepping physics" (``advance()``) then only moves the cursor: the getters hand out
views of the next frame in place (zero copy), exactly as a physics engine hands out its own state buffers, and the
pointers baked into a captured hipGraph are those of the ring.
Quaternions follow the physics convention (x, y, z, w).
"""
from __future__ import annotations

import torch


class SyntheticPhysicsView:
    FIELDS = ("root_transforms", "root_velocities", "dof_positions", "dof_velocities", "link_transforms",
              "link_velocities", "link_accelerations", "net_contact_forces")

    def __init__(self, num_envs: int, num_joints: int, num_bodies: int, device, num_contact_bodies: int | None = None,
                 with_links: bool = True, with_joint_forces: bool = False):
        self.count = num_envs
        self.num_joints = num_joints
        self.num_bodies = num_bodies
        self.num_contact_bodies = num_bodies if num_contact_bodies is None else num_contact_bodies
        self.device = torch.device(device)
        n, j, b, bc = num_envs, num_joints, num_bodies, self.num_contact_bodies
        shapes = {
            "root_transforms": (n, 7), "root_velocities": (n, 6), "dof_positions": (n, j), "dof_velocities": (n, j),
            "net_contact_forces": (n * bc, 3),
        }
        if with_links:
            shapes.update({"link_transforms": (n, b, 7), "link_velocities": (n, b, 6), "link_accelerations": (n, b, 6)})
        if with_joint_forces:      # ArticulationView.get_link_incoming_joint_force() (inhand_manipulation_env.py:121)
            shapes["link_incoming_joint_forces"] = (n, b, 6)
        self._shapes = shapes
        self._slices = {}
        off = 0
        for name, shp in shapes.items():
            size = 1
            for s in shp:
                size *= s
            size_al = (size + 3) // 4 * 4  # keep every field 16-byte aligned
            self._slices[name] = (off, size)
            off += size_al
        self.frame_size = off
        self.live = torch.zeros(off, dtype=torch.float32, device=self.device)
        self._views = {name: self.live[o : o + s].view(shapes[name]) for name, (o, s) in self._slices.items()}
        if "root_transforms" in self._views:
            self._views["root_transforms"][:, 6] = 1.0
        if with_links:
            self._views["link_transforms"][..., 6] = 1.0
        self.coms = torch.zeros((n, b, 7), dtype=torch.float32, device=self.device)
        self.coms[..., 6] = 1.0
        self._frames: torch.Tensor | None = None
        self._cursor = 0
        self._current: int | None = None       # ring frame the getters expose (None = ``live``)
        self._frame_views: dict[int, dict] = {}

    # -- getters with the PhysX view names -----------------------------------------------------------
    def _field(self, name: str) -> torch.Tensor:
        if self._current is None:
            return self._views[name]
        views = self._frame_views.get(self._current)
        if views is None:
            frame = self._frames[self._current]
            views = {k: frame[o : o + s].view(self._shapes[k]) for k, (o, s) in self._slices.items()}
            self._frame_views[self._current] = views
        return views[name]

    def get_root_transforms(self): return self._field("root_transforms")
    def get_root_velocities(self): return self._field("root_velocities")
    def get_dof_positions(self): return self._field("dof_positions")
    def get_dof_velocities(self): return self._field("dof_velocities")
    def get_link_transforms(self): return self._field("link_transforms")
    def get_link_velocities(self): return self._field("link_velocities")
    def get_link_accelerations(self): return self._field("link_accelerations")
    def get_link_incoming_joint_force(self): return self._field("link_incoming_joint_forces")
    def get_coms(self): return self.coms
    def get_net_contact_forces(self, dt: float | None = None): return self._field("net_contact_forces")
    def get_transforms(self): return self._field("root_transforms")

    # -- replay ---------------------------------------------------------------------------------------
    def field_view(self, frames: torch.Tensor, name: str) -> torch.Tensor:
        """View of field ``name`` inside a (T, frame_size) frame tensor (host or device)."""
        o, s = self._slices[name]
        return frames[:, o : o + s].view((frames.shape[0],) + tuple(self._shapes[name]))

    def set_frames(self, frames: torch.Tensor):
        if frames.dim() != 2 or frames.shape[1] != self.frame_size:
            raise ValueError(f"frames must be (T, {self.frame_size}), got {tuple(frames.shape)}")
        self._frames = frames.to(self.device, dtype=torch.float32).contiguous()
        self._cursor = 0
        self._current = None                   # ``live`` stays current until the first advance()
        self._frame_views = {}

    @property
    def num_frames(self) -> int:
        return 0 if self._frames is None else self._frames.shape[0]

    def advance(self):
        """'Physics step' in replay mode: the next frame of the ring becomes the one the getters expose."""
        if self._frames is None:
            return
        self._current = self._cursor
        self._cursor = (self._cursor + 1) % self._frames.shape[0]
