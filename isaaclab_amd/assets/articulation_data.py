"""``ArticulationData`` (``LAB/assets/articulation/articulation_data.py:33-854``): lazily evaluated,
time-stamped state buffers over a physics view, with the derived quantities computed by HIP kernels.

Buffers keep their identity (same storage every step) so fused manager programs can bind to them.
"""
from __future__ import annotations

import torch

from .. import _lib as L
from ..utils.buffers import TimestampedBuffer


class ArticulationData:
    def __init__(self, root_physx_view, device: str, num_bodies: int | None = None, body_names=None, joint_names=None):
        self.device = device
        self._root_physx_view = root_physx_view
        self._sim_timestamp = 0.0
        n = root_physx_view.count
        j = root_physx_view.num_joints
        b = root_physx_view.num_bodies if num_bodies is None else num_bodies
        self._num_instances, self._num_joints, self._num_bodies = n, j, b
        self.body_names = body_names
        self.joint_names = joint_names
        self.fixed_tendon_names = None
        # constants (articulation_data.py:60-66)
        self.GRAVITY_VEC_W = torch.tensor((0.0, 0.0, -1.0), device=device).repeat(n, 1)
        self.FORWARD_VEC_B = torch.tensor((1.0, 0.0, 0.0), device=device).repeat(n, 1)
        self._g = (0.0, 0.0, -1.0)
        f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=device)
        self._root_state_w = TimestampedBuffer(f(n, 13))
        self._root_link_state_w = TimestampedBuffer(f(n, 13))
        self._body_state_w = TimestampedBuffer(f(n, b, 13))
        self._body_link_state_w = TimestampedBuffer(f(n, b, 13))
        self._body_acc_w = TimestampedBuffer(f(n, b, 6))
        # the container owns its joint buffers (stable addresses for the fused programs); they are refreshed from
        # the physics view's current buffers, which may be a different ring frame every step
        self._joint_pos = TimestampedBuffer(f(n, j))
        self._joint_vel = TimestampedBuffer(f(n, j))
        self._joint_acc = TimestampedBuffer(f(n, j))
        self._joint_acc.timestamp = 0.0
        self._previous_joint_vel = f(n, j)
        # defaults / limits / targets (articulation_data.py:101-257)
        self.default_root_state = f(n, 13)
        self.default_root_state[:, 3] = 1.0
        self.default_joint_pos = f(n, j)
        self.default_joint_vel = f(n, j)
        self.joint_pos_target = f(n, j)
        self.joint_vel_target = f(n, j)
        self.joint_effort_target = f(n, j)
        self.computed_torque = f(n, j)
        self.applied_torque = f(n, j)
        self.joint_stiffness = f(n, j)
        self.joint_damping = f(n, j)
        self.default_joint_stiffness = f(n, j)
        self.default_joint_damping = f(n, j)
        self.default_joint_armature = f(n, j)
        self.default_joint_friction = f(n, j)
        self.joint_limits = f(n, j, 2)
        self.default_joint_limits = f(n, j, 2)
        self.joint_velocity_limits = f(n, j)
        self.joint_effort_limits = f(n, j)
        self.soft_joint_pos_limits = f(n, j, 2)
        self.soft_joint_vel_limits = f(n, j)
        self.gear_ratio = torch.ones((n, j), dtype=torch.float32, device=device)
        self.com_pos_b = f(n, b, 3)
        self.com_quat_b = f(n, b, 4)
        self.com_quat_b[..., 0] = 1.0

    def update(self, dt: float):
        """articulation_data.py:77-82: advance the timestamp and trigger the joint_acc finite difference."""
        self._sim_timestamp += dt
        # the joint buffers are this container's own copies (stable addresses bound by the fused programs): refresh
        # them eagerly, the reference's ``joint_pos`` is a live view of the physics buffer
        self.joint_pos
        self.joint_acc

    # -- fused scene update (InteractiveScene.update -> mdp_scene_update) --------------------------------------
    def fill_scene_update(self, d: "L.MdpSceneUpdate", dt: float):
        """Describes this container's share of the fused post-physics launch: root_state_w assembly, joint state
        refresh and the joint_acc finite difference, all from the physics view's CURRENT buffers."""
        v = self._root_physx_view
        d.num_envs, d.num_joints = self._num_instances, self._num_joints
        d.root_transforms = v.get_root_transforms().data_ptr()
        d.root_velocities = v.get_root_velocities().data_ptr()
        d.root_state_w = self._root_state_w.data.data_ptr()
        if self._num_joints > 0:
            q, qd = v.get_dof_positions(), v.get_dof_velocities()
            d.joint_pos_src, d.joint_vel_src, d.ld_joint_src = q.data_ptr(), qd.data_ptr(), q.stride(0)
            d.joint_pos, d.joint_vel = self._joint_pos.data.data_ptr(), self._joint_vel.data.data_ptr()
            d.prev_joint_vel, d.joint_acc = self._previous_joint_vel.data_ptr(), self._joint_acc.data.data_ptr()
            d.acc_elapsed = (self._sim_timestamp + dt) - self._joint_acc.timestamp

    def commit_scene_update(self, dt: float):
        """Host-side bookkeeping after the fused launch: the buffers it wrote are fresh at the new timestamp."""
        self._sim_timestamp += dt
        for buf in (self._root_state_w, self._joint_pos, self._joint_vel, self._joint_acc):
            buf.timestamp = self._sim_timestamp

    # -- lazily refreshed buffers ---------------------------------------------------------------------
    def _pack(self, pose, vel, out):
        count = out.numel() // 13
        L.check(L.lib.mdp_state_pack(count, L.ptr(pose), L.ptr(vel), L.ptr(out), L.stream()))

    @property
    def root_state_w(self):
        if self._root_state_w.timestamp < self._sim_timestamp:
            self._pack(self._root_physx_view.get_root_transforms(), self._root_physx_view.get_root_velocities(),
                       self._root_state_w.data)
            self._root_state_w.timestamp = self._sim_timestamp
        return self._root_state_w.data

    @property
    def root_link_state_w(self):
        if self._root_link_state_w.timestamp < self._sim_timestamp:
            com = self.com_pos_b[:, 0, :].contiguous()     # named: stays allocated until the launch is queued
            L.check(L.lib.mdp_link_state_from_com(self._num_instances, L.ptr(self.root_state_w), L.ptr(com),
                                                  L.ptr(self._root_link_state_w.data), L.stream()))
            self._root_link_state_w.timestamp = self._sim_timestamp
        return self._root_link_state_w.data

    @property
    def body_state_w(self):
        if self._body_state_w.timestamp < self._sim_timestamp:
            self._pack(self._root_physx_view.get_link_transforms(), self._root_physx_view.get_link_velocities(),
                       self._body_state_w.data)
            self._body_state_w.timestamp = self._sim_timestamp
        return self._body_state_w.data

    def body_state_rows(self, body_ids: torch.Tensor):
        """``body_state_w`` with only the rows of ``body_ids`` (device int32) brought up to date when the buffer is stale
        (``mdp_state_pack_rows``): for kernels that read a few bodies.  The buffer's timestamp is left alone, so any
        other reader still gets the full assembly."""
        if self._body_state_w.timestamp < self._sim_timestamp:
            v = self._root_physx_view
            L.check(L.lib.mdp_state_pack_rows(self._num_instances, self._num_bodies, L.ptr(body_ids), int(body_ids.numel()),
                                              L.ptr(v.get_link_transforms()), L.ptr(v.get_link_velocities()),
                                              L.ptr(self._body_state_w.data), L.stream()))
        return self._body_state_w.data

    @property
    def body_link_state_w(self):
        if self._body_link_state_w.timestamp < self._sim_timestamp:
            com = self.com_pos_b.contiguous()
            L.check(L.lib.mdp_link_state_from_com(self._num_instances * self._num_bodies, L.ptr(self.body_state_w),
                                                  L.ptr(com), L.ptr(self._body_link_state_w.data), L.stream()))
            self._body_link_state_w.timestamp = self._sim_timestamp
        return self._body_link_state_w.data

    @property
    def body_acc_w(self):
        if self._body_acc_w.timestamp < self._sim_timestamp:
            self._body_acc_w.data.copy_(self._root_physx_view.get_link_accelerations())
            self._body_acc_w.timestamp = self._sim_timestamp
        pending = self.__dict__.pop("_body_acc_zero", None)
        if pending:
            for mask, stamp in pending:
                if stamp == self._sim_timestamp:       # (older entries: the buffer has been read again from the simulation)
                    self._body_acc_w.data.masked_fill_(mask.bool().view(-1, 1, 1), 0.0)
        return self._body_acc_w.data

    def defer_body_acc_zero(self, mask: torch.Tensor):
        """``body_acc_w[env_ids] = 0`` of ``write_root_velocity_to_sim`` (articulation.py:418) for a launch that wrote the
        root velocities of the envs flagged in ``mask`` (device, num_envs bytes, read when the buffer is next accessed)
        without touching this lazily refreshed buffer: the rows are zeroed at the next read of this physics step, which
        is all the reference's write can be seen by (the next step reads the accelerations from the simulation again)."""
        self.__dict__.setdefault("_body_acc_zero", []).append((mask, self._sim_timestamp))

    @property
    def joint_pos(self):
        if self._joint_pos.timestamp < self._sim_timestamp:
            self._joint_pos.data.copy_(self._root_physx_view.get_dof_positions())
            self._joint_pos.timestamp = self._sim_timestamp
        return self._joint_pos.data

    @property
    def joint_vel(self):
        if self._joint_vel.timestamp < self._sim_timestamp:
            self._joint_vel.data.copy_(self._root_physx_view.get_dof_velocities())
            self._joint_vel.timestamp = self._sim_timestamp
        return self._joint_vel.data

    @property
    def joint_acc(self):
        """articulation_data.py:446-455 (finite difference over the elapsed time, updates the previous velocity)."""
        if self._joint_acc.timestamp < self._sim_timestamp:
            time_elapsed = self._sim_timestamp - self._joint_acc.timestamp
            qd, ld = L.rows(self.joint_vel, "joint_vel")
            L.check(L.lib.mdp_joint_acc_update(self._num_instances, self._num_joints, qd.data_ptr(), ld,
                                               self._previous_joint_vel.data_ptr(), self._joint_acc.data.data_ptr(),
                                               float(time_elapsed), L.stream()))
            self._joint_acc.timestamp = self._sim_timestamp
        return self._joint_acc.data

    # -- derived (recomputed on every access like the reference, articulation_data.py:412-425,503-518) ----
    def _derive(self, which: str) -> torch.Tensor:
        n = self._num_instances
        st = self.root_state_w
        out = torch.empty((n, 3) if which != "heading" else (n,), dtype=torch.float32, device=st.device)
        args = {"lin": [out.data_ptr(), None, None, None], "ang": [None, out.data_ptr(), None, None],
                "grav": [None, None, out.data_ptr(), None], "heading": [None, None, None, out.data_ptr()]}[which]
        g = self._g
        L.check(L.lib.mdp_root_derive(n, st.data_ptr(), 13, g[0], g[1], g[2], *args, L.stream()))
        return out

    @property
    def projected_gravity_b(self): return self._derive("grav")
    @property
    def heading_w(self): return self._derive("heading")
    @property
    def root_lin_vel_b(self): return self._derive("lin")
    @property
    def root_ang_vel_b(self): return self._derive("ang")

    # -- slices ---------------------------------------------------------------------------------------
    @property
    def root_pos_w(self): return self.root_state_w[:, :3]
    @property
    def root_quat_w(self): return self.root_state_w[:, 3:7]
    @property
    def root_vel_w(self): return self.root_state_w[:, 7:13]
    @property
    def root_lin_vel_w(self): return self.root_state_w[:, 7:10]
    @property
    def root_ang_vel_w(self): return self.root_state_w[:, 10:13]
    @property
    def root_link_pos_w(self): return self.root_link_state_w[:, :3]
    @property
    def root_link_quat_w(self): return self.root_link_state_w[:, 3:7]
    @property
    def body_pos_w(self): return self.body_state_w[..., :3]
    @property
    def body_quat_w(self): return self.body_state_w[..., 3:7]
    @property
    def body_vel_w(self): return self.body_state_w[..., 7:13]
    @property
    def body_lin_vel_w(self): return self.body_state_w[..., 7:10]
    @property
    def body_ang_vel_w(self): return self.body_state_w[..., 10:13]
    @property
    def body_lin_acc_w(self): return self.body_acc_w[..., 0:3]
    @property
    def body_ang_acc_w(self): return self.body_acc_w[..., 3:6]
