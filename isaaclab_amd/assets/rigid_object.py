"""Minimal ``RigidObject`` (``LAB/assets/rigid_object``): root state of a free body (the Shadow-hand cube)."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..utils.buffers import TimestampedBuffer


class RigidObjectData:
    def __init__(self, root_physx_view, device: str):
        self.device = device
        self._root_physx_view = root_physx_view
        self._sim_timestamp = 0.0
        n = root_physx_view.count
        self._root_state_w = TimestampedBuffer(torch.zeros((n, 13), dtype=torch.float32, device=device))
        self.default_root_state = torch.zeros((n, 13), dtype=torch.float32, device=device)
        self.default_root_state[:, 3] = 1.0

    def update(self, dt: float):
        self._sim_timestamp += dt

    @property
    def root_state_w(self):
        if self._root_state_w.timestamp < self._sim_timestamp:
            L.check(L.lib.mdp_state_pack(self._root_physx_view.count, L.ptr(self._root_physx_view.get_transforms()),
                                         L.ptr(self._root_physx_view.get_root_velocities()),
                                         L.ptr(self._root_state_w.data), L.stream()))
            self._root_state_w.timestamp = self._sim_timestamp
        return self._root_state_w.data

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


class RigidObject:
    def __init__(self, root_physx_view, device: str, body_names=("object",)):
        self.root_physx_view = root_physx_view
        self.device = device
        self.body_names = list(body_names)
        self.num_bodies = len(self.body_names)
        self.num_instances = root_physx_view.count
        self.data = RigidObjectData(root_physx_view, device)

    def find_bodies(self, name_keys, preserve_order: bool = False):
        from ..utils import string as string_utils

        return string_utils.resolve_matching_names(name_keys, self.body_names, preserve_order)

    def update(self, dt: float):
        self.data.update(dt)

    def reset(self, env_ids=None):
        pass

    def write_data_to_sim(self):
        pass
