"""Synthetic physics-state generator: seeded host-side frames for the replay feed.

Produces a (T, frame_size) ring for a :class:`SyntheticPhysicsView` plus per-step actions; distributions and
seeds follow the measurement protocol (root pose around the env origin, joint state around the defaults,
Bernoulli contact forces with temporal hold so the air-time logic is exercised)."""
from __future__ import annotations

import numpy as np
import torch

from ..assets.physics_view import SyntheticPhysicsView


def _quat_from_euler_xyz_xyzw(roll, pitch, yaw):
    cy, sy = np.cos(yaw * 0.5), np.sin(yaw * 0.5)
    cr, sr = np.cos(roll * 0.5), np.sin(roll * 0.5)
    cp, sp = np.cos(pitch * 0.5), np.sin(pitch * 0.5)
    w = cy * cr * cp + sy * sr * sp
    x = cy * sr * cp - sy * cr * sp
    y = cy * cr * sp + sy * sr * cp
    z = sy * cr * cp - cy * sr * sp
    q = np.stack([x, y, z, w], axis=-1)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def generate_frames(view: SyntheticPhysicsView, num_frames: int, seed: int, env_origins: np.ndarray,
                    default_joint_pos: np.ndarray, body_names: list[str], with_links: bool = False) -> torch.Tensor:
    """Returns a host tensor (T, frame_size) laid out like ``view.live``."""
    rng = np.random.RandomState(seed)
    n, j, b = view.count, view.num_joints, view.num_contact_bodies
    frames = torch.zeros((num_frames, view.frame_size), dtype=torch.float32)
    fv = lambda name: view.field_view(frames, name).numpy()
    root_t, root_v = fv("root_transforms"), fv("root_velocities")
    q, qd, forces = fv("dof_positions"), fv("dof_velocities"), fv("net_contact_forces")
    # contact probabilities by body role
    p_contact = np.full(b, 0.005)
    for i, name in enumerate(body_names[:b]):
        low = name.lower()
        if "foot" in low or "ankle" in low:
            p_contact[i] = 0.5
        elif "thigh" in low or "knee" in low:
            p_contact[i] = 0.02
    state = rng.uniform(size=(n, b)) < p_contact
    base_xy = rng.uniform(-0.5, 0.5, size=(n, 2))
    base_yaw = rng.uniform(-np.pi, np.pi, size=n)
    for t in range(num_frames):
        pos = np.zeros((n, 3))
        pos[:, :2] = env_origins[:, :2] + base_xy + rng.normal(0, 0.02, size=(n, 2))
        pos[:, 2] = env_origins[:, 2] + rng.uniform(0.45, 0.75, size=n)
        quat = _quat_from_euler_xyz_xyzw(rng.normal(0, 0.15, n), rng.normal(0, 0.15, n),
                                         base_yaw + rng.normal(0, 0.05, n))
        root_t[t, :, :3] = pos
        root_t[t, :, 3:] = quat
        root_v[t] = rng.normal(0, 0.5, size=(n, 6))
        q[t] = default_joint_pos + rng.normal(0, 0.2, size=(n, j))
        qd[t] = rng.normal(0, 2.0, size=(n, j))
        # Markov hold: each (env, body) keeps its contact state with prob 0.9, else re-draws
        redraw = rng.uniform(size=(n, b)) < 0.1
        state = np.where(redraw, rng.uniform(size=(n, b)) < p_contact, state)
        fz = np.abs(rng.normal(0, 60.0, size=(n, b))) * state
        f = np.zeros((n, b, 3))
        f[..., 2] = fz
        f[..., :2] = rng.normal(0, 2.0, size=(n, b, 2)) * state[..., None]
        forces[t] = f.reshape(n * b, 3)
        if with_links and "link_transforms" in view._shapes:
            lt, lv, la = fv("link_transforms"), fv("link_velocities"), fv("link_accelerations")
            nb = view.num_bodies
            lt[t, :, :, :3] = pos[:, None, :] + rng.normal(0, 0.3, size=(n, nb, 3))
            lt[t, :, :, 3:] = quat[:, None, :]
            lv[t] = rng.normal(0, 0.5, size=(n, nb, 6))
            la[t] = rng.normal(0, 2.0, size=(n, nb, 6))
    return frames


def generate_actions(num_steps: int, num_envs: int, action_dim: int, seed: int) -> torch.Tensor:
    rng = np.random.RandomState(seed + 7919)
    return torch.from_numpy(np.clip(rng.normal(0, 1, size=(num_steps, num_envs, action_dim)), -3, 3).astype(np.float32))
