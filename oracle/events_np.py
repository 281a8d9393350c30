"""numpy restatement of the event terms on the step path (TEST INFRASTRUCTURE: imported by tests/ and the oracle only).

Reference: LAB/envs/mdp/events.py:610-866 (reset_root_state_uniform, reset_joints_by_scale / _by_offset,
push_by_setting_velocity), LAB/managers/event_manager.py:199-225 (interval timers), the write_*_to_sim methods of
LAB/assets/articulation/articulation.py:316-341, 400-420, 472-508 (which data buffers the writes land in) and
sample_uniform, LAB/utils/math.py:1331.  Random numbers are passed in as uniforms in [0, 1): row e of ``u`` holds the
numbers env e consumes, in the order the reference draws them."""
from __future__ import annotations

import numpy as np

from . import math_np as M

F = np.float32
KEYS = ("x", "y", "z", "roll", "pitch", "yaw")


def dict_range(rng: dict):
    """The reference builds a float32 tensor of (lo, hi) rows: scale = hi - lo in fp32."""
    r = np.array([rng.get(k, (0.0, 0.0)) for k in KEYS], dtype=np.float32)
    return r[:, 0].copy(), (r[:, 1] - r[:, 0]).astype(np.float32)


def tuple_range(rng):
    """Python floats in the reference: (hi - lo) in double, rounded when it meets the fp32 tensor."""
    return F(rng[0]), F(rng[1] - rng[0])


def sample(u, lo, scale):
    return (u.astype(np.float32) * scale + lo).astype(np.float32)


def reset_root_state_uniform(default_root_state, env_origins, ids, pose_range, velocity_range, u):
    """events.py:638-680 -> (root_pose (k,7) wxyz, root_velocity (k,6)) for the envs ``ids``; u: (N,12)."""
    d = default_root_state[ids].astype(np.float32)
    plo, psc = dict_range(pose_range)
    vlo, vsc = dict_range(velocity_range)
    s = sample(u[ids, 0:6], plo, psc)
    pos = ((d[:, 0:3] + env_origins[ids].astype(np.float32)) + s[:, 0:3]).astype(np.float32)
    delta = M.quat_from_euler_xyz(s[:, 3], s[:, 4], s[:, 5])
    quat = M.quat_mul(d[:, 3:7], delta)
    vel = (d[:, 7:13] + sample(u[ids, 6:12], vlo, vsc)).astype(np.float32)
    return np.concatenate([pos, quat], axis=1).astype(np.float32), vel


def reset_joints(default_q, default_qd, soft_pos_limits, soft_vel_limits, ids, position_range, velocity_range, u,
                 by_offset=False):
    """events.py:802-866 -> (joint_pos (k,J), joint_vel (k,J)); u: (N,2J), J position then J velocity uniforms."""
    j = default_q.shape[1]
    plo, psc = tuple_range(position_range)
    vlo, vsc = tuple_range(velocity_range)
    sp, sv = sample(u[ids, :j], plo, psc), sample(u[ids, j:], vlo, vsc)
    q, qd = default_q[ids].astype(np.float32), default_qd[ids].astype(np.float32)
    q = (q + sp) if by_offset else (q * sp)
    qd = (qd + sv) if by_offset else (qd * sv)
    lim = soft_pos_limits[ids]
    q = np.minimum(np.maximum(q, lim[..., 0]), lim[..., 1])           # Tensor.clamp_(min, max)
    vl = soft_vel_limits[ids]
    qd = np.minimum(np.maximum(qd, -vl), vl)
    return q.astype(np.float32), qd.astype(np.float32)


def push_by_setting_velocity(root_vel_w, ids, velocity_range, u):
    """events.py:610-635 -> root velocity (k,6) of the envs ``ids``; u: (N,6)."""
    lo, sc = dict_range(velocity_range)
    return (root_vel_w[ids].astype(np.float32) + sample(u[ids], lo, sc)).astype(np.float32)


def interval_tick(time_left, dt, interval_range_s, u):
    """event_manager.py:203-225 (per-env timers), in place -> ids whose timer fired; u: (N,)."""
    time_left -= F(dt)
    ids = np.nonzero(time_left < F(1e-6))[0]
    if ids.size:
        lo, sc = tuple_range(interval_range_s)
        time_left[ids] = sample(u[ids], lo, sc)
    return ids


def terrain_levels_vel(root_state, env_origins, command, ids, terrain_levels, terrain_types, terrain_origins,
                       tile_size_x, max_episode_length_s, random_levels):
    """curriculums.py:26-55 + terrain_importer.py:307-324, in place on ``terrain_levels`` / ``env_origins`` -> mean
    level over all envs.  ``random_levels``: (N,) the torch.randint_like draws for envs beyond the last level."""
    d = root_state[ids, :2].astype(np.float32) - env_origins[ids, :2].astype(np.float32)
    distance = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)
    move_up = distance > F(tile_size_x / 2)
    c = command[ids, :2].astype(np.float32)
    cn = np.sqrt(c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]).astype(np.float32)
    move_down = distance < (cn * F(max_episode_length_s)).astype(np.float32) * F(0.5)
    move_down &= ~move_up
    lv = terrain_levels[ids] + move_up.astype(np.int64) - move_down.astype(np.int64)
    max_level = terrain_origins.shape[0]
    lv = np.where(lv >= max_level, random_levels[ids], np.clip(lv, 0, None))
    terrain_levels[ids] = lv
    env_origins[ids] = terrain_origins[lv, terrain_types[ids]]
    return np.float32(np.mean(terrain_levels.astype(np.float32), dtype=np.float32))
