"""numpy restatement of the sensors, term library, managers and command generator of the hot path."""
from __future__ import annotations

import numpy as np

from . import math_np as M

F = np.float32


def _f(x):
    return np.asarray(x, dtype=np.float32)


# ---- state container -----------------------------------------------------------------------------
def state_pack(pose_xyzw, vel):  # articulation_data.py:264-279
    pose = _f(pose_xyzw)
    quat = np.roll(pose[..., 3:7], 1, axis=-1)
    return np.concatenate([pose[..., :3], quat, _f(vel)], axis=-1)


def link_state_from_com(state, com_pos_b):  # articulation_data.py:288-302
    state = _f(state).copy()
    off = M.quat_rotate(state[..., 3:7], -_f(com_pos_b))
    state[..., 7:10] += np.cross(state[..., 10:13], off)
    return state


def root_lin_vel_b(st): return M.quat_rotate_inverse(st[:, 3:7], st[:, 7:10])       # :503-509
def root_ang_vel_b(st): return M.quat_rotate_inverse(st[:, 3:7], st[:, 10:13])      # :511-518
def projected_gravity_b(st, g=(0, 0, -1)):                                           # :412-414
    return M.quat_rotate_inverse(st[:, 3:7], np.tile(_f(g), (st.shape[0], 1)))
def heading_w(st):                                                                   # :417-425
    fwd = M.quat_apply(st[:, 3:7], np.tile(_f([1, 0, 0]), (st.shape[0], 1)))
    return np.arctan2(fwd[:, 1], fwd[:, 0]).astype(np.float32)
def joint_acc(qd, qd_prev, elapsed): return ((_f(qd) - _f(qd_prev)) / F(elapsed)).astype(np.float32)  # :446-455


# ---- contact sensor --------------------------------------------------------------------------------
class ContactSensor:  # contact_sensor.py:136-373 + sensor_base.py:182-205
    def __init__(self, n, b, history, update_period, force_threshold=1.0, track_air_time=True):
        self.n, self.b, self.h = n, b, history
        self.period, self.thr = update_period, force_threshold
        self.net = np.zeros((n, b, 3), np.float32)
        self.hist = np.zeros((n, max(history, 1), b, 3), np.float32)
        self.last_air = np.zeros((n, b), np.float32)
        self.cur_air = np.zeros((n, b), np.float32)
        self.last_con = np.zeros((n, b), np.float32)
        self.cur_con = np.zeros((n, b), np.float32)
        self.ts = np.zeros(n, np.float32)
        self.ts_last = np.zeros(n, np.float32)
        self.outdated = np.ones(n, bool)

    def update(self, forces, dt):
        self.ts = (self.ts + F(dt)).astype(np.float32)
        self.outdated |= (self.ts - self.ts_last + F(1e-6)) >= F(self.period)
        ids = np.nonzero(self.outdated)[0]
        if ids.size == 0:
            return
        f = _f(forces).reshape(self.n, self.b, 3)
        self.net[ids] = f[ids]
        if self.h > 0:
            self.hist[ids, 1:] = self.hist[ids, :-1].copy()
            self.hist[ids, 0] = self.net[ids]
        else:
            self.hist[:, 0] = self.net
        elapsed = (self.ts[ids] - self.ts_last[ids])[:, None]
        is_contact = M.norm(self.net[ids]) > F(self.thr)
        first_contact = (self.cur_air[ids] > 0) & is_contact
        first_detached = (self.cur_con[ids] > 0) & ~is_contact
        self.last_air[ids] = np.where(first_contact, self.cur_air[ids] + elapsed, self.last_air[ids])
        self.cur_air[ids] = np.where(~is_contact, self.cur_air[ids] + elapsed, 0.0)
        self.last_con[ids] = np.where(first_detached, self.cur_con[ids] + elapsed, self.last_con[ids])
        self.cur_con[ids] = np.where(is_contact, self.cur_con[ids] + elapsed, 0.0)
        self.ts_last[ids] = self.ts[ids]
        self.outdated[ids] = False

    def compute_first_contact(self, dt, abs_tol=1.0e-8):  # :169-203
        return (self.cur_con > 0.0) & (self.cur_con < F(dt + abs_tol))


# ---- ray caster ------------------------------------------------------------------------------------
def grid_pattern(resolution, size, direction=(0.0, 0.0, -1.0)):  # patterns.py:16-58 (ordering "xy")
    x = np.arange(-size[0] / 2, size[0] / 2 + 1.0e-9, resolution, dtype=np.float64).astype(np.float32)
    y = np.arange(-size[1] / 2, size[1] / 2 + 1.0e-9, resolution, dtype=np.float64).astype(np.float32)
    gx, gy = np.meshgrid(x, y, indexing="xy")
    starts = np.zeros((gx.size, 3), np.float32)
    starts[:, 0], starts[:, 1] = gx.flatten(), gy.flatten()
    dirs = np.tile(_f(direction), (gx.size, 1))
    return starts, dirs


def raycaster_rays(pos_w, quat_w, starts_local, dirs_local, attach_yaw_only):  # ray_caster.py:242-252
    n, r = pos_w.shape[0], starts_local.shape[0]
    q = np.repeat(_f(quat_w)[:, None, :], r, axis=1).reshape(-1, 4)
    s = np.tile(starts_local[None], (n, 1, 1)).reshape(-1, 3)
    d = np.tile(dirs_local[None], (n, 1, 1)).reshape(-1, 3)
    if attach_yaw_only:
        sw = M.quat_apply_yaw(q, s).reshape(n, r, 3) + _f(pos_w)[:, None, :]
        dw = d.reshape(n, r, 3)
    else:
        sw = M.quat_apply(q, s).reshape(n, r, 3) + _f(pos_w)[:, None, :]
        dw = M.quat_apply(q, d).reshape(n, r, 3)
    return sw.astype(np.float32), dw.astype(np.float32)


# ---- term library ------------------------------------------------------------------------------------
def _sel(x, ids):
    return x if ids is None else x[:, ids]


def contact_hist_max(hist, body_ids):
    h = hist if body_ids is None else hist[:, :, body_ids]
    return np.max(M.norm(h), axis=1)


class Terms:
    """Every term takes the env-like dict ``E`` with numpy state (see velocity_step.py)."""

    # observations.py
    @staticmethod
    def base_lin_vel(E): return root_lin_vel_b(E["root_state"])
    @staticmethod
    def base_ang_vel(E): return root_ang_vel_b(E["root_state"])
    @staticmethod
    def projected_gravity(E): return projected_gravity_b(E["root_state"])
    @staticmethod
    def base_pos_z(E): return E["root_state"][:, 2:3]
    @staticmethod
    def root_pos_w(E): return E["root_state"][:, :3] - E["env_origins"]
    @staticmethod
    def root_quat_w(E, make_quat_unique=False):
        q = E["root_state"][:, 3:7]
        return M.quat_unique(q) if make_quat_unique else q
    @staticmethod
    def root_lin_vel_w(E): return E["root_state"][:, 7:10]
    @staticmethod
    def root_ang_vel_w(E): return E["root_state"][:, 10:13]
    @staticmethod
    def joint_pos(E, ids=None): return _sel(E["joint_pos"], ids)
    @staticmethod
    def joint_pos_rel(E, ids=None): return _sel(E["joint_pos"], ids) - _sel(E["default_joint_pos"], ids)
    @staticmethod
    def joint_pos_limit_normalized(E, ids=None):
        lim = E["soft_joint_pos_limits"]
        return M.scale_transform(_sel(E["joint_pos"], ids), _sel(lim[..., 0], ids), _sel(lim[..., 1], ids))
    @staticmethod
    def joint_vel(E, ids=None): return _sel(E["joint_vel"], ids)
    @staticmethod
    def joint_vel_rel(E, ids=None): return _sel(E["joint_vel"], ids) - _sel(E["default_joint_vel"], ids)
    @staticmethod
    def height_scan(E, offset=0.5):
        return (E["sensor_pos_w"][:, 2:3] - E["ray_hits_w"][..., 2] - F(offset)).astype(np.float32)
    @staticmethod
    def last_action(E): return E["action"]
    @staticmethod
    def generated_commands(E): return E["command"]

    # rewards.py
    @staticmethod
    def is_alive(E): return (~E["terminated"]).astype(np.float32)
    @staticmethod
    def is_terminated(E): return E["terminated"].astype(np.float32)
    @staticmethod
    def is_terminated_term(E, rows):
        s = np.zeros(E["terminated"].shape[0], np.float32)
        for r in rows:
            s += E["term_dones"][r]
        return (s * (~E["time_outs"])).astype(np.float32)
    @staticmethod
    def lin_vel_z_l2(E): return np.square(root_lin_vel_b(E["root_state"])[:, 2])
    @staticmethod
    def ang_vel_xy_l2(E): return np.sum(np.square(root_ang_vel_b(E["root_state"])[:, :2]), axis=1)
    @staticmethod
    def flat_orientation_l2(E): return np.sum(np.square(projected_gravity_b(E["root_state"])[:, :2]), axis=1)
    @staticmethod
    def base_height_l2(E, target_height, use_sensor=False):
        target = F(target_height)
        if use_sensor:
            target = target + np.mean(E["ray_hits_w"][..., 2], axis=1, dtype=np.float32)
        return np.square(E["root_state"][:, 2] - target).astype(np.float32)
    @staticmethod
    def body_lin_acc_l2(E, ids=None):
        a = E["body_acc_w"] if ids is None else E["body_acc_w"][:, ids]
        return np.sum(M.norm(a[..., 0:3]), axis=1)
    @staticmethod
    def joint_torques_l2(E, ids=None): return np.sum(np.square(_sel(E["applied_torque"], ids)), axis=1)
    @staticmethod
    def joint_vel_l1(E, ids=None): return np.sum(np.abs(_sel(E["joint_vel"], ids)), axis=1)
    @staticmethod
    def joint_vel_l2(E, ids=None): return np.sum(np.square(_sel(E["joint_vel"], ids)), axis=1)
    @staticmethod
    def joint_acc_l2(E, ids=None): return np.sum(np.square(_sel(E["joint_acc"], ids)), axis=1)
    @staticmethod
    def joint_deviation_l1(E, ids=None):
        return np.sum(np.abs(_sel(E["joint_pos"], ids) - _sel(E["default_joint_pos"], ids)), axis=1)
    @staticmethod
    def joint_pos_limits(E, ids=None):
        lim = E["soft_joint_pos_limits"]
        q = _sel(E["joint_pos"], ids)
        out = -np.clip(q - _sel(lim[..., 0], ids), None, 0.0)
        out = out + np.clip(q - _sel(lim[..., 1], ids), 0.0, None)
        return np.sum(out, axis=1)
    @staticmethod
    def joint_vel_limits(E, soft_ratio, ids=None):
        out = np.abs(_sel(E["joint_vel"], ids)) - _sel(E["soft_joint_vel_limits"], ids) * F(soft_ratio)
        return np.sum(np.clip(out, 0.0, 1.0), axis=1)
    @staticmethod
    def applied_torque_limits(E, ids=None):
        return np.sum(np.abs(_sel(E["applied_torque"], ids) - _sel(E["computed_torque"], ids)), axis=1)
    @staticmethod
    def action_rate_l2(E): return np.sum(np.square(E["action"] - E["prev_action"]), axis=1)
    @staticmethod
    def action_l2(E): return np.sum(np.square(E["action"]), axis=1)
    @staticmethod
    def undesired_contacts(E, threshold, ids=None):
        return np.sum(contact_hist_max(E["contact_hist"], ids) > F(threshold), axis=1).astype(np.float32)
    @staticmethod
    def contact_forces(E, threshold, ids=None):
        return np.sum(np.clip(contact_hist_max(E["contact_hist"], ids) - F(threshold), 0.0, None), axis=1)
    @staticmethod
    def track_lin_vel_xy_exp(E, std):
        err = np.sum(np.square(E["command"][:, :2] - root_lin_vel_b(E["root_state"])[:, :2]), axis=1)
        return np.exp(-err / F(std**2)).astype(np.float32)
    @staticmethod
    def track_ang_vel_z_exp(E, std):
        err = np.square(E["command"][:, 2] - root_ang_vel_b(E["root_state"])[:, 2])
        return np.exp(-err / F(std**2)).astype(np.float32)

    # velocity/mdp/rewards.py
    @staticmethod
    def feet_air_time(E, threshold, ids=None):
        first = ((E["cur_contact_time"] > 0.0) & (E["cur_contact_time"] < F(E["step_dt"] + 1.0e-8)))
        first = _sel(first, ids)
        last_air = _sel(E["last_air_time"], ids)
        r = np.sum((last_air - F(threshold)) * first, axis=1, dtype=np.float32)
        return (r * (M.norm(E["command"][:, :2]) > F(0.1))).astype(np.float32)
    @staticmethod
    def feet_air_time_positive_biped(E, threshold, ids=None):
        air, con = _sel(E["cur_air_time"], ids), _sel(E["cur_contact_time"], ids)
        in_contact = con > 0.0
        mode = np.where(in_contact, con, air)
        single = np.sum(in_contact.astype(np.int32), axis=1) == 1
        r = np.min(np.where(single[:, None], mode, 0.0), axis=1)
        r = np.minimum(r, F(threshold))
        return (r * (M.norm(E["command"][:, :2]) > F(0.1))).astype(np.float32)
    @staticmethod
    def feet_slide(E, sensor_ids, body_ids):
        contacts = contact_hist_max(E["contact_hist"], sensor_ids) > F(1.0)
        vel = E["body_state"][:, body_ids, 7:9]
        return np.sum(M.norm(vel) * contacts, axis=1, dtype=np.float32)
    @staticmethod
    def track_lin_vel_xy_yaw_frame_exp(E, std):
        st = E["root_state"]
        v = M.quat_rotate_inverse(M.yaw_quat(st[:, 3:7]), st[:, 7:10])
        err = np.sum(np.square(E["command"][:, :2] - v[:, :2]), axis=1)
        return np.exp(-err / F(std**2)).astype(np.float32)
    @staticmethod
    def track_ang_vel_z_world_exp(E, std):
        err = np.square(E["command"][:, 2] - E["root_state"][:, 12])
        return np.exp(-err / F(std**2)).astype(np.float32)

    # terminations.py
    @staticmethod
    def time_out(E): return E["episode_length_buf"] >= E["max_episode_length"]
    @staticmethod
    def command_resample(E, num_resamples=1):
        return (E["time_left"] <= F(E["step_dt"])) & (E["command_counter"] == num_resamples)
    @staticmethod
    def bad_orientation(E, limit_angle):
        return np.abs(np.arccos(-projected_gravity_b(E["root_state"])[:, 2])) > F(limit_angle)
    @staticmethod
    def root_height_below_minimum(E, minimum_height): return E["root_state"][:, 2] < F(minimum_height)
    @staticmethod
    def joint_pos_out_of_manual_limit(E, bounds, ids=None):
        q = _sel(E["joint_pos"], ids)
        return np.any(q > F(bounds[1]), axis=1) | np.any(q < F(bounds[0]), axis=1)
    @staticmethod
    def joint_vel_out_of_limit(E, ids=None):
        return np.any(np.abs(_sel(E["joint_vel"], ids)) > _sel(E["soft_joint_vel_limits"], ids), axis=1)
    @staticmethod
    def joint_vel_out_of_manual_limit(E, max_velocity, ids=None):
        return np.any(np.abs(_sel(E["joint_vel"], ids)) > F(max_velocity), axis=1)
    @staticmethod
    def joint_effort_out_of_limit(E, ids=None):
        a, b = _sel(E["computed_torque"], ids), _sel(E["applied_torque"], ids)
        return np.any(np.isclose(a, b, rtol=1e-5, atol=1e-8), axis=1)
    @staticmethod
    def illegal_contact(E, threshold, ids=None):
        return np.any(contact_hist_max(E["contact_hist"], ids) > F(threshold), axis=1)
    @staticmethod
    def terrain_out_of_bounds(E, map_width, map_height, distance_buffer=3.0):
        p = E["root_state"]
        return (np.abs(p[:, 0]) > F(0.5 * map_width - distance_buffer)) | (np.abs(p[:, 1]) > F(0.5 * map_height - distance_buffer))


# ---- managers ------------------------------------------------------------------------------------------
def obs_postprocess(obs, noise=None, clip=None, scale=None):
    """observation_manager.py:313-318; noise = (kind, op, a, b, sample) with an injected sample array."""
    obs = _f(obs).copy()
    if noise is not None:
        kind, op, a, b, u = noise
        if kind == "uniform":      # noise_model.py:45-70
            n = _f(u) * F(b - a)
            obs = {"add": obs + n + F(a), "scale": obs * (n + F(a)), "abs": n + F(a)}[op]
        elif kind == "gaussian":   # :73-98
            n = F(b) * _f(u)
            obs = {"add": obs + F(a) + n, "scale": obs * (F(a) + n), "abs": F(a) + n}[op]
        else:                      # :20-42
            obs = {"add": obs + F(a), "scale": obs * F(a), "abs": np.zeros_like(obs) + F(a)}[op]
    if clip is not None:
        obs = np.clip(obs, F(clip[0]), F(clip[1]))
    if scale is not None:
        obs = obs * _f(scale)
    return obs.astype(np.float32)


def reward_compute(values, weights, dt, episode_sums, step_reward):
    """reward_manager.py:128-157; values: list of per-term raw arrays (None when weight == 0)."""
    n = episode_sums.shape[1]
    reward = np.zeros(n, np.float32)
    for i, (v, w) in enumerate(zip(values, weights)):
        if w == 0.0:
            continue
        value = (_f(v) * F(w) * F(dt)).astype(np.float32)
        reward += value
        episode_sums[i] += value
        step_reward[:, i] = value / F(dt)
    return reward


def termination_compute(values, time_out_flags):
    """termination_manager.py:151-174"""
    n = values[0].shape[0]
    trunc, term = np.zeros(n, bool), np.zeros(n, bool)
    for v, to in zip(values, time_out_flags):
        if to:
            trunc |= v
        else:
            term |= v
    return trunc, term


class VelocityCommand:  # velocity_command.py:48-160 + command_manager.py:151-187
    def __init__(self, n, cfg, step_dt):
        self.cfg, self.step_dt = cfg, step_dt
        self.cmd = np.zeros((n, 3), np.float32)
        self.heading_target = np.zeros(n, np.float32)
        self.is_heading = np.zeros(n, bool)
        self.is_standing = np.zeros(n, bool)
        self.time_left = np.zeros(n, np.float32)
        self.counter = np.zeros(n, np.int64)
        self.m_xy = np.zeros(n, np.float32)
        self.m_yaw = np.zeros(n, np.float32)

    def _resample(self, ids, u):
        c = self.cfg
        uni = lambda col, lo, hi: (_f(u[ids, col]) * F(hi - lo) + F(lo)).astype(np.float32)
        self.time_left[ids] = uni(0, *c["resampling_time_range"])
        self.counter[ids] += 1
        self.cmd[ids, 0] = uni(1, *c["lin_vel_x"])
        self.cmd[ids, 1] = uni(2, *c["lin_vel_y"])
        self.cmd[ids, 2] = uni(3, *c["ang_vel_z"])
        if c["heading_command"]:
            self.heading_target[ids] = uni(4, *c["heading"])
            self.is_heading[ids] = _f(u[ids, 5]) <= F(c["rel_heading_envs"])
        self.is_standing[ids] = _f(u[ids, 6]) <= F(c["rel_standing_envs"])

    def compute(self, root_state, dt, u):
        c = self.cfg
        max_step = F(c["resampling_time_range"][1] / self.step_dt)
        self.m_xy += M.norm(self.cmd[:, :2] - root_lin_vel_b(root_state)[:, :2]) / max_step
        self.m_yaw += np.abs(self.cmd[:, 2] - root_ang_vel_b(root_state)[:, 2]) / max_step
        self.time_left = (self.time_left - F(dt)).astype(np.float32)
        ids = np.nonzero(self.time_left <= 0.0)[0]
        if ids.size:
            self._resample(ids, u)
        if c["heading_command"]:
            hid = np.nonzero(self.is_heading)[0]
            err = M.wrap_to_pi(self.heading_target[hid] - heading_w(root_state)[hid])
            self.cmd[hid, 2] = np.clip(F(c["heading_control_stiffness"]) * err, F(c["ang_vel_z"][0]), F(c["ang_vel_z"][1]))
        self.cmd[np.nonzero(self.is_standing)[0], :] = 0.0

    def reset(self, ids, u):
        self.m_xy[ids] = 0.0
        self.m_yaw[ids] = 0.0
        self.counter[ids] = 0
        self._resample(ids, u)
        # CommandTerm.reset does not call _update_command; the fused HIP reset path does (documented)
