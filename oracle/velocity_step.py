"""CPU oracle of one full MDP env-step of the velocity task (TEST INFRASTRUCTURE; cpu_baseline of bench.py).

Follows ``ManagerBasedRLEnv.step`` (``LAB/envs/manager_based_rl_env.py:152-241``): action processing ->
decimation x {actuator, joint_acc, contact sensor} -> terminations -> rewards -> commands -> observations,
with the term set of ``velocity_env_cfg.py`` (and the per-robot overrides) expressed as plain lists."""
from __future__ import annotations

import math

import numpy as np

from . import actuators_np as A
from . import math_np as M
from . import mdp_np as P
from .mdp_np import Terms as T

F = np.float32


class VelocityStepOracle:
    def __init__(self, spec: dict):
        """spec keys: num_envs, joint_names, body_names, default_joint_pos (J,), soft_limit_factor, joint_pos_limits,
        actuator {kind, kp, kd, effort_limit, velocity_limit, saturation, [lstm_weights, decoder]},
        rewards [(name, fn, weight)], terminations [(name, fn, time_out)], obs_terms [(name, fn, noise(a,b)|None, clip)],
        command cfg dict, sim_dt, decimation, episode_length_s, env_origins, contact {history, period, threshold},
        height_scan {starts, dirs, caster, update_period, offset_z} or None, action_scale."""
        self.s = s = spec
        n, j, b = s["num_envs"], len(s["joint_names"]), len(s["body_names"])
        self.n, self.j, self.b = n, j, b
        self.dt, self.dec = s["sim_dt"], s["decimation"]
        self.step_dt = self.dt * self.dec
        self.max_episode_length = math.ceil(s["episode_length_s"] / self.step_dt)
        self.default_q = np.tile(np.asarray(s["default_joint_pos"], np.float32), (n, 1))
        lo, hi = s["joint_pos_limits"]
        mean, rng = (lo + hi) / 2, hi - lo
        self.soft_limits = np.zeros((n, j, 2), np.float32)
        self.soft_limits[..., 0] = mean - 0.5 * rng * s["soft_limit_factor"]
        self.soft_limits[..., 1] = mean + 0.5 * rng * s["soft_limit_factor"]
        self.action = np.zeros((n, j), np.float32)
        self.prev_action = np.zeros((n, j), np.float32)
        self.target = self.default_q.copy()
        self.prev_qd = np.zeros((n, j), np.float32)
        self.joint_acc = np.zeros((n, j), np.float32)
        self.computed = np.zeros((n, j), np.float32)
        self.applied = np.zeros((n, j), np.float32)
        c = s["contact"]
        self.contact = P.ContactSensor(n, b, c["history"], c["period"], c["threshold"])
        self.command = P.VelocityCommand(n, s["command"], self.step_dt)
        self.episode_length = np.zeros(n, np.int64)
        nr = len(s["rewards"])
        self.episode_sums = np.zeros((nr, n), np.float32)
        self.step_reward = np.zeros((n, nr), np.float32)
        self.term_dones = np.zeros((len(s["terminations"]), n), bool)
        self.terminated = np.zeros(n, bool)
        self.time_outs = np.zeros(n, bool)
        hs = s.get("height_scan")
        if hs is not None:
            r = hs["starts"].shape[0]
            self.ray_hits = np.zeros((n, r, 3), np.float32)
            self.sensor_pos = np.zeros((n, 3), np.float32)
            self.rc_ts = np.zeros(n, np.float32)
            self.rc_last = np.zeros(n, np.float32)
            self.rc_outdated = np.ones(n, bool)
        self._body_state = None
        if s.get("curriculum") is not None:
            s["env_origins"] = np.array(s["env_origins"], np.float32, copy=True)    # the curriculum moves them
            self.terrain_levels = np.asarray(s["curriculum"]["terrain_levels0"], np.int64).copy()
            self.curriculum_state = None
        if s.get("events") and s["events"].get("push_robot") is not None:
            self.push_time_left = np.asarray(s["events"]["push_time_left0"], np.float32).copy()
        act = s["actuator"]
        if act["kind"] == "lstm":
            L, H = len(act["lstm_weights"]), act["lstm_weights"][0][1].shape[1]
            self.h = np.zeros((L, n * j, H), np.float32)
            self.c = np.zeros((L, n * j, H), np.float32)

    def env_dict(self, root_state, q, qd):
        E = dict(root_state=root_state, joint_pos=q, joint_vel=qd, joint_acc=self.joint_acc,
                 default_joint_pos=self.default_q, default_joint_vel=np.zeros_like(self.default_q),
                 soft_joint_pos_limits=self.soft_limits,
                 soft_joint_vel_limits=np.full((self.n, self.j), self.s["actuator"]["velocity_limit"], np.float32),
                 applied_torque=self.applied, computed_torque=self.computed, action=self.action,
                 prev_action=self.prev_action, command=self.command.cmd, contact_hist=self.contact.hist,
                 cur_contact_time=self.contact.cur_con, cur_air_time=self.contact.cur_air,
                 last_air_time=self.contact.last_air, step_dt=self.step_dt, terminated=self.terminated,
                 time_outs=self.time_outs, term_dones=self.term_dones, episode_length_buf=self.episode_length,
                 max_episode_length=self.max_episode_length, env_origins=self.s["env_origins"],
                 time_left=self.command.time_left, command_counter=self.command.counter)
        if self.s.get("height_scan") is not None:
            E["ray_hits_w"] = self.ray_hits
            E["sensor_pos_w"] = self.sensor_pos
        if self._body_state is not None:
            E["body_state"] = self._body_state
        return E

    def _actuator(self, q, qd):
        a = self.s["actuator"]
        zeros = np.zeros_like(q)
        if a["kind"] == "dc_motor":
            self.computed, self.applied = A.dc_motor(self.target, zeros, zeros, q, qd, a["kp"], a["kd"],
                                                     a["effort_limit"], a["velocity_limit"], a["saturation"])
        elif a["kind"] in ("ideal_pd", "implicit"):
            self.computed, self.applied = A.ideal_pd(self.target, zeros, zeros, q, qd, a["kp"], a["kd"], a["effort_limit"])
        elif a["kind"] == "lstm":
            lim = np.full_like(q, a["effort_limit"])
            vl = np.full_like(q, a["velocity_limit"])
            self.computed, self.applied, self.h, self.c = A.actuator_net_lstm(
                self.target, q, qd, self.h, self.c, a["lstm_weights"], a["decoder"], lim, vl, a["saturation"])
        else:
            raise ValueError(a["kind"])

    def _raycast(self, root_state):
        hs = self.s["height_scan"]
        ids = np.nonzero(self.rc_outdated)[0]
        if ids.size == 0:
            return
        pos = root_state[ids, :3]
        quat = root_state[ids, 3:7]
        sw, dw = P.raycaster_rays(pos, quat, hs["starts"], hs["dirs"], True)
        hits = hs["caster"].cast(sw.reshape(-1, 3), dw.reshape(-1, 3), 1e6, hs.get("threads", 0))[0]
        self.ray_hits[ids] = hits.reshape(ids.size, -1, 3)
        self.sensor_pos[ids] = pos
        self.rc_last[ids] = self.rc_ts[ids]
        self.rc_outdated[ids] = False

    def _apply_reset_events(self, ids, root, q, qd, eu):
        """EventManager.apply(mode="reset") of the velocity task (velocity_env_cfg.py:187-210): the writes land in the
        asset's data buffers (articulation.py:332, 417-418, 498-501) -- here ``root`` / ``q`` / ``qd`` of this step."""
        from . import events_np as EV

        ev = self.s["events"]
        if ev.get("reset_base") is not None:
            pose, vel = EV.reset_root_state_uniform(ev["default_root_state"], self.s["env_origins"], ids,
                                                    ev["reset_base"]["pose_range"], ev["reset_base"]["velocity_range"],
                                                    eu["reset_root_state_uniform"])
            root[ids, :7] = pose
            root[ids, 7:] = vel
        if ev.get("reset_robot_joints") is not None:
            c = ev["reset_robot_joints"]
            vl = np.full((self.n, self.j), self.s["actuator"]["velocity_limit"], np.float32)
            jq, jqd = EV.reset_joints(self.default_q, np.zeros_like(self.default_q), self.soft_limits, vl, ids,
                                      c["position_range"], c["velocity_range"], eu["reset_joints_by_scale"])
            q[ids], qd[ids] = jq, jqd
            self.prev_qd[ids] = jqd
            self.joint_acc[ids] = 0.0

    def reset_idx(self, ids, cmd_uniforms):
        """ManagerBasedRLEnv._reset_idx (manager_based_rl_env.py:346-391) without curriculum / events (physics side):
        scene.reset -> manager resets in the reference order -> episode counter.  Returns the ``extras["log"]`` dict."""
        s, log = self.s, {}
        # scene.reset: actuators (actuator_net.py:61-64), sensors (sensor_base.py:182-195, contact_sensor.py:136-155,
        # ray_caster.py:106-113 with drift_range (0, 0))
        if s["actuator"]["kind"] == "lstm":
            L, H = self.h.shape[0], self.h.shape[2]
            self.h.reshape(L, self.n, self.j, H)[:, ids] = 0.0
            self.c.reshape(L, self.n, self.j, H)[:, ids] = 0.0
        ct = self.contact
        ct.ts[ids] = 0.0; ct.ts_last[ids] = 0.0; ct.outdated[ids] = True
        ct.net[ids] = 0.0; ct.hist[ids] = 0.0
        for buf in (ct.cur_air, ct.last_air, ct.cur_con, ct.last_con):
            buf[ids] = 0.0
        if s.get("height_scan") is not None:
            self.rc_ts[ids] = 0.0; self.rc_last[ids] = 0.0; self.rc_outdated[ids] = True
        # action manager (action_manager.py reset): prev_action, action, raw actions
        self.prev_action[ids] = 0.0
        self.action[ids] = 0.0
        # reward manager (reward_manager.py:100-126)
        max_len_s = F(s["episode_length_s"])
        for i, (name, _, _) in enumerate(s["rewards"]):
            log["Episode_Reward/" + name] = np.float32(np.mean(self.episode_sums[i, ids], dtype=np.float32) / max_len_s)
            self.episode_sums[i, ids] = 0.0
        # command manager (command_manager.py:120-149)
        log["Metrics/base_velocity/error_vel_xy"] = np.float32(np.mean(self.command.m_xy[ids], dtype=np.float32))
        log["Metrics/base_velocity/error_vel_yaw"] = np.float32(np.mean(self.command.m_yaw[ids], dtype=np.float32))
        self.command.reset(ids, cmd_uniforms)
        # termination manager (termination_manager.py:127-150)
        for i, (name, _, _) in enumerate(s["terminations"]):
            log["Episode_Termination/" + name] = int(np.count_nonzero(self.term_dones[i, ids]))
        self.episode_length[ids] = 0
        return log

    def step(self, action, substep_frames, cmd_uniforms=None, obs_noise=None, auto_reset=False, event_uniforms=None):
        """substep_frames: list of `decimation` dicts {root_state (N,13 wxyz), joint_pos, joint_vel, forces (N,B,3)}.
        ``event_uniforms`` (with spec["events"]): {event function name: (N,k) uniforms, "interval": (N,)}."""
        s = self.s
        self.prev_action = self.action.copy()
        self.action = np.asarray(action, np.float32).copy()
        self.target = (self.action * F(s["action_scale"]) + self.default_q).astype(np.float32)
        for fr in substep_frames:  # write_data_to_sim uses the state of the PREVIOUS physics step
            q, qd = self._last_q, self._last_qd
            self._actuator(q, qd)
            # physics.advance(): the new frame becomes current; scene.update(dt)
            self._last_q, self._last_qd = fr["joint_pos"], fr["joint_vel"]
            self.joint_acc = P.joint_acc(fr["joint_vel"], self.prev_qd, self.dt)
            self.prev_qd = fr["joint_vel"].copy()
            self.contact.update(fr["forces"], self.dt)
            if s.get("height_scan") is not None:
                self.rc_ts = (self.rc_ts + F(self.dt)).astype(np.float32)
                self.rc_outdated |= (self.rc_ts - self.rc_last + F(1e-6)) >= F(s["height_scan"]["update_period"])
        fr = substep_frames[-1]
        events = s.get("events")
        # with events the reset / push terms overwrite rows of the data buffers: work on copies of the frame
        root = fr["root_state"].copy() if events else fr["root_state"]
        q_now = fr["joint_pos"].copy() if events else fr["joint_pos"]
        qd_now = fr["joint_vel"].copy() if events else fr["joint_vel"]
        self._body_state = fr.get("body_state")
        self.episode_length += 1
        E = self.env_dict(root, q_now, qd_now)
        # terminations
        vals = [fn(E) for _, fn, _ in s["terminations"]]
        for i, v in enumerate(vals):
            self.term_dones[i] = v
        self.time_outs, self.terminated = P.termination_compute(vals, [to for _, _, to in s["terminations"]])
        E["terminated"], E["time_outs"] = self.terminated, self.time_outs
        # rewards (terms reading ray hits trigger the lazy sensor update first)
        if s.get("height_scan") is not None and s.get("reward_needs_rays", False):
            self._raycast(root)
        values = [fn(E) if w != 0.0 else None for _, fn, w in s["rewards"]]
        reward = P.reward_compute(values, [w for _, _, w in s["rewards"]], self.step_dt, self.episode_sums,
                                  self.step_reward)
        # commands
        if cmd_uniforms is None:
            cmd_uniforms = np.zeros((self.n, 7), np.float32)
        # reset of the envs that are done (manager_based_rl_env.py:214-229), between the rewards and the commands
        self.log = None
        if auto_reset:
            ids = np.nonzero(self.terminated | self.time_outs)[0]
            if ids.size:
                if s.get("curriculum") is not None:     # curriculum_manager.compute(env_ids), manager_based_rl_env.py:353
                    from . import events_np as EV

                    c = s["curriculum"]
                    self.curriculum_state = EV.terrain_levels_vel(
                        root, s["env_origins"], self.command.cmd, ids, self.terrain_levels, c["terrain_types"],
                        c["terrain_origins"], c["tile_size_x"], s["episode_length_s"], event_uniforms["terrain_levels_vel"])
                if events:      # scene.reset -> reset events -> manager resets (manager_based_rl_env.py:355-359)
                    self._apply_reset_events(ids, root, q_now, qd_now, event_uniforms)
                self.log = self.reset_idx(ids, cmd_uniforms)
                if s.get("curriculum") is not None:
                    self.log["Curriculum/terrain_levels"] = self.curriculum_state      # curriculum_manager.py:88-117
                # scene.write_data_to_sim() of manager_based_rl_env.py:221: one more actuator evaluation of EVERY env
                # (it advances the LSTM state) in the steps that reset at least one
                self._actuator(q_now, qd_now)
        self.command.compute(root, self.step_dt, cmd_uniforms)
        E["command"] = self.command.cmd
        if events and events.get("push_robot") is not None:     # interval events (manager_based_rl_env.py:233-235)
            from . import events_np as EV

            c = events["push_robot"]
            fired = EV.interval_tick(self.push_time_left, self.step_dt, c["interval_range_s"], event_uniforms["interval"])
            if fired.size:
                root[fired, 7:] = EV.push_by_setting_velocity(root[:, 7:], fired, c["velocity_range"],
                                                              event_uniforms["push_by_setting_velocity"])
        if events:
            # the next write_data_to_sim reads the joint state the reset left in the data buffers
            self._last_q, self._last_qd = q_now, qd_now
            E["joint_acc"] = self.joint_acc
        # observations
        if s.get("height_scan") is not None:
            self._raycast(root)
            E["ray_hits_w"], E["sensor_pos_w"] = self.ray_hits, self.sensor_pos
        cols = []
        for k, (name, fn, noise, clip) in enumerate(s["obs_terms"]):
            v = fn(E)
            nz = None
            if noise is not None and obs_noise is not None:
                nz = ("uniform", "add", noise[0], noise[1], obs_noise[name])
            cols.append(P.obs_postprocess(v, nz, clip))
        obs = np.concatenate(cols, axis=1)
        return obs, reward, self.terminated.copy(), self.time_outs.copy()

    def prime(self, joint_pos, joint_vel):
        """State visible to the first write_data_to_sim (what the physics view holds before the first step)."""
        self._last_q, self._last_qd = np.asarray(joint_pos, np.float32), np.asarray(joint_vel, np.float32)


def anymal_spec(num_envs, env_origins, rough=True, caster=None, ray_starts=None, ray_dirs=None, actuator=None,
                flat_weights=False, threads=0):
    """Term tables of Isaac-Velocity-{Rough,Flat}-Anymal-D-v0 (velocity_env_cfg.py:116-265, flat_env_cfg.py:13-28)."""
    legs = ("LF", "LH", "RF", "RH")
    joint_names = [f"{l}_{j}" for j in ("HAA", "HFE", "KFE") for l in legs]
    body_names = ["base"] + [f"{l}_{b}" for b in ("HIP", "THIGH", "SHANK", "FOOT") for l in legs]
    dq = [0.0] * 4 + [0.4, -0.4, 0.4, -0.4] + [-0.8, 0.8, -0.8, 0.8]
    feet = [i for i, b in enumerate(body_names) if b.endswith("FOOT")]
    thighs = [i for i, b in enumerate(body_names) if b.endswith("THIGH")]
    std = math.sqrt(0.25)
    w_flat, w_tau, w_air = (-5.0, -2.5e-5, 0.5) if flat_weights else (0.0, -1.0e-5, 0.125)
    rewards = [
        ("track_lin_vel_xy_exp", lambda E: T.track_lin_vel_xy_exp(E, std), 1.0),
        ("track_ang_vel_z_exp", lambda E: T.track_ang_vel_z_exp(E, std), 0.5),
        ("lin_vel_z_l2", T.lin_vel_z_l2, -2.0),
        ("ang_vel_xy_l2", T.ang_vel_xy_l2, -0.05),
        ("dof_torques_l2", T.joint_torques_l2, w_tau),
        ("dof_acc_l2", T.joint_acc_l2, -2.5e-7),
        ("action_rate_l2", T.action_rate_l2, -0.01),
        ("feet_air_time", lambda E: T.feet_air_time(E, 0.5, feet), w_air),
        ("undesired_contacts", lambda E: T.undesired_contacts(E, 1.0, thighs), -1.0),
        ("flat_orientation_l2", T.flat_orientation_l2, w_flat),
        ("dof_pos_limits", T.joint_pos_limits, 0.0),
    ]
    terminations = [("time_out", T.time_out, True), ("base_contact", lambda E: T.illegal_contact(E, 1.0, [0]), False)]
    obs_terms = [
        ("base_lin_vel", T.base_lin_vel, (-0.1, 0.1), None), ("base_ang_vel", T.base_ang_vel, (-0.2, 0.2), None),
        ("projected_gravity", T.projected_gravity, (-0.05, 0.05), None),
        ("velocity_commands", T.generated_commands, None, None),
        ("joint_pos", T.joint_pos_rel, (-0.01, 0.01), None), ("joint_vel", T.joint_vel_rel, (-1.5, 1.5), None),
        ("actions", T.last_action, None, None),
    ]
    height = None
    if rough:
        obs_terms.append(("height_scan", T.height_scan, (-0.1, 0.1), (-1.0, 1.0)))
        height = dict(starts=ray_starts, dirs=ray_dirs, caster=caster, update_period=0.02, threads=threads)
    if actuator is None:
        actuator = dict(kind="dc_motor", kp=F(40.0), kd=F(5.0), effort_limit=F(80.0), velocity_limit=F(7.5), saturation=120.0)
    return dict(
        num_envs=num_envs, joint_names=joint_names, body_names=body_names, default_joint_pos=dq,
        soft_limit_factor=0.95, joint_pos_limits=(-2.4, 2.4), actuator=actuator, rewards=rewards,
        terminations=terminations, obs_terms=obs_terms, sim_dt=0.005, decimation=4, episode_length_s=20.0,
        env_origins=np.asarray(env_origins, np.float32), action_scale=0.5,
        contact=dict(history=3, period=0.005, threshold=1.0), height_scan=height,
        command=dict(resampling_time_range=(10.0, 10.0), lin_vel_x=(-1.0, 1.0), lin_vel_y=(-1.0, 1.0),
                     ang_vel_z=(-1.0, 1.0), heading=(-math.pi, math.pi), heading_command=True,
                     rel_heading_envs=1.0, rel_standing_envs=0.02, heading_control_stiffness=0.5),
    )


def _match(names, patterns):
    import re

    pats = [patterns] if isinstance(patterns, str) else list(patterns)
    return [i for i, n in enumerate(names) if any(re.fullmatch(p, n) for p in pats)]


def h1_spec(num_envs, env_origins, caster, ray_starts, ray_dirs, threads=0):
    """Term tables of Isaac-Velocity-Rough-H1-v0 (config/h1/rough_env_cfg.py:19-115 over velocity_env_cfg.py:116-265;
    robot constants ASSETS/robots/unitree.py:216-260: 19 joints, three implicit-actuator groups)."""
    joint_names = ["left_hip_yaw", "right_hip_yaw", "torso", "left_hip_roll", "right_hip_roll", "left_shoulder_pitch",
                   "right_shoulder_pitch", "left_hip_pitch", "right_hip_pitch", "left_shoulder_roll",
                   "right_shoulder_roll", "left_knee", "right_knee", "left_shoulder_yaw", "right_shoulder_yaw",
                   "left_ankle", "right_ankle", "left_elbow", "right_elbow"]
    body_names = ["pelvis"] + [n + "_link" for n in joint_names]
    j = len(joint_names)

    def per_joint(table, default=0.0):
        out = np.full(j, default, np.float32)
        for pat, val in table.items():
            out[_match(joint_names, pat)] = val
        return out

    dq = per_joint({".*_hip_yaw": 0.0, ".*_hip_roll": 0.0, ".*_hip_pitch": -0.28, ".*_knee": 0.79, ".*_ankle": -0.52,
                    "torso": 0.0, ".*_shoulder_pitch": 0.28, ".*_shoulder_roll": 0.0, ".*_shoulder_yaw": 0.0,
                    ".*_elbow": 0.52})
    kp = per_joint({".*_hip_yaw": 150.0, ".*_hip_roll": 150.0, ".*_hip_pitch": 200.0, ".*_knee": 200.0, "torso": 200.0,
                    ".*_ankle": 20.0, ".*_shoulder_.*": 40.0, ".*_elbow": 40.0})
    kd = per_joint({".*_hip_.*": 5.0, ".*_knee": 5.0, "torso": 5.0, ".*_ankle": 4.0, ".*_shoulder_.*": 10.0,
                    ".*_elbow": 10.0})
    lim = per_joint({".*": 300.0})
    lim[_match(joint_names, ".*_ankle")] = 100.0
    feet = _match(body_names, ".*ankle_link")
    torso = _match(body_names, ".*torso_link")
    ankles = _match(joint_names, ".*_ankle")
    hips = _match(joint_names, [".*_hip_yaw", ".*_hip_roll"])
    arms = _match(joint_names, [".*_shoulder_.*", ".*_elbow"])
    torso_j = _match(joint_names, "torso")
    std = math.sqrt(0.25)
    rewards = [
        ("track_lin_vel_xy_exp", lambda E: T.track_lin_vel_xy_yaw_frame_exp(E, 0.5), 1.0),
        ("track_ang_vel_z_exp", lambda E: T.track_ang_vel_z_world_exp(E, 0.5), 1.0),
        ("ang_vel_xy_l2", T.ang_vel_xy_l2, -0.05),
        ("dof_torques_l2", T.joint_torques_l2, 0.0),
        ("dof_acc_l2", T.joint_acc_l2, -1.25e-7),
        ("action_rate_l2", T.action_rate_l2, -0.005),
        ("feet_air_time", lambda E: T.feet_air_time_positive_biped(E, 0.4, feet), 0.25),
        ("flat_orientation_l2", T.flat_orientation_l2, -1.0),
        ("dof_pos_limits", lambda E: T.joint_pos_limits(E, ankles), -1.0),
        ("termination_penalty", T.is_terminated, -200.0),
        ("feet_slide", lambda E: T.feet_slide(E, feet, feet), -0.25),
        ("joint_deviation_hip", lambda E: T.joint_deviation_l1(E, hips), -0.2),
        ("joint_deviation_arms", lambda E: T.joint_deviation_l1(E, arms), -0.2),
        ("joint_deviation_torso", lambda E: T.joint_deviation_l1(E, torso_j), -0.1),
    ]
    del std
    terminations = [("time_out", T.time_out, True), ("base_contact", lambda E: T.illegal_contact(E, 1.0, torso), False)]
    obs_terms = [
        ("base_lin_vel", T.base_lin_vel, (-0.1, 0.1), None), ("base_ang_vel", T.base_ang_vel, (-0.2, 0.2), None),
        ("projected_gravity", T.projected_gravity, (-0.05, 0.05), None),
        ("velocity_commands", T.generated_commands, None, None),
        ("joint_pos", T.joint_pos_rel, (-0.01, 0.01), None), ("joint_vel", T.joint_vel_rel, (-1.5, 1.5), None),
        ("actions", T.last_action, None, None),
        ("height_scan", T.height_scan, (-0.1, 0.1), (-1.0, 1.0)),
    ]
    return dict(
        num_envs=num_envs, joint_names=joint_names, body_names=body_names, default_joint_pos=dq,
        soft_limit_factor=0.9, joint_pos_limits=(-2.6, 2.6),
        actuator=dict(kind="implicit", kp=kp, kd=kd, effort_limit=lim, velocity_limit=F(100.0)),
        rewards=rewards, terminations=terminations, obs_terms=obs_terms, sim_dt=0.005, decimation=4,
        episode_length_s=20.0, env_origins=np.asarray(env_origins, np.float32), action_scale=0.5,
        contact=dict(history=3, period=0.005, threshold=1.0),
        height_scan=dict(starts=ray_starts, dirs=ray_dirs, caster=caster, update_period=0.02, threads=threads),
        command=dict(resampling_time_range=(10.0, 10.0), lin_vel_x=(0.0, 1.0), lin_vel_y=(0.0, 0.0),
                     ang_vel_z=(-1.0, 1.0), heading=(-math.pi, math.pi), heading_command=True,
                     rel_heading_envs=1.0, rel_standing_envs=0.02, heading_control_stiffness=0.5),
    )
