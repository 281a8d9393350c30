#!/usr/bin/env python3
"""Times the REFERENCE's own torch path on the host cores of the BUILD container.

Needs /root/reference, so it never runs on the GPU box; nothing here travels.  The reference classes are imported
with the name-only stubs of tests/golden/ref_loader.py and driven by the duck-typed env of tests/golden/make_golden.py
(the way the reference's own manager tests mock the env), `device="cpu"`.

One timed MDP env-step, in the order of ManagerBasedRLEnv.step (LAB/envs/manager_based_rl_env.py:152-241):

    decimation x [ ActuatorBase.compute (the config's actuator class) + joint_acc finite difference
                   + ContactSensor._update_buffers_impl ]
    -> root-frame derived state (ArticulationData.root_lin_vel_b / root_ang_vel_b / projected_gravity_b / heading_w)
    -> TerminationManager.compute -> RewardManager.compute -> UniformVelocityCommand.compute
    -> ObservationManager.compute (rough configs: height_scan over the sensor's ray_hits_w)

The ray cast itself has NO runnable reference (warp is absent): its CPU time is this repo's own C/OpenMP caster
(oracle/raycast_ref.c through oracle.harness), timed separately and labelled "port".
C5 (Shadow hand) runs the DirectRLEnv.step order (direct_rl_env.py:293-380) through the reference methods
_apply_action / _get_dones / compute_rewards / compute_full_observations.

    python tools/ref_cpu_baseline.py [--steps 200 --warmup 20] [--configs C1,C2,...] [--write]
--write patches the table at the end of BASELINE.md and drops profiles/r02_ref_cpu_baseline.json.
"""
from __future__ import annotations

import argparse
import io
import json
import math
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_golden as MG  # noqa: E402  (installs the stubs; reference modules import after this)
import ref_loader as R  # noqa: E402

_NS = MG._NS


def build_velocity_case(n: int, joints: int, bodies: int, actuator: str, rough: bool, seed: int = 5):
    """Reference managers + actuator over a duck-typed env of n envs; returns step(k) running one MDP env-step."""
    m = R.ref_math()
    obs = R.ref_mdp("observations")
    rew = R.ref_mdp("rewards")
    term = R.ref_mdp("terminations")
    vrew = R.ref_velocity_mdp("rewards")
    mg = R.ref_import("isaaclab.managers")
    act = R.ref_import("isaaclab.actuators")
    cmdmod, _ = R.ref_velocity_command()
    cs = R.ref_import("isaaclab.sensors.contact_sensor.contact_sensor")
    from isaaclab.managers import SceneEntityCfg
    from isaaclab.utils.noise import UniformNoiseCfg as Unoise
    from isaaclab.utils.types import ArticulationActions

    g = torch.Generator().manual_seed(seed)
    rays = 187 if rough else 11
    d = MG.make_state(g, n, j=joints, b=bodies, h=3, r=rays)
    env = MG.make_env(d, m)
    robot = env.scene["robot"]
    names = [f"j{i}" for i in range(joints)]
    # ---- actuator of the config ----
    if actuator == "dc_motor":       # ANYDRIVE_3_SIMPLE_ACTUATOR_CFG (ASSETS/robots/anymal.py:34-41)
        cfg = act.DCMotorCfg(joint_names_expr=[".*"], saturation_effort=120.0, effort_limit=80.0, velocity_limit=7.5,
                             stiffness={".*": 40.0}, damping={".*": 5.0})
        a = act.DCMotor(cfg, names, slice(None), n, "cpu")
    elif actuator == "lstm":         # ANYDRIVE_3_LSTM_ACTUATOR_CFG (anymal.py:44-51) with the stand-in network
        from isaaclab_amd.envs.synthetic import synthetic_anydrive_lstm

        class SeaNet(torch.nn.Module):
            def __init__(self, src):
                super().__init__()
                self.lstm, self.decoder = src.lstm, src.decoder

            def forward(self, x, hc: tuple[torch.Tensor, torch.Tensor]):
                o, (h, c) = self.lstm(x, hc)
                return self.decoder(o[:, -1, :]), (h, c)

        buf = io.BytesIO()
        torch.jit.save(torch.jit.script(SeaNet(synthetic_anydrive_lstm())), buf)
        blob = buf.getvalue()
        import isaaclab.actuators.actuator_net as an

        an.read_file = lambda path, _b=blob: io.BytesIO(_b)
        cfg = act.ActuatorNetLSTMCfg(joint_names_expr=[".*"], network_file="unused", saturation_effort=120.0,
                                     effort_limit=80.0, velocity_limit=7.5)
        a = act.ActuatorNetLSTM(cfg, names, slice(None), n, "cpu")
    elif actuator == "implicit":     # H1_MINIMAL_CFG (ASSETS/robots/unitree.py): ImplicitActuator groups
        cfg = act.ImplicitActuatorCfg(joint_names_expr=[".*"], effort_limit=300.0, velocity_limit=100.0,
                                      stiffness={".*": 150.0}, damping={".*": 5.0})
        a = act.ImplicitActuator(cfg, names, slice(None), n, "cpu")
    else:
        raise ValueError(actuator)
    # ---- contact sensor (reference update code on a fake self, as in make_golden.gen_sensors_state) ----
    b, h = bodies, 3
    sensor = _NS()
    sensor.cfg = _NS(history_length=h, filter_prim_paths_expr=[], track_pose=False, track_air_time=True,
                     force_threshold=1.0, update_period=0.005)
    sensor._num_envs, sensor._num_bodies, sensor._sim_physics_dt = n, b, 0.005
    sensor._data = _NS(net_forces_w=torch.zeros(n, b, 3), net_forces_w_history=d["contact_hist"],
                       last_air_time=d["last_air"], current_air_time=d["cur_air"],
                       last_contact_time=d["last_contact"], current_contact_time=d["cur_contact"])
    sensor._timestamp = torch.zeros(n)
    sensor._timestamp_last_update = torch.zeros(n)
    ring = 64
    forces = torch.randn(ring, n * b, 3, generator=g) * 30.0 * (torch.rand(ring, n * b, 1, generator=g) < 0.3)
    qd_ring = torch.randn(ring, n, joints, generator=g) * 2.0
    q_ring = torch.randn(ring, n, joints, generator=g) * 0.2
    cur = {"t": 0}
    sensor.contact_physx_view = _NS(get_net_contact_forces=lambda dt: forces[cur["t"] % ring])
    all_envs = torch.arange(n)
    # ---- managers with the velocity-task term tables (TASKS/.../velocity/velocity_env_cfg.py) ----
    feet = list(range(bodies - 4, bodies))
    thighs = list(range(5, 9))

    def ent(name, body_ids=None):
        c = SceneEntityCfg(name)
        if body_ids is not None:
            c.body_ids = body_ids
        c.resolve = lambda scene: None
        return c

    rcfg = {
        "track_lin_vel_xy_exp": mg.RewardTermCfg(func=rew.track_lin_vel_xy_exp, weight=1.0,
                                                 params={"command_name": "base_velocity", "std": math.sqrt(0.25)}),
        "track_ang_vel_z_exp": mg.RewardTermCfg(func=rew.track_ang_vel_z_exp, weight=0.5,
                                                params={"command_name": "base_velocity", "std": math.sqrt(0.25)}),
        "lin_vel_z_l2": mg.RewardTermCfg(func=rew.lin_vel_z_l2, weight=-2.0),
        "ang_vel_xy_l2": mg.RewardTermCfg(func=rew.ang_vel_xy_l2, weight=-0.05),
        "dof_torques_l2": mg.RewardTermCfg(func=rew.joint_torques_l2, weight=-1.0e-5),
        "dof_acc_l2": mg.RewardTermCfg(func=rew.joint_acc_l2, weight=-2.5e-7),
        "action_rate_l2": mg.RewardTermCfg(func=rew.action_rate_l2, weight=-0.01),
        "feet_air_time": mg.RewardTermCfg(func=vrew.feet_air_time, weight=0.125,
                                          params={"sensor_cfg": ent("contact_forces", feet),
                                                  "command_name": "base_velocity", "threshold": 0.5}),
        "undesired_contacts": mg.RewardTermCfg(func=rew.undesired_contacts, weight=-1.0,
                                               params={"sensor_cfg": ent("contact_forces", thighs), "threshold": 1.0}),
        "flat_orientation_l2": mg.RewardTermCfg(func=rew.flat_orientation_l2, weight=0.0 if rough else -5.0),
        "dof_pos_limits": mg.RewardTermCfg(func=rew.joint_pos_limits, weight=0.0),
    }
    rm = mg.RewardManager(rcfg, env)
    tcfg = {
        "time_out": mg.TerminationTermCfg(func=term.time_out, time_out=True),
        "base_contact": mg.TerminationTermCfg(func=term.illegal_contact,
                                              params={"sensor_cfg": ent("contact_forces", [0]), "threshold": 1.0}),
    }
    tm = mg.TerminationManager(tcfg, env)
    grp = mg.ObservationGroupCfg()
    grp.enable_corruption = True
    grp.concatenate_terms = True
    grp.base_lin_vel = mg.ObservationTermCfg(func=obs.base_lin_vel, noise=Unoise(n_min=-0.1, n_max=0.1))
    grp.base_ang_vel = mg.ObservationTermCfg(func=obs.base_ang_vel, noise=Unoise(n_min=-0.2, n_max=0.2))
    grp.projected_gravity = mg.ObservationTermCfg(func=obs.projected_gravity, noise=Unoise(n_min=-0.05, n_max=0.05))
    grp.velocity_commands = mg.ObservationTermCfg(func=obs.generated_commands, params={"command_name": "base_velocity"})
    grp.joint_pos = mg.ObservationTermCfg(func=obs.joint_pos_rel, noise=Unoise(n_min=-0.01, n_max=0.01))
    grp.joint_vel = mg.ObservationTermCfg(func=obs.joint_vel_rel, noise=Unoise(n_min=-1.5, n_max=1.5))
    grp.actions = mg.ObservationTermCfg(func=obs.last_action)
    if rough:
        grp.height_scan = mg.ObservationTermCfg(func=obs.height_scan, params={"sensor_cfg": ent("height_scanner")},
                                                noise=Unoise(n_min=-0.1, n_max=0.1), clip=(-1.0, 1.0))
    om = mg.ObservationManager({"policy": grp}, env)
    # ---- command term (UniformVelocityCommand, velocity_command.py:48-160) ----
    ccfg = _NS(asset_name="robot", resampling_time_range=(10.0, 10.0), rel_standing_envs=0.02, rel_heading_envs=1.0,
               heading_command=True, heading_control_stiffness=0.5, debug_vis=False,
               ranges=_NS(lin_vel_x=(-1.0, 1.0), lin_vel_y=(-1.0, 1.0), ang_vel_z=(-1.0, 1.0), heading=(-math.pi, math.pi)))
    ct = cmdmod.UniformVelocityCommand.__new__(cmdmod.UniformVelocityCommand)
    ct.cfg, ct._env, ct.robot = ccfg, env, robot
    ct._debug_vis_handle = None
    ct.metrics = {"error_vel_xy": torch.zeros(n), "error_vel_yaw": torch.zeros(n)}
    ct.time_left = torch.rand(n, generator=g) * 10.0
    ct.command_counter = torch.zeros(n, dtype=torch.long)
    ct.vel_command_b = d["command"]
    ct.heading_target = torch.zeros(n)
    ct.is_heading_env = torch.zeros(n, dtype=torch.bool)
    ct.is_standing_env = torch.zeros(n, dtype=torch.bool)
    env.scene["robot"] = robot
    gravity = torch.tensor([0.0, 0.0, -1.0]).repeat(n, 1)
    fwd = torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1)
    prev_qd = torch.zeros(n, joints)
    decimation = 4
    raw = torch.randn(ring, n, joints, generator=g)
    default_q = d["default_joint_pos"]

    def step(k: int):
        # ActionManager.process_action + JointPositionAction (action_manager.py:318-348, joint_actions.py:130-160)
        am = env.action_manager
        am.prev_action[:] = am.action
        am.action[:] = raw[k % ring]
        target = am.action * 0.5 + default_q
        for s in range(decimation):
            t = (k * decimation + s) % ring
            cur["t"] = t
            q, qd = q_ring[t], qd_ring[t]
            ca = ArticulationActions(joint_positions=target, joint_velocities=torch.zeros_like(target),
                                     joint_efforts=torch.zeros_like(target))
            a.compute(ca, q, qd)                                   # ActuatorBase.compute of the config
            robot.data.applied_torque, robot.data.computed_torque = a.applied_effort, a.computed_effort
            # ArticulationData.joint_acc (articulation_data.py:455-470) + ContactSensor update of this sub-step
            robot.data.joint_acc = (qd - prev_qd) / 0.005
            prev_qd.copy_(qd)
            robot.data.joint_pos, robot.data.joint_vel = q, qd
            cs.ContactSensor._update_buffers_impl(sensor, all_envs)
        # root-frame derived state the terms read (articulation_data.py:716-760)
        quat = robot.data.root_quat_w
        robot.data.root_lin_vel_b = m.quat_rotate_inverse(quat, robot.data.root_lin_vel_w)
        robot.data.root_ang_vel_b = m.quat_rotate_inverse(quat, robot.data.root_ang_vel_w)
        robot.data.projected_gravity_b = m.quat_rotate_inverse(quat, gravity)
        f = m.quat_apply(quat, fwd)
        robot.data.heading_w = torch.atan2(f[:, 1], f[:, 0])
        env.episode_length_buf += 1
        tm.compute()
        rm.compute(0.02)
        ct.compute(0.02)
        return om.compute()["policy"]

    return step


def build_shadow_case(n: int, seed: int = 6):
    ih = R.ref_inhand()
    g = torch.Generator().manual_seed(seed)
    D, Fg, B, A = 24, 5, 26, 20
    obj = torch.zeros(n, 13)
    obj[:, :3] = torch.randn(n, 3, generator=g) * 0.05 + torch.tensor([0.0, -0.39, 0.6])
    obj[:, 3:7] = MG.rand_quat(g, n)
    obj[:, 7:] = torch.randn(n, 6, generator=g)
    origins = torch.randn(n, 3, generator=g) * 3.0
    obj[:, :3] += origins
    in_hand = torch.tensor([0.0, -0.39, 0.56]).repeat(n, 1)
    goal = MG.rand_quat(g, n)
    lower = -1.0 - torch.rand(n, D, generator=g)
    upper = 1.0 + torch.rand(n, D, generator=g)
    act_idx = sorted(torch.randperm(D, generator=g)[:A].tolist())
    fingers = sorted(torch.randperm(B, generator=g)[:Fg].tolist())
    ring = 16
    actions = torch.rand(ring, n, A, generator=g) * 2 - 1
    q = lower + (upper - lower) * torch.rand(n, D, generator=g)
    qd = torch.randn(n, D, generator=g)
    body = torch.randn(n, B, 13, generator=g)
    fake = _NS(cfg=_NS(act_moving_average=1.0, vel_obs_scale=0.2, fall_dist=0.24, max_consecutive_success=0,
                       success_tolerance=0.1, dist_reward_scale=-10.0, rot_reward_scale=1.0, rot_eps=0.1,
                       action_penalty_scale=-0.0002, reach_goal_bonus=250.0, fall_penalty=0.0, av_factor=0.1),
               num_envs=n, num_fingertips=Fg, finger_bodies=fingers, actions=actions[0], actuated_dof_indices=act_idx,
               hand_dof_lower_limits=lower, hand_dof_upper_limits=upper, prev_targets=torch.zeros(n, D),
               cur_targets=torch.zeros(n, D),
               hand=_NS(set_joint_position_target=lambda *a, **k: None,
                        data=_NS(body_pos_w=body[..., :3].clone(), body_quat_w=body[..., 3:7], body_vel_w=body[..., 7:13],
                                 joint_pos=q, joint_vel=qd)),
               object=_NS(data=_NS(root_pos_w=obj[:, :3], root_quat_w=obj[:, 3:7], root_vel_w=obj[:, 7:13],
                                   root_lin_vel_w=obj[:, 7:10], root_ang_vel_w=obj[:, 10:13])),
               scene=_NS(env_origins=origins), in_hand_pos=in_hand, goal_rot=goal, successes=torch.zeros(n),
               consecutive_successes=torch.zeros(1), episode_length_buf=torch.zeros(n, dtype=torch.long),
               max_episode_length=600, reset_buf=torch.zeros(n, dtype=torch.bool),
               reset_goal_buf=torch.zeros(n, dtype=torch.bool), extras={})
    cls = ih.InHandManipulationEnv
    fake._compute_intermediate_values = types.MethodType(cls._compute_intermediate_values, fake)

    def step(k: int):
        fake.actions = actions[k % ring]                          # _pre_physics_step
        for _ in range(2):                                        # decimation 2
            cls._apply_action(fake)
        fake.episode_length_buf += 1
        oor, to = cls._get_dones(fake)
        fake.reset_buf = oor | to
        # _get_rewards (:140-177) = compute_rewards + bookkeeping; called through the reference method's own body
        (rew_, fake.reset_goal_buf, fake.successes, fake.consecutive_successes) = ih.compute_rewards(
            fake.reset_buf, fake.reset_goal_buf, fake.successes, fake.consecutive_successes, 600.0,
            fake.object_pos, fake.object_rot, fake.in_hand_pos, fake.goal_rot, -10.0, 1.0, 0.1, fake.actions, -0.0002,
            0.1, 250.0, 0.24, 0.0, 0.1)
        fake.reset_goal_buf = torch.zeros_like(fake.reset_goal_buf)   # goals are re-drawn on the reset path (8f N1)
        return cls.compute_full_observations(fake)

    return step


def raycast_port_ms(n: int, threads: int) -> float | None:
    """This repo's C/OpenMP caster over the benchmark terrain (no reference exists for warp's mesh_query_ray)."""
    try:
        from oracle import raycast_np as RC
        from isaaclab_amd.terrains import SyntheticTerrain, SyntheticTerrainCfg
        from isaaclab_amd.terrains.synthetic import TerrainGeneratorCfg
    except Exception as e:  # pragma: no cover
        print(f"[ref_cpu_baseline] ray-cast port unavailable: {e}", file=sys.stderr)
        return None
    terr = SyntheticTerrain(SyntheticTerrainCfg(terrain_generator=TerrainGeneratorCfg(), max_init_terrain_level=5), n)
    rng = np.random.RandomState(3)
    origins = np.asarray(terr.env_origins, np.float32)
    gx, gy = np.meshgrid(np.arange(-0.8, 0.8 + 1e-6, 0.1), np.arange(-0.5, 0.5 + 1e-6, 0.1), indexing="xy")
    local = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, 20.0)], 1).astype(np.float32)
    starts = (origins[:, None, :] + local[None] + rng.uniform(-0.5, 0.5, (n, 1, 3)).astype(np.float32) * [1, 1, 0]).astype(np.float32)
    dirs = np.tile(np.array([0, 0, -1], np.float32), (n, local.shape[0], 1))
    caster = RC.GridCaster(terr.vertices, terr.triangles, threads=threads) if hasattr(RC, "GridCaster") else None
    if caster is None:
        return None
    caster.cast(starts.reshape(-1, 3), dirs.reshape(-1, 3), 1e6)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        caster.cast(starts.reshape(-1, 3), dirs.reshape(-1, 3), 1e6)
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


CONFIGS = {
    # tag: (description, builder kwargs, envs per GPU share)
    "C1": ("Flat Anymal-D, DCMotor", dict(joints=12, bodies=17, actuator="dc_motor", rough=False), 4096),
    "C2": ("Rough Anymal-D, DCMotor, height scan", dict(joints=12, bodies=17, actuator="dc_motor", rough=True), 4096),
    "C3": ("Rough Anymal-C, ActuatorNetLSTM", dict(joints=12, bodies=17, actuator="lstm", rough=True), 16384),
    "C4": ("Rough H1 (per-GPU share of 65536)", dict(joints=19, bodies=20, actuator="implicit", rough=True), 8192),
    "C5": ("Shadow hand direct (per-GPU share of 131072)", None, 16384),
}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--configs", default="C1,C2,C3,C4,C5")
    p.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    p.add_argument("--write", action="store_true")
    a = p.parse_args()
    torch.set_num_threads(a.threads)
    rows = {}
    for tag in a.configs.split(","):
        desc, kw, n = CONFIGS[tag]
        step = build_shadow_case(n) if kw is None else build_velocity_case(n, **kw)
        steps = a.steps if n <= 8192 else max(50, a.steps // 2)
        for k in range(a.warmup):
            step(k)
        ts = []
        for k in range(steps):
            t0 = time.perf_counter()
            step(a.warmup + k)
            ts.append(time.perf_counter() - t0)
        ts = np.asarray(ts)
        med = float(np.median(ts))
        rows[tag] = {"config": desc, "envs": n, "steps": steps, "warmup": a.warmup, "threads": torch.get_num_threads(),
                     "cores": os.cpu_count(), "ms_median": 1e3 * med, "ms_p10": 1e3 * float(np.percentile(ts, 10)),
                     "ms_p90": 1e3 * float(np.percentile(ts, 90)), "env_steps_per_s": n / med,
                     "ray_cast": "excluded (warp absent): height_scan reads the sensor's ray_hits_w buffer"
                     if kw and kw.get("rough") else "n/a"}
        print(f"{tag} {desc}: N={n} median {1e3 * med:.2f} ms/step (p10 {rows[tag]['ms_p10']:.2f}, p90 {rows[tag]['ms_p90']:.2f})"
              f" -> {n / med / 1e3:.1f} k env-steps/s on {torch.get_num_threads()} threads", flush=True)
    doc = {"what": "reference torch path on CPU, build container", "torch": torch.__version__, "rows": rows}
    print(json.dumps(doc))
    if a.write:
        with open(os.path.join(ROOT, "profiles", "r02_ref_cpu_baseline.json"), "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
