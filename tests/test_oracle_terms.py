"""CPU: numpy oracle of the MDP term library, managers, sensors, state container, command generator and the
in-hand direct env against reference-generated fixtures (tests/golden/{terms,managers,sensors_state,
command_action}.npz; see tests/golden/make_golden.py)."""
import math

import numpy as np
import pytest

from conftest import assert_close
from oracle import math_np as M
from oracle import mdp_np as O

F = np.float32
FEET, THIGHS, SOME_J = [13, 14, 15, 16], [5, 6, 7, 8], [0, 3, 7, 11]


def env_dict(G, p):
    """golden 'in_*' arrays -> the oracle's env dict."""
    g = lambda k: G[p + "in_" + k]
    return dict(
        root_state=g("root_state"), env_origins=g("env_origins"), joint_pos=g("joint_pos"), joint_vel=g("joint_vel"),
        joint_acc=g("joint_acc"), default_joint_pos=g("default_joint_pos"), default_joint_vel=g("default_joint_vel"),
        soft_joint_pos_limits=g("soft_joint_pos_limits"), soft_joint_vel_limits=g("soft_joint_vel_limits"),
        applied_torque=g("applied_torque"), computed_torque=g("computed_torque"), action=g("action"),
        prev_action=g("prev_action"), command=g("command"), body_state=g("body_state"), body_acc_w=g("body_acc"),
        contact_hist=g("contact_hist"), cur_air_time=g("cur_air"), cur_contact_time=g("cur_contact"),
        last_air_time=g("last_air"), last_contact_time=g("last_contact"), ray_hits_w=g("ray_hits"),
        sensor_pos_w=g("sensor_pos"), episode_length_buf=g("episode_length"), max_episode_length=1000,
        time_left=g("time_left"), command_counter=g("command_counter"), terminated=g("terminated"),
        time_outs=g("time_outs"), term_dones=g("term_dones"), step_dt=0.02,
    )


def term_table(E):
    """name -> oracle value, for every stock term in tests/golden/terms.npz (shared with the GPU tests)."""
    T = O.Terms
    obs = {
        "base_pos_z": T.base_pos_z(E), "base_lin_vel": T.base_lin_vel(E), "base_ang_vel": T.base_ang_vel(E),
        "projected_gravity": T.projected_gravity(E), "root_pos_w": T.root_pos_w(E), "root_quat_w": T.root_quat_w(E),
        "root_quat_w_unique": T.root_quat_w(E, True), "root_lin_vel_w": T.root_lin_vel_w(E),
        "root_ang_vel_w": T.root_ang_vel_w(E), "joint_pos": T.joint_pos(E), "joint_pos_sel": T.joint_pos(E, SOME_J),
        "joint_pos_rel": T.joint_pos_rel(E), "joint_pos_rel_sel": T.joint_pos_rel(E, SOME_J),
        "joint_pos_limit_normalized": T.joint_pos_limit_normalized(E), "joint_vel": T.joint_vel(E),
        "joint_vel_rel": T.joint_vel_rel(E), "height_scan": T.height_scan(E), "last_action": T.last_action(E),
        "generated_commands": T.generated_commands(E),
    }
    rew = {
        "is_alive": T.is_alive(E), "is_terminated": T.is_terminated(E), "is_terminated_term": T.is_terminated_term(E, [0, 1]),
        "lin_vel_z_l2": T.lin_vel_z_l2(E), "ang_vel_xy_l2": T.ang_vel_xy_l2(E),
        "flat_orientation_l2": T.flat_orientation_l2(E), "base_height_l2": T.base_height_l2(E, 0.55),
        "base_height_l2_sensor": T.base_height_l2(E, 0.55, True), "body_lin_acc_l2": T.body_lin_acc_l2(E, FEET),
        "joint_torques_l2": T.joint_torques_l2(E), "joint_vel_l1": T.joint_vel_l1(E, SOME_J),
        "joint_vel_l2": T.joint_vel_l2(E), "joint_acc_l2": T.joint_acc_l2(E),
        "joint_deviation_l1": T.joint_deviation_l1(E, SOME_J), "joint_pos_limits": T.joint_pos_limits(E),
        "joint_vel_limits": T.joint_vel_limits(E, 0.9), "applied_torque_limits": T.applied_torque_limits(E),
        "action_rate_l2": T.action_rate_l2(E), "action_l2": T.action_l2(E),
        "undesired_contacts": T.undesired_contacts(E, 1.0, THIGHS), "contact_forces": T.contact_forces(E, 1.0, FEET),
        "track_lin_vel_xy_exp": T.track_lin_vel_xy_exp(E, math.sqrt(0.25)),
        "track_ang_vel_z_exp": T.track_ang_vel_z_exp(E, math.sqrt(0.25)),
        "feet_air_time": T.feet_air_time(E, 0.5, FEET),
        "feet_air_time_positive_biped": T.feet_air_time_positive_biped(E, 0.4, FEET[:2]),
        "feet_slide": T.feet_slide(E, FEET, FEET),
        "track_lin_vel_xy_yaw_frame_exp": T.track_lin_vel_xy_yaw_frame_exp(E, 0.5),
        "track_ang_vel_z_world_exp": T.track_ang_vel_z_world_exp(E, 0.5),
    }
    term = {
        "time_out": T.time_out(E), "command_resample": T.command_resample(E, 1),
        "bad_orientation": T.bad_orientation(E, 1.0), "root_height_below_minimum": T.root_height_below_minimum(E, 0.55),
        "joint_pos_out_of_manual_limit": T.joint_pos_out_of_manual_limit(E, (-0.9, 0.9), SOME_J),
        "joint_vel_out_of_limit": T.joint_vel_out_of_limit(E),
        "joint_vel_out_of_manual_limit": T.joint_vel_out_of_manual_limit(E, 6.0),
        "joint_effort_out_of_limit": T.joint_effort_out_of_limit(E, SOME_J),
        "illegal_contact": T.illegal_contact(E, 1.0, [0]),
    }
    return obs, rew, term


# sums of up to 17x3 squared fp32 values: order of accumulation differs between numpy (pairwise) and torch
SUM_TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n", (1, 7, 64))
def test_term_library(golden, n):
    G = golden("terms")
    p = f"n{n}_"
    obs, rew, term = term_table(env_dict(G, p))
    for k, v in obs.items():
        assert_close(v, G[p + "obs_" + k], msg="obs " + k)
    for k, v in rew.items():
        assert_close(np.asarray(v, np.float32), G[p + "rew_" + k], msg="rew " + k, **SUM_TOL)
    for k, v in term.items():
        assert_close(v, G[p + "term_" + k], msg="term " + k)
    # terminations.py:75-87 indexes a (N, J, 2) tensor with [..., 0] of a 2-D slice: the reference raises
    assert int(G[p + "term_joint_pos_out_of_limit_raises"]) == 1


@pytest.mark.parametrize("n", (7, 64))
def test_managers(golden, n):
    G = golden("managers")
    p = f"n{n}_"
    E = env_dict(G, p)
    T = O.Terms
    values = [T.track_lin_vel_xy_exp(E, math.sqrt(0.25)), T.track_ang_vel_z_exp(E, math.sqrt(0.25)), T.lin_vel_z_l2(E),
              T.ang_vel_xy_l2(E), T.joint_torques_l2(E), T.joint_acc_l2(E), T.action_rate_l2(E),
              T.feet_air_time(E, 0.5, FEET), T.undesired_contacts(E, 1.0, THIGHS), None, None]
    weights = [1.0, 0.5, -2.0, -0.05, -1.0e-5, -2.5e-7, -0.01, 0.125, -1.0, 0.0, 0.0]
    sums = np.zeros((len(weights), n), np.float32)
    step = np.zeros((n, len(weights)), np.float32)
    r1 = O.reward_compute(values, weights, 0.02, sums, step)
    r2 = O.reward_compute(values, weights, 0.02, sums, step)
    assert_close(r1, G[p + "reward_1"], **SUM_TOL)
    assert_close(r2, G[p + "reward_2"], **SUM_TOL)
    assert_close(sums, G[p + "episode_sums"], **SUM_TOL)
    assert_close(step, G[p + "step_reward"], rtol=1e-5, atol=1e-4)
    tvals = [T.time_out(E), T.illegal_contact(E, 1.0, [0]), T.bad_orientation(E, 1.0)]
    trunc, term = O.termination_compute(tvals, [True, False, False])
    assert_close(trunc | term, G[p + "dones"])
    assert_close(term, G[p + "terminated"])
    assert_close(trunc, G[p + "time_outs"])
    assert_close(np.stack(tvals), G[p + "term_dones"])
    obs = np.concatenate([
        O.obs_postprocess(T.base_lin_vel(E), noise=("constant", "add", 0.05, 0, None)),
        O.obs_postprocess(T.base_ang_vel(E), scale=0.25),
        O.obs_postprocess(T.projected_gravity(E), clip=(-0.5, 0.5)),
        T.generated_commands(E),
        O.obs_postprocess(T.joint_pos_rel(E), noise=("constant", "scale", 1.1, 0, None)),
        O.obs_postprocess(T.joint_vel_rel(E), scale=[0.05 * (i + 1) for i in range(12)]),
        T.last_action(E),
        O.obs_postprocess(T.height_scan(E), noise=("constant", "add", -0.02, 0, None), clip=(-1.0, 1.0)),
    ], axis=1)
    assert_close(obs, G[p + "obs"])
    # history (observation_manager.py:296-312): oldest first, first push back-fills the whole history
    from oracle.actuators_np import CircularBuffer

    def push(cbs, E):
        outs = []
        for cb, v in zip(cbs, (T.base_lin_vel(E), T.joint_pos_rel(E))):
            cb.append(v)
            outs.append(cb.buffer.reshape(n, -1))
        return np.concatenate(outs, axis=1)

    cbs = [CircularBuffer(3, n), CircularBuffer(3, n)]
    assert_close(push(cbs, E), G[p + "obs_hist_1"])
    E2 = dict(E, joint_pos=E["joint_pos"] + F(0.1))
    assert_close(push(cbs, E2), G[p + "obs_hist_2"])


def test_contact_sensor_sequence(golden):
    G = golden("sensors_state")
    seq = G["contact_forces_seq"]
    T_, n, b, _ = seq.shape
    s = O.ContactSensor(n, b, 3, 0.005)
    for t in range(T_):
        s.update(seq[t], 0.005)
        assert_close(s.hist, G["contact_hist"][t], msg=f"hist t={t}")
        for k, a in (("last_air", s.last_air), ("cur_air", s.cur_air), ("last_con", s.last_con), ("cur_con", s.cur_con),
                     ("ts", s.ts)):
            assert_close(a, G["contact_" + k][t], msg=f"{k} t={t}")
        assert_close(s.compute_first_contact(0.02), G["contact_first_contact"][t], msg=f"first t={t}")
        if t == 6:
            ids = [1, 4]
            s.ts[ids] = 0.0
            s.ts_last[ids] = 0.0
            s.outdated[ids] = True
            for a in (s.net, s.hist, s.cur_air, s.last_air, s.cur_con, s.last_con):
                a[ids] = 0.0


def test_ray_generation(golden):
    G = golden("sensors_state")
    starts, dirs = O.grid_pattern(0.1, [1.6, 1.0])
    assert_close(starts, G["grid_starts"], atol=1e-7)
    assert_close(dirs, G["grid_dirs"])
    assert starts.shape[0] == 187  # the Anymal height scanner (rough_env_cfg: 1.6 x 1.0 at 0.1)
    for tag, yaw_only in (("yaw", True), ("full", False)):
        pose = G[f"rays_{tag}_pose_xyzw"]
        pos = pose[:, :3] + G[f"rays_{tag}_drift"]
        quat = np.roll(pose[:, 3:7], 1, axis=-1)
        sl = starts + np.array([0, 0, 20.0], np.float32)
        sw, dw = O.raycaster_rays(pos, quat, sl, dirs, yaw_only)
        assert_close(sw, G[f"rays_{tag}_starts_w"])
        assert_close(dw, G[f"rays_{tag}_dirs_w"])
        assert_close(pos, G[f"rays_{tag}_pos_w"])
        assert_close(quat, G[f"rays_{tag}_quat_w"])


def test_articulation_state(golden):
    G = golden("sensors_state")
    st = O.state_pack(G["ad_pose_xyzw"], G["ad_vel"])
    assert_close(st, G["ad_root_state_w"])
    body = O.state_pack(G["ad_lpose_xyzw"], G["ad_lvel"])
    assert_close(body, G["ad_body_state_w"])
    assert_close(O.link_state_from_com(body, G["ad_coms"][..., :3]), G["ad_body_link_state_w"])
    assert_close(O.link_state_from_com(st, G["ad_coms"][:, 0, :3]), G["ad_root_link_state_w"])
    assert_close(O.projected_gravity_b(st), G["ad_projected_gravity_b"])
    assert_close(O.heading_w(st), G["ad_heading_w"])
    assert_close(O.root_lin_vel_b(st), G["ad_root_lin_vel_b"])
    assert_close(O.root_ang_vel_b(st), G["ad_root_ang_vel_b"])
    prev = np.zeros_like(G["ad_qd_seq"][0])
    for t, dt in enumerate((0.005, 0.005, 0.01)):
        assert_close(O.joint_acc(G["ad_qd_seq"][t], prev, dt), G["ad_joint_acc"][t], rtol=2e-5, msg=f"t={t}")
        prev = G["ad_qd_seq"][t]


CFG_A = dict(resampling_time_range=(10.0, 10.0), rel_standing_envs=0.02, rel_heading_envs=1.0, heading_command=True,
             heading_control_stiffness=0.5, lin_vel_x=(-1.0, 1.0), lin_vel_y=(-1.0, 1.0), ang_vel_z=(-1.0, 1.0),
             heading=(-math.pi, math.pi))
CFG_B = dict(CFG_A, rel_standing_envs=0.0, lin_vel_x=(0.7, 0.7), lin_vel_y=(-0.3, -0.3), heading=(0.4, 0.4))


def test_velocity_command(golden):
    G = golden("command_action")
    st = G["root_state"]
    n = st.shape[0]
    c = O.VelocityCommand(n, CFG_A, 0.02)
    c.cmd, c.heading_target = G["A_in_cmd"].copy(), G["A_in_heading_target"].copy()
    c.is_heading, c.is_standing = G["A_in_is_heading"].copy(), G["A_in_is_standing"].copy()
    c.time_left = np.full(n, 5.0, np.float32)
    u = np.full((n, 7), 0.5, np.float32)
    c.compute(st, 0.02, u)
    c.compute(st, 0.02, u)
    assert_close(c.cmd, G["A_cmd"], atol=1e-6)
    assert_close(c.time_left, G["A_time_left"])
    assert_close(c.m_xy, G["A_m_xy"])
    assert_close(c.m_yaw, G["A_m_yaw"])
    assert_close(c.counter, G["A_counter"])
    # case B: degenerate ranges -> the uniforms do not matter except ang_vel_z (overwritten by heading control)
    c.cfg = CFG_B
    c.cmd, c.time_left = G["B_in_cmd"].copy(), G["B_in_time_left"].copy()
    c.heading_target = G["B_in_heading_target"].copy()
    c.is_heading[:] = False
    c.is_standing[:] = False
    c.compute(st, 0.02, u)
    resampled = np.arange(n) % 2 == 0
    assert_close(c.cmd[:, :2], G["B_cmd"][:, :2], atol=1e-6)
    assert_close(c.cmd[resampled, 2], G["B_cmd"][resampled, 2], atol=1e-6)   # heading-controlled
    assert_close(c.time_left, G["B_time_left"], atol=1e-6)
    assert_close(c.counter, G["B_counter"])
    assert_close(c.heading_target, G["B_heading_target"], atol=1e-6)
    assert_close(c.is_heading, G["B_is_heading"])
    assert_close(c.is_standing, G["B_is_standing"])
