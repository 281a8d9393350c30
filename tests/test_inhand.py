"""Shadow-hand in-hand manipulation (SURVEY.md 8a row T-shadow; TASKS/direct/inhand_manipulation/
inhand_manipulation_env.py:97-117,179-199,262-433).

* not gpu: the numpy oracle against the reference-generated fixture tests/golden/inhand.npz;
* gpu: the host mirror ``InHandManipulationEnv`` (HIP through the C ABI) against the same fixture, and against the
  oracle at the per-GPU size of BASELINE config C5 (131072 envs / 8 GPUs = 16384 envs).
Masks / counters bit-exact, fp32 <= 1e-5 relative with the 2e-6 absolute floor of conftest.py (rewards span -10 .. 260;
measured worst cases in profiles/r02_parity_error_table.txt)."""
import numpy as np
import pytest

from conftest import assert_close, to_dev, to_np

KW = dict(dist_reward_scale=-10.0, rot_reward_scale=1.0, rot_eps=0.1, action_penalty_scale=-0.0002,
          success_tolerance=0.1, reach_goal_bonus=250.0, fall_dist=0.24, fall_penalty=0.0, av_factor=0.1)


@pytest.mark.parametrize("n", (7, 64))
def test_oracle_inhand(golden, n):
    from oracle import inhand_np as O

    G = golden("inhand")
    p = f"n{n}_"
    assert_close(O.rotation_distance(G[p + "obj"][:, 3:7], G[p + "goal"]), G[p + "rotation_distance"])
    rew, gr, succ, cons = O.compute_rewards(G[p + "reset_buf"], G[p + "reset_goal"], G[p + "successes"], G[p + "cons"],
                                            G[p + "obj"], G[p + "origins"], G[p + "in_hand"], G[p + "goal"],
                                            G[p + "actions"], **KW)
    assert_close(rew, G[p + "reward"], msg=f"oracle reward n={n}")
    np.testing.assert_array_equal(gr, G[p + "goal_resets"])
    assert_close(succ, G[p + "successes_out"])
    assert_close(cons, G[p + "cons_out"])
    cur, prev = O.apply_action(G[p + "actions"], G[p + "act_idx"], G[p + "lower"], G[p + "upper"],
                               np.zeros_like(G[p + "prev"]), G[p + "prev"], 0.3)
    assert_close(cur, G[p + "cur_out"])
    assert_close(prev, G[p + "prev_out"])
    oor, to, ep = O.dones(G[p + "obj"], G[p + "origins"], G[p + "in_hand"], G[p + "goal"], G[p + "successes"],
                          G[p + "ep"], 0.24, 0.1, 3, 600)
    np.testing.assert_array_equal(oor, G[p + "out_of_reach"])
    np.testing.assert_array_equal(to, G[p + "time_out"])
    np.testing.assert_array_equal(ep, G[p + "ep_out"])
    obs = O.full_observations(G[p + "q"], G[p + "qd"], G[p + "lower"], G[p + "upper"], G[p + "obj"], G[p + "origins"],
                              G[p + "in_hand"], G[p + "goal"], G[p + "body"], G[p + "fingers"], G[p + "actions"], 0.2)
    assert obs.shape == (n, 157)
    assert_close(obs, G[p + "full_obs"])


def _reset_state(G, p, n):
    """buffers of the fixture's _reset_idx case, keyed like oracle.inhand_np.reset_idx's ``st``"""
    x = np.tile(np.array([1.0, 0.0, 0.0], np.float32), (n, 1))
    y = np.tile(np.array([0.0, 1.0, 0.0], np.float32), (n, 1))
    return dict(goal_rot=G[p + "goal"], reset_goal_buf=G[p + "reset_goal_in"], object_default_root_state=G[p + "object_default"],
                env_origins=G[p + "origins"], object_root_state_w=G[p + "obj"], default_joint_pos=G[p + "default_joint_pos"],
                default_joint_vel=G[p + "default_joint_vel"], lower=G[p + "lower"], upper=G[p + "upper"],
                prev_targets=G[p + "in_prev_targets"], cur_targets=G[p + "in_cur_targets"],
                hand_dof_targets=G[p + "in_hand_dof_targets"], joint_pos_target=G[p + "in_joint_pos_target"],
                joint_pos=G[p + "q"], joint_vel=G[p + "qd"], prev_joint_vel=G[p + "in_prev_joint_vel"],
                joint_acc=G[p + "in_joint_acc"], successes=G[p + "in_successes"], episode_length_buf=G[p + "in_ep"],
                x_unit=x, y_unit=y)


_RESET_OUT = (("goal_rot", "r_goal"), ("object_root_state_w", "r_object_state"), ("prev_targets", "r_prev_targets"),
              ("cur_targets", "r_cur_targets"), ("hand_dof_targets", "r_hand_dof_targets"),
              ("joint_pos_target", "r_joint_pos_target"), ("joint_pos", "r_joint_pos"), ("joint_vel", "r_joint_vel"),
              ("prev_joint_vel", "r_prev_joint_vel"), ("joint_acc", "r_joint_acc"), ("successes", "r_successes"))


@pytest.mark.parametrize("n", (7, 48))
def test_oracle_inhand_reset_and_other_layouts(golden, n):
    """compute_reduced_observations, compute_full_state, _reset_target_pose, _reset_idx of the oracle against vectors the
    reference methods produced from recorded torch.rand draws (tests/golden/make_golden.py:gen_inhand_reset)."""
    from oracle import inhand_np as O

    G = golden("inhand_reset")
    p = f"n{n}_"
    fingers = G[p + "fingers"]
    red = O.reduced_observations(G[p + "obj"], G[p + "origins"], G[p + "goal"], G[p + "body"], fingers, G[p + "actions"])
    assert red.shape == (n, 42)
    assert_close(red, G[p + "reduced_obs"])
    st = O.full_state(G[p + "q"], G[p + "qd"], G[p + "lower"], G[p + "upper"], G[p + "obj"], G[p + "origins"], G[p + "in_hand"],
                      G[p + "goal"], G[p + "body"], fingers, G[p + "forces"][:, fingers], G[p + "actions"], 0.2, 10.0)
    assert st.shape == (n, 187)
    assert_close(st, G[p + "full_state"])
    x = np.tile(np.array([1.0, 0.0, 0.0], np.float32), (n, 1))
    y = np.tile(np.array([0.0, 1.0, 0.0], np.float32), (n, 1))
    mask = np.zeros(n, bool); mask[G[p + "ids"]] = True
    goal, rg = O.reset_target_pose(G[p + "goal"], G[p + "reset_goal_in"], mask, G[p + "u"], x, y)
    assert_close(goal, G[p + "goal_after_target_pose"])
    np.testing.assert_array_equal(rg, G[p + "reset_goal_after_target_pose"])
    mask2 = np.zeros(n, bool); mask2[G[p + "ids2"]] = True
    out = O.reset_idx(mask2, G[p + "u"], _reset_state(G, p, n), 0.01, 0.2, 0.3)
    for key, ref in _RESET_OUT:
        assert_close(out[key], G[p + ref], msg=f"{key} n={n}")
    np.testing.assert_array_equal(out["reset_goal_buf"], G[p + "r_reset_goal"])
    np.testing.assert_array_equal(out["episode_length_buf"], G[p + "r_ep"])


def _make_env(n, device, origins, lower, upper, act_idx=None, fingers=None, cfg_cls=None, **cfg_over):
    import torch

    from isaaclab_amd.tasks.inhand import InHandManipulationEnv, ShadowHandEnvCfg

    cfg = (cfg_cls or ShadowHandEnvCfg)(num_envs=n, **cfg_over)
    limits = torch.stack([torch.from_numpy(lower), torch.from_numpy(upper)], dim=-1)
    env = InHandManipulationEnv(cfg, str(device), env_origins=torch.from_numpy(origins), joint_pos_limits=limits)
    if act_idx is not None:       # the fixtures use random index sets, not the named Shadow joints
        env.actuated_dof_indices = [int(i) for i in act_idx]
        env._act_idx = torch.tensor(env.actuated_dof_indices, dtype=torch.int32, device=device)
    if fingers is not None:
        env.finger_bodies = [int(i) for i in fingers]
        env._fingers = torch.tensor(env.finger_bodies, dtype=torch.int32, device=device)
    return env


def _feed(env, device, obj, body=None, q=None, qd=None):
    xyzw = np.concatenate([obj[:, :3], obj[:, 4:7], obj[:, 3:4]], axis=1)
    env.object_view.transforms.copy_(to_dev(xyzw, device))
    env.object_view.velocities.copy_(to_dev(obj[:, 7:], device))
    if body is not None:
        lt = np.concatenate([body[..., :3], body[..., 4:7], body[..., 3:4]], axis=-1)
        env.hand_view.get_link_transforms().copy_(to_dev(lt, device))
        env.hand_view.get_link_velocities().copy_(to_dev(body[..., 7:], device))
    if q is not None:
        env.hand_view.get_dof_positions().copy_(to_dev(q, device))
        env.hand_view.get_dof_velocities().copy_(to_dev(qd, device))
    env.hand.data.update(env.physics_dt)
    env.object.update(env.physics_dt)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", (True, False), ids=("fused_step", "one_launch_per_statement"))
@pytest.mark.parametrize("n", (7, 64))
def test_gpu_inhand_vs_reference_fixture(golden, device, n, fused):
    G = golden("inhand")
    p = f"n{n}_"
    env = _make_env(n, device, G[p + "origins"], G[p + "lower"], G[p + "upper"], G[p + "act_idx"], G[p + "fingers"],
                    act_moving_average=0.3, max_consecutive_success=3, episode_length_s=10.0)
    env.fused_step = fused
    assert env.max_episode_length == 600 and env.num_full_obs == 157
    env.in_hand_pos.copy_(to_dev(G[p + "in_hand"], device))
    env.goal_rot.copy_(to_dev(G[p + "goal"], device))
    env._pre_physics_step(to_dev(G[p + "actions"], device))
    # _apply_action
    env.prev_targets.copy_(to_dev(G[p + "prev"], device))
    env._apply_action()
    np.testing.assert_array_equal(to_np(env.actions), G[p + "actions"])   # the clone of _pre_physics_step (fused: same launch)
    assert_close(to_np(env.cur_targets), G[p + "cur_out"])
    assert_close(to_np(env.prev_targets), G[p + "prev_out"])
    assert_close(to_np(env.hand.joint_pos_target), G[p + "cur_out"])
    # _get_dones (max_consecutive_success = 3 in the fixture)
    _feed(env, device, G[p + "obj"], G[p + "body"], G[p + "q"], G[p + "qd"])
    env.successes.copy_(to_dev(G[p + "successes"], device))
    env.episode_length_buf.copy_(to_dev(G[p + "ep"], device))
    oor, to = env._get_dones()
    np.testing.assert_array_equal(to_np(oor), G[p + "out_of_reach"])
    np.testing.assert_array_equal(to_np(to), G[p + "time_out"])
    np.testing.assert_array_equal(to_np(env.episode_length_buf), G[p + "ep_out"])
    # the same through the step form: episode_length_buf += 1, _get_dones, reset_terminated / _time_outs / reset_buf
    env.episode_length_buf.copy_(to_dev(G[p + "ep"] - 1, device))
    env._dones_step()
    np.testing.assert_array_equal(to_np(env.reset_terminated), G[p + "out_of_reach"])
    np.testing.assert_array_equal(to_np(env.reset_time_outs), G[p + "time_out"])
    np.testing.assert_array_equal(to_np(env.reset_buf), G[p + "out_of_reach"] | G[p + "time_out"])
    np.testing.assert_array_equal(to_np(env.episode_length_buf), G[p + "ep_out"])
    # observations
    assert_close(to_np(env._get_observations()["policy"]), G[p + "full_obs"])
    # _get_rewards
    env.reset_buf.copy_(to_dev(G[p + "reset_buf"], device))
    env.reset_goal_buf.copy_(to_dev(G[p + "reset_goal"], device))
    env.consecutive_successes.copy_(to_dev(G[p + "cons"], device))
    rew = env._get_rewards()
    assert_close(to_np(rew), G[p + "reward"], msg=f"reward n={n}")
    np.testing.assert_array_equal(to_np(env.reset_goal_buf), G[p + "goal_resets"])
    assert_close(to_np(env.successes), G[p + "successes_out"])
    assert_close(to_np(env.consecutive_successes), G[p + "cons_out"])
    assert float(env.extras["log"]["consecutive_successes"]) == pytest.approx(float(G[p + "cons_out"][0]), rel=1e-5)
    if fused:      # the partial sums are left zeroed for the next step (mdp_inhand_finish_rewards_step)
        assert to_np(env._partials).tolist() == [0.0, 0.0]
        rew2 = env._get_rewards()      # a second evaluation starts from clean sums: same reset count, same average
        assert np.isfinite(to_np(rew2)).all() and to_np(env._partials).tolist() == [0.0, 0.0]


@pytest.mark.gpu
@pytest.mark.parametrize("fused", (True, False), ids=("fused_step", "one_launch_per_statement"))
def test_gpu_inhand_step_full_size_vs_oracle(device, fused):
    """DirectRLEnv.step order at the per-GPU size of BASELINE config C5 (16384 envs, 24 DoF, 5 fingertips)."""
    from oracle import inhand_np as O

    n, D, B, A = 16384, 24, 26, 20
    rng = np.random.RandomState(11)
    f = lambda *s: rng.randn(*s).astype(np.float32)
    origins = f(n, 3) * 3
    lower = -1.0 - rng.rand(n, D).astype(np.float32)
    upper = 1.0 + rng.rand(n, D).astype(np.float32)
    env = _make_env(n, device, origins, lower, upper, act_moving_average=0.7)
    env.fused_step = fused
    quat = f(n, 4); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    goal = f(n, 4); goal /= np.linalg.norm(goal, axis=1, keepdims=True)
    goal[::9] = quat[::9]
    obj = np.concatenate([f(n, 3) * 0.08 + np.array(env.cfg.object_init_pos, np.float32) + origins, quat, f(n, 6)], 1)
    obj = obj.astype(np.float32)
    body = f(n, B, 13)
    q = (lower + (upper - lower) * rng.rand(n, D)).astype(np.float32)
    qd = f(n, D)
    actions = (rng.rand(n, A) * 2 - 1).astype(np.float32)
    env.goal_rot.copy_(to_dev(goal, device))
    prev = f(n, D) * 0.5
    env.prev_targets.copy_(to_dev(prev, device))
    succ0 = rng.randint(0, 5, n).astype(np.float32)
    env.successes.copy_(to_dev(succ0, device))
    ep0 = rng.randint(0, 400, n)
    ep0[::50] = env.max_episode_length - 2
    env.episode_length_buf.copy_(to_dev(ep0, device))
    env.consecutive_successes.fill_(1.3)
    # the synthetic feed: hold the same frame through both sub-steps
    _feed(env, device, obj, body, q, qd)
    env.hand_view.advance = lambda: None
    # reset path on (the reference's step always resets): goal reset inside _get_rewards, masked _reset_idx; the draws are
    # injected so that the oracle consumes the same numbers
    assert env.auto_reset == "masked" and env.reset_goals
    u = rng.rand(n, 7 + 2 * D).astype(np.float32)
    env.inject_uniforms(to_dev(u, device))
    dq0 = (f(n, D) * 0.2).astype(np.float32)
    env.hand.data.default_joint_pos.copy_(to_dev(dq0, device))
    obs, rew, term, tout, extras = env.step(to_dev(actions, device))
    # oracle, same order (direct_rl_env.py:323-372)
    act_idx = np.array(env.actuated_dof_indices)
    cur, pv = np.zeros((n, D), np.float32), prev
    for _ in range(env.cfg.decimation):
        cur, pv = O.apply_action(actions, act_idx, lower, upper, cur, pv, 0.7)
    in_hand = to_np(env.in_hand_pos)
    oor, to, ep = O.dones(obj, origins, in_hand, goal, succ0, ep0 + 1, 0.24, 0.1, 0, env.max_episode_length)
    r, gr, succ, cons = O.compute_rewards(oor | to, np.zeros(n, bool), succ0, np.array([1.3], np.float32), obj, origins,
                                          in_hand, goal, actions, **KW)
    # :172-175 goal reset of the envs that reached the goal, then direct_rl_env.py:355-358 _reset_idx of the done envs
    xu = np.tile(np.array([1.0, 0.0, 0.0], np.float32), (n, 1))
    yu = np.tile(np.array([0.0, 1.0, 0.0], np.float32), (n, 1))
    goal2, rg2 = O.reset_target_pose(goal, gr, gr.astype(bool), u, xu, yu)
    done = oor | to
    obj_default = np.zeros((n, 13), np.float32); obj_default[:, 3] = 1.0
    obj_default[:, :3] = np.array(env.cfg.object_init_pos, np.float32)
    tgt = np.zeros((n, D), np.float32); tgt[:, act_idx] = cur[:, act_idx]
    st = O.reset_idx(done, u, dict(
        goal_rot=goal2, reset_goal_buf=rg2, object_default_root_state=obj_default, env_origins=origins,
        object_root_state_w=obj, default_joint_pos=dq0, default_joint_vel=np.zeros((n, D), np.float32), lower=lower,
        upper=upper, prev_targets=pv, cur_targets=cur, hand_dof_targets=np.zeros((n, D), np.float32), joint_pos_target=tgt,
        joint_pos=q, joint_vel=qd, prev_joint_vel=qd, joint_acc=np.zeros((n, D), np.float32), successes=succ,
        episode_length_buf=ep, x_unit=xu, y_unit=yu), env.cfg.reset_position_noise, env.cfg.reset_dof_pos_noise,
        env.cfg.reset_dof_vel_noise)
    o = O.full_observations(st["joint_pos"], st["joint_vel"], lower, upper, st["object_root_state_w"], origins, in_hand,
                            st["goal_rot"], body, np.array(env.finger_bodies), actions, 0.2)
    np.testing.assert_array_equal(to_np(term), oor)
    np.testing.assert_array_equal(to_np(tout), to)
    assert to.sum() > 0 and oor.sum() > 0 and gr.sum() > 0 and (gr.astype(bool) & ~done).sum() > 0
    assert not to_np(env.reset_goal_buf).any()              # every goal that was reached has been replaced (:260)
    assert_close(to_np(rew), r, msg="reward 16384 envs vs oracle")
    assert_close(to_np(env.consecutive_successes), cons)    # from the successes BEFORE the reset (:424-431)
    assert_close(to_np(env.successes), st["successes"])
    assert_close(to_np(env.goal_rot), st["goal_rot"])
    assert_close(to_np(obs["policy"]), o)
    np.testing.assert_array_equal(to_np(env.reset_buf), done)
    np.testing.assert_array_equal(to_np(env.episode_length_buf), st["episode_length_buf"])
    assert_close(to_np(env.cur_targets), st["cur_targets"])
    assert_close(to_np(env.prev_targets), st["prev_targets"])
    assert_close(to_np(env.hand.joint_pos_target), st["joint_pos_target"])
    assert_close(to_np(env.object.data.root_state_w), st["object_root_state_w"])
    # joint state buffers of the articulation after the two scene updates (articulation_data.py:77-82, 446-455) and the
    # reset's write_joint_state_to_sim
    assert_close(to_np(env.hand.data.joint_pos), st["joint_pos"])
    assert_close(to_np(env.hand.data.joint_vel), st["joint_vel"])
    assert np.abs(to_np(env.hand.data.joint_acc)).max() == 0.0      # the frame is held: no velocity change
    # the physics view sees the reset state of the done envs (set_transforms xyzw / set_velocities / set_dof_*)
    xf = to_np(env.object_view.transforms)[done]
    ro = st["object_root_state_w"][done]
    assert_close(xf, np.concatenate([ro[:, :3], ro[:, 4:7], ro[:, 3:4]], axis=1))
    assert np.abs(to_np(env.object_view.velocities)[done]).max() == 0.0
    assert_close(to_np(env.hand_view.get_dof_positions())[done], st["joint_pos"][done])


@pytest.mark.gpu
def test_gpu_inhand_fused_step_equals_per_statement_step_over_changing_frames(device):
    """Three steps with a NEW physics frame every sub-step (written in place into the view's buffers, the harshest case
    for the launches that carry the previous sub-step's hand update): the four-launch step and the one-launch-per-
    statement step leave every buffer bit-identical -- joint state incl. the joint_acc finite difference, targets, dones,
    episode counters, rewards, successes, their moving average, observations."""
    import torch

    n, D, B, A, steps = 1000, 24, 26, 20, 3
    rng = np.random.RandomState(5)
    f = lambda *s: rng.randn(*s).astype(np.float32)
    origins = f(n, 3) * 3
    lower = -1.0 - rng.rand(n, D).astype(np.float32)
    upper = 1.0 + rng.rand(n, D).astype(np.float32)
    goal = f(n, 4); goal /= np.linalg.norm(goal, axis=1, keepdims=True)
    frames = []
    for _ in range(steps * 2):
        quat = f(n, 4); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
        quat[::7] = goal[::7]
        obj = np.concatenate([f(n, 3) * 0.12 + np.array([0.0, -0.39, 0.6], np.float32) + origins, quat, f(n, 6)], 1)
        frames.append((obj.astype(np.float32), f(n, B, 13), (lower + (upper - lower) * rng.rand(n, D)).astype(np.float32), f(n, D)))
    actions = [(rng.rand(n, A) * 2 - 1).astype(np.float32) for _ in range(steps)]
    prev = f(n, D) * 0.5
    succ0 = rng.randint(0, 5, n).astype(np.float32)
    ep0 = rng.randint(0, 400, n)
    results = []
    for fused in (True, False):
        env = _make_env(n, device, origins, lower, upper, act_moving_average=0.7)
        env.fused_step = fused
        ep0[::50] = env.max_episode_length - 2
        env.goal_rot.copy_(to_dev(goal, device))
        env.prev_targets.copy_(to_dev(prev, device))
        env.successes.copy_(to_dev(succ0, device))
        env.episode_length_buf.copy_(to_dev(ep0, device))
        env.consecutive_successes.fill_(1.3)
        _feed(env, device, *frames[0])
        cursor = [0]

        def advance(env=env, cursor=cursor):
            obj, body, q, qd = frames[cursor[0] % len(frames)]
            cursor[0] += 1
            xyzw = np.concatenate([obj[:, :3], obj[:, 4:7], obj[:, 3:4]], axis=1)
            env.object_view.transforms.copy_(to_dev(xyzw, device))
            env.object_view.velocities.copy_(to_dev(obj[:, 7:], device))
            lt = np.concatenate([body[..., :3], body[..., 4:7], body[..., 3:4]], axis=-1)
            env.hand_view.get_link_transforms().copy_(to_dev(lt, device))
            env.hand_view.get_link_velocities().copy_(to_dev(body[..., 7:], device))
            env.hand_view.get_dof_positions().copy_(to_dev(q, device))
            env.hand_view.get_dof_velocities().copy_(to_dev(qd, device))

        env.hand_view.advance = advance
        out = []
        for t in range(steps):
            obs, rew, term, tout, extras = env.step(to_dev(actions[t], device))
            torch.cuda.synchronize()
            out.append([to_np(x).copy() for x in (
                obs["policy"], rew, term, tout, env.reset_buf, env.episode_length_buf, env.successes,
                env.consecutive_successes, env.reset_goal_buf, env.cur_targets, env.prev_targets, env.hand.joint_pos_target,
                env.actions, env.hand.data.joint_pos, env.hand.data.joint_vel, env.hand.data.joint_acc,
                env.object.data.root_state_w, env.goal_rot)] + [np.float32(float(extras["log"]["consecutive_successes"]))])
        results.append(out)
    names = ("obs", "reward", "terminated", "time_outs", "reset_buf", "episode_length", "successes", "consecutive_successes",
             "reset_goal_buf", "cur_targets", "prev_targets", "joint_pos_target", "actions", "joint_pos", "joint_vel",
             "joint_acc", "object_root_state", "goal_rot", "logged")
    for t in range(steps):
        for name, a, b in zip(names, results[0][t], results[1][t]):
            np.testing.assert_array_equal(a, b, err_msg=f"{name}, step {t}")
    assert np.abs(results[0][-1][15]).max() > 0.0            # joint_acc: the velocities did change between sub-steps
    # the reset path ran in both forms (goal resets inside _get_rewards, masked _reset_idx of the done envs) from the same
    # per-env generator epochs: goals were replaced, nothing is left flagged
    assert results[0][-1][2].sum() > 0 and results[0][-1][8].sum() == 0
    assert np.abs(results[0][-1][17] - goal).max(axis=1).astype(bool).sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("fused", (True, False), ids=("fused_step", "one_launch_per_statement"))
@pytest.mark.parametrize("n", (7, 48))
def test_gpu_inhand_layouts_and_resets_vs_reference_fixture(golden, device, n, fused):
    """compute_reduced_observations (42), compute_full_state (187), the asymmetric _get_observations (both from ONE
    launch), _reset_target_pose and _reset_idx through the C ABI against vectors the reference methods produced from
    recorded draws (tests/golden/inhand_reset.npz)."""
    import torch

    from isaaclab_amd.tasks.inhand import ShadowHandOpenAIEnvCfg

    G = golden("inhand_reset")
    p = f"n{n}_"
    D = 24
    env = _make_env(n, device, G[p + "origins"], G[p + "lower"], G[p + "upper"], None, G[p + "fingers"],
                    cfg_cls=ShadowHandOpenAIEnvCfg, action_noise_model=None, observation_noise_model=None,
                    reset_dof_vel_noise=0.3)
    env.fused_step = fused
    assert env.num_reduced_obs == 42 and env.num_full_state == 187 and env.cfg.asymmetric_obs
    env.in_hand_pos.copy_(to_dev(G[p + "in_hand"], device))
    env.goal_rot.copy_(to_dev(G[p + "goal"], device))
    env.actions.copy_(to_dev(G[p + "actions"], device))
    _feed(env, device, G[p + "obj"], G[p + "body"], G[p + "q"], G[p + "qd"])
    env.hand_view.get_link_incoming_joint_force().copy_(to_dev(G[p + "forces"], device))
    assert_close(to_np(env.compute_reduced_observations()), G[p + "reduced_obs"])
    assert_close(to_np(env.compute_full_state()), G[p + "full_state"])
    both = env._get_observations()
    assert both["policy"].shape == (n, 42) and both["critic"].shape == (n, 187)
    assert_close(to_np(both["policy"]), G[p + "reduced_obs"])
    assert_close(to_np(both["critic"]), G[p + "full_state"])
    # _reset_target_pose(env_ids) with the recorded draws
    env.inject_uniforms(to_dev(G[p + "u"], device))
    env.reset_goal_buf.copy_(to_dev(G[p + "reset_goal_in"], device))
    env._reset_target_pose(torch.from_numpy(G[p + "ids"]).to(device))
    assert_close(to_np(env.goal_rot), G[p + "goal_after_target_pose"])
    np.testing.assert_array_equal(to_np(env.reset_goal_buf), G[p + "reset_goal_after_target_pose"])
    # _reset_idx(env_ids)
    env.goal_rot.copy_(to_dev(G[p + "goal"], device))
    env.reset_goal_buf.copy_(to_dev(G[p + "reset_goal_in"], device))
    env.object.data.default_root_state.copy_(to_dev(G[p + "object_default"], device))
    hd = env.hand.data
    hd.default_joint_pos.copy_(to_dev(G[p + "default_joint_pos"], device))
    hd.default_joint_vel.copy_(to_dev(G[p + "default_joint_vel"], device))
    for buf, key in ((env.prev_targets, "in_prev_targets"), (env.cur_targets, "in_cur_targets"),
                     (env.hand_dof_targets, "in_hand_dof_targets"), (env.hand.joint_pos_target, "in_joint_pos_target"),
                     (hd._previous_joint_vel, "in_prev_joint_vel"), (hd.joint_acc, "in_joint_acc"),
                     (env.successes, "in_successes"), (env.episode_length_buf, "in_ep")):
        buf.copy_(to_dev(G[p + key], device))
    ids2 = G[p + "ids2"]
    env._reset_idx(torch.from_numpy(ids2).to(device))
    got = dict(goal_rot=env.goal_rot, object_root_state_w=env.object.data.root_state_w, prev_targets=env.prev_targets,
               cur_targets=env.cur_targets, hand_dof_targets=env.hand_dof_targets, joint_pos_target=env.hand.joint_pos_target,
               joint_pos=hd.joint_pos, joint_vel=hd.joint_vel, prev_joint_vel=hd._previous_joint_vel, joint_acc=hd.joint_acc,
               successes=env.successes)
    for key, ref in _RESET_OUT:
        assert_close(to_np(got[key]), G[p + ref], msg=f"{key} n={n}")
    np.testing.assert_array_equal(to_np(env.reset_goal_buf), G[p + "r_reset_goal"])
    np.testing.assert_array_equal(to_np(env.episode_length_buf), G[p + "r_ep"])
    # what the reference handed to the physics view for the reset envs
    assert_close(to_np(env.object_view.transforms)[ids2], G[p + "r_sim_transforms"][ids2])
    assert_close(to_np(env.object_view.velocities)[ids2], G[p + "r_sim_velocities"][ids2])
    assert_close(to_np(env.hand_view.get_dof_positions())[ids2], G[p + "r_sim_dof_pos"][ids2])
    assert_close(to_np(env.hand_view.get_dof_velocities())[ids2], G[p + "r_sim_dof_vel"][ids2])
    # rejected arguments: the critic layout without force sensors, injected rows that are too narrow
    with pytest.raises(ValueError):
        env.inject_uniforms(to_dev(G[p + "u"][:, :5].copy(), device))
        env._reset_idx(torch.from_numpy(ids2).to(device))


@pytest.mark.gpu
def test_gpu_inhand_reset_draws_from_the_device_generator_and_noise_hooks(device):
    """Without injected uniforms the masked resets draw from the per-env device generator: unit quaternions, joint
    positions inside the limits, new numbers on every call -- also when the call is replayed from a hipGraph.  The
    DirectRLEnv noise models (direct_rl_env.py:319-320, 376-377) corrupt the action and the policy observation, never
    the critic's state."""
    import torch

    from isaaclab_amd.tasks.inhand import ShadowHandOpenAIEnvCfg

    n, D, B = 512, 24, 26
    rng = np.random.RandomState(3)
    f = lambda *s: rng.randn(*s).astype(np.float32)
    origins = f(n, 3)
    lower = -1.0 - rng.rand(n, D).astype(np.float32)
    upper = 1.0 + rng.rand(n, D).astype(np.float32)
    env = _make_env(n, device, origins, lower, upper, cfg_cls=ShadowHandOpenAIEnvCfg)
    mask = torch.zeros(n, dtype=torch.bool, device=device)
    mask[::3] = True
    g0 = env.goal_rot.clone()
    env._reset_idx_masked(mask)
    g1 = env.goal_rot.clone()
    env._reset_idx_masked(mask)
    g2 = env.goal_rot.clone()
    torch.cuda.synchronize()
    m = to_np(mask)
    assert np.array_equal(to_np(g1)[~m], to_np(g0)[~m])
    assert_close(np.linalg.norm(to_np(g1)[m], axis=1), np.ones(m.sum(), np.float32))
    assert np.abs(to_np(g1)[m] - to_np(g2)[m]).max(axis=1).min() > 0.0          # every flagged env drew new numbers
    qn = to_np(env.hand.data.joint_pos)[m]
    assert (qn >= lower[m]).all() and (qn <= upper[m]).all() and np.abs(qn).max() > 0.0
    # graph replay keeps drawing
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env._reset_target_pose_masked(mask)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        env._reset_target_pose_masked(mask)
    seen = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        seen.append(to_np(env.goal_rot)[m].copy())
    assert np.abs(seen[0] - seen[1]).max() > 0.0 and np.abs(seen[1] - seen[2]).max() > 0.0
    # noise hooks: one step with the OpenAI cfg (noise models on) against the same env state without them
    quat = f(n, 4); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    obj = np.concatenate([f(n, 3) * 0.05 + np.array(env.cfg.object_init_pos, np.float32) + origins, quat, f(n, 6)], 1).astype(np.float32)
    _feed(env, device, obj, f(n, B, 13), (lower + (upper - lower) * rng.rand(n, D)).astype(np.float32), f(n, D))
    env.hand_view.advance = lambda: None
    env.auto_reset = "off"
    env.reset_goals = False
    actions = to_dev((rng.rand(n, 20) * 2 - 1).astype(np.float32), device)
    obs, *_ = env.step(actions)
    noisy_policy, critic = obs["policy"].clone(), obs["critic"].clone()
    assert np.abs(to_np(env.actions) - to_np(actions)).max() > 1e-3              # action noise, std 0.05
    clean = env._get_observations()
    d = to_np(noisy_policy) - to_np(clean["policy"])
    assert 0.0005 < d.std() < 0.01 and np.abs(d).max() < 0.05                    # gaussian std 0.002 + |bias| ~ 1e-4
    np.testing.assert_array_equal(to_np(critic), to_np(clean["critic"]))
