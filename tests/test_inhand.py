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


def _make_env(n, device, origins, lower, upper, act_idx=None, fingers=None, **cfg_over):
    import torch

    from isaaclab_amd.tasks.inhand import InHandManipulationEnv, ShadowHandEnvCfg

    cfg = ShadowHandEnvCfg(num_envs=n, **cfg_over)
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
    o = O.full_observations(q, qd, lower, upper, obj, origins, in_hand, goal, body, np.array(env.finger_bodies), actions, 0.2)
    assert_close(to_np(env.cur_targets), cur)
    np.testing.assert_array_equal(to_np(term), oor)
    np.testing.assert_array_equal(to_np(tout), to)
    assert to.sum() > 0 and oor.sum() > 0 and gr.sum() > 0
    np.testing.assert_array_equal(to_np(env.reset_goal_buf), gr)
    assert_close(to_np(rew), r, msg="reward 16384 envs vs oracle")
    assert_close(to_np(env.successes), succ)
    assert_close(to_np(env.consecutive_successes), cons)
    assert_close(to_np(obs["policy"]), o)
    np.testing.assert_array_equal(to_np(env.reset_buf), oor | to)
    np.testing.assert_array_equal(to_np(env.episode_length_buf), ep)
    assert_close(to_np(env.hand.joint_pos_target)[:, act_idx], cur[:, act_idx])
    # joint state buffers of the articulation after the two scene updates (articulation_data.py:77-82, 446-455)
    assert_close(to_np(env.hand.data.joint_pos), q)
    assert_close(to_np(env.hand.data.joint_vel), qd)
    assert np.abs(to_np(env.hand.data.joint_acc)).max() == 0.0      # the frame is held: no velocity change


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
                env.object.data.root_state_w)] + [np.float32(float(extras["log"]["consecutive_successes"]))])
        results.append(out)
    names = ("obs", "reward", "terminated", "time_outs", "reset_buf", "episode_length", "successes", "consecutive_successes",
             "reset_goal_buf", "cur_targets", "prev_targets", "joint_pos_target", "actions", "joint_pos", "joint_vel",
             "joint_acc", "object_root_state", "logged")
    for t in range(steps):
        for name, a, b in zip(names, results[0][t], results[1][t]):
            np.testing.assert_array_equal(a, b, err_msg=f"{name}, step {t}")
    assert np.abs(results[0][-1][15]).max() > 0.0            # joint_acc: the velocities did change between sub-steps
    assert results[0][-1][2].sum() > 0 and results[0][-1][8].sum() > 0
