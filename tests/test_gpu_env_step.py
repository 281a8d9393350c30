"""GPU: full MDP env-step (ManagerBasedRLEnv.step order) of the velocity task over a replayed synthetic
physics feed, HIP path vs the CPU oracle on the same seeded frames / actions / injected random numbers.
Bit-exact: terminated / time_outs masks.  fp32: 1e-5 relative; absolute floors where terms cancel: 2e-6 on rewards,
4e-6 on observation columns, 1e-5 on the height-scan columns (conftest.assert_obs_close says where that comes from)."""
import numpy as np
import pytest

from conftest import assert_close, assert_obs_close, to_np

pytestmark = pytest.mark.gpu


def run_env(name, n, steps, device, seed, fuse_cast=None, fused=True, persistent=False, scan_in_cast=None,
            single_launch=True, **kw):
    import torch

    from isaaclab_amd.envs.synthetic import make_velocity_env

    env, rec = make_velocity_env(name, num_envs=n, device=str(device), steps=steps, seed=seed, **kw)
    if fuse_cast is not None:
        env.scene.sensors["height_scanner"].fuse_into_observations = fuse_cast
    if scan_in_cast is not None:
        env.scene.sensors["height_scanner"].scan_in_cast = scan_in_cast
    # persistent observation buffers let the step put the sensor-owned columns and the rest on two streams
    env.observation_manager.persistent_output = persistent
    env.single_launch_post_physics = single_launch
    if not fused:      # the per-entity / per-manager launches of the reference call structure
        env.scene.fused_update = False
        env.fuse_post_physics = False
        env.fuse_substeps = False
        env.fold_action = False
        env.concurrent_post_physics = False
    outs = []
    for t in range(steps):
        rec.feed(t)
        obs, rew, term, trunc, _ = env.step(rec.actions[t].to(device))
        outs.append((to_np(obs["policy"]), to_np(rew), to_np(term), to_np(trunc)))
    torch.cuda.synchronize()
    return env, rec, outs


def compare(outs, ref):
    for t, (got, exp) in enumerate(zip(outs, ref)):
        assert_close(got[2], exp[2], msg=f"terminated step {t}")
        assert_close(got[3], exp[3], msg=f"time_outs step {t}")
        assert_close(got[1], exp[1], rtol=1e-5, atol=2e-6, msg=f"reward step {t}")
        # every task here puts the 187-ray height scan last (flat tasks: 48 columns, no scan)
        assert_obs_close(got[0], exp[0], scan_cols=187 if got[0].shape[1] > 187 else 0, msg=f"obs step {t}")


@pytest.mark.parametrize("n,persistent", ((7, False), (256, False), (64, True), (203, True)))
def test_anymal_d_flat(device, n, persistent):
    """BASELINE config C1 (flat, DC motor, 48-dim observation).  With persistent observation buffers everything after
    the last physics sub-step is ONE launch here too (mdp_post_physics_step without a cast: manager blocks only; 203
    envs: ragged last block)."""
    from oracle.harness import replay_oracle

    env, rec, outs = run_env("anymal_d_flat", n, 6, device, seed=11, persistent=persistent)
    assert outs[0][0].shape == (n, 48)
    assert getattr(env, "_single_launch_steps", 0) == (5 if persistent else 0)
    compare(outs, replay_oracle(rec, 6))


@pytest.mark.parametrize("obs_waves", (0, 1, 4, 6, 11))
def test_flat_post_physics_wave_roles(device, obs_waves, monkeypatch):
    """The manager blocks of a launch without cast blocks are 12 waves wide; `obs_waves` of them write the observations
    while the others evaluate the reward terms (k_post_physics<.., ROLES>; default 6 at this size).  Every split gives
    the oracle's step and the same bits as every other split (0: all waves do every phase, the observation operands
    that lie in the command buffer still come from the block's LDS copy of the new command)."""
    from oracle.harness import replay_oracle

    monkeypatch.setenv("MDP_POST_OBS_WAVES", str(obs_waves))
    env, rec, outs = run_env("anymal_d_flat", 203, 6, device, seed=13, persistent=True)
    assert getattr(env, "_single_launch_steps", 0) == 5
    compare(outs, replay_oracle(rec, 6))
    monkeypatch.delenv("MDP_POST_OBS_WAVES")
    _, _, base = run_env("anymal_d_flat", 203, 6, device, seed=13, persistent=True)
    for a, b in zip(outs, base):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


def test_staged_cast_in_the_post_physics_launch(device, monkeypatch):
    """MDP_RC_STAGE=1: the cast blocks stage the cell records their rays name in LDS (raycaster_update_device.hpp,
    RcStage) instead of one walk per ray -- same records, same order, same arithmetic: the step's outputs and the
    sensor's hit points are bit-identical to the per-ray walk's."""
    runs = []
    for stage in ("1", "0"):
        monkeypatch.setenv("MDP_RC_STAGE", stage)
        env, rec, outs = run_env("anymal_d_rough", 203, 4, device, seed=14, small_terrain=True, persistent=True,
                                 scan_in_cast=True, mesh_accel="grid")
        assert getattr(env, "_single_launch_steps", 0) == 3
        runs.append((outs, to_np(env.scene["height_scanner"].data.ray_hits_w)))
    for a, b in zip(runs[0][0], runs[1][0]):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    assert np.isfinite(runs[0][1]).all()


def test_one_env_cast_blocks_in_the_post_physics_launch(device, monkeypatch):
    """The post-physics launch's cast blocks take the loop-free routine when a block owns one env with one ray per lane
    (`raycaster_update_down_block_one`, instantiation `k_post_physics<.., ONE>`); MDP_POST_CAST_ONE=0 keeps the general
    routine.  Same statements per ray: the step's outputs, the sensor's hit points and its bookkeeping are bit-identical
    (203 envs: a ragged last manager block; update period > 0 so that some steps do not cast)."""
    runs = []
    for one in ("1", "0"):
        monkeypatch.setenv("MDP_POST_CAST_ONE", one)
        env, rec, outs = run_env("anymal_c_rough", 203, 5, device, seed=15, small_terrain=True, persistent=True,
                                 scan_in_cast=True, mesh_accel="grid")
        assert getattr(env, "_single_launch_steps", 0) == 4
        sensor = env.scene["height_scanner"]
        runs.append((outs, to_np(sensor.data.ray_hits_w), to_np(sensor.data.pos_w), to_np(sensor._timestamp)))
    for a, b in zip(runs[0][0], runs[1][0]):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    for k in (1, 2, 3):
        np.testing.assert_array_equal(runs[0][k], runs[1][k])
    assert np.isfinite(runs[0][1]).all()


@pytest.mark.parametrize("n,fuse_cast,fused,persistent,scan_in_cast,single_launch",
                         ((5, False, True, False, True, True), (128, False, True, True, True, False),
                          (128, True, True, False, True, True), (37, False, False, False, True, True),
                          (64, False, True, True, False, True), (128, False, True, True, True, True),
                          (203, False, True, True, True, True)))
def test_anymal_d_rough(device, n, fuse_cast, fused, persistent, scan_in_cast, single_launch):
    """BASELINE config C2 (rough, 187-ray height scan, 235-dim observation): the default launch structure (fused
    scene update, fused termination + reward, vertical-ray cast kernel + lean observation kernel), the height scan
    cast inside the observation kernel, the unfused per-entity / per-manager launches, persistent observation buffers
    (two-stream split of the observation), the height scan read back from the sensor buffers, and everything after
    the last physics sub-step as one launch (mdp_post_physics_step; 203 envs: ragged last 64-env block)."""
    from oracle.harness import replay_oracle

    # the small terrain's flat border triangles make "auto" pick the BVH; the grid (what the full-size terrain gets)
    # is what the vertical-ray kernel and the single post-physics launch need
    accel = "grid" if (persistent and scan_in_cast and fused) or n == 5 else "auto"
    env, rec, outs = run_env("anymal_d_rough", n, 6, device, seed=12, small_terrain=True, fuse_cast=fuse_cast, fused=fused,
                              persistent=persistent, scan_in_cast=scan_in_cast, single_launch=single_launch,
                              mesh_accel=accel)
    assert outs[0][0].shape == (n, 235)
    if persistent and scan_in_cast and fused:
        assert getattr(env, "_single_launch_steps", 0) == (5 if single_launch else 0)
    compare(outs, replay_oracle(rec, 6))
    # sensor state after the run.  The rays themselves come from quat_apply_yaw on device vs host libm (1-ulp
    # differences in the ray origins), so hits agree to fp32 tolerance here; bit-exactness of the caster on
    # identical rays is tests/test_raycast.py.
    from oracle.harness import build_oracle, oracle_step

    orc = build_oracle(rec)
    for k in range(6):
        oracle_step(orc, rec, k)
    hits = to_np(env.scene["height_scanner"].data.ray_hits_w)
    assert_close(hits, orc.ray_hits, rtol=1e-5, atol=1e-5)


def test_episode_sums_and_logs(device):
    from oracle.harness import build_oracle, oracle_step

    env, rec, outs = run_env("anymal_d_flat", 64, 5, device, seed=13)
    orc = build_oracle(rec)
    for k in range(5):
        oracle_step(orc, rec, k)
    rm = env.reward_manager
    sums = np.stack([to_np(rm._episode_sums[name]) for name in rm.active_terms])
    assert_close(sums, orc.episode_sums, rtol=1e-5, atol=2e-6)
    assert_close(to_np(env.termination_manager.get_term("base_contact")), orc.term_dones[1])
    cmd = env.command_manager.get_term("base_velocity")
    assert_close(to_np(cmd.command), orc.command.cmd, rtol=1e-5, atol=1e-6)
    assert_close(to_np(cmd.time_left), orc.command.time_left, rtol=1e-5, atol=1e-6)
    assert_close(to_np(cmd.metrics["error_vel_xy"]), orc.command.m_xy, rtol=1e-5, atol=1e-6)


def test_anymal_c_rough_lstm(device):
    """BASELINE metric configuration: rough terrain + ANYdrive LSTM actuator net (seeded random weights)."""
    from oracle.harness import replay_oracle

    env, rec, outs = run_env("anymal_c_rough", 96, 5, device, seed=14, small_terrain=True, persistent=True)
    assert outs[0][0].shape == (96, 235)
    ref = replay_oracle(rec, 5)
    # the MFMA LSTM sums in a different order than numpy: torques agree to ~1e-5 abs, which the
    # dof_torques_l2 reward (weight 1e-5) damps far below the reward tolerance
    compare(outs, ref)
    act = env.scene["robot"].actuators["legs"]
    assert act.sea_hidden_state.abs().max().item() > 0.0


@pytest.mark.parametrize("n,fused,merged", ((48, True, True), (19, False, True), (33, True, False)))
def test_h1_rough(device, n, fused, merged, monkeypatch):
    """BASELINE config C4 shape (Isaac-Velocity-Rough-H1-v0: 19 joints in three implicit-actuator groups addressed by
    index lists, 20 bodies, biped reward terms incl. feet_slide on body velocities, 256-dim observation).  The three
    groups run as ONE launch over the full (N, J) buffers (Articulation.merge_actuator_groups, the default) or as one
    launch + gathers / scatters per group; the groups' own efforts read through to the merged buffers."""
    from isaaclab_amd.assets.articulation import Articulation
    from oracle.harness import build_oracle, replay_oracle

    monkeypatch.setattr(Articulation, "merge_actuator_groups", merged)
    env, rec, outs = run_env("h1_rough", n, 5, device, seed=15, small_terrain=True, fused=fused)
    assert outs[0][0].shape == (n, 256)
    orc = build_oracle(rec)
    assert env.reward_manager.active_terms == [name for name, _, _ in orc.s["rewards"]]
    robot = env.scene["robot"]
    assert len(robot.actuators) == 3
    assert (robot._merged_actuator is not None) == merged
    compare(outs, replay_oracle(rec, 5))
    for act in robot.actuators.values():
        ji = act.joint_indices
        np.testing.assert_array_equal(to_np(act.applied_effort), to_np(robot.data.applied_torque[:, ji]))
        np.testing.assert_array_equal(to_np(act.computed_effort), to_np(robot.data.computed_torque[:, ji]))
    assert float(robot.data.computed_torque.abs().max()) > 0.0
    if merged:
        # a group's gains edited in place after construction (the reference's randomize_actuator_gains pattern) reach the
        # merged launch by themselves: the next actuator evaluation uses them
        import torch

        robot.write_data_to_sim()                     # same joint state and targets as the evaluation after the edit
        before = robot.data.computed_torque.clone()
        legs = robot.actuators["legs"]
        ji = legs.joint_indices
        legs.stiffness[:] = legs.stiffness * 2.0
        legs.damping[:] = 0.0
        robot.write_data_to_sim()
        torch.cuda.synchronize()
        d = robot.data
        exp = legs.stiffness * (d.joint_pos_target[:, ji] - d.joint_pos[:, ji])
        assert_close(to_np(d.computed_torque[:, ji]), to_np(exp), rtol=1e-5, atol=1e-5, msg="edited gains of a merged group")
        other = [k for k in range(robot.num_joints) if k not in set(ji.tolist())]
        assert float((d.computed_torque[:, other] - before[:, other]).abs().max()) == 0.0


def test_post_physics_step_rejects_what_it_does_not_cover(device):
    """mdp_post_physics_step only takes casts that qualify for the vertical-ray kernel (grid-accelerated mesh): a BVH
    mesh comes back as MDP_ERR_UNSUPPORTED -> NotImplementedError, and the env keeps the separate launches; missing
    programs are a ValueError."""
    import ctypes as C

    from isaaclab_amd import _lib as L

    env, rec, outs = run_env("anymal_d_rough", 16, 3, device, seed=21, small_terrain=True, persistent=True, mesh_accel="bvh")
    assert env.mesh.info()["accel"] == "bvh" and getattr(env, "_single_launch_steps", 0) == 0
    env._mesh_has_grid = lambda mesh: True          # skip the host-side check: the C entry point must refuse itself
    with pytest.raises(NotImplementedError, match="vertical-ray"):
        env._post_physics_single_launch()
    tr = L.MdpTerminationRewardArgs()
    with pytest.raises(ValueError, match="programs required"):
        L.check(L.lib.mdp_post_physics_step(16, None, 0.0, C.byref(tr), None, None, None, L.stream(device)))
    with pytest.raises(ValueError, match="null reset arguments"):
        L.check(L.lib.mdp_post_physics_step_with_reset(16, None, 0.0, C.byref(tr), None, None, None, None, L.stream(device)))
    # the flagged recast covers the same casts as the launch it follows
    sensor = env.scene["height_scanner"]
    cast = sensor.fill_post_physics(None, env.physics_dt)
    with pytest.raises(NotImplementedError, match="vertical-ray"):
        L.check(L.lib.mdp_raycaster_update_flagged(16, C.byref(cast), None, L.stream(device)))


@pytest.mark.parametrize("name,mode,single_launch",
                         (("anymal_d_flat", True, False), ("anymal_d_flat", "masked", False),
                          ("anymal_c_rough", True, False), ("anymal_c_rough", "masked", False),
                          ("anymal_c_rough", "masked", True), ("anymal_d_rough", "masked", True)))
def test_auto_reset_inside_the_step(device, name, mode, single_launch):
    """ManagerBasedRLEnv.step with its reset branch (manager_based_rl_env.py:214-229, 346-391; events and curriculum
    are physics-side and out of scope): envs start close to their time limit, so every step resets some of them --
    LSTM states, contact-sensor and ray-caster bookkeeping, action buffers, episode sums, command metrics / counter /
    resampling, episode counter.  ``True`` is the reference's id-list path (nonzero() on the host), ``"masked"`` the
    sync-free one (mdp_masked_fill + masked command kernel); both must reproduce the oracle, statistics included.
    ``single_launch``: the reset branch runs inside mdp_post_physics_step_with_reset (statistics accumulated by the
    kernel), followed by the ray caster's own reset + recast of the flagged envs."""
    import torch

    from isaaclab_amd.envs.synthetic import make_velocity_env
    from oracle.harness import replay_oracle

    n, steps = (203 if name == "anymal_d_rough" else 96), 8      # 203: ragged last manager block
    kw = dict(small_terrain=True) if "rough" in name else {}
    if single_launch:
        kw["mesh_accel"] = "grid"
    env, rec = make_velocity_env(name, num_envs=n, device=str(device), steps=steps, seed=31, **kw)
    env.auto_reset = mode
    env.observation_manager.persistent_output = single_launch
    g = torch.Generator().manual_seed(5)
    start = env.max_episode_length - torch.randint(1, steps + 3, (n,), generator=g)
    env.episode_length_buf.copy_(start)
    outs, logs = [], []
    for t in range(steps):
        rec.feed(t)
        env.extras.pop("log", None)
        obs, rew, term, trunc, extras = env.step(rec.actions[t].to(device))
        outs.append((to_np(obs["policy"]), to_np(rew), to_np(term), to_np(trunc)))
        logs.append({k: float(v) for k, v in extras["log"].items()} if "log" in extras else None)
    ref_logs = []
    ref = replay_oracle(rec, steps, auto_reset=True, logs=ref_logs, episode_length0=start.numpy())
    compare(outs, ref)
    assert sum(o[3].sum() for o in outs) >= n // 2          # the time limit did fire
    checked = 0
    for t in range(steps):
        if ref_logs[t] is None:
            continue                                        # no env done: the reference does not log
        assert logs[t] is not None
        for key, exp in ref_logs[t].items():
            assert key in logs[t], key
            np.testing.assert_allclose(logs[t][key], exp, rtol=1e-4, atol=1e-6, err_msg=f"{key} step {t}")
            checked += 1
    assert checked > 0
    assert getattr(env, "_single_launch_steps", 0) == (steps - 1 if single_launch else 0)
    if "rough" in name:
        # ray-caster bookkeeping after the run (reset envs: timestamps back to 0, recast in the same step)
        from oracle.harness import build_oracle, oracle_step

        orc = build_oracle(rec)
        orc.episode_length[:] = start.numpy()
        for k in range(steps):
            oracle_step(orc, rec, k, auto_reset=True)
        sensor = env.scene["height_scanner"]
        assert_close(to_np(sensor._timestamp), orc.rc_ts, rtol=1e-6, atol=1e-7)
        assert_close(to_np(sensor._timestamp_last_update), orc.rc_last, rtol=1e-6, atol=1e-7)
        assert_close(to_np(sensor._is_outdated).astype(bool), orc.rc_outdated)
        assert_close(to_np(sensor.data.ray_hits_w), orc.ray_hits, rtol=1e-5, atol=1e-5)


def test_raycaster_update_flagged(device):
    """mdp_raycaster_update_flagged = RayCaster.reset(mask) + the lazy recast: flagged envs get zeroed timestamps and are
    cast at the current pose, outdated envs are cast without a reset, every other env keeps hits, pose and timestamps
    (the sensor pose moved in between, so a cast that should not have happened would show)."""
    import ctypes as C

    import torch

    from isaaclab_amd import _lib as L

    n = 40
    env, rec, outs = run_env("anymal_d_rough", n, 3, device, seed=23, small_terrain=True, persistent=True, mesh_accel="grid")
    sensor = env.scene["height_scanner"]
    torch.cuda.synchronize()
    hits0, ts0, last0 = sensor.data.ray_hits_w.clone(), sensor._timestamp.clone(), sensor._timestamp_last_update.clone()
    assert not sensor._is_outdated.any() and (ts0 > 0).all()
    pose = sensor._view.get_root_transforms()
    pose[:, 0] += 0.37                                     # the robots moved since the last cast
    mask = torch.zeros(n, dtype=torch.uint8, device=device)
    mask[[3, 17, 39]] = 1
    sensor._is_outdated[8] = True                          # outdated for another reason: cast, but not reset
    cast = sensor.fill_post_physics(None, env.physics_dt)
    L.check(L.lib.mdp_raycaster_update_flagged(n, C.byref(cast), mask.data_ptr(), L.stream(device)))
    torch.cuda.synchronize()
    touched = torch.zeros(n, dtype=torch.bool, device=device)
    touched[[3, 17, 39, 8]] = True
    assert not sensor._is_outdated.any()
    assert torch.equal(sensor.data.ray_hits_w[~touched], hits0[~touched])
    assert torch.equal(sensor._timestamp[~touched], ts0[~touched]) and torch.equal(sensor._timestamp_last_update[~touched], last0[~touched])
    assert (sensor._timestamp[[3, 17, 39]] == 0).all() and (sensor._timestamp_last_update[[3, 17, 39]] == 0).all()
    assert sensor._timestamp[8] == ts0[8] and sensor._timestamp_last_update[8] == ts0[8]
    moved = (sensor.data.ray_hits_w[touched][..., 0] - hits0[touched][..., 0])
    assert torch.allclose(moved[torch.isfinite(moved)], torch.full_like(moved[torch.isfinite(moved)], 0.37), atol=1e-4)
    # the same envs through the regular update (all flagged outdated): identical hits for the four, new hits for the rest
    got = sensor.data.ray_hits_w.clone()
    sensor._is_outdated.fill_(True)
    sensor._update_outdated_buffers(from_physics=True)
    torch.cuda.synchronize()
    a, b = got[touched], sensor.data.ray_hits_w[touched]
    assert torch.equal(torch.nan_to_num(a, posinf=1e30), torch.nan_to_num(b, posinf=1e30))
    assert not torch.equal(sensor.data.ray_hits_w[~touched], hits0[~touched])


def test_flagged_recast_draws_its_own_noise(device):
    """The recast of the reset envs and the per-step cast read the same per-block epoch words; on ONE generator stream the
    noise a reset env got in its recast came back sample for sample in its next per-step cast.  Static pose, device
    generator on: the height-scan columns written by the recast must differ from those of the next step's cast (and
    two per-step casts from each other)."""
    import ctypes as C

    import torch

    from isaaclab_amd import _lib as L
    from isaaclab_amd.envs.synthetic import make_velocity_env

    n = 48
    env, rec = make_velocity_env("anymal_d_rough", num_envs=n, device=str(device), steps=2, seed=29, small_terrain=True,
                                 inject_random=False, mesh_accel="grid")
    env.observation_manager.persistent_output = True
    env.step(rec.actions[0].to(device))
    env.step(rec.actions[1].to(device))
    oa = env.observation_manager.post_physics_args()
    assert oa is not None
    _, sensor, scan, _, _, out = oa
    lo, R = scan.col_offset, sensor.num_rays
    cols = lambda: out[:, lo:lo + R].clone()

    def cast():
        sensor._is_outdated.fill_(True)
        sensor._update_outdated_buffers(scan=scan, from_physics=True)
        return cols()

    a = cast()
    mask = torch.ones(n, dtype=torch.uint8, device=device)
    args = sensor.fill_post_physics(scan, 0.0)
    L.check(L.lib.mdp_raycaster_update_flagged(n, C.byref(args), mask.data_ptr(), L.stream(device)))
    f = cols()
    b = cast()
    c = cast()
    torch.cuda.synchronize()
    free = (a.abs() < 0.99) & (f.abs() < 0.99) & (b.abs() < 0.99) & (c.abs() < 0.99)   # not pinned by the clip
    assert free.float().mean() > 0.5
    for x, y in ((f, b), (f, c), (f, a), (a, b), (b, c)):
        assert ((x != y) & free).sum() > 0.98 * free.sum()
        assert (x - y)[free].abs().max() <= 0.2 + 1e-6          # same hits, noise of +-0.1 each


@pytest.mark.parametrize("name,mode,folded", (("anymal_d_flat", True, False), ("anymal_d_flat", "masked", False),
                                              ("anymal_c_rough", "masked", False), ("anymal_d_rough", True, False),
                                              ("anymal_c_rough", "masked", True), ("anymal_d_rough", "masked", True)))
def test_reset_and_interval_events(device, name, mode, folded):
    """The step with the task's event terms (velocity_env_cfg.py:187-218): reset_root_state_uniform and
    reset_joints_by_scale for the envs that are done, push_by_setting_velocity when an env's interval timer fires.
    The events write robot state (root pose / velocity, joint state, zeroed joint_acc) that the command update, the
    height scan and the observations of the SAME step read -- everything must follow the oracle's restatement, on
    the id-list path and on the sync-free masked path (device masks, mdp_event_* kernels).  ``folded``: curriculum and
    event terms run INSIDE the reset branch of the post-physics launch (mdp_post_physics_step_with_reset_events: one
    launch for everything after the last physics sub-step but the flagged recast and the gated actuator step)."""
    import torch

    from isaaclab_amd.envs.synthetic import make_velocity_env
    from oracle.harness import build_oracle, oracle_step

    n, steps = 64, 8
    kw = dict(small_terrain=True) if "rough" in name else {}
    if folded:
        kw["mesh_accel"] = "grid"
    env, rec = make_velocity_env(name, num_envs=n, device=str(device), steps=steps, seed=41, events=True, **kw)
    env.auto_reset = mode
    env.observation_manager.persistent_output = folded
    g = torch.Generator().manual_seed(6)
    start = env.max_episode_length - torch.randint(1, steps + 3, (n,), generator=g)
    env.episode_length_buf.copy_(start)
    orc = build_oracle(rec)
    orc.episode_length[:] = start.numpy()
    robot = env.scene["robot"]
    rough = "rough" in name
    if rough:
        # terrain curriculum (velocity_env_cfg.py:269-272): a third of the envs start on another terrain level than the
        # one their recorded positions belong to -- they have "walked" a tile length and move up (or, beyond the last
        # level, to a random one); the others move down or stay
        st = env.scene.terrain.device_state(device)
        sel = torch.arange(0, n, 3, device=device)
        rows = st["terrain_origins"].shape[0]
        st["terrain_levels"][sel] = (st["terrain_levels"][sel] + 1) % rows
        env.scene.env_origins[sel] = st["terrain_origins"][st["terrain_levels"][sel], st["terrain_types"][sel]]
        orc.terrain_levels[:] = to_np(st["terrain_levels"])
        orc.s["env_origins"][:] = to_np(env.scene.env_origins)
        levels0 = orc.terrain_levels.copy()
    pushed = 0
    for t in range(steps):
        rec.feed(t)
        env.extras.pop("log", None)
        obs, rew, term, trunc, extras = env.step(rec.actions[t].to(device))
        timers_before = orc.push_time_left.copy()
        exp = oracle_step(orc, rec, t, auto_reset=True)
        pushed += int((timers_before - np.float32(env.step_dt) < 1e-6).sum())
        compare([(to_np(obs["policy"]), to_np(rew), to_np(term), to_np(trunc))], [exp])
        if orc.log is not None:
            for key, val in orc.log.items():
                np.testing.assert_allclose(float(extras["log"][key]), val, rtol=1e-4, atol=1e-6, err_msg=f"{key} step {t}")
        if rough:
            assert_close(to_np(st["terrain_levels"]), orc.terrain_levels, msg=f"terrain levels step {t}")
            assert_close(to_np(env.scene.env_origins), orc.s["env_origins"], msg=f"env origins step {t}")
        # the data buffers the events wrote (what the next step's actuator and terms read)
        assert_close(to_np(robot.data.joint_pos), orc._last_q, rtol=1e-6, atol=1e-6, msg=f"joint_pos step {t}")
        assert_close(to_np(robot.data.joint_vel), orc._last_qd, rtol=1e-6, atol=1e-6, msg=f"joint_vel step {t}")
        assert_close(to_np(robot.data._previous_joint_vel), orc.prev_qd, rtol=1e-6, atol=1e-6)
        assert_close(to_np(env.event_manager._interval_term_time_left[0]), orc.push_time_left, rtol=1e-6, atol=1e-6)
    assert getattr(env, "_single_launch_steps", 0) == (steps - 1 if folded else 0)
    assert pushed >= n // 2                                 # the interval timers did fire
    last = rec.field("dof_positions", steps * rec.decimation)
    assert np.abs(orc._last_q - last).max() > 0.05           # ... and the last step's resets left their joint state
    if rough:
        moved = orc.terrain_levels - levels0
        assert (moved > 0).any() and (moved < 0).any()        # the curriculum moved envs both ways


def test_event_terms_with_device_generator(device):
    """The event kernels with their own generator (per-env Philox epochs): samples stay inside the ranges, differ
    between envs and between calls, and envs outside the mask are not touched (ids and masks select the same envs)."""
    import torch

    from isaaclab_amd.envs import mdp
    from isaaclab_amd.envs.synthetic import make_velocity_env

    n = 256
    env, rec = make_velocity_env("anymal_d_flat", num_envs=n, device=str(device), steps=2, seed=43, events=True,
                                 inject_random=False)
    rec.feed(0)
    env.step(rec.actions[0].to(device))
    robot = env.scene["robot"]
    d = robot.data
    before = d.root_state_w.clone()
    ids = torch.arange(0, n, 3, device=device)
    mask = torch.zeros(n, dtype=torch.bool, device=device)
    mask[ids] = True
    rng = dict(pose_range={"x": (-0.5, 0.5), "yaw": (-1.0, 1.0)}, velocity_range={"z": (0.1, 0.2)})
    mdp.reset_root_state_uniform(env, ids, **rng)
    first = d.root_state_w.clone()
    assert torch.equal(first[~mask], before[~mask])
    default = d.default_root_state
    dx = first[mask, 0] - (default[mask, 0] + env.scene.env_origins[mask, 0])
    assert dx.min() >= -0.5 - 1e-5 and dx.max() <= 0.5 + 1e-5 and dx.std() > 0.2
    assert torch.allclose(first[mask, 1], default[mask, 1] + env.scene.env_origins[mask, 1])
    vz = first[mask, 9] - default[mask, 9]
    assert vz.min() >= 0.1 - 1e-6 and vz.max() <= 0.2 + 1e-6
    assert torch.allclose(first[mask, 3:7].norm(dim=1), torch.ones(int(mask.sum()), device=device), atol=1e-5)
    view = robot.root_physx_view
    assert torch.equal(view.get_root_transforms()[mask][:, [6, 3, 4, 5]], first[mask, 3:7])       # xyzw in the sim
    assert torch.equal(view.get_root_velocities()[mask], first[mask, 7:13])
    mdp.reset_root_state_uniform(env, mask, **rng)                                                # mask form, new numbers
    second = d.root_state_w.clone()
    assert torch.equal(second[~mask], before[~mask]) and (second[mask, 0] != first[mask, 0]).float().mean() > 0.99
    # joints: by offset, clamped to the soft limits; every env (env_ids=None)
    mdp.reset_joints_by_offset(env, None, position_range=(-0.2, 0.2), velocity_range=(-0.1, 0.1))
    off = d.joint_pos - d.default_joint_pos
    assert off.min() >= -0.2 - 1e-6 and off.max() <= 0.2 + 1e-6 and off.std() > 0.08
    assert torch.equal(d._previous_joint_vel, d.joint_vel) and (d.joint_acc == 0).all()
    assert torch.equal(view.get_dof_positions(), d.joint_pos)
    # interval timers: global timer fires for all envs at once
    from isaaclab_amd.managers import EventManager, EventTermCfg

    calls = []
    cfg = {"tick": EventTermCfg(func=lambda env, env_ids: calls.append(env_ids.clone()), mode="interval",
                                interval_range_s=(0.05, 0.05), is_global_time=True)}
    em = EventManager(cfg, env)
    for _ in range(5):
        em.apply("interval", dt=0.02)
    fired = [int(c.sum().item()) for c in calls]
    assert fired == [0, 0, n, 0, 0], fired


@pytest.mark.parametrize("mode", (False, "masked"))
def test_normalised_observation_written_by_the_producing_kernels(device, mode):
    """mdp_obs_normalize_t: on the single-launch path the manager blocks and the cast blocks also store
    the standardised, clipped observation into a slot of a rollout storage -- bit-identical to mdp_scaler_apply on the
    finished observation, within 1e-5 of the oracle's restatement of RunningStandardScaler; step 0 (separate launches)
    goes through mdp_scaler_apply.  With resets the flagged recast rewrites the columns of the envs it casts again."""
    import torch

    from isaaclab_amd.envs.synthetic import make_velocity_env
    from isaaclab_amd.rl import RunningStandardScaler
    from oracle.scaler_np import RunningStandardScalerNP

    n, steps = 80, 5
    env, rec = make_velocity_env("anymal_c_rough", num_envs=n, device=str(device), steps=steps, seed=61, small_terrain=True,
                                 mesh_accel="grid")
    env.observation_manager.persistent_output = True
    env.auto_reset = mode
    if mode:
        g = torch.Generator().manual_seed(7)
        env.episode_length_buf.copy_(env.max_episode_length - torch.randint(1, steps + 2, (n,), generator=g))
    sc, orc = RunningStandardScaler(235, device=device), RunningStandardScalerNP(235)
    warm = (np.random.RandomState(1).normal(size=(300, 235)) * 0.3 + 0.1).astype(np.float32)
    sc(torch.from_numpy(warm).to(device), train=True)
    orc.compute(warm, train=True)
    storage = torch.full((steps, n, 235 + 5), 7.0, device=device)
    for t in range(steps):
        rec.feed(t)
        env.set_obs_normalizer(sc, out=storage[t, :, :235])
        obs, *_ = env.step(rec.actions[t].to(device))
        raw = obs["policy"]
        assert torch.equal(storage[t, :, :235], sc(raw)), f"step {t}"
        assert_close(to_np(storage[t, :, :235]), orc.compute(to_np(raw)), rtol=1e-5, atol=1e-6, msg=f"step {t}")
    assert getattr(env, "_single_launch_steps", 0) == steps - 1
    assert (storage[..., 235:] == 7.0).all() and float(storage[..., :235].abs().max()) <= 5.0


@pytest.mark.parametrize("single_launch", (True, False))
def test_reset_log_ring(device, single_launch):
    """``env.reset_log_ring = R``: the masked reset publishes extras["log"] as views of a ring of R buffers instead of a
    fresh copy per step -- same values as the default, and a step's log stays intact for R - 1 further steps."""
    import torch

    from isaaclab_amd.envs.synthetic import make_velocity_env

    n, steps, R = 96, 8, 4
    runs = []
    for ring in (0, R):
        env, rec = make_velocity_env("anymal_c_rough", num_envs=n, device=str(device), steps=steps, seed=31,
                                     small_terrain=True, mesh_accel="grid")
        env.auto_reset = "masked"
        env.reset_log_ring = ring
        env.observation_manager.persistent_output = single_launch
        g = torch.Generator().manual_seed(5)
        env.episode_length_buf.copy_(env.max_episode_length - torch.randint(1, steps + 3, (n,), generator=g))
        held, at_step = [], []
        for t in range(steps):
            rec.feed(t)
            env.step(rec.actions[t].to(device))
            held.append(env.extras["log"])                                  # kept as the caller would keep it
            at_step.append({k: float(v) for k, v in env.extras["log"].items()})
            if ring and t >= R - 1:                                         # R - 1 steps later the views still hold
                old = {k: float(v) for k, v in held[t - (R - 1)].items()}
                assert old == at_step[t - (R - 1)]
        runs.append(at_step)
    assert runs[0] == runs[1]
    assert any(v != 0.0 for step in runs[0] for v in step.values())


def test_episode_statistics_in_one_launch_equal_the_per_family_launches(device):
    """EpisodeStats.fill: all families through mdp_stats_masked_sums_multi (one launch) against one
    mdp_stats_masked_sums_typed launch per family, with the reset mask and without one; bit-identical payloads."""
    import torch

    from isaaclab_amd import _lib as L
    from isaaclab_amd.distributed import EpisodeStats
    from isaaclab_amd.envs.synthetic import make_velocity_env

    env, rec = make_velocity_env("anymal_c_rough", num_envs=192, device=str(device), steps=3, seed=4, small_terrain=True)
    for t in range(3):
        rec.feed(t)
        env.step(rec.actions[t].to(device))
    one, per_family = EpisodeStats(env), EpisodeStats(env)
    one._separate, per_family._separate = False, True
    assert len(one.names) >= 10 and any(n.startswith("Episode_Termination/") for n in one.names)
    g = torch.Generator().manual_seed(2)
    mask = (torch.rand(192, generator=g) < 0.4).to(device)
    for m in (mask, None):
        a = torch.full_like(one.payload, float("nan"))
        b = torch.full_like(one.payload, float("nan"))
        one.fill(a, m, L.stream())
        per_family.fill(b, m, L.stream())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(to_np(a), to_np(b))
        counts = to_np(a).reshape(-1, 2)[:, 1]
        expect = int(mask.sum()) if m is not None else 192
        assert np.all(counts[:-1] == expect) and counts[-1] == 192     # Curriculum/terrain_levels: over ALL envs
