"""Allocation-free entry point ``mdp_post_physics_step_with_reset_noalloc`` (no counterpart in the reference; for callers
that cannot hipMalloc): same outputs, bit for bit, as ``mdp_post_physics_step_with_reset`` with a caller workspace -- on
two runs in a row through the C ABI, the second with fewer envs (the static scratch array keeps the first run's
contents), and a size beyond the array's limit is rejected with a clear error."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(n, noalloc, device, steps=6, seed=31):
    import torch

    from isaaclab_amd.envs.synthetic import make_velocity_env

    env, rec = make_velocity_env("anymal_c_rough", num_envs=n, device=str(device), steps=steps, seed=seed,
                                 small_terrain=True, mesh_accel="grid")
    env.auto_reset = "masked"
    env.post_reset_noalloc = noalloc
    env.observation_manager.persistent_output = True
    g = torch.Generator().manual_seed(5)
    # envs start next to their time limit: every step resets some of them
    env.episode_length_buf.copy_(env.max_episode_length - torch.randint(1, steps + 3, (n,), generator=g))
    outs = []
    for t in range(steps):
        rec.feed(t)
        env.extras.pop("log", None)
        obs, rew, term, trunc, extras = env.step(rec.actions[t].to(device))
        log = {k: v.clone() for k, v in extras.get("log", {}).items()}
        outs.append((obs["policy"].clone(), rew.clone(), term.clone(), trunc.clone(), log))
    torch.cuda.synchronize()
    assert env._single_launch_steps == steps - 1, "the post-physics launch with the reset branch did not run"
    resets = sum(int(o[3].sum()) + int(o[2].sum()) for o in outs)
    assert resets > 0
    sensor = env.scene["height_scanner"]
    state = [env.episode_length_buf.clone(), sensor.data.ray_hits_w.clone(), env.reward_manager._episode_sums_block.clone()]
    return outs, state


def test_noalloc_matches_workspace_variant_twice_in_a_row(device):
    import torch

    big, small = 203, 72            # 203: ragged last manager block; the second run is smaller than the first
    ref_big = _run(big, False, device)
    ref_small = _run(small, False, device)
    got_big = _run(big, True, device)          # first run on the static array
    got_small = _run(small, True, device)      # second, smaller: the array still holds the first run's slots
    for (outs, state), (routs, rstate) in ((got_big, ref_big), (got_small, ref_small)):
        for t, (o, r) in enumerate(zip(outs, routs)):
            for a, b, what in zip(o[:4], r[:4], ("obs", "reward", "terminated", "time_outs")):
                assert torch.equal(a, b), f"{what} differs at step {t}"
            assert o[4].keys() == r[4].keys()
            for k in o[4]:
                assert torch.equal(o[4][k], r[4][k]), f"log {k} differs at step {t}"
        for a, b in zip(state, rstate):
            assert torch.equal(a, b)


def test_noalloc_rejects_sizes_beyond_the_static_array(device):
    from isaaclab_amd import _lib as L
    from isaaclab_amd.envs.synthetic import make_velocity_env

    env, rec = make_velocity_env("anymal_d_rough", num_envs=16, device=str(device), steps=2, seed=3,
                                 small_terrain=True, mesh_accel="grid")
    env.auto_reset = "masked"
    env.observation_manager.persistent_output = True
    for t in range(2):                 # builds the programs
        rec.feed(t)
        env.step(rec.actions[t].to(device))
    tm, rm = env.termination_manager, env.reward_manager
    tr = L.MdpTerminationRewardArgs()
    tr.termination, tr.reward = tm.prepare_launch().handle, rm.prepare_launch().handle
    ra = L.MdpPostResetArgs()
    ra.max_episode_length_s = 20.0
    too_many = 8 * 131072              # 131072 manager-block slots x >= 5 rows > the static array
    rc = L.lib.mdp_post_physics_step_with_reset_noalloc(too_many, None, 0.005, C.byref(tr), None, None, None,
                                                        C.byref(ra), L.stream(device))
    assert rc == L.MDP_ERR_INVALID
    assert "static array holds" in L.last_error()
    # a caller workspace is not accepted by this variant
    ra.log_sums = env._post_reset_sums.data_ptr()
    rc = L.lib.mdp_post_physics_step_with_reset_noalloc(16, None, 0.005, C.byref(tr), None, None, None, C.byref(ra),
                                                        L.stream(device))
    assert rc == L.MDP_ERR_INVALID and "log_sums must be NULL" in L.last_error()
