"""CPU: the oracle's restatement of the event terms (oracle/events_np.py) against plain torch fp32 evaluations of the
reference expressions (LAB/envs/mdp/events.py:610-866, math.py:252-278, 1331) on the same uniforms, and the interval
timer semantics of event_manager.py:203-225.  (The reference's own tests hold no golden vectors for these terms: the
random draws are "parity unpinned", the arithmetic on given uniforms is pinned here.)"""
import numpy as np
import torch

from oracle import events_np as EV


def _torch_sample(u, lo, hi):
    return u * (hi - lo) + lo                      # sample_uniform, math.py:1331


def test_reset_root_state_uniform_matches_torch_expression():
    rng = np.random.RandomState(0)
    n = 97
    d = rng.normal(size=(n, 13)).astype(np.float32)
    d[:, 3:7] /= np.linalg.norm(d[:, 3:7], axis=1, keepdims=True)
    org = rng.normal(scale=10.0, size=(n, 3)).astype(np.float32)
    u = rng.uniform(size=(n, 12)).astype(np.float32)
    pose_range = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "yaw": (-3.14, 3.14), "roll": (-0.2, 0.1)}
    vel_range = {k: (-0.5, 0.5) for k in EV.KEYS}
    ids = np.arange(0, n, 2)
    pose, vel = EV.reset_root_state_uniform(d, org, ids, pose_range, vel_range, u)
    # the reference's expression tree in torch fp32
    td, torg, tu = torch.from_numpy(d[ids]), torch.from_numpy(org[ids]), torch.from_numpy(u[ids])
    r = torch.tensor([pose_range.get(k, (0.0, 0.0)) for k in EV.KEYS])
    s = _torch_sample(tu[:, :6], r[:, 0], r[:, 1])
    pos = td[:, 0:3] + torg + s[:, 0:3]
    roll, pitch, yaw = s[:, 3], s[:, 4], s[:, 5]
    cy, sy, cr, sr, cp, sp = (torch.cos(yaw * 0.5), torch.sin(yaw * 0.5), torch.cos(roll * 0.5), torch.sin(roll * 0.5),
                              torch.cos(pitch * 0.5), torch.sin(pitch * 0.5))
    q2 = torch.stack([cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp,
                      sy * cr * cp - cy * sr * sp], dim=-1)
    q1 = td[:, 3:7]
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    ww, yy, zz = (z1 + x1) * (x2 + y2), (w1 - y1) * (w2 + z2), (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    quat = torch.stack([qq - ww + (z1 - y1) * (y2 - z2), qq - xx + (x1 + w1) * (x2 + w2), qq - yy + (w1 - x1) * (y2 + z2),
                        qq - zz + (z1 + y1) * (w2 - x2)], dim=-1)
    rv = torch.tensor([vel_range.get(k, (0.0, 0.0)) for k in EV.KEYS])
    v = td[:, 7:13] + _torch_sample(tu[:, 6:], rv[:, 0], rv[:, 1])
    np.testing.assert_allclose(pose[:, :3], pos.numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(pose[:, 3:], quat.numpy(), rtol=1e-6, atol=1e-7)      # libm vs torch cos / sin
    np.testing.assert_allclose(vel, v.numpy(), rtol=0, atol=0)


def test_reset_joints_and_push_match_torch_expression():
    rng = np.random.RandomState(1)
    n, j = 33, 12
    dq = rng.normal(size=(n, j)).astype(np.float32)
    dqd = rng.normal(size=(n, j)).astype(np.float32)
    lim = np.stack([np.full((n, j), -0.7, np.float32), np.full((n, j), 0.9, np.float32)], axis=-1)
    vl = np.full((n, j), 0.5, np.float32)
    u = rng.uniform(size=(n, 2 * j)).astype(np.float32)
    ids = np.array([0, 5, 32])
    for by_offset in (False, True):
        q, qd = EV.reset_joints(dq, dqd, lim, vl, ids, (0.5, 1.5), (-0.3, 0.4), u, by_offset=by_offset)
        tu = torch.from_numpy(u[ids])
        sp, sv = _torch_sample(tu[:, :j], 0.5, 1.5), _torch_sample(tu[:, j:], -0.3, 0.4)
        tq, tqd = torch.from_numpy(dq[ids]).clone(), torch.from_numpy(dqd[ids]).clone()
        tq, tqd = (tq + sp, tqd + sv) if by_offset else (tq * sp, tqd * sv)
        tl = torch.from_numpy(lim[ids])
        tq = tq.clamp_(tl[..., 0], tl[..., 1])
        tv = torch.from_numpy(vl[ids])
        tqd = tqd.clamp_(-tv, tv)
        np.testing.assert_array_equal(q, tq.numpy())
        np.testing.assert_array_equal(qd, tqd.numpy())
    vel = rng.normal(size=(n, 6)).astype(np.float32)
    u6 = rng.uniform(size=(n, 6)).astype(np.float32)
    out = EV.push_by_setting_velocity(vel, ids, {"x": (-0.5, 0.5), "y": (-0.5, 0.5)}, u6)
    r = torch.tensor([{"x": (-0.5, 0.5), "y": (-0.5, 0.5)}.get(k, (0.0, 0.0)) for k in EV.KEYS])
    exp = torch.from_numpy(vel[ids]) + _torch_sample(torch.from_numpy(u6[ids]), r[:, 0], r[:, 1])
    np.testing.assert_array_equal(out, exp.numpy())
    assert np.array_equal(out[:, 2:], vel[ids][:, 2:])           # axes without a range keep their velocity


def test_interval_timers():
    tl = np.array([0.05, 0.019, 0.02, 1.0], np.float32)
    u = np.array([0.0, 0.5, 1.0 - 2 ** -24, 0.3], np.float32)
    fired = EV.interval_tick(tl, 0.02, (10.0, 15.0), u)
    assert fired.tolist() == [1, 2]                              # 0.019 - 0.02 < 1e-6 and 0.02 - 0.02 < 1e-6
    np.testing.assert_allclose(tl, [0.03, 12.5, 15.0, 0.98], rtol=1e-6)
    assert EV.interval_tick(tl, 0.02, (10.0, 15.0), u).size == 0


def test_terrain_levels_vel_matches_torch_expression():
    """curriculums.py:26-55 + terrain_importer.py:307-324 evaluated with torch on the same inputs."""
    rng = np.random.RandomState(2)
    n, rows, cols = 60, 4, 5
    origins = rng.normal(scale=8.0, size=(rows, cols, 3)).astype(np.float32)
    levels = rng.randint(0, rows, size=n).astype(np.int64)
    types = rng.randint(0, cols, size=n).astype(np.int64)
    env_origins = origins[levels, types].copy()
    root = np.zeros((n, 13), np.float32)
    root[:, :2] = env_origins[:, :2] + rng.uniform(-7, 7, size=(n, 2)).astype(np.float32)
    cmd = rng.uniform(-0.6, 0.6, size=(n, 3)).astype(np.float32)
    rand = rng.randint(0, rows, size=n).astype(np.int64)
    ids = np.arange(0, n, 2)
    lv, eo = levels.copy(), env_origins.copy()
    mean = EV.terrain_levels_vel(root, eo, cmd, ids, lv, types, origins, 8.0, 20.0, rand)
    t_ids = torch.from_numpy(ids)
    t_lv, t_eo = torch.from_numpy(levels.copy()), torch.from_numpy(env_origins.copy())
    distance = torch.norm(torch.from_numpy(root)[t_ids, :2] - t_eo[t_ids, :2], dim=1)
    move_up = distance > 8.0 / 2
    move_down = distance < torch.norm(torch.from_numpy(cmd)[t_ids, :2], dim=1) * 20.0 * 0.5
    move_down *= ~move_up
    t_lv[t_ids] += 1 * move_up - 1 * move_down
    t_lv[t_ids] = torch.where(t_lv[t_ids] >= rows, torch.from_numpy(rand)[t_ids], torch.clip(t_lv[t_ids], 0))
    t_eo[t_ids] = torch.from_numpy(origins)[t_lv[t_ids], torch.from_numpy(types)[t_ids]]
    np.testing.assert_array_equal(lv, t_lv.numpy())
    np.testing.assert_array_equal(eo, t_eo.numpy())
    np.testing.assert_allclose(mean, torch.mean(t_lv.float()).item(), rtol=1e-6)
    assert (lv != levels).any() and ((lv - levels) > 0).any() and ((lv - levels) < 0).any()
