"""Golden vectors produced by the REFERENCE code itself (tests/golden/make_golden.py::gen_events: events.py:610-866 with
the real Articulation.write_*_to_sim, EventManager.apply("interval"), terrain_levels_vel + TerrainImporter.
update_env_origins, my_RunningStandardScaler) with injected random draws.  CPU: the oracle against them.  GPU: the HIP
entry points against them (fp32 results within 1e-5 relative / 1e-6 absolute -- the device's sinf / cosf / sqrtf vs
torch's; masks, levels and untouched rows bit-exact)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import assert_close, to_np
from oracle import events_np as EV
from oracle.scaler_np import RunningStandardScalerNP

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "events.npz"))
POSE = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "yaw": (-3.14, 3.14), "roll": (-0.3, 0.2)}
VEL = {k: (-0.5, 0.5) for k in EV.KEYS}
PUSH = {"x": (-0.5, 0.5), "y": (-0.5, 0.5)}
JOINTS = {"scale": ((0.5, 1.5), (0.0, 0.0), False), "offset": ((-0.4, 0.7), (-0.6, 0.5), True)}
IDS = G["ids"].astype(np.int64)


def _xyzw(pose_wxyz):
    return np.concatenate([pose_wxyz[:, :3], pose_wxyz[:, 4:7], pose_wxyz[:, 3:4]], axis=1)


def test_oracle_events_against_reference_vectors():
    n = G["root_state_in"].shape[0]
    rest = np.setdiff1d(np.arange(n), IDS)
    pose, vel = EV.reset_root_state_uniform(G["root_default"], G["env_origins"], IDS, POSE, VEL, G["root_u"])
    np.testing.assert_allclose(pose, G["root_state_out"][IDS, :7], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(vel, G["root_state_out"][IDS, 7:])
    np.testing.assert_array_equal(G["root_state_out"][rest], G["root_state_in"][rest])
    assert (G["root_body_acc_out"][IDS] == 0).all()
    np.testing.assert_array_equal(G["root_sim_transforms"][IDS], _xyzw(G["root_state_out"][IDS, :7]))
    np.testing.assert_array_equal(G["root_sim_velocities"][IDS], G["root_state_out"][IDS, 7:])
    for tag, (prange, vrange, by_offset) in JOINTS.items():
        q, qd = EV.reset_joints(G[f"joints_{tag}_default_pos"], G[f"joints_{tag}_default_vel"], G[f"joints_{tag}_pos_limits"],
                                G[f"joints_{tag}_vel_limits"], IDS, prange, vrange, G[f"joints_{tag}_u"], by_offset=by_offset)
        np.testing.assert_array_equal(q, G[f"joints_{tag}_pos_out"][IDS])
        np.testing.assert_array_equal(qd, G[f"joints_{tag}_vel_out"][IDS])
        np.testing.assert_array_equal(G[f"joints_{tag}_prev_vel_out"][IDS], qd)
        assert (G[f"joints_{tag}_acc_out"][IDS] == 0).all()
        np.testing.assert_array_equal(G[f"joints_{tag}_pos_out"][rest], G[f"joints_{tag}_pos_in"][rest])
    out = EV.push_by_setting_velocity(G["push_state_in"][:, 7:], IDS, PUSH, G["push_u"])
    np.testing.assert_array_equal(out, G["push_state_out"][IDS, 7:])
    tl = G["interval_time_left_in"].copy()
    for k in range(G["interval_u"].shape[0]):
        fired = EV.interval_tick(tl, 0.02, (10.0, 15.0), G["interval_u"][k])
        mask = np.zeros(n, bool)
        mask[fired] = True
        np.testing.assert_array_equal(mask, G["interval_fired"][k])
        np.testing.assert_array_equal(tl, G["interval_time_left_out"][k])
    assert G["interval_fired"].any(axis=1).sum() >= 3
    lv, eo = G["cur_levels_in"].copy(), G["cur_env_origins_in"].copy()
    state = EV.terrain_levels_vel(G["cur_root_state"], eo, G["cur_command"], IDS, lv, G["cur_types"], G["cur_terrain_origins"],
                                  8.0, 20.0, G["cur_random_levels"])
    np.testing.assert_array_equal(lv, G["cur_levels_out"])
    np.testing.assert_array_equal(eo, G["cur_env_origins_out"])
    np.testing.assert_allclose(state, G["cur_state"][0], rtol=1e-6)
    assert (lv > G["cur_levels_in"]).any() and (lv < G["cur_levels_in"]).any()


def test_oracle_scaler_against_reference_vectors():
    orc = RunningStandardScalerNP(G["scaler_x"].shape[2])
    for k in range(G["scaler_x"].shape[0]):
        # the reference reduces the fp32 batch in fp32 (torch.mean / torch.var), the oracle in fp64: outputs agree to that
        np.testing.assert_allclose(orc.compute(G["scaler_x"][k], train=True), G["scaler_y"][k], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(orc.compute(G["scaler_x"][:2], train=True), G["scaler_y3"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(orc.running_mean, G["scaler_mean"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(orc.running_variance, G["scaler_var"], rtol=1e-4)
    assert orc.current_count == G["scaler_count"][0]
    np.testing.assert_allclose(orc.compute(G["scaler_y"][0], inverse=True), G["scaler_inverse"], rtol=1e-4, atol=1e-4)
    # same statistics -> the elementwise map is exact
    orc.running_mean, orc.running_variance = G["scaler_mean"].copy(), G["scaler_var"].copy()
    np.testing.assert_array_equal(orc.compute(G["scaler_y"][0], inverse=True), G["scaler_inverse"])


def _dev(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return t.to(dtype) if dtype is not None else t


def _ranges(rng, lo, scale):
    l, s = EV.dict_range(rng)
    for k in range(6):
        lo[k], scale[k] = float(l[k]), float(s[k])


@pytest.mark.gpu
def test_gpu_event_kernels_against_reference_vectors(device):
    from isaaclab_amd import _lib as L

    n = G["root_state_in"].shape[0]
    mask = torch.zeros(n, dtype=torch.uint8, device=device)
    mask[_dev(IDS, device)] = 1
    st = L.stream(device)
    # reset_root_state_uniform
    a = L.MdpEventRootState()
    keep = dict(d=_dev(G["root_default"], device), o=_dev(G["env_origins"], device), u=_dev(G["root_u"], device),
                r=_dev(G["root_state_in"], device), t=torch.zeros(n, 7, device=device), v=torch.zeros(n, 6, device=device),
                acc=torch.ones(n, 17, 6, device=device))
    a.env_mask, a.default_root_state, a.env_origins, a.injected_u = mask.data_ptr(), keep["d"].data_ptr(), keep["o"].data_ptr(), keep["u"].data_ptr()
    _ranges(POSE, a.pose_lo, a.pose_scale)
    _ranges(VEL, a.vel_lo, a.vel_scale)
    a.root_state_w, a.sim_root_transforms, a.sim_root_velocities = keep["r"].data_ptr(), keep["t"].data_ptr(), keep["v"].data_ptr()
    a.body_acc_w, a.num_bodies = keep["acc"].data_ptr(), 17
    L.check(L.lib.mdp_event_reset_root_state_uniform(n, C.byref(a), st))
    assert_close(to_np(keep["r"]), G["root_state_out"], rtol=1e-5, atol=1e-6)
    rest = np.setdiff1d(np.arange(n), IDS)
    np.testing.assert_array_equal(to_np(keep["r"])[rest], G["root_state_in"][rest])
    assert_close(to_np(keep["t"])[IDS], G["root_sim_transforms"][IDS], rtol=1e-5, atol=1e-6)
    assert_close(to_np(keep["v"])[IDS], G["root_sim_velocities"][IDS], rtol=1e-6, atol=1e-7)
    assert (to_np(keep["acc"])[IDS] == 0).all() and (to_np(keep["acc"])[rest] == 1).all() and (to_np(keep["t"])[rest] == 0).all()
    # joints
    for tag, (prange, vrange, by_offset) in JOINTS.items():
        b = L.MdpEventJoints()
        k = {name: _dev(G[f"joints_{tag}_{name}"], device) for name in
             ("default_pos", "default_vel", "pos_limits", "vel_limits", "u", "pos_in", "vel_in")}
        k.update(prev=torch.full((n, 12), 9.0, device=device), acc=torch.full((n, 12), 9.0, device=device),
                 sq=torch.zeros(n, 12, device=device), sv=torch.zeros(n, 12, device=device))
        b.env_mask, b.num_joints, b.mode = mask.data_ptr(), 12, 1 if by_offset else 0
        b.default_joint_pos, b.default_joint_vel = k["default_pos"].data_ptr(), k["default_vel"].data_ptr()
        plo, psc = EV.tuple_range(prange)
        vlo, vsc = EV.tuple_range(vrange)
        b.pos_lo, b.pos_scale, b.vel_lo, b.vel_scale = float(plo), float(psc), float(vlo), float(vsc)
        b.soft_joint_pos_limits, b.soft_joint_vel_limits = k["pos_limits"].data_ptr(), k["vel_limits"].data_ptr()
        b.injected_u = k["u"].data_ptr()
        b.joint_pos, b.joint_vel, b.prev_joint_vel, b.joint_acc = (k["pos_in"].data_ptr(), k["vel_in"].data_ptr(),
                                                                   k["prev"].data_ptr(), k["acc"].data_ptr())
        b.sim_dof_positions, b.sim_dof_velocities, b.ld_sim = k["sq"].data_ptr(), k["sv"].data_ptr(), 12
        L.check(L.lib.mdp_event_reset_joints(n, C.byref(b), st))
        np.testing.assert_array_equal(to_np(k["pos_in"]), G[f"joints_{tag}_pos_out"])
        np.testing.assert_array_equal(to_np(k["vel_in"]), G[f"joints_{tag}_vel_out"])
        np.testing.assert_array_equal(to_np(k["prev"])[IDS], G[f"joints_{tag}_prev_vel_out"][IDS])
        assert (to_np(k["acc"])[IDS] == 0).all() and (to_np(k["acc"])[rest] == 9).all()
        np.testing.assert_array_equal(to_np(k["sq"])[IDS], G[f"joints_{tag}_sim_pos"][IDS])
        np.testing.assert_array_equal(to_np(k["sv"])[IDS], G[f"joints_{tag}_sim_vel"][IDS])
    # push
    p = L.MdpEventPush()
    k = dict(r=_dev(G["push_state_in"], device), u=_dev(G["push_u"], device), v=torch.zeros(n, 6, device=device))
    p.env_mask, p.injected_u, p.root_state_w, p.sim_root_velocities = mask.data_ptr(), k["u"].data_ptr(), k["r"].data_ptr(), k["v"].data_ptr()
    _ranges(PUSH, p.vel_lo, p.vel_scale)
    L.check(L.lib.mdp_event_push_by_setting_velocity(n, C.byref(p), st))
    np.testing.assert_array_equal(to_np(k["r"]), G["push_state_out"])
    np.testing.assert_array_equal(to_np(k["v"])[IDS], G["push_sim_velocities"][IDS])
    # interval timers
    tl = _dev(G["interval_time_left_in"], device)
    trig = torch.zeros(n, dtype=torch.uint8, device=device)
    lo, sc = EV.tuple_range((10.0, 15.0))
    for i in range(G["interval_u"].shape[0]):
        u = _dev(G["interval_u"][i], device)
        L.check(L.lib.mdp_event_interval_tick(n, tl.data_ptr(), 0.02, float(lo), float(sc), 0, u.data_ptr(), None, trig.data_ptr(), st))
        np.testing.assert_array_equal(to_np(trig).astype(bool), G["interval_fired"][i])
        np.testing.assert_array_equal(to_np(tl), G["interval_time_left_out"][i])
    # terrain curriculum
    c = L.MdpCurriculumTerrain()
    k = dict(r=_dev(G["cur_root_state"], device), cmd=_dev(G["cur_command"], device), lv=_dev(G["cur_levels_in"], device, torch.int64),
             ty=_dev(G["cur_types"], device, torch.int64), to=_dev(G["cur_terrain_origins"], device),
             eo=_dev(G["cur_env_origins_in"], device), rnd=_dev(G["cur_random_levels"], device, torch.int64))
    c.env_mask, c.root_state_w, c.ld_state, c.max_terrain_level = mask.data_ptr(), k["r"].data_ptr(), 13, k["to"].shape[0]
    c.command, c.half_tile_size, c.max_episode_length_s = k["cmd"].data_ptr(), 4.0, 20.0
    c.terrain_levels, c.terrain_types, c.terrain_origins, c.num_cols = k["lv"].data_ptr(), k["ty"].data_ptr(), k["to"].data_ptr(), k["to"].shape[1]
    c.env_origins, c.injected_levels = k["eo"].data_ptr(), k["rnd"].data_ptr()
    L.check(L.lib.mdp_curriculum_terrain_levels_vel(n, C.byref(c), st))
    np.testing.assert_array_equal(to_np(k["lv"]), G["cur_levels_out"])
    np.testing.assert_array_equal(to_np(k["eo"]), G["cur_env_origins_out"])


@pytest.mark.gpu
def test_gpu_scaler_against_reference_vectors(device):
    from isaaclab_amd.rl import RunningStandardScaler

    sc = RunningStandardScaler(G["scaler_x"].shape[2], device=device)
    # two references: (1) the vectors the reference class produced -- it takes torch.mean / torch.var of the fp32 batch in
    # fp32, so they carry ITS reduction error (1.2e-4 relative) and only pin the statements; (2) the same statements
    # evaluated in fp64 (the oracle), which is what the contract's 1e-5 is measured against
    orc = RunningStandardScalerNP(G["scaler_x"].shape[2])
    for k in range(G["scaler_x"].shape[0]):
        y = sc(_dev(G["scaler_x"][k], device), train=True)
        assert_close(to_np(y), G["scaler_y"][k], rtol=1e-4, atol=2e-5, msg=f"batch {k}")   # reference reduces in fp32
        assert_close(to_np(y), orc.compute(G["scaler_x"][k], train=True), rtol=1e-5, atol=2e-6, msg=f"batch {k} vs fp64")
        np.testing.assert_allclose(to_np(sc.running_mean), orc.running_mean, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(to_np(sc.running_variance), orc.running_variance, rtol=1e-11)
    y3 = sc(_dev(G["scaler_x"][:2], device), train=True)
    assert_close(to_np(y3), G["scaler_y3"], rtol=1e-4, atol=2e-5)
    assert_close(to_np(y3), orc.compute(G["scaler_x"][:2], train=True), rtol=1e-5, atol=2e-6, msg="3-d batch vs fp64")
    np.testing.assert_allclose(to_np(sc.running_mean), G["scaler_mean"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(to_np(sc.running_variance), G["scaler_var"], rtol=1e-4)
    assert float(sc.current_count) == G["scaler_count"][0]
    # with the reference's own statistics the elementwise maps must agree to fp32 rounding of one division / sqrt
    sc.running_mean.copy_(_dev(G["scaler_mean"], device))
    sc.running_variance.copy_(_dev(G["scaler_var"], device))
    sc._prepare()
    assert_close(to_np(sc(_dev(G["scaler_y"][0], device), inverse=True)), G["scaler_inverse"], rtol=1e-6, atol=1e-6)
