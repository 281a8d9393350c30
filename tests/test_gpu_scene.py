"""GPU: state container (S1), contact sensor (S2), ray generation (S3), velocity command (G6/G7), joint action (G8),
modifiers / constant noise (T-mod, T-noise) and the statistics kernels, through the host-side mirrors of the reference
classes (HIP through the C ABI), against the reference-generated fixtures tests/golden/{sensors_state,command_action,
buffers_noise_modifiers}.npz.  fp32 <= 1e-5 relative, masks / counters bit-exact."""
import math
from types import SimpleNamespace as NS

import numpy as np
import pytest

from conftest import assert_close, to_dev, to_np

pytestmark = pytest.mark.gpu


def _view(n, j, b, device, bc=None, with_links=True):
    from isaaclab_amd.assets import SyntheticPhysicsView

    return SyntheticPhysicsView(n, j, b, device, num_contact_bodies=bc, with_links=with_links)


def test_contact_sensor_sequence(golden, device):
    """contact_sensor.py:314-373 + sensor_base.py:182-205 replayed on the reference's own schedule, incl. a mid-sequence
    reset of two envs (the behavioural spec is test_contact_sensor.py:371-482)."""
    from isaaclab_amd.sensors import ContactSensor, ContactSensorCfg

    G = golden("sensors_state")
    seq = G["contact_forces_seq"]
    T_, n, b, _ = seq.shape
    view = _view(n, 1, b, device, with_links=False)
    cfg = ContactSensorCfg(history_length=3, track_air_time=True, force_threshold=1.0, update_period=0.005)
    s = ContactSensor(cfg, view, [f"b{i}" for i in range(b)], n, str(device), 0.005)
    for t in range(T_):
        view.get_net_contact_forces().copy_(to_dev(seq[t].reshape(-1, 3), device))
        s.update(0.005)
        d = s.data
        assert_close(to_np(d.net_forces_w_history), G["contact_hist"][t], msg=f"hist t={t}")
        for k, a in (("last_air", d.last_air_time), ("cur_air", d.current_air_time), ("last_con", d.last_contact_time),
                     ("cur_con", d.current_contact_time), ("ts", s._timestamp)):
            assert_close(to_np(a), G["contact_" + k][t], msg=f"{k} t={t}")
        np.testing.assert_array_equal(to_np(s.compute_first_contact(0.02)), G["contact_first_contact"][t].astype(bool),
                                      err_msg=f"first t={t}")
        if t == 6:
            s.reset([1, 4])


def test_articulation_data(golden, device):
    """articulation_data.py:264-455,503-518 over a fake physics view (same arrays as the reference run)."""
    from isaaclab_amd.assets.articulation_data import ArticulationData

    G = golden("sensors_state")
    n, b = G["ad_lpose_xyzw"].shape[:2]
    j = G["ad_qd_seq"].shape[2]
    view = _view(n, j, b, device)
    view.get_root_transforms().copy_(to_dev(G["ad_pose_xyzw"], device))
    view.get_root_velocities().copy_(to_dev(G["ad_vel"], device))
    view.get_link_transforms().copy_(to_dev(G["ad_lpose_xyzw"], device))
    view.get_link_velocities().copy_(to_dev(G["ad_lvel"], device))
    data = ArticulationData(view, str(device))
    data.com_pos_b.copy_(to_dev(G["ad_coms"][..., :3], device))
    for t, dt in enumerate((0.005, 0.005, 0.01)):
        view.get_dof_velocities().copy_(to_dev(G["ad_qd_seq"][t], device))
        data._sim_timestamp += dt
        assert_close(to_np(data.joint_acc), G["ad_joint_acc"][t], rtol=2e-5, msg=f"joint_acc t={t}")
    assert_close(to_np(data.root_state_w), G["ad_root_state_w"])
    assert_close(to_np(data.root_link_state_w), G["ad_root_link_state_w"])
    assert_close(to_np(data.body_state_w), G["ad_body_state_w"])
    assert_close(to_np(data.body_link_state_w), G["ad_body_link_state_w"])
    assert_close(to_np(data.projected_gravity_b), G["ad_projected_gravity_b"])
    assert_close(to_np(data.heading_w), G["ad_heading_w"])
    assert_close(to_np(data.root_lin_vel_b), G["ad_root_lin_vel_b"])
    assert_close(to_np(data.root_ang_vel_b), G["ad_root_ang_vel_b"])


@pytest.mark.parametrize("tag,yaw_only", (("yaw", True), ("full", False)))
def test_raycaster_ray_generation(golden, device, tag, yaw_only):
    """ray_caster.py:220-260: the kernel generates the rays in registers; cast against the plane z = -30 the hit of
    ray (s, d) is s + t d with t = (-30 - s.z) / d.z, so the reference's captured (starts_w, dirs_w) pin the generation."""
    from isaaclab_amd.sensors import RayCaster, RayCasterCfg, patterns
    from isaaclab_amd.utils.warp import convert_to_warp_mesh

    G = golden("sensors_state")
    pose = G[f"rays_{tag}_pose_xyzw"]
    n = pose.shape[0]
    view = _view(n, 1, 1, device, with_links=False)
    view.get_root_transforms().copy_(to_dev(pose, device))
    z0 = -30.0
    pts = np.array([[-500, -500, z0], [500, -500, z0], [500, 500, z0], [-500, 500, z0]], np.float32)
    tris = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    mesh = convert_to_warp_mesh(pts, tris, str(device))
    cfg = RayCasterCfg(mesh_prim_paths=["/World/ground"], offset=RayCasterCfg.OffsetCfg(pos=(0.0, 0.0, 20.0)),
                       attach_yaw_only=yaw_only, pattern_cfg=patterns.GridPatternCfg(resolution=0.1, size=[1.6, 1.0]))
    rc = RayCaster(cfg, view, mesh, n, str(device), 0.005)
    assert rc.num_rays == 187
    assert_close(to_np(rc.ray_starts[0]) - np.array([0, 0, 20.0], np.float32), G["grid_starts"], atol=1e-6)
    assert_close(to_np(rc.ray_directions[0]), G["grid_dirs"])
    rc.drift.copy_(to_dev(G[f"rays_{tag}_drift"], device))
    hits = to_np(rc.data.ray_hits_w)
    s, d = G[f"rays_{tag}_starts_w"].astype(np.float64), G[f"rays_{tag}_dirs_w"].astype(np.float64)
    t = (z0 - s[..., 2]) / d[..., 2]
    ok = t > 0
    exp = s + t[..., None] * d
    assert ok.sum() > 0.5 * ok.size
    # distances are O(50 m): 1e-5 relative on the ray parameter
    np.testing.assert_allclose(hits[ok], exp[ok], rtol=1e-5, atol=1e-3)
    assert np.isinf(hits[~ok]).all()
    assert_close(to_np(rc.data.pos_w), G[f"rays_{tag}_pos_w"])
    assert_close(to_np(rc.data.quat_w), G[f"rays_{tag}_quat_w"])


def _command_term(G, device, cfg_kwargs):
    from isaaclab_amd.envs.mdp.commands import UniformVelocityCommand, UniformVelocityCommandCfg

    st = to_dev(G["root_state"], device)
    n = st.shape[0]
    robot = NS(data=NS(root_state_w=st))
    env = NS(num_envs=n, device=str(device), step_dt=0.02, scene={"robot": robot})
    r = UniformVelocityCommandCfg.Ranges(lin_vel_x=cfg_kwargs["lin_vel_x"], lin_vel_y=cfg_kwargs["lin_vel_y"],
                                         ang_vel_z=cfg_kwargs["ang_vel_z"], heading=cfg_kwargs["heading"])
    cfg = UniformVelocityCommandCfg(asset_name="robot", resampling_time_range=(10.0, 10.0),
                                    rel_standing_envs=cfg_kwargs["rel_standing_envs"], rel_heading_envs=1.0,
                                    heading_command=True, heading_control_stiffness=0.5, ranges=r)
    term = UniformVelocityCommand(cfg, env)
    term.injected_uniforms = to_dev(np.full((n, 7), 0.5, np.float32), device)
    return term, n


def test_velocity_command(golden, device):
    """velocity_command.py:111-160 + command_manager.py:151-187 (RNG-free cases of the fixture)."""
    import torch

    G = golden("command_action")
    A = dict(rel_standing_envs=0.02, lin_vel_x=(-1.0, 1.0), lin_vel_y=(-1.0, 1.0), ang_vel_z=(-1.0, 1.0),
             heading=(-math.pi, math.pi))
    term, n = _command_term(G, device, A)
    term.vel_command_b.copy_(to_dev(G["A_in_cmd"], device))
    term.heading_target.copy_(to_dev(G["A_in_heading_target"], device))
    term.is_heading_env.copy_(to_dev(G["A_in_is_heading"], device))
    term.is_standing_env.copy_(to_dev(G["A_in_is_standing"], device))
    term.time_left.fill_(5.0)
    term.compute(0.02)
    term.compute(0.02)
    assert_close(to_np(term.command), G["A_cmd"], atol=1e-6)
    assert_close(to_np(term.time_left), G["A_time_left"])
    assert_close(to_np(term.metrics["error_vel_xy"]), G["A_m_xy"])
    assert_close(to_np(term.metrics["error_vel_yaw"]), G["A_m_yaw"])
    np.testing.assert_array_equal(to_np(term.command_counter), G["A_counter"])
    heading_in = to_np(term.heading_target).copy()
    B = dict(rel_standing_envs=0.0, lin_vel_x=(0.7, 0.7), lin_vel_y=(-0.3, -0.3), ang_vel_z=(-1.0, 1.0), heading=(0.4, 0.4))
    termB, _ = _command_term(G, device, B)
    termB.vel_command_b.copy_(to_dev(G["B_in_cmd"], device))
    termB.time_left.copy_(to_dev(G["B_in_time_left"], device))
    termB.heading_target.copy_(to_dev(G["B_in_heading_target"], device))
    termB.command_counter.copy_(term.command_counter)
    termB.compute(0.02)
    resampled = np.arange(n) % 2 == 0
    got = to_np(termB.command)
    assert_close(got[:, :2], G["B_cmd"][:, :2], atol=1e-6)
    assert_close(got[resampled, 2], G["B_cmd"][resampled, 2], atol=1e-6)
    assert_close(to_np(termB.time_left), G["B_time_left"], atol=1e-6)
    np.testing.assert_array_equal(to_np(termB.command_counter), G["B_counter"])
    assert_close(to_np(termB.heading_target), G["B_heading_target"], atol=1e-6)
    np.testing.assert_array_equal(to_np(termB.is_heading_env), G["B_is_heading"])
    np.testing.assert_array_equal(to_np(termB.is_standing_env), G["B_is_standing"])
    assert heading_in.shape == (n,) and torch.is_tensor(termB.command)
    # device generator: draws stay inside the configured ranges and differ between launches (hipGraph-safe epochs)
    termB.injected_uniforms = None
    termB.time_left.fill_(0.0)
    termB.compute(0.02)
    tl1 = to_np(termB.time_left).copy()
    assert np.all(tl1 == np.float32(10.0))
    term.injected_uniforms = None
    term.time_left.fill_(0.0)
    term.compute(0.02)
    c1 = to_np(term.command).copy()
    term.time_left.fill_(0.0)
    term.compute(0.02)
    c2 = to_np(term.command)
    moving = ~to_np(term.is_standing_env)
    assert np.all(np.abs(c1[:, :2]) <= 1.0) and np.all(np.abs(c2) <= 1.0 + 1e-6)
    assert np.mean(c1[moving, 0] != c2[moving, 0]) > 0.9


def test_joint_action(golden, device):
    """joint_actions.py:130-139 + action_manager.py:331-332 through the raw C ABI entry point."""
    import torch

    from isaaclab_amd import _lib as L

    G = golden("command_action")
    raw, default = to_dev(G["ja_raw"], device), to_dev(G["ja_default"], device)
    n, a = raw.shape
    action = torch.full((n, a), 7.0, device=device)
    prev = torch.zeros((n, a), device=device)
    processed = torch.zeros((n, a), device=device)
    L.check(L.lib.mdp_joint_action_process(n, a, L.ptr(raw), a, L.ptr(action), L.ptr(prev), L.ptr(processed), None,
                                           0.5, L.ptr(default), None, L.stream()))
    assert_close(to_np(processed), G["ja_processed"])
    assert_close(to_np(action), G["ja_raw"])
    assert np.all(to_np(prev) == 7.0)


def test_modifiers_and_constant_noise(golden, device):
    """modifier.py:22-259, noise_model.py:20-42 (deterministic members of the families)."""
    from isaaclab_amd.utils import modifiers as mods
    from isaaclab_amd.utils import noise

    G = golden("buffers_noise_modifiers")
    d = to_dev(G["mod_data"], device)
    assert_close(to_np(mods.scale(d, 1.7)), G["mod_scale"])
    assert_close(to_np(mods.clip(d, (-0.5, None))), G["mod_clip"])
    assert_close(to_np(mods.bias(d, 0.3)), G["mod_bias"])
    assert_close(to_np(d), G["mod_data"])   # modifiers must not edit their input
    assert_close(to_np(noise.constant_noise(d, noise.ConstantNoiseCfg(bias=0.25))), G["noise_const_add"])
    assert_close(to_np(noise.constant_noise(d, noise.ConstantNoiseCfg(bias=0.25, operation="scale"))), G["noise_const_scale"])
    assert_close(to_np(noise.constant_noise(d, noise.ConstantNoiseCfg(bias=0.25, operation="abs"))), G["noise_const_abs"])
    filt = mods.DigitalFilter(mods.DigitalFilterCfg(A=[0.3, -0.1], B=[0.5, 0.2, 0.1]), (6, 4), str(device))
    integ = mods.Integrator(mods.IntegratorCfg(dt=0.02), (6, 4), str(device))
    seq = G["mod_seq"]
    for t in range(seq.shape[0]):
        x = to_dev(seq[t], device)
        assert_close(to_np(filt(x)), G["mod_filter"][t], msg=f"filter t={t}")
        assert_close(to_np(integ(x)), G["mod_integrator"][t], msg=f"integrator t={t}")
        if t == 2:
            filt.reset([0, 5])
            integ.reset([0, 5])
    # statistical members: uniform / gaussian noise draw from the device generator inside the configured law
    big = to_dev(np.zeros((4096, 8), np.float32), device)
    u = to_np(noise.uniform_noise(big, noise.UniformNoiseCfg(n_min=-0.1, n_max=0.3)))
    assert u.min() >= -0.1 and u.max() <= 0.3 and abs(u.mean() - 0.1) < 5e-3
    g = to_np(noise.gaussian_noise(big, noise.GaussianNoiseCfg(mean=0.5, std=2.0)))
    assert abs(g.mean() - 0.5) < 0.05 and abs(g.std() - 2.0) < 0.05


def test_stats_kernels(device):
    """Per-rank partials of the cross-GPU reductions: fp64 moments and masked sums vs numpy."""
    import torch

    from isaaclab_amd import _lib as L

    rng = np.random.RandomState(5)
    for n, dim in ((1, 3), (777, 48), (4096, 235)):
        x = (rng.randn(n, dim) * 3 + 1).astype(np.float32)
        xd = to_dev(x, device)
        out = torch.zeros(1 + 2 * dim, dtype=torch.float64, device=device)
        L.check(L.lib.mdp_stats_moments(L.ptr(xd), n, dim, dim, L.ptr(out), L.stream()))
        o = to_np(out)
        x64 = x.astype(np.float64)
        assert o[0] == n
        np.testing.assert_allclose(o[1:1 + dim], x64.sum(0), rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(o[1 + dim:], ((x64 - x64.mean(0)) ** 2).sum(0), rtol=1e-9, atol=1e-9)
    vals = rng.randn(5, 1000).astype(np.float32)
    mask = rng.rand(1000) < 0.3
    out = torch.zeros(10, dtype=torch.float64, device=device)
    d_vals, d_mask = to_dev(vals, device), to_dev(mask, device)     # named: the launch borrows the pointers
    L.check(L.lib.mdp_stats_masked_sums(L.ptr(d_vals), 5, 1000, L.ptr(d_mask), L.ptr(out), L.stream()))
    o = to_np(out).reshape(5, 2)
    np.testing.assert_allclose(o[:, 0], vals.astype(np.float64)[:, mask].sum(1), rtol=1e-12, atol=1e-9)
    assert np.all(o[:, 1] == mask.sum())


def test_masked_sums_of_every_family_in_one_launch(device):
    """mdp_stats_masked_sums_multi == one mdp_stats_masked_sums_typed call per family, bit for bit (float rows, uint8
    done flags, int64 levels over ALL envs), and equal to numpy; bad arguments are rejected before any launch."""
    import torch

    from isaaclab_amd import _lib as L

    rng = np.random.RandomState(11)
    for n in (1, 1000, 4096):
        fam = [((rng.randn(10, n) * 5).astype(np.float32), 0, True), (rng.randn(1, n).astype(np.float32), 0, True),
               ((rng.rand(2, n) < 0.4).astype(np.uint8), 1, True), (rng.randint(0, 10, (1, n)).astype(np.int64), 2, False)]
        mask = rng.rand(n) < 0.3
        d_mask = to_dev(mask, device)
        dev = [to_dev(v, device) for v, _, _ in fam]
        rows = sum(v.shape[0] for v, _, _ in fam)
        one = torch.full((2 * rows,), float("nan"), dtype=torch.float64, device=device)
        off = 0
        for d, (v, kind, masked) in zip(dev, fam):
            L.check(L.lib.mdp_stats_masked_sums_typed(L.ptr(d), kind, v.shape[0], n, L.ptr(d_mask) if masked else None,
                                                      one[off:].data_ptr(), L.stream()))
            off += 2 * v.shape[0]
        table = (L.MdpStatsSource * len(fam))()
        for t, d, (v, kind, masked) in zip(table, dev, fam):
            t.values, t.elem_kind, t.rows, t.masked = d.data_ptr(), kind, v.shape[0], int(masked)
        multi = torch.full((2 * rows,), float("nan"), dtype=torch.float64, device=device)
        L.check(L.lib.mdp_stats_masked_sums_multi(table, len(fam), n, L.ptr(d_mask), L.ptr(multi), L.stream()))
        np.testing.assert_array_equal(to_np(multi), to_np(one))
        o, k = to_np(multi).reshape(rows, 2), 0
        for v, kind, masked in fam:
            sel = mask if masked else np.ones(n, dtype=bool)
            for r in range(v.shape[0]):
                np.testing.assert_allclose(o[k, 0], v[r].astype(np.float64)[sel].sum(), rtol=1e-12, atol=1e-9)
                assert o[k, 1] == sel.sum()
                k += 1
    with pytest.raises(ValueError):      # a masked family without a mask
        L.check(L.lib.mdp_stats_masked_sums_multi(table, len(fam), n, None, L.ptr(multi), L.stream()))
    with pytest.raises(ValueError):
        L.check(L.lib.mdp_stats_masked_sums_multi(table, L.MDP_STATS_MAX_SOURCES + 1, n, L.ptr(d_mask), L.ptr(multi), L.stream()))


def test_masked_fill(device):
    """mdp_masked_fill == ``buf[env_ids] = value`` for every buffer of the call: env-major (N, k) float / int64 / bool
    buffers, a term-major (k, N) block, untouched rows bit-identical; empty mask = no-op; too many buffers = ValueError."""
    import ctypes as C
    import struct

    import torch

    from isaaclab_amd import _lib as L

    n = 333
    g = torch.Generator().manual_seed(3)
    mask = (torch.rand(n, generator=g) < 0.3).to(device)
    a = torch.randn(n, 7, generator=g).to(device)
    b = torch.randint(-5, 5, (n,), generator=g).to(device)                      # int64
    c = torch.zeros(n, dtype=torch.bool, device=device)
    d = torch.randn(5, n, generator=g).to(device)                               # term-major
    ref = [t.clone() for t in (a, b, c, d)]
    ref[0][mask] = 0.5
    ref[1][mask] = 0
    ref[2][mask] = True
    ref[3][:, mask] = 0.0

    def rec(t, count, env_stride, elem_stride, value):
        f = L.MdpFill()
        f.ptr, f.elem_bytes, f.count, f.env_stride, f.elem_stride = t.data_ptr(), t.element_size(), count, env_stride, elem_stride
        f.value_bits = struct.unpack("<I", struct.pack("<f", value))[0] if t.dtype == torch.float32 else int(value)
        return f

    fills = [rec(a, 7, 7, 1, 0.5), rec(b, 1, 1, 1, 0), rec(c, 1, 1, 1, 1), rec(d, 5, 1, n, 0.0)]
    arr = (L.MdpFill * len(fills))(*fills)
    L.check(L.lib.mdp_masked_fill(arr, len(fills), mask.data_ptr(), n, L.stream(device)))
    for got, exp in zip((a, b, c, d), ref):
        assert torch.equal(got, exp)
    none = torch.zeros(n, dtype=torch.bool, device=device)
    L.check(L.lib.mdp_masked_fill(arr, len(fills), none.data_ptr(), n, L.stream(device)))
    for got, exp in zip((a, b, c, d), ref):
        assert torch.equal(got, exp)
    many = (L.MdpFill * 49)(*([fills[0]] * 49))
    with pytest.raises(ValueError, match="at most"):
        L.check(L.lib.mdp_masked_fill(many, 49, mask.data_ptr(), n, L.stream(device)))
