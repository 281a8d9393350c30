"""GPU: actuator models, buffers, interpolation (HIP through the C ABI, driven by the host mirror of
``isaaclab.actuators`` / ``isaaclab.utils.buffers``) against reference-generated fixtures and the oracle.
fp32 tolerance 1e-5 relative; MFMA-evaluated nets 1e-5 relative with a 1e-5 absolute floor (sum order)."""
import numpy as np
import pytest

from conftest import assert_close, to_dev, to_np
from test_oracle_actuators import lstm_weights_from_golden, mlp_layers_from_golden

pytestmark = pytest.mark.gpu

NAMES = [f"j{i}" for i in range(12)]


def _action(io, device, keys=("q_des", "qd_des", "tau_ff")):
    from isaaclab_amd.utils.types import ArticulationActions

    vals = [to_dev(io[k], device) if k in io and io[k] is not None else None for k in keys]
    return ArticulationActions(joint_positions=vals[0], joint_velocities=vals[1], joint_efforts=vals[2])


@pytest.mark.parametrize("n", (1, 7, 64))
def test_pd_family(golden, device, n):
    import isaaclab_amd.actuators as act

    G = golden("actuators")
    for kind, cls, cfgcls, extra in (
        ("implicit", act.ImplicitActuator, act.ImplicitActuatorCfg, {}),
        ("ideal_pd", act.IdealPDActuator, act.IdealPDActuatorCfg, {}),
        ("dc_motor", act.DCMotor, act.DCMotorCfg, {"saturation_effort": 120.0}),
    ):
        cfg = cfgcls(joint_names_expr=[".*"], effort_limit=80.0, velocity_limit=7.5, stiffness={".*": 40.0},
                     damping={".*": 5.0}, **extra)
        a = cls(cfg, NAMES, slice(None), n, device)
        p = f"{kind}_n{n}_"
        io = {k: G[p + k] for k in ("q_des", "qd_des", "tau_ff", "q", "qd")}
        out = a.compute(_action(io, device), to_dev(io["q"], device), to_dev(io["qd"], device))
        assert_close(to_np(a.computed_effort), G[p + "computed"], msg=p)
        assert_close(to_np(a.applied_effort), G[p + "applied"], msg=p)
        if kind == "implicit":  # actuator_pd.py:62-87 returns the action untouched
            assert_close(to_np(out.joint_positions), io["q_des"])
        else:
            assert out.joint_positions is None and out.joint_velocities is None
            assert_close(to_np(out.joint_efforts), G[p + "applied"])


def test_dc_motor_infinite_saturation(golden, device):
    import isaaclab_amd.actuators as act

    G = golden("actuators")
    cfg = act.DCMotorCfg(joint_names_expr=[".*"], effort_limit=80.0, velocity_limit=7.5, stiffness=40.0, damping=5.0,
                         saturation_effort=None)
    a = act.DCMotor(cfg, NAMES, slice(None), 7, device)
    io = {k: G["dc_inf_" + k] for k in ("q_des", "qd_des", "tau_ff", "q", "qd")}
    a.compute(_action(io, device), to_dev(io["q"], device), to_dev(io["qd"], device))
    assert_close(to_np(a.computed_effort), G["dc_inf_computed"])
    assert_close(to_np(a.applied_effort), G["dc_inf_applied"])
    with pytest.raises(ValueError, match="velocity limit"):
        act.DCMotor(act.DCMotorCfg(joint_names_expr=[".*"], stiffness=1.0, damping=1.0, velocity_limit=None,
                                   saturation_effort=1.0), NAMES, slice(None), 2, device)


def test_delayed_pd_sequence(golden, device):
    import torch

    import isaaclab_amd.actuators as act

    G = golden("actuators")
    seq = G["delayed_inputs"]
    n, j = seq.shape[2], seq.shape[3]
    cfg = act.DelayedPDActuatorCfg(joint_names_expr=[".*"], effort_limit=50.0, velocity_limit=10.0, stiffness=20.0,
                                   damping=1.0, min_delay=0, max_delay=3)
    a = act.DelayedPDActuator(cfg, NAMES[:j], slice(None), n, device)
    lags = torch.from_numpy(G["delayed_lags"]).to(device)
    bufs = (a.positions_delay_buffer, a.velocities_delay_buffer, a.efforts_delay_buffer)
    for b in bufs:
        b.set_time_lag(lags)
        b.reset()
    for t in range(seq.shape[0]):
        io = dict(zip(("q_des", "qd_des", "tau_ff", "q", "qd"), seq[t]))
        a.compute(_action(io, device), to_dev(io["q"], device), to_dev(io["qd"], device))
        assert_close(to_np(a.computed_effort), G["delayed_computed"][t], msg=f"t={t}")
        assert_close(to_np(a.applied_effort), G["delayed_applied"][t], msg=f"t={t}")
        if t == 3:
            for b in bufs:
                b.reset([1, 4])
    with pytest.raises(ValueError):   # delay_buffer.py:109-113
        bufs[0].set_time_lag(7)


def test_remotized_pd(golden, device):
    import isaaclab_amd.actuators as act

    G = golden("actuators")
    n, j = G["remotized_q"].shape
    cfg = act.RemotizedPDActuatorCfg(joint_names_expr=[".*"], stiffness=60.0, damping=2.0, min_delay=0, max_delay=0,
                                     joint_parameter_lookup=G["remotized_lut"].tolist())
    a = act.RemotizedPDActuator(cfg, NAMES[:j], slice(None), n, device)
    a.reset(slice(None))
    io = {k: G["remotized_" + k] for k in ("q_des", "qd_des", "tau_ff", "q", "qd")}
    a.compute(_action(io, device), to_dev(io["q"], device), to_dev(io["qd"], device))
    assert_close(to_np(a.computed_effort), G["remotized_computed"])
    assert_close(to_np(a.applied_effort), G["remotized_applied"])


def _sea_net(G, tag):
    import torch

    lstm, dec = lstm_weights_from_golden(G, tag)
    hidden, layers = lstm[0][1].shape[1], len(lstm)

    class SeaNet(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lstm = torch.nn.LSTM(2, hidden, layers, batch_first=True)
            mods, d = [], hidden
            for w, _, _ in dec[:-1]:
                mods += [torch.nn.Linear(d, w.shape[0]), torch.nn.ELU()]
                d = w.shape[0]
            mods.append(torch.nn.Linear(d, 1))
            self.decoder = torch.nn.Sequential(*mods)

    net = SeaNet()
    sd = {k[len(tag) + 3:]: torch.from_numpy(v) for k, v in G.items() if k.startswith(f"{tag}_w_")}
    net.load_state_dict(sd)
    return net


@pytest.mark.parametrize("tag", ("lstm8", "lstm16"))
def test_actuator_net_lstm(golden, device, tag):
    import isaaclab_amd.actuators as act

    G = golden("actuators")
    seq = G[f"{tag}_inputs"]
    n, j = seq.shape[2], seq.shape[3]
    cfg = act.ActuatorNetLSTMCfg(joint_names_expr=[".*"], network_file=_sea_net(G, tag), saturation_effort=120.0,
                                 effort_limit=80.0, velocity_limit=7.5)
    a = act.ActuatorNetLSTM(cfg, NAMES, slice(None), n, device)
    for t in range(seq.shape[0]):
        io = dict(zip(("q_des", "q", "qd"), seq[t]))
        a.compute(_action(io, device, ("q_des", "_", "_")), to_dev(io["q"], device), to_dev(io["qd"], device))
        if t == 1:
            a.reset([2, 5])
        assert_close(to_np(a.computed_effort), G[f"{tag}_computed"][t], rtol=1e-5, atol=4e-6, msg=f"{tag} t={t}")
        assert_close(to_np(a.applied_effort), G[f"{tag}_applied"][t], rtol=1e-5, atol=4e-6, msg=f"{tag} t={t}")
        assert_close(to_np(a.sea_hidden_state), G[f"{tag}_h"][t], rtol=1e-5, atol=2e-6, msg=f"{tag} h t={t}")
        assert_close(to_np(a.sea_cell_state), G[f"{tag}_c"][t], rtol=1e-5, atol=2e-6, msg=f"{tag} c t={t}")


@pytest.mark.parametrize("order", ("pos_vel", "vel_pos"))
def test_actuator_net_mlp(golden, device, order):
    import torch

    import isaaclab_amd.actuators as act

    G = golden("actuators")
    mlp = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Softsign(), torch.nn.Linear(32, 32), torch.nn.Softsign(),
                              torch.nn.Linear(32, 1))
    mlp.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in G.items() if k.startswith("mlp_w_")})
    seq = G[f"mlp_{order}_inputs"]
    n, j = seq.shape[2], seq.shape[3]
    cfg = act.ActuatorNetMLPCfg(joint_names_expr=[".*"], network_file=mlp, saturation_effort=23.7, effort_limit=23.7,
                                velocity_limit=30.0, pos_scale=-1.0, vel_scale=1.0, torque_scale=1.0, input_order=order,
                                input_idx=[0, 1, 2])
    a = act.ActuatorNetMLP(cfg, NAMES, slice(None), n, device)
    for t in range(seq.shape[0]):
        io = dict(zip(("q_des", "q", "qd"), seq[t]))
        a.compute(_action(io, device, ("q_des", "_", "_")), to_dev(io["q"], device), to_dev(io["qd"], device))
        assert_close(to_np(a.computed_effort), G[f"mlp_{order}_computed"][t], rtol=1e-5, atol=4e-6, msg=f"t={t}")
        assert_close(to_np(a.applied_effort), G[f"mlp_{order}_applied"][t], rtol=1e-5, atol=4e-6, msg=f"t={t}")
        if t == 2:
            a.reset([0, 3])


def test_actuator_net_lstm_full_size_vs_oracle(golden, device):
    """4096 envs x 12 joints (config C3 shape): three steps of the MFMA LSTM against the numpy oracle."""
    import isaaclab_amd.actuators as act
    from oracle import actuators_np as A

    G = golden("actuators")
    lstm, dec = lstm_weights_from_golden(G, "lstm8")
    n, j = 4096, 12
    rng = np.random.default_rng(3)
    cfg = act.ActuatorNetLSTMCfg(joint_names_expr=[".*"], network_file=_sea_net(G, "lstm8"), saturation_effort=120.0,
                                 effort_limit=80.0, velocity_limit=7.5)
    a = act.ActuatorNetLSTM(cfg, NAMES, slice(None), n, device)
    h = np.zeros((2, n * j, 8), np.float32)
    c = np.zeros_like(h)
    for t in range(3):
        q_des = (rng.standard_normal((n, j)) * 0.5).astype(np.float32)
        q = (rng.standard_normal((n, j)) * 0.5).astype(np.float32)
        qd = (rng.standard_normal((n, j)) * 4).astype(np.float32)
        comp, app, h, c = A.actuator_net_lstm(q_des, q, qd, h, c, lstm, dec, np.float32(80.0), np.float32(7.5), 120.0)
        a.compute(_action({"q_des": q_des}, device, ("q_des", "_", "_")), to_dev(q, device), to_dev(qd, device))
        assert_close(to_np(a.computed_effort), comp, rtol=1e-5, atol=4e-6, msg=f"t={t}")
        assert_close(to_np(a.applied_effort), app, rtol=1e-5, atol=4e-6, msg=f"t={t}")
        assert_close(to_np(a.sea_hidden_state), h, rtol=1e-5, atol=4e-6)


def test_circular_and_delay_buffers(golden, device):
    import torch

    from isaaclab_amd.utils.buffers import CircularBuffer, DelayBuffer

    G = golden("buffers_noise_modifiers")
    data = to_dev(G["cb_data"], device)
    cb = CircularBuffer(4, 5, device)
    with pytest.raises(RuntimeError):     # circular_buffer.py:148-151: read before the first append
        cb[torch.zeros(5, dtype=torch.long, device=device)]
    for t in range(7):
        cb.append(data[t])
        assert_close(to_np(cb.buffer), G["cb_buffers"][t], msg=f"t={t}")
        if t == 3:
            cb.reset([1, 3])
    assert_close(to_np(cb[torch.tensor([0, 1, 2, 3, 9], device=device)]), G["cb_lookup"])
    with pytest.raises(ValueError):       # circular_buffer.py:103-106
        cb.append(data[0][:3])
    with pytest.raises(ValueError):       # circular_buffer.py:14-22
        CircularBuffer(0, 5, device)
    db = DelayBuffer(3, 5, device)
    db.set_time_lag(torch.tensor([0, 1, 2, 3, 2], dtype=torch.int, device=device))
    for t in range(7):
        assert_close(to_np(db.compute(data[t])), G["db_out"][t], msg=f"t={t}")
        if t == 4:
            db.reset([0, 4])


def test_linear_interpolation(golden, device):
    from isaaclab_amd.utils.interpolation import LinearInterpolation

    G = golden("buffers_noise_modifiers")
    li = LinearInterpolation(to_dev(G["li_x"], device), to_dev(G["li_y"], device), device)
    assert_close(to_np(li.compute(to_dev(G["li_q"], device))), G["li_out"])
    with pytest.raises(ValueError):       # linear_interpolation.py:38-48
        LinearInterpolation(to_dev(G["li_x"], device), to_dev(G["li_y"][:3], device), device)


@pytest.mark.parametrize("n", (3, 517))
def test_action_processed_inside_the_actuator_launch(device, n):
    """mdp_actuator_pd_compute_with_action == mdp_joint_action_process followed by mdp_actuator_pd_compute, bit for
    bit (action / prev_action / processed buffers and both effort outputs), with scale vector, offset and clip."""
    import ctypes as C

    import torch

    from isaaclab_amd import _lib as L

    J = 12
    g = torch.Generator().manual_seed(n)
    rnd = lambda *s: torch.randn(*s, generator=g).to(device)
    raw, q, qd = rnd(n, J + 3)[:, :J], rnd(n, J), rnd(n, J)      # strided raw actions
    scale_vec, offset = rnd(J).abs() + 0.1, rnd(n, J)
    clip = torch.stack([offset - 0.3, offset + 0.3], dim=-1).contiguous()
    kp, kd = rnd(n, J).abs() * 40, rnd(n, J).abs() * 2
    lim, vlim = torch.full((n, J), 80.0, device=device), torch.full((n, J), 7.5, device=device)

    def run(fused):
        action, prev = rnd(n, J) * 0 + 0.25, torch.zeros(n, J, device=device)
        processed = torch.zeros(n, J, device=device)
        computed, applied = torch.zeros(n, J, device=device), torch.zeros(n, J, device=device)
        a = L.MdpJointAction()
        a.actions_in, a.ld_in = raw.data_ptr(), raw.stride(0)
        a.action, a.prev_action, a.processed = action.data_ptr(), prev.data_ptr(), processed.data_ptr()
        a.scale_vec, a.scale, a.offset, a.clip = scale_vec.data_ptr(), 1.0, offset.data_ptr(), clip.data_ptr()
        tail = (None, 0, None, 0, q.data_ptr(), J, qd.data_ptr(), J, kp.data_ptr(), kd.data_ptr(), lim.data_ptr(),
                vlim.data_ptr(), 120.0, computed.data_ptr(), applied.data_ptr(), L.stream(device))
        kind = L.ACT_KINDS["MDP_ACT_DC_MOTOR"]
        if fused:
            L.check(L.lib.mdp_actuator_pd_compute_with_action(C.byref(a), kind, n, J, *tail))
        else:
            L.check(L.lib.mdp_joint_action_process(n, J, a.actions_in, a.ld_in, a.action, a.prev_action, a.processed,
                                                   a.scale_vec, a.scale, a.offset, a.clip, L.stream(device)))
            L.check(L.lib.mdp_actuator_pd_compute(kind, n, J, processed.data_ptr(), J, *tail))
        torch.cuda.synchronize()
        return [to_np(t) for t in (action, prev, processed, computed, applied)]

    for got, exp in zip(run(True), run(False)):
        np.testing.assert_array_equal(got, exp)
    exp_p = np.clip(to_np(raw) * to_np(scale_vec) + to_np(offset), to_np(clip)[..., 0], to_np(clip)[..., 1])
    assert_close(run(True)[2], exp_p)
    with pytest.raises(ValueError, match="Invalid action shape"):
        a = L.MdpJointAction()
        a.actions_in, a.ld_in = raw.data_ptr(), J - 1
        a.action = a.prev_action = a.processed = q.data_ptr()
        L.check(L.lib.mdp_actuator_pd_compute_with_action(
            C.byref(a), L.ACT_KINDS["MDP_ACT_IDEAL_PD"], n, J, None, 0, None, 0, q.data_ptr(), J, qd.data_ptr(), J,
            kp.data_ptr(), kd.data_ptr(), lim.data_ptr(), vlim.data_ptr(), 0.0, q.data_ptr(), q.data_ptr(),
            L.stream(device)))


@pytest.mark.parametrize("name", ("anymal_c_rough", "anymal_d_flat"))
def test_device_conditional_actuator_step(device, name):
    """mdp_actuator_{lstm,pd}_compute_gated: the write_data_to_sim() the reference runs only in steps that reset an env
    (manager_based_rl_env.py:214-222) with the condition on the device -- gate == 0 leaves every output and the LSTM
    state untouched, gate > 0 is bit-identical to the plain call."""
    import torch

    from isaaclab_amd.envs.synthetic import make_velocity_env

    kw = dict(small_terrain=True) if "rough" in name else {}
    env, rec = make_velocity_env(name, num_envs=48, device=str(device), steps=2, seed=51, **kw)
    rec.feed(0)
    env.step(rec.actions[0].to(device))
    robot = env.scene["robot"]
    act = next(iter(robot.actuators.values()))
    state = [act.computed_effort, act.applied_effort]
    if hasattr(act, "sea_hidden_state"):
        state += [act.sea_hidden_state, act.sea_cell_state]
    start = [t.clone() for t in state]
    robot.write_data_to_sim(gate=torch.zeros(1, device=device))
    torch.cuda.synchronize()
    for t, ref in zip(state, start):
        assert torch.equal(t, ref)
    robot.write_data_to_sim(gate=torch.full((1,), 3.0, device=device))
    gated = [t.clone() for t in state]
    for t, ref in zip(state, start):
        t.copy_(ref)
    robot.write_data_to_sim()
    torch.cuda.synchronize()
    for t, ref, g in zip(state, start, gated):
        assert torch.equal(t, g)
    if hasattr(act, "sea_hidden_state"):
        assert not torch.equal(act.sea_hidden_state, start[2])      # the open gate did advance the LSTM state
