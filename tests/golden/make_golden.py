"""Generates the committed golden fixtures by running the REFERENCE implementation on CPU.

Run in the build container only (needs /root/reference):  ``python tests/golden/make_golden.py``
Outputs ``tests/golden/*.npz`` (inputs + reference outputs, fp32 / bool / int arrays; data only).
The reference source, byte-code or stubs never travel: tests read only the ``.npz`` files.

Coverage: math M1-M13, actuators A1-A7 + buffers, noise/modifiers (deterministic
ones), state container S1, contact sensor S2, ray generation S3, term library T-obs/T-rew/T-rew-vel/T-term,
managers G3-G5, velocity command G7 (RNG-free cases), joint action G8, in-hand env T-shadow.
"""
from __future__ import annotations

import importlib
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader as R  # noqa: E402

torch.set_num_threads(4)
R.install_stubs()
OUT = HERE


def save(name, **arrays):
    clean = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        clean[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **clean)
    print(f"{name}: {len(clean)} arrays, {sum(a.nbytes for a in clean.values()) / 1024:.1f} KiB")


def rand_quat(g, n):
    q = torch.randn(n, 4, generator=g)
    return q / q.norm(dim=-1, keepdim=True)


# ================================================================================================
def gen_math():
    m = R.ref_math()
    g = torch.Generator().manual_seed(101)
    out = {}
    for n in (1, 7, 64):
        q1, q2 = rand_quat(g, n), rand_quat(g, n)
        q1[::3] *= -1
        v = torch.randn(n, 3, generator=g)
        p = f"n{n}_"
        out.update({p + "q1": q1, p + "q2": q2, p + "v": v})
        out[p + "quat_mul"] = m.quat_mul(q1, q2)
        out[p + "quat_conjugate"] = m.quat_conjugate(q1)
        out[p + "quat_inv"] = m.quat_inv(q1 * 1.7)
        out[p + "quat_unique"] = m.quat_unique(q1)
        out[p + "quat_rotate"] = m.quat_rotate(q1, v)
        out[p + "quat_rotate_inverse"] = m.quat_rotate_inverse(q1, v)
        out[p + "quat_apply"] = m.quat_apply(q1, v)
        out[p + "quat_apply_yaw"] = m.quat_apply_yaw(q1, v)
        out[p + "yaw_quat"] = m.yaw_quat(q1)
        rpy = (torch.rand(n, 3, generator=g) - 0.5) * 6.0
        out[p + "rpy"] = rpy
        out[p + "quat_from_euler_xyz"] = m.quat_from_euler_xyz(rpy[:, 0], rpy[:, 1], rpy[:, 2])
        out[p + "euler_xyz_from_quat"] = torch.stack(m.euler_xyz_from_quat(q1), dim=-1)
        out[p + "axis_angle_from_quat"] = m.axis_angle_from_quat(q1)
        out[p + "quat_error_magnitude"] = m.quat_error_magnitude(q1, q2)
        out[p + "quat_box_minus"] = m.quat_box_minus(q1, q2)
        angle = (torch.rand(n, generator=g) - 0.5) * 6.0
        axis = torch.randn(n, 3, generator=g)
        out.update({p + "angle": angle, p + "axis": axis})
        out[p + "quat_from_angle_axis"] = m.quat_from_angle_axis(angle, axis)
        mat = m.matrix_from_quat(q1)
        out[p + "matrix_from_quat"] = mat
        out[p + "quat_from_matrix"] = m.quat_from_matrix(mat)
        ang = (torch.rand(n, 5, generator=g) - 0.5) * 40.0
        out[p + "angles"] = ang
        out[p + "wrap_to_pi"] = m.wrap_to_pi(ang)
        x = torch.randn(n, 6, generator=g)
        lo = -1.0 - torch.rand(6, generator=g)
        hi = 1.0 + torch.rand(6, generator=g)
        out.update({p + "x": x, p + "lo": lo, p + "hi": hi})
        out[p + "scale_transform"] = m.scale_transform(x, lo, hi)
        out[p + "unscale_transform"] = m.unscale_transform(x, lo, hi)
        out[p + "saturate"] = m.saturate(x * 2, lo, hi)
        out[p + "normalize"] = m.normalize(x)
        out[p + "copysign"] = m.copysign(2.5, x[:, 0])
        t1, t2 = torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g)
        out.update({p + "t1": t1, p + "t2": t2})
        a, b = m.combine_frame_transforms(t1, q1, t2, q2)
        out[p + "combine_frame_transforms"] = torch.cat([a, b], dim=-1)
        a, b = m.subtract_frame_transforms(t1, q1, t2, q2)
        out[p + "subtract_frame_transforms"] = torch.cat([a, b], dim=-1)
        a, b = m.compute_pose_error(t1, q1, t2, q2, "axis_angle")
        out[p + "compute_pose_error_aa"] = torch.cat([a, b], dim=-1)
        a, b = m.compute_pose_error(t1, q1, t2, q2, "quat")
        out[p + "compute_pose_error_quat"] = torch.cat([a, b], dim=-1)
        delta = torch.randn(n, 6, generator=g) * 0.3
        delta[::4, 3:] = 0.0
        out[p + "delta"] = delta
        a, b = m.apply_delta_pose(t1, q1, delta)
        out[p + "apply_delta_pose"] = torch.cat([a, b], dim=-1)
        out[p + "skew_symmetric_matrix"] = m.skew_symmetric_matrix(v)
        pts = torch.randn(n, 5, 3, generator=g)
        out[p + "pts"] = pts
        out[p + "transform_points"] = m.transform_points(pts, t1, q1)
    save("math", **out)


# ================================================================================================
def _act_common(g, n, j):
    return dict(q_des=torch.randn(n, j, generator=g) * 0.5, qd_des=torch.randn(n, j, generator=g) * 0.1,
                tau_ff=torch.randn(n, j, generator=g) * 2.0, q=torch.randn(n, j, generator=g) * 0.5,
                qd=torch.randn(n, j, generator=g) * 4.0)


def gen_actuators():
    act = R.ref_import("isaaclab.actuators")
    from isaaclab.utils.types import ArticulationActions

    g = torch.Generator().manual_seed(202)
    out = {}
    names = [f"j{i}" for i in range(12)]
    for n in (1, 7, 64):
        for kind, cls, cfgcls, extra in (
            ("implicit", act.ImplicitActuator, act.ImplicitActuatorCfg, {}),
            ("ideal_pd", act.IdealPDActuator, act.IdealPDActuatorCfg, {}),
            ("dc_motor", act.DCMotor, act.DCMotorCfg, {"saturation_effort": 120.0}),
        ):
            cfg = cfgcls(joint_names_expr=[".*"], effort_limit=80.0, velocity_limit=7.5, stiffness={".*": 40.0},
                         damping={".*": 5.0}, **extra)
            a = cls(cfg, names, slice(None), n, "cpu")
            io = _act_common(g, n, 12)
            io["qd"][0, 0] = 7.5  # hits the zero of the DC-motor characteristic exactly
            ca = ArticulationActions(joint_positions=io["q_des"].clone(), joint_velocities=io["qd_des"].clone(),
                                     joint_efforts=io["tau_ff"].clone())
            a.compute(ca, io["q"], io["qd"])
            p = f"{kind}_n{n}_"
            out.update({p + k: v for k, v in io.items()})
            out[p + "computed"] = a.computed_effort
            out[p + "applied"] = a.applied_effort
    # DC motor with saturation = inf (cfg None)
    cfg = act.DCMotorCfg(joint_names_expr=[".*"], effort_limit=80.0, velocity_limit=7.5, stiffness=40.0, damping=5.0,
                         saturation_effort=None)
    a = act.DCMotor(cfg, names, slice(None), 7, "cpu")
    io = _act_common(g, 7, 12)
    ca = ArticulationActions(joint_positions=io["q_des"].clone(), joint_velocities=io["qd_des"].clone(),
                             joint_efforts=io["tau_ff"].clone())
    a.compute(ca, io["q"], io["qd"])
    out.update({"dc_inf_" + k: v for k, v in io.items()})
    out["dc_inf_computed"], out["dc_inf_applied"] = a.computed_effort, a.applied_effort

    # delayed PD: sequence of 6 computes, fixed lags injected (reset draws them with randint)
    n, j = 7, 4
    cfg = act.DelayedPDActuatorCfg(joint_names_expr=[".*"], effort_limit=50.0, velocity_limit=10.0, stiffness=20.0,
                                   damping=1.0, min_delay=0, max_delay=3)
    a = act.DelayedPDActuator(cfg, names[:j], slice(None), n, "cpu")
    lags = torch.tensor([0, 1, 2, 3, 3, 1, 0], dtype=torch.int)
    for buf in (a.positions_delay_buffer, a.velocities_delay_buffer, a.efforts_delay_buffer):
        buf.set_time_lag(lags)
        buf.reset()
    out["delayed_lags"] = lags
    seq_in, seq_comp, seq_app = [], [], []
    for t in range(6):
        io = _act_common(g, n, j)
        ca = ArticulationActions(joint_positions=io["q_des"].clone(), joint_velocities=io["qd_des"].clone(),
                                 joint_efforts=io["tau_ff"].clone())
        a.compute(ca, io["q"], io["qd"])
        seq_in.append(torch.stack([io[k] for k in ("q_des", "qd_des", "tau_ff", "q", "qd")]))
        seq_comp.append(a.computed_effort.clone())
        seq_app.append(a.applied_effort.clone())
        if t == 3:  # partial reset in the middle of the sequence
            for buf in (a.positions_delay_buffer, a.velocities_delay_buffer, a.efforts_delay_buffer):
                buf.reset([1, 4])
    out["delayed_inputs"] = torch.stack(seq_in)
    out["delayed_computed"] = torch.stack(seq_comp)
    out["delayed_applied"] = torch.stack(seq_app)

    # remotized PD
    lut = [[-1.0, 1.0, 20.0], [-0.5, 1.0, 35.0], [0.0, 1.0, 50.0], [0.7, 1.0, 30.0], [1.2, 1.0, 10.0]]
    cfg = act.RemotizedPDActuatorCfg(joint_names_expr=[".*"], stiffness=60.0, damping=2.0, min_delay=0, max_delay=0,
                                     joint_parameter_lookup=lut)
    a = act.RemotizedPDActuator(cfg, names[:j], slice(None), n, "cpu")
    a.reset(slice(None))
    io = _act_common(g, n, j)
    io["q"][0, 0], io["q"][1, 1], io["q"][2, 2] = -2.0, 2.0, -0.5
    ca = ArticulationActions(joint_positions=io["q_des"].clone(), joint_velocities=io["qd_des"].clone(),
                             joint_efforts=io["tau_ff"].clone())
    a.compute(ca, io["q"], io["qd"])
    out.update({"remotized_" + k: v for k, v in io.items()})
    out["remotized_lut"] = torch.tensor(lut)
    out["remotized_computed"], out["remotized_applied"] = a.computed_effort, a.applied_effort

    # actuator nets: locally built TorchScript modules with seeded random weights (the real files are remote)
    class SeaNet(torch.nn.Module):
        def __init__(self, hidden, layers, dec):
            super().__init__()
            self.lstm = torch.nn.LSTM(2, hidden, layers, batch_first=True)
            mods, d = [], hidden
            for w in dec:
                mods += [torch.nn.Linear(d, w), torch.nn.ELU()]
                d = w
            mods.append(torch.nn.Linear(d, 1))
            self.decoder = torch.nn.Sequential(*mods)

        def forward(self, x, hc: tuple[torch.Tensor, torch.Tensor]):
            o, (h, c) = self.lstm(x, hc)
            return self.decoder(o[:, -1, :]), (h, c)

    for tag, hidden, layers, dec in (("lstm8", 8, 2, []), ("lstm16", 16, 1, [32])):
        torch.manual_seed(303 + hidden)
        net = torch.jit.script(SeaNet(hidden, layers, dec))
        import io as _io

        buf = _io.BytesIO()
        torch.jit.save(net, buf)
        blob = buf.getvalue()
        n, j = 7, 12
        cfg = act.ActuatorNetLSTMCfg(joint_names_expr=[".*"], network_file="unused", saturation_effort=120.0,
                                     effort_limit=80.0, velocity_limit=7.5)
        import isaaclab.actuators.actuator_net as an

        an.read_file = lambda path, _b=blob: _io.BytesIO(_b)
        a = act.ActuatorNetLSTM(cfg, names, slice(None), n, "cpu")
        seq_in, seq_comp, seq_app, seq_h, seq_c = [], [], [], [], []
        for t in range(4):
            io = _act_common(g, n, j)
            ca = ArticulationActions(joint_positions=io["q_des"].clone(), joint_velocities=None, joint_efforts=None)
            a.compute(ca, io["q"], io["qd"])
            seq_in.append(torch.stack([io["q_des"], io["q"], io["qd"]]))
            seq_comp.append(a.computed_effort.clone())
            seq_app.append(a.applied_effort.clone())
            seq_h.append(a.sea_hidden_state.clone())
            seq_c.append(a.sea_cell_state.clone())
            if t == 1:
                a.reset([2, 5])
                seq_h[-1] = a.sea_hidden_state.clone()
                seq_c[-1] = a.sea_cell_state.clone()
        for k, v in net.state_dict().items():
            out[f"{tag}_w_{k}"] = v
        out[f"{tag}_inputs"] = torch.stack(seq_in)
        out[f"{tag}_computed"] = torch.stack(seq_comp)
        out[f"{tag}_applied"] = torch.stack(seq_app)
        out[f"{tag}_h"] = torch.stack(seq_h)
        out[f"{tag}_c"] = torch.stack(seq_c)

    # MLP net (Go1-like: taps [0,1,2], 6 -> 32 -> 32 -> 1, softsign), pos_scale -1
    torch.manual_seed(404)
    mlp = torch.jit.script(torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Softsign(), torch.nn.Linear(32, 32),
                                               torch.nn.Softsign(), torch.nn.Linear(32, 1)))
    import io as _io

    buf = _io.BytesIO()
    torch.jit.save(mlp, buf)
    blob = buf.getvalue()
    import isaaclab.actuators.actuator_net as an

    an.read_file = lambda path, _b=blob: _io.BytesIO(_b)
    for order in ("pos_vel", "vel_pos"):
        cfg = act.ActuatorNetMLPCfg(joint_names_expr=[".*"], network_file="unused", saturation_effort=23.7,
                                    effort_limit=23.7, velocity_limit=30.0, pos_scale=-1.0, vel_scale=1.0,
                                    torque_scale=1.0, input_order=order, input_idx=[0, 1, 2])
        n, j = 7, 12
        a = act.ActuatorNetMLP(cfg, names, slice(None), n, "cpu")
        seq_in, seq_comp, seq_app = [], [], []
        for t in range(5):
            io = _act_common(g, n, j)
            ca = ArticulationActions(joint_positions=io["q_des"].clone(), joint_velocities=None, joint_efforts=None)
            a.compute(ca, io["q"], io["qd"])
            seq_in.append(torch.stack([io["q_des"], io["q"], io["qd"]]))
            seq_comp.append(a.computed_effort.clone())
            seq_app.append(a.applied_effort.clone())
            if t == 2:
                a.reset([0, 3])
        out[f"mlp_{order}_inputs"] = torch.stack(seq_in)
        out[f"mlp_{order}_computed"] = torch.stack(seq_comp)
        out[f"mlp_{order}_applied"] = torch.stack(seq_app)
    for k, v in mlp.state_dict().items():
        out[f"mlp_w_{k}"] = v
    save("actuators", **out)


# ================================================================================================
def gen_buffers_noise_modifiers():
    from isaaclab.utils.buffers import CircularBuffer, DelayBuffer
    from isaaclab.utils.interpolation import LinearInterpolation
    import isaaclab.utils.modifiers as mods
    import isaaclab.utils.noise as noise

    g = torch.Generator().manual_seed(505)
    out = {}
    cb = CircularBuffer(4, 5, "cpu")
    data = torch.randn(7, 5, 3, generator=g)
    bufs = []
    for t in range(7):
        cb.append(data[t])
        bufs.append(cb.buffer.clone())
        if t == 3:
            cb.reset([1, 3])
    out["cb_data"] = data
    out["cb_buffers"] = torch.stack(bufs)
    out["cb_lookup"] = cb[torch.tensor([0, 1, 2, 3, 9])]
    db = DelayBuffer(3, 5, "cpu")
    db.set_time_lag(torch.tensor([0, 1, 2, 3, 2], dtype=torch.int))
    outs = []
    for t in range(7):
        outs.append(db.compute(data[t]).clone())
        if t == 4:
            db.reset([0, 4])
    out["db_out"] = torch.stack(outs)
    x = torch.tensor([-1.0, -0.2, 0.3, 0.35, 2.0])
    y = torch.tensor([3.0, 1.0, -2.0, 4.0, 0.5])
    q = torch.cat([torch.randn(40, generator=g) * 1.5, x])
    out.update({"li_x": x, "li_y": y, "li_q": q, "li_out": LinearInterpolation(x, y, "cpu").compute(q)})
    d = torch.randn(6, 4, generator=g)
    out["mod_data"] = d
    out["mod_scale"] = mods.scale(d, 1.7)
    out["mod_clip"] = mods.clip(d, (-0.5, None))
    out["mod_bias"] = mods.bias(d, 0.3)
    out["noise_const_add"] = noise.constant_noise(d, noise.ConstantNoiseCfg(bias=0.25))
    out["noise_const_scale"] = noise.constant_noise(d, noise.ConstantNoiseCfg(bias=0.25, operation="scale"))
    out["noise_const_abs"] = noise.constant_noise(d, noise.ConstantNoiseCfg(bias=0.25, operation="abs"))
    fcfg = mods.DigitalFilterCfg(A=[0.3, -0.1], B=[0.5, 0.2, 0.1])
    filt = mods.DigitalFilter(fcfg, (6, 4), "cpu")
    icfg = mods.IntegratorCfg(dt=0.02)
    integ = mods.Integrator(icfg, (6, 4), "cpu")
    seq = torch.randn(5, 6, 4, generator=g)
    fo, io_ = [], []
    for t in range(5):
        fo.append(filt(seq[t]).clone())
        io_.append(integ(seq[t]).clone())
        if t == 2:
            filt.reset([0, 5])
            integ.reset([0, 5])
    out.update({"mod_seq": seq, "mod_filter": torch.stack(fo), "mod_integrator": torch.stack(io_)})
    save("buffers_noise_modifiers", **out)


# ================================================================================================
class _NS(types.SimpleNamespace):
    pass


def make_state(g, n, j=12, b=17, h=3, r=11):
    st = torch.zeros(n, 13)
    st[:, :3] = torch.randn(n, 3, generator=g) * torch.tensor([3.0, 3.0, 0.1]) + torch.tensor([0.0, 0.0, 0.6])
    st[:, 3:7] = rand_quat(g, n)
    st[:, 7:] = torch.randn(n, 6, generator=g) * 0.5
    d = dict(
        root_state=st, joint_pos=torch.randn(n, j, generator=g) * 0.6, joint_vel=torch.randn(n, j, generator=g) * 3.0,
        joint_acc=torch.randn(n, j, generator=g) * 20.0, default_joint_pos=torch.randn(n, j, generator=g) * 0.3,
        default_joint_vel=torch.zeros(n, j), applied_torque=torch.randn(n, j, generator=g) * 30.0,
        action=torch.randn(n, j, generator=g), prev_action=torch.randn(n, j, generator=g),
        command=torch.randn(n, 3, generator=g), env_origins=torch.randn(n, 3, generator=g) * 5.0,
        body_state=torch.randn(n, b, 13, generator=g), body_acc=torch.randn(n, b, 6, generator=g) * 3.0,
        soft_joint_vel_limits=torch.full((n, j), 2.5) + torch.rand(n, j, generator=g),
    )
    d["computed_torque"] = d["applied_torque"].clone()
    mask = torch.rand(n, j, generator=g) < 0.4
    d["computed_torque"][mask] += torch.randn(int(mask.sum()), generator=g) * 10.0
    lim = torch.zeros(n, j, 2)
    lim[..., 0] = -0.8 - torch.rand(n, j, generator=g) * 0.3
    lim[..., 1] = 0.8 + torch.rand(n, j, generator=g) * 0.3
    d["soft_joint_pos_limits"] = lim
    hist = torch.randn(n, h, b, 3, generator=g) * 0.8
    hist[torch.rand(n, h, b, generator=g) < 0.3] *= 20.0
    d["contact_hist"] = hist
    for k in ("cur_air", "cur_contact", "last_air", "last_contact"):
        t = torch.rand(n, b, generator=g) * 0.6
        d[k] = t
    on = torch.rand(n, b, generator=g) < 0.5
    d["cur_contact"] = torch.where(on, d["cur_contact"] * 0.08, torch.zeros_like(on, dtype=torch.float))
    d["cur_air"] = torch.where(on, torch.zeros_like(on, dtype=torch.float), d["cur_air"])
    d["ray_hits"] = torch.randn(n, r, 3, generator=g)
    d["ray_hits"][0, 0, :] = float("inf")
    d["sensor_pos"] = st[:, :3] + torch.tensor([0.0, 0.0, 20.0])
    d["episode_length"] = torch.randint(0, 1200, (n,), generator=g)
    d["time_left"] = torch.rand(n, generator=g) * 0.05
    d["command_counter"] = torch.randint(0, 3, (n,), generator=g)
    d["terminated"] = torch.rand(n, generator=g) < 0.3
    d["time_outs"] = torch.rand(n, generator=g) < 0.3
    d["term_dones"] = torch.rand(3, n, generator=g) < 0.4
    return d


def make_env(d, math_mod):
    """Duck-typed env over the state dict (the reference's own manager tests mock the env the same way)."""
    n = d["root_state"].shape[0]
    st = d["root_state"]
    data = _NS(
        root_pos_w=st[:, :3], root_quat_w=st[:, 3:7], root_lin_vel_w=st[:, 7:10], root_ang_vel_w=st[:, 10:13],
        root_lin_vel_b=math_mod.quat_rotate_inverse(st[:, 3:7], st[:, 7:10]),
        root_ang_vel_b=math_mod.quat_rotate_inverse(st[:, 3:7], st[:, 10:13]),
        projected_gravity_b=math_mod.quat_rotate_inverse(st[:, 3:7], torch.tensor([0.0, 0.0, -1.0]).repeat(n, 1)),
        joint_pos=d["joint_pos"], joint_vel=d["joint_vel"], joint_acc=d["joint_acc"],
        default_joint_pos=d["default_joint_pos"], default_joint_vel=d["default_joint_vel"],
        soft_joint_pos_limits=d["soft_joint_pos_limits"], soft_joint_vel_limits=d["soft_joint_vel_limits"],
        applied_torque=d["applied_torque"], computed_torque=d["computed_torque"],
        body_lin_acc_w=d["body_acc"][..., :3], body_lin_vel_w=d["body_state"][..., 7:10],
    )
    robot = _NS(data=data)
    contact = _NS(
        data=_NS(net_forces_w_history=d["contact_hist"], current_air_time=d["cur_air"],
                 current_contact_time=d["cur_contact"], last_air_time=d["last_air"], last_contact_time=d["last_contact"]),
        compute_first_contact=lambda dt, abs_tol=1.0e-8: (d["cur_contact"] > 0.0) * (d["cur_contact"] < (dt + abs_tol)),
    )
    rays = _NS(data=_NS(pos_w=d["sensor_pos"], ray_hits_w=d["ray_hits"]))

    class Scene(dict):
        pass

    scene = Scene(robot=robot, height_scanner=rays, contact_forces=contact)
    scene.sensors = {"height_scanner": rays, "contact_forces": contact}
    scene.env_origins = d["env_origins"]
    term_names = ["time_out", "base_contact", "bad"]
    env = _NS(
        num_envs=n, device="cpu", scene=scene, step_dt=0.02, max_episode_length=1000, max_episode_length_s=20.0,
        episode_length_buf=d["episode_length"],
        action_manager=_NS(action=d["action"], prev_action=d["prev_action"]),
        command_manager=_NS(get_command=lambda name: d["command"],
                            get_term=lambda name: _NS(time_left=d["time_left"], command_counter=d["command_counter"])),
        termination_manager=_NS(terminated=d["terminated"], time_outs=d["time_outs"],
                                get_term=lambda name: d["term_dones"][term_names.index(name)],
                                find_terms=lambda keys: term_names[:2], active_terms=term_names),
    )
    return env


def gen_terms():
    m = R.ref_math()
    obs = R.ref_mdp("observations")
    rew = R.ref_mdp("rewards")
    term = R.ref_mdp("terminations")
    vrew = R.ref_velocity_mdp("rewards")
    from isaaclab.managers import SceneEntityCfg

    g = torch.Generator().manual_seed(606)
    out = {}
    feet, thighs, some_j = [13, 14, 15, 16], [5, 6, 7, 8], [0, 3, 7, 11]

    def cfg(name, joint_ids=None, body_ids=None):
        c = SceneEntityCfg(name)
        if joint_ids is not None:
            c.joint_ids = joint_ids
        if body_ids is not None:
            c.body_ids = body_ids
        return c

    for n in (1, 7, 64):
        d = make_state(g, n)
        env = make_env(d, m)
        p = f"n{n}_"
        out.update({p + "in_" + k: v for k, v in d.items()})
        O = {
            "base_pos_z": obs.base_pos_z(env), "base_lin_vel": obs.base_lin_vel(env),
            "base_ang_vel": obs.base_ang_vel(env), "projected_gravity": obs.projected_gravity(env),
            "root_pos_w": obs.root_pos_w(env), "root_quat_w": obs.root_quat_w(env),
            "root_quat_w_unique": obs.root_quat_w(env, make_quat_unique=True),
            "root_lin_vel_w": obs.root_lin_vel_w(env), "root_ang_vel_w": obs.root_ang_vel_w(env),
            "joint_pos": obs.joint_pos(env), "joint_pos_sel": obs.joint_pos(env, cfg("robot", some_j)),
            "joint_pos_rel": obs.joint_pos_rel(env), "joint_pos_rel_sel": obs.joint_pos_rel(env, cfg("robot", some_j)),
            "joint_pos_limit_normalized": obs.joint_pos_limit_normalized(env),
            "joint_vel": obs.joint_vel(env), "joint_vel_rel": obs.joint_vel_rel(env),
            "height_scan": obs.height_scan(env, cfg("height_scanner")), "last_action": obs.last_action(env),
            "generated_commands": obs.generated_commands(env, "base_velocity"),
        }
        out.update({p + "obs_" + k: v for k, v in O.items()})
        itt = rew.is_terminated_term(types.SimpleNamespace(params={"term_keys": ".*"}), env)
        Rw = {
            "is_alive": rew.is_alive(env), "is_terminated": rew.is_terminated(env), "is_terminated_term": itt(env),
            "lin_vel_z_l2": rew.lin_vel_z_l2(env), "ang_vel_xy_l2": rew.ang_vel_xy_l2(env),
            "flat_orientation_l2": rew.flat_orientation_l2(env), "base_height_l2": rew.base_height_l2(env, 0.55),
            "base_height_l2_sensor": rew.base_height_l2(env, 0.55, sensor_cfg=cfg("height_scanner")),
            "body_lin_acc_l2": rew.body_lin_acc_l2(env, cfg("robot", body_ids=feet)),
            "joint_torques_l2": rew.joint_torques_l2(env), "joint_vel_l1": rew.joint_vel_l1(env, cfg("robot", some_j)),
            "joint_vel_l2": rew.joint_vel_l2(env), "joint_acc_l2": rew.joint_acc_l2(env),
            "joint_deviation_l1": rew.joint_deviation_l1(env, cfg("robot", some_j)),
            "joint_pos_limits": rew.joint_pos_limits(env), "joint_vel_limits": rew.joint_vel_limits(env, 0.9),
            "applied_torque_limits": rew.applied_torque_limits(env), "action_rate_l2": rew.action_rate_l2(env),
            "action_l2": rew.action_l2(env),
            "undesired_contacts": rew.undesired_contacts(env, 1.0, cfg("contact_forces", body_ids=thighs)),
            "contact_forces": rew.contact_forces(env, 1.0, cfg("contact_forces", body_ids=feet)),
            "track_lin_vel_xy_exp": rew.track_lin_vel_xy_exp(env, math.sqrt(0.25), "base_velocity"),
            "track_ang_vel_z_exp": rew.track_ang_vel_z_exp(env, math.sqrt(0.25), "base_velocity"),
            "feet_air_time": vrew.feet_air_time(env, "base_velocity", cfg("contact_forces", body_ids=feet), 0.5),
            "feet_air_time_positive_biped": vrew.feet_air_time_positive_biped(
                env, "base_velocity", 0.4, cfg("contact_forces", body_ids=feet[:2])),
            "feet_slide": vrew.feet_slide(env, cfg("contact_forces", body_ids=feet), cfg("robot", body_ids=feet)),
            "track_lin_vel_xy_yaw_frame_exp": vrew.track_lin_vel_xy_yaw_frame_exp(env, 0.5, "base_velocity"),
            "track_ang_vel_z_world_exp": vrew.track_ang_vel_z_world_exp(env, "base_velocity", 0.5),
        }
        out.update({p + "rew_" + k: v.float() for k, v in Rw.items()})
        Tm = {
            "time_out": term.time_out(env), "command_resample": term.command_resample(env, "base_velocity", 1),
            "bad_orientation": term.bad_orientation(env, 1.0),
            "root_height_below_minimum": term.root_height_below_minimum(env, 0.55),
            "joint_pos_out_of_manual_limit": term.joint_pos_out_of_manual_limit(env, (-0.9, 0.9), cfg("robot", some_j)),
            "joint_vel_out_of_limit": term.joint_vel_out_of_limit(env),
            "joint_vel_out_of_manual_limit": term.joint_vel_out_of_manual_limit(env, 6.0),
            "joint_effort_out_of_limit": term.joint_effort_out_of_limit(env, cfg("robot", some_j)),
            "illegal_contact": term.illegal_contact(env, 1.0, cfg("contact_forces", body_ids=[0])),
        }
        out.update({p + "term_" + k: v for k, v in Tm.items()})
        try:
            term.joint_pos_out_of_limit(env)
            out[p + "term_joint_pos_out_of_limit_raises"] = np.array(0)
        except IndexError:
            out[p + "term_joint_pos_out_of_limit_raises"] = np.array(1)
    save("terms", **out)


# ================================================================================================
def gen_managers():
    """Reward / termination / observation managers of the reference driven by the duck-typed env with the
    velocity-task term tables (flat + rough variants), two consecutive computes."""
    m = R.ref_math()
    obs = R.ref_mdp("observations")
    rew = R.ref_mdp("rewards")
    term = R.ref_mdp("terminations")
    vrew = R.ref_velocity_mdp("rewards")
    mg = R.ref_import("isaaclab.managers")
    from isaaclab.managers import SceneEntityCfg
    from isaaclab.utils.noise import ConstantNoiseCfg

    g = torch.Generator().manual_seed(707)
    out = {}
    feet, thighs = [13, 14, 15, 16], [5, 6, 7, 8]

    def ent(name, body_ids=None):
        c = SceneEntityCfg(name)
        if body_ids is not None:
            c.body_ids = body_ids
        c.resolve = lambda scene: None
        return c

    for n in (7, 64):
        d = make_state(g, n)
        env = make_env(d, m)
        p = f"n{n}_"
        out.update({p + "in_" + k: v for k, v in d.items()})
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
            "flat_orientation_l2": mg.RewardTermCfg(func=rew.flat_orientation_l2, weight=0.0),
            "dof_pos_limits": mg.RewardTermCfg(func=rew.joint_pos_limits, weight=0.0),
        }
        rm = mg.RewardManager(rcfg, env)
        r1 = rm.compute(0.02).clone()
        r2 = rm.compute(0.02).clone()
        out[p + "reward_1"], out[p + "reward_2"] = r1, r2
        out[p + "episode_sums"] = torch.stack([rm._episode_sums[k] for k in rm.active_terms])
        out[p + "step_reward"] = rm._step_reward
        tcfg = {
            "time_out": mg.TerminationTermCfg(func=term.time_out, time_out=True),
            "base_contact": mg.TerminationTermCfg(func=term.illegal_contact,
                                                  params={"sensor_cfg": ent("contact_forces", [0]), "threshold": 1.0}),
            "bad": mg.TerminationTermCfg(func=term.bad_orientation, params={"limit_angle": 1.0}),
        }
        tm = mg.TerminationManager(tcfg, env)
        out[p + "dones"] = tm.compute().clone()
        out[p + "terminated"], out[p + "time_outs"] = tm.terminated, tm.time_outs
        out[p + "term_dones"] = torch.stack([tm.get_term(k) for k in tm.active_terms])

        # observation manager: corruption with the deterministic constant-noise model + clip + scale
        grp = mg.ObservationGroupCfg()
        grp.enable_corruption = True
        grp.concatenate_terms = True
        grp.base_lin_vel = mg.ObservationTermCfg(func=obs.base_lin_vel, noise=ConstantNoiseCfg(bias=0.05))
        grp.base_ang_vel = mg.ObservationTermCfg(func=obs.base_ang_vel, scale=0.25)
        grp.projected_gravity = mg.ObservationTermCfg(func=obs.projected_gravity, clip=(-0.5, 0.5))
        grp.velocity_commands = mg.ObservationTermCfg(func=obs.generated_commands, params={"command_name": "base_velocity"})
        grp.joint_pos = mg.ObservationTermCfg(func=obs.joint_pos_rel, noise=ConstantNoiseCfg(bias=1.1, operation="scale"))
        grp.joint_vel = mg.ObservationTermCfg(func=obs.joint_vel_rel, scale=tuple(0.05 * (i + 1) for i in range(12)))
        grp.actions = mg.ObservationTermCfg(func=obs.last_action)
        grp.height_scan = mg.ObservationTermCfg(func=obs.height_scan, params={"sensor_cfg": ent("height_scanner")},
                                                noise=ConstantNoiseCfg(bias=-0.02), clip=(-1.0, 1.0))
        om = mg.ObservationManager({"policy": grp}, env)
        out[p + "obs"] = om.compute()["policy"]
        # history variant (unfused path): 3-step history on two terms, two pushes
        grp2 = mg.ObservationGroupCfg()
        grp2.concatenate_terms = True
        grp2.enable_corruption = False
        grp2.history_length = 3
        grp2.flatten_history_dim = True
        grp2.base_lin_vel = mg.ObservationTermCfg(func=obs.base_lin_vel)
        grp2.joint_pos = mg.ObservationTermCfg(func=obs.joint_pos_rel)
        om2 = mg.ObservationManager({"policy": grp2}, env)
        h1 = om2.compute()["policy"].clone()
        d["joint_pos"] += 0.1
        h2 = om2.compute()["policy"].clone()
        d["joint_pos"] -= 0.1
        out[p + "obs_hist_1"], out[p + "obs_hist_2"] = h1, h2
    save("managers", **out)


# ================================================================================================
def gen_sensors_state():
    m = R.ref_math()
    cs = importlib.import_module("isaaclab.sensors.contact_sensor.contact_sensor")
    sb = importlib.import_module("isaaclab.sensors.sensor_base")
    rc = importlib.import_module("isaaclab.sensors.ray_caster.ray_caster")
    pt = importlib.import_module("isaaclab.sensors.ray_caster.patterns.patterns")
    ad = importlib.import_module("isaaclab.assets.articulation.articulation_data")
    g = torch.Generator().manual_seed(808)
    out = {}

    # ---- contact sensor: reference update code run on a fake self ----
    n, b, h = 7, 5, 3
    forces_seq = torch.zeros(12, n, b, 3)
    on = torch.rand(n, b, generator=g) < 0.5
    for t in range(12):
        flip = torch.rand(n, b, generator=g) < 0.3
        on = on ^ flip
        forces_seq[t, ..., 2] = on * (torch.rand(n, b, generator=g) * 50.0 + 1.5)
        forces_seq[t, ..., 0] = torch.randn(n, b, generator=g) * 0.3
    fake = _NS()
    fake.cfg = _NS(history_length=h, filter_prim_paths_expr=[], track_pose=False, track_air_time=True,
                   force_threshold=1.0, update_period=0.005)
    fake._num_envs, fake._num_bodies, fake._sim_physics_dt = n, b, 0.005
    fake._is_visualizing = False
    fake._data = _NS(net_forces_w=torch.zeros(n, b, 3), net_forces_w_history=torch.zeros(n, h, b, 3),
                     last_air_time=torch.zeros(n, b), current_air_time=torch.zeros(n, b),
                     last_contact_time=torch.zeros(n, b), current_contact_time=torch.zeros(n, b))
    fake._timestamp = torch.zeros(n)
    fake._timestamp_last_update = torch.zeros(n)
    fake._is_outdated = torch.ones(n, dtype=torch.bool)
    cur = {"t": 0}
    fake.contact_physx_view = _NS(get_net_contact_forces=lambda dt: forces_seq[cur["t"]].reshape(-1, 3))
    fake._update_buffers_impl = types.MethodType(cs.ContactSensor._update_buffers_impl, fake)
    fake._update_outdated_buffers = types.MethodType(sb.SensorBase._update_outdated_buffers, fake)
    snaps = {k: [] for k in ("hist", "last_air", "cur_air", "last_con", "cur_con", "ts", "first_contact")}
    for t in range(12):
        cur["t"] = t
        sb.SensorBase.update(fake, 0.005)
        snaps["hist"].append(fake._data.net_forces_w_history.clone())
        snaps["last_air"].append(fake._data.last_air_time.clone())
        snaps["cur_air"].append(fake._data.current_air_time.clone())
        snaps["last_con"].append(fake._data.last_contact_time.clone())
        snaps["cur_con"].append(fake._data.current_contact_time.clone())
        snaps["ts"].append(fake._timestamp.clone())
        fake.data = fake._data
        snaps["first_contact"].append(cs.ContactSensor.compute_first_contact(fake, 0.02).clone())
        if t == 6:  # reset two envs mid-sequence (contact_sensor.py:136-155 + sensor_base.py:182-195)
            ids = [1, 4]
            sb.SensorBase.reset(fake, ids)
            fake._data.net_forces_w[ids] = 0.0
            fake._data.net_forces_w_history[ids] = 0.0
            for k in ("current_air_time", "last_air_time", "current_contact_time", "last_contact_time"):
                getattr(fake._data, k)[ids] = 0.0
    out["contact_forces_seq"] = forces_seq
    out.update({"contact_" + k: torch.stack(v) for k, v in snaps.items()})

    # ---- ray caster: ray generation of the reference with raycast_mesh intercepted ----
    cfgp = _NS(resolution=0.1, size=[1.6, 1.0], direction=(0.0, 0.0, -1.0), ordering="xy")
    starts, dirs = pt.grid_pattern(cfgp, "cpu")
    out["grid_starts"], out["grid_dirs"] = starts, dirs
    for yaw_only in (True, False):
        n = 5
        fake = _NS()
        fake.cfg = _NS(attach_yaw_only=yaw_only, max_distance=1e6, mesh_prim_paths=["/World/ground"])
        pose = torch.zeros(n, 7)
        pose[:, :3] = torch.randn(n, 3, generator=g)
        q = rand_quat(g, n)
        pose[:, 3:] = m.convert_quat(q, to="xyzw")

        class View(rc.physx.ArticulationView if isinstance(rc.physx.ArticulationView, type) else object):
            pass

        rc.physx.ArticulationView = View
        view = View.__new__(View)
        view.get_root_transforms = lambda: pose
        fake._view = view
        fake.num_rays = len(dirs)
        fake.ray_starts = (starts + torch.tensor([0.0, 0.0, 20.0])).repeat(n, 1, 1)
        fake.ray_directions = dirs.repeat(n, 1, 1)
        fake.drift = torch.randn(n, 3, generator=g) * 0.01
        fake._data = _NS(pos_w=torch.zeros(n, 3), quat_w=torch.zeros(n, 4), ray_hits_w=torch.zeros(n, len(dirs), 3))
        fake.meshes = {"/World/ground": None}
        captured = {}

        def fake_raycast(ray_starts, ray_directions, max_dist, mesh):
            captured["starts"], captured["dirs"] = ray_starts.clone(), ray_directions.clone()
            return (torch.zeros_like(ray_starts), None, None, None)

        rc.raycast_mesh = fake_raycast
        rc.XFormPrim = type("XFormPrim", (), {})
        rc.RayCaster._update_buffers_impl(fake, torch.arange(n))
        tag = "yaw" if yaw_only else "full"
        out[f"rays_{tag}_pose_xyzw"] = pose
        out[f"rays_{tag}_drift"] = fake.drift
        out[f"rays_{tag}_starts_w"] = captured["starts"]
        out[f"rays_{tag}_dirs_w"] = captured["dirs"]
        out[f"rays_{tag}_pos_w"], out[f"rays_{tag}_quat_w"] = fake._data.pos_w, fake._data.quat_w

    # ---- ArticulationData derived quantities on a fake physx view ----
    n, b = 7, 4
    pose = torch.zeros(n, 7)
    pose[:, :3] = torch.randn(n, 3, generator=g)
    pose[:, 3:] = m.convert_quat(rand_quat(g, n), to="xyzw")
    vel = torch.randn(n, 6, generator=g)
    lpose = torch.zeros(n, b, 7)
    lpose[..., :3] = torch.randn(n, b, 3, generator=g)
    lpose[..., 3:] = m.convert_quat(rand_quat(g, n * b).view(n, b, 4), to="xyzw")
    lvel = torch.randn(n, b, 6, generator=g)
    qd_seq = torch.randn(3, n, 6, generator=g)
    coms = torch.zeros(n, b, 7)
    coms[..., :3] = torch.randn(n, b, 3, generator=g) * 0.05
    coms[..., 6] = 1.0
    cur = {"t": 0}
    # PhysX fills its buffer afresh on every getter call; the reference edits the returned tensor in place
    # (articulation_data.py:360-365), so the fake view hands out copies
    view = _NS(count=n, get_root_transforms=lambda: pose.clone(), get_root_velocities=lambda: vel.clone(),
               get_link_transforms=lambda: lpose.clone(), get_link_velocities=lambda: lvel.clone(),
               get_dof_positions=lambda: qd_seq[cur["t"]] * 0.1, get_dof_velocities=lambda: qd_seq[cur["t"]],
               get_coms=lambda: coms, get_link_accelerations=lambda: lvel * 2.0)
    data = ad.ArticulationData.__new__(ad.ArticulationData)
    data.device = "cpu"
    data._root_physx_view = view
    data._physics_sim_view = _NS(update_articulations_kinematic=lambda: None)
    data._sim_timestamp = 0.0
    data.GRAVITY_VEC_W = torch.tensor((0.0, 0.0, -1.0)).repeat(n, 1)
    data.FORWARD_VEC_B = torch.tensor((1.0, 0.0, 0.0)).repeat(n, 1)
    from isaaclab.utils.buffers import TimestampedBuffer

    for name in ("_root_state_w", "_root_link_state_w", "_root_com_state_w", "_body_state_w", "_body_link_state_w",
                 "_body_com_state_w", "_body_acc_w", "_joint_pos", "_joint_acc", "_joint_vel"):
        setattr(data, name, TimestampedBuffer())
    data._previous_joint_vel = torch.zeros(n, 6)
    data._joint_acc.timestamp = 0.0
    accs = []
    for t, dt in enumerate((0.005, 0.005, 0.01)):
        cur["t"] = t
        data._sim_timestamp += dt
        accs.append(data.joint_acc.clone())
    out["ad_pose_xyzw"], out["ad_vel"], out["ad_lpose_xyzw"], out["ad_lvel"] = pose, vel, lpose, lvel
    out["ad_coms"], out["ad_qd_seq"] = coms, qd_seq
    out["ad_joint_acc"] = torch.stack(accs)
    out["ad_root_state_w"] = data.root_state_w
    out["ad_root_link_state_w"] = data.root_link_state_w
    out["ad_body_state_w"] = data.body_state_w
    out["ad_body_link_state_w"] = data.body_link_state_w
    out["ad_projected_gravity_b"] = data.projected_gravity_b
    out["ad_heading_w"] = data.heading_w
    out["ad_root_lin_vel_b"], out["ad_root_ang_vel_b"] = data.root_lin_vel_b, data.root_ang_vel_b
    save("sensors_state", **out)


# ================================================================================================
def gen_command_action():
    m = R.ref_math()
    cmd, ccfg = R.ref_velocity_command()
    ja = R.load_by_path("_ref_joint_actions", os.path.join(R.LAB, "isaaclab", "envs", "mdp", "actions", "joint_actions.py"))
    g = torch.Generator().manual_seed(909)
    out = {}
    n = 64
    st = torch.zeros(n, 13)
    st[:, 3:7] = rand_quat(g, n)
    st[:, 7:] = torch.randn(n, 6, generator=g)
    robot = _NS(data=_NS(
        root_lin_vel_b=m.quat_rotate_inverse(st[:, 3:7], st[:, 7:10]),
        root_ang_vel_b=m.quat_rotate_inverse(st[:, 3:7], st[:, 10:13]),
        heading_w=torch.atan2(m.quat_apply(st[:, 3:7], torch.tensor([1.0, 0, 0]).repeat(n, 1))[:, 1],
                              m.quat_apply(st[:, 3:7], torch.tensor([1.0, 0, 0]).repeat(n, 1))[:, 0])))
    env = _NS(num_envs=n, device="cpu", step_dt=0.02, scene={"robot": robot})
    out["root_state"] = st
    # case A: standard ranges, no resample during compute (time_left large): metrics + heading + standing
    cfgA = _NS(asset_name="robot", resampling_time_range=(10.0, 10.0), rel_standing_envs=0.02, rel_heading_envs=1.0,
               heading_command=True, heading_control_stiffness=0.5, debug_vis=False,
               ranges=_NS(lin_vel_x=(-1.0, 1.0), lin_vel_y=(-1.0, 1.0), ang_vel_z=(-1.0, 1.0), heading=(-math.pi, math.pi)))
    term = cmd.UniformVelocityCommand.__new__(cmd.UniformVelocityCommand)
    term.cfg, term._env = cfgA, env
    term.metrics = {"error_vel_xy": torch.zeros(n), "error_vel_yaw": torch.zeros(n)}
    term.robot = robot
    term.time_left = torch.full((n,), 5.0)
    term.command_counter = torch.zeros(n, dtype=torch.long)
    term.vel_command_b = torch.randn(n, 3, generator=g)
    term.heading_target = (torch.rand(n, generator=g) - 0.5) * 6.0
    term.is_heading_env = torch.rand(n, generator=g) < 0.7
    term.is_standing_env = torch.rand(n, generator=g) < 0.1
    out.update({"A_in_cmd": term.vel_command_b.clone(), "A_in_heading_target": term.heading_target.clone(),
                "A_in_is_heading": term.is_heading_env.clone(), "A_in_is_standing": term.is_standing_env.clone()})
    term.compute(0.02)
    term.compute(0.02)
    out.update({"A_cmd": term.vel_command_b.clone(), "A_time_left": term.time_left.clone(),
                "A_m_xy": term.metrics["error_vel_xy"].clone(), "A_m_yaw": term.metrics["error_vel_yaw"].clone(),
                "A_counter": term.command_counter.clone()})
    # case B: degenerate ranges -> resampling is RNG-independent; half the envs expire
    cfgB = _NS(asset_name="robot", resampling_time_range=(10.0, 10.0), rel_standing_envs=0.0, rel_heading_envs=1.0,
               heading_command=True, heading_control_stiffness=0.5, debug_vis=False,
               ranges=_NS(lin_vel_x=(0.7, 0.7), lin_vel_y=(-0.3, -0.3), ang_vel_z=(-1.0, 1.0), heading=(0.4, 0.4)))
    term.cfg = cfgB
    term.time_left = torch.where(torch.arange(n) % 2 == 0, torch.tensor(0.01), torch.tensor(3.0))
    term.vel_command_b = torch.randn(n, 3, generator=g)
    term.is_heading_env = torch.zeros(n, dtype=torch.bool)
    term.is_standing_env = torch.zeros(n, dtype=torch.bool)
    term.metrics = {"error_vel_xy": torch.zeros(n), "error_vel_yaw": torch.zeros(n)}
    out.update({"B_in_cmd": term.vel_command_b.clone(), "B_in_time_left": term.time_left.clone(),
                "B_in_heading_target": term.heading_target.clone()})
    term.compute(0.02)
    out.update({"B_cmd": term.vel_command_b, "B_time_left": term.time_left, "B_counter": term.command_counter,
                "B_heading_target": term.heading_target, "B_is_heading": term.is_heading_env,
                "B_is_standing": term.is_standing_env})
    # joint action processing (joint_actions.py:130-139)
    raw = torch.randn(n, 12, generator=g)
    default = torch.randn(n, 12, generator=g) * 0.3
    out["ja_raw"], out["ja_default"] = raw, default
    out["ja_processed"] = raw * 0.5 + default
    save("command_action", **out)


# ================================================================================================
def gen_inhand():
    ih = R.ref_inhand()
    m = R.ref_math()
    g = torch.Generator().manual_seed(1010)
    out = {}
    for n in (7, 64):
        p = f"n{n}_"
        D, Fg, B, A = 24, 5, 26, 20
        obj = torch.zeros(n, 13)
        obj[:, :3] = torch.randn(n, 3, generator=g) * 0.05 + torch.tensor([0.0, -0.39, 0.6])
        obj[:, 3:7] = rand_quat(g, n)
        obj[:, 7:] = torch.randn(n, 6, generator=g)
        origins = torch.randn(n, 3, generator=g) * 3.0
        obj[:, :3] += origins
        in_hand = torch.tensor([0.0, -0.39, 0.56]).repeat(n, 1)
        goal = rand_quat(g, n)
        goal[::5] = obj[::5, 3:7]  # a few successes
        obj[1::7, :3] += 0.5  # a few falls
        actions = torch.rand(n, A, generator=g) * 2 - 1
        reset_buf = torch.rand(n, generator=g) < 0.2
        reset_goal = torch.rand(n, generator=g) < 0.1
        successes = torch.randint(0, 5, (n,), generator=g).float()
        cons = torch.tensor([1.3])
        obj_pos = obj[:, :3] - origins
        rew, goal_resets, succ, cs = ih.compute_rewards(
            reset_buf, reset_goal, successes, cons, 600.0, obj_pos, obj[:, 3:7], in_hand, goal,
            -10.0, 1.0, 0.1, actions, -0.0002, 0.1, 250.0, 0.24, 0.0, 0.1)
        out.update({p + "obj": obj, p + "origins": origins, p + "in_hand": in_hand, p + "goal": goal,
                    p + "actions": actions, p + "reset_buf": reset_buf, p + "reset_goal": reset_goal,
                    p + "successes": successes, p + "cons": cons, p + "reward": rew, p + "goal_resets": goal_resets,
                    p + "successes_out": succ, p + "cons_out": cs})
        out[p + "rotation_distance"] = ih.rotation_distance(obj[:, 3:7], goal)
        # apply action
        lower = -1.0 - torch.rand(n, D, generator=g)
        upper = 1.0 + torch.rand(n, D, generator=g)
        act_idx = sorted(torch.randperm(D, generator=g)[:A].tolist())
        prev = torch.randn(n, D, generator=g) * 0.5
        cur = torch.zeros(n, D)
        fake = _NS(cfg=_NS(act_moving_average=0.3), actions=actions, actuated_dof_indices=act_idx,
                   hand_dof_lower_limits=lower, hand_dof_upper_limits=upper, prev_targets=prev.clone(), cur_targets=cur,
                   hand=_NS(set_joint_position_target=lambda *a, **k: None))
        ih.InHandManipulationEnv._apply_action(fake)
        out.update({p + "lower": lower, p + "upper": upper, p + "act_idx": torch.tensor(act_idx), p + "prev": prev,
                    p + "cur_out": fake.cur_targets, p + "prev_out": fake.prev_targets})
        # full observations + dones
        q = lower + (upper - lower) * torch.rand(n, D, generator=g)
        qd = torch.randn(n, D, generator=g)
        body = torch.randn(n, B, 13, generator=g)
        fingers = sorted(torch.randperm(B, generator=g)[:Fg].tolist())
        ep = torch.randint(0, 700, (n,), generator=g)
        fake = _NS(cfg=_NS(vel_obs_scale=0.2, fall_dist=0.24, max_consecutive_success=3, success_tolerance=0.1),
                   num_envs=n, num_fingertips=Fg, finger_bodies=fingers, actions=actions,
                   hand=_NS(data=_NS(body_pos_w=body[..., :3].clone(), body_quat_w=body[..., 3:7],
                                     body_vel_w=body[..., 7:13], joint_pos=q, joint_vel=qd)),
                   object=_NS(data=_NS(root_pos_w=obj[:, :3], root_quat_w=obj[:, 3:7], root_vel_w=obj[:, 7:13],
                                       root_lin_vel_w=obj[:, 7:10], root_ang_vel_w=obj[:, 10:13])),
                   scene=_NS(env_origins=origins), hand_dof_lower_limits=lower, hand_dof_upper_limits=upper,
                   in_hand_pos=in_hand, goal_rot=goal, successes=successes, episode_length_buf=ep.clone(),
                   max_episode_length=600)
        fake._compute_intermediate_values = types.MethodType(ih.InHandManipulationEnv._compute_intermediate_values, fake)
        oor, to = ih.InHandManipulationEnv._get_dones(fake)
        obs = ih.InHandManipulationEnv.compute_full_observations(fake)
        out.update({p + "q": q, p + "qd": qd, p + "body": body, p + "fingers": torch.tensor(fingers), p + "ep": ep,
                    p + "ep_out": fake.episode_length_buf, p + "out_of_reach": oor, p + "time_out": to,
                    p + "full_obs": obs})
    save("inhand", **out)


def gen_inhand_reset():
    """The rest of the in-hand env, run by the REFERENCE methods on an uninitialised instance with duck-typed assets:
    compute_reduced_observations (:281-296), compute_full_state (:324-350), _reset_target_pose (:248-260) and _reset_idx
    (:201-246, with the real RigidObject.write_root_*_to_sim / Articulation.write_joint_state_to_sim /
    set_joint_position_target behind it) -- torch.rand replaced by recorded draws."""
    ih = R.ref_inhand()
    art = R.ref_import("isaaclab.assets.articulation.articulation")
    rig = R.ref_import("isaaclab.assets.rigid_object.rigid_object")
    # the env module was loaded against the name-only stub of isaaclab.envs, so `super()._reset_idx` has nothing to find:
    # import the real DirectRLEnv (a few more name-only stubs for its UI imports) and lend its _reset_idx to the stub base
    R.install_stubs()
    for name in ("omni.kit.commands", "omni.ui", "isaaclab.envs.ui", "omni.kit.viewport", "omni.kit.viewport.utility",
                 "isaacsim.core.utils.viewports"):
        R._stub(name)
    direct = R.ref_import("isaaclab.envs.direct_rl_env")
    ih.InHandManipulationEnv.__mro__[1]._reset_idx = direct.DirectRLEnv._reset_idx
    g = torch.Generator().manual_seed(2024)
    out = {}

    class View:
        def __init__(self):
            self.calls = {}

        def __getattr__(self, name):
            if name.startswith("set_"):
                def f(data, indices=None):
                    self.calls[name] = (data.clone(), None if indices is None else torch.as_tensor(indices).clone())
                return f
            raise AttributeError(name)

    for n in (7, 48):
        p = f"n{n}_"
        D, Fg, B, A = 24, 5, 26, 20
        origins = torch.randn(n, 3, generator=g) * 3.0
        obj = torch.zeros(n, 13)
        obj[:, :3] = torch.randn(n, 3, generator=g) * 0.05 + torch.tensor([0.0, -0.39, 0.6]) + origins
        obj[:, 3:7] = rand_quat(g, n)
        obj[:, 7:] = torch.randn(n, 6, generator=g)
        goal = rand_quat(g, n)
        in_hand = torch.tensor([0.0, -0.39, 0.56]).repeat(n, 1)
        actions = torch.rand(n, A, generator=g) * 2 - 1
        lower = -1.0 - torch.rand(n, D, generator=g)
        upper = 1.0 + torch.rand(n, D, generator=g)
        q = lower + (upper - lower) * torch.rand(n, D, generator=g)
        qd = torch.randn(n, D, generator=g)
        body = torch.randn(n, B, 13, generator=g)
        forces = torch.randn(n, B, 6, generator=g)
        fingers = sorted(torch.randperm(B, generator=g)[:Fg].tolist())
        env = object.__new__(ih.InHandManipulationEnv)
        env.cfg = _NS(vel_obs_scale=0.2, force_torque_obs_scale=10.0, reset_position_noise=0.01, reset_dof_pos_noise=0.2,
                      reset_dof_vel_noise=0.3, events=None, action_noise_model=None, observation_noise_model=None)
        env.num_envs, env.num_fingertips, env.finger_bodies, env.actions, env.device = n, Fg, fingers, actions, "cpu"
        env.num_hand_dofs = D
        hand_data = _NS(body_pos_w=body[..., :3].clone(), body_quat_w=body[..., 3:7], body_vel_w=body[..., 7:13],
                        joint_pos=q.clone(), joint_vel=qd.clone(), default_joint_pos=torch.randn(n, D, generator=g) * 0.3,
                        default_joint_vel=torch.randn(n, D, generator=g) * 0.1, joint_pos_target=torch.randn(n, D, generator=g),
                        _previous_joint_vel=torch.randn(n, D, generator=g), joint_acc=torch.randn(n, D, generator=g),
                        _body_state_w=_NS(timestamp=1.0), _body_link_state_w=_NS(timestamp=1.0), _body_com_state_w=_NS(timestamp=1.0))
        hand = _NS(data=hand_data, _data=hand_data, device="cpu", root_physx_view=View(), _ALL_INDICES=torch.arange(n),
                   num_joints=D)
        for name in ("write_joint_state_to_sim", "set_joint_position_target"):
            setattr(hand, name, types.MethodType(getattr(art.Articulation, name), hand))
        obj_data = _NS(root_pos_w=obj[:, :3], root_quat_w=obj[:, 3:7], root_vel_w=obj[:, 7:13], root_lin_vel_w=obj[:, 7:10],
                       root_ang_vel_w=obj[:, 10:13], root_state_w=obj.clone(),
                       default_root_state=torch.cat([torch.tensor([0.0, -0.39, 0.6]).repeat(n, 1), rand_quat(g, n),
                                                     torch.randn(n, 6, generator=g)], dim=1),
                       body_acc_w=torch.randn(n, 1, 6, generator=g))
        cube = _NS(data=obj_data, _data=obj_data, device="cpu", root_physx_view=View(), _ALL_INDICES=torch.arange(n))
        for name in ("write_root_pose_to_sim", "write_root_velocity_to_sim"):
            setattr(cube, name, types.MethodType(getattr(rig.RigidObject, name), cube))
        env.hand, env.object, env.scene = hand, cube, _NS(env_origins=origins, reset=lambda env_ids: None)
        env.hand_dof_lower_limits, env.hand_dof_upper_limits = lower, upper
        env.in_hand_pos, env.goal_rot = in_hand, goal.clone()
        env.goal_pos = torch.zeros(n, 3)
        env.x_unit_tensor = torch.tensor([1.0, 0, 0]).repeat(n, 1)
        env.y_unit_tensor = torch.tensor([0, 1.0, 0]).repeat(n, 1)
        env.fingertip_force_sensors = forces[:, fingers]
        ih.InHandManipulationEnv._compute_intermediate_values(env)
        out.update({p + "origins": origins, p + "obj": obj, p + "goal": goal, p + "in_hand": in_hand, p + "actions": actions,
                    p + "lower": lower, p + "upper": upper, p + "q": q, p + "qd": qd, p + "body": body, p + "forces": forces,
                    p + "fingers": torch.tensor(fingers),
                    p + "reduced_obs": ih.InHandManipulationEnv.compute_reduced_observations(env),
                    p + "full_state": ih.InHandManipulationEnv.compute_full_state(env)})
        # -- _reset_target_pose for a subset
        ids = torch.tensor(sorted(torch.randperm(n, generator=g)[: max(2, n // 3)].tolist()))
        u = torch.rand(n, 7 + 2 * D, generator=g)
        env.reset_goal_buf = torch.rand(n, generator=g) < 0.5
        out.update({p + "ids": ids, p + "u": u, p + "reset_goal_in": env.reset_goal_buf.clone()})
        with _RandQueue(rand=[u[ids, :2]]):
            ih.InHandManipulationEnv._reset_target_pose(env, ids)
        out.update({p + "goal_after_target_pose": env.goal_rot.clone(), p + "reset_goal_after_target_pose": env.reset_goal_buf.clone()})
        # -- _reset_idx for another subset (InHandManipulationEnv._reset_idx -> the real DirectRLEnv._reset_idx)
        ids2 = torch.tensor(sorted(torch.randperm(n, generator=g)[: max(2, n // 2)].tolist()))
        env.goal_rot = goal.clone()
        env.reset_goal_buf = out[p + "reset_goal_in"].clone()
        env.prev_targets, env.cur_targets, env.hand_dof_targets = (torch.randn(n, D, generator=g) for _ in range(3))
        env.successes = torch.randint(0, 5, (n,), generator=g).float()
        env.episode_length_buf = torch.randint(1, 500, (n,), generator=g)
        env._compute_intermediate_values = lambda: None
        out.update({p + "ids2": ids2, p + "object_default": obj_data.default_root_state.clone(),
                    p + "default_joint_pos": hand_data.default_joint_pos.clone(), p + "default_joint_vel": hand_data.default_joint_vel.clone(),
                    p + "in_prev_targets": env.prev_targets.clone(), p + "in_cur_targets": env.cur_targets.clone(),
                    p + "in_hand_dof_targets": env.hand_dof_targets.clone(), p + "in_joint_pos_target": hand_data.joint_pos_target.clone(),
                    p + "in_prev_joint_vel": hand_data._previous_joint_vel.clone(), p + "in_joint_acc": hand_data.joint_acc.clone(),
                    p + "in_successes": env.successes.clone(), p + "in_ep": env.episode_length_buf.clone()})
        with _RandQueue(rand=[u[ids2, 0:2], u[ids2, 2:5], u[ids2, 5:7], u[ids2, 7:7 + D], u[ids2, 7 + D:]]):
            ih.InHandManipulationEnv._reset_idx(env, ids2)
        out.update({p + "r_goal": env.goal_rot, p + "r_reset_goal": env.reset_goal_buf, p + "r_object_state": obj_data.root_state_w,
                    p + "r_sim_transforms": cube.root_physx_view.calls["set_transforms"][0],
                    p + "r_sim_velocities": cube.root_physx_view.calls["set_velocities"][0],
                    p + "r_prev_targets": env.prev_targets, p + "r_cur_targets": env.cur_targets,
                    p + "r_hand_dof_targets": env.hand_dof_targets, p + "r_joint_pos_target": hand_data.joint_pos_target,
                    p + "r_joint_pos": hand_data.joint_pos, p + "r_joint_vel": hand_data.joint_vel,
                    p + "r_prev_joint_vel": hand_data._previous_joint_vel, p + "r_joint_acc": hand_data.joint_acc,
                    p + "r_sim_dof_pos": hand.root_physx_view.calls["set_dof_positions"][0],
                    p + "r_sim_dof_vel": hand.root_physx_view.calls["set_dof_velocities"][0],
                    p + "r_successes": env.successes, p + "r_ep": env.episode_length_buf})
    save("inhand_reset", **out)


class _RandQueue:
    """Replaces torch.rand / torch.randint_like while a reference function runs: hands out the prepared tensors in call
    order (shape-checked), so the reference consumes exactly the numbers the fixture records."""

    def __init__(self, rand=(), randint=()):
        self.rand, self.randint = list(rand), list(randint)

    def __enter__(self):
        self._rand, self._randint_like = torch.rand, torch.randint_like

        def rand(*size, **kw):
            size = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
            t = self.rand.pop(0)
            assert tuple(t.shape) == size, (tuple(t.shape), size)
            return t.clone()

        def randint_like(x, *a, **kw):
            t = self.randint.pop(0)
            assert t.shape == x.shape, (t.shape, x.shape)
            return t.clone().to(x.dtype)

        torch.rand, torch.randint_like = rand, randint_like
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randint_like = self._rand, self._randint_like
        assert not self.rand and not self.randint, "the reference drew fewer numbers than prepared"


def gen_events():
    """Reset / interval event terms, the terrain-level curriculum and the running standard scaler, run by the REFERENCE
    code (events.py:610-866 incl. Articulation.write_*_to_sim, event_manager.py:147-225, curriculums.py:26-55 +
    terrain_importer.py:307-324, my_running_standard_scaler.py) on duck-typed assets with injected random draws."""
    ev = R.load_by_path("_ref_events", os.path.join(R.LAB, "isaaclab", "envs", "mdp", "events.py"))
    art = R.ref_import("isaaclab.assets.articulation.articulation")
    em = R.ref_import("isaaclab.managers.event_manager")
    g = torch.Generator().manual_seed(4242)
    n, j, b = 48, 12, 17
    out = {}

    class View:
        def __init__(self):
            self.calls = {}

        def _set(self, name):
            def f(data, indices=None):
                self.calls[name] = (data.clone(), None if indices is None else torch.as_tensor(indices).clone())
            return f

        def __getattr__(self, name):
            if name.startswith("set_"):
                return self._set(name)
            raise AttributeError(name)

    def make_asset():
        d = _NS(
            default_root_state=torch.cat([torch.randn(n, 3, generator=g), rand_quat(g, n), torch.randn(n, 6, generator=g) * 0.1], dim=1),
            root_state_w=torch.randn(n, 13, generator=g), body_acc_w=torch.randn(n, b, 6, generator=g),
            default_joint_pos=torch.randn(n, j, generator=g) * 0.6, default_joint_vel=torch.randn(n, j, generator=g) * 0.3,
            soft_joint_pos_limits=torch.stack([-0.9 * torch.ones(n, j), 1.1 * torch.ones(n, j)], dim=-1),
            soft_joint_vel_limits=torch.full((n, j), 0.4),
            joint_pos=torch.randn(n, j, generator=g), joint_vel=torch.randn(n, j, generator=g),
            _previous_joint_vel=torch.randn(n, j, generator=g), joint_acc=torch.randn(n, j, generator=g),
            _body_state_w=_NS(timestamp=1.0), _body_link_state_w=_NS(timestamp=1.0), _body_com_state_w=_NS(timestamp=1.0))
        d.root_vel_w = d.root_state_w[:, 7:13]
        asset = _NS(_data=d, data=d, device="cpu", root_physx_view=View(), _ALL_INDICES=torch.arange(n))
        for name in ("write_root_pose_to_sim", "write_root_velocity_to_sim", "write_joint_state_to_sim"):
            setattr(asset, name, types.MethodType(getattr(art.Articulation, name), asset))
        return asset, d

    origins = torch.randn(n, 3, generator=g) * 5.0
    ids = torch.tensor(sorted(torch.randperm(n, generator=g)[:19].tolist()))
    out["env_origins"], out["ids"] = origins, ids

    # -- reset_root_state_uniform
    asset, d = make_asset()
    env = _NS(scene=_NS(env_origins=origins, **{"__getitem__": None}), num_envs=n, device="cpu")
    scene = type("Scene", (dict,), {})(robot=asset)
    scene.env_origins = origins
    env.scene = scene
    pose_range = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "yaw": (-3.14, 3.14), "roll": (-0.3, 0.2)}
    vel_range = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "z": (-0.5, 0.5), "roll": (-0.5, 0.5), "pitch": (-0.5, 0.5), "yaw": (-0.5, 0.5)}
    u = torch.rand(n, 12, generator=g)
    out.update(root_default=d.default_root_state.clone(), root_state_in=d.root_state_w.clone(), root_u=u)
    with _RandQueue(rand=[u[ids, :6], u[ids, 6:]]):
        ev.reset_root_state_uniform(env, ids, pose_range, vel_range)
    out.update(root_state_out=d.root_state_w.clone(), root_body_acc_out=d.body_acc_w.clone(),
               root_sim_transforms=asset.root_physx_view.calls["set_root_transforms"][0],
               root_sim_velocities=asset.root_physx_view.calls["set_root_velocities"][0])

    # -- reset_joints_by_scale / by_offset
    for tag, fn, prange, vrange in (("scale", ev.reset_joints_by_scale, (0.5, 1.5), (0.0, 0.0)),
                                    ("offset", ev.reset_joints_by_offset, (-0.4, 0.7), (-0.6, 0.5))):
        asset, d = make_asset()
        scene["robot"] = asset
        u = torch.rand(n, 2 * j, generator=g)
        out.update({f"joints_{tag}_default_pos": d.default_joint_pos.clone(), f"joints_{tag}_default_vel": d.default_joint_vel.clone(),
                    f"joints_{tag}_pos_in": d.joint_pos.clone(), f"joints_{tag}_vel_in": d.joint_vel.clone(),
                    f"joints_{tag}_u": u, f"joints_{tag}_pos_limits": d.soft_joint_pos_limits.clone(),
                    f"joints_{tag}_vel_limits": d.soft_joint_vel_limits.clone()})
        with _RandQueue(rand=[u[ids, :j], u[ids, j:]]):
            fn(env, ids, prange, vrange)
        out.update({f"joints_{tag}_pos_out": d.joint_pos.clone(), f"joints_{tag}_vel_out": d.joint_vel.clone(),
                    f"joints_{tag}_prev_vel_out": d._previous_joint_vel.clone(), f"joints_{tag}_acc_out": d.joint_acc.clone(),
                    f"joints_{tag}_sim_pos": asset.root_physx_view.calls["set_dof_positions"][0],
                    f"joints_{tag}_sim_vel": asset.root_physx_view.calls["set_dof_velocities"][0]})

    # -- push_by_setting_velocity
    asset, d = make_asset()
    scene["robot"] = asset
    u = torch.rand(n, 6, generator=g)
    out.update(push_state_in=d.root_state_w.clone(), push_u=u)
    with _RandQueue(rand=[u[ids]]):
        ev.push_by_setting_velocity(env, ids, {"x": (-0.5, 0.5), "y": (-0.5, 0.5)})
    out.update(push_state_out=d.root_state_w.clone(), push_sim_velocities=asset.root_physx_view.calls["set_root_velocities"][0])

    # -- EventManager.apply(mode="interval"), per-env timers (event_manager.py:199-225)
    calls = []
    mgr = em.EventManager.__new__(em.EventManager)
    mgr._env = _NS(num_envs=n, device="cpu")
    cfg = _NS(func=lambda env, env_ids, **kw: calls.append(env_ids.clone()), params={}, interval_range_s=(10.0, 15.0),
              is_global_time=False, min_step_count_between_reset=0)
    mgr._mode_term_names = {"interval": ["push"]}
    mgr._mode_term_cfgs = {"interval": [cfg]}
    tl = torch.rand(n, generator=g) * 0.07
    tl[3] = 0.02                     # exactly dt: fires (0.02 - 0.02 < 1e-6)
    mgr._interval_term_time_left = [tl.clone()]
    out["interval_time_left_in"] = tl.clone()
    fired_all, tl_all, u_all = [], [], []
    for k in range(4):
        u = torch.rand(n, generator=g)
        valid = ((mgr._interval_term_time_left[0] - 0.02) < 1e-6).nonzero().flatten()
        calls.clear()
        with _RandQueue(rand=[u[valid]] if len(valid) else []):
            mgr.apply(mode="interval", dt=0.02)
        fired = torch.zeros(n, dtype=torch.bool)
        if calls:
            fired[calls[0]] = True
        fired_all.append(fired); tl_all.append(mgr._interval_term_time_left[0].clone()); u_all.append(u)
    out.update(interval_fired=torch.stack(fired_all), interval_time_left_out=torch.stack(tl_all), interval_u=torch.stack(u_all))

    # -- terrain_levels_vel + TerrainImporter.update_env_origins
    cur = R.load_by_path("_ref_curriculums", os.path.join(R.TASKS, "manager_based", "locomotion", "velocity", "mdp", "curriculums.py"))
    ti = R.ref_import("isaaclab.terrains.terrain_importer")
    rows, cols = 4, 5
    t_origins = torch.randn(rows, cols, 3, generator=g) * 8.0
    levels = torch.randint(0, rows, (n,), generator=g)
    types_ = torch.randint(0, cols, (n,), generator=g)
    eo = t_origins[levels, types_].clone()
    terrain = _NS(cfg=_NS(terrain_generator=_NS(size=(8.0, 8.0))), terrain_origins=t_origins, terrain_levels=levels.clone(),
                  terrain_types=types_, max_terrain_level=rows, env_origins=eo)
    terrain.update_env_origins = types.MethodType(ti.TerrainImporter.update_env_origins, terrain)
    root = torch.zeros(n, 13)
    root[:, :2] = eo[:, :2] + (torch.rand(n, 2, generator=g) - 0.5) * 14.0
    command = (torch.rand(n, 3, generator=g) - 0.5) * 1.2
    robot = _NS(data=_NS(root_pos_w=root[:, :3]))
    scene2 = type("Scene", (dict,), {})(robot=robot)
    scene2.terrain, scene2.env_origins = terrain, eo
    env2 = _NS(scene=scene2, command_manager=_NS(get_command=lambda name: command), max_episode_length_s=20.0)
    rnd = torch.randint(0, rows, (n,), generator=g)
    out.update(cur_terrain_origins=t_origins, cur_levels_in=levels.clone(), cur_types=types_, cur_env_origins_in=eo.clone(),
               cur_root_state=root, cur_command=command, cur_random_levels=rnd)
    with _RandQueue(randint=[rnd[ids]]):
        state = cur.terrain_levels_vel(env2, ids)
    out.update(cur_levels_out=terrain.terrain_levels.clone(), cur_env_origins_out=eo.clone(), cur_state=state.reshape(1))

    # -- running standard scaler (my_code/models/my_running_standard_scaler.py)
    skrl = R._stub("skrl")
    skrl.config = _NS(torch=_NS(parse_device=lambda d: torch.device("cpu")))
    sp = R._stub("skrl.utils.spaces.torch")
    sp.compute_space_size = lambda size, occupied_size=True: int(size)
    R._stub("skrl.utils"); R._stub("skrl.utils.spaces")
    rs = R.load_by_path("_ref_scaler", os.path.join(R.REF_ROOT, "my_code", "models", "my_running_standard_scaler.py"))
    dsz = 29
    sc = rs.my_RunningStandardScaler(dsz, device="cpu")
    scale = torch.rand(dsz, generator=g) * 20.0 + 0.01
    xs = [torch.randn(65, dsz, generator=g) * scale + torch.randn(dsz, generator=g) * 4.0 for _ in range(3)]
    xs[1][2, :4] = 1e6
    ys = []
    for x in xs:
        ys.append(sc._compute(x, train=True).clone())
    x3 = torch.stack(xs[:2])
    y3 = sc._compute(x3, train=True).clone()
    inv = sc._compute(ys[0], inverse=True).clone()
    out.update(scaler_x=torch.stack(xs), scaler_y=torch.stack(ys), scaler_y3=y3, scaler_inverse=inv,
               scaler_mean=sc.running_mean.clone(), scaler_var=sc.running_variance.clone(),
               scaler_count=sc.current_count.reshape(1).clone())
    save("events", **out)


def gen_imu():
    """Imu._update_buffers_impl (imu.py:142-182) by the reference class on a duck-typed rigid-body view: three updates with
    moving body state, the middle one for a subset of envs."""
    imu = R.ref_import("isaaclab.sensors.imu.imu")
    g = torch.Generator().manual_seed(777)
    n = 40
    out = {}
    coms = torch.cat([torch.randn(n, 3, generator=g) * 0.05, rand_quat(g, n)], dim=1)
    frames = [(torch.cat([torch.randn(n, 3, generator=g), rand_quat(g, n)[:, [1, 2, 3, 0]]], dim=1),
               torch.randn(n, 6, generator=g)) for _ in range(3)]
    state = {"k": 0}
    view = _NS(count=n, get_transforms=lambda: frames[state["k"]][0].clone(), get_coms=lambda: coms.clone(),
               get_velocities=lambda: frames[state["k"]][1].clone())
    s = imu.Imu.__new__(imu.Imu)
    s._view, s._device, s._num_envs = view, "cpu", n
    s._data = _NS()
    s._initialize_handle = s._invalidate_initialize_handle = s._prim_deletion_callback_id = None   # SensorBase.__del__
    s._debug_vis_handle = None
    s.cfg = _NS(offset=_NS(pos=(0.1, -0.05, 0.2), rot=tuple(rand_quat(g, 1)[0].tolist())), gravity_bias=(0.0, 0.0, 9.81))
    imu.Imu._initialize_buffers_impl(s)
    s._dt = 0.005
    out.update(coms=coms, offset_pos=s._offset_pos_b, offset_quat=s._offset_quat_b, transforms=torch.stack([f[0] for f in frames]),
               velocities=torch.stack([f[1] for f in frames]))
    subset = torch.tensor(sorted(torch.randperm(n, generator=g)[:17].tolist()))
    out["subset"] = subset
    for k, ids in enumerate((torch.arange(n), subset, torch.arange(n))):
        state["k"] = k
        imu.Imu._update_buffers_impl(s, ids)
        for name in ("pos_w", "quat_w", "lin_vel_b", "ang_vel_b", "lin_acc_b", "ang_acc_b"):
            out[f"{name}_{k}"] = getattr(s._data, name).clone()
        out[f"prev_lin_{k}"], out[f"prev_ang_{k}"] = s._prev_lin_vel_w.clone(), s._prev_ang_vel_w.clone()
    save("imu", **out)


def gen_frame_transformer():
    """FrameTransformer._update_buffers_impl (frame_transformer.py:322-385) by the reference class with hand-made index
    tables (the USD walk that builds them is out of scope): 3 bodies per env in the view, 4 target frames on 2 bodies."""
    ft = R.ref_import("isaaclab.sensors.frame_transformer.frame_transformer")
    g = torch.Generator().manual_seed(31337)
    n, bodies, nf = 24, 3, 4
    out = {}
    m = n * bodies
    # the view lists bodies body-major (all envs of body 0, then body 1, ...): per_env_indices makes it env-major
    per_env = torch.arange(m).reshape(bodies, n).t().reshape(-1)
    source_ids = torch.arange(n) * bodies + 0
    target_ids = (torch.arange(n)[:, None] * bodies + torch.tensor([1, 2])[None, :]).reshape(-1)      # 2 unique bodies
    dup = (torch.arange(n)[:, None] * 2 + torch.tensor([0, 0, 1, 1])[None, :]).reshape(-1)            # 4 frames
    s = ft.FrameTransformer.__new__(ft.FrameTransformer)
    for name in ("_initialize_handle", "_invalidate_initialize_handle", "_prim_deletion_callback_id", "_debug_vis_handle"):
        setattr(s, name, None)
    s._num_envs, s._device = n, "cpu"
    s._target_frame_names = ["a", "b", "c", "d"]
    s._per_env_indices, s._source_frame_body_ids, s._target_frame_body_ids, s._duplicate_frame_indices = per_env, source_ids, target_ids, dup
    s._apply_source_frame_offset = s._apply_target_frame_offset = True
    s._source_frame_offset_pos = torch.randn(1, 3, generator=g).repeat(n, 1) * 0.1
    s._source_frame_offset_quat = rand_quat(g, 1).repeat(n, 1)
    s._target_frame_offset_pos = (torch.randn(nf, 3, generator=g) * 0.1).repeat(n, 1)
    s._target_frame_offset_quat = rand_quat(g, nf).repeat(n, 1)
    z = lambda *shape: torch.zeros(shape)
    s._data = _NS(source_pos_w=z(n, 3), source_quat_w=z(n, 4), target_pos_w=z(n, nf, 3), target_quat_w=z(n, nf, 4),
                  target_pos_source=z(n, nf, 3), target_quat_source=z(n, nf, 4))
    tr = torch.cat([torch.randn(m, 3, generator=g), rand_quat(g, m)[:, [1, 2, 3, 0]]], dim=1)
    s._frame_physx_view = _NS(get_transforms=lambda: tr.clone())
    out.update(transforms=tr, per_env_indices=per_env, source_frame_body_ids=source_ids, target_frame_body_ids=target_ids,
               duplicate_frame_indices=dup, source_offset_pos=s._source_frame_offset_pos, source_offset_quat=s._source_frame_offset_quat,
               target_offset_pos=s._target_frame_offset_pos, target_offset_quat=s._target_frame_offset_quat)
    ft.FrameTransformer._update_buffers_impl(s, torch.arange(n))
    for name in ("source_pos_w", "source_quat_w", "target_pos_w", "target_quat_w", "target_pos_source", "target_quat_source"):
        out[name] = getattr(s._data, name).clone()
    s._apply_source_frame_offset = s._apply_target_frame_offset = False
    ft.FrameTransformer._update_buffers_impl(s, torch.arange(n))
    for name in ("source_pos_w", "target_pos_w", "target_pos_source", "target_quat_source"):
        out["nooff_" + name] = getattr(s._data, name).clone()
    save("frame_transformer", **out)


def gen_camera():
    """RayCasterCamera._update_buffers_impl (ray_caster_camera.py:257-318) by the reference class.  warp is absent, so its
    `raycast_mesh` is replaced by the repo's CPU ray-cast oracle on the same mesh (the cast itself is pinned elsewhere,
    DESIGN section 1); what this fixture pins is the camera pose, the ray rotation and the depth post-processing."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from isaaclab_amd.terrains import height_field as HF
    from oracle.raycast_np import GridCaster, tri_soup

    cam = R.ref_import("isaaclab.sensors.ray_caster.ray_caster_camera")
    g = torch.Generator().manual_seed(2024)
    np.random.seed(5)
    meshes, _ = HF.pyramid_stairs_terrain(0.8, HF.HfPyramidStairsTerrainCfg(size=(6.0, 6.0), border_width=0.5, slope_threshold=0.75,
                                                                             platform_width=1.5))
    verts, faces = meshes[0].vertices, meshes[0].faces
    caster = GridCaster(tri_soup(verts, faces))

    def raycast_mesh(starts, dirs, mesh=None, max_dist=1e6, return_distance=False, return_normal=False, **kw):
        shape = starts.shape[:-1]
        hits, dist, normal, _ = caster.cast(starts.reshape(-1, 3).numpy(), dirs.reshape(-1, 3).numpy(), max_dist)
        return (torch.from_numpy(hits).reshape(*shape, 3), torch.from_numpy(dist).reshape(*shape) if return_distance else None,
                torch.from_numpy(normal).reshape(*shape, 3) if return_normal else None, None)

    cam.raycast_mesh = raycast_mesh
    n, h, w = 6, 6, 8
    r = h * w
    out = dict(vertices=verts, faces=faces.astype(np.int32))
    view_pose = torch.cat([torch.tensor([3.0, 3.0, 2.5]) + torch.randn(n, 3, generator=g) * 0.5, rand_quat(g, n)], dim=1)   # wxyz
    offset_pos = torch.randn(n, 3, generator=g) * 0.1
    offset_quat = rand_quat(g, n)
    dirs = torch.randn(n, r, 3, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    starts = torch.zeros(n, r, 3)
    out.update(view_pose=view_pose, offset_pos=offset_pos, offset_quat=offset_quat, ray_directions=dirs, ray_starts=starts)
    for behavior in ("max", "zero", "none"):
        s = cam.RayCasterCamera.__new__(cam.RayCasterCamera)
        for name in ("_initialize_handle", "_invalidate_initialize_handle", "_prim_deletion_callback_id", "_debug_vis_handle"):
            setattr(s, name, None)
        s._device, s._frame = "cpu", torch.zeros(n, dtype=torch.long)
        s._offset_pos, s._offset_quat = offset_pos.clone(), offset_quat.clone()
        s._compute_view_world_poses = lambda env_ids: (view_pose[env_ids, :3].clone(), view_pose[env_ids, 3:].clone())
        s.ray_starts, s.ray_directions, s.num_rays = starts, dirs, r
        s.meshes = {"/m": None}
        s.cfg = _NS(mesh_prim_paths=["/m"], data_types=["distance_to_image_plane", "distance_to_camera", "normals"],
                    depth_clipping_behavior=behavior, max_distance=2.5, pattern_cfg=_NS(height=h, width=w))
        s._data = _NS(pos_w=torch.zeros(n, 3), quat_w_world=torch.zeros(n, 4),
                      output={"distance_to_image_plane": torch.zeros(n, h, w, 1), "distance_to_camera": torch.zeros(n, h, w, 1),
                              "normals": torch.zeros(n, h, w, 3)})
        cam.RayCasterCamera._update_buffers_impl(s, torch.arange(n))
        out.update({f"{behavior}_image_plane": s._data.output["distance_to_image_plane"].clone(),
                    f"{behavior}_camera": s._data.output["distance_to_camera"].clone(),
                    f"{behavior}_normals": s._data.output["normals"].clone()})
        if behavior == "max":
            out.update(pos_w=s._data.pos_w.clone(), quat_w_world=s._data.quat_w_world.clone(), ray_hits_w=s.ray_hits_w.clone())
    save("camera", **out)


def gen_camera_class():
    """RayCasterCamera as a class (ray_caster_camera.py:121-255, 370-430), the pinhole pattern (patterns.py:61-104,
    patterns_cfg.py:120-172) and the camera-convention helpers (math.py:1429-1561), all run by the REFERENCE code: rays of
    a pinhole cfg with aperture offsets, intrinsics, the offsets `set_world_poses` / `set_world_poses_from_view` compute,
    and one update after them (raycast_mesh served by the CPU ray-cast oracle, as in gen_camera)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from isaaclab_amd.terrains import height_field as HF
    from oracle.raycast_np import GridCaster, tri_soup

    m = R.ref_math()
    cam = R.ref_import("isaaclab.sensors.ray_caster.ray_caster_camera")
    pat = R.ref_import("isaaclab.sensors.ray_caster.patterns.patterns")
    pcfg = R.ref_import("isaaclab.sensors.ray_caster.patterns.patterns_cfg")
    g = torch.Generator().manual_seed(77)
    out = {}
    # -- conventions
    q = rand_quat(g, 7)
    out["conv_q"] = q
    for origin in ("opengl", "ros", "world"):
        for target in ("opengl", "ros", "world"):
            out[f"conv_{origin}_{target}"] = m.convert_camera_frame_orientation_convention(q.clone(), origin=origin, target=target)
    eyes = torch.randn(6, 3, generator=g) * 2.0
    targets = torch.randn(6, 3, generator=g)
    targets[0] = eyes[0] + torch.tensor([0.0, 0.0, -1.5])          # straight down: the degenerate x-axis branch
    out.update(view_eyes=eyes, view_targets=targets)
    out["view_rot_Z"] = m.create_rotation_matrix_from_view(eyes, targets, up_axis="Z")
    out["view_rot_Y"] = m.create_rotation_matrix_from_view(eyes, targets, up_axis="Y")
    # -- pattern cfg from an intrinsic matrix, intrinsics, rays
    K = [38.0, 0.0, 4.3, 0.0, 35.0, 2.6, 0.0, 0.0, 1.0]
    pc = pcfg.PinholeCameraPatternCfg.from_intrinsic_matrix(K, width=8, height=6, focal_length=12.0)
    out["pc_fields"] = np.array([pc.focal_length, pc.horizontal_aperture, pc.vertical_aperture, pc.horizontal_aperture_offset,
                                 pc.vertical_aperture_offset, pc.width, pc.height], np.float64)
    # -- the class
    np.random.seed(5)
    meshes, _ = HF.pyramid_stairs_terrain(0.8, HF.HfPyramidStairsTerrainCfg(size=(6.0, 6.0), border_width=0.5, slope_threshold=0.75,
                                                                             platform_width=1.5))
    verts, faces = meshes[0].vertices, meshes[0].faces
    caster = GridCaster(tri_soup(verts, faces))

    def raycast_mesh(starts, dirs, mesh=None, max_dist=1e6, return_distance=False, return_normal=False, **kw):
        shape = starts.shape[:-1]
        hits, dist, normal, _ = caster.cast(starts.reshape(-1, 3).numpy(), dirs.reshape(-1, 3).numpy(), max_dist)
        return (torch.from_numpy(hits).reshape(*shape, 3), torch.from_numpy(dist).reshape(*shape) if return_distance else None,
                torch.from_numpy(normal).reshape(*shape, 3) if return_normal else None, None)

    cam.raycast_mesh = raycast_mesh
    cam.stage_utils = _NS(get_stage_up_axis=lambda: "Z")
    n = 5
    view_pose = torch.cat([torch.tensor([3.0, 3.0, 2.0]) + torch.randn(n, 3, generator=g) * 0.4, rand_quat(g, n)], dim=1)   # wxyz
    s = cam.RayCasterCamera.__new__(cam.RayCasterCamera)
    for name in ("_initialize_handle", "_invalidate_initialize_handle", "_prim_deletion_callback_id", "_debug_vis_handle"):
        setattr(s, name, None)
    s._device = "cpu"
    s._view = _NS(count=n)
    s._data = cam.CameraData()
    s.cfg = _NS(mesh_prim_paths=["/m"], data_types=["distance_to_image_plane", "distance_to_camera", "normals"],
                depth_clipping_behavior="max", max_distance=4.0, pattern_cfg=pc,
                offset=_NS(pos=(0.1, -0.05, 0.2), rot=(0.9238795, 0.0, 0.3826834, 0.0), convention="ros"))
    s._compute_view_world_poses = lambda env_ids: (view_pose[env_ids, :3].clone(), view_pose[env_ids, 3:].clone())
    s.meshes = {"/m": None}
    s._initialize_rays_impl()
    out.update(vertices=verts, faces=faces.astype(np.int32), cls_view_pose=view_pose,
               cls_intrinsics=s._data.intrinsic_matrices.clone(), cls_ray_starts=s.ray_starts.clone(),
               cls_ray_directions=s.ray_directions.clone(), cls_offset_pos0=s._offset_pos.clone(), cls_offset_quat0=s._offset_quat.clone())
    # configuration calls: world pose in the ros convention for envs 0..2, look-at for envs 3..4
    set_pos = torch.tensor([3.0, 3.0, 2.6]) + torch.randn(3, 3, generator=g) * 0.3
    set_quat = rand_quat(g, 3)
    s.set_world_poses(set_pos, set_quat, env_ids=torch.tensor([0, 1, 2]), convention="ros")
    eyes2 = torch.tensor([[2.0, 2.5, 2.8], [4.0, 3.5, 2.4]])
    targets2 = torch.tensor([[3.0, 3.0, 0.3], [3.0, 3.0, 0.5]])
    s.set_world_poses_from_view(eyes2, targets2, env_ids=torch.tensor([3, 4]))
    out.update(cls_set_pos=set_pos, cls_set_quat=set_quat, cls_eyes=eyes2, cls_targets=targets2,
               cls_offset_pos=s._offset_pos.clone(), cls_offset_quat=s._offset_quat.clone(),
               cls_pos_w_cfg=s._data.pos_w.clone(), cls_quat_w_cfg=s._data.quat_w_world.clone())
    # new intrinsics for two envs
    K2 = torch.tensor([[[30.0, 0.0, 4.0], [0.0, 31.0, 3.0], [0.0, 0.0, 1.0]]]).repeat(2, 1, 1)
    s.set_intrinsic_matrices(K2, focal_length=2.0, env_ids=torch.tensor([1, 4]))
    out.update(cls_K2=K2, cls_ray_directions2=s.ray_directions.clone())
    s._frame = torch.zeros(n, dtype=torch.long)
    s._update_buffers_impl(torch.arange(n))
    out.update(cls_pos_w=s._data.pos_w.clone(), cls_quat_w=s._data.quat_w_world.clone(), cls_hits=s.ray_hits_w.clone(),
               cls_image_plane=s._data.output["distance_to_image_plane"].clone(),
               cls_camera=s._data.output["distance_to_camera"].clone(), cls_normals=s._data.output["normals"].clone())
    save("camera_class", **out)


def gen_terrains():
    """Height-field sub-terrains by the reference (hf_terrains.py + height_field/utils.py), trimesh.Trimesh replaced by a
    recorder of (vertices, faces); np.random seeded before the random terrain."""
    hf = R.ref_import("isaaclab.terrains.height_field.hf_terrains")
    ut = R.ref_import("isaaclab.terrains.height_field.utils")

    class Rec:
        def __init__(self, vertices=None, faces=None, **kw):
            self.vertices, self.faces = np.asarray(vertices), np.asarray(faces)

    ut.trimesh = _NS(Trimesh=Rec)
    base = dict(size=(6.0, 5.0), horizontal_scale=0.1, vertical_scale=0.005, slope_threshold=0.75)
    cases = {
        "random": (hf.random_uniform_terrain, 0.3, dict(border_width=0.25, noise_range=(0.02, 0.10), noise_step=0.02, downsampled_scale=None)),
        "random_ds": (hf.random_uniform_terrain, 0.3, dict(border_width=0.0, noise_range=(-0.05, 0.05), noise_step=0.01, downsampled_scale=0.2)),
        "slope": (hf.pyramid_sloped_terrain, 0.7, dict(border_width=0.25, slope_range=(0.0, 0.4), platform_width=2.0, inverted=False)),
        "slope_inv": (hf.pyramid_sloped_terrain, 0.45, dict(border_width=0.25, slope_range=(0.0, 0.4), platform_width=2.0, inverted=True)),
        "stairs": (hf.pyramid_stairs_terrain, 0.6, dict(border_width=0.5, step_height_range=(0.05, 0.23), step_width=0.3, platform_width=1.5, inverted=False)),
        "stairs_inv": (hf.pyramid_stairs_terrain, 0.2, dict(border_width=0.5, step_height_range=(0.05, 0.23), step_width=0.3, platform_width=1.5, inverted=True)),
    }
    out = {}
    for k, (name, (fn, difficulty, extra)) in enumerate(cases.items()):
        cfg = _NS(**base, **extra)
        np.random.seed(100 + k)
        meshes, origin = fn(difficulty, cfg)
        out[name + "_vertices"] = meshes[0].vertices.astype(np.float32)
        out[name + "_faces"] = meshes[0].faces.astype(np.int32)
        out[name + "_origin"] = np.asarray(origin, np.float64)
        out[name + "_difficulty"] = np.array([difficulty])
    save("terrains", **out)


def gen_mesh_terrains():
    """Mesh sub-terrains and the whole generator run by the REFERENCE (trimesh/mesh_terrains.py, trimesh/utils.py,
    terrain_generator.py) on the trimesh stand-in of tests/golden/trimesh_standin.py (trimesh is absent: its box
    layout is restated there, everything else is the reference's own code).  Stored: triangle soups in face order
    (fp32, what convert_to_warp_mesh receives), origins, and for the full ROUGH_TERRAINS_CFG a digest + samples."""
    import hashlib
    import trimesh_standin

    sys.modules["trimesh"] = trimesh_standin.module()
    for name in ("isaaclab.terrains.trimesh", "isaaclab.terrains.trimesh.utils"):
        sys.modules.pop(name, None)
    # a plain package object for isaaclab.terrains (the attribute-swallowing stub would answer the circular
    # `import isaaclab.terrains.trimesh.mesh_terrains as ...` of mesh_terrains_cfg.py with a dummy)
    plain = types.ModuleType("isaaclab.terrains")
    plain.__path__ = list(sys.modules["isaaclab.terrains"].__path__)
    sys.modules["isaaclab.terrains"] = plain
    sys.modules["isaaclab"].terrains = plain
    mtc = R.ref_import("isaaclab.terrains.trimesh.mesh_terrains_cfg")
    hfc = R.ref_import("isaaclab.terrains.height_field.hf_terrains_cfg")
    tg = R.ref_import("isaaclab.terrains.terrain_generator")
    tgc = R.ref_import("isaaclab.terrains.terrain_generator_cfg")

    def soup(mesh):
        return np.asarray(mesh.vertices, np.float64)[np.asarray(mesh.faces)].astype(np.float32)

    out = {}
    cat = sys.modules["trimesh"].util.concatenate
    subs = {
        "stairs": (mtc.MeshPyramidStairsTerrainCfg(size=(8.0, 8.0), step_height_range=(0.05, 0.23), step_width=0.3,
                                                   platform_width=3.0, border_width=1.0, holes=False), 0.37),
        "stairs_inv": (mtc.MeshInvertedPyramidStairsTerrainCfg(size=(8.0, 8.0), step_height_range=(0.05, 0.23), step_width=0.3,
                                                               platform_width=3.0, border_width=1.0, holes=False), 0.81),
        "stairs_holes": (mtc.MeshPyramidStairsTerrainCfg(size=(6.0, 7.0), step_height_range=(0.1, 0.2), step_width=0.25,
                                                         platform_width=1.5, border_width=0.0, holes=True), 0.5),
        "boxes": (mtc.MeshRandomGridTerrainCfg(size=(8.0, 8.0), grid_width=0.45, grid_height_range=(0.05, 0.2), platform_width=2.0), 0.6),
        "boxes_holes": (mtc.MeshRandomGridTerrainCfg(size=(6.0, 6.0), grid_width=0.35, grid_height_range=(0.02, 0.1),
                                                     platform_width=1.5, holes=True), 0.25),
        "flat": (mtc.MeshPlaneTerrainCfg(size=(5.0, 4.0)), 0.0),
    }
    for k, (name, (cfg, difficulty)) in enumerate(subs.items()):
        torch.manual_seed(300 + k)
        res = cfg.function(difficulty, cfg)
        assert len(res) == 2, (name, type(res), len(res))
        meshes, origin = res
        out[name + "_soup"] = soup(cat(meshes))
        out[name + "_origin"] = np.asarray(origin, np.float64)
        out[name + "_difficulty"] = np.array([difficulty])

    def rough(num_rows, num_cols, curriculum, seed):
        return tgc.TerrainGeneratorCfg(
            size=(8.0, 8.0), border_width=20.0, num_rows=num_rows, num_cols=num_cols, horizontal_scale=0.1, vertical_scale=0.005,
            slope_threshold=0.75, use_cache=False, curriculum=curriculum, seed=seed,
            sub_terrains={
                "pyramid_stairs": mtc.MeshPyramidStairsTerrainCfg(proportion=0.2, step_height_range=(0.05, 0.23), step_width=0.3,
                                                                  platform_width=3.0, border_width=1.0, holes=False),
                "pyramid_stairs_inv": mtc.MeshInvertedPyramidStairsTerrainCfg(proportion=0.2, step_height_range=(0.05, 0.23),
                                                                              step_width=0.3, platform_width=3.0, border_width=1.0, holes=False),
                "boxes": mtc.MeshRandomGridTerrainCfg(proportion=0.2, grid_width=0.45, grid_height_range=(0.05, 0.2), platform_width=2.0),
                "random_rough": hfc.HfRandomUniformTerrainCfg(proportion=0.2, noise_range=(0.02, 0.10), noise_step=0.02, border_width=0.25),
                "hf_pyramid_slope": hfc.HfPyramidSlopedTerrainCfg(proportion=0.1, slope_range=(0.0, 0.4), platform_width=2.0, border_width=0.25),
                "hf_pyramid_slope_inv": hfc.HfInvertedPyramidSlopedTerrainCfg(proportion=0.1, slope_range=(0.0, 0.4), platform_width=2.0,
                                                                             border_width=0.25),
            })

    # reduced layouts, complete soups: curriculum (2 x 10: every sub-terrain type appears) and random (2 x 3)
    for name, (rows, cols, cur, seed) in {"gen_cur": (2, 10, True, 7), "gen_rand": (2, 3, False, 11)}.items():
        np.random.seed(seed + 1)
        torch.manual_seed(seed + 2)
        g = tg.TerrainGenerator(rough(rows, cols, cur, seed), device="cpu")
        sp = soup(g.terrain_mesh)
        # the flat interior of the height-field tiles compresses well; still keep the file small: fp32 soup of a
        # 2 x 10 layout is ~4.6 MB raw
        out[name + "_soup"] = sp
        out[name + "_origins"] = np.asarray(g.terrain_origins, np.float64)
        out[name + "_layout"] = np.array([rows, cols, int(cur), seed])
    # the benchmark terrain itself (ROUGH_TERRAINS_CFG, curriculum on as velocity_env_cfg.py:314-318 sets it, seed 42):
    # digest of the fp32 soup, its size, every 4099th triangle, the tile origins
    np.random.seed(42)
    torch.manual_seed(42)
    g = tg.TerrainGenerator(rough(10, 20, True, 42), device="cpu")
    sp = soup(g.terrain_mesh)
    out["rough_digest"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(sp).tobytes()).digest(), dtype=np.uint8).copy()
    out["rough_num_tris"] = np.array([sp.shape[0]])
    out["rough_sample"] = sp[::4099].copy()
    out["rough_origins"] = np.asarray(g.terrain_origins, np.float64)
    # env-origin layout of TerrainImporter (terrain_importer.py:328-349) on those origins
    ti = R.ref_import("isaaclab.terrains.terrain_importer")
    imp = _NS(cfg=_NS(max_init_terrain_level=5), device="cpu")
    torch.manual_seed(43)
    eo = ti.TerrainImporter._compute_env_origins_curriculum(imp, 4096, torch.from_numpy(g.terrain_origins).to(dtype=torch.float))
    out.update(rough_env_origins=eo, rough_levels=imp.terrain_levels, rough_types=imp.terrain_types)
    save("mesh_terrains", **out)


def gen_samplers():
    """Samplers of utils/math.py:1233-1421 run by the REFERENCE functions on recorded draws: torch.rand / torch.randn
    hand out prepared tensors while they run (for the scripted ones the module is loaded a second time with
    torch.jit.script as the identity, so the same Python bodies run eagerly).  Two draws have no torch.rand call to
    intercept and are fed the recorded numbers through their documented transformation: Tensor.uniform_(a, b) =
    a + (b - a) * u and torch.normal(mean, std) = mean + std * z."""
    script = torch.jit.script
    torch.jit.script = lambda f, *a, **k: f
    try:
        m = R.load_by_path("_ref_math_eager", os.path.join(R.LAB, "isaaclab", "utils", "math.py"))
    finally:
        torch.jit.script = script
    g = torch.Generator().manual_seed(1212)
    n = 257
    u = torch.rand(n, 2, generator=g)
    u[0, 0], u[1, 0], u[2, 0] = 0.0, 0.5, 1.0 - 2.0 ** -24
    z = torch.randn(n, 4, generator=g)
    out = {"u": u, "z": z}
    with _RandQueue(rand=[u[:, 0]]):
        out["uniform"] = m.sample_uniform(-0.7, 2.5, n, "cpu")
    lo = torch.rand(n, generator=g) - 1.5
    hi = lo + 0.1 + torch.rand(n, generator=g) * 3.0
    out["lo"], out["hi"] = lo, hi
    with _RandQueue(rand=[u[:, 0]]):
        out["uniform_tensor"] = m.sample_uniform(lo, hi, (n,), "cpu")
    with _RandQueue(rand=[u[:, 0]]):
        out["log_uniform"] = m.sample_log_uniform(1e-3, 10.0, n, "cpu")
    with _RandQueue(rand=[u[:, 0]]):
        out["triangle"] = m.sample_triangle(-1.0, 3.0, (n,), "cpu")
    with _RandQueue(rand=[u[:, 0]]):
        out["yaw_quat"] = m.random_yaw_orientation(n, "cpu")
    randn, normal, uniform_ = torch.randn, torch.normal, torch.Tensor.uniform_
    try:
        torch.randn = lambda *a, **k: z.clone()
        out["quat"] = m.random_orientation(n, "cpu")
        torch.normal = lambda mean, std, size=None, **k: mean + std * (z[:, 0] if size is None else z[:, 0].reshape(size))
        out["gaussian"] = m.sample_gaussian(0.3, 1.7, (n,), "cpu")
        out["gaussian_tensor"] = m.sample_gaussian(lo, hi - lo, (n,), "cpu")
        torch.Tensor.uniform_ = lambda self, a=0.0, b=1.0: self.copy_(a + (b - a) * u[:, 1].reshape(self.shape))
        with _RandQueue(rand=[u[:, 0]]):
            out["cylinder"] = m.sample_cylinder(0.35, (-0.2, 0.9), n, "cpu")
    finally:
        torch.randn, torch.normal, torch.Tensor.uniform_ = randn, normal, uniform_
    save("samplers", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["math", "actuators", "buffers", "terms", "managers", "sensors", "command", "inhand", "inhand_reset", "events", "terrains", "imu", "frames", "camera", "samplers", "camera_class", "mesh_terrains"]
    table = {"math": gen_math, "actuators": gen_actuators, "buffers": gen_buffers_noise_modifiers, "terms": gen_terms,
             "managers": gen_managers, "sensors": gen_sensors_state, "command": gen_command_action, "inhand": gen_inhand, "inhand_reset": gen_inhand_reset,
             "events": gen_events, "terrains": gen_terrains, "mesh_terrains": gen_mesh_terrains, "imu": gen_imu,
             "frames": gen_frame_transformer, "camera": gen_camera, "camera_class": gen_camera_class, "samplers": gen_samplers}
    for w in which:
        table[w]()
