"""CPU: numpy oracle of the actuator models / buffers / interpolation against reference-generated fixtures
(tests/golden/actuators.npz, buffers_noise_modifiers.npz; see tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import assert_close
from oracle import actuators_np as A

F = np.float32


def lstm_weights_from_golden(G, tag):
    layers = []
    l = 0
    while f"{tag}_w_lstm.weight_ih_l{l}" in G:
        layers.append(tuple(G[f"{tag}_w_lstm.{k}_l{l}"] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")))
        l += 1
    dec, idx = [], 0
    keys = sorted({int(k.split(".")[1]) for k in G if k.startswith(f"{tag}_w_decoder.") and k.endswith(".weight")})
    for n_, i in enumerate(keys):
        act = 1 if n_ < len(keys) - 1 else 0  # ELU between layers, identity on the last
        dec.append((G[f"{tag}_w_decoder.{i}.weight"], G[f"{tag}_w_decoder.{i}.bias"], act))
    return layers, dec


def mlp_layers_from_golden(G):
    keys = sorted({int(k.split("_w_")[1].split(".")[0]) for k in G if k.startswith("mlp_w_") and k.endswith(".weight")})
    return [(G[f"mlp_w_{i}.weight"], G[f"mlp_w_{i}.bias"], 4 if n_ < len(keys) - 1 else 0) for n_, i in enumerate(keys)]


@pytest.mark.parametrize("n", (1, 7, 64))
def test_pd_family(golden, n):
    G = golden("actuators")
    for kind in ("implicit", "ideal_pd", "dc_motor"):
        p = f"{kind}_n{n}_"
        io = [G[p + k] for k in ("q_des", "qd_des", "tau_ff", "q", "qd")]
        if kind == "dc_motor":
            comp, app = A.dc_motor(*io, F(40.0), F(5.0), F(80.0), F(7.5), 120.0)
        else:  # implicit actuator computes the same approximate torques (actuator_pd.py:62-87)
            comp, app = A.ideal_pd(*io, F(40.0), F(5.0), F(80.0))
        assert_close(comp, G[p + "computed"], msg=p + "computed")
        assert_close(app, G[p + "applied"], msg=p + "applied")


def test_dc_motor_infinite_saturation(golden):
    G = golden("actuators")
    io = [G["dc_inf_" + k] for k in ("q_des", "qd_des", "tau_ff", "q", "qd")]
    comp, app = A.dc_motor(*io, F(40.0), F(5.0), F(80.0), F(7.5), np.inf)
    assert_close(comp, G["dc_inf_computed"])
    assert_close(app, G["dc_inf_applied"])


def test_delayed_pd_sequence(golden):
    G = golden("actuators")
    seq = G["delayed_inputs"]  # (T, 5, N, J)
    n = seq.shape[2]
    bufs = [A.DelayBuffer(3, n) for _ in range(3)]
    for b in bufs:
        b.lags = G["delayed_lags"].astype(np.int32)
    for t in range(seq.shape[0]):
        q_des, qd_des, tau_ff, q, qd = seq[t]
        d = [b.compute(x) for b, x in zip(bufs, (q_des, qd_des, tau_ff))]
        comp, app = A.ideal_pd(d[0], d[1], d[2], q, qd, F(20.0), F(1.0), F(50.0))
        assert_close(comp, G["delayed_computed"][t], msg=f"t={t}")
        assert_close(app, G["delayed_applied"][t], msg=f"t={t}")
        if t == 3:
            for b in bufs:
                b.cb.reset([1, 4])


def test_remotized_pd(golden):
    G = golden("actuators")
    io = [G["remotized_" + k] for k in ("q_des", "qd_des", "tau_ff", "q", "qd")]
    lut = G["remotized_lut"]
    limit = A.linear_interp(lut[:, 0], lut[:, 2], io[3])
    tau = A.pd_torque(*io, F(60.0), F(2.0))
    assert_close(tau, G["remotized_computed"])
    assert_close(A.tclip(tau, -limit, limit), G["remotized_applied"])


@pytest.mark.parametrize("tag", ("lstm8", "lstm16"))
def test_actuator_net_lstm(golden, tag):
    G = golden("actuators")
    lstm, dec = lstm_weights_from_golden(G, tag)
    seq = G[f"{tag}_inputs"]  # (T, 3, N, J)
    n, j = seq.shape[2], seq.shape[3]
    L, H = len(lstm), lstm[0][1].shape[1]
    h = np.zeros((L, n * j, H), np.float32)
    c = np.zeros_like(h)
    for t in range(seq.shape[0]):
        q_des, q, qd = seq[t]
        comp, app, h, c = A.actuator_net_lstm(q_des, q, qd, h, c, lstm, dec, F(80.0), F(7.5), 120.0)
        if t == 1:  # actuator_net.py:64-69 reset of envs [2, 5]
            hv, cv = h.reshape(L, n, j, H), c.reshape(L, n, j, H)
            hv[:, [2, 5]] = 0.0
            cv[:, [2, 5]] = 0.0
        assert_close(comp, G[f"{tag}_computed"][t], rtol=1e-5, atol=1e-5, msg=f"{tag} t={t}")
        assert_close(app, G[f"{tag}_applied"][t], rtol=1e-5, atol=1e-5, msg=f"{tag} t={t}")
        assert_close(h, G[f"{tag}_h"][t], rtol=1e-5, atol=2e-6, msg=f"{tag} h t={t}")
        assert_close(c, G[f"{tag}_c"][t], rtol=1e-5, atol=2e-6, msg=f"{tag} c t={t}")


@pytest.mark.parametrize("order", ("pos_vel", "vel_pos"))
def test_actuator_net_mlp(golden, order):
    G = golden("actuators")
    layers = mlp_layers_from_golden(G)
    seq = G[f"mlp_{order}_inputs"]
    n, j = seq.shape[2], seq.shape[3]
    ph = np.zeros((n, 3, j), np.float32)
    vh = np.zeros_like(ph)
    for t in range(seq.shape[0]):
        q_des, q, qd = seq[t]
        comp, app, ph, vh = A.actuator_net_mlp(q_des, q, qd, ph, vh, layers, [0, 1, 2], order, -1.0, 1.0, 1.0,
                                               F(23.7), F(30.0), 23.7)
        assert_close(comp, G[f"mlp_{order}_computed"][t], rtol=1e-5, atol=1e-5, msg=f"t={t}")
        assert_close(app, G[f"mlp_{order}_applied"][t], rtol=1e-5, atol=1e-5, msg=f"t={t}")
        if t == 2:  # actuator_net.py:142-145
            ph[[0, 3]] = 0.0
            vh[[0, 3]] = 0.0


def test_circular_and_delay_buffers(golden):
    G = golden("buffers_noise_modifiers")
    data = G["cb_data"]
    cb = A.CircularBuffer(4, 5)
    for t in range(7):
        cb.append(data[t])
        assert_close(cb.buffer, G["cb_buffers"][t], msg=f"t={t}")
        if t == 3:
            cb.reset([1, 3])
    assert_close(cb[np.array([0, 1, 2, 3, 9])], G["cb_lookup"])
    db = A.DelayBuffer(3, 5)
    db.lags = np.array([0, 1, 2, 3, 2], np.int32)
    for t in range(7):
        assert_close(db.compute(data[t]), G["db_out"][t], msg=f"t={t}")
        if t == 4:
            db.cb.reset([0, 4])


def test_linear_interpolation(golden):
    G = golden("buffers_noise_modifiers")
    assert_close(A.linear_interp(G["li_x"], G["li_y"], G["li_q"]), G["li_out"])
