"""numpy restatement of ``LAB/actuators`` and the buffers / interpolation they use."""
from __future__ import annotations

import numpy as np

F = np.float32


def _f(x):
    return np.asarray(x, dtype=np.float32)


def tclip(x, lo, hi):
    """torch.clip(x, min=lo, max=hi): NaN in x propagates."""
    with np.errstate(invalid="ignore"):
        return np.minimum(np.maximum(x, lo), hi).astype(np.float32)


def pd_torque(q_des, qd_des, tau_ff, q, qd, kp, kd):  # actuator_pd.py:135-138
    error_pos = _f(q_des) - _f(q)
    error_vel = _f(qd_des) - _f(qd)
    return (_f(kp) * error_pos + _f(kd) * error_vel + _f(tau_ff)).astype(np.float32)


def clip_effort_base(effort, effort_limit):  # actuator_base.py:252-261
    return tclip(_f(effort), -_f(effort_limit), _f(effort_limit))


def clip_effort_dc(effort, joint_vel, saturation, effort_limit, velocity_limit):  # actuator_pd.py:223-233
    with np.errstate(invalid="ignore", over="ignore"):
        sat = F(saturation)
        max_eff = sat * (F(1.0) - _f(joint_vel) / _f(velocity_limit))
        max_eff = tclip(max_eff, np.zeros_like(max_eff), _f(effort_limit))
        min_eff = sat * (F(-1.0) - _f(joint_vel) / _f(velocity_limit))
        min_eff = tclip(min_eff, -_f(effort_limit), np.zeros_like(min_eff))
        return tclip(_f(effort), min_eff, max_eff)


def ideal_pd(q_des, qd_des, tau_ff, q, qd, kp, kd, effort_limit):  # actuator_pd.py:131-145 (and :62-87)
    tau = pd_torque(q_des, qd_des, tau_ff, q, qd, kp, kd)
    return tau, clip_effort_base(tau, effort_limit)


def dc_motor(q_des, qd_des, tau_ff, q, qd, kp, kd, effort_limit, velocity_limit, saturation):  # :211-233
    tau = pd_torque(q_des, qd_des, tau_ff, q, qd, kp, kd)
    return tau, clip_effort_dc(tau, qd, saturation, effort_limit, velocity_limit)


class CircularBuffer:  # circular_buffer.py:10-166
    def __init__(self, max_len, batch):
        self.L, self.N = max_len, batch
        self.pushes = np.zeros(batch, dtype=np.int64)
        self.ptr = -1
        self.buf = None

    def reset(self, ids=None):
        ids = slice(None) if ids is None else ids
        self.pushes[ids] = 0
        if self.buf is not None:
            self.buf[:, ids] = 0.0

    def append(self, data):
        data = _f(data)
        if self.buf is None:
            self.ptr = -1
            self.buf = np.empty((self.L,) + data.shape, dtype=np.float32)
        self.ptr = (self.ptr + 1) % self.L
        self.buf[self.ptr] = data
        fill = np.nonzero(self.pushes == 0)[0]
        if fill.size:
            self.buf[:, fill] = data[fill]
        self.pushes += 1

    @property
    def buffer(self):  # :78-86
        b = np.roll(self.buf.copy(), self.L - self.ptr - 1, axis=0)
        return np.swapaxes(b, 0, 1)

    def __getitem__(self, key):  # :137-166
        valid = np.minimum(np.asarray(key, dtype=np.int64), self.pushes - 1)
        idx = np.remainder(self.ptr - valid, self.L)
        return self.buf[idx, np.arange(self.N)]


class DelayBuffer:  # delay_buffer.py:15-177
    def __init__(self, history, batch):
        self.cb = CircularBuffer(max(0, history) + 1, batch)
        self.lags = np.zeros(batch, dtype=np.int32)

    def compute(self, data):
        self.cb.append(data)
        return self.cb[self.lags].copy()


def linear_interp(x, y, q):  # linear_interpolation.py:54-85
    x, y, q = _f(x), _f(y), _f(q)
    q1 = q.reshape(-1)
    n_smaller = np.sum(x[:, None] < q1[None, :], axis=0).astype(np.int32)
    lb = np.clip(n_smaller - 1, 0, None)
    ub = np.clip(n_smaller, None, x.size - 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = (q1 - x[lb]) / (x[ub] - x[lb])
    w[ub == lb] = 0.0
    fq = y[lb] + w * (y[ub] - y[lb])
    return fq.reshape(q.shape).astype(np.float32)


def sigmoid(x):
    return (F(1) / (F(1) + np.exp(-x))).astype(np.float32)


_ACTS = {0: lambda x: x, 1: lambda x: np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(np.float32),
         2: lambda x: np.maximum(x, 0), 3: np.tanh, 4: lambda x: x / (1 + np.abs(x)), 5: sigmoid}


def mlp_forward(x, layers):
    """layers: list of (W (out,in), b (out), act id)."""
    for w, b, a in layers:
        x = _ACTS[a]((x @ _f(w).T + _f(b)).astype(np.float32)).astype(np.float32)
    return x


def lstm_step(x, h, c, lstm_weights):
    """torch.nn.LSTM single time step, gate order i,f,g,o.  x (R,in); h,c (L,R,H) updated copies returned."""
    h, c = h.copy(), c.copy()
    inp = _f(x)
    for l, (w_ih, w_hh, b_ih, b_hh) in enumerate(lstm_weights):
        gates = (inp @ _f(w_ih).T + _f(b_ih) + h[l] @ _f(w_hh).T + _f(b_hh)).astype(np.float32)
        H = h.shape[-1]
        i, f, g, o = gates[:, :H], gates[:, H:2 * H], gates[:, 2 * H:3 * H], gates[:, 3 * H:]
        c[l] = sigmoid(f) * c[l] + sigmoid(i) * np.tanh(g)
        h[l] = sigmoid(o) * np.tanh(c[l])
        inp = h[l]
    return inp, h, c


def actuator_net_lstm(q_des, q, qd, h, c, lstm_weights, decoder, effort_limit, velocity_limit, saturation):
    """actuator_net.py:76-99"""
    n, j = q.shape
    x = np.stack([(_f(q_des) - _f(q)).reshape(-1), _f(qd).reshape(-1)], axis=-1)
    top, h, c = lstm_step(x, h, c, lstm_weights)
    tau = mlp_forward(top, decoder).reshape(n, j)
    return tau, clip_effort_dc(tau, qd, saturation, effort_limit, velocity_limit), h, c


def actuator_net_mlp(q_des, q, qd, pos_hist, vel_hist, layers, input_idx, input_order, pos_scale, vel_scale,
                     torque_scale, effort_limit, velocity_limit, saturation):
    """actuator_net.py:147-188; histories (N, Hlen, J), returns updated copies."""
    n, j = q.shape
    pos_hist = np.roll(pos_hist, 1, axis=1)
    pos_hist[:, 0] = _f(q_des) - _f(q)
    vel_hist = np.roll(vel_hist, 1, axis=1)
    vel_hist[:, 0] = _f(qd)
    pos_in = np.concatenate([pos_hist[:, i][:, :, None] for i in input_idx], axis=2).reshape(n * j, -1)
    vel_in = np.concatenate([vel_hist[:, i][:, :, None] for i in input_idx], axis=2).reshape(n * j, -1)
    if input_order == "pos_vel":
        net_in = np.concatenate([pos_in * F(pos_scale), vel_in * F(vel_scale)], axis=1)
    else:
        net_in = np.concatenate([vel_in * F(vel_scale), pos_in * F(pos_scale)], axis=1)
    tau = (mlp_forward(net_in.astype(np.float32), layers).reshape(n, j) * F(torque_scale)).astype(np.float32)
    return tau, clip_effort_dc(tau, qd, saturation, effort_limit, velocity_limit), pos_hist, vel_hist
