"""numpy restatement of Imu._update_buffers_impl (LAB/sensors/imu/imu.py:142-182).  TEST INFRASTRUCTURE."""
from __future__ import annotations

import numpy as np

from . import math_np as M


class ImuNP:
    def __init__(self, n, offset_pos, offset_quat, gravity_bias=(0.0, 0.0, 9.81)):
        f = lambda *s: np.zeros(s, np.float32)
        self.pos_w, self.quat_w = f(n, 3), f(n, 4)
        self.quat_w[:, 0] = 1.0
        self.lin_vel_b, self.ang_vel_b, self.lin_acc_b, self.ang_acc_b = f(n, 3), f(n, 3), f(n, 3), f(n, 3)
        self.prev_lin, self.prev_ang = f(n, 3), f(n, 3)
        self.off_p = np.broadcast_to(np.asarray(offset_pos, np.float32), (n, 3)).copy()
        self.off_q = np.broadcast_to(np.asarray(offset_quat, np.float32), (n, 4)).copy()
        self.g = np.broadcast_to(np.asarray(gravity_bias, np.float32), (n, 3)).copy()

    def update(self, ids, transforms_xyzw, velocities, coms, dt):
        t = np.asarray(transforms_xyzw, np.float32)[ids]
        pos, q = t[:, :3], t[:, [6, 3, 4, 5]]
        self.pos_w[ids] = pos + M.quat_rotate(q, self.off_p[ids])
        self.quat_w[ids] = M.quat_mul(q, self.off_q[ids])
        v = np.asarray(velocities, np.float32)[ids]
        lin, ang = v[:, :3].copy(), v[:, 3:]
        lin = (lin + np.cross(ang, M.quat_rotate(q, self.off_p[ids] - np.asarray(coms, np.float32)[ids, :3]))).astype(np.float32)
        dt = np.float32(dt)
        lin_acc = ((lin - self.prev_lin[ids]) / dt + self.g[ids]).astype(np.float32)
        ang_acc = ((ang - self.prev_ang[ids]) / dt).astype(np.float32)
        qs = self.quat_w[ids]
        self.lin_vel_b[ids] = M.quat_rotate_inverse(qs, lin)
        self.ang_vel_b[ids] = M.quat_rotate_inverse(qs, ang)
        self.lin_acc_b[ids] = M.quat_rotate_inverse(qs, lin_acc)
        self.ang_acc_b[ids] = M.quat_rotate_inverse(qs, ang_acc)
        self.prev_lin[ids], self.prev_ang[ids] = lin, ang
