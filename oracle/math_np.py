"""numpy fp32 restatement of ``LAB/utils/math.py`` (on-path functions)."""
from __future__ import annotations

import numpy as np

F = np.float32
PI = F(np.pi)


def _f(x):
    return np.asarray(x, dtype=np.float32)


def scale_transform(x, lower, upper):  # math.py:22-40
    x, lower, upper = _f(x), _f(lower), _f(upper)
    offset = (lower + upper) * F(0.5)
    return F(2) * (x - offset) / (upper - lower)


def unscale_transform(x, lower, upper):  # math.py:43-61
    x, lower, upper = _f(x), _f(lower), _f(upper)
    offset = (lower + upper) * F(0.5)
    return x * (upper - lower) * F(0.5) + offset


def saturate(x, lower, upper):  # math.py:64-78
    return np.maximum(np.minimum(_f(x), _f(upper)), _f(lower))


def norm(x, axis=-1):
    x = _f(x)
    return np.sqrt(np.sum(x * x, axis=axis, dtype=np.float32))


def normalize(x, eps=1e-9):  # math.py:81-92
    x = _f(x)
    return x / np.maximum(norm(x), F(eps))[..., None]


def wrap_to_pi(angles):  # math.py:95-117
    angles = _f(angles)
    wrapped = np.mod(angles + PI, F(2) * PI).astype(np.float32)
    return np.where((wrapped == 0) & (angles > 0), PI, wrapped - PI).astype(np.float32)


def copysign(mag, other):  # math.py:120-135
    return np.abs(F(mag)) * np.sign(_f(other))


def matrix_from_quat(q):  # math.py:143-174
    q = _f(q)
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    two_s = F(2.0) / np.sum(q * q, axis=-1, dtype=np.float32)
    o = np.stack([
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], axis=-1)
    return o.reshape(q.shape[:-1] + (3, 3)).astype(np.float32)


def quat_conjugate(q):  # math.py:223-235
    q = _f(q)
    return np.concatenate([q[..., 0:1], -q[..., 1:]], axis=-1)


def quat_inv(q):  # math.py:238-248
    return normalize(quat_conjugate(q))


def quat_from_euler_xyz(roll, pitch, yaw):  # math.py:251-278
    roll, pitch, yaw = _f(roll), _f(pitch), _f(yaw)
    cy, sy = np.cos(yaw * F(0.5)), np.sin(yaw * F(0.5))
    cr, sr = np.cos(roll * F(0.5)), np.sin(roll * F(0.5))
    cp, sp = np.cos(pitch * F(0.5)), np.sin(pitch * F(0.5))
    return np.stack([cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp,
                     cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp], axis=-1).astype(np.float32)


def quat_from_matrix(m):  # math.py:294-349
    m = _f(m)
    batch = m.shape[:-2]
    f = m.reshape(batch + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [f[..., i] for i in range(9)]
    x = np.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22],
                 axis=-1).astype(np.float32)
    q_abs = np.where(x > 0, np.sqrt(np.maximum(x, 0)), 0).astype(np.float32)
    cand = np.stack([
        np.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], axis=-1),
        np.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], axis=-1),
        np.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], axis=-1),
        np.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], axis=-1)], axis=-2).astype(np.float32)
    cand = cand / (F(2.0) * np.maximum(q_abs[..., None], F(0.1)))
    best = np.argmax(q_abs, axis=-1)
    return np.take_along_axis(cand, best[..., None, None].repeat(4, -1), axis=-2)[..., 0, :].astype(np.float32)


def euler_xyz_from_quat(q):  # math.py:413-444
    q = _f(q)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    roll = np.arctan2(F(2.0) * (w * x + y * z), 1 - 2 * (x * x + y * y))
    sin_pitch = F(2.0) * (w * y - z * x)
    pitch = np.where(np.abs(sin_pitch) >= 1, copysign(np.pi / 2.0, sin_pitch), np.arcsin(np.clip(sin_pitch, -1, 1)))
    yaw = np.arctan2(F(2.0) * (w * z + x * y), 1 - 2 * (y * y + z * z))
    two_pi = F(2) * PI
    return (np.mod(roll.astype(np.float32), two_pi), np.mod(pitch.astype(np.float32), two_pi),
            np.mod(yaw.astype(np.float32), two_pi))


def quat_unique(q):  # math.py:447-460
    q = _f(q)
    return np.where(q[..., 0:1] < 0, -q, q)


def quat_mul(q1, q2):  # math.py:463-499
    q1, q2 = _f(q1), _f(q2)
    shape = q1.shape
    q1, q2 = q1.reshape(-1, 4), q2.reshape(-1, 4)
    w1, x1, y1, z1 = q1.T
    w2, x2, y2, z2 = q2.T
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = F(0.5) * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return np.stack([w, x, y, z], axis=-1).reshape(shape).astype(np.float32)


def quat_box_minus(q1, q2):  # math.py:502-518
    d = quat_mul(q1, quat_conjugate(q2))
    re, im = d[:, 0], d[:, 1:]
    nim = norm(im)
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = F(2.0) * np.where(nim > 1.0e-7, np.arctan2(nim, re) / nim, np.sign(re))
    return (scale[:, None] * im).astype(np.float32)


def yaw_quat(q):  # math.py:521-542
    q = _f(q)
    shape = q.shape
    q = q.reshape(-1, 4)
    w, x, y, z = q.T
    yaw = np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z)).astype(np.float32)
    out = np.zeros_like(q)
    out[:, 3] = np.sin(yaw / F(2))
    out[:, 0] = np.cos(yaw / F(2))
    return normalize(out).reshape(shape)


def quat_apply(q, v):  # math.py:545-564
    q, v = _f(q), _f(v)
    shape = v.shape
    q, v = q.reshape(-1, 4), v.reshape(-1, 3)
    xyz = q[:, 1:]
    t = np.cross(xyz, v) * F(2)
    return (v + q[:, 0:1] * t + np.cross(xyz, t)).reshape(shape).astype(np.float32)


def quat_apply_yaw(q, v):  # math.py:567-579
    return quat_apply(yaw_quat(q), v)


def _rot(q, v, sign):
    q, v = _f(q), _f(v)
    qw = q[..., 0]
    qv = q[..., 1:]
    a = v * (F(2.0) * qw**2 - F(1.0))[..., None]
    b = np.cross(qv, v) * qw[..., None] * F(2.0)
    c = qv * np.sum(qv * v, axis=-1, dtype=np.float32)[..., None] * F(2.0)
    return (a + sign * b + c).astype(np.float32)


def quat_rotate(q, v):  # math.py:582-602
    return _rot(q, v, F(1))


def quat_rotate_inverse(q, v):  # math.py:605-625
    return _rot(q, v, F(-1))


def quat_from_angle_axis(angle, axis):  # math.py:628-642
    theta = (_f(angle) / F(2))[..., None]
    xyz = normalize(axis) * np.sin(theta)
    return normalize(np.concatenate([np.cos(theta), xyz], axis=-1).astype(np.float32))


def axis_angle_from_quat(q, eps=1.0e-6):  # math.py:645-674
    q = _f(q)
    q = q * (F(1.0) - F(2.0) * (q[..., 0:1] < 0.0))
    mag = norm(q[..., 1:])
    half = np.arctan2(mag, q[..., 0]).astype(np.float32)
    angle = F(2.0) * half
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(np.abs(angle) > eps, np.sin(half) / angle, F(0.5) - angle * angle / F(48))
    return (q[..., 1:4] / s[..., None]).astype(np.float32)


def quat_error_magnitude(q1, q2):  # math.py:677-689
    return norm(axis_angle_from_quat(quat_mul(q1, quat_conjugate(q2))))


def skew_symmetric_matrix(v):  # math.py:692-720
    v = _f(v)
    if v.ndim == 1:
        v = v[None]
    m = np.zeros((v.shape[0], 3, 3), dtype=np.float32)
    m[:, 0, 1], m[:, 0, 2], m[:, 1, 2] = -v[:, 2], v[:, 1], -v[:, 0]
    m[:, 1, 0], m[:, 2, 0], m[:, 2, 1] = v[:, 2], -v[:, 1], v[:, 0]
    return m


def combine_frame_transforms(t01, q01, t12=None, q12=None):  # math.py:749-781
    q02 = quat_mul(q01, q12) if q12 is not None else _f(q01)
    t02 = _f(t01) + quat_apply(q01, t12) if t12 is not None else _f(t01)
    return t02, q02


def subtract_frame_transforms(t01, q01, t02=None, q02=None):  # math.py:785-816
    q10 = quat_inv(q01)
    q12 = quat_mul(q10, q02) if q02 is not None else q10
    t12 = quat_apply(q10, _f(t02) - _f(t01)) if t02 is not None else quat_apply(q10, -_f(t01))
    return t12, q12


def compute_pose_error(t01, q01, t02, q02, rot_error_type="axis_angle"):  # math.py:820-869
    nrm = quat_mul(q01, quat_conjugate(q01))[:, 0]
    inv = quat_conjugate(q01) / nrm[:, None]
    qe = quat_mul(q02, inv)
    pe = _f(t02) - _f(t01)
    return (pe, qe) if rot_error_type == "quat" else (pe, axis_angle_from_quat(qe))


def apply_delta_pose(pos, rot, delta, eps=1.0e-6):  # math.py:872-910
    pos, rot, delta = _f(pos), _f(rot), _f(delta)
    tp = pos + delta[:, 0:3]
    ra = delta[:, 3:6]
    angle = norm(ra)
    with np.errstate(divide="ignore", invalid="ignore"):
        axis = ra / angle[:, None]
        dq = quat_from_angle_axis(angle, axis)
    ident = np.tile(np.array([1.0, 0, 0, 0], dtype=np.float32), (pos.shape[0], 1))
    dq = np.where((angle > eps)[:, None], dq, ident)
    return tp, quat_mul(dq, rot)
