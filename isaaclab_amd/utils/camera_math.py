"""Camera-frame helpers of ``LAB/utils/math.py:1429-1561`` used when a ``RayCasterCamera`` is CONFIGURED (offset
convention, ``set_world_poses*``): ``convert_camera_frame_orientation_convention`` and
``create_rotation_matrix_from_view``.  They run once per configuration call on a handful of quaternions, so they are
host-side numpy in fp32 (each step rounded like the reference's fp32 torch ops); nothing here is on the per-step path."""
from __future__ import annotations

import math

import numpy as np

_F = np.float32


def _matrix_from_quat(q: np.ndarray) -> np.ndarray:
    """math.py:143-174."""
    q = q.astype(_F)
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    two_s = (_F(2.0) / (q * q).sum(-1, dtype=_F)).astype(_F)
    o = np.stack([
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1).astype(_F)
    return o.reshape(q.shape[:-1] + (3, 3))


def _quat_from_matrix(m: np.ndarray) -> np.ndarray:
    """math.py:294-349 (largest-component branch selection, as in the reference)."""
    m = m.astype(_F)
    batch = m.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [m[..., a, b] for a in range(3) for b in range(3)]

    def sqrt_pos(x):
        return np.where(x > 0, np.sqrt(np.maximum(x, 0, dtype=_F), dtype=_F), _F(0)).astype(_F)

    q_abs = sqrt_pos(np.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], -1).astype(_F))
    cand = np.stack([
        np.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        np.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
        np.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
        np.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2).astype(_F)
    cand = (cand / (_F(2.0) * np.maximum(q_abs[..., None], _F(0.1)))).astype(_F)
    best = q_abs.argmax(-1)
    return np.take_along_axis(cand, best[..., None, None].repeat(4, -1), -2).reshape(batch + (4,))


def _matrix_from_euler_xyz(a: float, b: float, c: float) -> np.ndarray:
    """math.py:1229-ff ``matrix_from_euler(angles, "XYZ")`` = Rx(a) @ Ry(b) @ Rz(c), fp32."""
    def rot(axis, t):
        t = _F(t)
        cs, sn = np.cos(t, dtype=_F), np.sin(t, dtype=_F)
        if axis == "X":
            return np.array([[1, 0, 0], [0, cs, -sn], [0, sn, cs]], _F)
        if axis == "Y":
            return np.array([[cs, 0, sn], [0, 1, 0], [-sn, 0, cs]], _F)
        return np.array([[cs, -sn, 0], [sn, cs, 0], [0, 0, 1]], _F)

    return (rot("X", a) @ rot("Y", b) @ rot("Z", c)).astype(_F)


def convert_camera_frame_orientation_convention(orientation, origin: str = "opengl", target: str = "ros") -> np.ndarray:
    """math.py:1429-1509.  ``orientation``: (..., 4) wxyz, array-like; returns a float32 numpy array."""
    q = np.array(orientation, dtype=_F)
    if target == origin:
        return q.copy()
    flat = q.reshape(-1, 4)
    world = _matrix_from_euler_xyz(math.pi / 2, -math.pi / 2, 0.0)
    if origin == "ros":
        rotm = _matrix_from_quat(flat)
        rotm[:, :, 2] = -rotm[:, :, 2]
        rotm[:, :, 1] = -rotm[:, :, 1]
        quat_gl = _quat_from_matrix(rotm)
    elif origin == "world":
        quat_gl = _quat_from_matrix((_matrix_from_quat(flat) @ world).astype(_F))
    else:
        quat_gl = flat
    if target == "ros":
        rotm = _matrix_from_quat(quat_gl)
        rotm[:, :, 2] = -rotm[:, :, 2]
        rotm[:, :, 1] = -rotm[:, :, 1]
        out = _quat_from_matrix(rotm)
    elif target == "world":
        out = _quat_from_matrix((_matrix_from_quat(quat_gl) @ world.T).astype(_F))
    else:
        out = quat_gl.copy()
    return out.reshape(q.shape).astype(_F)


def _normalize(v: np.ndarray, eps: float) -> np.ndarray:
    n = np.sqrt((v * v).sum(-1, keepdims=True, dtype=_F), dtype=_F)
    return (v / np.maximum(n, _F(eps))).astype(_F)


def create_rotation_matrix_from_view(eyes, targets, up_axis: str = "Z") -> np.ndarray:
    """math.py:1512-1561: (N, 3, 3) rotation from the eye / look-at pairs, OpenGL camera axes."""
    eyes, targets = np.array(eyes, dtype=_F).reshape(-1, 3), np.array(targets, dtype=_F).reshape(-1, 3)
    if up_axis == "Y":
        up = np.tile(np.array((0, 1, 0), _F), (eyes.shape[0], 1))
    elif up_axis == "Z":
        up = np.tile(np.array((0, 0, 1), _F), (eyes.shape[0], 1))
    else:
        raise ValueError(f"Invalid up axis: {up_axis}. Valid options are 'Y' and 'Z'.")
    z_axis = -_normalize(targets - eyes, 1e-5)
    x_axis = _normalize(np.cross(up, z_axis).astype(_F), 1e-5)
    y_axis = _normalize(np.cross(z_axis, x_axis).astype(_F), 1e-5)
    is_close = np.isclose(x_axis, 0.0, atol=5e-3).all(1, keepdims=True)
    if is_close.any():
        x_axis = np.where(is_close, _normalize(np.cross(y_axis, z_axis).astype(_F), 1e-5), x_axis)
    rot = np.concatenate((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), 1)
    return rot.transpose(0, 2, 1).astype(_F)
