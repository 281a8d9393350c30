"""``isaaclab.utils.math`` on the MI355X: every function launches the matching row kernel of
``libmdp_hip.so`` (``mdp_math_op``).  Same names / argument meaning / shapes as ``LAB/utils/math.py``
(line numbers cited per function).  Inputs must be fp32 HIP tensors; there is no torch fallback.
"""
from __future__ import annotations

import math
from typing import Literal

import numpy as np
import torch

from .. import _lib as L

_OPS = L.MATH_OPS


def _flat(t: torch.Tensor, width: int, name: str):
    t = L.f32(t, name)
    if t.shape[-1] != width:
        raise ValueError(f"Expected last dimension {width} for {name}, got shape {tuple(t.shape)}.")
    t2 = t.reshape(-1, width)
    if t2.stride(-1) != 1 or (t2.shape[0] > 1 and t2.stride(0) < width):
        t2 = t2.contiguous()
    return t2, (t2.stride(0) if t2.shape[0] > 1 else width)


def _launch(op: str, n: int, ins, out: torch.Tensor, i0: int = 0, f0: float = 0.0):
    args = []
    for item in ins:
        if item is None:
            args += [None, 0]
        else:
            t, ld = item
            args += [t.data_ptr(), ld]
    while len(args) < 8:
        args += [None, 0]
    L.check(L.lib.mdp_math_op(_OPS[op], n, *args, out.data_ptr(), out.stride(0) if out.dim() > 1 and out.shape[0] > 1 else (out.shape[-1] if out.dim() > 1 else 1), i0, f0, L.stream(out.device)))
    return out


def _rowop(op: str, inputs: list[tuple[torch.Tensor, int] | None], out_width: int, out_shape, i0=0, f0=0.0):
    first = next(t for t in inputs if t is not None)[0]
    n = first.shape[0]
    out = torch.empty((n, out_width), dtype=torch.float32, device=first.device)
    _launch(op, n, inputs, out, i0, f0)
    return out.reshape(out_shape)


def _bcast(t: torch.Tensor, n: int, width: int, name: str):
    """(dims,) bounds broadcast over rows via a zero row stride."""
    t = L.f32(t, name)
    if t.dim() == 1 or (t.dim() == 2 and t.shape[0] == 1 and n != 1):
        return t.reshape(-1).contiguous(), 0
    return _flat(t, width, name)


# -- general ------------------------------------------------------------------------------------

def _elementwise3(op: str, x: torch.Tensor, lower: torch.Tensor, upper: torch.Tensor):
    """math.py:22-78 use torch broadcasting of (N, dims) with (dims,) or (N, dims)."""
    x = L.f32(x, "x")
    lower = torch.as_tensor(lower, dtype=torch.float32, device=x.device)
    upper = torch.as_tensor(upper, dtype=torch.float32, device=x.device)
    shape = torch.broadcast_shapes(x.shape, lower.shape, upper.shape)
    xe = x.expand(shape).contiguous().reshape(-1, 1)
    le = lower.expand(shape).contiguous().reshape(-1, 1)
    ue = upper.expand(shape).contiguous().reshape(-1, 1)
    return _rowop(op, [(xe, 1), (le, 1), (ue, 1)], 1, shape)


def scale_transform(x, lower, upper):
    """math.py:22-40"""
    return _elementwise3("MDP_MATH_SCALE_TRANSFORM", x, lower, upper)


def unscale_transform(x, lower, upper):
    """math.py:43-61"""
    return _elementwise3("MDP_MATH_UNSCALE_TRANSFORM", x, lower, upper)


def saturate(x, lower, upper):
    """math.py:64-78"""
    return _elementwise3("MDP_MATH_SATURATE", x, lower, upper)


def normalize(x: torch.Tensor, eps: float = 1e-9) -> torch.Tensor:
    """math.py:81-92"""
    w = x.shape[-1]
    a = _flat(x, w, "x")
    return _rowop("MDP_MATH_NORMALIZE", [a], w, x.shape, i0=w, f0=eps)


def wrap_to_pi(angles: torch.Tensor) -> torch.Tensor:
    """math.py:95-117"""
    a = L.f32(angles, "angles").contiguous().reshape(-1, 1)
    return _rowop("MDP_MATH_WRAP_TO_PI", [(a, 1)], 1, angles.shape)


def copysign(mag: float, other: torch.Tensor) -> torch.Tensor:
    """math.py:120-135"""
    a = L.f32(other, "other").contiguous().reshape(-1, 1)
    return _rowop("MDP_MATH_COPYSIGN", [(a, 1)], 1, other.shape, f0=float(mag))


# -- rotations ----------------------------------------------------------------------------------

def matrix_from_quat(quaternions: torch.Tensor) -> torch.Tensor:
    """math.py:143-174"""
    a = _flat(quaternions, 4, "quaternions")
    return _rowop("MDP_MATH_MATRIX_FROM_QUAT", [a], 9, quaternions.shape[:-1] + (3, 3))


def convert_quat(quat, to: Literal["xyzw", "wxyz"] = "xyzw"):
    """math.py:177-220 (pure re-indexing; numpy supported like the reference)."""
    if quat.shape[-1] != 4:
        raise ValueError(f"Expected input quaternion shape mismatch: {quat.shape} != (..., 4).")
    if to not in ["xyzw", "wxyz"]:
        raise ValueError(f"Expected input argument `to` to be 'xyzw' or 'wxyz'. Received: {to}.")
    shift = -1 if to == "xyzw" else 1
    if isinstance(quat, np.ndarray):
        return np.roll(quat, shift, axis=-1)
    if not isinstance(quat, torch.Tensor):
        quat = torch.tensor(quat, dtype=float)
    return quat.roll(shift, dims=-1)


def quat_conjugate(q: torch.Tensor) -> torch.Tensor:
    """math.py:223-235"""
    return _rowop("MDP_MATH_QUAT_CONJUGATE", [_flat(q, 4, "q")], 4, q.shape)


def quat_inv(q: torch.Tensor) -> torch.Tensor:
    """math.py:238-248"""
    return _rowop("MDP_MATH_QUAT_INV", [_flat(q, 4, "q")], 4, q.shape)


def quat_from_euler_xyz(roll: torch.Tensor, pitch: torch.Tensor, yaw: torch.Tensor) -> torch.Tensor:
    """math.py:251-278"""
    rpy = torch.stack([L.f32(roll, "roll"), L.f32(pitch, "pitch"), L.f32(yaw, "yaw")], dim=-1)
    return _rowop("MDP_MATH_QUAT_FROM_EULER_XYZ", [_flat(rpy, 3, "rpy")], 4, roll.shape + (4,))


def quat_from_matrix(matrix: torch.Tensor) -> torch.Tensor:
    """math.py:294-349"""
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    m = L.f32(matrix, "matrix").reshape(matrix.shape[:-2] + (9,))
    return _rowop("MDP_MATH_QUAT_FROM_MATRIX", [_flat(m, 9, "matrix")], 4, matrix.shape[:-2] + (4,))


def euler_xyz_from_quat(quat: torch.Tensor):
    """math.py:413-444 (angles in [0, 2*pi) like the reference)."""
    out = _rowop("MDP_MATH_EULER_XYZ_FROM_QUAT", [_flat(quat, 4, "quat")], 3, quat.shape[:-1] + (3,))
    return out[..., 0], out[..., 1], out[..., 2]


def quat_unique(q: torch.Tensor) -> torch.Tensor:
    """math.py:447-460"""
    return _rowop("MDP_MATH_QUAT_UNIQUE", [_flat(q, 4, "q")], 4, q.shape)


def quat_mul(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    """math.py:463-499"""
    if q1.shape != q2.shape:
        raise ValueError(f"Expected input quaternion shape mismatch: {q1.shape} != {q2.shape}.")
    return _rowop("MDP_MATH_QUAT_MUL", [_flat(q1, 4, "q1"), _flat(q2, 4, "q2")], 4, q1.shape)


def quat_box_minus(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    """math.py:502-518"""
    return _rowop("MDP_MATH_QUAT_BOX_MINUS", [_flat(q1, 4, "q1"), _flat(q2, 4, "q2")], 3, q1.shape[:-1] + (3,))


def yaw_quat(quat: torch.Tensor) -> torch.Tensor:
    """math.py:521-542"""
    return _rowop("MDP_MATH_YAW_QUAT", [_flat(quat, 4, "quat")], 4, quat.shape)


def _quat_vec(op: str, quat: torch.Tensor, vec: torch.Tensor):
    q = _flat(quat, 4, "quat")
    v = _flat(vec, 3, "vec")
    if q[0].shape[0] != v[0].shape[0]:
        raise ValueError(f"Batch mismatch between quaternions {tuple(quat.shape)} and vectors {tuple(vec.shape)}.")
    return _rowop(op, [q, v], 3, vec.shape)


def quat_apply(quat: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    """math.py:545-564"""
    return _quat_vec("MDP_MATH_QUAT_APPLY", quat, vec)


def quat_apply_yaw(quat: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    """math.py:567-579"""
    return _quat_vec("MDP_MATH_QUAT_APPLY_YAW", quat, vec)


def quat_rotate(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """math.py:582-602"""
    return _quat_vec("MDP_MATH_QUAT_ROTATE", q, v)


def quat_rotate_inverse(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """math.py:605-625"""
    return _quat_vec("MDP_MATH_QUAT_ROTATE_INVERSE", q, v)


def quat_from_angle_axis(angle: torch.Tensor, axis: torch.Tensor) -> torch.Tensor:
    """math.py:628-642"""
    a = L.f32(angle, "angle").contiguous().reshape(-1, 1)
    return _rowop("MDP_MATH_QUAT_FROM_ANGLE_AXIS", [(a, 1), _flat(axis, 3, "axis")], 4, angle.shape + (4,))


def axis_angle_from_quat(quat: torch.Tensor, eps: float = 1.0e-6) -> torch.Tensor:
    """math.py:645-674"""
    return _rowop("MDP_MATH_AXIS_ANGLE_FROM_QUAT", [_flat(quat, 4, "quat")], 3, quat.shape[:-1] + (3,), f0=eps)


def quat_error_magnitude(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    """math.py:677-689"""
    return _rowop("MDP_MATH_QUAT_ERROR_MAGNITUDE", [_flat(q1, 4, "q1"), _flat(q2, 4, "q2")], 1, q1.shape[:-1])


def skew_symmetric_matrix(vec: torch.Tensor) -> torch.Tensor:
    """math.py:692-720"""
    if vec.shape[-1] != 3:
        raise ValueError(f"Expected input vector shape mismatch: {vec.shape} != (..., 3).")
    if vec.ndim == 1:
        vec = vec.unsqueeze(0)
    return _rowop("MDP_MATH_SKEW", [_flat(vec, 3, "vec")], 9, (vec.shape[0], 3, 3))


# -- transformations ----------------------------------------------------------------------------

def is_identity_pose(pos: torch.Tensor, rot: torch.Tensor) -> bool:
    """math.py:728-746 (host-side check, returns a python bool like the reference)."""
    pos_identity = torch.zeros_like(pos)
    rot_identity = torch.zeros_like(rot)
    rot_identity[..., 0] = 1
    return torch.allclose(pos, pos_identity) and torch.allclose(rot, rot_identity)


def combine_frame_transforms(t01, q01, t12=None, q12=None):
    """math.py:749-781"""
    ins = [_flat(t01, 3, "t01"), _flat(q01, 4, "q01"), _flat(t12, 3, "t12") if t12 is not None else None,
           _flat(q12, 4, "q12") if q12 is not None else None]
    out = _rowop("MDP_MATH_COMBINE_FRAME", ins, 7, t01.shape[:-1] + (7,))
    return out[..., :3], out[..., 3:]


def subtract_frame_transforms(t01, q01, t02=None, q02=None):
    """math.py:785-816"""
    ins = [_flat(t01, 3, "t01"), _flat(q01, 4, "q01"), _flat(t02, 3, "t02") if t02 is not None else None,
           _flat(q02, 4, "q02") if q02 is not None else None]
    out = _rowop("MDP_MATH_SUBTRACT_FRAME", ins, 7, t01.shape[:-1] + (7,))
    return out[..., :3], out[..., 3:]


def compute_pose_error(t01, q01, t02, q02, rot_error_type: Literal["quat", "axis_angle"] = "axis_angle"):
    """math.py:820-869"""
    if rot_error_type not in ("quat", "axis_angle"):
        raise ValueError(f"Unsupported orientation error type: {rot_error_type}. Valid: 'quat', 'axis_angle'.")
    as_quat = rot_error_type == "quat"
    w = 7 if as_quat else 6
    ins = [_flat(t01, 3, "t01"), _flat(q01, 4, "q01"), _flat(t02, 3, "t02"), _flat(q02, 4, "q02")]
    out = _rowop("MDP_MATH_COMPUTE_POSE_ERROR", ins, w, t01.shape[:-1] + (w,), i0=1 if as_quat else 0)
    return out[..., :3], out[..., 3:]


def apply_delta_pose(source_pos, source_rot, delta_pose, eps: float = 1.0e-6):
    """math.py:872-910"""
    ins = [_flat(source_pos, 3, "source_pos"), _flat(source_rot, 4, "source_rot"), _flat(delta_pose, 6, "delta_pose")]
    out = _rowop("MDP_MATH_APPLY_DELTA_POSE", ins, 7, source_pos.shape[:-1] + (7,), f0=eps)
    return out[..., :3], out[..., 3:]


def transform_points(points: torch.Tensor, pos: torch.Tensor | None = None, quat: torch.Tensor | None = None):
    """math.py:914-982: p' = R(quat) p + pos with (N,P,3)/(P,3) points and batched or single poses."""
    is_batched = points.dim() == 3
    pts = points if is_batched else points[None]
    if pts.dim() != 3:
        raise ValueError(f"Expected points to have dim = 2 or dim = 3: got shape {points.shape}")
    if not (pos is None or pos.dim() in (1, 2)):
        raise ValueError(f"Expected pos to have dim = 1 or dim = 2: got shape {pos.shape}")
    if not (quat is None or quat.dim() in (1, 2)):
        raise ValueError(f"Expected quat to have dim = 1 or dim = 2: got shape {quat.shape}")
    n, p, _ = pts.shape
    out = L.f32(pts, "points")
    if quat is not None:
        q = quat if quat.dim() == 2 else quat[None]
        q = q[:, None, :].expand(n, p, 4)
        out = quat_rotate(q.reshape(-1, 4), out.reshape(-1, 3)).reshape(n, p, 3)
    if pos is not None:
        t = pos if pos.dim() == 2 else pos[None]
        zero = torch.zeros((1, 4), device=out.device)
        zero[:, 0] = 1.0
        tt = t[:, None, :].expand(n, p, 3).reshape(-1, 3)
        o, _ = combine_frame_transforms(tt, zero.expand(n * p, 4), out.reshape(-1, 3), None)
        out = o.reshape(n, p, 3)
    return out if is_batched else out.squeeze(0)


# -- samplers (math.py:1233-1421) on the device generator (``mdp_sample``) ----------------------------------------
# The reference draws from torch's global generator; streams cannot be matched number for number, so parity is the
# transformation of a given draw (``_inject``: tests/test_gpu_samplers.py against vectors the reference functions
# produced from recorded draws) plus ranges and moments.  Generator state {seed, epoch} lives on the device, one pair
# per device, seeded from ``torch.initial_seed()`` on first use (``seed_samplers`` re-seeds); the epoch advances on the
# device in stream order, so a call captured in a hipGraph keeps drawing fresh numbers.
_SAMPLE_KINDS = L._parse_enum(L._HDR, "MDP_SAMPLE_")
_sampler_state: dict[int, torch.Tensor] = {}
_sampler_seed: dict[int, int | None] = {}      # torch.initial_seed() the state was derived from (None: seed_samplers)


def seed_samplers(seed: int, device="cuda:0") -> None:
    """Explicit seed of the device generator of ``device`` (stays until the next call of this function)."""
    dev = torch.device(device)
    idx = L.device_index(dev)
    _sampler_state[idx] = torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=dev)
    _sampler_seed[idx] = None


def _state(device) -> torch.Tensor:
    dev = torch.device(device)
    if dev.type != "cuda":
        # no torch / CPU arithmetic anywhere in this package's product path: the reference's CPU
        # call sites of these samplers (physics-property randomisation at start-up, events.py:115, 999-1003) are
        # out of scope; a CPU tensor here is a caller's mistake, not something to paper over with a second code path
        raise ValueError(f"samplers run on the HIP device generator; got device '{device}' (there is no CPU path)")
    idx = L.device_index(dev)
    # like the reference's draws, these follow torch's global seed: torch.manual_seed() / env.seed() after the first draw
    # starts a new stream (the state is rebuilt when torch.initial_seed() no longer is the seed it was derived from)
    seed_now = torch.initial_seed()
    if idx not in _sampler_state or (_sampler_seed.get(idx) is not None and _sampler_seed[idx] != seed_now):
        _sampler_state[idx] = torch.tensor([int(seed_now) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=dev)
        _sampler_seed[idx] = seed_now
    return _sampler_state[idx]


def _sample(kind: str, shape: tuple, device, a=None, b=None, scalars=(0.0, 0.0, 0.0), width: int = 1,
            _inject: torch.Tensor | None = None) -> torch.Tensor:
    # tensor bounds broadcast against ``size`` by torch's rules, as in the reference's ``rand(*size) * (upper - lower) +
    # lower``: the result has the broadcast shape
    for t in (a, b):
        if t is not None:
            shape = tuple(torch.broadcast_shapes(tuple(shape), tuple(t.shape)))
    n = 1
    for d_ in shape:
        n *= int(d_)
    out = torch.empty(tuple(shape) + ((width,) if width > 1 else ()), dtype=torch.float32, device=device)
    arrs = []
    for t in (a, b):
        if t is None:
            arrs.append(None)
        else:
            arrs.append(L.f32(t.to(device), "bound").expand(shape).contiguous())
    inj = None if _inject is None else L.f32(_inject, "injected numbers").contiguous()
    if inj is not None and inj.numel() != n * (4 if kind == "MDP_SAMPLE_QUAT" else 2):
        raise ValueError(f"injected numbers: expected {n * (4 if kind == 'MDP_SAMPLE_QUAT' else 2)} values, got {inj.numel()}")
    state = None if inj is not None else _state(device)
    L.check(L.lib.mdp_sample(_SAMPLE_KINDS[kind], n, L.ptr(arrs[0]), L.ptr(arrs[1]), float(scalars[0]), float(scalars[1]),
                             float(scalars[2]), L.ptr(inj), out.data_ptr(), L.ptr(state), L.stream(out.device)))
    return out


def _size(size) -> tuple:
    return (size,) if isinstance(size, int) else tuple(size)


def _bounds(lower, upper, device):
    """(a, b, scalars) of ``_sample``: tensors stay per-element arrays, numbers become kernel scalars."""
    ta = lower if isinstance(lower, torch.Tensor) else None
    tb = upper if isinstance(upper, torch.Tensor) else None
    if (ta is None) != (tb is None):       # mixed: promote the number
        if ta is None:
            ta = torch.full_like(tb, float(lower))
        else:
            tb = torch.full_like(ta, float(upper))
    sc = (0.0 if ta is not None else float(lower), 0.0 if tb is not None else float(upper), 0.0)
    return ta, tb, sc


def sample_triangle(lower: float, upper: float, size, device, _inject=None) -> torch.Tensor:
    """math.py:1288-1310"""
    return _sample("MDP_SAMPLE_TRIANGLE", _size(size), device, scalars=(lower, upper, 0.0), _inject=_inject)


def sample_uniform(lower, upper, size, device, _inject=None) -> torch.Tensor:
    """math.py:1313-1331"""
    a, b, sc = _bounds(lower, upper, device)
    return _sample("MDP_SAMPLE_UNIFORM", _size(size), device, a, b, sc, _inject=_inject)


def sample_log_uniform(lower, upper, size, device, _inject=None) -> torch.Tensor:
    """math.py:1334-1362 (log of the bounds, uniform draw and exp all inside the launch)"""
    a, b, sc = _bounds(lower, upper, device)
    return _sample("MDP_SAMPLE_LOG_UNIFORM", _size(size), device, a, b, sc, _inject=_inject)


def sample_gaussian(mean, std, size, device, _inject=None) -> torch.Tensor:
    """math.py:1365-1384: float mean -> ``size`` samples; tensor mean -> one sample per element of ``mean``."""
    if isinstance(mean, float):
        return _sample("MDP_SAMPLE_GAUSSIAN", _size(size), device, scalars=(mean, float(std), 0.0), _inject=_inject)
    std_t = std if isinstance(std, torch.Tensor) else torch.full_like(mean, float(std))
    return _sample("MDP_SAMPLE_GAUSSIAN", tuple(mean.shape), device, mean, std_t, _inject=_inject)


def sample_cylinder(radius: float, h_range, size, device, _inject=None) -> torch.Tensor:
    """math.py:1387-1421: points on the lateral surface of a z-aligned cylinder, shape (*size, 3)."""
    return _sample("MDP_SAMPLE_CYLINDER", _size(size), device, scalars=(radius, h_range[0], h_range[1]), width=3,
                   _inject=_inject)


def default_orientation(num: int, device: str) -> torch.Tensor:
    """math.py:1233-1247 (a constant: one fill)"""
    quat = torch.zeros((num, 4), dtype=torch.float, device=device)
    quat[..., 0] = 1.0
    return quat


def random_yaw_orientation(num: int, device: str, _inject=None) -> torch.Tensor:
    """math.py:1270-1285"""
    return _sample("MDP_SAMPLE_YAW_QUAT", (num,), device, width=4, _inject=_inject)


def random_orientation(num: int, device: str, _inject=None) -> torch.Tensor:
    """math.py:1250-1267"""
    return _sample("MDP_SAMPLE_QUAT", (num,), device, width=4, _inject=_inject)
