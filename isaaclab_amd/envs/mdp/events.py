"""Event terms (mirror of ``isaaclab.envs.mdp.events``, LAB/envs/mdp/events.py:610-866): same names and arguments; the
arithmetic runs in ``csrc/events.hip``.

``env_ids`` may be what the reference passes -- ``None`` (every env) or a tensor of ids -- or a byte / bool MASK of
``num_envs`` entries, which is what the sync-free reset path and the interval timers of ``EventManager`` hand over
(an id list has to come from ``nonzero()`` on the host; a mask does not).

Random numbers come from a per-term device state (Philox, one epoch word per env).  Tests make the HIP path and the
oracle consume the same numbers through ``env.event_uniforms[<function name>]`` (a ``(N, k)`` tensor of uniforms,
row e = the numbers env e draws, in the order the reference draws them)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _lib as L
from ...managers.scene_entity_cfg import SceneEntityCfg

__all__ = ["reset_root_state_uniform", "reset_joints_by_scale", "reset_joints_by_offset", "push_by_setting_velocity"]

_KEYS = ("x", "y", "z", "roll", "pitch", "yaw")


def _mask_ptr(env, env_ids, keep: list):
    """None -> NULL (all envs); uint8 / bool tensor of num_envs entries -> itself; id tensor -> scattered mask."""
    if env_ids is None or isinstance(env_ids, slice):
        return None
    t = torch.as_tensor(env_ids, device=env.device)
    if t.dtype in (torch.uint8, torch.bool):
        if t.shape != (env.num_envs,):
            raise ValueError(f"env mask must have shape ({env.num_envs},), got {tuple(t.shape)}")
        t = t.view(torch.uint8) if t.dtype == torch.bool else t
    else:
        m = torch.zeros(env.num_envs, dtype=torch.uint8, device=env.device)
        m[t.long()] = 1
        t = m
    keep.append(t)
    return t.data_ptr()


def _state(env, key, seed_tag: int):
    """Per-term persistent device state: {seed, one epoch word per env}."""
    states = env.__dict__.setdefault("_event_rng_states", {})
    st = states.get(key)
    if st is None:
        st = torch.zeros(1 + env.num_envs, dtype=torch.int64, device=env.device)
        st[0] = (int(getattr(getattr(env, "cfg", None), "seed", 0) or 0) * 0x9E3779B1 + seed_tag * 0x85EBCA77 + len(states) + 1) & 0x7FFFFFFFFFFFFFFF
        states[key] = st
    return st


def _injected(env, name: str, width: int):
    u = getattr(env, "event_uniforms", {}).get(name)
    if u is None:
        return None
    if u.shape != (env.num_envs, width) or u.dtype != torch.float32 or not u.is_contiguous():
        raise ValueError(f"event_uniforms['{name}'] must be a contiguous float32 ({env.num_envs}, {width}) tensor")
    return u


def _dict_range(rng: dict, lo_out, scale_out):
    """Dict ranges become a float32 tensor in the reference: scale = fp32(hi) - fp32(lo), computed in fp32."""
    for k, key in enumerate(_KEYS):
        lo, hi = rng.get(key, (0.0, 0.0))
        lo32, hi32 = np.float32(lo), np.float32(hi)
        lo_out[k] = float(lo32)
        scale_out[k] = float(np.float32(hi32 - lo32))


def _fill_reset_root_state(env, env_ids, pose_range: dict, velocity_range: dict, asset_cfg: SceneEntityCfg, keep: list,
                           body_acc: bool = True):
    """``mdp_event_root_state_t`` of the term (no launch).  ``body_acc=False``: the caller zeroes ``body_acc_w`` itself
    (the launch that carries the term defers it, ``ArticulationData.defer_body_acc_zero``)."""
    asset = env.scene[asset_cfg.name]
    d = asset.data
    a = L.MdpEventRootState()
    a.env_mask = _mask_ptr(env, env_ids, keep)
    a.default_root_state, a.env_origins = d.default_root_state.data_ptr(), env.scene.env_origins.data_ptr()
    _dict_range(pose_range, a.pose_lo, a.pose_scale)
    _dict_range(velocity_range, a.vel_lo, a.vel_scale)
    u = _injected(env, "reset_root_state_uniform", 12)
    a.injected_u = u.data_ptr() if u is not None else None
    a.rng_state = None if u is not None else _state(env, ("root", asset_cfg.name), 1).data_ptr()
    view = asset.root_physx_view
    a.root_state_w = d.root_state_w.data_ptr()           # refreshed first if stale, then overwritten for the envs
    a.sim_root_transforms, a.sim_root_velocities = view.get_root_transforms().data_ptr(), view.get_root_velocities().data_ptr()
    if body_acc:
        a.body_acc_w, a.num_bodies = d.body_acc_w.data_ptr(), asset.num_bodies
    return a


def reset_root_state_uniform(env, env_ids, pose_range: dict, velocity_range: dict,
                             asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """events.py:638-680."""
    keep: list = []
    a = _fill_reset_root_state(env, env_ids, pose_range, velocity_range, asset_cfg, keep)
    L.check(L.lib.mdp_event_reset_root_state_uniform(env.num_envs, C.byref(a), L.stream(env.device)))
    _invalidate_body_state(env.scene[asset_cfg.name].data)


def _fill_reset_joints(env, env_ids, position_range, velocity_range, asset_cfg, mode: int, name: str, keep: list):
    """``mdp_event_joints_t`` of the term (no launch)."""
    asset = env.scene[asset_cfg.name]
    d = asset.data
    a = L.MdpEventJoints()
    a.env_mask = _mask_ptr(env, env_ids, keep)
    a.num_joints, a.mode = asset.num_joints, mode
    a.default_joint_pos, a.default_joint_vel = d.default_joint_pos.data_ptr(), d.default_joint_vel.data_ptr()
    # tuple ranges are Python floats in the reference: (hi - lo) in double, rounded to fp32 by the tensor op
    a.pos_lo, a.pos_scale = float(position_range[0]), float(position_range[1] - position_range[0])
    a.vel_lo, a.vel_scale = float(velocity_range[0]), float(velocity_range[1] - velocity_range[0])
    a.soft_joint_pos_limits, a.soft_joint_vel_limits = d.soft_joint_pos_limits.data_ptr(), d.soft_joint_vel_limits.data_ptr()
    u = _injected(env, name, 2 * asset.num_joints)
    a.injected_u = u.data_ptr() if u is not None else None
    a.rng_state = None if u is not None else _state(env, ("joints", asset_cfg.name, mode), 2 + mode).data_ptr()
    a.joint_pos, a.joint_vel = d.joint_pos.data_ptr(), d.joint_vel.data_ptr()
    d.joint_acc                                           # stale finite difference first, then zeroed for the envs
    a.prev_joint_vel, a.joint_acc = d._previous_joint_vel.data_ptr(), d._joint_acc.data.data_ptr()
    view = asset.root_physx_view
    q, qd = view.get_dof_positions(), view.get_dof_velocities()
    a.sim_dof_positions, a.sim_dof_velocities, a.ld_sim = q.data_ptr(), qd.data_ptr(), q.stride(0)
    return a


def _reset_joints(env, env_ids, position_range, velocity_range, asset_cfg, mode: int, name: str):
    keep: list = []
    a = _fill_reset_joints(env, env_ids, position_range, velocity_range, asset_cfg, mode, name, keep)
    L.check(L.lib.mdp_event_reset_joints(env.num_envs, C.byref(a), L.stream(env.device)))
    _invalidate_body_state(env.scene[asset_cfg.name].data)


def reset_joints_by_scale(env, env_ids, position_range: tuple, velocity_range: tuple,
                          asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """events.py:802-832."""
    _reset_joints(env, env_ids, position_range, velocity_range, asset_cfg, 0, "reset_joints_by_scale")


def reset_joints_by_offset(env, env_ids, position_range: tuple, velocity_range: tuple,
                           asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """events.py:835-866."""
    _reset_joints(env, env_ids, position_range, velocity_range, asset_cfg, 1, "reset_joints_by_offset")


def _fill_push(env, env_ids, velocity_range: dict, asset_cfg: SceneEntityCfg, keep: list, body_acc: bool = True):
    """``mdp_event_push_t`` of the term (no launch); ``body_acc`` as in ``_fill_reset_root_state``."""
    asset = env.scene[asset_cfg.name]
    d = asset.data
    a = L.MdpEventPush()
    a.env_mask = _mask_ptr(env, env_ids, keep)
    _dict_range(velocity_range, a.vel_lo, a.vel_scale)
    u = _injected(env, "push_by_setting_velocity", 6)
    a.injected_u = u.data_ptr() if u is not None else None
    a.rng_state = None if u is not None else _state(env, ("push", asset_cfg.name), 4).data_ptr()
    a.root_state_w = d.root_state_w.data_ptr()
    a.sim_root_velocities = asset.root_physx_view.get_root_velocities().data_ptr()
    if body_acc:
        a.body_acc_w, a.num_bodies = d.body_acc_w.data_ptr(), asset.num_bodies
    return a


def push_by_setting_velocity(env, env_ids, velocity_range: dict, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """events.py:610-635."""
    keep: list = []
    a = _fill_push(env, env_ids, velocity_range, asset_cfg, keep)
    L.check(L.lib.mdp_event_push_by_setting_velocity(env.num_envs, C.byref(a), L.stream(env.device)))


def _invalidate_body_state(d):
    """articulation.py:337-339, 503-505: the body states are read again from the simulation."""
    for name in ("_body_state_w", "_body_link_state_w", "_body_com_state_w"):
        buf = getattr(d, name, None)
        if buf is not None:
            buf.timestamp = -1.0
