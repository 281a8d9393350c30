"""Termination terms (``LAB/envs/mdp/terminations.py:30-158``): ``func(env, **params) -> (N,) bool`` + ``.lower``."""
from __future__ import annotations

import torch

from ..._terms import TERMINATION, TermSpec, ids_to_list, lowered
from ...managers.scene_entity_cfg import SceneEntityCfg
from ._common import contact_history, f32, mat, root_state


@lowered(TERMINATION)
def time_out(env):
    """terminations.py:30-32"""
    return TermSpec(TERMINATION, "MDP_TERM_TIME_OUT", p={0: env.episode_length_buf}, l0=int(env.max_episode_length))


@lowered(TERMINATION)
def command_resample(env, command_name: str, num_resamples: int = 1):
    """terminations.py:35-42"""
    command = env.command_manager.get_term(command_name)
    return TermSpec(TERMINATION, "MDP_TERM_COMMAND_RESAMPLE", p={0: command.time_left, 1: command.command_counter},
                    i={0: num_resamples}, f={0: f32(env.step_dt)})


@lowered(TERMINATION)
def bad_orientation(env, limit_angle: float, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """terminations.py:50-59"""
    st, ld, before = root_state(env.scene[asset_cfg.name])
    return TermSpec(TERMINATION, "MDP_TERM_BAD_ORIENTATION", p={0: st}, i={0: ld},
                    f={0: 0.0, 1: 0.0, 2: -1.0, 3: limit_angle}, before=before)


@lowered(TERMINATION)
def root_height_below_minimum(env, minimum_height: float, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """terminations.py:62-72"""
    st, ld, before = root_state(env.scene[asset_cfg.name])
    return TermSpec(TERMINATION, "MDP_TERM_ROOT_HEIGHT_BELOW_MINIMUM", p={0: st}, i={0: ld}, f={0: minimum_height},
                    before=before)


def joint_pos_out_of_limit(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")) -> torch.Tensor:
    """terminations.py:80-87.  The reference reduces over the joint axis with ``torch.any(..., dim=1)`` and
    then indexes the resulting 1-D mask with ``[:, joint_ids]``, which raises ``IndexError`` on every call;
    the behaviour is reproduced rather than fixed (hard parts)."""
    raise IndexError("too many indices for tensor of dimension 1")


def _joint_term(op: str, env, asset_cfg, getter, getter2=None, f=None):
    asset = env.scene[asset_cfg.name]
    a, ld = mat(getter(asset.data), "joint quantity")
    p, i = {0: a}, {0: ld, 1: asset.num_joints}
    if getter2 is not None:
        b, ld_b = mat(getter2(asset.data), "joint quantity 2")
        if op == "MDP_TERM_JOINT_EFFORT_OUT_OF_LIMIT" and ld_b != ld:
            b = b.contiguous()
            a = a.contiguous()
            p[0], i[0] = a, a.shape[1]
        p[1] = b
    return TermSpec(TERMINATION, op, p=p, i=i, f=f or {}, idx=ids_to_list(asset_cfg.joint_ids, asset.num_joints))


@lowered(TERMINATION)
def joint_pos_out_of_manual_limit(env, bounds: tuple[float, float],
                                  asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """terminations.py:90-105"""
    if asset_cfg.joint_ids is None:
        asset_cfg.joint_ids = slice(None)
    return _joint_term("MDP_TERM_JOINT_POS_OUT_OF_MANUAL_LIMIT", env, asset_cfg, lambda d: d.joint_pos,
                       f={0: bounds[0], 1: bounds[1]})


@lowered(TERMINATION)
def joint_vel_out_of_limit(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """terminations.py:108-114"""
    return _joint_term("MDP_TERM_JOINT_VEL_OUT_OF_LIMIT", env, asset_cfg, lambda d: d.joint_vel,
                       lambda d: d.soft_joint_vel_limits)


@lowered(TERMINATION)
def joint_vel_out_of_manual_limit(env, max_velocity: float, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """terminations.py:117-124"""
    return _joint_term("MDP_TERM_JOINT_VEL_OUT_OF_MANUAL_LIMIT", env, asset_cfg, lambda d: d.joint_vel,
                       f={0: max_velocity})


@lowered(TERMINATION)
def joint_effort_out_of_limit(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """terminations.py:127-142 (flags joints whose computed torque is *close to* the applied one, as written)."""
    return _joint_term("MDP_TERM_JOINT_EFFORT_OUT_OF_LIMIT", env, asset_cfg, lambda d: d.computed_torque,
                       lambda d: d.applied_torque)


@lowered(TERMINATION)
def illegal_contact(env, threshold: float, sensor_cfg: SceneEntityCfg):
    """terminations.py:150-158"""
    sensor = env.scene.sensors[sensor_cfg.name]
    hist, h, b, before = contact_history(sensor)
    return TermSpec(TERMINATION, "MDP_TERM_ILLEGAL_CONTACT", p={0: hist}, i={0: h, 1: b}, f={0: threshold},
                    idx=ids_to_list(sensor_cfg.body_ids, b), before=before)
