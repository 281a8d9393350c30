"""Reward terms (``LAB/envs/mdp/rewards.py:31-309``): ``func(env, **params) -> (N,) tensor`` + ``.lower``."""
from __future__ import annotations

import torch

from ..._terms import REWARD, TermSpec, ids_to_list, lowered
from ...managers.manager_base import ManagerTermBase
from ...managers.scene_entity_cfg import SceneEntityCfg
from ._common import contact_history, f32, mat, root_state


@lowered(REWARD)
def is_alive(env):
    """rewards.py:31-33"""
    return TermSpec(REWARD, "MDP_REW_IS_ALIVE", p={0: env.termination_manager.terminated})


@lowered(REWARD)
def is_terminated(env):
    """rewards.py:36-38"""
    return TermSpec(REWARD, "MDP_REW_IS_TERMINATED", p={0: env.termination_manager.terminated})


class is_terminated_term(ManagerTermBase):
    """rewards.py:41-68"""

    def __init__(self, cfg, env):
        super().__init__(cfg, env)
        term_keys = cfg.params.get("term_keys", ".*")
        self._term_names = env.termination_manager.find_terms(term_keys)

    def lower(self, env, term_keys: str | list[str] = ".*"):
        tm = env.termination_manager
        rows = [tm.active_terms.index(name) for name in self._term_names]
        return TermSpec(REWARD, "MDP_REW_IS_TERMINATED_TERM", p={0: tm._term_dones_block, 1: tm.time_outs},
                        i={1: len(tm.active_terms)}, idx=rows)

    def __call__(self, env, term_keys: str | list[str] = ".*") -> torch.Tensor:
        from ..._terms import evaluate

        return evaluate(self.lower(env, term_keys), env.num_envs, env.device)


def _root_term(op: str, env, asset_cfg, **f):
    st, ld, before = root_state(env.scene[asset_cfg.name])
    return TermSpec(REWARD, op, p={0: st}, i={0: ld}, f=f.get("f", {}), before=before)


@lowered(REWARD)
def lin_vel_z_l2(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:76-80"""
    return _root_term("MDP_REW_LIN_VEL_Z_L2", env, asset_cfg)


@lowered(REWARD)
def ang_vel_xy_l2(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:83-87"""
    return _root_term("MDP_REW_ANG_VEL_XY_L2", env, asset_cfg)


@lowered(REWARD)
def flat_orientation_l2(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:90-97"""
    return _root_term("MDP_REW_FLAT_ORIENTATION_L2", env, asset_cfg, f={0: 0.0, 1: 0.0, 2: -1.0})


@lowered(REWARD)
def base_height_l2(env, target_height: float, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot"),
                   sensor_cfg: SceneEntityCfg | None = None):
    """rewards.py:100-122"""
    st, ld, before = root_state(env.scene[asset_cfg.name])
    p = {0: st}
    i = {0: ld}
    lazy = []
    if sensor_cfg is not None:
        sensor = env.scene[sensor_cfg.name]
        p[1] = sensor._data.ray_hits_w
        i[1] = sensor.num_rays
        before = before + [lambda s=sensor: s.data]
        lazy = [sensor]
    return TermSpec(REWARD, "MDP_REW_BASE_HEIGHT_L2", p=p, i=i, f={0: target_height}, before=before, lazy_sensors=lazy)


@lowered(REWARD)
def body_lin_acc_l2(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:125-128"""
    asset = env.scene[asset_cfg.name]
    acc = asset.data.body_acc_w
    return TermSpec(REWARD, "MDP_REW_BODY_LIN_ACC_L2", p={0: acc}, i={1: asset.num_bodies},
                    idx=ids_to_list(asset_cfg.body_ids, asset.num_bodies), before=[lambda d=asset.data: d.body_acc_w])


def _joint_reduce(op: str, env, asset_cfg, getter, getter2=None, f=None, before=None):
    asset = env.scene[asset_cfg.name]
    a, ld = mat(getter(asset.data), "joint quantity")
    p, i = {0: a}, {0: ld, 1: asset.num_joints}
    if getter2 is not None:
        b, ld_b = mat(getter2(asset.data), "joint quantity 2")
        p[1] = b
        i[2] = ld_b
    return TermSpec(REWARD, op, p=p, i=i, f=f or {}, idx=ids_to_list(asset_cfg.joint_ids, asset.num_joints),
                    before=before or [])


@lowered(REWARD)
def joint_torques_l2(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:136-143"""
    return _joint_reduce("MDP_REW_SUM_SQ", env, asset_cfg, lambda d: d.applied_torque)


@lowered(REWARD)
def joint_vel_l1(env, asset_cfg: SceneEntityCfg):
    """rewards.py:146-150"""
    return _joint_reduce("MDP_REW_SUM_ABS", env, asset_cfg, lambda d: d.joint_vel)


@lowered(REWARD)
def joint_vel_l2(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:153-160"""
    return _joint_reduce("MDP_REW_SUM_SQ", env, asset_cfg, lambda d: d.joint_vel)


@lowered(REWARD)
def joint_acc_l2(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:163-170"""
    data = env.scene[asset_cfg.name].data
    return _joint_reduce("MDP_REW_SUM_SQ", env, asset_cfg, lambda d: d.joint_acc, before=[lambda d=data: d.joint_acc])


@lowered(REWARD)
def joint_deviation_l1(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:173-179"""
    return _joint_reduce("MDP_REW_SUM_ABS_DIFF", env, asset_cfg, lambda d: d.joint_pos, lambda d: d.default_joint_pos)


@lowered(REWARD)
def joint_pos_limits(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:182-196"""
    asset = env.scene[asset_cfg.name]
    q, ld = mat(asset.data.joint_pos, "joint_pos")
    return TermSpec(REWARD, "MDP_REW_JOINT_POS_LIMITS", p={0: q, 1: asset.data.soft_joint_pos_limits},
                    i={0: ld, 1: asset.num_joints}, idx=ids_to_list(asset_cfg.joint_ids, asset.num_joints))


@lowered(REWARD)
def joint_vel_limits(env, soft_ratio: float, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:199-218"""
    asset = env.scene[asset_cfg.name]
    qd, ld = mat(asset.data.joint_vel, "joint_vel")
    return TermSpec(REWARD, "MDP_REW_JOINT_VEL_LIMITS", p={0: qd, 1: asset.data.soft_joint_vel_limits},
                    i={0: ld, 1: asset.num_joints}, f={0: soft_ratio},
                    idx=ids_to_list(asset_cfg.joint_ids, asset.num_joints))


@lowered(REWARD)
def applied_torque_limits(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:226-242"""
    return _joint_reduce("MDP_REW_SUM_ABS_DIFF", env, asset_cfg, lambda d: d.applied_torque, lambda d: d.computed_torque)


@lowered(REWARD)
def action_rate_l2(env):
    """rewards.py:245-247"""
    a, ld = mat(env.action_manager.action, "action")
    b, ld_b = mat(env.action_manager.prev_action, "prev_action")
    return TermSpec(REWARD, "MDP_REW_SUM_SQ_DIFF", p={0: a, 1: b}, i={0: ld, 1: a.shape[1], 2: ld_b})


@lowered(REWARD)
def action_l2(env):
    """rewards.py:250-252"""
    a, ld = mat(env.action_manager.action, "action")
    return TermSpec(REWARD, "MDP_REW_SUM_SQ", p={0: a}, i={0: ld, 1: a.shape[1]})


def _contact_term(op: str, env, threshold: float, sensor_cfg):
    sensor = env.scene.sensors[sensor_cfg.name]
    hist, h, b, before = contact_history(sensor)
    return TermSpec(REWARD, op, p={0: hist}, i={0: h, 1: b}, f={0: threshold},
                    idx=ids_to_list(sensor_cfg.body_ids, b), before=before)


@lowered(REWARD)
def undesired_contacts(env, threshold: float, sensor_cfg: SceneEntityCfg):
    """rewards.py:260-268 (the reference returns an int64 count; the value is the same count as fp32)."""
    return _contact_term("MDP_REW_UNDESIRED_CONTACTS", env, threshold, sensor_cfg)


@lowered(REWARD)
def contact_forces(env, threshold: float, sensor_cfg: SceneEntityCfg):
    """rewards.py:271-279"""
    return _contact_term("MDP_REW_CONTACT_FORCES", env, threshold, sensor_cfg)


def _track(op: str, env, std: float, command_name: str, asset_cfg):
    st, ld, before = root_state(env.scene[asset_cfg.name])
    c, ld_c = mat(env.command_manager.get_command(command_name), "command")
    return TermSpec(REWARD, op, p={0: st, 1: c}, i={0: ld, 1: ld_c}, f={0: f32(std**2)}, before=before)


@lowered(REWARD)
def track_lin_vel_xy_exp(env, std: float, command_name: str, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:287-298"""
    return _track("MDP_REW_TRACK_LIN_VEL_XY_EXP", env, std, command_name, asset_cfg)


@lowered(REWARD)
def track_ang_vel_z_exp(env, std: float, command_name: str, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:301-309"""
    return _track("MDP_REW_TRACK_ANG_VEL_Z_EXP", env, std, command_name, asset_cfg)
