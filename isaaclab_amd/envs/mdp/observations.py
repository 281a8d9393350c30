"""Observation terms (``LAB/envs/mdp/observations.py:33-173,512-531``).  Each function is
``func(env, **params) -> (N, d) tensor`` and carries ``.lower`` for the fused ObservationManager kernel."""
from __future__ import annotations

import torch

from ..._terms import OBS, TermSpec, ids_to_list, lowered
from ...managers.scene_entity_cfg import SceneEntityCfg
from ._common import contact_history, mat, root_state


def _copy(t: torch.Tensor, ld: int, off: int, dim: int, before=None, idx=None) -> TermSpec:
    return TermSpec(OBS, "MDP_OBS_COPY", dim=dim, p={0: t}, i={0: ld, 1: off}, idx=idx, before=before or [])


@lowered(OBS)
def base_pos_z(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:33-37"""
    st, ld, before = root_state(env.scene[asset_cfg.name])
    return _copy(st, ld, 2, 1, before)


@lowered(OBS)
def base_lin_vel(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:40-44"""
    st, ld, before = root_state(env.scene[asset_cfg.name])
    return TermSpec(OBS, "MDP_OBS_BASE_LIN_VEL", dim=3, p={0: st}, i={0: ld}, before=before)


@lowered(OBS)
def base_ang_vel(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:47-51"""
    st, ld, before = root_state(env.scene[asset_cfg.name])
    return TermSpec(OBS, "MDP_OBS_BASE_ANG_VEL", dim=3, p={0: st}, i={0: ld}, before=before)


@lowered(OBS)
def projected_gravity(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:54-58"""
    st, ld, before = root_state(env.scene[asset_cfg.name])
    return TermSpec(OBS, "MDP_OBS_PROJECTED_GRAVITY", dim=3, p={0: st}, i={0: ld}, f={0: 0.0, 1: 0.0, 2: -1.0},
                    before=before)


@lowered(OBS)
def root_pos_w(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:61-65"""
    st, ld, before = root_state(env.scene[asset_cfg.name])
    org, ld_o = mat(env.scene.env_origins, "env_origins")
    return TermSpec(OBS, "MDP_OBS_SUB", dim=3, p={0: st, 1: org}, i={0: ld, 1: 0, 2: ld_o, 3: 0}, before=before)


@lowered(OBS)
def root_quat_w(env, make_quat_unique: bool = False, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:68-82"""
    st, ld, before = root_state(env.scene[asset_cfg.name])
    if make_quat_unique:
        return TermSpec(OBS, "MDP_OBS_ROOT_QUAT_UNIQUE", dim=4, p={0: st}, i={0: ld}, before=before)
    return _copy(st, ld, 3, 4, before)


@lowered(OBS)
def root_lin_vel_w(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:85-89"""
    st, ld, before = root_state(env.scene[asset_cfg.name])
    return _copy(st, ld, 7, 3, before)


@lowered(OBS)
def root_ang_vel_w(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:92-96"""
    st, ld, before = root_state(env.scene[asset_cfg.name])
    return _copy(st, ld, 10, 3, before)


def _joint_sel(asset, asset_cfg):
    idx = ids_to_list(asset_cfg.joint_ids, asset.num_joints)
    dim = asset.num_joints if idx is None else len(idx)
    return idx, dim


@lowered(OBS)
def joint_pos(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:104-111"""
    asset = env.scene[asset_cfg.name]
    q, ld = mat(asset.data.joint_pos, "joint_pos")
    idx, dim = _joint_sel(asset, asset_cfg)
    return _copy(q, ld, 0, dim, idx=idx)


@lowered(OBS)
def joint_pos_rel(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:114-121"""
    asset = env.scene[asset_cfg.name]
    q, ld = mat(asset.data.joint_pos, "joint_pos")
    d, ld_d = mat(asset.data.default_joint_pos, "default_joint_pos")
    idx, dim = _joint_sel(asset, asset_cfg)
    return TermSpec(OBS, "MDP_OBS_SUB", dim=dim, p={0: q, 1: d}, i={0: ld, 1: 0, 2: ld_d, 3: 0}, idx=idx)


@lowered(OBS)
def joint_pos_limit_normalized(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:124-137"""
    asset = env.scene[asset_cfg.name]
    q, ld = mat(asset.data.joint_pos, "joint_pos")
    idx, dim = _joint_sel(asset, asset_cfg)
    return TermSpec(OBS, "MDP_OBS_JOINT_POS_LIMIT_NORMALIZED", dim=dim,
                    p={0: q, 1: asset.data.soft_joint_pos_limits}, i={0: ld, 1: asset.num_joints}, idx=idx)


@lowered(OBS)
def joint_vel(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:140-147"""
    asset = env.scene[asset_cfg.name]
    qd, ld = mat(asset.data.joint_vel, "joint_vel")
    idx, dim = _joint_sel(asset, asset_cfg)
    return _copy(qd, ld, 0, dim, idx=idx)


@lowered(OBS)
def joint_vel_rel(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """observations.py:150-157"""
    asset = env.scene[asset_cfg.name]
    qd, ld = mat(asset.data.joint_vel, "joint_vel")
    d, ld_d = mat(asset.data.default_joint_vel, "default_joint_vel")
    idx, dim = _joint_sel(asset, asset_cfg)
    return TermSpec(OBS, "MDP_OBS_SUB", dim=dim, p={0: qd, 1: d}, i={0: ld, 1: 0, 2: ld_d, 3: 0}, idx=idx)


@lowered(OBS)
def height_scan(env, sensor_cfg: SceneEntityCfg, offset: float = 0.5):
    """observations.py:165-173.  Inside an ObservationManager the ray cast itself is fused into the kernel."""
    sensor = env.scene.sensors[sensor_cfg.name]
    raw = sensor._data  # buffers are persistent; `before` triggers the lazy update when not fused
    return TermSpec(OBS, "MDP_OBS_HEIGHT_SCAN", dim=sensor.num_rays, p={0: raw.pos_w, 1: raw.ray_hits_w},
                    i={0: sensor.num_rays}, f={0: offset}, raycaster=sensor,
                    before=[lambda s=sensor: s.data])


@lowered(OBS)
def last_action(env, action_name: str | None = None):
    """observations.py:512-521"""
    if action_name is None:
        a, ld = mat(env.action_manager.action, "action")
    else:
        a, ld = mat(env.action_manager.get_term(action_name).raw_actions, "raw_actions")
    return _copy(a, ld, 0, a.shape[1])


@lowered(OBS)
def generated_commands(env, command_name: str):
    """observations.py:529-531"""
    c, ld = mat(env.command_manager.get_command(command_name), "command")
    return _copy(c, ld, 0, c.shape[1])
