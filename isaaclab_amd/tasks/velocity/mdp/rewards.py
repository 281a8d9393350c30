"""Task rewards (``TASKS/manager_based/locomotion/velocity/mdp/rewards.py:25-106``)."""
from __future__ import annotations

import torch

from ...._terms import REWARD, TermSpec, ids_to_list, lowered
from ....envs.mdp._common import contact_history, f32, mat, root_state
from ....managers.scene_entity_cfg import SceneEntityCfg


@lowered(REWARD)
def feet_air_time(env, command_name: str, sensor_cfg: SceneEntityCfg, threshold: float):
    """rewards.py:25-44 (first contact computed in-kernel: contact_sensor.py:169-203)."""
    sensor = env.scene.sensors[sensor_cfg.name]
    d = sensor.data
    c, ld_c = mat(env.command_manager.get_command(command_name), "command")
    b = sensor.num_bodies
    return TermSpec(REWARD, "MDP_REW_FEET_AIR_TIME", p={0: d.current_contact_time, 1: d.last_air_time, 2: c},
                    i={1: b, 2: ld_c}, f={0: threshold, 1: f32(env.step_dt + 1.0e-8)},
                    idx=ids_to_list(sensor_cfg.body_ids, b), before=[lambda s=sensor: s.data])


@lowered(REWARD)
def feet_air_time_positive_biped(env, command_name: str, threshold: float, sensor_cfg: SceneEntityCfg):
    """rewards.py:47-66"""
    sensor = env.scene.sensors[sensor_cfg.name]
    d = sensor.data
    c, ld_c = mat(env.command_manager.get_command(command_name), "command")
    b = sensor.num_bodies
    return TermSpec(REWARD, "MDP_REW_FEET_AIR_TIME_POSITIVE_BIPED",
                    p={0: d.current_air_time, 1: d.current_contact_time, 2: c}, i={1: b, 2: ld_c}, f={0: threshold},
                    idx=ids_to_list(sensor_cfg.body_ids, b), before=[lambda s=sensor: s.data])


@lowered(REWARD)
def feet_slide(env, sensor_cfg: SceneEntityCfg, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:69-83"""
    sensor = env.scene.sensors[sensor_cfg.name]
    hist, h, bs, before = contact_history(sensor)
    asset = env.scene[asset_cfg.name]
    body = asset.data.body_state_w
    idx = ids_to_list(sensor_cfg.body_ids, bs)
    idx2 = ids_to_list(asset_cfg.body_ids, asset.num_bodies)
    n1 = bs if idx is None else len(idx)
    n2 = asset.num_bodies if idx2 is None else len(idx2)
    if n1 != n2:
        raise RuntimeError(f"The size of tensor a ({n2}) must match the size of tensor b ({n1}) at non-singleton dimension 1")
    if idx2 is None or not hasattr(asset.data, "body_state_rows"):
        refresh = lambda d=asset.data: d.body_state_w
    else:
        # only the rows the term reads (H1: 2 of 20 bodies) instead of the whole (N, B, 13) assembly every step
        rows = torch.tensor(idx2, dtype=torch.int32, device=body.device)
        refresh = lambda d=asset.data, rows=rows: d.body_state_rows(rows)
    return TermSpec(REWARD, "MDP_REW_FEET_SLIDE", p={0: hist, 1: body}, i={0: h, 1: bs, 2: asset.num_bodies},
                    idx=idx, idx2=idx2, before=before + [refresh])


def _track(op, env, std, command_name, asset_cfg):
    st, ld, before = root_state(env.scene[asset_cfg.name])
    c, ld_c = mat(env.command_manager.get_command(command_name), "command")
    return TermSpec(REWARD, op, p={0: st, 1: c}, i={0: ld, 1: ld_c}, f={0: f32(std**2)}, before=before)


@lowered(REWARD)
def track_lin_vel_xy_yaw_frame_exp(env, std: float, command_name: str,
                                   asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:86-96"""
    return _track("MDP_REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP", env, std, command_name, asset_cfg)


@lowered(REWARD)
def track_ang_vel_z_world_exp(env, command_name: str, std: float,
                              asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")):
    """rewards.py:99-106"""
    return _track("MDP_REW_TRACK_ANG_VEL_Z_WORLD_EXP", env, std, command_name, asset_cfg)
