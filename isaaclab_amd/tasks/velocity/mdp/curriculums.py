"""``terrain_levels_vel`` (``TASKS/manager_based/locomotion/velocity/mdp/curriculums.py:26-55``): reset-time
term (off the per-step hot path); kept for API completeness of the task cfg."""
from __future__ import annotations

from collections.abc import Sequence

import torch

from ....managers.scene_entity_cfg import SceneEntityCfg


def terrain_levels_vel(env, env_ids: Sequence[int], asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")) -> torch.Tensor:
    asset = env.scene[asset_cfg.name]
    terrain = env.scene.terrain
    command = env.command_manager.get_command("base_velocity")
    distance = torch.norm(asset.data.root_pos_w[env_ids, :2] - env.scene.env_origins[env_ids, :2], dim=1)
    move_up = distance > terrain.cfg.terrain_generator.size[0] / 2
    move_down = distance < torch.norm(command[env_ids, :2], dim=1) * env.max_episode_length_s * 0.5
    move_down *= ~move_up
    terrain.update_env_origins(env_ids, move_up, move_down)
    return torch.mean(terrain.terrain_levels.float())
