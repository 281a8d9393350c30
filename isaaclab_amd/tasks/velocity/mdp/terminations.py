"""Task terminations (``TASKS/manager_based/locomotion/velocity/mdp/terminations.py:24-52``)."""
from __future__ import annotations

from ...._terms import TERMINATION, TermSpec, lowered
from ....envs.mdp._common import f32, root_state
from ....managers.scene_entity_cfg import SceneEntityCfg


@lowered(TERMINATION)
def terrain_out_of_bounds(env, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot"), distance_buffer: float = 3.0):
    terrain_type = env.scene.cfg.terrain.terrain_type
    if terrain_type == "plane":
        return False  # infinite terrain (terminations.py:32-33)
    elif terrain_type == "generator":
        gen = env.scene.terrain.cfg.terrain_generator
        grid_width, grid_length = gen.size
        map_width = gen.num_rows * grid_width + 2 * gen.border_width
        map_height = gen.num_cols * grid_length + 2 * gen.border_width
        st, ld, before = root_state(env.scene[asset_cfg.name])
        return TermSpec(TERMINATION, "MDP_TERM_TERRAIN_OUT_OF_BOUNDS", p={0: st}, i={0: ld},
                        f={0: f32(0.5 * map_width - distance_buffer), 1: f32(0.5 * map_height - distance_buffer)},
                        before=before)
    raise ValueError("Received unsupported terrain type, must be either 'plane' or 'generator'.")
