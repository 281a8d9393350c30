"""``terrain_levels_vel`` (``TASKS/manager_based/locomotion/velocity/mdp/curriculums.py:26-55``) with
``TerrainImporter.update_env_origins`` (``LAB/terrains/terrain_importer.py:307-324``) folded in: one masked launch
(``mdp_curriculum_terrain_levels_vel``).  ``env_ids`` may be ids, ``None`` / ``slice(None)``
(every env) or a byte / bool mask of ``num_envs`` entries (the sync-free reset path)."""
from __future__ import annotations

import ctypes as C
from collections.abc import Sequence

import torch

from .... import _lib as L
from ....envs.mdp.events import _mask_ptr, _state
from ....managers.scene_entity_cfg import SceneEntityCfg


def _fill_terrain_levels_vel(env, env_ids, asset_cfg: SceneEntityCfg, keep: list):
    """``mdp_curriculum_terrain_t`` of the term without the mean-level members (no launch)."""
    asset = env.scene[asset_cfg.name]
    terrain = env.scene.terrain
    command = env.command_manager.get_command("base_velocity")
    st = terrain.device_state(env.device)
    a = L.MdpCurriculumTerrain()
    a.env_mask = _mask_ptr(env, env_ids, keep)
    root = asset.data.root_state_w
    a.root_state_w, a.ld_state = root.data_ptr(), root.stride(0)
    a.max_terrain_level = terrain.max_terrain_level
    if not command.is_contiguous() or command.shape[1] != 3:
        raise ValueError("base_velocity command must be a contiguous (N, 3) tensor")
    a.command = command.data_ptr()
    a.half_tile_size = float(terrain.cfg.terrain_generator.size[0] / 2)
    a.max_episode_length_s = float(env.max_episode_length_s)
    a.terrain_levels, a.terrain_types = st["terrain_levels"].data_ptr(), st["terrain_types"].data_ptr()
    a.terrain_origins, a.num_cols = st["terrain_origins"].data_ptr(), st["terrain_origins"].shape[1]
    a.env_origins = env.scene.env_origins.data_ptr()
    inj = getattr(env, "event_uniforms", {}).get("terrain_levels_vel")       # tests: the randint draws, (N,) int64
    a.injected_levels = inj.data_ptr() if inj is not None else None
    a.rng_state = None if inj is not None else _state(env, ("terrain_levels", asset_cfg.name), 7).data_ptr()
    return a, st


def terrain_levels_vel(env, env_ids: Sequence[int], asset_cfg: SceneEntityCfg = SceneEntityCfg("robot")) -> torch.Tensor:
    keep: list = []
    a, st = _fill_terrain_levels_vel(env, env_ids, asset_cfg, keep)
    return _launch_terrain_levels_vel(env, a, st)


def _launch_terrain_levels_vel(env, a, st) -> torch.Tensor:
    # the term's return value, torch.mean(terrain_levels.float()) (:55), written by the launch itself: a fresh
    # one-element tensor per call like the reference's, or a slot of a ring (env.reset_log_ring, no copy launch)
    ws = st.get("mean_ws")
    if ws is None:
        ws = st["mean_ws"] = torch.zeros(2, dtype=torch.int64, device=env.device)
        st["mean_out"] = torch.zeros(1, dtype=torch.float32, device=env.device)
    ring = int(getattr(env, "reset_log_ring", 0) or 0)
    if ring > 0:
        slots = st.get("mean_ring")
        if slots is None or slots.shape[0] != ring:
            slots = st["mean_ring"] = torch.zeros(ring, dtype=torch.float32, device=env.device)
            st["mean_slot"] = 0
        out = slots[st["mean_slot"]: st["mean_slot"] + 1]
        st["mean_slot"] = (st["mean_slot"] + 1) % ring
    else:
        out = st["mean_out"]
    a.mean_level, a.mean_ws = out.data_ptr(), ws.data_ptr()
    L.check(L.lib.mdp_curriculum_terrain_levels_vel(env.num_envs, C.byref(a), L.stream(env.device)))
    return out[0] if ring > 0 else out[0].clone()
