"""Scene container (``LAB/scene/interactive_scene.py:432-513`` subset): entity lookup by name,
``write_data_to_sim`` / ``update`` / ``reset`` fan-out.  USD cloning / spawning is out of scope."""
from __future__ import annotations

import torch


class InteractiveScene:
    def __init__(self, cfg, num_envs: int, device: str, env_origins: torch.Tensor | None = None, terrain=None):
        self.cfg = cfg
        self.num_envs = num_envs
        self.device = device
        self.articulations: dict = dict()
        self.rigid_objects: dict = dict()
        self.sensors: dict = dict()
        self.terrain = terrain
        self.env_origins = (env_origins if env_origins is not None
                            else torch.zeros((num_envs, 3), dtype=torch.float32, device=device))
        self.lazy_sensor_update = True
        self.fused_update = True

    def keys(self) -> list[str]:
        keys = ["terrain"] if self.terrain is not None else []
        for group in (self.articulations, self.rigid_objects, self.sensors):
            keys += list(group.keys())
        return keys

    def __getitem__(self, key: str):
        if key == "terrain":
            return self.terrain
        for group in (self.articulations, self.rigid_objects, self.sensors):
            if key in group:
                return group[key]
        raise KeyError(f"Scene entity with key '{key}' not found. Available Entities: '{self.keys()}'")

    def reset(self, env_ids=None):
        for group in (self.articulations, self.rigid_objects, self.sensors):
            for entity in group.values():
                entity.reset(env_ids)

    def write_data_to_sim(self, joint_action=None, gate=None, recast=None):
        """interactive_scene.py:484-494.  ``joint_action = (asset, L.MdpJointAction)``: that asset's actuator launch
        also processes the action of the step (``ActionManager.deferred_action``)."""
        for articulation in self.articulations.values():
            if joint_action is not None and articulation is joint_action[0]:
                articulation.write_data_to_sim(joint_action=joint_action[1])
            else:
                articulation.write_data_to_sim(gate=gate, recast=recast)
        for obj in self.rigid_objects.values():
            obj.write_data_to_sim()

    def update(self, dt: float, skip_ticks=()) -> None:
        """interactive_scene.py:496-513.  With one articulation and sensors that know how to describe their share
        (``fill_scene_update`` / ``fill_scene_tick``) the whole fan-out is ONE kernel launch (``mdp_scene_update``);
        anything else goes through the per-entity ``update`` calls below."""
        if self.fused_update and self._update_fused(dt, skip_ticks):
            return
        if skip_ticks:
            raise RuntimeError("skip_ticks needs the fused scene update")
        for articulation in self.articulations.values():
            articulation.update(dt)
        for obj in self.rigid_objects.values():
            obj.update(dt)
        for sensor in self.sensors.values():
            sensor.update(dt, force_recompute=not self.lazy_sensor_update)

    def _scene_update_desc(self, dt: float, skip_ticks=()):
        """Descriptor of the fused post-physics launch, or None when some entity cannot describe its share.
        ``skip_ticks``: lazily evaluated sensors whose bookkeeping the caller does itself (cast kernel pre-tick)."""
        from .. import _lib as L

        if len(self.articulations) != 1 or self.rigid_objects:
            return None
        art = next(iter(self.articulations.values()))
        if not hasattr(art.data, "fill_scene_update"):
            return None
        d = L.MdpSceneUpdate()
        force = not self.lazy_sensor_update
        seen_contact = False
        for sensor in self.sensors.values():
            if hasattr(sensor, "fill_scene_update") and not seen_contact:
                if not sensor.fill_scene_update(d, force):
                    return None
                seen_contact = True
            elif hasattr(sensor, "fill_scene_tick"):
                if any(sensor is sk for sk in skip_ticks):
                    continue
                if not sensor.fill_scene_tick(d, force):
                    return None
            else:
                return None
        art.data.fill_scene_update(d, dt)
        return d, art

    def _update_fused(self, dt: float, skip_ticks=()) -> bool:
        import ctypes as C

        from .. import _lib as L

        plan = self._scene_update_desc(dt, skip_ticks)
        if plan is None:
            return False
        d, art = plan
        L.check(L.lib.mdp_scene_update(C.byref(d), float(dt), L.stream(self.device)))
        art.data.commit_scene_update(dt)
        return True

    def update_and_write(self, dt: float) -> None:
        """``update(dt)`` immediately followed by ``write_data_to_sim()`` -- the tail of physics sub-step k and the head
        of sub-step k+1 (manager_based_rl_env.py:184-195), which have no physics step between them.  When the single
        actuator group can carry it, both run as ONE launch (``mdp_actuator_*_compute_after_update``)."""
        plan = self._scene_update_desc(dt) if self.fused_update else None
        if plan is None or not plan[1].can_carry_update:
            self.update(dt)
            self.write_data_to_sim()
            return
        d, art = plan
        art.data.commit_scene_update(dt)      # the buffers are fresh once the launch below has run
        art.write_data_to_sim(carry_update=(d, dt))
