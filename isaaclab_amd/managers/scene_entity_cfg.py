"""``SceneEntityCfg`` (``LAB/managers/scene_entity_cfg.py:17-290``): entity name + joint/body selectors that
are resolved to index lists (or ``slice(None)`` for "all, in order") once at manager construction."""
from __future__ import annotations

from dataclasses import MISSING

from ..utils.configclass import configclass


@configclass
class SceneEntityCfg:
    name: str = MISSING
    joint_names: str | list[str] | None = None
    joint_ids: list[int] | slice = slice(None)
    fixed_tendon_names: str | list[str] | None = None
    fixed_tendon_ids: list[int] | slice = slice(None)
    body_names: str | list[str] | None = None
    body_ids: list[int] | slice = slice(None)
    object_collection_names: str | list[str] | None = None
    object_collection_ids: list[int] | slice = slice(None)
    preserve_order: bool = False

    def resolve(self, scene):
        if self.name not in scene.keys():
            raise ValueError(f"The scene entity '{self.name}' does not exist. Available entities: {scene.keys()}.")
        self._resolve("joint", "find_joints", "joint_names", "num_joints", scene, collapse=True)
        self._resolve("fixed_tendon", "find_fixed_tendons", "fixed_tendon_names", "num_fixed_tendons", scene, collapse=True)
        self._resolve("body", "find_bodies", "body_names", "num_bodies", scene, collapse=True)
        self._resolve("object_collection", "find_objects", "object_names", "num_objects", scene, collapse=False)

    def _resolve(self, kind: str, finder: str, entity_names_attr: str, count_attr: str, scene, collapse: bool):
        names_attr, ids_attr = f"{kind}_names", f"{kind}_ids"
        names, ids = getattr(self, names_attr), getattr(self, ids_attr)
        if names is None and ids == slice(None):
            return
        entity = scene[self.name]
        if isinstance(names, str):
            names = [names]
            setattr(self, names_attr, names)
        if isinstance(ids, int):
            ids = [ids]
            setattr(self, ids_attr, ids)
        all_names = getattr(entity, entity_names_attr)
        if names is not None and ids != slice(None):
            # both given: they must agree (scene_entity_cfg.py:150-163)
            found_ids, _ = getattr(entity, finder)(names, preserve_order=self.preserve_order)
            found_names = [all_names[i] for i in ids]
            if found_ids != ids or found_names != names:
                label = kind.replace("_", " ")
                raise ValueError(
                    f"Both '{names_attr}' and '{ids_attr}' are specified, and are not consistent."
                    f"\n\tfrom {label} names: {names} [{found_ids}]"
                    f"\n\tfrom {label} ids: {found_names} [{ids}]"
                    f"\nHint: Use either '{names_attr}' or '{ids_attr}' to avoid confusion."
                )
        elif names is not None:
            found_ids, _ = getattr(entity, finder)(names, preserve_order=self.preserve_order)
            # all entries in entity order -> slice (cheaper indexing, scene_entity_cfg.py:169-172)
            if collapse and len(found_ids) == getattr(entity, count_attr) and names == all_names:
                found_ids = slice(None)
            setattr(self, ids_attr, found_ids)
        else:
            setattr(self, names_attr, [all_names[i] for i in ids])
