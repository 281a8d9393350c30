"""``CurriculumManager`` (``LAB/managers/curriculum_manager.py:24-175``): terms are called at reset with the envs being
reset; their return value is the logged state.  ``env_ids`` is handed through as is (ids, ``slice(None)`` or a device
mask); ``reset`` keeps tensor states as device tensors when ``sync_free`` is set (the reference calls ``.item()``)."""
from __future__ import annotations

from collections.abc import Sequence

import torch

from .manager_base import ManagerBase, ManagerTermBase
from .manager_term_cfg import CurriculumTermCfg


class CurriculumManager(ManagerBase):
    def __init__(self, cfg: object, env):
        self._term_names: list[str] = list()
        self._term_cfgs: list[CurriculumTermCfg] = list()
        self._class_term_cfgs: list[CurriculumTermCfg] = list()
        self.sync_free = False
        super().__init__(cfg, env)
        self._curriculum_state = dict()
        for term_name in self._term_names:
            self._curriculum_state[term_name] = None

    @property
    def active_terms(self) -> list[str]:
        return self._term_names

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, float]:
        """curriculum_manager.py:88-117: logs the state of every term (``env_ids`` is not used for that)."""
        extras = {}
        conv = (lambda v: v) if self.sync_free else (lambda v: v.item() if isinstance(v, torch.Tensor) else v)
        for term_name, term_state in self._curriculum_state.items():
            if term_state is not None:
                if isinstance(term_state, dict):
                    for key, value in term_state.items():
                        extras[f"Curriculum/{term_name}/{key}"] = conv(value)
                else:
                    extras[f"Curriculum/{term_name}"] = conv(term_state)
        for term_cfg in self._class_term_cfgs:
            term_cfg.func.reset(env_ids=env_ids)
        return extras

    def compute(self, env_ids: Sequence[int] | None = None):
        """curriculum_manager.py:119-134."""
        if env_ids is None:
            env_ids = slice(None)
        for name, term_cfg in zip(self._term_names, self._term_cfgs):
            self._curriculum_state[name] = term_cfg.func(self._env, env_ids, **term_cfg.params)

    def _prepare_terms(self):
        """curriculum_manager.py:151-175."""
        for term_name, term_cfg in self._cfg_items():
            if term_cfg is None:
                continue
            if not isinstance(term_cfg, CurriculumTermCfg):
                raise TypeError(
                    f"Configuration for the term '{term_name}' is not of type CurriculumTermCfg."
                    f" Received: '{type(term_cfg)}'."
                )
            self._resolve_common_term_cfg(term_name, term_cfg, min_argc=2)
            self._term_names.append(term_name)
            self._term_cfgs.append(term_cfg)
            if isinstance(term_cfg.func, ManagerTermBase):
                self._class_term_cfgs.append(term_cfg)
