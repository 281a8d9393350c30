"""``ManagerTermBase`` / ``ManagerBase`` (``LAB/managers/manager_base.py:26-298``): cfg parsing and the
error conventions shared by all managers (TypeError / ValueError / AttributeError at construction)."""
from __future__ import annotations

import copy
import inspect
from abc import ABC, abstractmethod
from collections.abc import Sequence
from typing import Any

from ..utils import string as string_utils
from ..utils.string import string_to_callable
from .manager_term_cfg import ManagerTermBaseCfg
from .scene_entity_cfg import SceneEntityCfg


class ManagerTermBase(ABC):
    def __init__(self, cfg: ManagerTermBaseCfg, env):
        self.cfg = cfg
        self._env = env

    @property
    def num_envs(self) -> int:
        return self._env.num_envs

    @property
    def device(self) -> str:
        return self._env.device

    def reset(self, env_ids: Sequence[int] | None = None) -> None:
        pass

    def __call__(self, *args) -> Any:
        raise NotImplementedError


class ManagerBase(ABC):
    def __init__(self, cfg: object, env):
        self.cfg = copy.deepcopy(cfg)
        self._env = env
        if self.cfg:
            self._prepare_terms()

    @property
    def num_envs(self) -> int:
        return self._env.num_envs

    @property
    def device(self) -> str:
        return self._env.device

    @property
    @abstractmethod
    def active_terms(self) -> list[str] | dict[str, list[str]]:
        raise NotImplementedError

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, float]:
        return {}

    def find_terms(self, name_keys: str | Sequence[str]) -> list[str]:
        if isinstance(self.active_terms, dict):
            names = [n for group in self.active_terms.values() for n in group]
        else:
            names = self.active_terms
        return string_utils.resolve_matching_names(name_keys, names)[1]

    def get_active_iterable_terms(self, env_idx: int):
        raise NotImplementedError

    @abstractmethod
    def _prepare_terms(self):
        raise NotImplementedError

    def _cfg_items(self, cfg=None):
        cfg = self.cfg if cfg is None else cfg
        return cfg.items() if isinstance(cfg, dict) else cfg.__dict__.items()

    def _resolve_common_term_cfg(self, term_name: str, term_cfg: ManagerTermBaseCfg, min_argc: int = 1):
        """What every manager does to a term cfg before it uses it (contract: ``LAB/managers/manager_base.py:219-298``;
        the exception TYPES are the reference's, see SURVEY.md 8b "Errors"): bind scene entities, turn ``func`` into a
        callable (``"module:attr"`` strings, class terms instantiated with ``cfg=`` / ``env=``) and make sure ``params``
        names exactly the arguments ``func`` takes after the ``min_argc`` the manager supplies itself."""
        if not isinstance(term_cfg, ManagerTermBaseCfg):
            raise TypeError(f"term '{term_name}': expected a ManagerTermBaseCfg, got {type(term_cfg)!r}")
        self._bind_scene_entities(term_name, term_cfg)
        term_cfg.func = self._as_callable(term_name, term_cfg)
        self._check_params(term_name, term_cfg, min_argc)

    def _bind_scene_entities(self, term_name: str, term_cfg: ManagerTermBaseCfg) -> None:
        for key, value in term_cfg.params.items():
            if not isinstance(value, SceneEntityCfg):
                continue
            try:
                value.resolve(self._env.scene)        # names -> joint_ids / body_ids (scene_entity_cfg.py:112-177)
            except ValueError as err:
                raise ValueError(f"term '{term_name}', parameter '{key}': {err}") from err

    def _as_callable(self, term_name: str, term_cfg: ManagerTermBaseCfg):
        func = term_cfg.func
        if isinstance(func, str):
            func = string_to_callable(func)
        if inspect.isclass(func):
            if not issubclass(func, ManagerTermBase):
                raise TypeError(f"term '{term_name}': a class term must derive from ManagerTermBase, got {func!r}")
            func = func(cfg=term_cfg, env=self._env)
        if not callable(func):
            raise AttributeError(f"term '{term_name}': {func!r} cannot be called")
        return func

    @staticmethod
    def _check_params(term_name: str, term_cfg: ManagerTermBaseCfg, min_argc: int) -> None:
        sig = inspect.signature(term_cfg.func).parameters
        required = [name for name, p in sig.items() if p.default is inspect.Parameter.empty]
        optional = [name for name, p in sig.items() if p.default is not inspect.Parameter.empty]
        # positional slots the manager fills (env, and env_ids for event / curriculum terms) come first
        from_cfg = (required + optional)[min_argc:]
        if not from_cfg:
            return
        given = set(term_cfg.params)
        if set(from_cfg) != given | set(optional):
            raise ValueError(
                f"term '{term_name}': params {sorted(given)} do not fit the signature -- required beyond the first"
                f" {min_argc}: {required[min_argc:]}, optional: {optional}"
            )
