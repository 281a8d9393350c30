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
        if not isinstance(term_cfg, ManagerTermBaseCfg):
            raise TypeError(
                f"Configuration for the term '{term_name}' is not of type ManagerTermBaseCfg."
                f" Received: '{type(term_cfg)}'."
            )
        for key, value in term_cfg.params.items():
            if isinstance(value, SceneEntityCfg):
                try:
                    value.resolve(self._env.scene)
                except ValueError as e:
                    raise ValueError(f"Error while parsing '{term_name}:{key}'. {e}")
            term_cfg.params[key] = value
        if isinstance(term_cfg.func, str):
            term_cfg.func = string_to_callable(term_cfg.func)
        if inspect.isclass(term_cfg.func):
            if not issubclass(term_cfg.func, ManagerTermBase):
                raise TypeError(
                    f"Configuration for the term '{term_name}' is not of type ManagerTermBase."
                    f" Received: '{type(term_cfg.func)}'."
                )
            term_cfg.func = term_cfg.func(cfg=term_cfg, env=self._env)
        if not callable(term_cfg.func):
            raise AttributeError(f"The term '{term_name}' is not callable. Received: {term_cfg.func}")
        # params must cover exactly the non-default arguments after the first `min_argc`
        term_params = list(term_cfg.params.keys())
        sig = inspect.signature(term_cfg.func).parameters
        with_defaults = [a for a in sig if sig[a].default is not inspect.Parameter.empty]
        without_defaults = [a for a in sig if sig[a].default is inspect.Parameter.empty]
        args = without_defaults + with_defaults
        if len(args) > min_argc:
            if set(args[min_argc:]) != set(term_params + with_defaults):
                raise ValueError(
                    f"The term '{term_name}' expects mandatory parameters: {without_defaults[min_argc:]}"
                    f" and optional parameters: {with_defaults}, but received: {term_params}."
                )
