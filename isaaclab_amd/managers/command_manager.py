"""CommandTerm / CommandManager (``LAB/managers/command_manager.py:32-394``)."""
from __future__ import annotations

import inspect
from abc import abstractmethod
from collections.abc import Sequence

import torch

from .manager_base import ManagerBase, ManagerTermBase
from .manager_term_cfg import CommandTermCfg


class CommandTerm(ManagerTermBase):
    """Base of command generators: owns ``metrics``, ``time_left`` and ``command_counter``."""

    def __init__(self, cfg: CommandTermCfg, env):
        super().__init__(cfg, env)
        self.metrics = dict()
        self.time_left = torch.zeros(self.num_envs, device=self.device)
        self.command_counter = torch.zeros(self.num_envs, device=self.device, dtype=torch.long)

    @property
    @abstractmethod
    def command(self) -> torch.Tensor:
        raise NotImplementedError

    @property
    def has_debug_vis_implementation(self) -> bool:
        return False

    def set_debug_vis(self, debug_vis: bool) -> bool:
        return False

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, float]:
        """command_manager.py:120-149 (generic path; subclasses may fuse it)."""
        if env_ids is None:
            env_ids = slice(None)
        extras = {}
        for metric_name, metric_value in self.metrics.items():
            extras[metric_name] = torch.mean(metric_value[env_ids]).item()
            metric_value[env_ids] = 0.0
        self.command_counter[env_ids] = 0
        self._resample(env_ids)
        return extras

    def compute(self, dt: float):
        """command_manager.py:151-166 (generic path with the reference's host sync; fused subclasses override)."""
        self._update_metrics()
        self.time_left -= dt
        resample_env_ids = (self.time_left <= 0.0).nonzero().flatten()
        if len(resample_env_ids) > 0:
            self._resample(resample_env_ids)
        self._update_command()

    def _resample(self, env_ids: Sequence[int]):
        if isinstance(env_ids, slice) or len(env_ids) != 0:
            self.time_left[env_ids] = self.time_left[env_ids].uniform_(*self.cfg.resampling_time_range)
            self.command_counter[env_ids] += 1
            self._resample_command(env_ids)

    @abstractmethod
    def _update_metrics(self):
        raise NotImplementedError

    @abstractmethod
    def _resample_command(self, env_ids: Sequence[int]):
        raise NotImplementedError

    @abstractmethod
    def _update_command(self):
        raise NotImplementedError


class CommandManager(ManagerBase):
    def __init__(self, cfg: object, env):
        self._terms: dict[str, CommandTerm] = dict()
        super().__init__(cfg, env)
        self._commands = dict()
        if self.cfg:
            self.cfg.debug_vis = False if not hasattr(self.cfg, "debug_vis") else self.cfg.debug_vis

    def __str__(self) -> str:
        msg = f"<CommandManager> contains {len(self._terms.values())} active terms.\n"
        for index, (name, term) in enumerate(self._terms.items()):
            msg += f"  {index:3d} | {name:<32s} | {term.__class__.__name__}\n"
        return msg

    @property
    def active_terms(self) -> list[str]:
        return list(self._terms.keys())

    @property
    def has_debug_vis_implementation(self) -> bool:
        return any(term.has_debug_vis_implementation for term in self._terms.values())

    def get_active_iterable_terms(self, env_idx: int):
        return [(key, term.command[env_idx].cpu().tolist()) for key, term in self._terms.items()]

    def set_debug_vis(self, debug_vis: bool):
        for term in self._terms.values():
            term.set_debug_vis(debug_vis)

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, torch.Tensor]:
        if env_ids is None:
            env_ids = slice(None)
        extras = {}
        for name, term in self._terms.items():
            metrics = term.reset(env_ids=env_ids)
            for metric_name, metric_value in metrics.items():
                extras[f"Metrics/{name}/{metric_name}"] = metric_value
        return extras

    def compute(self, dt: float):
        for term in self._terms.values():
            term.compute(dt)

    def get_command(self, name: str) -> torch.Tensor:
        return self._terms[name].command

    def get_term(self, name: str) -> CommandTerm:
        return self._terms[name]

    def _prepare_terms(self):
        for term_name, term_cfg in self._cfg_items():
            if term_cfg is None or term_name == "debug_vis":
                continue
            if not isinstance(term_cfg, CommandTermCfg):
                raise TypeError(
                    f"Configuration for the term '{term_name}' is not of type CommandTermCfg."
                    f" Received: '{type(term_cfg)}'."
                )
            term = term_cfg.class_type(term_cfg, self._env)
            if not isinstance(term, CommandTerm):
                raise TypeError(f"Returned object for the term '{term_name}' is not of type CommandType.")
            self._terms[term_name] = term
