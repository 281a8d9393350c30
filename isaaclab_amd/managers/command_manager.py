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
        """Episode end of ``env_ids`` (contract: command_manager.py:120-149): report the mean of every metric over those
        envs, clear metrics and resample counters, draw a new command.  The means come back to the host together (one
        transfer for all metrics); fused subclasses keep even that on the device."""
        rows = slice(None) if env_ids is None else env_ids
        names = list(self.metrics)
        means = torch.stack([self.metrics[k][rows].mean() for k in names]).tolist() if names else []
        for k in names:
            self.metrics[k][rows] = 0.0
        self.command_counter[rows] = 0
        self._resample(rows)
        return dict(zip(names, means))

    def compute(self, dt: float):
        """One env step of the generator (contract: command_manager.py:151-166): metrics, countdown, a new command where
        the countdown ran out, then the per-step command update."""
        self._update_metrics()
        self.time_left -= dt
        self._resample_where(self.time_left <= 0.0)
        self._update_command()

    def _resample_where(self, expired: torch.Tensor) -> None:
        """Resampling for the envs of a device mask.  Countdown and counter are selected on the device; a term that
        implements ``_resample_command_masked(mask)`` never leaves it.  Terms that only have the reference's
        ``_resample_command(env_ids)`` get their id list from one ``nonzero()`` (the host sync the reference pays too)."""
        lo, hi = self.cfg.resampling_time_range
        fresh = torch.empty_like(self.time_left).uniform_(lo, hi)
        torch.where(expired, fresh, self.time_left, out=self.time_left)
        self.command_counter += expired
        masked = getattr(self, "_resample_command_masked", None)
        if masked is not None:
            masked(expired)
            return
        env_ids = expired.nonzero().flatten()
        if env_ids.numel() > 0:
            self._resample_command(env_ids)

    def _resample(self, env_ids: Sequence[int]):
        """The id-list form the reference exposes (``slice`` or ids; nothing happens for an empty list)."""
        if not isinstance(env_ids, slice) and len(env_ids) == 0:
            return
        lo, hi = self.cfg.resampling_time_range
        count = self.time_left[env_ids].shape
        self.time_left[env_ids] = torch.empty(count, device=self.device).uniform_(lo, hi)
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
        rows = [f"  {i:3d} | {name:<32s} | {type(term).__name__}" for i, (name, term) in enumerate(self._terms.items())]
        return "\n".join([f"<CommandManager> contains {len(self._terms)} active terms."] + rows) + "\n"

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
        """``Metrics/<term>/<metric>`` of the envs that end their episode (command_manager.py:351-356)."""
        rows = slice(None) if env_ids is None else env_ids
        return {f"Metrics/{name}/{metric}": value
                for name, term in self._terms.items() for metric, value in term.reset(env_ids=rows).items()}

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
