"""ActionTerm / ActionManager (``LAB/managers/action_manager.py:28-390``): splits the policy action
over terms and keeps ``action`` / ``prev_action`` for reward and observation terms."""
from __future__ import annotations

import inspect
from abc import abstractmethod
from collections.abc import Sequence

import torch

from .manager_base import ManagerBase, ManagerTermBase
from .manager_term_cfg import ActionTermCfg


class ActionTerm(ManagerTermBase):
    def __init__(self, cfg: ActionTermCfg, env):
        super().__init__(cfg, env)
        self._asset = self._env.scene[self.cfg.asset_name]

    @property
    @abstractmethod
    def action_dim(self) -> int:
        raise NotImplementedError

    @property
    @abstractmethod
    def raw_actions(self) -> torch.Tensor:
        raise NotImplementedError

    @property
    @abstractmethod
    def processed_actions(self) -> torch.Tensor:
        raise NotImplementedError

    @abstractmethod
    def process_actions(self, actions: torch.Tensor):
        raise NotImplementedError

    @abstractmethod
    def apply_actions(self):
        raise NotImplementedError


class ActionManager(ManagerBase):
    def __init__(self, cfg: object, env):
        if cfg is None:
            raise ValueError("Action manager configuration is None. Please provide a valid configuration.")
        self._terms: dict[str, ActionTerm] = dict()
        super().__init__(cfg, env)
        self._action = torch.zeros((self.num_envs, self.total_action_dim), device=self.device)
        self._prev_action = torch.zeros_like(self._action)
        # single joint-action term: let the term's fused kernel maintain action / prev_action as well
        self._fused_term = None
        if len(self._terms) == 1:
            term = next(iter(self._terms.values()))
            if hasattr(term, "process_actions_fused"):
                self._fused_term = term

    def __str__(self) -> str:
        msg = f"<ActionManager> contains {len(self._terms)} active terms.\n"
        for index, (name, term) in enumerate(self._terms.items()):
            msg += f"  {index:3d} | {name:<32s} | {term.action_dim}\n"
        return msg

    @property
    def total_action_dim(self) -> int:
        return sum(self.action_term_dim)

    @property
    def active_terms(self) -> list[str]:
        return list(self._terms.keys())

    @property
    def action_term_dim(self) -> list[int]:
        return [term.action_dim for term in self._terms.values()]

    @property
    def action(self) -> torch.Tensor:
        return self._action

    @property
    def prev_action(self) -> torch.Tensor:
        return self._prev_action

    def get_active_iterable_terms(self, env_idx: int):
        terms = []
        idx = 0
        for name, term in self._terms.items():
            term_actions = self._action[env_idx, idx : idx + term.action_dim].cpu()
            terms.append((name, term_actions.tolist()))
            idx += term.action_dim
        return terms

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, torch.Tensor]:
        if env_ids is None:
            env_ids = slice(None)
        self._prev_action[env_ids] = 0.0
        self._action[env_ids] = 0.0
        for term in self._terms.values():
            term.reset(env_ids=env_ids)
        return {}

    def process_action(self, action: torch.Tensor):
        """action_manager.py:318-339"""
        if self.total_action_dim != action.shape[1]:
            raise ValueError(f"Invalid action shape, expected: {self.total_action_dim}, received: {action.shape[1]}.")
        if self._fused_term is not None:
            self._fused_term.process_actions_fused(action, self._action, self._prev_action)
            return
        self._prev_action[:] = self._action
        self._action[:] = action.to(self.device)
        idx = 0
        for term in self._terms.values():
            term.process_actions(action[:, idx : idx + term.action_dim])
            idx += term.action_dim

    def deferred_action(self, action: torch.Tensor):
        """``process_action`` without its launch: returns ``(asset, MdpJointAction)`` for the caller to hand to that
        asset's first ``write_data_to_sim`` of the step (the actuator launch then processes the action itself), or
        ``None`` when the terms do not allow it -- the caller then calls ``process_action``."""
        term = self._fused_term
        if term is None or not term.can_fold_into_actuator():
            return None
        if self.total_action_dim != action.shape[1]:
            raise ValueError(f"Invalid action shape, expected: {self.total_action_dim}, received: {action.shape[1]}.")
        return term._asset, term.fill_joint_action(action, self._action, self._prev_action)

    def apply_action(self) -> None:
        for term in self._terms.values():
            term.apply_actions()

    def get_term(self, name: str) -> ActionTerm:
        return self._terms[name]

    def _prepare_terms(self):
        for term_name, term_cfg in self._cfg_items():
            if term_cfg is None:
                continue
            if not isinstance(term_cfg, ActionTermCfg):
                raise TypeError(
                    f"Configuration for the term '{term_name}' is not of type ActionTermCfg."
                    f" Received: '{type(term_cfg)}'."
                )
            term = term_cfg.class_type(term_cfg, self._env)
            if not isinstance(term, ActionTerm):
                raise TypeError(f"Returned object for the term '{term_name}' is not of type ActionType.")
            self._terms[term_name] = term
