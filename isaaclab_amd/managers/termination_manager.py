"""TerminationManager (``LAB/managers/termination_manager.py:49-268``): OR of the term masks into
``time_outs`` / ``terminated`` plus per-term masks, one fused kernel (``mdp_termination_compute``)."""
from __future__ import annotations

from collections.abc import Sequence

import torch

from .. import _lib as L
from .._terms import TERMINATION, TermSpec
from ._program import Program
from .manager_base import ManagerBase, ManagerTermBase
from .manager_term_cfg import TerminationTermCfg


class TerminationManager(ManagerBase):
    def __init__(self, cfg: object, env):
        self._term_names: list[str] = list()
        self._term_cfgs: list[TerminationTermCfg] = list()
        self._class_term_cfgs: list[TerminationTermCfg] = list()
        self._program: Program | None = None
        self._external: list = []
        super().__init__(cfg, env)
        n_terms = len(self._term_names)
        self._term_dones_block = torch.zeros((max(n_terms, 1), self.num_envs), device=self.device, dtype=torch.bool)
        self._term_dones = {name: self._term_dones_block[i] for i, name in enumerate(self._term_names)}
        self._truncated_buf = torch.zeros(self.num_envs, device=self.device, dtype=torch.bool)
        self._terminated_buf = torch.zeros_like(self._truncated_buf)
        self._dones_buf = torch.zeros_like(self._truncated_buf)

    def __str__(self) -> str:
        msg = f"<TerminationManager> contains {len(self._term_names)} active terms.\n"
        for index, (name, term_cfg) in enumerate(zip(self._term_names, self._term_cfgs)):
            msg += f"  {index:3d} | {name:<32s} | {term_cfg.time_out}\n"
        return msg

    @property
    def active_terms(self) -> list[str]:
        return self._term_names

    @property
    def dones(self) -> torch.Tensor:
        return self._dones_buf

    @property
    def time_outs(self) -> torch.Tensor:
        return self._truncated_buf

    @property
    def terminated(self) -> torch.Tensor:
        return self._terminated_buf

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, torch.Tensor]:
        if env_ids is None:
            env_ids = slice(None)
        extras = {}
        for key in self._term_dones.keys():
            extras["Episode_Termination/" + key] = torch.count_nonzero(self._term_dones[key][env_ids]).item()
        for term_cfg in self._class_term_cfgs:
            term_cfg.func.reset(env_ids=env_ids)
        return extras

    def _build_program(self) -> Program:
        specs = []
        self._external = []
        for term_cfg in self._term_cfgs:
            lower = getattr(term_cfg.func, "lower", None)
            spec = lower(self._env, **term_cfg.params) if lower is not None else None
            if not isinstance(spec, TermSpec):
                staging = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
                self._external.append((term_cfg, staging))
                spec = TermSpec(TERMINATION, "MDP_TERM_EXTERNAL", p={0: staging})
            specs.append(spec)

        def post(n, spec, t):
            t.out_offset = n
            t.i3 = 1 if self._term_cfgs[n].time_out else 0

        self._bound = tuple((id(c.func), bool(c.time_out)) for c in self._term_cfgs)
        return Program(TERMINATION, self.num_envs, self.device, specs, structs_post=post)

    def prepare_launch(self):
        """Everything ``compute`` does before its kernel launch (program rebuild, user terms, lazy inputs)."""
        bound = tuple((id(c.func), bool(c.time_out)) for c in self._term_cfgs)
        if self._program is None or bound != self._bound:
            self._program = self._build_program()
        for term_cfg, staging in self._external:
            value = term_cfg.func(self._env, **term_cfg.params)
            if isinstance(value, torch.Tensor):
                staging.copy_(value)
            else:
                staging.fill_(bool(value))
        self._program.run_before()
        return self._program

    def compute(self) -> torch.Tensor:
        if len(self._term_names) == 0:
            return self._dones_buf
        self.prepare_launch()
        L.check(L.lib.mdp_termination_compute(self._program.handle, self._term_dones_block.data_ptr(),
                                              self._truncated_buf.data_ptr(), self._terminated_buf.data_ptr(),
                                              self._dones_buf.data_ptr(), L.stream(self.device)))
        return self._dones_buf

    def get_term(self, name: str) -> torch.Tensor:
        return self._term_dones[name]

    def get_active_iterable_terms(self, env_idx: int):
        return [(key, [self._term_dones[key][env_idx].float().cpu().item()]) for key in self._term_dones.keys()]

    def set_term_cfg(self, term_name: str, cfg: TerminationTermCfg):
        if term_name not in self._term_names:
            raise ValueError(f"Termination term '{term_name}' not found.")
        self._term_cfgs[self._term_names.index(term_name)] = cfg
        self._program = None

    def get_term_cfg(self, term_name: str) -> TerminationTermCfg:
        if term_name not in self._term_names:
            raise ValueError(f"Termination term '{term_name}' not found.")
        return self._term_cfgs[self._term_names.index(term_name)]

    def _prepare_terms(self):
        for term_name, term_cfg in self._cfg_items():
            if term_cfg is None:
                continue
            if not isinstance(term_cfg, TerminationTermCfg):
                raise TypeError(
                    f"Configuration for the term '{term_name}' is not of type TerminationTermCfg."
                    f" Received: '{type(term_cfg)}'."
                )
            self._resolve_common_term_cfg(term_name, term_cfg, min_argc=1)
            self._term_names.append(term_name)
            self._term_cfgs.append(term_cfg)
            if isinstance(term_cfg.func, ManagerTermBase):
                self._class_term_cfgs.append(term_cfg)
