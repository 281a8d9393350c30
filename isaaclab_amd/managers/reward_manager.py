"""RewardManager (``LAB/managers/reward_manager.py:43-246``): ``reward = sum_i func_i * weight_i * dt`` with
episodic sums and the per-step term values, evaluated by one fused kernel (``mdp_reward_compute``)."""
from __future__ import annotations

from collections.abc import Sequence

import torch

from .. import _lib as L
from .._terms import REWARD, TermSpec
from ._program import Program
from .manager_base import ManagerBase, ManagerTermBase
from .manager_term_cfg import RewardTermCfg


class RewardManager(ManagerBase):
    def __init__(self, cfg: object, env):
        self._term_names: list[str] = list()
        self._term_cfgs: list[RewardTermCfg] = list()
        self._class_term_cfgs: list[RewardTermCfg] = list()
        self._program: Program | None = None
        self._bound_weights = None
        self._external: list = []
        super().__init__(cfg, env)
        n_terms = len(self._term_names)
        # one (T, N) block; the per-term dict entries are row views of it
        self._episode_sums_block = torch.zeros((max(n_terms, 1), self.num_envs), dtype=torch.float, device=self.device)
        self._episode_sums = {name: self._episode_sums_block[i] for i, name in enumerate(self._term_names)}
        self._reward_buf = torch.zeros(self.num_envs, dtype=torch.float, device=self.device)
        self._step_reward = torch.zeros((self.num_envs, n_terms), dtype=torch.float, device=self.device)

    def __str__(self) -> str:
        msg = f"<RewardManager> contains {len(self._term_names)} active terms.\n"
        for index, (name, term_cfg) in enumerate(zip(self._term_names, self._term_cfgs)):
            msg += f"  {index:3d} | {name:<32s} | {term_cfg.weight}\n"
        return msg

    @property
    def active_terms(self) -> list[str]:
        return self._term_names

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, torch.Tensor]:
        if env_ids is None:
            env_ids = slice(None)
        extras = {}
        for key in self._episode_sums.keys():
            episodic_sum_avg = torch.mean(self._episode_sums[key][env_ids])
            extras["Episode_Reward/" + key] = episodic_sum_avg / self._env.max_episode_length_s
            self._episode_sums[key][env_ids] = 0.0
        for term_cfg in self._class_term_cfgs:
            term_cfg.func.reset(env_ids=env_ids)
        return extras

    def _build_program(self) -> Program:
        specs = []
        self._external = []
        for term_cfg in self._term_cfgs:
            lower = getattr(term_cfg.func, "lower", None)
            if term_cfg.weight == 0.0:
                # skipped like the reference (reward_manager.py:145-146): never evaluated
                spec = TermSpec(REWARD, "MDP_REW_EXTERNAL", p={0: self._reward_buf}, i={0: 1})
            elif lower is not None:
                spec = lower(self._env, **term_cfg.params)
            else:
                staging = torch.zeros(self.num_envs, dtype=torch.float32, device=self.device)
                self._external.append((term_cfg, staging))
                spec = TermSpec(REWARD, "MDP_REW_EXTERNAL", p={0: staging}, i={0: 1})
            specs.append(spec)

        def post(n, spec, t):
            t.out_offset = n
            t.weight = float(self._term_cfgs[n].weight)

        self._bound_weights = tuple(float(c.weight) for c in self._term_cfgs)
        self._bound_funcs = tuple(id(c.func) for c in self._term_cfgs)
        return Program(REWARD, self.num_envs, self.device, specs, structs_post=post)

    def prepare_launch(self):
        """Everything ``compute`` does before its kernel launch: (re)build the program when weights / functions
        changed, evaluate user-defined (non-lowered) terms, refresh lazily evaluated inputs."""
        weights = tuple(float(c.weight) for c in self._term_cfgs)
        funcs = tuple(id(c.func) for c in self._term_cfgs)
        if self._program is None or weights != self._bound_weights or funcs != self._bound_funcs:
            self._program = self._build_program()
        for term_cfg, staging in self._external:
            staging.copy_(term_cfg.func(self._env, **term_cfg.params))
        self._program.run_before()
        return self._program

    def compute(self, dt: float) -> torch.Tensor:
        if len(self._term_names) == 0:
            self._reward_buf[:] = 0.0
            return self._reward_buf
        self.prepare_launch()
        L.check(L.lib.mdp_reward_compute(self._program.handle, float(dt), self._reward_buf.data_ptr(),
                                         self._episode_sums_block.data_ptr(), self._step_reward.data_ptr(),
                                         L.stream(self.device)))
        return self._reward_buf

    def set_term_cfg(self, term_name: str, cfg: RewardTermCfg):
        if term_name not in self._term_names:
            raise ValueError(f"Reward term '{term_name}' not found.")
        self._term_cfgs[self._term_names.index(term_name)] = cfg
        self._program = None

    def get_term_cfg(self, term_name: str) -> RewardTermCfg:
        if term_name not in self._term_names:
            raise ValueError(f"Reward term '{term_name}' not found.")
        return self._term_cfgs[self._term_names.index(term_name)]

    def get_active_iterable_terms(self, env_idx: int):
        return [(name, [self._step_reward[env_idx, idx].cpu().item()]) for idx, name in enumerate(self._term_names)]

    def _prepare_terms(self):
        for term_name, term_cfg in self._cfg_items():
            if term_cfg is None:
                continue
            if not isinstance(term_cfg, RewardTermCfg):
                raise TypeError(
                    f"Configuration for the term '{term_name}' is not of type RewardTermCfg."
                    f" Received: '{type(term_cfg)}'."
                )
            if not isinstance(term_cfg.weight, (float, int)):
                raise TypeError(
                    f"Weight for the term '{term_name}' is not of type float or int."
                    f" Received: '{type(term_cfg.weight)}'."
                )
            self._resolve_common_term_cfg(term_name, term_cfg, min_argc=1)
            self._term_names.append(term_name)
            self._term_cfgs.append(term_cfg)
            if isinstance(term_cfg.func, ManagerTermBase):
                self._class_term_cfgs.append(term_cfg)
