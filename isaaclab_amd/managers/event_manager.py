"""``EventManager`` (``LAB/managers/event_manager.py:59-330``) for the modes that sit on the step path: ``"reset"``
and ``"interval"`` (``"startup"`` / ``"prestartup"`` terms randomise physics properties once and are called as in the
reference, with ``env_ids=None``).

The reference turns the interval timers and the reset trigger test into id lists with ``nonzero()`` on the host
(event_manager.py:218, 256-258).  Here they stay masks on the device -- ``mdp_event_interval_tick`` updates the
timers and writes the trigger mask in one launch -- and the terms are called with that mask (the stock terms of
``isaaclab_amd.envs.mdp.events`` take masks; a user term that needs ids can call ``nonzero()`` itself)."""
from __future__ import annotations

from collections.abc import Sequence

import ctypes as C

import torch

from .. import _lib as L
from .manager_base import ManagerBase, ManagerTermBase
from .manager_term_cfg import EventTermCfg


class EventManager(ManagerBase):
    def __init__(self, cfg: object, env):
        self._mode_term_names: dict[str, list[str]] = dict()
        self._mode_term_cfgs: dict[str, list[EventTermCfg]] = dict()
        self._mode_class_term_cfgs: dict[str, list[EventTermCfg]] = dict()
        self._interval_term_time_left: list[torch.Tensor] = list()
        self._interval_masks: list[torch.Tensor] = list()
        self._interval_rng: list[torch.Tensor] = list()
        self._reset_term_last_triggered_step_id: list[torch.Tensor] = list()
        self._reset_term_last_triggered_once: list[torch.Tensor] = list()
        #: tests: ``interval_uniforms[i]`` = (N,) (or (1,) for a global timer) uniforms term i consumes when it fires
        self.interval_uniforms: dict[int, torch.Tensor] = dict()
        super().__init__(cfg, env)

    @property
    def active_terms(self) -> dict[str, list[str]]:
        return self._mode_term_names

    @property
    def available_modes(self) -> list[str]:
        return list(self._mode_term_names.keys())

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, float]:
        """event_manager.py:120-145.  (Like the reference, only the timers of CLASS interval terms are resampled here:
        the loop runs over ``_mode_class_term_cfgs``; function terms keep their timers across episodes.)"""
        for mode_cfg in self._mode_class_term_cfgs.values():
            for term_cfg in mode_cfg:
                term_cfg.func.reset(env_ids=env_ids)
        if "interval" in self._mode_class_term_cfgs:
            for index, term_cfg in enumerate(self._mode_class_term_cfgs["interval"]):
                if not term_cfg.is_global_time:
                    lower, upper = term_cfg.interval_range_s
                    tl = self._interval_term_time_left[index]
                    sampled = torch.rand(self.num_envs, device=self.device) * (upper - lower) + lower
                    if env_ids is None:
                        tl.copy_(sampled)
                    else:
                        ids = torch.as_tensor(env_ids, device=self.device)
                        if ids.dtype in (torch.uint8, torch.bool):
                            torch.where(ids.bool(), sampled, tl, out=tl)
                        else:
                            tl[ids] = sampled[: len(ids)]
        return {}

    def apply(self, mode: str, env_ids=None, dt: float | None = None, global_env_step_count: int | None = None):
        """event_manager.py:147-268.  ``env_ids``: None, ids, or a byte / bool mask of ``num_envs`` entries."""
        if mode not in self._mode_term_names:
            return
        if mode == "interval" and dt is None:
            raise ValueError(f"Event mode '{mode}' requires the time-step of the environment.")
        if mode == "interval" and env_ids is not None:
            raise ValueError(
                f"Event mode '{mode}' does not require environment indices. This is an undefined behavior"
                " as the environment indices are computed based on the time left for each environment."
            )
        if mode == "reset" and global_env_step_count is None:
            raise ValueError(f"Event mode '{mode}' requires the total number of environment steps to be provided.")

        marked = False
        if mode == "reset" and isinstance(env_ids, torch.Tensor) and env_ids.dtype in (torch.uint8, torch.bool) \
                and env_ids.is_cuda and env_ids.is_contiguous():
            # the sync-free reset path (a device mask): the bookkeeping of every term without a minimum step count
            # (event_manager.py:235-240) in ONE launch instead of two index-put kernels per term
            idx = [i for i, c in enumerate(self._mode_term_cfgs[mode]) if c.min_step_count_between_reset == 0]
            if 0 < len(idx) <= 8:
                a = L.MdpEventResetMark()
                a.num_terms, a.step_count = len(idx), int(global_env_step_count)
                for k, i in enumerate(idx):
                    a.last_triggered_step[k] = self._reset_term_last_triggered_step_id[i].data_ptr()
                    a.triggered_once[k] = self._reset_term_last_triggered_once[i].data_ptr()
                L.check(L.lib.mdp_event_reset_mark(self.num_envs, env_ids.data_ptr(), C.byref(a), L.stream(self.device)))
                marked = True
        for index, term_cfg in enumerate(self._mode_term_cfgs[mode]):
            if mode == "interval":
                lower, upper = term_cfg.interval_range_s
                mask = self._interval_masks[index]
                u = self.interval_uniforms.get(index)
                L.check(L.lib.mdp_event_interval_tick(
                    self.num_envs, self._interval_term_time_left[index].data_ptr(), float(dt), float(lower),
                    float(upper - lower), 1 if term_cfg.is_global_time else 0, u.data_ptr() if u is not None else None,
                    None if u is not None else self._interval_rng[index].data_ptr(), mask.data_ptr(),
                    L.stream(self.device)))
                # the term runs for the envs whose timer fired (a global timer fires for all of them or for none)
                term_cfg.func(self._env, mask, **term_cfg.params)
            elif mode == "reset":
                min_step_count = term_cfg.min_step_count_between_reset
                if min_step_count == 0:
                    if not marked:
                        sel = slice(None) if env_ids is None else self._as_index(env_ids)
                        self._reset_term_last_triggered_step_id[index][sel] = global_env_step_count
                        self._reset_term_last_triggered_once[index][sel] = True
                    term_cfg.func(self._env, env_ids, **term_cfg.params)
                else:
                    # event_manager.py:242-268 with masks instead of id lists
                    last = self._reset_term_last_triggered_step_id[index]
                    once = self._reset_term_last_triggered_once[index]
                    valid = (global_env_step_count - last) >= min_step_count
                    valid |= (last == 0) & ~once
                    valid &= self._as_mask(env_ids)
                    last.masked_fill_(valid, global_env_step_count)
                    once |= valid
                    term_cfg.func(self._env, valid, **term_cfg.params)
            else:
                term_cfg.func(self._env, env_ids, **term_cfg.params)

    # -- helpers ---------------------------------------------------------------------------------------
    def _as_mask(self, env_ids) -> torch.Tensor:
        if env_ids is None:
            return torch.ones(self.num_envs, dtype=torch.bool, device=self.device)
        t = torch.as_tensor(env_ids, device=self.device)
        if t.dtype in (torch.uint8, torch.bool):
            return t.bool()
        m = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        m[t.long()] = True
        return m

    def _as_index(self, env_ids):
        t = torch.as_tensor(env_ids, device=self.device)
        return t.bool() if t.dtype in (torch.uint8, torch.bool) else t.long()

    def set_term_cfg(self, term_name: str, cfg: EventTermCfg):
        for mode, names in self._mode_term_names.items():
            if term_name in names:
                self._mode_term_cfgs[mode][names.index(term_name)] = cfg
                return
        raise ValueError(f"Event term '{term_name}' not found.")

    def get_term_cfg(self, term_name: str) -> EventTermCfg:
        for mode, names in self._mode_term_names.items():
            if term_name in names:
                return self._mode_term_cfgs[mode][names.index(term_name)]
        raise ValueError(f"Event term '{term_name}' not found.")

    def _prepare_terms(self):
        """event_manager.py:298-372."""
        for term_name, term_cfg in self._cfg_items():
            if term_cfg is None:
                continue
            if not isinstance(term_cfg, EventTermCfg):
                raise TypeError(
                    f"Configuration for the term '{term_name}' is not of type EventTermCfg."
                    f" Received: '{type(term_cfg)}'."
                )
            if term_cfg.mode != "reset" and term_cfg.min_step_count_between_reset != 0:
                pass                                        # the reference only warns
            self._resolve_common_term_cfg(term_name, term_cfg, min_argc=2)
            if term_cfg.mode not in self._mode_term_names:
                self._mode_term_names[term_cfg.mode] = list()
                self._mode_term_cfgs[term_cfg.mode] = list()
                self._mode_class_term_cfgs[term_cfg.mode] = list()
            self._mode_term_names[term_cfg.mode].append(term_name)
            self._mode_term_cfgs[term_cfg.mode].append(term_cfg)
            if isinstance(term_cfg.func, ManagerTermBase):
                self._mode_class_term_cfgs[term_cfg.mode].append(term_cfg)
            if term_cfg.mode == "interval":
                if term_cfg.interval_range_s is None:
                    raise ValueError(
                        f"Event term '{term_name}' has mode 'interval' but 'interval_range_s' is not specified."
                    )
                lower, upper = term_cfg.interval_range_s
                count = 1 if term_cfg.is_global_time else self.num_envs
                # event_manager.py:344-356: the first interval is sampled at construction
                time_left = torch.rand(count, device=self.device if not term_cfg.is_global_time else "cpu") * (upper - lower) + lower
                self._interval_term_time_left.append(time_left.to(self.device).contiguous())
                self._interval_masks.append(torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device))
                rng = torch.zeros(1 + count, dtype=torch.int64, device=self.device)
                rng[0] = (int(getattr(getattr(self._env, "cfg", None), "seed", 0) or 0) * 0x9E3779B1 + 0x49564C + len(self._interval_rng)) & 0x7FFFFFFFFFFFFFFF
                self._interval_rng.append(rng)
            elif term_cfg.mode == "reset":
                if term_cfg.min_step_count_between_reset < 0:
                    raise ValueError(
                        f"Event term '{term_name}' has mode 'reset' but 'min_step_count_between_reset' is"
                        f" negative: {term_cfg.min_step_count_between_reset}. Please provide a non-negative value."
                    )
                self._reset_term_last_triggered_step_id.append(torch.zeros(self.num_envs, dtype=torch.int32, device=self.device))
                self._reset_term_last_triggered_once.append(torch.zeros(self.num_envs, dtype=torch.bool, device=self.device))
