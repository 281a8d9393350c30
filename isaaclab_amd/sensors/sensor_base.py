"""``SensorBase`` (``LAB/sensors/sensor_base.py:182-293``): per-env timestamps and lazy "outdated" updates.

The bookkeeping (timestamp += dt, outdated flag, last-update stamp) lives on the device and is
maintained by the sensor kernels themselves, so neither ``update`` nor ``.data`` syncs with the host
(the reference calls ``nonzero()`` in ``_update_outdated_buffers``, sensor_base.py:286-287).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from collections.abc import Sequence
from dataclasses import MISSING

import torch

from ..utils.configclass import configclass


@configclass
class SensorBaseCfg:
    class_type: type = MISSING
    prim_path: str = ""
    update_period: float = 0.0
    history_length: int = 0
    debug_vis: bool = False


class SensorBase(ABC):
    def __init__(self, cfg: SensorBaseCfg, num_envs: int, device: str, sim_physics_dt: float):
        if cfg.history_length < 0:
            raise ValueError(f"History length must be greater than 0! Received: {cfg.history_length}")
        self.cfg = cfg
        self._num_envs = num_envs
        self._device = device
        self._sim_physics_dt = sim_physics_dt
        self._is_initialized = True
        self._is_visualizing = False
        self._is_outdated = torch.ones(num_envs, dtype=torch.bool, device=device)
        self._timestamp = torch.zeros(num_envs, device=device)
        self._timestamp_last_update = torch.zeros_like(self._timestamp)

    @property
    def is_initialized(self) -> bool: return self._is_initialized
    @property
    def num_instances(self) -> int: return self._num_envs
    @property
    def device(self) -> str: return self._device

    @property
    @abstractmethod
    def data(self):
        raise NotImplementedError

    def reset(self, env_ids: Sequence[int] | None = None):
        if env_ids is None:
            env_ids = slice(None)
        self._timestamp[env_ids] = 0.0
        self._timestamp_last_update[env_ids] = 0.0
        self._is_outdated[env_ids] = True

    @abstractmethod
    def update(self, dt: float, force_recompute: bool = False):
        raise NotImplementedError

    @abstractmethod
    def _update_outdated_buffers(self):
        raise NotImplementedError
