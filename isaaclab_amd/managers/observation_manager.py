"""ObservationManager (``LAB/managers/observation_manager.py:65-494``).

Same cfg schema, properties, error behaviour and per-term pipeline (compute -> modifiers -> noise ->
clip -> scale -> history -> concatenate).  A group whose terms all lower to ``mdp_term_t`` records is
evaluated by ONE fused kernel (``mdp_obs_compute``) that writes the concatenated (N, O) tensor
directly; other groups run term by term (each stock term is still a HIP kernel).
"""
from __future__ import annotations

import inspect
from collections.abc import Sequence

import numpy as np
import torch

from .. import _lib as L
from .._terms import OBS, TermSpec
from ..utils import modifiers
from ..utils._post import fill_noise, postprocess_inplace
from ..utils.buffers import CircularBuffer
from ._program import Program
from .manager_base import ManagerBase, ManagerTermBase
from .manager_term_cfg import ObservationGroupCfg, ObservationTermCfg

_GROUP_SETTINGS = ["enable_corruption", "concatenate_terms", "history_length", "flatten_history_dim"]


class ObservationManager(ManagerBase):
    def __init__(self, cfg: object, env, seed: int | None = None):
        if cfg is None:
            raise ValueError("Observation manager configuration is None. Please provide a valid configuration.")
        self._seed = seed
        self._programs: dict[str, Program | None] = {}
        self._scans: dict[str, list] = {}
        self._skip_sensor_terms = False
        super().__init__(cfg, env)
        self._group_obs_dim: dict[str, tuple[int, ...] | list[tuple[int, ...]]] = dict()
        for group_name, group_term_dims in self._group_obs_term_dim.items():
            if self._group_obs_concatenate[group_name]:
                dims = [tuple(d) for d in group_term_dims]
                lead = {d[:-1] for d in dims}
                if any(len(d) == 0 for d in dims) or len(lead) > 1:
                    raise RuntimeError(
                        f"Unable to concatenate observation terms in group '{group_name}'."
                        f" The shapes of the terms are: {group_term_dims}."
                        " Please ensure that the shapes are compatible for concatenation."
                        " Otherwise, set 'concatenate_terms' to False in the group configuration."
                    )
                # the reference sums the per-term shapes element-wise (observation_manager.py:91-92)
                self._group_obs_dim[group_name] = tuple(int(sum(d[i] for d in dims)) for i in range(len(dims[0])))
            else:
                self._group_obs_dim[group_name] = group_term_dims
        self._obs_buffer = None

    def __str__(self) -> str:
        msg = f"<ObservationManager> contains {len(self._group_obs_term_names)} groups.\n"
        for group_name, group_dim in self._group_obs_dim.items():
            msg += f"Active Observation Terms in Group: '{group_name}'"
            if self._group_obs_concatenate[group_name]:
                msg += f" (shape: {group_dim})"
            msg += "\n"
            for index, (name, dims) in enumerate(
                zip(self._group_obs_term_names[group_name], self._group_obs_term_dim[group_name])
            ):
                msg += f"  {index:3d} | {name:<32s} | {tuple(dims)}\n"
        return msg

    def get_active_iterable_terms(self, env_idx: int):
        terms = []
        if self._obs_buffer is None:
            self.compute()
        obs_buffer = self._obs_buffer
        for group_name in self._group_obs_dim:
            if not self.group_obs_concatenate[group_name]:
                for name, term in obs_buffer[group_name].items():
                    terms.append((group_name + "-" + name, term[env_idx].cpu().tolist()))
                continue
            idx = 0
            data = obs_buffer[group_name]
            for name, shape in zip(self._group_obs_term_names[group_name], self._group_obs_term_dim[group_name]):
                data_length = int(np.prod(shape))
                terms.append((group_name + "-" + name, data[env_idx, idx : idx + data_length].cpu().tolist()))
                idx += data_length
        return terms

    # -- properties --------------------------------------------------------------------------------
    @property
    def active_terms(self) -> dict[str, list[str]]:
        return self._group_obs_term_names

    @property
    def group_obs_dim(self):
        return self._group_obs_dim

    @property
    def group_obs_term_dim(self):
        return self._group_obs_term_dim

    @property
    def group_obs_concatenate(self) -> dict[str, bool]:
        return self._group_obs_concatenate

    # -- operations --------------------------------------------------------------------------------
    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, float]:
        for group_name, group_cfg in self._group_obs_class_term_cfgs.items():
            for term_cfg in group_cfg:
                term_cfg.func.reset(env_ids=env_ids)
            for term_name in self._group_obs_term_names[group_name]:
                if term_name in self._group_obs_term_history_buffer[group_name]:
                    self._group_obs_term_history_buffer[group_name][term_name].reset(batch_ids=env_ids)
        for mod in self._group_obs_class_modifiers:
            mod.reset(env_ids=env_ids)
        return {}

    def compute(self, skip_sensor_terms: bool = False):
        """observation_manager.py:238-258.  ``skip_sensor_terms``: the caller has launched ``compute_sensor_terms()``
        itself (on another stream)."""
        obs_buffer = dict()
        self._skip_sensor_terms = skip_sensor_terms
        try:
            for group_name in self._group_obs_term_names:
                obs_buffer[group_name] = self.compute_group(group_name)
        finally:
            self._skip_sensor_terms = False
        self._obs_buffer = obs_buffer
        return obs_buffer

    def compute_group(self, group_name: str):
        if group_name not in self._group_obs_term_names:
            raise ValueError(
                f"Unable to find the group '{group_name}' in the observation manager."
                f" Available groups are: {list(self._group_obs_term_names.keys())}"
            )
        if self._group_fusable[group_name]:
            return self._compute_group_fused(group_name)
        return self._compute_group_unfused(group_name)

    # -- fused path --------------------------------------------------------------------------------
    def _build_program(self, group_name: str) -> Program:
        specs: list[TermSpec] = []
        cfgs = self._group_obs_term_cfgs[group_name]
        offset = 0
        self._external[group_name] = []
        offsets = []
        scans: list = []
        for term_cfg, dims in zip(cfgs, self._group_obs_term_dim[group_name]):
            dim = int(np.prod(dims))
            lower = getattr(term_cfg.func, "lower", None)
            if lower is not None:
                spec = lower(self._env, **term_cfg.params)
                if spec.op == "MDP_OBS_HEIGHT_SCAN" and spec.raycaster is not None and spec.raycaster.can_fuse:
                    spec.op = "MDP_OBS_HEIGHT_SCAN_CAST"
                    spec.before = []
                elif (spec.op == "MDP_OBS_HEIGHT_SCAN" and spec.raycaster is not None
                      and getattr(spec.raycaster, "scan_in_cast", False) and dim == spec.raycaster.num_rays):
                    # the cast kernel writes these columns (mdp_height_scan_obs_t); the program skips them
                    scan = L.MdpHeightScanObs()
                    scan.col_offset, scan.offset = offset, float(spec.f.get(0, 0.5))
                    scans.append((spec.raycaster, scan, len(specs)))
                    spec.op = "MDP_OBS_SKIP"
                    spec.before = []
                    spec.p = {}
            else:
                staging = torch.zeros((self.num_envs, dim), dtype=torch.float32, device=self.device)
                self._external[group_name].append((term_cfg, staging))
                spec = TermSpec(OBS, "MDP_OBS_COPY", dim=dim, p={0: staging}, i={0: dim, 1: 0})
            spec.dim = dim
            offsets.append(offset)
            specs.append(spec)
            offset += dim

        def post(n, spec, t):
            cfg = cfgs[n]
            t.out_offset = offsets[n]
            if cfg.noise:
                fill_noise(t, cfg.noise)
                injected = getattr(cfg, "_injected_noise", None)
                if injected is not None:
                    if injected.dtype != torch.float32 or not injected.is_contiguous():
                        raise ValueError("injected observation noise must be a contiguous float32 tensor (the term table"
                                         " keeps its address)")
                    t.noise_u = L.require_gpu(injected, "injected noise").data_ptr()
                    self._keepalive.append(injected)
            if cfg.clip:
                t.has_clip, t.clip_lo, t.clip_hi = 1, float(cfg.clip[0]), float(cfg.clip[1])
            if cfg.scale is not None:
                if isinstance(cfg.scale, torch.Tensor) and cfg.scale.numel() > 1:
                    t.has_scale, t.scale_vec = 2, cfg.scale.data_ptr()
                else:
                    t.has_scale, t.scale = 1, float(cfg.scale)

            for _, scan, term_index in scans:
                if term_index == n:
                    # the cast kernel applies this term's noise / clip / scale
                    for name in ("dim", "noise_kind", "noise_op", "n0", "n1", "noise_u", "has_clip", "clip_lo", "clip_hi",
                                 "has_scale", "scale", "scale_vec"):
                        setattr(scan.post, name, getattr(t, name))

        seed = self._seed if self._seed is not None else torch.initial_seed()
        prog = Program(OBS, self.num_envs, self.device, specs, structs_post=post, seed=seed)
        for k, (_, scan, _) in enumerate(scans):
            rng = torch.zeros(1 + self.num_envs, dtype=torch.int64, device=self.device)
            rng[0] = (seed + 7919 * (k + 1)) & 0x7FFFFFFFFFFFFFFF
            scan.rng = rng.data_ptr()
            self._keepalive.append(rng)
        self._scans[group_name] = [(sensor, scan) for sensor, scan, _ in scans]
        return prog

    def _compute_group_fused(self, group_name: str) -> torch.Tensor:
        prog = self._programs.get(group_name)
        if prog is None:
            prog = self._build_program(group_name)
            self._programs[group_name] = prog
        for term_cfg, staging in self._external[group_name]:
            value = term_cfg.func(self._env, **term_cfg.params)
            staging.copy_(value.reshape(self.num_envs, -1))
        prog.run_before()
        width = self._group_obs_dim[group_name][0]
        out = self._out_buffers.get(group_name) if self.persistent_output else None
        if out is None:
            out = torch.empty((self.num_envs, width), dtype=torch.float32, device=self.device)
            if self.persistent_output:
                self._out_buffers[group_name] = out
        if not self._skip_sensor_terms:
            self._launch_scans(group_name, out, width)
        L.check(L.lib.mdp_obs_compute(prog.handle, out.data_ptr(), width, L.stream(self.device)))
        return out

    def _launch_scans(self, group_name: str, out: torch.Tensor, width: int, tick_dt: float = 0.0,
                      from_physics: bool = False, ticked: set | None = None):
        for sensor, scan in self._scans.get(group_name, []):
            scan.obs, scan.ld_obs = out.data_ptr(), width
            dt = tick_dt
            if ticked is not None:          # a sensor shared by several groups is ticked by its first launch only
                if id(sensor) in ticked:
                    dt = 0.0
                ticked.add(id(sensor))
            sensor._update_outdated_buffers(scan=scan, tick_dt=dt, from_physics=from_physics)

    def scan_sensors(self) -> list:
        """Sensors whose observation columns are written by their own cast kernel."""
        out = []
        for scans in self._scans.values():
            for sensor, _ in scans:
                if not any(sensor is s for s in out):
                    out.append(sensor)
        return out

    def post_physics_args(self):
        """``(MdpObsArgs, sensor, scan)`` for ``mdp_post_physics_step``: one fused group with a persistent output whose
        only sensor-owned columns are one height scan written by its cast kernel; ``None`` when the observations do not
        have that shape."""
        if len(self._group_obs_term_names) != 1 or not self.can_split_sensor_terms():
            return None
        group_name = next(iter(self._group_obs_term_names))
        scans = self._scans.get(group_name, [])
        if len(scans) > 1 or self._external[group_name]:
            return None
        prog, out = self._programs[group_name], self._out_buffers[group_name]
        sensor, scan = scans[0] if scans else (None, None)      # a task without a height scan: no cast blocks
        if scan is not None:
            scan.obs, scan.ld_obs = out.data_ptr(), out.shape[1]
        a = L.MdpObsArgs()
        a.program, a.obs, a.ld_obs = prog.handle, out.data_ptr(), out.shape[1]
        return a, sensor, scan, prog, group_name, out

    def can_split_sensor_terms(self) -> bool:
        """Whether ``compute_sensor_terms()`` + ``compute(skip_sensor_terms=True)`` can replace ``compute()``: every
        group is fused, its program is built and its output buffer is persistent."""
        if not self.persistent_output:
            return False
        for group_name in self._group_obs_term_names:
            if not self._group_fusable[group_name]:
                return False
            if self._programs.get(group_name) is None or group_name not in self._out_buffers:
                return False
        return True

    def compute_sensor_terms(self, tick_dt: float = 0.0, from_physics: bool = False):
        """Launches only the sensor-owned observation columns (height scans written by their cast kernels) of every
        group into the persistent output buffers; ``compute(skip_sensor_terms=True)`` evaluates the rest.  Lets a caller
        put the two on different streams (``can_split_sensor_terms()`` must hold).  ``tick_dt`` / ``from_physics``:
        see ``RayCaster._update_outdated_buffers`` (the caller then leaves these sensors out of ``scene.update``)."""
        ticked: set = set()
        for group_name in self._group_obs_term_names:
            out = self._out_buffers[group_name]
            self._launch_scans(group_name, out, out.shape[1], tick_dt, from_physics, ticked)

    # -- term-by-term path -------------------------------------------------------------------------
    def _compute_group_unfused(self, group_name: str):
        group_term_names = self._group_obs_term_names[group_name]
        group_obs = dict.fromkeys(group_term_names, None)
        for term_name, term_cfg in zip(group_term_names, self._group_obs_term_cfgs[group_name]):
            obs = term_cfg.func(self._env, **term_cfg.params)
            obs = obs.float().clone().contiguous() if obs.dtype != torch.float32 else obs.clone().contiguous()
            if term_cfg.modifiers is not None:
                for modifier in term_cfg.modifiers:
                    obs = modifier.func(obs, **modifier.params)
            if term_cfg.noise:
                obs = term_cfg.noise.func(obs, term_cfg.noise)
            if term_cfg.clip or term_cfg.scale is not None:
                t = L.MdpTerm()
                if term_cfg.clip:
                    t.has_clip, t.clip_lo, t.clip_hi = 1, float(term_cfg.clip[0]), float(term_cfg.clip[1])
                if term_cfg.scale is not None:
                    if isinstance(term_cfg.scale, torch.Tensor) and term_cfg.scale.numel() > 1:
                        t.has_scale, t.scale_vec = 2, term_cfg.scale.data_ptr()
                    else:
                        t.has_scale, t.scale = 1, float(term_cfg.scale)
                obs = postprocess_inplace(obs.contiguous(), t)
            if term_cfg.history_length > 0:
                hist = self._group_obs_term_history_buffer[group_name][term_name]
                hist.append(obs)
                buf = hist.buffer
                group_obs[term_name] = buf.reshape(self._env.num_envs, -1) if term_cfg.flatten_history_dim else buf
            else:
                group_obs[term_name] = obs
        if self._group_obs_concatenate[group_name]:
            return torch.cat(list(group_obs.values()), dim=-1)
        return group_obs

    # -- cfg parsing -------------------------------------------------------------------------------
    def _prepare_terms(self):
        self._group_obs_term_names: dict[str, list[str]] = dict()
        self._group_obs_term_dim: dict[str, list[tuple[int, ...]]] = dict()
        self._group_obs_term_cfgs: dict[str, list[ObservationTermCfg]] = dict()
        self._group_obs_class_term_cfgs: dict[str, list[ObservationTermCfg]] = dict()
        self._group_obs_concatenate: dict[str, bool] = dict()
        self._group_obs_term_history_buffer: dict[str, dict] = dict()
        self._group_obs_class_modifiers: list[modifiers.ModifierBase] = list()
        self._group_fusable: dict[str, bool] = dict()
        self._external: dict[str, list] = dict()
        self._out_buffers: dict[str, torch.Tensor] = dict()
        self._keepalive: list = []
        self.persistent_output = False

        for group_name, group_cfg in self._cfg_items():
            if group_cfg is None:
                continue
            if not isinstance(group_cfg, ObservationGroupCfg):
                raise TypeError(
                    f"Observation group '{group_name}' is not of type 'ObservationGroupCfg'."
                    f" Received: '{type(group_cfg)}'."
                )
            self._group_obs_term_names[group_name] = list()
            self._group_obs_term_dim[group_name] = list()
            self._group_obs_term_cfgs[group_name] = list()
            self._group_obs_class_term_cfgs[group_name] = list()
            history_buffers: dict[str, CircularBuffer] = dict()
            self._group_obs_concatenate[group_name] = group_cfg.concatenate_terms
            fusable = bool(group_cfg.concatenate_terms)
            for term_name, term_cfg in self._cfg_items(group_cfg):
                if term_name in _GROUP_SETTINGS:
                    continue
                if term_cfg is None:
                    continue
                if not isinstance(term_cfg, ObservationTermCfg):
                    raise TypeError(
                        f"Configuration for the term '{term_name}' is not of type ObservationTermCfg."
                        f" Received: '{type(term_cfg)}'."
                    )
                self._resolve_common_term_cfg(f"{group_name}/{term_name}", term_cfg, min_argc=1)
                if not group_cfg.enable_corruption:
                    term_cfg.noise = None
                if group_cfg.history_length is not None:
                    term_cfg.history_length = group_cfg.history_length
                    term_cfg.flatten_history_dim = group_cfg.flatten_history_dim
                self._group_obs_term_names[group_name].append(term_name)
                self._group_obs_term_cfgs[group_name].append(term_cfg)
                # shape probe: one call of the term at construction (observation_manager.py:411)
                obs_dims = tuple(term_cfg.func(self._env, **term_cfg.params).shape)
                if term_cfg.history_length > 0:
                    history_buffers[term_name] = CircularBuffer(
                        max_len=term_cfg.history_length, batch_size=self._env.num_envs, device=self._env.device
                    )
                    old_dims = list(obs_dims)
                    old_dims.insert(1, term_cfg.history_length)
                    obs_dims = tuple(old_dims)
                    if term_cfg.flatten_history_dim:
                        obs_dims = (obs_dims[0], int(np.prod(obs_dims[1:])))
                self._group_obs_term_dim[group_name].append(obs_dims[1:])

                if term_cfg.scale is not None:
                    if not isinstance(term_cfg.scale, (float, int, tuple)):
                        raise TypeError(
                            f"Scale for observation term '{term_name}' in group '{group_name}'"
                            f" is not of type float, int or tuple. Received: '{type(term_cfg.scale)}'."
                        )
                    if isinstance(term_cfg.scale, tuple) and len(term_cfg.scale) != obs_dims[1]:
                        raise ValueError(
                            f"Scale for observation term '{term_name}' in group '{group_name}'"
                            f" does not match the dimensions of the observation. Expected: {obs_dims[1]}"
                            f" but received: {len(term_cfg.scale)}."
                        )
                    term_cfg.scale = torch.tensor(term_cfg.scale, dtype=torch.float, device=self._env.device)

                if term_cfg.modifiers is not None:
                    for mod_cfg in term_cfg.modifiers:
                        if isinstance(mod_cfg, modifiers.ModifierCfg):
                            if inspect.isclass(mod_cfg.func):
                                if not issubclass(mod_cfg.func, modifiers.ModifierBase):
                                    raise TypeError(
                                        f"Modifier function '{mod_cfg.func}' for observation term '{term_name}'"
                                        f" is not a subclass of 'ModifierBase'. Received: '{type(mod_cfg.func)}'."
                                    )
                                mod_cfg.func = mod_cfg.func(cfg=mod_cfg, data_dim=obs_dims, device=self._env.device)
                                self._group_obs_class_modifiers.append(mod_cfg.func)
                        else:
                            raise TypeError(
                                f"Modifier configuration '{mod_cfg}' of observation term '{term_name}' is not of"
                                f" required type ModifierCfg, Received: '{type(mod_cfg)}'"
                            )
                        if not callable(mod_cfg.func):
                            raise AttributeError(
                                f"Modifier '{mod_cfg}' of observation term '{term_name}' is not callable."
                                f" Received: {mod_cfg.func}"
                            )
                        term_params = list(mod_cfg.params.keys())
                        args = inspect.signature(mod_cfg.func).parameters
                        with_defaults = [a for a in args if args[a].default is not inspect.Parameter.empty]
                        without_defaults = [a for a in args if args[a].default is inspect.Parameter.empty]
                        args = without_defaults + with_defaults
                        if len(args) > 1:
                            if set(args[1:]) != set(term_params + with_defaults):
                                raise ValueError(
                                    f"Modifier '{mod_cfg}' of observation term '{term_name}' expects"
                                    f" mandatory parameters: {without_defaults[1:]}"
                                    f" and optional parameters: {with_defaults}, but received: {term_params}."
                                )
                if isinstance(term_cfg.func, ManagerTermBase):
                    self._group_obs_class_term_cfgs[group_name].append(term_cfg)
                    term_cfg.func.reset()
                # fusability: single row per env, no history / modifiers, scalar noise parameters
                if term_cfg.modifiers or term_cfg.history_length > 0 or len(obs_dims) != 2:
                    fusable = False
                if term_cfg.noise is not None:
                    try:
                        fill_noise(L.MdpTerm(), term_cfg.noise)
                    except NotImplementedError:
                        fusable = False
            self._group_obs_term_history_buffer[group_name] = history_buffers
            self._group_fusable[group_name] = fusable and len(self._group_obs_term_names[group_name]) > 0
