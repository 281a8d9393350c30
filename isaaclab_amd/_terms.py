"""Lowering of manager terms to ``mdp_term_t`` records.

A stock term function (``isaaclab_amd.envs.mdp.*``) does two things:

* called as ``func(env, **params)`` (the reference contract, ``LAB/managers/manager_base.py:250-267``)
  it evaluates ONE lowered term on the GPU through ``mdp_term_eval``;
* it carries ``func.lower(env, **params) -> TermSpec`` so the managers can fuse all recognised
  terms of a group into one kernel launch (``mdp_program_create`` + ``mdp_*_compute``).

User-defined terms without a ``lower`` attribute are called as plain Python and enter the fused
kernel through the ``*_EXTERNAL`` / ``MDP_OBS_COPY`` opcodes (their output tensor is read by pointer).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Any

import torch

from . import _lib as L

OBS, REWARD, TERMINATION = L.PROG_OBS, L.PROG_REWARD, L.PROG_TERMINATION


def ids_to_list(ids, count: int) -> list[int] | None:
    """SceneEntityCfg ids (slice(None) | slice | list | tensor | int) -> explicit list or None for 'all'."""
    if ids is None:
        return None
    if isinstance(ids, slice):
        if ids == slice(None):
            return None
        return list(range(*ids.indices(count)))
    if isinstance(ids, torch.Tensor):
        return [int(v) for v in ids.tolist()]
    if isinstance(ids, int):
        return [ids]
    return [int(v) for v in ids]


@dataclass
class TermSpec:
    kind: int
    op: str
    dim: int = 1
    p: dict[int, torch.Tensor | None] = field(default_factory=dict)   # slot -> tensor
    i: dict[int, int] = field(default_factory=dict)
    f: dict[int, float] = field(default_factory=dict)
    l0: int = 0
    idx: list[int] | None = None
    idx2: list[int] | None = None
    raycaster: Any = None           # sensor object for MDP_OBS_HEIGHT_SCAN_CAST
    before: list = field(default_factory=list)  # callables run before each launch (e.g. sensor lazy update)
    lazy_sensors: list = field(default_factory=list)  # lazily evaluated sensors the `before` hooks refresh
    out_dtype: torch.dtype = torch.float32

    def to_struct(self, pool: list[int]) -> L.MdpTerm:
        t = L.MdpTerm()
        t.op = L.OPS[self.op]
        t.dim = int(self.dim)
        t.l0 = int(self.l0)
        for k, v in self.i.items():
            setattr(t, f"i{k}", int(v))
        for k, v in self.f.items():
            setattr(t, f"f{k}", float(v))
        for k, v in self.p.items():
            if v is not None:
                L.require_gpu(v, f"{self.op}.p{k}")
                setattr(t, f"p{k}", v.data_ptr())
        for name, lst in (("", self.idx), ("2", self.idx2)):
            if lst is None:
                setattr(t, f"n_idx{name}", -1)
                setattr(t, f"idx{name}_offset", 0)
            else:
                setattr(t, f"n_idx{name}", len(lst))
                setattr(t, f"idx{name}_offset", len(pool))
                pool.extend(int(v) for v in lst)
        return t


def make_term(op: str, dim: int = 1, **kw) -> L.MdpTerm:
    """Quick struct builder for internal single-kernel helpers (no index lists)."""
    t = L.MdpTerm()
    t.op = L.OPS[op]
    t.dim = dim
    t.n_idx = -1
    t.n_idx2 = -1
    t._keep = []  # keep tensors alive for the duration of the python object
    for k, v in kw.items():
        if k.startswith("p") and isinstance(v, torch.Tensor):
            L.require_gpu(v, k)
            setattr(t, k, v.data_ptr())
            t._keep.append(v)
        else:
            setattr(t, k, v)
    return t


_pool_cache: dict[tuple, torch.Tensor] = {}


def device_pool(pool: list[int], device) -> torch.Tensor | None:
    if not pool:
        return None
    key = (tuple(pool), str(device))
    t = _pool_cache.get(key)
    if t is None:
        t = torch.tensor(pool, dtype=torch.int32, device=device)
        _pool_cache[key] = t
    return t


_epoch = [0]


def run_obs_single(term: L.MdpTerm, num_envs: int, device, pool: torch.Tensor | None = None) -> torch.Tensor:
    out = torch.empty((num_envs, term.dim), dtype=torch.float32, device=device)
    _epoch[0] += 1
    L.check(L.lib.mdp_term_eval(OBS, C.byref(term), L.ptr(pool) if pool is not None else None, num_envs,
                                out.data_ptr(), term.dim, torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, _epoch[0],
                                L.stream(device)))
    return out


def evaluate(spec: TermSpec, num_envs: int, device) -> torch.Tensor:
    """Stand-alone evaluation of one lowered term."""
    for fn in spec.before:
        fn()
    pool: list[int] = []
    if spec.op == "MDP_OBS_HEIGHT_SCAN_CAST":
        raise RuntimeError("fused height-scan terms are only evaluated inside an ObservationManager program")
    term = spec.to_struct(pool)
    dpool = device_pool(pool, device)
    if spec.kind == OBS:
        return run_obs_single(term, num_envs, device, dpool)
    if spec.kind == REWARD:
        out = torch.empty((num_envs,), dtype=torch.float32, device=device)
    else:
        out = torch.empty((num_envs,), dtype=torch.bool, device=device)
    L.check(L.lib.mdp_term_eval(spec.kind, C.byref(term), L.ptr(dpool) if dpool is not None else None, num_envs,
                                out.data_ptr(), 1, 0, 0, L.stream(device)))
    return out


def lowered(kind: int):
    """Decorator: ``@lowered(OBS) def lower_fn(env, **params) -> TermSpec`` becomes a term function
    ``func(env, **params) -> Tensor`` with the lowering attached as ``func.lower``."""

    def deco(lower_fn):
        def term_fn(env, *args, **kwargs):
            spec = lower_fn(env, *args, **kwargs)
            if isinstance(spec, torch.Tensor) or not isinstance(spec, TermSpec):
                return spec  # e.g. python-bool shortcuts of the reference (terrain_out_of_bounds on a plane)
            return evaluate(spec, env.num_envs, env.device)

        term_fn.__name__ = lower_fn.__name__
        term_fn.__qualname__ = lower_fn.__qualname__
        term_fn.__doc__ = lower_fn.__doc__
        term_fn.__module__ = lower_fn.__module__
        term_fn.__wrapped__ = lower_fn  # inspect.signature follows __wrapped__ -> param checks see the real signature
        term_fn.lower = lower_fn
        term_fn.mdp_kind = kind
        return term_fn

    return deco


def rows2d(t: torch.Tensor, name: str) -> tuple[torch.Tensor, int]:
    """fp32 2-D view with dense last dim -> (tensor, row stride)."""
    return L.rows(t, name)
