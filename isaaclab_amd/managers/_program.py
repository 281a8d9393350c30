"""Host-side owner of a device term program (``mdp_program_t``)."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib as L
from .._terms import TermSpec


class Program:
    """Builds and owns one ``mdp_program_t``; keeps every bound tensor alive."""

    def __init__(self, kind: int, num_envs: int, device, specs: list[TermSpec], structs_post=None, seed: int = 0):
        self.kind = kind
        self.num_envs = num_envs
        self.device = torch.device(device)
        self.specs = specs
        pool: list[int] = []
        terms = (L.MdpTerm * max(len(specs), 1))()
        self._keep = []
        self._descs = []
        for n, spec in enumerate(specs):
            t = spec.to_struct(pool)
            if structs_post is not None:
                structs_post(n, spec, t)
            if spec.op == "MDP_OBS_HEIGHT_SCAN_CAST":
                desc = spec.raycaster.binding_desc()
                self._descs.append(desc)
                t.p4 = C.addressof(desc)
            self._keep.extend(v for v in spec.p.values() if v is not None)
            terms[n] = t
        pool_arr = (C.c_int32 * max(len(pool), 1))(*pool)
        self._handle = C.c_void_p()
        L.check(L.lib.mdp_program_create(kind, L.device_index(self.device), num_envs, C.byref(terms), len(specs),
                                         C.byref(pool_arr), len(pool), seed & 0xFFFFFFFFFFFFFFFF,
                                         C.byref(self._handle)))
        self.before = [fn for spec in specs for fn in spec.before]

    @property
    def handle(self):
        return self._handle

    def run_before(self):
        for fn in self.before:
            fn()

    def __del__(self):
        try:
            if self._handle:
                L.lib.mdp_program_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass
