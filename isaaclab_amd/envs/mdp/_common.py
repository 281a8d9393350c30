"""Helpers shared by the term lowerings."""
from __future__ import annotations

import numpy as np
import torch

from ... import _lib as L
from ..._terms import TermSpec, ids_to_list


def f32(v: float) -> float:
    """Round a python float to fp32 the way torch does when it meets an fp32 tensor."""
    return float(np.float32(v))


def mat(t: torch.Tensor, name: str) -> tuple[torch.Tensor, int]:
    """(N, W) fp32 matrix with dense last dim -> (tensor, row stride)."""
    return L.rows(t, name)


def root_state(asset) -> tuple[torch.Tensor, int, list]:
    """Persistent (N,13) root state + hook that refreshes it lazily before a launch."""
    data = asset.data
    st = data.root_state_w
    return st, st.stride(0), [lambda d=data: d.root_state_w]


def contact_history(sensor) -> tuple[torch.Tensor, int, int, list]:
    d = sensor.data
    hist = d.net_forces_w_history
    return hist, hist.shape[1], hist.shape[2], [lambda s=sensor: s.data]
