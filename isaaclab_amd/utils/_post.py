"""Shared helper: run the noise -> clip -> scale post-processing kernel in place on an (N, d) tensor."""
from __future__ import annotations

import ctypes as C
import itertools

import torch

from .. import _lib as L

_call_counter = itertools.count(1)

NOISE_NONE, NOISE_UNIFORM, NOISE_GAUSSIAN, NOISE_CONSTANT = 0, 1, 2, 3
_OPS = {"add": 0, "scale": 1, "abs": 2}


def noise_op_id(operation: str) -> int:
    if operation not in _OPS:
        raise ValueError(f"Unknown operation in noise: {operation}")
    return _OPS[operation]


def _scalar(v, name: str) -> float:
    if isinstance(v, torch.Tensor):
        if v.numel() != 1:
            raise NotImplementedError(f"tensor-valued noise parameter '{name}' with more than one element")
        return float(v.item())
    return float(v)


def fill_noise(term: L.MdpTerm, noise_cfg) -> None:
    """Encode a NoiseCfg (``LAB/utils/noise/noise_cfg.py:18-92``) into the post-processing fields."""
    kind = type(noise_cfg).__name__
    if "Uniform" in kind:
        lo, hi = _scalar(noise_cfg.n_min, "n_min"), _scalar(noise_cfg.n_max, "n_max")
        term.noise_kind, term.n0, term.n1 = NOISE_UNIFORM, lo, hi - lo  # range evaluated in double like python
    elif "Gaussian" in kind:
        term.noise_kind, term.n0, term.n1 = NOISE_GAUSSIAN, _scalar(noise_cfg.mean, "mean"), _scalar(noise_cfg.std, "std")
    elif "Constant" in kind:
        term.noise_kind, term.n0, term.n1 = NOISE_CONSTANT, _scalar(noise_cfg.bias, "bias"), 0.0
    else:
        raise NotImplementedError(f"noise cfg {kind} has no HIP lowering")
    term.noise_op = noise_op_id(noise_cfg.operation)


def postprocess_inplace(data: torch.Tensor, term: L.MdpTerm, seed: int | None = None) -> torch.Tensor:
    L.require_gpu(data, "data")
    if data.dtype != torch.float32 or not data.is_contiguous():
        raise ValueError("post-processing expects a contiguous fp32 tensor")
    n = data.shape[0] if data.dim() > 0 else 1
    d = data.numel() // max(n, 1)
    term.dim = max(d, 1)
    if seed is None:
        seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
    L.check(L.lib.mdp_obs_postprocess(C.byref(term), n, data.data_ptr(), term.dim, seed, next(_call_counter), L.stream()))
    return data
