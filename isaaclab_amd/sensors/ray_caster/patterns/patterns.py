"""Ray patterns (``LAB/sensors/ray_caster/patterns/patterns.py:16-58``): init-time host geometry."""
from __future__ import annotations

import torch


def grid_pattern(cfg, device: str) -> tuple[torch.Tensor, torch.Tensor]:
    if cfg.ordering not in ["xy", "yx"]:
        raise ValueError(f"Ordering must be 'xy' or 'yx'. Received: '{cfg.ordering}'.")
    if cfg.resolution <= 0:
        raise ValueError(f"Resolution must be greater than 0. Received: '{cfg.resolution}'.")
    indexing = cfg.ordering if cfg.ordering == "xy" else "ij"
    # built on the host in fp32 exactly like torch.arange on the reference device, then uploaded
    x = torch.arange(start=-cfg.size[0] / 2, end=cfg.size[0] / 2 + 1.0e-9, step=cfg.resolution)
    y = torch.arange(start=-cfg.size[1] / 2, end=cfg.size[1] / 2 + 1.0e-9, step=cfg.resolution)
    grid_x, grid_y = torch.meshgrid(x, y, indexing=indexing)
    num_rays = grid_x.numel()
    ray_starts = torch.zeros(num_rays, 3)
    ray_starts[:, 0] = grid_x.flatten()
    ray_starts[:, 1] = grid_y.flatten()
    ray_directions = torch.zeros_like(ray_starts)
    ray_directions[..., :] = torch.tensor(list(cfg.direction))
    return ray_starts.to(device), ray_directions.to(device)
