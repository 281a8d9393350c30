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


def pinhole_camera_pattern(cfg, intrinsic_matrices: torch.Tensor, device: str) -> tuple[torch.Tensor, torch.Tensor]:
    """``patterns.py:61-104``: one ray per pixel centre through the inverse intrinsic matrix, expressed in the robotics
    camera frame (x forward, y left, z up) and normalised; starts at the sensor origin.  Init-time host geometry in fp32
    (like ``grid_pattern``), uploaded once; shapes (N, H * W, 3)."""
    k = intrinsic_matrices.detach().to("cpu", torch.float32)
    grid = torch.meshgrid(torch.arange(start=0, end=cfg.width, dtype=torch.int32),
                          torch.arange(start=0, end=cfg.height, dtype=torch.int32), indexing="xy")
    pixels = torch.vstack(list(map(torch.ravel, grid))).T
    pixels = torch.hstack([pixels, torch.ones((len(pixels), 1))])
    pixels += torch.tensor([[0.5, 0.5, 0]])
    pix_in_cam_frame = torch.matmul(torch.inverse(k), pixels.T)
    transform_vec = torch.tensor([1, -1, -1]).unsqueeze(0).unsqueeze(2)
    pix_in_cam_frame = pix_in_cam_frame[:, [2, 0, 1], :] * transform_vec
    ray_directions = (pix_in_cam_frame / torch.norm(pix_in_cam_frame, dim=1, keepdim=True)).permute(0, 2, 1)
    ray_starts = torch.zeros_like(ray_directions)
    return ray_starts.contiguous().to(device), ray_directions.contiguous().to(device)
