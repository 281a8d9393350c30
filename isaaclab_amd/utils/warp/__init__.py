"""Replacement of ``isaaclab.utils.warp`` (``LAB/utils/warp/ops.py``): same call signatures, HIP kernels."""
from .ops import Mesh, convert_to_warp_mesh, raycast_mesh

__all__ = ["Mesh", "convert_to_warp_mesh", "raycast_mesh"]
