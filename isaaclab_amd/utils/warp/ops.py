"""``raycast_mesh`` / ``convert_to_warp_mesh`` (``LAB/utils/warp/ops.py:24-145``) on gfx950.

``Mesh`` stands where ``wp.Mesh`` stood: an opaque handle owning the device-side triangle soup and
its acceleration structures (uniform xy cell grid + LBVH, built on the host at creation).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _lib as L


class Mesh:
    """Device mesh handle (counterpart of ``wp.Mesh``)."""

    def __init__(self, points: np.ndarray, indices: np.ndarray, device: str = "cuda:0", accel: str = "auto"):
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.float32).reshape(-1, 3))
        idx = np.ascontiguousarray(np.asarray(indices, dtype=np.int32).reshape(-1, 3))
        self.device = torch.device(device)
        self._handle = C.c_void_p()
        accel_id = {"auto": 0, "grid": 1, "bvh": 2}[accel]
        L.check(L.lib.mdp_mesh_create(pts.ctypes.data, pts.shape[0], idx.ctypes.data, idx.shape[0],
                                      L.device_index(self.device), accel_id, C.byref(self._handle)))
        self.num_points = pts.shape[0]
        self.num_tris = idx.shape[0]

    @property
    def handle(self) -> int:
        return self._handle.value

    @property
    def id(self) -> int:  # name used by warp call sites
        return self._handle.value

    def info(self) -> dict:
        vals = [C.c_int64() for _ in range(5)]
        accel = C.c_int()
        L.check(L.lib.mdp_mesh_info(self._handle, *[C.byref(v) for v in vals], C.byref(accel)))
        keys = ["num_tris", "num_cells", "num_cell_refs", "num_bvh_nodes", "device_bytes"]
        out = {k: v.value for k, v in zip(keys, vals)}
        out["accel"] = {1: "grid", 2: "bvh"}.get(accel.value, "brute")
        return out

    def __del__(self):
        try:
            if self._handle:
                L.lib.mdp_mesh_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass


_ACCEL = {None: 0, "auto": 0, "grid": 1, "bvh": 2, "brute": 3}


def raycast_mesh(
    ray_starts: torch.Tensor,
    ray_directions: torch.Tensor,
    mesh: Mesh,
    max_dist: float = 1e6,
    return_distance: bool = False,
    return_normal: bool = False,
    return_face_id: bool = False,
    accel: str | None = None,
):
    """Closest hit per ray; misses give ``inf`` hits/distances/normals and ``-1`` face ids (ops.py:46-58)."""
    shape = ray_starts.shape
    device = ray_starts.device
    starts = L.f32(ray_starts.to(mesh.device), "ray_starts").reshape(-1, 3).contiguous()
    dirs = L.f32(ray_directions.to(mesh.device), "ray_directions").reshape(-1, 3).contiguous()
    if starts.shape != dirs.shape:
        raise ValueError(f"ray_starts {tuple(ray_starts.shape)} and ray_directions {tuple(ray_directions.shape)} differ")
    n = starts.shape[0]
    hits = torch.empty((n, 3), dtype=torch.float32, device=mesh.device)
    dist = torch.empty((n,), dtype=torch.float32, device=mesh.device) if return_distance else None
    normal = torch.empty((n, 3), dtype=torch.float32, device=mesh.device) if return_normal else None
    face = torch.empty((n,), dtype=torch.int32, device=mesh.device) if return_face_id else None
    L.check(L.lib.mdp_raycast(mesh.handle, n, L.ptr(starts), L.ptr(dirs), float(max_dist), L.ptr(hits),
                              L.ptr(dist) if dist is not None else None, L.ptr(normal) if normal is not None else None,
                              L.ptr(face) if face is not None else None, _ACCEL[accel], L.stream(mesh.device)))
    if return_distance:
        dist = dist.to(device).view(shape[:-1])
    if return_normal:
        normal = normal.to(device).view(shape)
    if return_face_id:
        face = face.to(device).view(shape[:-1])
    return hits.to(device).view(shape), dist, normal, face


def convert_to_warp_mesh(points: np.ndarray, indices: np.ndarray, device: str, accel: str = "auto") -> Mesh:
    """ops.py:130-145"""
    return Mesh(points, indices, device=device, accel=accel)
