"""Rough-terrain mesh generator used as the ray-cast target of the benchmark / tests.

Restates the numpy-level parts of the reference terrain pipeline without trimesh:
``convert_height_field_to_mesh`` (``LAB/terrains/height_field/utils.py:79-173``, vectorised),
``random_uniform_terrain`` and ``pyramid_sloped_terrain`` (``hf_terrains.py:21-148``), laid out on the
10 x 20 grid of 8 m tiles of ``ROUGH_TERRAINS_CFG`` (``LAB/terrains/config/rough.py:12-51``) with the same
sub-terrain proportions (stairs 0.2 / inverted stairs 0.2 / boxes 0.2 / random rough 0.2 / slopes 0.1+0.1).
Stair and box tiles are plain box meshes.  Mesh *generation* is an input generator, not a perf target.
"""
from __future__ import annotations

from dataclasses import MISSING

import numpy as np
from scipy import interpolate

from ..utils.configclass import configclass


def convert_height_field_to_mesh(height_field: np.ndarray, horizontal_scale: float, vertical_scale: float,
                                 slope_threshold: float | None = None) -> tuple[np.ndarray, np.ndarray]:
    num_rows, num_cols = height_field.shape
    y = np.linspace(0, (num_cols - 1) * horizontal_scale, num_cols)
    x = np.linspace(0, (num_rows - 1) * horizontal_scale, num_rows)
    yy, xx = np.meshgrid(y, x)
    hf = height_field.astype(np.float64).copy()
    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale
        move_x = np.zeros((num_rows, num_cols))
        move_y = np.zeros((num_rows, num_cols))
        move_c = np.zeros((num_rows, num_cols))
        move_x[:-1, :] += hf[1:, :] - hf[:-1, :] > thr
        move_x[1:, :] -= hf[:-1, :] - hf[1:, :] > thr
        move_y[:, :-1] += hf[:, 1:] - hf[:, :-1] > thr
        move_y[:, 1:] -= hf[:, :-1] - hf[:, 1:] > thr
        move_c[:-1, :-1] += hf[1:, 1:] - hf[:-1, :-1] > thr
        move_c[1:, 1:] -= hf[:-1, :-1] - hf[1:, 1:] > thr
        xx = xx + (move_x + move_c * (move_x == 0)) * horizontal_scale
        yy = yy + (move_y + move_c * (move_y == 0)) * horizontal_scale
    vertices = np.zeros((num_rows * num_cols, 3), dtype=np.float32)
    vertices[:, 0] = xx.flatten()
    vertices[:, 1] = yy.flatten()
    vertices[:, 2] = hf.flatten() * vertical_scale
    i = np.arange(num_rows - 1)[:, None]
    j = np.arange(num_cols - 1)[None, :]
    ind0 = (i * num_cols + j).reshape(-1)
    ind1, ind2 = ind0 + 1, ind0 + num_cols
    ind3 = ind2 + 1
    tris = np.empty((2 * ind0.size, 3), dtype=np.int32)
    tris[0::2] = np.stack([ind0, ind3, ind1], axis=1)
    tris[1::2] = np.stack([ind0, ind2, ind3], axis=1)
    return vertices, tris


def make_plane(size=(2e6, 2e6), height: float = 0.0) -> tuple[np.ndarray, np.ndarray]:
    """Two-triangle ground plane (``LAB/terrains/trimesh/utils.py`` make_plane, ray_caster.py:188)."""
    sx, sy = size[0] / 2, size[1] / 2
    v = np.array([[-sx, -sy, height], [sx, -sy, height], [sx, sy, height], [-sx, sy, height]], dtype=np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    return v, f


def _box(cx, cy, cz, sx, sy, sz):
    """Axis-aligned box as 8 vertices / 12 triangles (centre, full extents)."""
    hx, hy, hz = sx / 2, sy / 2, sz / 2
    v = np.array([[cx + dx * hx, cy + dy * hy, cz + dz * hz] for dx in (-1, 1) for dy in (-1, 1) for dz in (-1, 1)],
                 dtype=np.float32)
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6],
                  [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], dtype=np.int32)
    return v, f


@configclass
class TerrainGeneratorCfg:
    size: tuple = (8.0, 8.0)
    border_width: float = 20.0
    num_rows: int = 10
    num_cols: int = 20
    horizontal_scale: float = 0.1
    vertical_scale: float = 0.005
    slope_threshold: float = 0.75
    seed: int = 42


@configclass
class SyntheticTerrainCfg:
    terrain_type: str = "generator"
    terrain_generator: TerrainGeneratorCfg = TerrainGeneratorCfg()
    max_init_terrain_level: int = 5


class SyntheticTerrain:
    """Holds the merged mesh (vertices, triangles), tile origins and per-env origins."""

    def __init__(self, cfg: SyntheticTerrainCfg, num_envs: int):
        self.cfg = cfg
        g = cfg.terrain_generator
        rng = np.random.RandomState(g.seed)
        verts, faces, offset = [], [], 0
        self.terrain_origins = np.zeros((g.num_rows, g.num_cols, 3), dtype=np.float32)
        total_x, total_y = g.num_rows * g.size[0], g.num_cols * g.size[1]
        for col in range(g.num_cols):
            frac = col / g.num_cols
            for row in range(g.num_rows):
                difficulty = (row + rng.uniform()) / g.num_rows
                if frac < 0.2:
                    v, f, oz = self._stairs(difficulty, g, inverted=False)
                elif frac < 0.4:
                    v, f, oz = self._stairs(difficulty, g, inverted=True)
                elif frac < 0.6:
                    v, f, oz = self._boxes(difficulty, g, rng)
                elif frac < 0.8:
                    v, f, oz = self._random_rough(g, rng)
                elif frac < 0.9:
                    v, f, oz = self._slope(difficulty, g, inverted=False)
                else:
                    v, f, oz = self._slope(difficulty, g, inverted=True)
                ox = row * g.size[0] - total_x / 2
                oy = col * g.size[1] - total_y / 2
                v = v.copy()
                v[:, 0] += ox
                v[:, 1] += oy
                verts.append(v)
                faces.append(f + offset)
                offset += v.shape[0]
                self.terrain_origins[row, col] = (ox + g.size[0] / 2, oy + g.size[1] / 2, oz)
        # border frame at z = 0 (four boxes)
        b = g.border_width
        for cx, cy, sx, sy in ((0, -(total_y + b) / 2, total_x + 2 * b, b), (0, (total_y + b) / 2, total_x + 2 * b, b),
                               (-(total_x + b) / 2, 0, b, total_y), ((total_x + b) / 2, 0, b, total_y)):
            v, f = _box(cx, cy, -0.5, sx, sy, 1.0)
            verts.append(v)
            faces.append(f + offset)
            offset += v.shape[0]
        self.vertices = np.concatenate(verts).astype(np.float32)
        self.triangles = np.concatenate(faces).astype(np.int32)
        # env origins (terrain_importer.py:307-349): random level up to max_init_terrain_level, types round-robin
        max_level = min(cfg.max_init_terrain_level, g.num_rows - 1)
        self.terrain_levels = rng.randint(0, max_level + 1, size=num_envs)
        self.terrain_types = (np.arange(num_envs) * g.num_cols // max(num_envs, 1)).astype(np.int64)
        self.env_origins = self.terrain_origins[self.terrain_levels, self.terrain_types].astype(np.float32)

    @property
    def max_terrain_level(self) -> int:
        return self.terrain_origins.shape[0]                   # terrain_importer.py:333

    def device_state(self, device):
        """Device copies of what the terrain curriculum reads and writes (terrain_importer.py:307-324): levels (N) int64,
        types (N) int64, tile origins (rows, cols, 3) fp32.  Built once; ``terrain_levels`` is the live curriculum state."""
        import torch

        st = self.__dict__.get("_device_state")
        if st is None or st["device"] != str(device):
            st = dict(device=str(device),
                      terrain_levels=torch.as_tensor(np.asarray(self.terrain_levels, np.int64), device=device).clone(),
                      terrain_types=torch.as_tensor(np.asarray(self.terrain_types, np.int64), device=device).clone(),
                      terrain_origins=torch.as_tensor(self.terrain_origins, device=device).contiguous().clone())
            self._device_state = st
        return st

    @staticmethod
    def _hf_tile(z: np.ndarray, g) -> tuple[np.ndarray, np.ndarray, float]:
        width = int(g.size[0] / g.horizontal_scale) + 1
        length = int(g.size[1] / g.horizontal_scale) + 1
        heights = np.zeros((width, length), dtype=np.int16)
        heights[: z.shape[0], : z.shape[1]] = z
        v, f = convert_height_field_to_mesh(heights, g.horizontal_scale, g.vertical_scale, g.slope_threshold)
        c0, c1 = width // 2 - 10, width // 2 + 10
        oz = float(np.max(heights[c0:c1, c0:c1]) * g.vertical_scale)
        return v, f, oz

    def _random_rough(self, g, rng):
        """hf_terrains.py:21-79 with noise_range=(0.02, 0.10), noise_step=0.02, downsampled to 0.1 m."""
        wp, lp = int(g.size[0] / g.horizontal_scale), int(g.size[1] / g.horizontal_scale)
        hmin, hmax, hstep = int(0.02 / g.vertical_scale), int(0.10 / g.vertical_scale), int(0.02 / g.vertical_scale)
        hrange = np.arange(hmin, hmax + hstep, hstep)
        coarse = rng.choice(hrange, size=(wp, lp))
        x = np.linspace(0, g.size[0] * g.horizontal_scale, wp)
        y = np.linspace(0, g.size[1] * g.horizontal_scale, lp)
        z = interpolate.RectBivariateSpline(x, y, coarse)(x, y)
        return self._hf_tile(np.rint(z).astype(np.int16), g)

    def _slope(self, difficulty, g, inverted):
        """hf_terrains.py:82-148 with slope_range=(0.0, 0.4), platform_width=2.0."""
        slope = 0.0 + difficulty * 0.4
        if inverted:
            slope = -slope
        wp, lp = int(g.size[0] / g.horizontal_scale), int(g.size[1] / g.horizontal_scale)
        hmax = int(slope * g.size[0] / 2 / g.vertical_scale)
        cx, cy = int(wp / 2), int(lp / 2)
        xx = (cx - np.abs(cx - np.arange(wp))) / cx
        yy = (cy - np.abs(cy - np.arange(lp))) / cy
        hf = hmax * xx.reshape(wp, 1) * yy.reshape(1, lp)
        pw = int(2.0 / g.horizontal_scale / 2)
        z_pf = hf[wp // 2 - pw, lp // 2 - pw]
        hf = np.clip(hf, min(0, z_pf), max(0, z_pf))
        return self._hf_tile(np.rint(hf).astype(np.int16), g)

    @staticmethod
    def _stairs(difficulty, g, inverted):
        """Concentric box rings: step height 0.05..0.23 m, step width 0.3 m, 2 m platform (rough.py:14-31)."""
        step_h = 0.05 + difficulty * (0.23 - 0.05)
        step_w, platform = 0.3, 3.0
        size = g.size[0]
        n_steps = int((size - platform) / (2 * step_w))
        verts, faces, off = [], [], 0
        c = size / 2
        v, f = _box(c, c, -0.25, size, size, 0.5)  # ground slab, top at z = 0
        verts.append(v); faces.append(f); off += 8
        for k in range(n_steps):
            outer = size - 2 * k * step_w
            h = (k + 1) * step_h
            if inverted:
                h = (n_steps - k) * step_h
            v, f = _box(c, c, h / 2, outer, outer, h)
            verts.append(v); faces.append(f + off); off += 8
        top = (n_steps * step_h) if not inverted else step_h
        return np.concatenate(verts), np.concatenate(faces), float(top)

    @staticmethod
    def _boxes(difficulty, g, rng):
        """Random grid of boxes: cell 0.45 m, height 0.05..0.2 m, 2 m platform (rough.py:32-34)."""
        size, cell = g.size[0], 0.45
        hmax = 0.05 + difficulty * (0.2 - 0.05)
        n = int(size / cell)
        verts, faces, off = [], [], 0
        v, f = _box(size / 2, size / 2, -0.25, size, size, 0.5)
        verts.append(v); faces.append(f); off += 8
        for i in range(n):
            for j in range(n):
                cx, cy = (i + 0.5) * cell, (j + 0.5) * cell
                if abs(cx - size / 2) < 1.0 and abs(cy - size / 2) < 1.0:
                    continue
                h = rng.uniform(0.01, hmax)
                v, f = _box(cx, cy, h / 2, cell, cell, h)
                verts.append(v); faces.append(f + off); off += 8
        return np.concatenate(verts), np.concatenate(faces), 0.0
