"""Height-field sub-terrains without trimesh (first slice): the numpy level of
``LAB/terrains/height_field/utils.py:20-173`` (border frame + origin + height field -> mesh, via
``synthetic.convert_height_field_to_mesh``) and of ``hf_terrains.py:21-232`` (``random_uniform_terrain``,
``pyramid_sloped_terrain``, ``pyramid_stairs_terrain``).  Same names, ``(difficulty, cfg)`` arguments, discretisation
(int16 heights in units of ``vertical_scale``) and random stream (``np.random.choice`` on the global generator) as the
reference, so a seeded reference run and a seeded run here build the same triangles; a sub-terrain comes back as
``([TriMesh(vertices, faces)], origin)`` where the reference returns ``trimesh.Trimesh`` objects.
Host-side input generation: no kernels, not a performance target."""
from __future__ import annotations

import functools
from typing import NamedTuple

import numpy as np
from scipy import interpolate

from ..utils.configclass import configclass
from .synthetic import convert_height_field_to_mesh


class TriMesh(NamedTuple):
    vertices: np.ndarray   # (V, 3) float32
    faces: np.ndarray      # (F, 3) int32


@configclass
class HfTerrainBaseCfg:
    """hf_terrains_cfg.py:16-32 (+ the sub-terrain fields the functions read)."""

    function: object = None            # the generator (hf_terrains_cfg.py:41,69,108)
    proportion: float = 1.0
    flat_patch_sampling: dict | None = None
    size: tuple = (8.0, 8.0)
    border_width: float = 0.0
    horizontal_scale: float = 0.1
    vertical_scale: float = 0.005
    slope_threshold: float | None = None


def _pixels(cfg):
    return int(cfg.size[0] / cfg.horizontal_scale), int(cfg.size[1] / cfg.horizontal_scale)


def height_field_to_mesh(func):
    """utils.py:20-76: the generated field sits inside a zero-height frame of ``border_width``, the mesh comes from
    ``convert_height_field_to_mesh`` and the origin is the tile centre at the highest point of its middle 2 m."""

    @functools.wraps(func)
    def wrapper(difficulty: float, cfg: HfTerrainBaseCfg):
        hs = cfg.horizontal_scale
        if 0 < cfg.border_width < hs:
            raise ValueError(
                f"The border width ({cfg.border_width}) must be greater than or equal to the"
                f" horizontal scale ({cfg.horizontal_scale})."
            )
        full = tuple(cfg.size)
        wp, lp = int(full[0] / hs) + 1, int(full[1] / hs) + 1
        bp = int(cfg.border_width / hs) + 1
        heights = np.zeros((wp, lp), dtype=np.int16)
        cfg.size = ((wp - 2 * bp) * hs, (lp - 2 * bp) * hs)      # the function generates the inner part only
        try:
            inner = func(difficulty, cfg)
        finally:
            cfg.size = full
        heights[bp:-bp, bp:-bp] = inner
        vertices, faces = convert_height_field_to_mesh(heights, hs, cfg.vertical_scale, cfg.slope_threshold)
        lo_x, hi_x = int((full[0] * 0.5 - 1) / hs), int((full[0] * 0.5 + 1) / hs)
        lo_y, hi_y = int((full[1] * 0.5 - 1) / hs), int((full[1] * 0.5 + 1) / hs)
        origin = np.array([0.5 * full[0], 0.5 * full[1], np.max(heights[lo_x:hi_x, lo_y:hi_y]) * cfg.vertical_scale])
        return [TriMesh(vertices, faces)], origin

    return wrapper


@height_field_to_mesh
def random_uniform_terrain(difficulty: float, cfg: HfRandomUniformTerrainCfg) -> np.ndarray:
    """hf_terrains.py:21-79: heights drawn on a coarse grid, bicubic-spline upsampled, rounded to int16."""
    if cfg.downsampled_scale is None:
        cfg.downsampled_scale = cfg.horizontal_scale
    elif cfg.downsampled_scale < cfg.horizontal_scale:
        raise ValueError(
            "Downsampled scale must be larger than or equal to the horizontal scale:"
            f" {cfg.downsampled_scale} < {cfg.horizontal_scale}."
        )
    wp, lp = _pixels(cfg)
    wd, ld = int(cfg.size[0] / cfg.downsampled_scale), int(cfg.size[1] / cfg.downsampled_scale)
    lo, hi, step = (int(v / cfg.vertical_scale) for v in (cfg.noise_range[0], cfg.noise_range[1], cfg.noise_step))
    coarse = np.random.choice(np.arange(lo, hi + step, step), size=(wd, ld))
    # (the reference's sample positions span size * horizontal_scale, not size: kept, the spline only sees ratios)
    ex, ey = cfg.size[0] * cfg.horizontal_scale, cfg.size[1] * cfg.horizontal_scale
    spline = interpolate.RectBivariateSpline(np.linspace(0, ex, wd), np.linspace(0, ey, ld), coarse)
    return np.rint(spline(np.linspace(0, ex, wp), np.linspace(0, ey, lp))).astype(np.int16)


@height_field_to_mesh
def pyramid_sloped_terrain(difficulty: float, cfg: HfPyramidSlopedTerrainCfg) -> np.ndarray:
    """hf_terrains.py:82-148: product of two tent functions scaled to the slope, cut flat at the platform height."""
    span = cfg.slope_range[1] - cfg.slope_range[0]
    slope = (-cfg.slope_range[0] - difficulty * span) if cfg.inverted else (cfg.slope_range[0] + difficulty * span)
    wp, lp = _pixels(cfg)
    peak = int(slope * cfg.size[0] / 2 / cfg.vertical_scale)
    cx, cy = int(wp / 2), int(lp / 2)
    tent_x = ((cx - np.abs(cx - np.arange(wp))) / cx).reshape(wp, 1)
    tent_y = ((cy - np.abs(cy - np.arange(lp))) / cy).reshape(1, lp)
    field = peak * tent_x * tent_y
    half = int(cfg.platform_width / cfg.horizontal_scale / 2)
    level = field[wp // 2 - half, lp // 2 - half]
    return np.rint(np.clip(field, min(0, level), max(0, level))).astype(np.int16)


@height_field_to_mesh
def pyramid_stairs_terrain(difficulty: float, cfg: HfPyramidStairsTerrainCfg) -> np.ndarray:
    """hf_terrains.py:151-232: nested squares, one step higher (lower when inverted) per ring, down to the platform."""
    rise = cfg.step_height_range[0] + difficulty * (cfg.step_height_range[1] - cfg.step_height_range[0])
    if cfg.inverted:
        rise *= -1
    wp, lp = _pixels(cfg)
    run = int(cfg.step_width / cfg.horizontal_scale)
    rise = int(rise / cfg.vertical_scale)
    platform = int(cfg.platform_width / cfg.horizontal_scale)
    field = np.zeros((wp, lp))
    x0, x1, y0, y1, level = 0, wp, 0, lp, 0
    while (x1 - x0) > platform and (y1 - y0) > platform:
        x0, x1, y0, y1, level = x0 + run, x1 - run, y0 + run, y1 - run, level + rise
        field[x0:x1, y0:y1] = level
    return np.rint(field).astype(np.int16)


# ---- configs (hf_terrains_cfg.py:34-136): defined behind the generators they name ----


@configclass
class HfRandomUniformTerrainCfg(HfTerrainBaseCfg):
    function = random_uniform_terrain
    noise_range: tuple = (0.02, 0.10)
    noise_step: float = 0.02
    downsampled_scale: float | None = None


@configclass
class HfPyramidSlopedTerrainCfg(HfTerrainBaseCfg):
    function = pyramid_sloped_terrain
    slope_range: tuple = (0.0, 0.4)
    platform_width: float = 1.0
    inverted: bool = False


@configclass
class HfPyramidStairsTerrainCfg(HfTerrainBaseCfg):
    function = pyramid_stairs_terrain
    step_height_range: tuple = (0.05, 0.23)
    step_width: float = 0.3
    platform_width: float = 1.0
    inverted: bool = False


@configclass
class HfInvertedPyramidSlopedTerrainCfg(HfPyramidSlopedTerrainCfg):
    """hf_terrains_cfg.py:88-97."""

    inverted: bool = True


@configclass
class HfInvertedPyramidStairsTerrainCfg(HfPyramidStairsTerrainCfg):
    """hf_terrains_cfg.py:127-136."""

    inverted: bool = True

