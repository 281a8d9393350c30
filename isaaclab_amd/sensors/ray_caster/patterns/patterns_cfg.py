"""Pattern cfgs (``LAB/sensors/ray_caster/patterns/patterns_cfg.py``)."""
from __future__ import annotations

from collections.abc import Callable
from dataclasses import MISSING
from typing import Literal

from ....utils.configclass import configclass
from . import patterns


@configclass
class PatternBaseCfg:
    func: Callable = MISSING


@configclass
class GridPatternCfg(PatternBaseCfg):
    func: Callable = patterns.grid_pattern
    resolution: float = MISSING
    size: tuple = MISSING
    direction: tuple = (0.0, 0.0, -1.0)
    ordering: Literal["xy", "yx"] = "xy"


@configclass
class PinholeCameraPatternCfg(PatternBaseCfg):
    """``patterns_cfg.py:72-172``: focal length / apertures in cm (a tenth of the world unit)."""

    func: Callable = patterns.pinhole_camera_pattern
    focal_length: float = 24.0
    horizontal_aperture: float = 20.955
    vertical_aperture: float | None = None
    horizontal_aperture_offset: float = 0.0
    vertical_aperture_offset: float = 0.0
    width: int = MISSING
    height: int = MISSING

    @classmethod
    def from_intrinsic_matrix(cls, intrinsic_matrix: list, width: int, height: int, focal_length: float = 24.0):
        """Row-major [f_x, 0, c_x, 0, f_y, c_y, 0, 0, 1] -> apertures and offsets (patterns_cfg.py:120-172)."""
        f_x, c_x, f_y, c_y = intrinsic_matrix[0], intrinsic_matrix[2], intrinsic_matrix[4], intrinsic_matrix[5]
        return cls(focal_length=focal_length, horizontal_aperture=width * focal_length / f_x,
                   vertical_aperture=height * focal_length / f_y, horizontal_aperture_offset=(c_x - width / 2) / f_x,
                   vertical_aperture_offset=(c_y - height / 2) / f_y, width=width, height=height)
