from .patterns import grid_pattern, pinhole_camera_pattern
from .patterns_cfg import GridPatternCfg, PatternBaseCfg, PinholeCameraPatternCfg
