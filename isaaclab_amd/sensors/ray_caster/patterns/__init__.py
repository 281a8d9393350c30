from .patterns import grid_pattern
from .patterns_cfg import GridPatternCfg, PatternBaseCfg
