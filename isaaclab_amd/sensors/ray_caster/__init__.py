from . import patterns
from .ray_caster import RayCaster, RayCasterCfg, RayCasterData
