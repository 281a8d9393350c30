from . import patterns
from .ray_caster import RayCaster, RayCasterCfg, RayCasterData
from .ray_caster_camera import CameraData, RayCasterCamera, RayCasterCameraCfg
