"""Sensors on the MDP hot path (mirror of ``isaaclab.sensors``): contact sensor state machine and ray caster."""
from .contact_sensor import ContactSensor, ContactSensorCfg, ContactSensorData
from .frame_transformer import FrameTransformer, FrameTransformerCfg, FrameTransformerData
from .imu import Imu, ImuCfg, ImuData
from .ray_caster import RayCaster, RayCasterCfg, RayCasterData, patterns
from .sensor_base import SensorBase, SensorBaseCfg
