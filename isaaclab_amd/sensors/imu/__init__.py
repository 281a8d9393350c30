from .imu import Imu, ImuCfg, ImuData

__all__ = ["Imu", "ImuCfg", "ImuData"]
