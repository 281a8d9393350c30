from .commands_cfg import UniformVelocityCommandCfg
from .velocity_command import UniformVelocityCommand

__all__ = ["UniformVelocityCommand", "UniformVelocityCommandCfg"]
