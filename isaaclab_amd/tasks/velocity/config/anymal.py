"""ANYmal env cfgs (``config/anymal_d/{rough,flat}_env_cfg.py``, ``config/anymal_c/rough_env_cfg.py``).

BASELINE configs: C1 = ``AnymalDFlatEnvCfg`` (DC motor, no height scan), C2 = ``AnymalDRoughEnvCfg`` (DC motor +
height scan), C3 = ``AnymalCRoughEnvCfg`` (LSTM actuator net)."""
from __future__ import annotations

from ....utils.configclass import configclass
from ...robots import ANYMAL_D_DCMOTOR_CFG, anydrive_3_lstm_actuator_cfg, anymal_cfg
from ..velocity_env_cfg import LocomotionVelocityRoughEnvCfg


@configclass
class AnymalDRoughEnvCfg(LocomotionVelocityRoughEnvCfg):
    def __post_init__(self):
        super().__post_init__()
        self.scene.robot = ANYMAL_D_DCMOTOR_CFG.copy()


@configclass
class AnymalDFlatEnvCfg(AnymalDRoughEnvCfg):
    def __post_init__(self):
        super().__post_init__()
        # flat_env_cfg.py:13-28
        self.rewards.flat_orientation_l2.weight = -5.0
        self.rewards.dof_torques_l2.weight = -2.5e-5
        self.rewards.feet_air_time.weight = 0.5
        self.scene.height_scanner = None
        self.observations.policy.height_scan = None


@configclass
class AnymalCRoughEnvCfg(LocomotionVelocityRoughEnvCfg):
    network_file: object = None

    def __post_init__(self):
        super().__post_init__()
        self.scene.robot = anymal_cfg({"legs": anydrive_3_lstm_actuator_cfg(self.network_file)})
