"""H1 rough env cfg (``config/h1/rough_env_cfg.py:19-115``)."""
from __future__ import annotations

from ....managers import RewardTermCfg as RewTerm
from ....managers import SceneEntityCfg
from ....managers import TerminationTermCfg as DoneTerm
from ....utils.configclass import configclass
from ...robots import H1_CFG
from .. import mdp
from ..velocity_env_cfg import LocomotionVelocityRoughEnvCfg, RewardsCfg


@configclass
class H1Rewards(RewardsCfg):
    termination_penalty = RewTerm(func=mdp.is_terminated, weight=-200.0)
    lin_vel_z_l2 = None
    track_lin_vel_xy_exp = RewTerm(
        func=mdp.track_lin_vel_xy_yaw_frame_exp, weight=1.0, params={"command_name": "base_velocity", "std": 0.5}
    )
    track_ang_vel_z_exp = RewTerm(
        func=mdp.track_ang_vel_z_world_exp, weight=1.0, params={"command_name": "base_velocity", "std": 0.5}
    )
    feet_air_time = RewTerm(
        func=mdp.feet_air_time_positive_biped,
        weight=0.25,
        params={"command_name": "base_velocity",
                "sensor_cfg": SceneEntityCfg("contact_forces", body_names=".*ankle_link"), "threshold": 0.4},
    )
    feet_slide = RewTerm(
        func=mdp.feet_slide,
        weight=-0.25,
        params={"sensor_cfg": SceneEntityCfg("contact_forces", body_names=".*ankle_link"),
                "asset_cfg": SceneEntityCfg("robot", body_names=".*ankle_link")},
    )
    dof_pos_limits = RewTerm(func=mdp.joint_pos_limits, weight=-1.0,
                             params={"asset_cfg": SceneEntityCfg("robot", joint_names=".*_ankle")})
    joint_deviation_hip = RewTerm(func=mdp.joint_deviation_l1, weight=-0.2,
                                  params={"asset_cfg": SceneEntityCfg("robot", joint_names=[".*_hip_yaw", ".*_hip_roll"])})
    joint_deviation_arms = RewTerm(func=mdp.joint_deviation_l1, weight=-0.2,
                                   params={"asset_cfg": SceneEntityCfg("robot", joint_names=[".*_shoulder_.*", ".*_elbow"])})
    joint_deviation_torso = RewTerm(func=mdp.joint_deviation_l1, weight=-0.1,
                                    params={"asset_cfg": SceneEntityCfg("robot", joint_names="torso")})


@configclass
class H1RoughEnvCfg(LocomotionVelocityRoughEnvCfg):
    rewards: H1Rewards = H1Rewards()

    def __post_init__(self):
        super().__post_init__()
        self.scene.robot = H1_CFG.copy()
        if self.scene.height_scanner:
            self.scene.height_scanner.prim_path = "{ENV_REGEX_NS}/Robot/torso_link"
        self.rewards.undesired_contacts = None
        self.rewards.flat_orientation_l2.weight = -1.0
        self.rewards.dof_torques_l2.weight = 0.0
        self.rewards.action_rate_l2.weight = -0.005
        self.rewards.dof_acc_l2.weight = -1.25e-7
        self.commands.base_velocity.ranges.lin_vel_x = (0.0, 1.0)
        self.commands.base_velocity.ranges.lin_vel_y = (0.0, 0.0)
        self.commands.base_velocity.ranges.ang_vel_z = (-1.0, 1.0)
        self.terminations.base_contact.params["sensor_cfg"].body_names = ".*torso_link"
