"""Robot constants consumed by the MDP path (``source/isaaclab_assets/isaaclab_assets/robots``): joint / body
names in PhysX (breadth-first) order, default joint positions, actuator groups.  USD assets, spawn and
physics properties are out of scope."""
from __future__ import annotations

from ..actuators import ActuatorNetLSTMCfg, DCMotorCfg, ImplicitActuatorCfg
from ..assets import ArticulationCfg

_LEGS = ("LF", "LH", "RF", "RH")
ANYMAL_JOINT_NAMES = [f"{leg}_{j}" for j in ("HAA", "HFE", "KFE") for leg in _LEGS]
ANYMAL_BODY_NAMES = ["base"] + [f"{leg}_{b}" for b in ("HIP", "THIGH", "SHANK", "FOOT") for leg in _LEGS]

# anymal.py:34-41
ANYDRIVE_3_SIMPLE_ACTUATOR_CFG = DCMotorCfg(
    joint_names_expr=[".*HAA", ".*HFE", ".*KFE"], saturation_effort=120.0, effort_limit=80.0, velocity_limit=7.5,
    stiffness={".*": 40.0}, damping={".*": 5.0},
)


def anydrive_3_lstm_actuator_cfg(network_file) -> ActuatorNetLSTMCfg:
    """anymal.py:45-51; the network file lives on a remote server in the reference, so it is a parameter here."""
    return ActuatorNetLSTMCfg(
        joint_names_expr=[".*HAA", ".*HFE", ".*KFE"], network_file=network_file, saturation_effort=120.0,
        effort_limit=80.0, velocity_limit=7.5,
    )


_ANYMAL_DEFAULT_JOINT_POS = {".*HAA": 0.0, ".*F_HFE": 0.4, ".*H_HFE": -0.4, ".*F_KFE": -0.8, ".*H_KFE": 0.8}


def anymal_cfg(actuators: dict) -> ArticulationCfg:
    """anymal.py:93-159 (ANYmal-C/D share everything the MDP path reads)."""
    return ArticulationCfg(
        joint_names=ANYMAL_JOINT_NAMES, body_names=ANYMAL_BODY_NAMES, default_joint_pos=_ANYMAL_DEFAULT_JOINT_POS,
        soft_joint_pos_limit_factor=0.95, joint_pos_limits=(-2.4, 2.4), joint_vel_limits=7.5,
        joint_effort_limits=80.0, actuators=actuators,
    )


ANYMAL_D_DCMOTOR_CFG = anymal_cfg({"legs": ANYDRIVE_3_SIMPLE_ACTUATOR_CFG})

# unitree.py:216-260 (H1): 19 joints, three implicit-actuator groups addressed by index lists
H1_JOINT_NAMES = [
    "left_hip_yaw", "right_hip_yaw", "torso", "left_hip_roll", "right_hip_roll", "left_shoulder_pitch",
    "right_shoulder_pitch", "left_hip_pitch", "right_hip_pitch", "left_shoulder_roll", "right_shoulder_roll",
    "left_knee", "right_knee", "left_shoulder_yaw", "right_shoulder_yaw", "left_ankle", "right_ankle",
    "left_elbow", "right_elbow",
]
H1_BODY_NAMES = [
    "pelvis", "left_hip_yaw_link", "right_hip_yaw_link", "torso_link", "left_hip_roll_link", "right_hip_roll_link",
    "left_shoulder_pitch_link", "right_shoulder_pitch_link", "left_hip_pitch_link", "right_hip_pitch_link",
    "left_shoulder_roll_link", "right_shoulder_roll_link", "left_knee_link", "right_knee_link",
    "left_shoulder_yaw_link", "right_shoulder_yaw_link", "left_ankle_link", "right_ankle_link",
    "left_elbow_link", "right_elbow_link",
]
H1_CFG = ArticulationCfg(
    joint_names=H1_JOINT_NAMES, body_names=H1_BODY_NAMES,
    default_joint_pos={".*_hip_yaw": 0.0, ".*_hip_roll": 0.0, ".*_hip_pitch": -0.28, ".*_knee": 0.79,
                       ".*_ankle": -0.52, "torso": 0.0, ".*_shoulder_pitch": 0.28, ".*_shoulder_roll": 0.0,
                       ".*_shoulder_yaw": 0.0, ".*_elbow": 0.52},
    soft_joint_pos_limit_factor=0.9, joint_pos_limits=(-2.6, 2.6), joint_vel_limits=100.0, joint_effort_limits=300.0,
    actuators={
        "legs": ImplicitActuatorCfg(
            joint_names_expr=[".*_hip_yaw", ".*_hip_roll", ".*_hip_pitch", ".*_knee", "torso"],
            effort_limit=300, velocity_limit=100.0,
            stiffness={".*_hip_yaw": 150.0, ".*_hip_roll": 150.0, ".*_hip_pitch": 200.0, ".*_knee": 200.0, "torso": 200.0},
            damping={".*_hip_yaw": 5.0, ".*_hip_roll": 5.0, ".*_hip_pitch": 5.0, ".*_knee": 5.0, "torso": 5.0},
        ),
        "feet": ImplicitActuatorCfg(joint_names_expr=[".*_ankle"], effort_limit=100, velocity_limit=100.0,
                                    stiffness={".*_ankle": 20.0}, damping={".*_ankle": 4.0}),
        "arms": ImplicitActuatorCfg(
            joint_names_expr=[".*_shoulder_pitch", ".*_shoulder_roll", ".*_shoulder_yaw", ".*_elbow"],
            effort_limit=300, velocity_limit=100.0,
            stiffness={".*_shoulder_pitch": 40.0, ".*_shoulder_roll": 40.0, ".*_shoulder_yaw": 40.0, ".*_elbow": 40.0},
            damping={".*_shoulder_pitch": 10.0, ".*_shoulder_roll": 10.0, ".*_shoulder_yaw": 10.0, ".*_elbow": 10.0},
        ),
    },
)
