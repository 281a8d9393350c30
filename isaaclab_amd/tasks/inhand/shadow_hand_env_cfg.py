"""``ShadowHandEnvCfg`` (``TASKS/direct/shadow_hand/shadow_hand_env_cfg.py:117-227``): the constants the MDP path
reads.  USD assets, spawn / physics properties and markers are out of scope."""
from __future__ import annotations

from ...utils.configclass import configclass
from ...utils.noise import GaussianNoiseCfg, NoiseModelWithAdditiveBiasCfg

# joint / body names of the Shadow hand in PhysX order (24 DoF, 26 bodies): the MDP path only resolves the actuated
# joints and the fingertip bodies by name (inhand_manipulation_env.py:44-56)
_FINGERS = ("FF", "MF", "RF")
SHADOW_HAND_JOINT_NAMES = (
    ["robot0_WRJ1", "robot0_WRJ0"]
    + [f"robot0_{f}J{k}" for f in _FINGERS for k in (3, 2, 1, 0)]
    + [f"robot0_LFJ{k}" for k in (4, 3, 2, 1, 0)]
    + [f"robot0_THJ{k}" for k in (4, 3, 2, 1, 0)]
)
SHADOW_HAND_BODY_NAMES = (
    ["robot0_forearm", "robot0_wrist", "robot0_palm"]
    + [f"robot0_{f.lower()}{seg}" for f in _FINGERS for seg in ("knuckle", "proximal", "middle", "distal")]
    + ["robot0_lfmetacarpal"] + [f"robot0_lf{seg}" for seg in ("knuckle", "proximal", "middle", "distal")]
    + [f"robot0_th{seg}" for seg in ("base", "proximal", "hub", "middle", "distal")]
    + ["robot0_mount"]
)


@configclass
class ShadowHandEnvCfg:
    # env (:119-126)
    decimation: int = 2
    episode_length_s: float = 10.0
    action_space: int = 20
    observation_space: int = 157
    state_space: int = 0
    asymmetric_obs: bool = False
    obs_type: str = "full"
    sim_dt: float = 1 / 120
    num_envs: int = 8192
    joint_names: list = SHADOW_HAND_JOINT_NAMES
    body_names: list = SHADOW_HAND_BODY_NAMES
    # :148-176
    actuated_joint_names: list = [
        "robot0_WRJ1", "robot0_WRJ0", "robot0_FFJ3", "robot0_FFJ2", "robot0_FFJ1", "robot0_MFJ3", "robot0_MFJ2",
        "robot0_MFJ1", "robot0_RFJ3", "robot0_RFJ2", "robot0_RFJ1", "robot0_LFJ4", "robot0_LFJ3", "robot0_LFJ2",
        "robot0_LFJ1", "robot0_THJ4", "robot0_THJ3", "robot0_THJ2", "robot0_THJ1", "robot0_THJ0",
    ]
    fingertip_body_names: list = ["robot0_ffdistal", "robot0_mfdistal", "robot0_rfdistal", "robot0_lfdistal",
                                  "robot0_thdistal"]
    object_init_pos: tuple = (0.0, -0.39, 0.6)
    # reset (:209-212)
    reset_position_noise: float = 0.01
    reset_dof_pos_noise: float = 0.2
    reset_dof_vel_noise: float = 0.0
    # reward scales (:213-226)
    dist_reward_scale: float = -10.0
    rot_reward_scale: float = 1.0
    rot_eps: float = 0.1
    action_penalty_scale: float = -0.0002
    reach_goal_bonus: float = 250
    fall_penalty: float = 0
    fall_dist: float = 0.24
    vel_obs_scale: float = 0.2
    success_tolerance: float = 0.1
    max_consecutive_success: int = 0
    av_factor: float = 0.1
    act_moving_average: float = 1.0
    force_torque_obs_scale: float = 10.0
    # DirectRLEnvCfg (direct_rl_env_cfg.py): noise models applied by DirectRLEnv.step (direct_rl_env.py:319-320, 376-377)
    action_noise_model: object = None
    observation_noise_model: object = None
    seed: int = 0


@configclass
class ShadowHandOpenAIEnvCfg(ShadowHandEnvCfg):
    """:228-283.  The ``events`` of the reference cfg randomise physics properties (materials, masses, gains, gravity):
    physics side, out of scope."""
    decimation: int = 3
    episode_length_s: float = 8.0
    observation_space: int = 42
    state_space: int = 187
    asymmetric_obs: bool = True
    obs_type: str = "openai"
    sim_dt: float = 1 / 60
    fall_penalty: float = -50
    success_tolerance: float = 0.4
    max_consecutive_success: int = 50
    act_moving_average: float = 0.3
    action_noise_model: object = NoiseModelWithAdditiveBiasCfg(
        noise_cfg=GaussianNoiseCfg(mean=0.0, std=0.05, operation="add"),
        bias_noise_cfg=GaussianNoiseCfg(mean=0.0, std=0.015, operation="abs"),
    )
    observation_noise_model: object = NoiseModelWithAdditiveBiasCfg(
        noise_cfg=GaussianNoiseCfg(mean=0.0, std=0.002, operation="add"),
        bias_noise_cfg=GaussianNoiseCfg(mean=0.0, std=0.0001, operation="abs"),
    )
