"""Shadow-hand in-hand cube reorientation, direct workflow
(``TASKS/direct/inhand_manipulation``, ``TASKS/direct/shadow_hand/shadow_hand_env_cfg.py``)."""
from .inhand_manipulation_env import InHandManipulationEnv
from .shadow_hand_env_cfg import ShadowHandEnvCfg, ShadowHandOpenAIEnvCfg

__all__ = ["InHandManipulationEnv", "ShadowHandEnvCfg", "ShadowHandOpenAIEnvCfg"]
