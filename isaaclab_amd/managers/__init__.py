"""Managers of the MDP hot path (mirror of ``isaaclab.managers``)."""
from .action_manager import ActionManager, ActionTerm
from .command_manager import CommandManager, CommandTerm
from .curriculum_manager import CurriculumManager
from .event_manager import EventManager
from .manager_base import ManagerBase, ManagerTermBase
from .manager_term_cfg import (ActionTermCfg, CommandTermCfg, CurriculumTermCfg, EventTermCfg, ManagerTermBaseCfg,
                               ObservationGroupCfg, ObservationTermCfg, RewardTermCfg, TerminationTermCfg)
from .observation_manager import ObservationManager
from .reward_manager import RewardManager
from .scene_entity_cfg import SceneEntityCfg
from .termination_manager import TerminationManager
