"""Action cfgs (``LAB/envs/mdp/actions/actions_cfg.py``), joint actions only."""
from __future__ import annotations

from dataclasses import MISSING

from ....managers.manager_term_cfg import ActionTermCfg
from ....utils.configclass import configclass
from . import joint_actions


@configclass
class JointActionCfg(ActionTermCfg):
    joint_names: list = MISSING
    scale: float | dict = 1.0
    offset: float | dict = 0.0
    preserve_order: bool = False


@configclass
class JointPositionActionCfg(JointActionCfg):
    class_type: type = joint_actions.JointPositionAction
    use_default_offset: bool = True


@configclass
class JointVelocityActionCfg(JointActionCfg):
    class_type: type = joint_actions.JointVelocityAction
    use_default_offset: bool = True


@configclass
class JointEffortActionCfg(JointActionCfg):
    class_type: type = joint_actions.JointEffortAction
