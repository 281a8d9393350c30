from .actions_cfg import JointActionCfg, JointEffortActionCfg, JointPositionActionCfg, JointVelocityActionCfg
from .joint_actions import JointAction, JointEffortAction, JointPositionAction, JointVelocityAction
