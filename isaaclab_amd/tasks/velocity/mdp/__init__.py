"""``isaaclab_tasks...velocity.mdp``: the stock library plus the task-specific terms."""
from ....envs.mdp import *  # noqa: F401,F403
from .curriculums import *  # noqa: F401,F403
from .rewards import *  # noqa: F401,F403
from .terminations import *  # noqa: F401,F403
