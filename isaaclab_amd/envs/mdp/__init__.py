"""Stock MDP term library (mirror of ``isaaclab.envs.mdp``)."""
from .actions import *  # noqa: F401,F403
from .commands import *  # noqa: F401,F403
from .events import *  # noqa: F401,F403
from .observations import *  # noqa: F401,F403
from .rewards import *  # noqa: F401,F403
from .terminations import *  # noqa: F401,F403
