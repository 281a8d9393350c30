"""Per-robot env cfgs (``TASKS/manager_based/locomotion/velocity/config``)."""
from .anymal import (AnymalCRoughEnvCfg, AnymalDFlatEnvCfg, AnymalDRoughEnvCfg)
from .h1 import H1RoughEnvCfg
