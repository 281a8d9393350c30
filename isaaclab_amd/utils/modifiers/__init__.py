from .modifier_cfg import DigitalFilterCfg, IntegratorCfg, ModifierCfg
from .modifier_base import ModifierBase
from .modifier import DigitalFilter, Integrator, bias, clip, scale
