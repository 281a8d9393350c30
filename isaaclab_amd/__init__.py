"""isaaclab_amd: MI355X-native implementation of Isaac Lab's per-step vectorised MDP compute.

Python host code (this package) mirrors the reference operator interface; all arithmetic runs in
hand-written gfx950 HIP kernels behind the C ABI of ``include/mdp_hip.h`` (``libmdp_hip.so``).
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library has not been built)

__version__ = "0.1.0"
