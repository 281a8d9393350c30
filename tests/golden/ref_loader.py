"""Loader that imports the *reference* hot-path modules in the build container.

Used ONLY by ``tests/golden/make_golden.py`` to generate the committed fixtures
(``tests/golden/*.npz``).  It needs ``/root/reference`` and therefore never runs
on the GPU box; nothing in ``tests/`` / ``bench.py`` / ``smoke()`` imports it.

Recipe: pre-seed ``sys.modules`` with name-only stubs for the
third-party modules that are missing in this image (toml, warp, prettytable,
omni.*, carb, h5py, ...) and for the simulator-bound ``isaaclab`` sub-packages
that are only used in type annotations, then put ``source/isaaclab`` on
``sys.path``.  No reference source text is copied: the stubs only carry names.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("ISAACLAB_REFERENCE", "/root/reference")
LAB = os.path.join(REF_ROOT, "source", "isaaclab")
TASKS = os.path.join(REF_ROOT, "source", "isaaclab_tasks", "isaaclab_tasks")


class _Swallow(types.ModuleType):
    """Module whose every attribute is another swallow / a dummy class."""

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        obj = _Dummy
        setattr(self, name, obj)
        return obj


class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy


class _Dummy(metaclass=_DummyMeta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getitem__(self, key):
        return _Dummy()

    def __setitem__(self, key, value):
        pass

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy()


def _stub(name: str, **attrs) -> types.ModuleType:
    mod = _Swallow(name)
    mod.__path__ = []  # mark as package so sub-imports resolve
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


_installed = False


def install_stubs():
    global _installed
    if _installed:
        return
    _installed = True
    # -- third party
    toml = _stub("toml")
    toml.load = lambda *a, **k: {"package": {"version": "0.0.0"}}
    wp = _stub("warp")
    wp.kernel = lambda *a, **k: (lambda f: f) if not (a and callable(a[0])) else a[0]

    class _PT:
        def __init__(self, *a, **k):
            self.align = {}
            self.title = ""
            self.field_names = []

        def add_row(self, *a, **k):
            pass

        def get_string(self, *a, **k):
            return ""

    _stub("prettytable", PrettyTable=_PT)
    for name in (
        "omni", "omni.log", "omni.kit", "omni.kit.app", "omni.usd", "omni.timeline", "omni.client",
        "omni.physics", "omni.physics.tensors", "omni.physics.tensors.impl", "omni.physics.tensors.impl.api",
        "carb", "h5py", "pxr", "isaacsim", "isaacsim.core", "isaacsim.core.utils", "isaacsim.core.utils.stage",
        "isaacsim.core.utils.torch", "isaacsim.core.prims", "isaacsim.core.simulation_manager",
        "isaacsim.core.version", "isaacsim.core.cloner", "gymnasium", "trimesh",
    ):
        _stub(name)
    log = sys.modules["omni.log"]
    for fn in ("info", "warn", "error", "verbose"):
        setattr(log, fn, lambda *a, **k: None)
    sys.modules["omni"].log = log
    # -- simulator-bound isaaclab sub-packages (annotation-only names)
    if LAB not in sys.path:
        sys.path.insert(0, LAB)
    import isaaclab  # noqa: F401  (real package __init__, needs the toml stub)

    for name in ("assets", "sensors", "markers", "terrains", "scene", "envs", "sim", "app", "controllers", "devices"):
        full = f"isaaclab.{name}"
        mod = _stub(full)
        # real search path: sub-modules (e.g. isaaclab.sensors.sensor_base) import from the reference tree,
        # while the package __init__ (which pulls the simulator) is never executed
        mod.__path__ = [os.path.join(LAB, "isaaclab", name)]
        setattr(sys.modules["isaaclab"], name, mod)
    _stub("isaaclab.markers.config")
    _stub("isaaclab.markers.visualization_markers")
    _stub("isaaclab.sim.spawners")
    _stub("isaaclab.sim.spawners.from_files")
    _stub("isaaclab.sim.utils")
    _stub("isaaclab.sensors.camera")
    _stub("isaaclab.sensors.camera.utils")
    _stub("isaaclab.terrains.trimesh")
    _stub("isaaclab.terrains.trimesh.utils")
    _stub("isaaclab_tasks")
    _stub("isaaclab_tasks.direct")
    _stub("isaaclab_tasks.direct.allegro_hand")
    _stub("isaaclab_tasks.direct.allegro_hand.allegro_hand_env_cfg")
    _stub("isaaclab_tasks.direct.shadow_hand")
    _stub("isaaclab_tasks.direct.shadow_hand.shadow_hand_env_cfg")


def load_by_path(mod_name: str, path: str):
    """Load a reference source file under a private module name."""
    install_stubs()
    spec = importlib.util.spec_from_file_location(mod_name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[mod_name] = mod
    spec.loader.exec_module(mod)
    return mod


def ref_math():
    """``isaaclab.utils.math`` loads stand-alone (only torch/numpy)."""
    return load_by_path("_ref_math", os.path.join(LAB, "isaaclab", "utils", "math.py"))


def ref_import(name: str):
    install_stubs()
    return importlib.import_module(name)


def ref_mdp(which: str):
    """which in {observations, rewards, terminations}."""
    return load_by_path(f"_ref_mdp_{which}", os.path.join(LAB, "isaaclab", "envs", "mdp", f"{which}.py"))


def ref_velocity_command():
    install_stubs()
    # the commands package __init__ pulls pose commands etc.; load the two files directly
    pkg = types.ModuleType("_ref_cmds")
    pkg.__path__ = [os.path.join(LAB, "isaaclab", "envs", "mdp", "commands")]
    sys.modules["_ref_cmds"] = pkg
    cfg = load_by_path("_ref_cmds.commands_cfg", os.path.join(pkg.__path__[0], "commands_cfg.py"))
    cmd = load_by_path("_ref_cmds.velocity_command", os.path.join(pkg.__path__[0], "velocity_command.py"))
    return cmd, cfg


def ref_velocity_mdp(which: str):
    return load_by_path(
        f"_ref_vel_mdp_{which}",
        os.path.join(TASKS, "manager_based", "locomotion", "velocity", "mdp", f"{which}.py"),
    )


def ref_inhand():
    return load_by_path(
        "_ref_inhand",
        os.path.join(TASKS, "direct", "inhand_manipulation", "inhand_manipulation_env.py"),
    )
