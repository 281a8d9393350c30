"""Synthetic physics-state generator: seeded host-side frames for the replay feed.

Produces a (T, frame_size) ring for a :class:`SyntheticPhysicsView` plus per-step actions; distributions and
seeds follow the measurement protocol (root pose around the env origin, joint state around the defaults,
Bernoulli contact forces with temporal hold so the air-time logic is exercised)."""
from __future__ import annotations

import numpy as np
import torch

from ..assets.physics_view import SyntheticPhysicsView


def _quat_from_euler_xyz_xyzw(roll, pitch, yaw):
    cy, sy = np.cos(yaw * 0.5), np.sin(yaw * 0.5)
    cr, sr = np.cos(roll * 0.5), np.sin(roll * 0.5)
    cp, sp = np.cos(pitch * 0.5), np.sin(pitch * 0.5)
    w = cy * cr * cp + sy * sr * sp
    x = cy * sr * cp - sy * cr * sp
    y = cy * cr * sp + sy * sr * cp
    z = sy * cr * cp - cy * sr * sp
    q = np.stack([x, y, z, w], axis=-1)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def generate_frames(view: SyntheticPhysicsView, num_frames: int, seed: int, env_origins: np.ndarray,
                    default_joint_pos: np.ndarray, body_names: list[str], with_links: bool = False) -> torch.Tensor:
    """Returns a host tensor (T, frame_size) laid out like ``view.live``."""
    rng = np.random.RandomState(seed)
    n, j, b = view.count, view.num_joints, view.num_contact_bodies
    frames = torch.zeros((num_frames, view.frame_size), dtype=torch.float32)
    fv = lambda name: view.field_view(frames, name).numpy()
    root_t, root_v = fv("root_transforms"), fv("root_velocities")
    q, qd, forces = fv("dof_positions"), fv("dof_velocities"), fv("net_contact_forces")
    # contact probabilities by body role
    p_contact = np.full(b, 0.005)
    for i, name in enumerate(body_names[:b]):
        low = name.lower()
        if "foot" in low or "ankle" in low:
            p_contact[i] = 0.5
        elif "thigh" in low or "knee" in low:
            p_contact[i] = 0.02
    state = rng.uniform(size=(n, b)) < p_contact
    base_xy = rng.uniform(-0.5, 0.5, size=(n, 2))
    base_yaw = rng.uniform(-np.pi, np.pi, size=n)
    for t in range(num_frames):
        pos = np.zeros((n, 3))
        pos[:, :2] = env_origins[:, :2] + base_xy + rng.normal(0, 0.02, size=(n, 2))
        pos[:, 2] = env_origins[:, 2] + rng.uniform(0.45, 0.75, size=n)
        quat = _quat_from_euler_xyz_xyzw(rng.normal(0, 0.15, n), rng.normal(0, 0.15, n),
                                         base_yaw + rng.normal(0, 0.05, n))
        root_t[t, :, :3] = pos
        root_t[t, :, 3:] = quat
        root_v[t] = rng.normal(0, 0.5, size=(n, 6))
        q[t] = default_joint_pos + rng.normal(0, 0.2, size=(n, j))
        qd[t] = rng.normal(0, 2.0, size=(n, j))
        # Markov hold: each (env, body) keeps its contact state with prob 0.9, else re-draws
        redraw = rng.uniform(size=(n, b)) < 0.1
        state = np.where(redraw, rng.uniform(size=(n, b)) < p_contact, state)
        fz = np.abs(rng.normal(0, 60.0, size=(n, b))) * state
        f = np.zeros((n, b, 3))
        f[..., 2] = fz
        f[..., :2] = rng.normal(0, 2.0, size=(n, b, 2)) * state[..., None]
        forces[t] = f.reshape(n * b, 3)
        if with_links and "link_transforms" in view._shapes:
            lt, lv, la = fv("link_transforms"), fv("link_velocities"), fv("link_accelerations")
            nb = view.num_bodies
            lt[t, :, :, :3] = pos[:, None, :] + rng.normal(0, 0.3, size=(n, nb, 3))
            lt[t, :, :, 3:] = quat[:, None, :]
            lv[t] = rng.normal(0, 0.5, size=(n, nb, 6))
            la[t] = rng.normal(0, 2.0, size=(n, nb, 6))
    return frames


def generate_actions(num_steps: int, num_envs: int, action_dim: int, seed: int) -> torch.Tensor:
    rng = np.random.RandomState(seed + 7919)
    return torch.from_numpy(np.clip(rng.normal(0, 1, size=(num_steps, num_envs, action_dim)), -3, 3).astype(np.float32))


# ---------------------------------------------------------------------------------------------------------
# Replay harness (SURVEY.md section 8c "MdpStepHarness"): env over a pre-generated ring of physics frames
# ---------------------------------------------------------------------------------------------------------
class Recording:
    """Pure data describing one synthetic run: everything a CPU checker needs to redo the same steps."""

    def __init__(self):
        self.name = ""
        self.num_envs = 0
        self.decimation = 4
        self.frames: torch.Tensor | None = None      # host (1 + steps * decimation, frame_size); frame 0 = initial state
        self.fields: dict[str, tuple[int, int, tuple]] = {}
        self.actions: torch.Tensor | None = None     # host (steps, N, A)
        self.env_origins: np.ndarray | None = None
        self.vertices: np.ndarray | None = None      # terrain mesh (None for the flat configs)
        self.triangles: np.ndarray | None = None
        self.cmd_uniforms: torch.Tensor | None = None  # host (steps, N, 7) or None (device generator)
        self.obs_noise: dict[str, torch.Tensor] = {}   # term name -> host (steps, N, dim) uniforms in [0, 1)
        self.network = None                           # actuator net module (anymal_c_rough)
        self._dev_noise: dict[str, torch.Tensor] = {}
        self._dev_cmd: torch.Tensor | None = None
        self.curriculum: dict | None = None           # terrain curriculum: initial levels, types, tile origins, tile size
        self.events: dict | None = None               # event cfg values + initial state (default root state, push timers)
        self.event_uniforms: dict[str, torch.Tensor] = {}   # event function name / "interval" -> host (steps, N, k)
        self._dev_events: dict[str, torch.Tensor] = {}

    def field(self, name: str, t: int) -> np.ndarray:
        o, s, shape = self.fields[name]
        return self.frames[t, o : o + s].view(shape).numpy()

    def feed(self, t: int):
        """Copies the injected random numbers of env-step ``t`` into the device tensors the kernels read."""
        for name, dev in self._dev_noise.items():
            dev.copy_(self.obs_noise[name][t], non_blocking=True)
        if self._dev_cmd is not None:
            self._dev_cmd.copy_(self.cmd_uniforms[t], non_blocking=True)
        for name, dev in self._dev_events.items():
            dev.copy_(self.event_uniforms[name][t], non_blocking=True)


def synthetic_anydrive_lstm(seed: int = 303, hidden: int = 8, layers: int = 2):
    """Stand-in for ``anydrive_3_lstm_jit.pt`` (remote in the reference, ASSETS/robots/anymal.py:47): a module with the
    attributes the actuator inspects (``.lstm`` state dict, actuator_net.py:53-54) -- LSTM(2 -> hidden, layers) +
    Linear(hidden -> 1), seeded random weights."""

    class SeaNet(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lstm = torch.nn.LSTM(2, hidden, layers, batch_first=True)
            self.decoder = torch.nn.Sequential(torch.nn.Linear(hidden, 1))

    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    net = SeaNet()
    torch.random.set_rng_state(gen_state)
    return net


def _env_cfg(name: str, num_envs: int, network=None):
    from ..tasks.velocity.config import anymal, h1

    if name == "anymal_d_flat":
        cfg = anymal.AnymalDFlatEnvCfg()
    elif name == "anymal_d_rough":
        cfg = anymal.AnymalDRoughEnvCfg()
    elif name == "anymal_c_rough":
        cfg = anymal.AnymalCRoughEnvCfg(network_file=network)
    elif name == "h1_rough":
        cfg = h1.H1RoughEnvCfg()
    else:
        raise ValueError(f"unknown velocity config '{name}' (anymal_d_flat | anymal_d_rough | anymal_c_rough | h1_rough)")
    cfg.scene.num_envs = num_envs
    return cfg


def make_rough_terrain(num_envs: int, small: bool = False, seed: int = 42, generator: str | None = None):
    """The terrain of the rough velocity tasks: ``ROUGH_TERRAINS_CFG`` (LAB/terrains/config/rough.py:12-51) through the
    restated ``TerrainGenerator`` with the curriculum layout the task cfg switches on (velocity_env_cfg.py:314-318) and
    ``max_init_terrain_level = 5`` (velocity_env_cfg.py:60) -- 10 x 20 tiles, ~1.19 M triangles; ``small``: 2 x 10 tiles
    (every sub-terrain type once per row), levels 0..1.  The generator reads numpy's and torch's global generators (like
    the reference); both are seeded for the build and restored afterwards.
    ``generator='synthetic'`` (or ``MDP_TERRAIN=synthetic``): the round-1 look-alike (``SyntheticTerrain``)."""
    import os

    from ..terrains import SyntheticTerrain, SyntheticTerrainCfg
    from ..terrains.synthetic import TerrainGeneratorCfg as SyntheticGeneratorCfg
    from ..terrains.terrain_generator import ROUGH_TERRAINS_CFG, GeneratedTerrain, TerrainImporterCfg

    if (generator or os.environ.get("MDP_TERRAIN", "generator")) == "synthetic":
        gen = SyntheticGeneratorCfg(num_rows=2, num_cols=5) if small else SyntheticGeneratorCfg()
        return SyntheticTerrain(SyntheticTerrainCfg(terrain_generator=gen, max_init_terrain_level=1 if small else 5), num_envs)
    cfg = ROUGH_TERRAINS_CFG.copy()
    cfg.curriculum, cfg.seed = True, seed
    if small:
        cfg.num_rows, cfg.num_cols = 2, 10
    np_state = np.random.get_state()
    with torch.random.fork_rng(devices=[]):
        np.random.seed(seed)
        torch.manual_seed(seed)
        try:
            return GeneratedTerrain(TerrainImporterCfg(terrain_generator=cfg, max_init_terrain_level=1 if small else 5,
                                                       num_envs=num_envs), num_envs)
        finally:
            np.random.set_state(np_state)


def make_velocity_env(name: str, num_envs: int, device: str = "cuda:0", steps: int = 8, seed: int = 0,
                      small_terrain: bool = False, inject_random: bool = True, network=None, terrain=None,
                      mesh_accel: str = "auto", events: bool = False):
    """Builds a velocity-task env over a replayed synthetic physics feed.  Returns ``(env, recording)``.

    ``inject_random=True`` replaces the device generators of the command term and of the observation noise by
    recorded uniforms (parity tests); ``False`` leaves the device-side Philox generators on (benchmark)."""
    from .manager_based_rl_env import ManagerBasedRLEnv

    if name == "anymal_c_rough" and network is None:
        network = synthetic_anydrive_lstm()
    cfg = _env_cfg(name, num_envs, network)
    if events:                              # reset + interval event terms of the task (off in the headline workload)
        from ..tasks.velocity.velocity_env_cfg import EventCfg

        cfg.events = EventCfg()
    rough = cfg.scene.height_scanner is not None
    if events and rough:                    # ... and the terrain-level curriculum where there is a generated terrain
        from ..tasks.velocity.velocity_env_cfg import CurriculumCfg

        cfg.curriculum = CurriculumCfg()
    if terrain is None and rough:
        terrain = make_rough_terrain(num_envs, small=small_terrain)
    mesh = None
    if rough and mesh_accel != "auto":     # tests: pin the acceleration structure (and with it the cast kernel)
        from ..utils.warp import convert_to_warp_mesh

        mesh = convert_to_warp_mesh(terrain.vertices, terrain.triangles, device, accel=mesh_accel)
    env = ManagerBasedRLEnv(cfg, device=device, terrain=terrain if rough else None, mesh=mesh)
    view = env.physics
    robot = env.scene["robot"]
    rec = Recording()
    rec.name, rec.num_envs, rec.decimation, rec.network = name, num_envs, cfg.decimation, network
    rec.env_origins = env.scene.env_origins.cpu().numpy()
    if rough:
        rec.vertices, rec.triangles = terrain.vertices, terrain.triangles
    rec.fields = {k: (o, s, tuple(view._shapes[k])) for k, (o, s) in view._slices.items()}
    nframes = 1 + steps * cfg.decimation
    rec.frames = generate_frames(view, nframes, seed, rec.env_origins, robot.data.default_joint_pos[0].cpu().numpy(),
                                 robot.body_names, with_links=True)
    rec.actions = generate_actions(steps, num_envs, env.action_manager.total_action_dim, seed)
    view.live.copy_(rec.frames[0])
    view.set_frames(rec.frames[1:])
    if inject_random:
        rng = np.random.RandomState(seed + 104729)
        rec.cmd_uniforms = torch.from_numpy(rng.uniform(size=(steps, num_envs, 7)).astype(np.float32))
        term = env.command_manager.get_term("base_velocity")
        rec._dev_cmd = torch.zeros((num_envs, 7), dtype=torch.float32, device=device)
        term.injected_uniforms = rec._dev_cmd
        om = env.observation_manager
        for tname, tcfg, dims in zip(om.active_terms["policy"], om._group_obs_term_cfgs["policy"],
                                     om.group_obs_term_dim["policy"]):
            if tcfg.noise is None:
                continue
            dim = int(np.prod(dims))
            rec.obs_noise[tname] = torch.from_numpy(rng.uniform(size=(steps, num_envs, dim)).astype(np.float32))
            dev = torch.zeros((num_envs, dim), dtype=torch.float32, device=device)
            tcfg._injected_noise = dev
            rec._dev_noise[tname] = dev
    if events:
        # a default root state that exercises the whole pose arithmetic, push timers that fire within the first steps
        rng = np.random.RandomState(seed + 7919)
        d = np.zeros((num_envs, 13), np.float32)
        d[:, 2] = 0.6
        ang = rng.uniform(-0.3, 0.3, size=(num_envs, 3)).astype(np.float32)
        cr, sr, cp, sp, cy, sy = (f(ang[:, k] * 0.5) for k in range(3) for f in (np.cos, np.sin))
        quat = np.stack([cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp,
                         sy * cr * cp - cy * sr * sp], axis=1)
        d[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
        d[:, 7:13] = rng.uniform(-0.05, 0.05, size=(num_envs, 6)).astype(np.float32)
        robot.data.default_root_state.copy_(torch.from_numpy(d))
        push0 = rng.uniform(0.0, 0.12, size=num_envs).astype(np.float32)
        env.event_manager._interval_term_time_left[0].copy_(torch.from_numpy(push0))
        ecfg = cfg.events
        rec.events = dict(default_root_state=d, push_time_left0=push0,
                          reset_base=dict(ecfg.reset_base.params), reset_robot_joints=dict(ecfg.reset_robot_joints.params),
                          push_robot=dict(velocity_range=ecfg.push_robot.params["velocity_range"],
                                          interval_range_s=ecfg.push_robot.interval_range_s))
        if rough:
            rec.curriculum = dict(terrain_levels0=np.asarray(terrain.terrain_levels, np.int64).copy(),
                                  terrain_types=np.asarray(terrain.terrain_types, np.int64).copy(),
                                  terrain_origins=terrain.terrain_origins.copy(),
                                  tile_size_x=float(terrain.cfg.terrain_generator.size[0]))
        if inject_random:
            j = robot.num_joints
            widths = {"reset_root_state_uniform": 12, "reset_joints_by_scale": 2 * j, "push_by_setting_velocity": 6}
            for fname, w in widths.items():
                rec.event_uniforms[fname] = torch.from_numpy(rng.uniform(size=(steps, num_envs, w)).astype(np.float32))
                dev = torch.zeros((num_envs, w), dtype=torch.float32, device=device)
                env.event_uniforms[fname] = dev
                rec._dev_events[fname] = dev
            if rough:
                rows = terrain.max_terrain_level
                rec.event_uniforms["terrain_levels_vel"] = torch.from_numpy(
                    rng.randint(0, rows, size=(steps, num_envs)).astype(np.int64))
                dev = torch.zeros(num_envs, dtype=torch.int64, device=device)
                env.event_uniforms["terrain_levels_vel"] = dev
                rec._dev_events["terrain_levels_vel"] = dev
            rec.event_uniforms["interval"] = torch.from_numpy(rng.uniform(size=(steps, num_envs)).astype(np.float32))
            dev = torch.zeros(num_envs, dtype=torch.float32, device=device)
            env.event_manager.interval_uniforms[0] = dev
            rec._dev_events["interval"] = dev
    return env, rec
