"""Replays a ``isaaclab_amd.envs.synthetic.Recording`` through the CPU oracle (TEST INFRASTRUCTURE: used by
tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py only)."""
from __future__ import annotations

import numpy as np

from . import mdp_np as P
from .actuators_np import F
from .velocity_step import VelocityStepOracle, anymal_spec, h1_spec


def net_weights_from_module(network):
    """(lstm_weights, decoder) in the oracle's layout from an nn.Module / ScriptModule with .lstm and Linear layers."""
    sd = {k: v.detach().cpu().numpy() for k, v in network.state_dict().items()}
    lstm, l = [], 0
    while f"lstm.weight_ih_l{l}" in sd:
        lstm.append(tuple(sd[f"lstm.{k}_l{l}"] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")))
        l += 1
    ids = sorted({int(k.split(".")[1]) for k in sd if k.startswith("decoder.") and k.endswith(".weight")})
    dec = [(sd[f"decoder.{i}.weight"], sd[f"decoder.{i}.bias"], 1 if n < len(ids) - 1 else 0) for n, i in enumerate(ids)]
    return lstm, dec


def build_oracle(rec, threads: int = 0) -> VelocityStepOracle:
    rough = rec.vertices is not None
    caster = starts = dirs = None
    if rough:
        from .raycast_np import GridCaster, tri_soup

        caster = GridCaster(tri_soup(rec.vertices, rec.triangles))
        starts, dirs = P.grid_pattern(0.1, [1.6, 1.0])
        starts = (starts + np.array([0.0, 0.0, 20.0], np.float32)).astype(np.float32)
    actuator = None
    if rec.name == "anymal_c_rough":
        lstm, dec = net_weights_from_module(rec.network)
        actuator = dict(kind="lstm", lstm_weights=lstm, decoder=dec, effort_limit=F(80.0), velocity_limit=F(7.5),
                        saturation=120.0)
    if rec.name.startswith("anymal"):
        spec = anymal_spec(rec.num_envs, rec.env_origins, rough=rough, caster=caster, ray_starts=starts, ray_dirs=dirs,
                           actuator=actuator, flat_weights=rec.name == "anymal_d_flat", threads=threads)
    elif rec.name == "h1_rough":
        spec = h1_spec(rec.num_envs, rec.env_origins, caster, starts, dirs, threads=threads)
    else:
        raise NotImplementedError(f"no oracle spec for config '{rec.name}'")
    if getattr(rec, "events", None) is not None:
        spec["events"] = rec.events
    if getattr(rec, "curriculum", None) is not None:
        spec["curriculum"] = rec.curriculum
    orc = VelocityStepOracle(spec)
    orc.prime(rec.field("dof_positions", 0), rec.field("dof_velocities", 0))
    return orc


def frame_dict(rec, t: int) -> dict:
    n = rec.num_envs
    d = dict(root_state=P.state_pack(rec.field("root_transforms", t), rec.field("root_velocities", t)),
             joint_pos=rec.field("dof_positions", t), joint_vel=rec.field("dof_velocities", t),
             forces=rec.field("net_contact_forces", t).reshape(n, -1, 3))
    if "link_transforms" in rec.fields:
        d["body_state"] = P.state_pack(rec.field("link_transforms", t), rec.field("link_velocities", t))
    return d


def oracle_step(orc: VelocityStepOracle, rec, k: int, auto_reset: bool = False):
    d = rec.decimation
    frames = [frame_dict(rec, 1 + k * d + i) for i in range(d)]
    cmd_u = rec.cmd_uniforms[k].numpy() if rec.cmd_uniforms is not None else None
    noise = {name: u[k].numpy() for name, u in rec.obs_noise.items()} if rec.obs_noise else None
    ev_u = {name: u[k].numpy() for name, u in rec.event_uniforms.items()} if getattr(rec, "event_uniforms", None) else None
    return orc.step(rec.actions[k].numpy(), frames, cmd_u, noise, auto_reset=auto_reset, event_uniforms=ev_u)


def replay_oracle(rec, steps: int, threads: int = 0, auto_reset: bool = False, logs: list | None = None,
                  episode_length0=None):
    """-> list of (obs, reward, terminated, time_outs) per env step.  ``auto_reset``: done envs are reset inside the step
    like ManagerBasedRLEnv.step does; ``logs`` (a list) then receives the step's ``extras["log"]`` dict or None."""
    orc = build_oracle(rec, threads)
    if episode_length0 is not None:
        orc.episode_length[:] = np.asarray(episode_length0, np.int64)
    out = []
    for k in range(steps):
        out.append(oracle_step(orc, rec, k, auto_reset))
        if logs is not None:
            logs.append(orc.log)
    return out
