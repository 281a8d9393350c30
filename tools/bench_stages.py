#!/usr/bin/env python3
"""Per-stage GPU time of one env step (GPU box): each stage is captured REPS times into a hipGraph and the replay is
timed with events, so the figure is kernel time + the graph's launch gap, not Python/ctypes overhead.
    python tools/bench_stages.py [--config anymal_c_rough] [--envs 4096]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isaaclab_amd.envs.synthetic import make_velocity_env


def graph_time(fn, reps=20, replays=10):
    """Average seconds per ``fn()`` call inside a replayed graph."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(replays):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / (reps * replays)


def stage_functions(env):
    robot = env.scene["robot"]
    sensors = env.scene.sensors
    fns = {
        "actuator": lambda: robot.write_data_to_sim(),
        "scene_update": lambda: env.scene.update(env.physics_dt),
        "action": lambda: env.action_manager.process_action(env.action_manager.action),
        "term_reward": lambda: env._termination_reward_fused(),
        "command": lambda: env.command_manager.compute(dt=env.step_dt),
    }
    if "height_scanner" in sensors:
        hs = sensors["height_scanner"]

        def cast():
            hs._is_outdated.fill_(True)
            env.observation_manager.compute_sensor_terms()

        fns["fill(1B x N)"] = lambda: hs._is_outdated.fill_(True)
        fns["fill+raycast"] = cast
    fns["observation"] = lambda: env.observation_manager.compute(skip_sensor_terms=True)
    return fns


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--config", default="anymal_c_rough")
    p.add_argument("--envs", type=int, default=4096)
    p.add_argument("--small-terrain", action="store_true")
    args = p.parse_args()
    env, rec = make_velocity_env(args.config, num_envs=args.envs, device="cuda:0", steps=4, seed=5,
                                 small_terrain=args.small_terrain, inject_random=False)
    env.observation_manager.persistent_output = True
    acts = rec.actions.to("cuda:0")
    for k in range(4):
        env.step(acts[k % acts.shape[0]])
    torch.cuda.synchronize()
    if hasattr(env, "mesh"):
        print(env.mesh.info())
    for name, fn in stage_functions(env).items():
        print(f"{name:<16} {1e6 * graph_time(fn):8.2f} us")


if __name__ == "__main__":
    main()
