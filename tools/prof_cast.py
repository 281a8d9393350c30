#!/usr/bin/env python3
"""Profiling target: a handful of stand-alone height-scan casts and post-physics launches at the benchmark size, launched
directly (no graph) so rocprofv3 --pmc / --kernel-trace see individual dispatches.
    rocprofv3 --pmc <counters> -d gpurun_out/x -o x -- python3 tools/prof_cast.py"""
import os, sys; sys.path.insert(0, '.')
import torch
from isaaclab_amd.envs.synthetic import make_velocity_env
N = int(os.environ.get("BENCH_POST_ENVS", "4096"))
env, rec = make_velocity_env("anymal_c_rough", num_envs=N, device="cuda:0", steps=4, seed=5, inject_random=False)
env.observation_manager.persistent_output = True
acts = rec.actions.to("cuda:0")
for k in range(4):
    env.step(acts[k % acts.shape[0]])
env.scene.sensors['height_scanner'].cfg.update_period = 0.0
torch.cuda.synchronize()
om = env.observation_manager
for _ in range(int(os.environ.get("PROF_REPS", "8"))):
    om.compute_sensor_terms(tick_dt=env.physics_dt, from_physics=True)
    torch.cuda.synchronize()
    env._post_physics_single_launch()
    torch.cuda.synchronize()
