#!/usr/bin/env python3
"""Same-box A/B timing of the two big launches (GPU box): the stand-alone height-scan cast and the post-physics launch,
each as 20 copies in a hipGraph.  Run-to-run differences between boxes are ~+-1.5 us, so compare builds in ONE call:
    for v in old new; do MDP_HIP_LIB=isaaclab_amd/_lib/libmdp_hip_$v.so python tools/bench_post.py; done"""
import sys, os; sys.path.insert(0, '.')
import torch
from isaaclab_amd.envs.synthetic import make_velocity_env
from tools.bench_stages import graph_time
N = int(os.environ.get("BENCH_POST_ENVS", "4096"))
env, rec = make_velocity_env("anymal_c_rough", num_envs=N, device="cuda:0", steps=4, seed=5, inject_random=False,
                             small_terrain=os.environ.get("BENCH_POST_SMALL") == "1")
if os.environ.get("BENCH_POST_SAMEPOSE") == "1":      # every env at env 0's pose in every frame: all casts share lines
    fr = env.physics._frames
    rt = env.physics.field_view(fr, "root_transforms")
    rt[:] = rt[:, :1]
env.observation_manager.persistent_output = True
acts = rec.actions.to("cuda:0")
for k in range(4):
    env.step(acts[k % acts.shape[0]])
env.scene.sensors['height_scanner'].cfg.update_period = 0.0
torch.cuda.synchronize()
om = env.observation_manager
for _ in range(3):
    print("cast launch %.2f us" % (1e6 * graph_time(lambda: om.compute_sensor_terms(tick_dt=env.physics_dt, from_physics=True))),
          "post launch %.2f us" % (1e6 * graph_time(lambda: env._post_physics_single_launch())))
