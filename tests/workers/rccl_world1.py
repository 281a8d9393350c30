"""Worker of tests/test_zzz_rccl.py: ONE rank on backend "nccl" (= RCCL on ROCm) on a one-GPU box.  Every collective of the
multi-GPU layer is issued with ``force_collective`` so that the branch an 8-GPU node takes -- communicator
set-up, the collective on RCCL's own stream, ``Work.wait()`` as a device-side wait, a side stream next to hipGraph
replays of the env step -- runs here once.  With one rank an all-reduce(sum) is the identity, so every merged value
must equal the single-process value.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    port = int(sys.argv[1])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    from isaaclab_amd import distributed as D_
    from isaaclab_amd.envs.synthetic import make_velocity_env
    from isaaclab_amd.rl import RunningStandardScaler

    out = {"backend": dist.get_backend()}
    n, steps = 256, 4
    env, rec = make_velocity_env("anymal_d_rough", num_envs=n, device="cuda:0", steps=steps, seed=5, small_terrain=True,
                                 mesh_accel="grid", inject_random=False)
    env.observation_manager.persistent_output = True
    actions = rec.actions.to("cuda:0")
    for t in range(steps):
        obs = env.step(actions[t])[0]["policy"]
    torch.cuda.synchronize()
    # the env step as a hipGraph (what bench.py replays); collectives run next to its replays
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for t in range(steps):
            env.step(actions[t])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        for t in range(steps):
            obs = env.step(actions[t])[0]["policy"]
    graph.replay()
    torch.cuda.synchronize()
    mask = torch.rand(n, device="cuda:0") < 0.3
    width = obs.shape[1]

    # (a) episode statistics: side-stream partial sums + forced RCCL all-reduce next to graph replays
    stats = D_.EpisodeStats(env, 1)
    for _ in range(3):
        graph.replay()
        stats.reduce_async(mask, force_collective=True)
    got = stats.result()
    torch.cuda.synchronize()
    ref = D_.EpisodeStats(env, 1)
    ref.reduce(mask)                                     # same buffers, no collective
    exp = ref.result()
    out["stats_names"] = sorted(got)
    out["stats_max_abs_diff"] = max(abs(got[k] - exp[k]) for k in exp)
    out["families"] = sorted({k.split("/")[0] for k in got})

    # (b) normaliser row + statistics in ONE forced all-reduce (FusedStatistics), against the scaler's own local update
    sc_a = RunningStandardScaler(width, device="cuda:0")
    sc_b = RunningStandardScaler(width, device="cuda:0")
    fused = D_.FusedStatistics(D_.EpisodeStats(env, 1), sc_a, 1)
    for _ in range(3):
        graph.replay()
        fused.step(obs, mask, force_collective=True)
        sc_b(obs, train=True)                            # world size 1: the local Chan merge
    torch.cuda.synchronize()
    fs = fused.result()
    ref2 = D_.EpisodeStats(env, 1)
    ref2.reduce(mask)                                    # the buffers as the last fused step saw them, no collective
    exp2 = ref2.result()
    out["fused_stats_max_abs_diff"] = max(abs(fs[k] - exp2[k]) for k in exp2)
    out["scaler_mean_max_rel"] = float(((sc_a.running_mean - sc_b.running_mean).abs()
                                        / sc_b.running_mean.abs().clamp(min=1e-3)).max())
    out["scaler_var_max_rel"] = float(((sc_a.running_variance - sc_b.running_variance).abs()
                                       / sc_b.running_variance.abs().clamp(min=1e-6)).max())
    out["scaler_count"] = [float(sc_a.current_count), float(sc_b.current_count)]

    # (c) the scaler's own cross-rank path and the Shadow hand's partial sums through the process group
    sc_c = RunningStandardScaler(width, device="cuda:0")
    row = sc_c.local_row(obs)
    work = D_.allreduce_payload(row, 1, force=True)
    work.wait()
    sc_c.apply_merged_row(row)
    sc_d = RunningStandardScaler(width, device="cuda:0")
    sc_d(obs, train=True)
    torch.cuda.synchronize()
    out["scaler_row_path_max_rel"] = float(((sc_c.running_variance - sc_d.running_variance).abs()
                                            / sc_d.running_variance.abs().clamp(min=1e-6)).max())
    from isaaclab_amd.tasks.inhand import InHandManipulationEnv, ShadowHandEnvCfg

    vals = []
    for group in (None, dist.group.WORLD):
        torch.manual_seed(3)
        hand = InHandManipulationEnv(ShadowHandEnvCfg(num_envs=512), "cuda:0", group=group)
        hand.episode_length_buf.fill_(hand.max_episode_length - 2)        # every env times out: resets are counted
        hand.successes.copy_(torch.arange(512, device="cuda:0") % 5)
        hand.consecutive_successes.fill_(1.0)
        hand.step(torch.zeros(512, 20, device="cuda:0"))
        torch.cuda.synchronize()
        vals.append(float(hand.consecutive_successes[0]))
    out["consecutive_successes"] = vals
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
