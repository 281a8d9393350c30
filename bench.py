#!/usr/bin/env python3
"""Benchmark of the MDP hot path: env-steps/sec of full MDP compute on MI355X.

Workload (the configuration BASELINE.json's metric is quoted on): Isaac-Velocity-Rough-Anymal-C-v0 term set -- 4096
envs per GPU, 12 joints, 17 contact bodies, decimation 4 (ANYdrive LSTM actuator net + joint_acc + contact sensor per
substep; `--config anymal_d_rough` swaps in the DC-motor actuator of configs[1]), then terminations,
11 reward terms, the velocity command generator and the 235-wide observation with the 187-ray height scan cast
against a ~1.16 M-triangle rough terrain.  Physics state is a synthetic, seeded ring of frames resident in HBM
(SURVEY.md 8d); one "step" = ManagerBasedRLEnv.step over all envs of the rank.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Multi-GPU: envs are sharded (fixed 4096 per rank, weak scaling), no data-path
collective; the episode statistics are merged by one fused RCCL all-reduce per rollout of `--stats-every` steps.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Runtime setting, before HIP loads: hipGraph launches go through the runtime's ordinary dispatch path instead of its
# pre-captured AQL packets.  On this stack (ROCm 7.2) the packet-capture path starts a 100-node graph ~40 us later: the
# driver's command (--steps 20: one 20-step graph per timed region) 70.05 -> 68.02 us/step, the default 256-step region
# 67.65 -> 67.2, flat Anymal-D 42.8 -> 41.8 (same box, profiles/r03_experiments.txt exp58).  The work is the same;
# reported as config.hip_graph_packet_capture.  An explicit setting in the environment wins.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    "anymal_c_rough": "Isaac-Velocity-Rough-Anymal-C-v0 MDP compute (ANYdrive LSTM actuator net, seeded random weights)",
    "anymal_d_rough": "Isaac-Velocity-Rough-Anymal-D-v0 MDP compute (DC-motor actuator)",
    "anymal_d_flat": "Isaac-Velocity-Flat-Anymal-D-v0 MDP compute (DC-motor actuator, no height scan)",
    "h1_rough": "Isaac-Velocity-Rough-H1-v0 MDP compute (implicit actuators, per-GPU share of BASELINE configs[3])",
    "shadow": "Isaac-Repose-Cube-Shadow-Direct-v0 MDP compute (DirectRLEnv.step order, per-GPU share of BASELINE configs[4])",
}
# bodies referenced by terms / feet ("Bc contact bodies actually referenced by terms")
TERM_BODIES = {"h1_rough": (3, 2)}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=256)
    p.add_argument("--warmup", type=int, default=32)
    p.add_argument("--envs", type=int, default=0,
                   help="envs per GPU (default: 4096; h1_rough 8192 and shadow 16384 = their per-GPU share)")
    p.add_argument("--config", default="anymal_c_rough",
                   help="anymal_c_rough (BASELINE metric: LSTM actuator net + height scan) | anymal_d_rough | anymal_d_flat"
                        " | h1_rough | shadow")
    p.add_argument("--ring", type=int, default=32,
                   help="env-steps in the replayed frame ring (x decimation frames); min(ring, steps) is used")
    p.add_argument("--ring-mib", type=int, default=640,
                   help="minimum size of the resident frame ring in MiB: the ring is `sets` x --ring env-steps of frames, one "
                        "captured graph per set, replayed in rotation, so that a frame is read again only after > 2 x the "
                        "256 MiB Infinity Cache of other frames (0: one set, the ring may then stay cache-resident)")
    p.add_argument("--regions", type=int, default=0,
                   help="how often the timed region of exactly --steps steps is repeated (median reported); 0 = auto:"
                        " at least 50 and about 2 s of timed work")
    p.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-steps", type=int, default=128, help="env-steps of the CPU baseline sample (~15 s)")
    p.add_argument("--stats-every", type=int, default=24, help="rollout length between episode-stat all-reduces")
    p.add_argument("--small-terrain", action="store_true")
    p.add_argument("--fuse-cast", action="store_true", help="cast the height scan inside the observation kernel")
    p.add_argument("--auto-reset", choices=("off", "masked"), default="off",
                   help="masked: every step also resets its done envs without a host sync (not the headline metric)")
    p.add_argument("--normalize-obs", action="store_true",
                   help="also leave the standardised, clipped observation in a storage slot (not the headline metric)")
    p.add_argument("--events", action="store_true",
                   help="with --auto-reset masked: the task's reset / interval event terms and terrain curriculum too")
    a = p.parse_args()
    if a.config not in WORKLOADS:
        raise SystemExit(f"unknown --config {a.config!r}: {' | '.join(WORKLOADS)}")
    if a.envs <= 0:
        a.envs = {"h1_rough": 8192, "shadow": 16384}.get(a.config, 4096)
    return a


# ---- algorithmic bytes (SURVEY.md 8d counting rules; table repeated in DESIGN.md) -------------------------
def algorithmic_bytes(env) -> dict:
    """Per-env algorithmic HBM bytes of each launch of one env-step and the launches per env-step
    (fp32 = 4 B; every distinct per-env input read once, every output written once; constants, mesh, weights and
    the generator excluded)."""
    robot = env.scene["robot"]
    J = robot.num_joints
    D = env.cfg.decimation
    H = 3
    Bc, Bf = TERM_BODIES.get(getattr(env, "_bench_config", ""), (9, 4))   # term-referenced bodies (Anymal: 4 feet + 4 thighs + base), feet
    O = env.observation_manager.group_obs_dim["policy"][0]
    Tr = sum(1 for c in env.reward_manager._term_cfgs if c.weight != 0.0)
    Tt = len(env.termination_manager.active_terms)
    R = env.scene.sensors["height_scanner"].num_rays if "height_scanner" in env.scene.sensors else 0
    lstm_state_bytes = 0
    for act in robot.actuators.values():
        if hasattr(act, "sea_hidden_state"):
            layers, _, hidden = act.sea_hidden_state.shape
            lstm_state_bytes = 2 * (2 * layers * hidden) * 4
    table = {
        # name: (bytes per env per launch, launches per env-step, kernel-name fragment in the rocprof traces)
        # PD family: read target,q,qd; write computed,applied.  LSTM net: + (h, c) of every layer read and written
        "actuator": (20 * J + lstm_state_bytes * J, D, "k_lstm" if lstm_state_bytes else "k_actuator_pd"),
        # fused scene.update: joint_acc (read qd, RMW prev_qd, write acc) + contact sensor (read force, RMW 4 timers,
        # history shift) + root_state_w assembly (13 floats in, 13 out)
        "scene_update": (12 * J + 44 * Bc + 104, D, "k_scene_update"),
        "action": (16 * J, 1, "k_joint_action"),               # read raw, RMW action/prev_action, write target
        # episode counter RMW + terminations (base contact history, flags) + rewards (state in; sums / step out)
        "term_reward": (16 + 4 * (3 * H * 1) + Tt + 2 + 4 * (13 + 4 * J + 3 + 3 * H * Bc + 2 * Bf + J) + 12 * Tr + 4, 1,
                        "k_term_reward"),
        "command": (4 * 13 + 16 + 12, 1, "k_velocity_command"),  # root state; command RMW; timers
    }
    if R:
        # cast: pose in; ray_hits + sensor pose + the height-scan observation columns out (written by the cast kernel)
        table["raycast"] = (4 * 7 + 12 * R + 28 + 4 * R, 1, "k_raycaster_update")
        table["observation"] = (4 * (13 + 3 * J + 3) + 4 * (O - R), 1, "k_obs")  # state in; remaining columns out
    else:
        table["observation"] = (4 * (13 + 3 * J + 3) + 4 * O, 1, "k_obs")
    if getattr(env, "_single_launch_steps", 0) > 0:
        # launch structure of the default step: the actuator launches of sub-steps 2..D carry the scene update of the
        # sub-step before them, and everything after the last physics step is ONE launch (mdp_post_physics_step)
        act, upd = table["actuator"][0], table["scene_update"][0]
        post = upd + sum(table[k][0] for k in ("term_reward", "command", "observation", "raycast") if k in table)
        folded = env.fold_action and env.action_manager._fused_term is not None \
            and env.action_manager._fused_term.can_fold_into_actuator()
        table = {
            # the first actuator launch of the step also processes the action (mdp_actuator_*_compute_with_action)
            **({"action+actuator": (table["action"][0] + act, 1, table["actuator"][2])} if folded else
               {"action": table["action"], "actuator": (act, 1, table["actuator"][2])}),
            "actuator+scene_update": (act + upd, D - 1, table["actuator"][2]),
            "post_physics": (post, 1, "k_post_physics"),
        }
    return table


def _trace(rank, msg):
    if os.environ.get("MDP_BENCH_TRACE") == "1":
        print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if os.environ.get("MDP_BENCH_WATCHDOG"):       # diagnosis of a stalled rehearsal: dump every thread's stack and leave
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["MDP_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU path")
    # rehearsal of the N > 1 path on a one-GPU box (never used by the driver): every rank on cuda:0, gloo transport
    rehearsal = os.environ.get("MDP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(device))
    from isaaclab_amd.distributed import EpisodeStats
    from isaaclab_amd.envs.synthetic import make_velocity_env

    _trace(rank, "process group up")
    torch.manual_seed(1234 + rank)
    K = args.steps
    if K < 1:
        raise SystemExit("--steps must be >= 1")
    # The captured hipGraph holds one pass over the whole frame ring, so the ring never exceeds the timed step count;
    # a second graph holds the K % ring steps left over: every timed step is a replayed one for any --steps.
    ring = max(1, min(args.ring, K))
    tail = K % ring
    shadow = args.config == "shadow"
    # Cold frames: `sets` segments of `ring` env-steps each, every segment with a graph of its own, replayed in rotation --
    # between two reads of a frame lie sets x ring x decimation other frames, more than twice the 256 MiB Infinity Cache
    # (MI355X_MICROARCH.md: FETCH_SIZE counts Infinity-Cache hits as well, so a ring that fits it is not "cold")
    sets = 1
    if args.ring_mib > 0 and not args.no_graph:
        per_set = ring * frame_bytes_per_step(args)
        sets = int(min(64, max(1, math.ceil(args.ring_mib * 2.0 ** 20 / per_set)), max(1, (4 << 30) // per_set)))
    if shadow:
        if args.auto_reset != "off" or args.events or args.normalize_obs:
            raise SystemExit("--auto-reset / --events / --normalize-obs belong to the velocity configs")
        env, rec, actions = make_shadow_env(args, device, ring * sets, seed=1234 + 1 + 1000 * rank)
        if world > 1:       # consecutive_successes (:424-431) is reduced over every rank's envs inside the step
            import torch.distributed as dist

            env.group = dist.group.WORLD
        stats = None
    else:
        env, rec = make_velocity_env(args.config, num_envs=args.envs, device=device, steps=ring * sets, seed=1234 + 1 + 1000 * rank,
                                     small_terrain=args.small_terrain, inject_random=False, events=args.events)
        env._bench_config = args.config
        env.observation_manager.persistent_output = True
        if args.auto_reset == "masked":
            env.auto_reset = "masked"
            env.reset_log_ring = ring      # extras["log"] as views of a ring of buffers (no copy launch per step)
        if args.normalize_obs:
            from isaaclab_amd.rl import RunningStandardScaler

            width = env.observation_manager.group_obs_dim["policy"][0]
            env.set_obs_normalizer(RunningStandardScaler(width, device=device))
        if "height_scanner" in env.scene.sensors:
            env.scene.sensors["height_scanner"].fuse_into_observations = args.fuse_cast
        # episode_length_buf ~ U{0..999}
        g = torch.Generator().manual_seed(99 + rank)
        env.episode_length_buf.copy_(torch.randint(0, 1000, (args.envs,), generator=g))
        actions = rec.actions.to(device)
        stats = EpisodeStats(env, world)

    total = ring * sets                               # env-steps of frames resident in HBM
    frames_per_step = env.physics.num_frames // total   # = decimation

    def eager_steps(k0, count):
        for k in range(k0, k0 + count):
            env.step(actions[k % total])

    # warm-up (also builds the term programs), W untimed steps
    warm = max(args.warmup, 1)
    warm = (warm + total - 1) // total * total    # keep the frame cursor aligned with the ring
    eager_steps(0, warm)
    torch.cuda.synchronize()
    _trace(rank, "warm-up done")

    graphs, tail_graph = [], None
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                eager_steps(0, ring)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for s_ in range(sets):
                env.physics._cursor = s_ * ring * frames_per_step     # this set's segment of the ring
                g_ = torch.cuda.CUDAGraph()
                # thread_local: helper threads of the process group (watchdog, transport) may touch the runtime meanwhile
                with torch.cuda.graph(g_, capture_error_mode="thread_local"):
                    eager_steps(s_ * ring, ring)
                g_.replay()
                torch.cuda.synchronize()
                graphs.append(g_)
            if tail:
                # the tail graph replays the first `tail` steps of the first segment
                env.physics._cursor = 0
                tail_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(tail_graph, capture_error_mode="thread_local"):
                    eager_steps(0, tail)
                tail_graph.replay()
                torch.cuda.synchronize()
            env.physics._cursor = 0                # captures moved the host cursor; replays do not depend on it
        except Exception as e:  # capture is an optimisation; eager launches compute the same thing
            if rank == 0:
                print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs, tail_graph = [], None
            torch.cuda.synchronize()
    graph = graphs[0] if graphs else None
    next_set = [0]

    if stats is None:                   # Shadow hand: its one statistic (consecutive_successes) is reduced inside the step
        reduce_stats = lambda: None
    else:
        reduce_stats = stats.reduce if os.environ.get("MDP_BENCH_SYNC_STATS") == "1" else stats.reduce_async
    replayed = [0]

    def run(k_steps):
        """Exactly ``k_steps`` env steps: whole-ring graph replays, then the tail graph (eager only without graphs)."""
        done, next_stats, reduced_at = 0, args.stats_every, 0
        replayed[0] = 0
        while done < k_steps:
            left = k_steps - done
            if graph is not None and left >= ring:
                graphs[next_set[0]].replay()       # the segments take turns: a frame is cold again when its turn comes
                next_set[0] = (next_set[0] + 1) % len(graphs)
                done += ring
                replayed[0] += ring
            elif tail_graph is not None and left == tail:
                tail_graph.replay()
                done += tail
                replayed[0] += tail
            else:
                env.step(actions[done % total])
                done += 1
            if done >= next_stats:          # end of a rollout: merge the episode statistics across ranks
                reduce_stats()
                reduced_at = done
                while done >= next_stats:
                    next_stats += args.stats_every
        if reduced_at < done:               # the steps behind the last whole rollout (all of them when K < --stats-every)
            reduce_stats()

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()

    def timed_region():
        """EXACTLY K steps between barrier + synchronize on both sides."""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize()
        barrier()
        return time.perf_counter() - t0

    _trace(rank, f"graph {'captured' if graph is not None else 'unavailable'}")
    first = timed_region()
    _trace(rank, "first timed region done")
    # The region is repeated and the median reported: K = 20 steps are under 2 ms of GPU work, far too short for one
    # sample of a host clock.  Every rank runs the same count (derived from rank 0's first region).
    regions = args.regions
    if regions <= 0:
        est = torch.tensor([first], dtype=torch.float64, device=device)
        if world > 1:
            import torch.distributed as dist

            dist.broadcast(est, src=0)
        regions = int(min(2000, max(50, math.ceil(2.0 / max(float(est.item()), 1e-6)))))
        regions = max(5, min(regions, int(30.0 / max(float(est.item()), 1e-6))))   # ... and at most ~30 s
    times = torch.tensor([timed_region() for _ in range(regions)], dtype=torch.float64, device=device)
    _trace(rank, "timed regions done")
    episode_stats = stats.result() if stats is not None else {}   # drains the statistics collectives before anything is captured again
    _trace(rank, f"episode statistics merged ({len(episode_stats)} entries)")
    if world > 1:
        import torch.distributed as dist

        dist.all_reduce(times, op=dist.ReduceOp.MAX)       # a region takes as long as its slowest rank
    times = times.cpu().numpy()
    elapsed = float(np.median(times))
    value = world * args.envs * K / elapsed
    # Secondary figure, labelled as such: the same K steps replaying ONE segment over and over (what this bench timed up
    # to round 2): its frames stay in the 256 MiB Infinity Cache, as state a simulator has just written would.
    warm_ms = None
    if len(graphs) > 1 and world == 1:
        def warm_region():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done = 0
            while K - done >= ring:
                graphs[0].replay()
                done += ring
            if tail_graph is not None and K - done == tail:
                tail_graph.replay()
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        warm_region()
        warm_ms = 1e3 * float(np.median([warm_region() for _ in range(min(len(times), 40))])) / (K - (0 if tail_graph is not None or tail == 0 else tail))

    out = None
    resets = None
    if getattr(env, "_post_reset_log", None) is not None:
        resets = float(env._post_reset_log[0].item())      # envs the last timed step reset (its own log count)
    if rank == 0:
        roof = measure_roofline_shadow(env, actions) if shadow else measure_roofline(env, args)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_shadow(rec, args) if shadow else cpu_baseline(rec, args)
        out = {
            # BASELINE.json's metric text for the configuration it is quoted on; any other --config / --envs says what it ran
            "metric": ("env-steps/sec of MDP compute (Anymal-C rough, 4096 envs/GPU) + % HBM peak"
                       if args.config == "anymal_c_rough" and args.envs == 4096 else
                       f"env-steps/sec of MDP compute ({args.config}, {args.envs} envs/GPU) + % HBM peak"),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text(env, args, total),
                       "envs_per_gpu": args.envs, "hip_graph": graph is not None and replayed[0] == K,
                       "hip_graph_packet_capture": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0",
                       "graph_replayed_steps": replayed[0], "graph_ring_steps": ring, "graph_tail_steps": tail,
                       "graph_sets": len(graphs), "frame_ring_mib": env.physics.num_frames * frame_bytes(env) / 2.0 ** 20,
                       # what the step reads of the ring between two visits of a frame (vs the 256 MiB Infinity Cache)
                       "frame_ring_touched_mib": total * frame_bytes_per_step(args) / 2.0 ** 20,
                       "frames_cold": bool(len(graphs) > 0 and total * frame_bytes_per_step(args) >= 2 * 256 * 2 ** 20),
                       "ms_per_step_cache_resident_ring": warm_ms,
                       "timed_regions": int(len(times)), "region_ms_median": 1e3 * elapsed,
                       "region_ms_p10": 1e3 * float(np.percentile(times, 10)),
                       "region_ms_p90": 1e3 * float(np.percentile(times, 90)), "first_region_ms": 1e3 * first,
                       "stats_allreduce_every": args.stats_every,
                       "auto_reset": env.auto_reset if shadow else args.auto_reset,
                       **({"reset_goals": bool(env.reset_goals)} if shadow else {}),
                       "events": bool(args.events), "normalize_obs": bool(args.normalize_obs),
                       "reset_envs_last_step": float(env.reset_buf.sum().item()) if shadow else resets,
                       "reset_log_ring": int(getattr(env, "reset_log_ring", 0) or 0)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def frame_bytes(env) -> int:
    """Bytes of one synthetic physics frame resident in HBM."""
    fr = env.physics._frames
    return int(fr.shape[1] * fr.element_size())


def frame_bytes_per_step(args) -> int:
    """Frame bytes one env-step TOUCHES (decimation frames), from the robot's shape -- before the env exists.  A frame
    also holds link states the velocity tasks never read; what competes for the Infinity Cache is what is read."""
    # floats per env and frame: root transform 7 + root velocity 6 + 2 x joints + 3 x contact bodies (+ links for Shadow)
    shapes = {"anymal_c_rough": (12, 17, 4), "anymal_d_rough": (12, 17, 4), "anymal_d_flat": (12, 17, 4),
              "h1_rough": (19, 20, 4), "shadow": (24, 0, 2)}
    j, b, dec = shapes.get(args.config, (12, 17, 4))
    per_env = 13 + 2 * j + 3 * b
    if args.config == "shadow":
        per_env += 26 * 19 + 13 + 24 * 2       # link transforms / velocities / accelerations, object state, forces
    return int(per_env * 4 * args.envs * dec)


def workload_text(env, args, ring) -> str:
    if args.config == "shadow":
        return (f"{WORKLOADS[args.config]}: {args.envs} envs/GPU, {env.num_hand_dofs} joints, {env.hand.num_bodies} bodies,"
                f" {env.num_fingertips} fingertips, decimation {env.cfg.decimation}, {env.num_full_obs}-wide observation,"
                f" ring of {ring * env.cfg.decimation} synthetic state frames")
    robot = env.scene["robot"]
    scan = ""
    if "height_scanner" in env.scene.sensors:
        scan = (f" {env.scene.sensors['height_scanner'].num_rays}-ray height scan on a"
                f" {env.mesh.num_tris if getattr(env, 'mesh', None) is not None else 0}-triangle terrain,")
    return (f"{WORKLOADS[args.config]}: {args.envs} envs/GPU, {robot.num_joints} joints, {robot.num_bodies} bodies,"
            f" decimation {env.cfg.decimation},{scan} ring of {ring * env.cfg.decimation} synthetic state frames")


# ---- Shadow hand (BASELINE configs[4]): DirectRLEnv.step order over a synthetic feed -------------------------------
def make_shadow_env(args, device, ring, seed):
    """InHandManipulationEnv over a ring of random hand frames resident in HBM (generated on the device: the frames are
    36 MB each at 16384 envs) and a fixed object state near the palm; returns (env, recording dict, actions)."""
    from isaaclab_amd.tasks.inhand import InHandManipulationEnv, ShadowHandEnvCfg

    n = args.envs
    cfg = ShadowHandEnvCfg(num_envs=n)
    g = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device=device)
    u = lambda *s: torch.rand(*s, generator=g, device=device)
    D = len(cfg.joint_names)
    origins = r(n, 3) * 3.0
    limits = torch.stack([-1.0 - u(n, D), 1.0 + u(n, D)], dim=-1)
    env = InHandManipulationEnv(cfg, device, env_origins=origins, joint_pos_limits=limits)
    view = env.hand_view
    T = ring * cfg.decimation
    frames = torch.zeros(T, view.frame_size, device=device)
    lo, hi = env.hand_dof_lower_limits, env.hand_dof_upper_limits
    for t in range(T):
        view.field_view(frames, "dof_positions")[t] = lo + (hi - lo) * u(n, D)
        view.field_view(frames, "dof_velocities")[t] = r(n, D)
        lt = view.field_view(frames, "link_transforms")[t]
        lt[..., :3] = r(n, env.hand.num_bodies, 3) * 0.1 + origins[:, None, :]
        q = r(n, env.hand.num_bodies, 4)
        lt[..., 3:] = q / q.norm(dim=-1, keepdim=True)
        view.field_view(frames, "link_velocities")[t] = r(n, env.hand.num_bodies, 6)
    view.set_frames(frames)
    q = r(n, 4)
    q = q / q.norm(dim=-1, keepdim=True)
    obj_pos = r(n, 3) * 0.08 + torch.tensor(cfg.object_init_pos, device=device) + origins
    env.object_view.transforms.copy_(torch.cat([obj_pos, q[:, 1:], q[:, :1]], dim=1))
    env.object_view.velocities.copy_(r(n, 6))
    goal = r(n, 4)
    goal = goal / goal.norm(dim=-1, keepdim=True)
    goal[::9] = q[::9]
    env.goal_rot.copy_(goal)
    env.episode_length_buf.copy_(torch.randint(0, 400, (n,), generator=g, device=device))
    env.physics = view                      # the ring cursor the graph capture rewinds
    actions = u(ring, n, cfg.action_space) * 2.0 - 1.0
    rec = {"frames": frames, "actions": actions, "origins": origins, "limits": limits, "goal": goal,
           "obj": torch.cat([obj_pos, q, env.object_view.velocities], dim=1), "in_hand": env.in_hand_pos,
           "fingers": list(env.finger_bodies), "act_idx": list(env.actuated_dof_indices)}
    return env, rec, actions


def shadow_algorithmic_bytes(env) -> int:
    """read 2J + 13 + 13 x fingertips + actions + episode counter, write obs + reward + flags; actuator
    targets 20 B x J per sub-step."""
    J, Fg, A, D = env.num_hand_dofs, env.num_fingertips, env.cfg.action_space, env.cfg.decimation
    return 4 * (2 * J + 13 + 13 * Fg + A + 4) + 4 * (env.num_full_obs + 1 + 3) + 20 * J * D


def measure_roofline_shadow(env, actions) -> dict:
    """Per-launch times of the DirectRLEnv.step stages (hipGraph of 20 copies, HIP events) and the whole step."""
    N = env.num_envs
    J, Fg, A = env.num_hand_dofs, env.num_fingertips, env.cfg.action_space
    stages = {
        # name: (callable, algorithmic bytes per env, kernel-name fragment)
        "apply_action": (env._apply_action, 4 * A + 20 * J, "k_inhand_apply_action"),
        "dones": (env._get_dones, 4 * (7 + 3 + 4 + 1) + 8 + 2, "k_inhand_dones"),
        "rewards": (env._get_rewards, 4 * (7 + 3 + 4 + A + 1) + 2 + 4 + 1, "k_inhand_rewards"),
        "observations": (env.compute_full_observations, 4 * (2 * J + 13 + 13 * Fg + A) + 4 * env.num_full_obs,
                         "k_inhand_obs"),
        # masked _reset_idx of the step's done envs: one flag per env in, the reset rows of the few flagged envs out
        "reset_idx": (lambda: env._reset_idx_masked(env.reset_buf), 1, "k_inhand_reset_idx"),
    }
    per_kernel = {}
    for name, (fn, nbytes, frag) in stages.items():
        dur = _graph_time(fn)
        per_kernel[name] = {"kernel": frag, "us": 1e6 * dur, "launches_per_step": env.cfg.decimation if name == "apply_action" else 1,
                            "alg_bytes_per_env": nbytes, "GBps": nbytes * N / dur / 1e9}
    step_s = _graph_time(lambda: env.step(actions[0]), reps=8)
    dom = max(per_kernel, key=lambda k: per_kernel[k]["us"])
    d = per_kernel[dom]
    total = shadow_algorithmic_bytes(env)
    return {"bound": "hbm", "kernel": d["kernel"], "stage": dom, "achieved": d["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": d["GBps"] / HBM_PEAK_GBS, "traffic": _pmc_traffic(d["kernel"]), "traffic_source": _pmc_traffic_source(),
            "alg_bytes_per_launch": d["alg_bytes_per_env"] * N,
            "avg_launch_us": d["us"], "alg_bytes_per_env_step": total, "whole_step_us": 1e6 * step_s,
            "whole_step_GBps": total * N / step_s / 1e9, "kernels": per_kernel}


def cpu_baseline_shadow(rec, args) -> dict:
    """The numpy oracle (oracle/inhand_np.py) over the same inputs, bounded sample, one thread."""
    from oracle import inhand_np as O

    npy = lambda t: t.detach().cpu().numpy()
    lim = npy(rec["limits"])
    lower, upper = np.ascontiguousarray(lim[..., 0]), np.ascontiguousarray(lim[..., 1])
    origins, goal, obj, in_hand = npy(rec["origins"]), npy(rec["goal"]), npy(rec["obj"]), npy(rec["in_hand"])
    n, D = lower.shape
    from isaaclab_amd.tasks.inhand import ShadowHandEnvCfg

    cfg = ShadowHandEnvCfg(num_envs=n)
    B = len(cfg.body_names)
    rng = np.random.RandomState(0)
    q, qd = rng.randn(n, D).astype(np.float32), rng.randn(n, D).astype(np.float32)
    body = rng.randn(n, B, 13).astype(np.float32)
    act_idx, fingers = np.array(rec["act_idx"]), np.array(rec["fingers"])
    actions = npy(rec["actions"])
    succ, cons, ep = np.zeros(n, np.float32), np.zeros(1, np.float32), np.zeros(n, np.int64)
    cur, prev = np.zeros((n, D), np.float32), np.zeros((n, D), np.float32)
    kw = dict(dist_reward_scale=-10.0, rot_reward_scale=1.0, rot_eps=0.1, action_penalty_scale=-0.0002,
              success_tolerance=0.1, reach_goal_bonus=250.0, fall_dist=0.24, fall_penalty=0.0, av_factor=0.1)
    steps = max(8, args.cpu_steps)
    t0 = time.perf_counter()
    for k in range(steps):
        a = actions[k % actions.shape[0]]
        for _ in range(cfg.decimation):
            cur, prev = O.apply_action(a, act_idx, lower, upper, cur, prev, cfg.act_moving_average)
        ep = ep + 1
        oor, to, ep = O.dones(obj, origins, in_hand, goal, succ, ep, cfg.fall_dist, cfg.success_tolerance,
                              cfg.max_consecutive_success, 600)
        _, _, succ, cons = O.compute_rewards(oor | to, np.zeros(n, bool), succ, cons, obj, origins, in_hand, goal, a, **kw)
        O.full_observations(q, qd, lower, upper, obj, origins, in_hand, goal, body, fingers, a, cfg.vel_obs_scale)
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps} env-steps x {n} envs of the same workload (numpy oracle, single thread), {dt:.1f} s",
            "reference_torch_cpu": reference_torch_cpu("shadow")}


def _graph_time(fn, reps: int = 20, replays: int = 10) -> float:
    """Average duration (s) of ``fn`` -- one stage of the step, one kernel launch -- measured with HIP events on
    torch's current stream (the stream every launch goes to) around replays of a hipGraph holding ``reps`` copies of the
    launch: kernel time plus the graph's launch gap, no Python / ctypes time."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
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
    return 1e-3 * s.elapsed_time(e) / (reps * replays)


def _pmc_traffic(kernel_fragment: str):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, made by
    tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command).  Counters are in KiB;
    FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md section HBM; checked on k_lstm,
    whose compulsory reads are known)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    for name, row in table.get("kernels", {}).items():
        if kernel_fragment in name and "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            return (2.0 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024.0
    return None


def _pmc_traffic_source():
    """Where `roofline.traffic` comes from: it is NOT measured by this run (counters need their own rocprofv3 --pmc
    passes) but read from the committed table of the round's profile run; this names file, tag and commit."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    src = dict(table.get("source", {}))
    src["file"] = "profiles/pmc_traffic.json"
    src["measured_in_this_run"] = False
    return src


MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32, dense (MI355X_MICROARCH.md: equals the f32 vector peak)


def _mfma_summary(env, per_kernel) -> dict | None:
    """MFMA figures of the actuator-net launch (north_star: "MFMA utilisation for the actuator nets").  Instruction
    count and flop are exact properties of the kernel (one v_mfma_f32_16x16x4_f32 = 2048 flop per (env, joint) row per
    launch: 16 per 16-row wave, checked against SQ_INSTS_MFMA / SQ_INSTS_VALU_MFMA_MOPS_F32 of the committed counter
    pass); the rate uses the launch time measured live above.  `util_pmc` is rocprofv3's MfmaUtil definition,
    SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs), from profiles/pmc_mfma.json (own --pmc pass, C3 size)."""
    robot = env.scene["robot"]
    if not any(hasattr(a, "sea_hidden_state") for a in robot.actuators.values()):
        return None
    key = "actuator+scene_update" if "actuator+scene_update" in per_kernel else "actuator"
    if key not in per_kernel:
        return None
    rows = env.num_envs * robot.num_joints
    flops = 2048.0 * rows
    dur = per_kernel[key]["us"] * 1e-6
    out = {"kernel": "k_lstm8_reg", "stage": key, "mfma_insts_per_launch": rows, "flop_per_launch": flops,
           "achieved": flops / dur / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": flops / dur / 1e12 / MFMA_F32_PEAK_TFLOPS, "busy_cycles_per_launch": 32.0 * rows,
           "bound": "hbm (6 flop per algorithmic byte: the (h, c) state round-trips every sub-step)"}
    path = os.path.join(ROOT, "profiles", "pmc_mfma.json")
    if os.path.exists(path):
        with open(path) as f:
            pmc = json.load(f)
        out["util_pmc"] = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / (pmc["GRBM_GUI_ACTIVE_sum_over_8_xcds"] / 8.0 * pmc["simds"])
        out["util_pmc_envs"] = pmc["envs"]
        out["pmc_insts_per_row"] = pmc["SQ_INSTS_MFMA"] / pmc["rows"]
    return out


def measure_roofline(env, args) -> dict:
    """Times every launch of the step in isolation and reports the one with the largest share of the step."""
    table = algorithmic_bytes(env)
    N = env.num_envs
    robot = env.scene["robot"]
    sensors = env.scene.sensors
    split = env.observation_manager.can_split_sensor_terms()
    if "post_physics" in table:
        hs = sensors.get("height_scanner")
        if hs is not None:
            period = hs.cfg.update_period
            hs.cfg.update_period = 0.0      # every timed launch must cast (the step casts once per env-step)
        raw_action = env.action_manager.action.clone()

        def action_actuator():
            env.scene.write_data_to_sim(joint_action=env.action_manager.deferred_action(raw_action))

        fns = {
            "action": lambda: env.action_manager.process_action(raw_action),
            "actuator": lambda: robot.write_data_to_sim(),
            "action+actuator": action_actuator,
            "actuator+scene_update": lambda: env.scene.update_and_write(env.physics_dt),
            "post_physics": lambda: env._post_physics_single_launch(),
        }
        per_kernel = {}
        for name, fn in fns.items():
            if name not in table:
                continue
            dur = _graph_time(fn)
            nbytes, launches, frag = table[name]
            per_kernel[name] = {"kernel": frag, "us": 1e6 * dur, "launches_per_step": launches,
                                "alg_bytes_per_env": nbytes, "GBps": nbytes * N / dur / 1e9}
        if hs is not None:
            hs.cfg.update_period = period
        roof = _roofline_summary(table, per_kernel, N)
        mfma = _mfma_summary(env, per_kernel)
        if mfma is not None:
            roof["mfma"] = mfma
        return roof
    fns = {
        "actuator": lambda: robot.write_data_to_sim(),
        "scene_update": lambda: env.scene.update(env.physics_dt),
        "action": lambda: env.action_manager.process_action(env.action_manager.action),
        "term_reward": lambda: env._termination_reward_fused(),
        "command": lambda: env.command_manager.compute(dt=env.step_dt),
        "observation": lambda: env.observation_manager.compute(skip_sensor_terms=split),
    }
    fill_s = 0.0
    if "height_scanner" in sensors:
        hs = sensors["height_scanner"]

        def cast():                     # every timed launch must cast: re-arm the outdated flags first
            hs._is_outdated.fill_(True)
            if split:
                env.observation_manager.compute_sensor_terms()
            else:
                hs._update_outdated_buffers()

        fill_s = _graph_time(lambda: hs._is_outdated.fill_(True))
        fns["raycast"] = cast
    per_kernel = {}
    for name, fn in fns.items():
        dur = _graph_time(fn) - (fill_s if name == "raycast" else 0.0)
        nbytes, launches, frag = table[name]
        per_kernel[name] = {"kernel": frag, "us": 1e6 * dur, "launches_per_step": launches,
                            "alg_bytes_per_env": nbytes, "GBps": nbytes * N / dur / 1e9}
    return _roofline_summary(table, per_kernel, N)


def _roofline_summary(table, per_kernel, N) -> dict:
    # the dominant kernel: the launch with the longest duration (launch counts differ, but the roofline is per launch)
    dom = max(per_kernel, key=lambda k: per_kernel[k]["us"])
    d = per_kernel[dom]
    total_alg = sum(b * l for b, l, _ in table.values())
    return {"bound": "hbm", "kernel": d["kernel"], "stage": dom, "achieved": d["GBps"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": d["GBps"] / HBM_PEAK_GBS, "traffic": _pmc_traffic(d["kernel"]),
            "traffic_source": _pmc_traffic_source(),
            "alg_bytes_per_launch": d["alg_bytes_per_env"] * N, "avg_launch_us": d["us"],
            "alg_bytes_per_env_step": total_alg, "kernels": per_kernel}


def reference_torch_cpu(config: str):
    """The REFERENCE's own torch code for this path timed on CPU cores -- a stored figure, NOT measured by this run: it
    was taken in the build container (the reference cannot travel to the GPU box), for the BASELINE.md configuration
    that matches `config`; its ray cast is excluded (warp is not importable).  From profiles/r02_ref_cpu_baseline.json."""
    path = os.path.join(ROOT, "profiles", "r02_ref_cpu_baseline.json")
    row_of = {"anymal_d_flat": "C1", "anymal_d_rough": "C2", "anymal_c_rough": "C3", "h1_rough": "C4", "shadow": "C5"}
    if config not in row_of or not os.path.exists(path):
        return None
    with open(path) as f:
        rows = json.load(f).get("rows", {})
    r = rows.get(row_of[config])
    if r is None:
        return None
    return {"value": r["env_steps_per_s"], "unit": "env-steps/s", "cores": r["cores"], "kind": "reference",
            "row": row_of[config], "config": r["config"], "envs": r["envs"], "ray_cast": r["ray_cast"],
            "measured_in_this_run": False, "source": "profiles/r02_ref_cpu_baseline.json (build container, torch CPU)"}


def cpu_baseline(rec, args) -> dict:
    """The numpy + C oracle (a port of the reference algorithm) timed on the host cores, bounded sample."""
    from oracle.harness import build_oracle, oracle_step

    threads = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or min(os.cpu_count() or 1, 16)
    orc = build_oracle(rec, threads=threads)
    steps = args.cpu_steps
    ring = rec.actions.shape[0]
    oracle_step(orc, rec, 0)                       # warm-up (page-in, OpenMP pool)
    orc = build_oracle(rec, threads=threads)
    t0 = time.perf_counter()
    for k in range(steps):
        oracle_step(orc, rec, k % ring)            # the same ring of frames the GPU replays
    dt = time.perf_counter() - t0
    return {"value": rec.num_envs * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{steps} env-steps x {rec.num_envs} envs of the same workload and frames (numpy oracle; ray cast in C/OpenMP"
                      f" with {threads} threads, numpy part single-threaded), {dt:.1f} s",
            "reference_torch_cpu": reference_torch_cpu(args.config)}


if __name__ == "__main__":
    main()
