#!/usr/bin/env python3
"""Launch times of the SURVEY 8f "next" rows at the benchmark's env count (4096): event terms, interval timers, terrain
curriculum, reset statistics, masked fill, Imu, FrameTransformer, RayCasterCamera.  Each launch is replayed from a
hipGraph of 20 copies and timed with HIP events (kernel + graph gap, no Python); algorithmic bytes = inputs read once +
outputs written once for the rows the launch touches."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from isaaclab_amd import _lib as L
from isaaclab_amd.envs import mdp
from isaaclab_amd.envs.synthetic import make_velocity_env
from isaaclab_amd.sensors import FrameTransformer, FrameTransformerCfg, Imu, ImuCfg
from tools.bench_scaler import graph_time


def main():
    dev = "cuda:0"
    n = 4096
    env, rec = make_velocity_env("anymal_c_rough", num_envs=n, device=dev, steps=2, seed=3, small_terrain=True,
                                 mesh_accel="grid", events=True, inject_random=False)
    rec.feed(0)
    env.step(rec.actions[0].to(dev))
    g = torch.Generator(device="cpu").manual_seed(0)
    rows = []

    def add(name, fn, nbytes):
        t = graph_time(fn)
        rows.append((name, t, nbytes / t / 1e3 if nbytes else 0.0))

    for frac, label in ((22 / n, "22 envs"), (1.0, "all envs")):
        mask = (torch.rand(n, generator=g) < frac).to(torch.uint8).to(dev) if frac < 1 else torch.ones(n, dtype=torch.uint8, device=dev)
        k = int(mask.sum())
        add(f"reset_root_state_uniform ({label})", lambda: mdp.reset_root_state_uniform(
            env, mask, {"x": (-0.5, 0.5), "yaw": (-3.14, 3.14)}, {"x": (-0.5, 0.5)}), k * (13 + 3 + 13 + 7 + 6 + 17 * 6) * 4)
        add(f"reset_joints_by_scale ({label})", lambda: mdp.reset_joints_by_scale(env, mask, (0.5, 1.5), (0.0, 0.0)),
            k * 12 * (2 + 3 + 6) * 4)
        add(f"push_by_setting_velocity ({label})", lambda: mdp.push_by_setting_velocity(env, mask, {"x": (-0.5, 0.5)}),
            k * (6 + 6 + 6 + 17 * 6) * 4)
        arr, count, _ = env._masked_reset_plan()
        add(f"mdp_masked_fill, {count} buffers ({label})",
            lambda: L.check(L.lib.mdp_masked_fill(arr, count, mask.data_ptr(), n, L.stream(dev))), 0)
        from isaaclab_amd.tasks.velocity.mdp import terrain_levels_vel
        add(f"terrain_levels_vel ({label})", lambda: terrain_levels_vel(env, mask), k * (2 + 3 + 2 + 3 + 3) * 4)
    em = env.event_manager
    add("interval timers (mdp_event_interval_tick)", lambda: L.check(L.lib.mdp_event_interval_tick(
        n, em._interval_term_time_left[0].data_ptr(), 0.02, 10.0, 5.0, 0, None, em._interval_rng[0].data_ptr(),
        em._interval_masks[0].data_ptr(), L.stream(dev))), n * 9)
    rm, tm = env.reward_manager, env.termination_manager
    log = torch.zeros(1 + len(rm.active_terms) + 2 + len(tm.active_terms), device=dev)
    term = next(iter(env.command_manager._terms.values()))
    dones = (torch.rand(n, generator=g) < 0.01).to(torch.uint8).to(dev)
    add("mdp_reset_log (15 rows)", lambda: L.check(L.lib.mdp_reset_log(
        n, dones.data_ptr(), rm._episode_sums_block.data_ptr(), len(rm.active_terms), term.metrics["error_vel_xy"].data_ptr(),
        term.metrics["error_vel_yaw"].data_ptr(), tm._term_dones_block.data_ptr(), len(tm.active_terms), 20.0, log.data_ptr(),
        L.stream(dev))), n * log.numel())
    # sensors over body state
    tr = torch.randn(n, 7, device=dev)
    tr[:, 3:] /= tr[:, 3:].norm(dim=1, keepdim=True)
    vel, coms = torch.randn(n, 6, device=dev), torch.randn(n, 7, device=dev) * 0.05
    view = type("V", (), {"count": n, "get_transforms": lambda s: tr, "get_velocities": lambda s: vel, "get_coms": lambda s: coms})()
    imu = Imu(ImuCfg(), view, n, dev, 0.005)
    imu.update(0.005)

    def imu_step():
        imu._is_outdated.fill_(True)
        imu.data

    add("Imu update (+ flag fill)", imu_step, n * (7 + 6 + 3 + 3 + 4 + 3 + 6 + 6 + 19) * 4)
    bodies, nf = 3, 4
    m = n * bodies
    ftr = torch.randn(m, 7, device=dev)
    ftr[:, 3:] /= ftr[:, 3:].norm(dim=1, keepdim=True)
    tables = dict(target_frame_names=list("abcd"), per_env_indices=torch.arange(m).reshape(bodies, n).t().reshape(-1),
                  source_frame_body_ids=torch.arange(n) * bodies,
                  target_frame_body_ids=(torch.arange(n)[:, None] * bodies + torch.tensor([1, 2])[None]).reshape(-1),
                  duplicate_frame_indices=(torch.arange(n)[:, None] * 2 + torch.tensor([0, 0, 1, 1])[None]).reshape(-1))
    ft = FrameTransformer(FrameTransformerCfg(tables=tables), type("V", (), {"get_transforms": lambda s: ftr})(), n, dev, 0.005)
    ft.update(0.005)

    def ft_step():
        ft._is_outdated.fill_(True)
        ft.data

    add("FrameTransformer, 4 frames (+ flag fill, gate count)", ft_step, n * (bodies * 7 + 7 + nf * 14) * 4)
    # camera: 1024 envs x 32 x 24 rays on the env's terrain mesh
    nc, r = 1024, 32 * 24
    pose = torch.zeros(nc, 7, device=dev)
    pose[:, :3] = env.scene.env_origins[:nc] + torch.tensor([0.0, 0.0, 3.0], device=dev)
    pose[:, 3] = 1.0
    d = torch.randn(nc, r, 3, device=dev) * torch.tensor([0.4, 0.4, 0.0], device=dev) + torch.tensor([0.0, 0.0, -1.0], device=dev)
    d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
    bufs = dict(op=torch.zeros(nc, 3, device=dev), oq=torch.tensor([1.0, 0, 0, 0], device=dev).repeat(nc, 1), rs=torch.zeros(nc, r, 3, device=dev),
                pos=torch.zeros(nc, 3, device=dev), quat=torch.zeros(nc, 4, device=dev), hits=torch.zeros(nc, r, 3, device=dev),
                ip=torch.zeros(nc, r, device=dev), cam=torch.zeros(nc, r, device=dev))
    a = L.MdpRaycasterCameraArgs()
    a.mesh, a.view_pose, a.ld_pose, a.pose_xyzw = env.mesh.handle, pose.data_ptr(), 7, 0
    a.offset_pos, a.offset_quat, a.num_rays, a.depth_clipping, a.max_distance = bufs["op"].data_ptr(), bufs["oq"].data_ptr(), r, 1, 10.0
    a.ray_starts, a.ray_directions = bufs["rs"].data_ptr(), d.data_ptr()
    a.pos_w, a.quat_w_world, a.ray_hits_w = bufs["pos"].data_ptr(), bufs["quat"].data_ptr(), bufs["hits"].data_ptr()
    a.distance_to_image_plane, a.distance_to_camera = bufs["ip"].data_ptr(), bufs["cam"].data_ptr()
    add(f"RayCasterCamera, {nc} envs x 32 x 24 rays (grid mesh)", lambda: L.check(L.lib.mdp_raycaster_camera_update(nc, C.byref(a), L.stream(dev))),
        nc * r * (6 + 3 + 2) * 4)
    hit = float(torch.isfinite(bufs["cam"]).float().mean())
    for name, t, gbs in rows:
        print(f"{name:<62s} {t:8.2f} us" + (f"  {gbs:8.1f} GB/s algorithmic" if gbs else ""))
    print(f"(camera: {nc * r / rows[-1][1]:.0f} M rays/s, {hit:.2f} of the rays hit)")


if __name__ == "__main__":
    main()
