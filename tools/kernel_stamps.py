#!/usr/bin/env python3
"""Where the time goes INSIDE the two big launches of the step (GPU box, diagnostic build):

    make -C isaaclab_amd/csrc STAMPS=1
    MDP_HIP_LIB=isaaclab_amd/_lib/libmdp_hip_stamps.so python tools/kernel_stamps.py

Every block records wall-clock stamps (100 MHz) at its phase boundaries; the script launches ONE post-physics kernel
and ONE actuator+scene-update kernel and prints, per block role, when blocks start and finish relative to the first
block of the launch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from isaaclab_amd import _lib as L
from isaaclab_amd.envs.synthetic import make_velocity_env


def show(name, st, roles, phases):
    st = st.cpu().numpy().astype(np.int64)
    used = st[:, 0] != 0
    st = st[used].copy()
    hw = st[:, 7] >> 8
    st[:, 7] &= 0xFF
    t0 = st[:, 0].min()
    print(f"== {name}: {used.sum()} blocks, launch span {(st[:, 1:6].max() - t0) / 100.0:.2f} us")
    for tag, rname in roles.items():
        r = st[st[:, 7] == tag]
        if len(r) == 0:
            continue
        last = phases[tag]
        rel = (r[:, : last + 1] - t0) / 100.0
        line = f"  {rname:<8} n={len(r):5d} start min/med/max {rel[:, 0].min():6.2f}/{np.median(rel[:, 0]):6.2f}/{rel[:, 0].max():6.2f}"
        line += f"  end min/med/max {rel[:, last].min():6.2f}/{np.median(rel[:, last]):6.2f}/{rel[:, last].max():6.2f}"
        line += f"  block life med {np.median(rel[:, last] - rel[:, 0]):6.2f} us"
        print(line)
        if last > 1:
            d = np.diff(rel, axis=1)
            print("           phase medians (us): " + "  ".join(f"{np.median(d[:, k]):.2f}" for k in range(d.shape[1])))
    # residency: blocks alive at a few instants, and how the blocks spread over the XCDs / CUs (HW_ID, XCC_ID)
    end = st[:, 1:6].max(axis=1)
    ts = np.linspace(t0, end.max(), 12)[1:-1]
    print("  blocks alive at t (us): " + "  ".join(f"{(t - t0) / 100.0:.1f}:{int(((st[:, 0] <= t) & (end > t)).sum())}" for t in ts))
    xcc, cu, sh, se = (hw >> 32) & 0xF, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    per_cu = np.bincount(cuid)
    per_cu = per_cu[per_cu > 0]
    print(f"  distinct CUs {len(per_cu)}  blocks per CU min/med/max {per_cu.min()}/{int(np.median(per_cu))}/{per_cu.max()}  per XCC {np.bincount(xcc).tolist()}")
    alive_cu = []
    for t in ts[::3]:
        a = (st[:, 0] <= t) & (end > t)
        c = np.bincount(cuid[a], minlength=cuid.max() + 1)
        alive_cu.append(f"{(t - t0) / 100.0:.1f}: max {c.max()} mean {c[np.unique(cuid)].mean():.2f}")
    print("  blocks alive per CU at t: " + " | ".join(alive_cu))


def cast_phases(st):
    """cast blocks: slots 0 start, 2 flags known (two barriers), 3 pose + yaw quaternion, 4 cell walk done, 1 end"""
    st = st.cpu().numpy().astype(np.int64)
    st[:, 7] &= 0xFF
    r = st[(st[:, 0] != 0) & (st[:, 7] == 1)]
    seq = r[:, [0, 2, 3, 4, 1]] / 100.0
    d = np.diff(seq, axis=1)
    print("           cast phase medians (us): bookkeeping+barriers %.2f  pose+yaw %.2f  ray setup+cell walk %.2f  hit/obs stores %.2f"
          % tuple(np.median(d, axis=0)))
    if os.environ.get("MDP_RC_STAGE", "0") != "0" and (r[:, 5] != 0).any():
        # staged walk: 5 = every ray's list is in the table (first barrier), 6 = records staged (second barrier)
        fine = np.diff(r[:, [3, 5, 6, 4]] / 100.0, axis=1)
        print("           staged walk (us): cell + patch lookups, table %.2f  -> cooperative record fetch %.2f  -> walk out of LDS %.2f"
              % tuple(np.median(fine, axis=0)))
    elif (r[:, 5] != 0).any():   # finer stamps of the bookkeeping lane: 5 = its inputs arrived, 6 = yaw quaternion done
        b = r[r[:, 5] != 0]
        fine = np.diff(b[:, [0, 5, 6, 2]] / 100.0, axis=1)
        print("           bookkeeping lane (us): block start -> inputs arrived %.2f  -> yaw quaternion %.2f  -> stores + barrier passed %.2f"
              % tuple(np.median(fine, axis=0)))
        q = b[:, 0] > np.percentile(b[:, 0], 75)
        print("           same, last-quarter blocks: " + "  ".join("%.2f" % v for v in np.median(fine[q], axis=0)))
    early = r[:, 0] < np.percentile(r[:, 0], 25)
    print("           first-quarter blocks: " + "  ".join("%.2f" % v for v in np.median(d[early], axis=0)),
          "| last-quarter blocks: " + "  ".join("%.2f" % v for v in np.median(d[r[:, 0] > np.percentile(r[:, 0], 75)], axis=0)))


def main():
    n_envs = int(os.environ.get("STAMPS_ENVS", "4096"))     # stamp rows exist for 8192 blocks: up to ~7800 envs
    events = os.environ.get("STAMPS_EVENTS", "0") == "1"    # ... with curriculum and event terms folded into it
    env, rec = make_velocity_env("anymal_c_rough", num_envs=n_envs, device="cuda:0", steps=8, seed=5, inject_random=False,
                                 events=events)
    env.observation_manager.persistent_output = True
    reset = os.environ.get("STAMPS_RESET", "0") == "1"      # the launch with the reset branch in it
    if reset:
        env.auto_reset = "masked"
    acts = rec.actions.to("cuda:0")
    for k in range(8):
        env.step(acts[k % acts.shape[0]])
    env.scene.sensors["height_scanner"].cfg.update_period = 0.0
    torch.cuda.synchronize()
    full = torch.zeros((8192 * 8 + 4096 * 187,), dtype=torch.int64, device="cuda:0")   # stamps | per-ray walk words
    buf = full[: 8192 * 8].view(8192, 8)
    for setter in ("mdp_debug_stamps_programs", "mdp_debug_stamps_actuator_nets"):
        fn = getattr(L.lib, setter)
        fn.argtypes = [C.c_void_p]
        fn.restype = C.c_int
        assert fn(buf.data_ptr()) == 0
    # (the reset variant is followed by the conditional actuator launch: its stamps must not land in the same rows)
    assert L.lib.mdp_debug_stamps_actuator_nets(None) == 0
    for rep in range(3):
        buf.zero_()
        torch.cuda.synchronize()
        env._post_physics_single_launch(reset=reset, events=env._foldable_events() if events else None)
        torch.cuda.synchronize()
        if rep == 2:
            show("k_post_physics", buf, {2: "managers", 1: "cast"}, {2: 5, 1: 1})
            if reset:
                st = buf.cpu().numpy().astype(np.int64)
                st[:, 7] &= 0xFF
                m = st[(st[:, 0] != 0) & (st[:, 7] == 2)]
                last = m[np.argmax(m[:, 5])]
                t0 = st[st[:, 0] != 0][:, 0].min()
                d23 = (m[:, 3] - m[:, 2]) / 100.0
                dones = env.termination_manager._dones_buf.cpu().numpy().astype(np.int64)
                E = -(-len(dones) // len(m))                                                  # envs per manager block
                per_block = np.add.reduceat(dones, np.arange(0, len(dones), E))[: len(m)]
                ends = (m[:, 5] - t0) / 100.0
                print("           manager block ends (us): without done envs med %.2f max %.2f | with done envs med %.2f max %.2f"
                      % (np.median(ends[per_block == 0]), ends[per_block == 0].max(),
                         np.median(ends[per_block > 0]) if (per_block > 0).any() else 0.0,
                         ends[per_block > 0].max() if (per_block > 0).any() else 0.0))
                print("           rewards+reset phase: blocks without done envs med %.2f max %.2f | with done envs (%d blocks) med %.2f max %.2f"
                      % (np.median(d23[per_block == 0]), d23[per_block == 0].max(), int((per_block > 0).sum()),
                         np.median(d23[per_block > 0]) if (per_block > 0).any() else 0.0,
                         d23[per_block > 0].max() if (per_block > 0).any() else 0.0))
                print("           manager block that ends last: stamps (us) " +
                      "  ".join("%.2f" % ((last[k] - t0) / 100.0) for k in (0, 1, 2, 3, 4, 6, 5)) + "   (6 = before the log reduction)")
            cast_phases(buf)
            it = full[8192 * 8:].cpu().numpy().reshape(4096, 187)[: min(n_envs, 4096)]
            print("           rays answered by the lattice patch: %.3f" % (it == 0).mean())
            it = np.where(it == 0, 1010, it)
            dis, n, batches = it // 1000, (it % 1000) // 10, it % 10
            print("           rays in disjoint cells %.3f; records per cell mean %.2f; batches per ray mean %.2f  hist %s"
                  % (dis.mean(), n.mean(), batches.mean(), np.bincount(batches.ravel(), minlength=8)[:8].tolist()))
            print("           batches per ray, disjoint cells %.2f, other cells %.2f; max over each 64-ray wave: mean %.2f"
                  % (batches[dis == 1].mean(), batches[dis == 0].mean(),
                     np.mean([batches[:, w:w + 64].max(axis=1).mean() for w in (0, 64, 128)])))
    assert L.lib.mdp_debug_stamps_actuator_nets(buf.data_ptr()) == 0
    for rep in range(3):
        buf.zero_()
        torch.cuda.synchronize()
        env.scene.update_and_write(env.physics_dt)
        torch.cuda.synchronize()
        if rep == 2:
            show("k_lstm8_reg + scene update", buf, {1: "update", 2: "lstm"}, {1: 1, 2: 1})
    for rep in range(3):
        buf.zero_()
        torch.cuda.synchronize()
        env.scene["robot"].write_data_to_sim()
        torch.cuda.synchronize()
        if rep == 2:
            show("k_lstm8_reg alone", buf, {2: "lstm"}, {2: 1})


if __name__ == "__main__":
    main()
