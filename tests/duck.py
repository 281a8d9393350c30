"""Duck-typed env over device tensors for the GPU term / manager tests.

Mirrors the env that tests/golden/make_golden.py handed to the REFERENCE term functions (same fields, same
fixture arrays), the way the reference's own manager tests mock the env (test_observation_manager.py:102-104)."""
from types import SimpleNamespace as NS

import numpy as np
import torch

TERM_NAMES = ["time_out", "base_contact", "bad"]


def dev(x, device):
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


class Scene(dict):
    pass


def make_device_env(G, p, device):
    """G: golden dict, p: key prefix such as 'n7_'.  Returns (env, tensors) with every state tensor on `device`."""
    g = lambda k: dev(G[p + "in_" + k], device)
    n = G[p + "in_root_state"].shape[0]
    t = {k: g(k) for k in (
        "root_state", "joint_pos", "joint_vel", "joint_acc", "default_joint_pos", "default_joint_vel",
        "applied_torque", "computed_torque", "action", "prev_action", "command", "env_origins", "body_state",
        "body_acc", "soft_joint_vel_limits", "soft_joint_pos_limits", "contact_hist", "cur_air", "cur_contact",
        "last_air", "last_contact", "ray_hits", "sensor_pos", "episode_length", "time_left", "command_counter",
        "terminated", "time_outs", "term_dones")}
    j, b = t["joint_pos"].shape[1], t["body_state"].shape[1]
    data = NS(root_state_w=t["root_state"], joint_pos=t["joint_pos"], joint_vel=t["joint_vel"],
              joint_acc=t["joint_acc"], default_joint_pos=t["default_joint_pos"],
              default_joint_vel=t["default_joint_vel"], soft_joint_pos_limits=t["soft_joint_pos_limits"],
              soft_joint_vel_limits=t["soft_joint_vel_limits"], applied_torque=t["applied_torque"],
              computed_torque=t["computed_torque"], body_state_w=t["body_state"], body_acc_w=t["body_acc"])
    robot = NS(data=data, num_joints=j, num_bodies=b)
    cdata = NS(net_forces_w_history=t["contact_hist"], current_air_time=t["cur_air"],
               current_contact_time=t["cur_contact"], last_air_time=t["last_air"], last_contact_time=t["last_contact"])
    contact = NS(data=cdata, _data=cdata, num_bodies=b)
    rdata = NS(pos_w=t["sensor_pos"], ray_hits_w=t["ray_hits"])
    rays = NS(data=rdata, _data=rdata, num_rays=t["ray_hits"].shape[1], can_fuse=False)
    scene = Scene(robot=robot, height_scanner=rays, contact_forces=contact)
    scene.sensors = {"height_scanner": rays, "contact_forces": contact}
    scene.env_origins = t["env_origins"]
    cmd_term = NS(time_left=t["time_left"], command_counter=t["command_counter"])
    tm = NS(terminated=t["terminated"], time_outs=t["time_outs"], _term_dones_block=t["term_dones"],
            active_terms=TERM_NAMES, find_terms=lambda keys: TERM_NAMES[:2],
            get_term=lambda name: t["term_dones"][TERM_NAMES.index(name)])
    env = NS(num_envs=n, device=str(device), scene=scene, step_dt=0.02, max_episode_length=1000,
             max_episode_length_s=20.0, episode_length_buf=t["episode_length"],
             action_manager=NS(action=t["action"], prev_action=t["prev_action"]),
             command_manager=NS(get_command=lambda name: t["command"], get_term=lambda name: cmd_term),
             termination_manager=tm)
    return env, t
