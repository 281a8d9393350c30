"""ctypes binding of ``libmdp_hip.so`` (the C ABI declared in ``include/mdp_hip.h``).

The product path never falls back to torch/CPU arithmetic: if the shared library is missing the
import fails loudly, and every wrapper raises if a tensor does not live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MDP_HIP_LIB: diagnostic builds only (e.g. the in-kernel stamp build of tools/kernel_stamps.py)
LIB_PATH = os.environ.get("MDP_HIP_LIB") or os.path.join(_HERE, "_lib", "libmdp_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  isaaclab_amd has no CPU / torch fallback."
    )

lib = C.CDLL(LIB_PATH)

MDP_OK = 0
MDP_ERR_INVALID = -1
MDP_ERR_RUNTIME = -2
MDP_ERR_UNSUPPORTED = -3

lib.mdp_last_error.restype = C.c_char_p
lib.mdp_version.restype = C.c_int
lib.mdp_net_packed_size.restype = C.c_int64


class MdpTerm(C.Structure):
    """Mirror of ``mdp_term_t``."""

    _fields_ = [
        ("op", C.c_int32), ("out_offset", C.c_int32), ("dim", C.c_int32),
        ("n_idx", C.c_int32), ("idx_offset", C.c_int32), ("n_idx2", C.c_int32), ("idx2_offset", C.c_int32),
        ("i0", C.c_int32), ("i1", C.c_int32), ("i2", C.c_int32), ("i3", C.c_int32),
        ("l0", C.c_int64),
        ("f0", C.c_float), ("f1", C.c_float), ("f2", C.c_float), ("f3", C.c_float), ("f4", C.c_float), ("f5", C.c_float),
        ("p0", C.c_void_p), ("p1", C.c_void_p), ("p2", C.c_void_p), ("p3", C.c_void_p), ("p4", C.c_void_p), ("p5", C.c_void_p),
        ("noise_kind", C.c_int32), ("noise_op", C.c_int32), ("n0", C.c_float), ("n1", C.c_float),
        ("noise_u", C.c_void_p),
        ("has_clip", C.c_int32), ("clip_lo", C.c_float), ("clip_hi", C.c_float),
        ("has_scale", C.c_int32), ("scale", C.c_float), ("scale_vec", C.c_void_p),
        ("weight", C.c_float),
    ]


class MdpRaycasterDesc(C.Structure):
    _fields_ = [
        ("mesh", C.c_void_p), ("num_rays", C.c_int32), ("attach_yaw_only", C.c_int32), ("max_dist", C.c_float),
        ("ld_pose", C.c_int32), ("root_pose", C.c_void_p), ("drift", C.c_void_p),
        ("ray_starts_local", C.c_void_p), ("ray_dirs_local", C.c_void_p),
        ("pos_w", C.c_void_p), ("quat_w", C.c_void_p), ("ray_hits_w", C.c_void_p),
        ("timestamp", C.c_void_p), ("timestamp_last_update", C.c_void_p), ("is_outdated", C.c_void_p),
    ]


class MdpObsNormalize(C.Structure):
    """Mirror of ``mdp_obs_normalize_t``."""

    _fields_ = [("mean32", C.c_void_p), ("denom", C.c_void_p), ("out", C.c_void_p), ("ld_out", C.c_int32),
                ("clip", C.c_float)]


class MdpHeightScanObs(C.Structure):
    """Mirror of ``mdp_height_scan_obs_t``."""

    _fields_ = [
        ("obs", C.c_void_p), ("ld_obs", C.c_int32), ("col_offset", C.c_int32), ("offset", C.c_float),
        ("pad0", C.c_int32), ("rng", C.c_void_p), ("post", MdpTerm), ("normalize", MdpObsNormalize),
    ]


class MdpVelocityCommandCfg(C.Structure):
    _fields_ = [
        ("resampling_time_lo", C.c_float), ("resampling_time_hi", C.c_float),
        ("lin_vel_x_lo", C.c_float), ("lin_vel_x_hi", C.c_float),
        ("lin_vel_y_lo", C.c_float), ("lin_vel_y_hi", C.c_float),
        ("ang_vel_z_lo", C.c_float), ("ang_vel_z_hi", C.c_float),
        ("heading_lo", C.c_float), ("heading_hi", C.c_float),
        ("rel_standing_envs", C.c_float), ("rel_heading_envs", C.c_float),
        ("heading_control_stiffness", C.c_float), ("heading_command", C.c_int32),
        ("max_command_step", C.c_float),
    ]


class MdpTerminationRewardArgs(C.Structure):
    """Mirror of ``mdp_termination_reward_args_t``."""

    _fields_ = [
        ("termination", C.c_void_p), ("reward", C.c_void_p), ("dt", C.c_float), ("pad0", C.c_int32),
        ("episode_length_buf_inc", C.c_void_p), ("term_dones", C.c_void_p), ("truncated", C.c_void_p),
        ("terminated", C.c_void_p), ("dones", C.c_void_p), ("reward_buf", C.c_void_p),
        ("episode_sums", C.c_void_p), ("step_reward", C.c_void_p),
    ]


class MdpVelocityCommandArgs(C.Structure):
    """Mirror of ``mdp_velocity_command_args_t``."""

    _fields_ = [
        ("cfg", MdpVelocityCommandCfg), ("ld_state", C.c_int32), ("root_state", C.c_void_p),
        ("vel_command_b", C.c_void_p), ("heading_target", C.c_void_p), ("is_heading_env", C.c_void_p),
        ("is_standing_env", C.c_void_p), ("time_left", C.c_void_p), ("command_counter", C.c_void_p),
        ("metric_error_vel_xy", C.c_void_p), ("metric_error_vel_yaw", C.c_void_p), ("dt", C.c_float),
        ("pad0", C.c_int32), ("rng_state", C.c_void_p), ("injected_u", C.c_void_p),
    ]


class MdpObsArgs(C.Structure):
    """Mirror of ``mdp_obs_args_t``."""

    _fields_ = [("program", C.c_void_p), ("obs", C.c_void_p), ("ld_obs", C.c_int32), ("pad0", C.c_int32),
                ("normalize", MdpObsNormalize)]


class MdpRaycasterUpdateArgs(C.Structure):
    """Mirror of ``mdp_raycaster_update_args_t``."""

    _fields_ = [
        ("mesh", C.c_void_p), ("num_rays", C.c_int32), ("ld_pose", C.c_int32), ("pose_xyzw", C.c_int32),
        ("attach_yaw_only", C.c_int32), ("pattern_vertical_down", C.c_int32), ("pad0", C.c_int32),
        ("root_pose", C.c_void_p), ("drift", C.c_void_p), ("ray_starts_local", C.c_void_p),
        ("ray_dirs_local", C.c_void_p), ("max_dist", C.c_float), ("tick_dt", C.c_float),
        ("update_period", C.c_float), ("pad1", C.c_float), ("pos_w", C.c_void_p), ("quat_w", C.c_void_p),
        ("ray_hits_w", C.c_void_p), ("timestamp", C.c_void_p), ("timestamp_last_update", C.c_void_p),
        ("is_outdated", C.c_void_p), ("scan", C.c_void_p), ("pose_ws", C.c_void_p),
    ]


class MdpJointAction(C.Structure):
    """Mirror of ``mdp_joint_action_t``."""

    _fields_ = [
        ("actions_in", C.c_void_p), ("ld_in", C.c_int32), ("pad0", C.c_int32), ("action", C.c_void_p),
        ("prev_action", C.c_void_p), ("processed", C.c_void_p), ("scale_vec", C.c_void_p), ("scale", C.c_float),
        ("pad1", C.c_int32), ("offset", C.c_void_p), ("clip", C.c_void_p),
    ]


class MdpStatsSource(C.Structure):
    """Mirror of ``mdp_stats_source_t``."""

    _fields_ = [("values", C.c_void_p), ("elem_kind", C.c_int32), ("rows", C.c_int32), ("masked", C.c_int32), ("_pad", C.c_int32)]


class MdpFill(C.Structure):
    """Mirror of ``mdp_fill_t``."""

    _fields_ = [
        ("ptr", C.c_void_p), ("elem_bytes", C.c_int32), ("count", C.c_int32), ("env_stride", C.c_int64),
        ("elem_stride", C.c_int64), ("value_bits", C.c_uint64),
    ]


class MdpPostResetArgs(C.Structure):
    """Mirror of ``mdp_post_reset_args_t``."""

    _fields_ = [("fills", C.c_void_p), ("num_fills", C.c_int32), ("max_episode_length_s", C.c_float),
                ("log_sums", C.c_void_p), ("log", C.c_void_p)]


class MdpPostEvents(C.Structure):
    """Mirror of ``mdp_post_events_t``."""

    _fields_ = [("curriculum", C.c_void_p), ("reset_root_state", C.c_void_p), ("reset_joints", C.c_void_p),
                ("reset_mark", C.c_void_p), ("push", C.c_void_p), ("push_time_left", C.c_void_p),
                ("push_lo", C.c_float), ("push_scale", C.c_float), ("push_tick_u", C.c_void_p),
                ("push_tick_rng", C.c_void_p), ("push_trigger", C.c_void_p), ("interval_dt", C.c_float),
                ("pad0", C.c_int32)]


class MdpEventRootState(C.Structure):
    """Mirror of ``mdp_event_root_state_t``."""

    _fields_ = [("env_mask", C.c_void_p), ("default_root_state", C.c_void_p), ("env_origins", C.c_void_p),
                ("pose_lo", C.c_float * 6), ("pose_scale", C.c_float * 6), ("vel_lo", C.c_float * 6),
                ("vel_scale", C.c_float * 6), ("injected_u", C.c_void_p), ("rng_state", C.c_void_p),
                ("root_state_w", C.c_void_p), ("sim_root_transforms", C.c_void_p), ("sim_root_velocities", C.c_void_p),
                ("body_acc_w", C.c_void_p), ("num_bodies", C.c_int32), ("pad0", C.c_int32)]


class MdpEventJoints(C.Structure):
    """Mirror of ``mdp_event_joints_t``."""

    _fields_ = [("env_mask", C.c_void_p), ("num_joints", C.c_int32), ("mode", C.c_int32),
                ("default_joint_pos", C.c_void_p), ("default_joint_vel", C.c_void_p),
                ("pos_lo", C.c_float), ("pos_scale", C.c_float), ("vel_lo", C.c_float), ("vel_scale", C.c_float),
                ("soft_joint_pos_limits", C.c_void_p), ("soft_joint_vel_limits", C.c_void_p),
                ("injected_u", C.c_void_p), ("rng_state", C.c_void_p),
                ("joint_pos", C.c_void_p), ("joint_vel", C.c_void_p), ("prev_joint_vel", C.c_void_p),
                ("joint_acc", C.c_void_p), ("sim_dof_positions", C.c_void_p), ("sim_dof_velocities", C.c_void_p),
                ("ld_sim", C.c_int32), ("pad0", C.c_int32)]


class MdpEventPush(C.Structure):
    """Mirror of ``mdp_event_push_t``."""

    _fields_ = [("env_mask", C.c_void_p), ("vel_lo", C.c_float * 6), ("vel_scale", C.c_float * 6),
                ("injected_u", C.c_void_p), ("rng_state", C.c_void_p), ("root_state_w", C.c_void_p),
                ("sim_root_velocities", C.c_void_p), ("body_acc_w", C.c_void_p), ("num_bodies", C.c_int32),
                ("pad0", C.c_int32)]


class MdpCurriculumTerrain(C.Structure):
    """Mirror of ``mdp_curriculum_terrain_t``."""

    _fields_ = [("env_mask", C.c_void_p), ("root_state_w", C.c_void_p), ("ld_state", C.c_int32),
                ("max_terrain_level", C.c_int32), ("command", C.c_void_p), ("half_tile_size", C.c_float),
                ("max_episode_length_s", C.c_float), ("terrain_levels", C.c_void_p), ("terrain_types", C.c_void_p),
                ("terrain_origins", C.c_void_p), ("num_cols", C.c_int32), ("pad0", C.c_int32),
                ("env_origins", C.c_void_p), ("injected_levels", C.c_void_p), ("rng_state", C.c_void_p),
                ("mean_level", C.c_void_p), ("mean_ws", C.c_void_p)]


class MdpEventResetMark(C.Structure):
    """Mirror of ``mdp_event_reset_mark_t``."""

    _fields_ = [("num_terms", C.c_int32), ("step_count", C.c_int32), ("last_triggered_step", C.c_void_p * 8),
                ("triggered_once", C.c_void_p * 8)]


class MdpImuArgs(C.Structure):
    """Mirror of ``mdp_imu_args_t``."""

    _fields_ = [("is_outdated", C.c_void_p), ("timestamp", C.c_void_p), ("timestamp_last_update", C.c_void_p),
                ("transforms", C.c_void_p), ("velocities", C.c_void_p), ("coms", C.c_void_p), ("ld_coms", C.c_int32),
                ("dt", C.c_float), ("offset_pos_b", C.c_void_p), ("offset_quat_b", C.c_void_p),
                ("gravity_bias_w", C.c_void_p), ("prev_lin_vel_w", C.c_void_p), ("prev_ang_vel_w", C.c_void_p),
                ("pos_w", C.c_void_p), ("quat_w", C.c_void_p), ("lin_vel_b", C.c_void_p), ("ang_vel_b", C.c_void_p),
                ("lin_acc_b", C.c_void_p), ("ang_acc_b", C.c_void_p)]


class MdpFrameTransformerArgs(C.Structure):
    """Mirror of ``mdp_frame_transformer_args_t``."""

    _fields_ = [(n, C.c_void_p) for n in (
        "gate", "is_outdated", "timestamp", "timestamp_last_update", "transforms", "per_env_indices",
        "source_frame_body_ids", "target_frame_body_ids", "duplicate_frame_indices", "source_offset_pos",
        "source_offset_quat", "target_offset_pos", "target_offset_quat", "source_pos_w", "source_quat_w", "target_pos_w",
        "target_quat_w", "target_pos_source", "target_quat_source")]


class MdpRaycasterCameraArgs(C.Structure):
    """Mirror of ``mdp_raycaster_camera_args_t``."""

    _fields_ = [("mesh", C.c_void_p), ("is_outdated", C.c_void_p), ("timestamp", C.c_void_p),
                ("timestamp_last_update", C.c_void_p), ("view_pose", C.c_void_p), ("ld_pose", C.c_int32),
                ("pose_xyzw", C.c_int32), ("offset_pos", C.c_void_p), ("offset_quat", C.c_void_p),
                ("num_rays", C.c_int32), ("depth_clipping", C.c_int32), ("max_distance", C.c_float), ("pad0", C.c_int32),
                ("ray_starts", C.c_void_p), ("ray_directions", C.c_void_p), ("pos_w", C.c_void_p),
                ("quat_w_world", C.c_void_p), ("ray_hits_w", C.c_void_p), ("distance_to_image_plane", C.c_void_p),
                ("distance_to_camera", C.c_void_p), ("normals", C.c_void_p), ("frame", C.c_void_p)]


class MdpNetDesc(C.Structure):
    _fields_ = [
        ("lstm_layers", C.c_int32), ("lstm_hidden", C.c_int32), ("lstm_input", C.c_int32),
        ("mlp_layers", C.c_int32), ("mlp_dims", C.c_int32 * 9), ("mlp_act", C.c_int32 * 8),
    ]


class MdpInhandCfg(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("num_dofs", C.c_int32), ("num_actuated", C.c_int32),
        ("num_fingertips", C.c_int32), ("num_bodies", C.c_int32),
        ("act_moving_average", C.c_float), ("vel_obs_scale", C.c_float),
        ("dist_reward_scale", C.c_float), ("rot_reward_scale", C.c_float), ("rot_eps", C.c_float),
        ("action_penalty_scale", C.c_float), ("success_tolerance", C.c_float), ("reach_goal_bonus", C.c_float),
        ("fall_dist", C.c_float), ("fall_penalty", C.c_float), ("av_factor", C.c_float),
        ("max_consecutive_success", C.c_int32), ("max_episode_length", C.c_int64),
    ]


class MdpInhandGoalReset(C.Structure):
    """Mirror of ``mdp_inhand_goal_reset_t``."""

    _fields_ = [("env_mask", C.c_void_p), ("injected_u", C.c_void_p), ("ld_injected", C.c_int32), ("pad0", C.c_int32),
                ("rng_state", C.c_void_p), ("x_unit", C.c_void_p), ("y_unit", C.c_void_p), ("goal_rot", C.c_void_p),
                ("reset_goal_buf", C.c_void_p)]


class MdpInhandObsArgs(C.Structure):
    """Mirror of ``mdp_inhand_obs_args_t``."""

    _fields_ = [(n, C.c_void_p) for n in (
        "joint_pos", "joint_vel", "lower", "upper", "object_root_state", "env_origins", "in_hand_pos", "goal_rot",
        "body_state_w", "link_transforms", "link_velocities", "link_incoming_joint_force", "finger_bodies", "actions")] + [
        ("action_dim", C.c_int32), ("force_torque_obs_scale", C.c_float),
        ("policy_kind", C.c_int32), ("ld_policy", C.c_int32), ("policy", C.c_void_p),
        ("critic_kind", C.c_int32), ("ld_critic", C.c_int32), ("critic", C.c_void_p),
        ("partials", C.c_void_p), ("consecutive_successes", C.c_void_p), ("reset", C.c_void_p)]


class MdpInhandReset(C.Structure):
    """Mirror of ``mdp_inhand_reset_t``."""

    _fields_ = [("goal", MdpInhandGoalReset),
                ("reset_position_noise", C.c_float), ("reset_dof_pos_noise", C.c_float),
                ("reset_dof_vel_noise", C.c_float), ("pad0", C.c_float),
                ("object_default_root_state", C.c_void_p), ("env_origins", C.c_void_p),
                ("object_root_state_w", C.c_void_p), ("sim_object_transforms", C.c_void_p),
                ("sim_object_velocities", C.c_void_p),
                ("default_joint_pos", C.c_void_p), ("default_joint_vel", C.c_void_p), ("lower", C.c_void_p),
                ("upper", C.c_void_p),
                ("prev_targets", C.c_void_p), ("cur_targets", C.c_void_p), ("hand_dof_targets", C.c_void_p),
                ("joint_pos_target", C.c_void_p), ("ld_target", C.c_int32), ("ld_sim", C.c_int32),
                ("joint_pos", C.c_void_p), ("joint_vel", C.c_void_p), ("prev_joint_vel", C.c_void_p),
                ("joint_acc", C.c_void_p), ("sim_dof_positions", C.c_void_p), ("sim_dof_velocities", C.c_void_p),
                ("successes", C.c_void_p), ("episode_length_buf", C.c_void_p)]


class MdpSceneUpdate(C.Structure):
    """Mirror of ``mdp_scene_update_t``."""

    _fields_ = [
        ("num_envs", C.c_int32), ("num_joints", C.c_int32), ("num_contact_bodies", C.c_int32), ("history_len", C.c_int32),
        ("root_transforms", C.c_void_p), ("root_velocities", C.c_void_p), ("root_state_w", C.c_void_p),
        ("joint_pos_src", C.c_void_p), ("joint_vel_src", C.c_void_p), ("ld_joint_src", C.c_int32),
        ("acc_elapsed", C.c_float),
        ("joint_pos", C.c_void_p), ("joint_vel", C.c_void_p), ("prev_joint_vel", C.c_void_p), ("joint_acc", C.c_void_p),
        ("net_contact_forces", C.c_void_p), ("net_forces_w", C.c_void_p), ("net_forces_w_history", C.c_void_p),
        ("last_air_time", C.c_void_p), ("current_air_time", C.c_void_p), ("last_contact_time", C.c_void_p),
        ("current_contact_time", C.c_void_p), ("contact_timestamp", C.c_void_p),
        ("contact_timestamp_last_update", C.c_void_p), ("contact_is_outdated", C.c_void_p),
        ("contact_update_period", C.c_float), ("force_threshold", C.c_float), ("contact_eager", C.c_int32),
        ("num_ticks", C.c_int32),
        ("tick_timestamp", C.c_void_p * 4), ("tick_timestamp_last_update", C.c_void_p * 4),
        ("tick_is_outdated", C.c_void_p * 4), ("tick_update_period", C.c_float * 4),
    ]


def _parse_defines(path: str, prefix: str) -> dict[str, int]:
    out = {}
    with open(path) as f:
        for line in f:
            m = re.match(r"#define\s+(" + prefix + r"\w+)\s+(-?\d+)", line)
            if m:
                out[m.group(1)] = int(m.group(2))
    return out


OPS = _parse_defines(os.path.join(_HERE, "csrc", "term_ops.h"), "MDP_")


def _parse_enum(path: str, prefix: str) -> dict[str, int]:
    out = {}
    with open(path) as f:
        for m in re.finditer(r"(" + prefix + r"\w+)\s*=\s*(\d+)", f.read()):
            out[m.group(1)] = int(m.group(2))
    return out


_HDR = os.path.join(os.path.dirname(_HERE), "include", "mdp_hip.h")
MATH_OPS = _parse_enum(_HDR, "MDP_MATH_")
ACT_KINDS = _parse_enum(_HDR, "MDP_ACT_")
PROG_OBS, PROG_REWARD, PROG_TERMINATION = 0, 1, 2
with open(_HDR) as _f:
    MDP_STATS_MAX_SOURCES = int(re.search(r"#define\s+MDP_STATS_MAX_SOURCES\s+(\d+)", _f.read()).group(1))


def exported_symbols() -> list[str]:
    """Every function name declared in include/mdp_hip.h (used by the ABI test)."""
    names = []
    with open(_HDR) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for m in re.finditer(r"\b(mdp_\w+)\s*\(", text):
        names.append(m.group(1))
    return sorted(set(names))


_CTYPE = {
    "int": C.c_int, "int32_t": C.c_int32, "int64_t": C.c_int64, "uint64_t": C.c_uint64,
    "float": C.c_float, "double": C.c_double, "unsigned": C.c_uint,
}


def _bind_all():
    """Set argtypes/restype of every prototype in include/mdp_hip.h (so python floats/ints convert)."""
    with open(_HDR) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for m in re.finditer(r"\b(int64_t|int|const char\s*\*)\s+(mdp_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        fn = getattr(lib, name)  # AttributeError here == symbol missing from the .so
        argtypes = []
        for a in args.split(","):
            a = " ".join(a.split())
            if not a or a == "void":
                continue
            if "*" in a:
                argtypes.append(C.c_void_p)
            else:
                ty = a.rsplit(" ", 1)[0].replace("const", "").strip()
                argtypes.append(_CTYPE[ty])
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if ret == "int64_t" else (C.c_char_p if "char" in ret else C.c_int)


_bind_all()


def last_error() -> str:
    return (lib.mdp_last_error() or b"").decode()


def check(rc: int):
    if rc == MDP_OK:
        return
    msg = last_error()
    if rc == MDP_ERR_INVALID:
        raise ValueError(msg)
    if rc == MDP_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


def require_gpu(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t)}")
    if t.device.type != "cuda":
        raise RuntimeError(
            f"{name} lives on '{t.device}': isaaclab_amd only computes on a HIP device (MI355X); "
            "there is no CPU fallback."
        )
    return t


def ptr(t: torch.Tensor | None):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    require_gpu(t)
    return t.data_ptr()


def f32(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    require_gpu(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    return t


def rows(t: torch.Tensor, name: str = "tensor") -> tuple[torch.Tensor, int]:
    """Return (tensor, row stride) for a 2-D fp32 tensor whose last dim is dense; copies otherwise."""
    t = f32(t, name)
    if t.dim() != 2:
        raise ValueError(f"{name} must be 2-D, got shape {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()  # broadcast / expanded rows
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    return t, int(ld)


def stream(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def device_index(device) -> int:
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"isaaclab_amd needs a HIP device, got '{device}'")
    return d.index if d.index is not None else torch.cuda.current_device()
