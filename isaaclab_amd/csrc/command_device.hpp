// UniformVelocityCommand per-env update as a device routine, shared by mdp_velocity_command_compute (scene.hip) and the
// post-physics kernel (programs.hip).
#pragma once
#include "mdp_device.hpp"
#include "../../include/mdp_hip.h"

namespace mdp {

// command_manager.py:151-187 + velocity_command.py:111-160, one thread per env
__device__ __forceinline__ void velocity_command_env(
    const mdp_velocity_command_cfg_t& cfg, int e, const float* __restrict__ st, int ld, float* __restrict__ cmd,
    float* __restrict__ heading_target, uint8_t* __restrict__ is_heading, uint8_t* __restrict__ is_standing,
    float* __restrict__ time_left, int64_t* __restrict__ counter, float* __restrict__ m_xy,
    float* __restrict__ m_yaw, float dt, uint64_t seed, uint64_t epoch, const float* __restrict__ inj,
    const uint8_t* __restrict__ force, float* defer = nullptr) {
    // defer != nullptr (compute path only): the new command goes to defer[0..2] instead of cmd -- the caller stores it
    // once nobody reads the previous command any more (the post-physics kernel evaluates the command update next to
    // the termination terms, ahead of the rewards, which read the previous command); every other buffer is written here
    const float* r = st + (int64_t)e * ld;
    Q4 q = ld4(r + 3);
    float c0 = cmd[3 * e + 0], c1 = cmd[3 * e + 1], c2 = cmd[3 * e + 2];
    bool resample;

    if (force) {
        // CommandTerm.reset (command_manager.py:120-149): metrics zeroed, counter = 0 then += 1
        resample = force[e] != 0;
        if (resample) {
            m_xy[e] = 0.0f;
            m_yaw[e] = 0.0f;
            counter[e] = 0;
        }
    } else {
        // _update_metrics (velocity_command.py:111-121)
        V3 lin_b = quat_rotate_inverse(q, ld3(r + 7));
        V3 ang_b = quat_rotate_inverse(q, ld3(r + 10));
        m_xy[e] += norm2(c0 - lin_b.x, c1 - lin_b.y) / cfg.max_command_step;
        m_yaw[e] += fabsf(c2 - ang_b.z) / cfg.max_command_step;
        float tl = time_left[e] - dt;
        time_left[e] = tl;
        resample = tl <= 0.0f;
    }
    bool heading_env = is_heading[e] != 0;
    bool standing_env = is_standing[e] != 0;
    if (resample) {
        float u[7];
        if (inj) {
            for (int k = 0; k < 7; ++k) u[k] = inj[7 * (int64_t)e + k];
        } else {
            U4 a = rng4(seed, epoch, 0x434d4431u, (uint64_t)e * 2);
            U4 b = rng4(seed, epoch, 0x434d4431u, (uint64_t)e * 2 + 1);
            u[0] = u01(a.x); u[1] = u01(a.y); u[2] = u01(a.z); u[3] = u01(a.w);
            u[4] = u01(b.x); u[5] = u01(b.y); u[6] = u01(b.z);
        }
        // Tensor.uniform_(a, b) == u * (b - a) + a
        time_left[e] = u[0] * (cfg.resampling_time_hi - cfg.resampling_time_lo) + cfg.resampling_time_lo;
        counter[e] += 1;
        c0 = u[1] * (cfg.lin_vel_x_hi - cfg.lin_vel_x_lo) + cfg.lin_vel_x_lo;
        c1 = u[2] * (cfg.lin_vel_y_hi - cfg.lin_vel_y_lo) + cfg.lin_vel_y_lo;
        c2 = u[3] * (cfg.ang_vel_z_hi - cfg.ang_vel_z_lo) + cfg.ang_vel_z_lo;
        if (cfg.heading_command) {
            heading_target[e] = u[4] * (cfg.heading_hi - cfg.heading_lo) + cfg.heading_lo;
            heading_env = u[5] <= cfg.rel_heading_envs;
            is_heading[e] = heading_env ? 1 : 0;
        }
        standing_env = u[6] <= cfg.rel_standing_envs;
        is_standing[e] = standing_env ? 1 : 0;
    }
    if (force) {
        // CommandTerm.reset only resamples (command_manager.py:144-147); _update_command runs at the next compute
        if (resample) {
            cmd[3 * e + 0] = c0;
            cmd[3 * e + 1] = c1;
            cmd[3 * e + 2] = c2;
        }
        return;
    }
    // _update_command (velocity_command.py:140-160)
    if (cfg.heading_command && heading_env) {
        V3 f = quat_apply(q, V3{1.0f, 0.0f, 0.0f});
        float heading_w = atan2f(f.y, f.x);
        float err = wrap_to_pi(heading_target[e] - heading_w);
        c2 = tclamp(cfg.heading_control_stiffness * err, cfg.ang_vel_z_lo, cfg.ang_vel_z_hi);
    }
    if (standing_env) {
        c0 = 0.0f;
        c1 = 0.0f;
        c2 = 0.0f;
    }
    if (defer) {
        defer[0] = c0; defer[1] = c1; defer[2] = c2;
        return;
    }
    cmd[3 * e + 0] = c0;
    cmd[3 * e + 1] = c1;
    cmd[3 * e + 2] = c2;
}

}  // namespace mdp
