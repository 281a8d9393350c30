// Event terms of the reset / interval modes and the terrain curriculum as per-env device routines, shared by the masked
// kernels of events.hip (one launch per term, mdp_event_*) and by the reset-carrying post-physics kernel (programs.hip,
// mdp_post_physics_step_with_reset_events), which runs them for the envs its own termination phase has just flagged.
// Reference arithmetic: LAB/envs/mdp/events.py:610-866, LAB/managers/event_manager.py:199-225,
// LAB/assets/articulation/articulation.py:316-341, 400-420, 472-508, LAB/utils/math.py:252-278, 463-499, 1331,
// TASKS/manager_based/locomotion/velocity/mdp/curriculums.py:26-55, LAB/terrains/terrain_importer.py:307-324.
#pragma once
#include "mdp_device.hpp"
#include "uniforms_device.hpp"
#include "../../include/mdp_hip.h"

namespace mdp {

// math.py:252-278
__device__ __forceinline__ Q4 quat_from_euler_xyz(float roll, float pitch, float yaw) {
    float cy, sy, cr, sr, cp, sp;          // (one argument reduction per angle: sincosf returns sinf's and cosf's values)
    sincosf(yaw * 0.5f, &sy, &cy);
    sincosf(roll * 0.5f, &sr, &cr);
    sincosf(pitch * 0.5f, &sp, &cp);
    return Q4{cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp,
              sy * cr * cp - cy * sr * sp};
}

__device__ __forceinline__ void zero_body_acc(float* body_acc_w, int num_bodies, int e) {
    if (!body_acc_w) return;
    float* p = body_acc_w + (int64_t)e * num_bodies * 6;
    for (int i = 0; i < num_bodies * 6; ++i) p[i] = 0.0f;      // articulation.py:418
}

// events.py:638-680 + articulation.py:316-341, 400-420 for env e (the calling thread owns the env's epoch word), in two
// parts so that a caller with other work for the env can request everything first: the loads ...
struct RootStateIn {
    float d[13];             // default_root_state row
    uint64_t seed, epoch;    // generator words (unused with injected uniforms)
};
__device__ __forceinline__ RootStateIn event_reset_root_state_load(const mdp_event_root_state_t& a, int e) {
    RootStateIn in;
    const float* d = a.default_root_state + (int64_t)e * 13;
#pragma unroll
    for (int k = 0; k < 13; ++k) in.d[k] = d[k];
    in.seed = 0; in.epoch = 0;
    if (!a.injected_u && a.rng_state) { in.seed = a.rng_state[0]; in.epoch = a.rng_state[1 + e]; }
    return in;
}
// ... the draws and everything that does not need the env origin (a caller whose origin row is still in flight does this
// meanwhile) ...
struct RootStateDraw { float s[3]; Q4 q; float v[6]; };
__device__ __forceinline__ RootStateDraw event_reset_root_state_draw(const mdp_event_root_state_t& a, int e, const RootStateIn& in) {
    if (!a.injected_u && a.rng_state) a.rng_state[1 + e] = in.epoch + 1;
    Uniforms u{a.injected_u, 12, in.seed, in.epoch, 0x45525354u, e};
    const float* d = in.d;
    float s[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = u(k) * a.pose_scale[k] + a.pose_lo[k];
    RootStateDraw o;
    o.s[0] = s[0]; o.s[1] = s[1]; o.s[2] = s[2];
    o.q = quat_mul(Q4{d[3], d[4], d[5], d[6]}, quat_from_euler_xyz(s[3], s[4], s[5]));
#pragma unroll
    for (int k = 0; k < 6; ++k) o.v[k] = d[7 + k] + (u(6 + k) * a.vel_scale[k] + a.vel_lo[k]);
    return o;
}
// ... and the position + stores; org = the env's origin (env_origins row as the curriculum left it)
__device__ __forceinline__ void event_reset_root_state_finish(const mdp_event_root_state_t& a, int e, const RootStateIn& in,
                                                              const RootStateDraw& w, V3 org) {
    const float* d = in.d;
    const V3 pos{(d[0] + org.x) + w.s[0], (d[1] + org.y) + w.s[1], (d[2] + org.z) + w.s[2]};
    const Q4 q = w.q;
    float* r = a.root_state_w + (int64_t)e * 13;
    st3(r, pos);
    st4(r + 3, q);
#pragma unroll
    for (int k = 0; k < 6; ++k) r[7 + k] = w.v[k];
    float* t = a.sim_root_transforms + (int64_t)e * 7;         // (x, y, z, qx, qy, qz, qw)
    st3(t, pos);
    t[3] = q.x; t[4] = q.y; t[5] = q.z; t[6] = q.w;
    float* sv = a.sim_root_velocities + (int64_t)e * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) sv[k] = w.v[k];
    zero_body_acc(a.body_acc_w, a.num_bodies, e);
}
__device__ __forceinline__ void event_reset_root_state_apply(const mdp_event_root_state_t& a, int e, const RootStateIn& in,
                                                             V3 org) {
    const RootStateDraw w = event_reset_root_state_draw(a, e, in);
    event_reset_root_state_finish(a, e, in, w, org);
}
__device__ __forceinline__ void event_reset_root_state_env(const mdp_event_root_state_t& a, int e) {
    const RootStateIn in = event_reset_root_state_load(a, e);
    event_reset_root_state_apply(a, e, in, ld3(a.env_origins + 3 * (int64_t)e));
}

// events.py:802-866 + articulation.py:472-508 for joint j of env e; (seed, epoch): the env's generator words as read
// BEFORE any thread of the env advances the epoch (the caller advances it once per env).  Loads and arithmetic apart,
// as above.
struct JointResetIn { float q0, qd0, lo, hi, vl; };
__device__ __forceinline__ JointResetIn event_reset_joint_load(const mdp_event_joints_t& a, int e, int j) {
    const int64_t i = (int64_t)e * a.num_joints + j;
    return JointResetIn{a.default_joint_pos[i], a.default_joint_vel[i], a.soft_joint_pos_limits[2 * i],
                        a.soft_joint_pos_limits[2 * i + 1], a.soft_joint_vel_limits[i]};
}
__device__ __forceinline__ void event_reset_joint_apply(const mdp_event_joints_t& a, int e, int j, uint64_t seed, uint64_t epoch,
                                                        const JointResetIn& in) {
    const int J = a.num_joints;
    Uniforms u{a.injected_u, 2 * J, seed, epoch, 0x454a4e54u, e};
    const int64_t i = (int64_t)e * J + j;
    float sp = u(j) * a.pos_scale + a.pos_lo;
    float sv = u(J + j) * a.vel_scale + a.vel_lo;
    float q = a.mode == 0 ? in.q0 * sp : in.q0 + sp;
    float qd = a.mode == 0 ? in.qd0 * sv : in.qd0 + sv;
    q = tclamp(q, in.lo, in.hi);
    qd = tclamp(qd, -in.vl, in.vl);
    a.joint_pos[i] = q;
    a.joint_vel[i] = qd;
    a.prev_joint_vel[i] = qd;
    a.joint_acc[i] = 0.0f;
    a.sim_dof_positions[(int64_t)e * a.ld_sim + j] = q;
    a.sim_dof_velocities[(int64_t)e * a.ld_sim + j] = qd;
}
__device__ __forceinline__ void event_reset_joint(const mdp_event_joints_t& a, int e, int j, uint64_t seed, uint64_t epoch) {
    event_reset_joint_apply(a, e, j, seed, epoch, event_reset_joint_load(a, e, j));
}

// events.py:610-635 + articulation.py:400-420 for env e, loads and arithmetic apart
struct PushIn { float v[6]; uint64_t seed, epoch; };
__device__ __forceinline__ PushIn event_push_load(const mdp_event_push_t& a, int e) {
    PushIn in;
    const float* r = a.root_state_w + (int64_t)e * 13 + 7;
#pragma unroll
    for (int k = 0; k < 6; ++k) in.v[k] = r[k];
    in.seed = 0; in.epoch = 0;
    if (!a.injected_u && a.rng_state) { in.seed = a.rng_state[0]; in.epoch = a.rng_state[1 + e]; }
    return in;
}
__device__ __forceinline__ void event_push_apply(const mdp_event_push_t& a, int e, const PushIn& in) {
    if (!a.injected_u && a.rng_state) a.rng_state[1 + e] = in.epoch + 1;
    Uniforms u{a.injected_u, 6, in.seed, in.epoch, 0x45505348u, e};
    float* r = a.root_state_w + (int64_t)e * 13 + 7;
    float* sv = a.sim_root_velocities + (int64_t)e * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float v = in.v[k] + (u(k) * a.vel_scale[k] + a.vel_lo[k]);
        r[k] = v;
        sv[k] = v;
    }
    zero_body_acc(a.body_acc_w, a.num_bodies, e);
}
__device__ __forceinline__ void event_push_env(const mdp_event_push_t& a, int e) {
    event_push_apply(a, e, event_push_load(a, e));
}

// event_manager.py:203-225, the timer of env e: true when it fired (and was resampled).  `loaded`: time_left[e] as read
// by the caller (a kernel that requests it ahead of the work in front of the tick)
__device__ __forceinline__ bool event_interval_tick_env(float* __restrict__ time_left, float dt, float lo, float scale,
                                                        const float* __restrict__ inj, uint64_t* rng_state, int e,
                                                        float loaded) {
    float t = loaded - dt;
    const bool fire = t < 1e-6f;
    if (fire) {
        Uniforms u = uniforms_for(inj, 1, rng_state, 0x45495654u, e);
        t = u(0) * scale + lo;
    }
    time_left[e] = t;
    return fire;
}

// curriculums.py:26-55 (terrain_levels_vel) + terrain_importer.py:307-324 (update_env_origins) for env e, in two parts
// like the root-state event: the loads that depend on nothing ...
struct TerrainLevelsIn {
    float px, py, ox, oy, c0, c1;
    int64_t level, type, injected;
    uint64_t seed, epoch;
};
__device__ __forceinline__ TerrainLevelsIn terrain_levels_vel_load(int e, const mdp_curriculum_terrain_t& a) {
    TerrainLevelsIn in;
    const float* r = a.root_state_w + (int64_t)e * a.ld_state;
    const float* org = a.env_origins + 3 * (int64_t)e;
    in.px = r[0]; in.py = r[1]; in.ox = org[0]; in.oy = org[1];
    in.c0 = a.command[3 * e]; in.c1 = a.command[3 * e + 1];
    in.level = a.terrain_levels[e];
    in.type = a.terrain_types[e];
    in.injected = a.injected_levels ? a.injected_levels[e] : 0;
    in.seed = 0; in.epoch = 0;
    if (!a.injected_levels && a.rng_state) { in.seed = a.rng_state[0]; in.epoch = a.rng_state[1 + e]; }
    return in;
}
// ... the decision and the level store; returns the new level and REQUESTS its origin row (*org_out: in flight) ...
__device__ __forceinline__ int64_t terrain_levels_vel_decide(int e, const mdp_curriculum_terrain_t& a, const TerrainLevelsIn& in,
                                                             V3* org_out) {
    float distance = norm2(in.px - in.ox, in.py - in.oy);
    bool move_up = distance > a.half_tile_size;
    bool move_down = distance < (norm2(in.c0, in.c1) * a.max_episode_length_s) * 0.5f;
    move_down = move_down && !move_up;
    int64_t level = in.level + (move_up ? 1 : 0) - (move_down ? 1 : 0);
    if (level >= a.max_terrain_level) {
        // robots that solve the last level are sent to a random one (torch.randint_like(levels, max_level))
        if (a.injected_levels) {
            level = in.injected;
        } else {
            a.rng_state[1 + e] = in.epoch + 1;
            level = (int64_t)(rng4(in.seed, in.epoch, 0x43544c56u, (uint64_t)e).x % (uint32_t)a.max_terrain_level);
        }
    } else if (level < 0) {
        level = 0;
    }
    a.terrain_levels[e] = level;
    *org_out = ld3(a.terrain_origins + ((int64_t)level * a.num_cols + in.type) * 3);
    return level;
}
// ... and the origin store (terrain_importer.py:307-324)
__device__ __forceinline__ void terrain_levels_vel_commit(int e, const mdp_curriculum_terrain_t& a, V3 org) {
    st3(a.env_origins + 3 * (int64_t)e, org);
}
__device__ __forceinline__ int64_t terrain_levels_vel_apply(int e, const mdp_curriculum_terrain_t& a, const TerrainLevelsIn& in,
                                                            V3* org_out) {
    const int64_t level = terrain_levels_vel_decide(e, a, in, org_out);
    terrain_levels_vel_commit(e, a, *org_out);
    return level;
}
__device__ __forceinline__ int64_t terrain_levels_vel_env(int e, const mdp_curriculum_terrain_t& a) {
    const TerrainLevelsIn in = terrain_levels_vel_load(e, a);
    V3 org;
    return terrain_levels_vel_apply(e, a, in, &org);
}

}  // namespace mdp
