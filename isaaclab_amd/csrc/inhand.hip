// Shadow-hand in-hand cube reorientation (direct workflow).
// Reference: TASKS/direct/inhand_manipulation/inhand_manipulation_env.py:97-117,179-199,262-433
#include "mdp_device.hpp"
#include "mdp_host.hpp"
#include "scene_update_device.hpp"

using namespace mdp;

namespace {

// :353-355 scale, :358-360 unscale
MDP_DEV float ih_scale(float x, float lower, float upper) { return 0.5f * (x + 1.0f) * (upper - lower) + lower; }
MDP_DEV float ih_unscale(float x, float lower, float upper) { return ((2.0f * x - upper) - lower) / (upper - lower); }

// :370-374 rotation_distance
MDP_DEV float rotation_distance(Q4 obj, Q4 target) {
    Q4 d = quat_mul(obj, quat_conjugate(target));
    float n = norm3(V3{d.x, d.y, d.z});
    return 2.0f * asinf(tclamp_max(n, 1.0f));
}

// :97-117, one thread per (env, actuated joint); block `b` of `nb`
__device__ __forceinline__ void apply_action_blocks(const mdp_inhand_cfg_t& cfg, const float* __restrict__ actions,
                                                    const int32_t* __restrict__ act_idx, const float* __restrict__ lower,
                                                    const float* __restrict__ upper, float* __restrict__ cur,
                                                    float* __restrict__ prev, float* __restrict__ target, int ld_target,
                                                    float* __restrict__ actions_copy, unsigned b, unsigned nb) {
    int64_t total = (int64_t)cfg.num_envs * cfg.num_actuated;
    for (int64_t i = b * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)nb * blockDim.x) {
        int64_t e = i / cfg.num_actuated;
        int a = (int)(i - e * cfg.num_actuated);
        int j = act_idx[a];
        int64_t o = e * cfg.num_dofs + j;
        float lo = lower[o], hi = upper[o];
        const float act = actions[i];
        if (actions_copy) actions_copy[i] = act;     // `self.actions = actions.clone()` (:94-95) rides along
        float t = ih_scale(act, lo, hi);
        t = cfg.act_moving_average * t + (1.0f - cfg.act_moving_average) * prev[o];
        t = tmax(tmin(t, hi), lo);  // math.py:64-78 saturate
        cur[o] = t;
        prev[o] = t;
        // set_joint_position_target(cur_targets[:, actuated], joint_ids=actuated) (:115-117, articulation.py:807-830)
        if (target) target[e * ld_target + j] = t;
    }
}

__global__ void k_apply_action(mdp_inhand_cfg_t cfg, const float* __restrict__ actions,
                               const int32_t* __restrict__ act_idx, const float* __restrict__ lower,
                               const float* __restrict__ upper, float* __restrict__ cur, float* __restrict__ prev,
                               float* __restrict__ target, int ld_target, float* __restrict__ actions_copy) {
    apply_action_blocks(cfg, actions, act_idx, lower, upper, cur, prev, target, ld_target, actions_copy, blockIdx.x, gridDim.x);
}

// The same with the scene update of the PREVIOUS sub-step in front of the grid (`hand.data.update(dt)` after sim.step,
// direct_rl_env.py:330-339): the targets of this sub-step do not depend on it, so both go in one launch
__global__ void __launch_bounds__(256) k_apply_action_upd(mdp_inhand_cfg_t cfg, const float* __restrict__ actions,
                                                          const int32_t* __restrict__ act_idx,
                                                          const float* __restrict__ lower, const float* __restrict__ upper,
                                                          float* __restrict__ cur, float* __restrict__ prev,
                                                          float* __restrict__ target, int ld_target,
                                                          float* __restrict__ actions_copy, mdp_scene_update_t upd, float dt,
                                                          unsigned upd_total, unsigned nb_root, unsigned nb_joint,
                                                          unsigned nb_contact, int upd_envs) {
    const unsigned task_blocks = gridDim.x - upd_total;
    if (blockIdx.x >= task_blocks) {
        scene_update_carried(upd, dt, blockIdx.x - task_blocks, nb_root, nb_joint, nb_contact, upd_envs);
        return;
    }
    apply_action_blocks(cfg, actions, act_idx, lower, upper, cur, prev, target, ld_target, actions_copy, blockIdx.x,
                        task_blocks);
}

// :179-199 for env e; returns the env's reset flag (direct_rl_env.py:346) and leaves the object's root state in row[13]
__device__ __forceinline__ uint8_t dones_env(const mdp_inhand_cfg_t& cfg, int e, const float* obj,
                                             const float* __restrict__ origins, const float* __restrict__ in_hand,
                                             const float* __restrict__ goal_rot, const float* successes,
                                             int64_t* __restrict__ ep_len, uint8_t* __restrict__ out_of_reach,
                                             uint8_t* __restrict__ time_out, int increment, uint8_t* __restrict__ reset_buf,
                                             const float* __restrict__ obj_xf, const float* __restrict__ obj_vel,
                                             float* obj_out, float* row) {
    if (increment) ep_len[e] += 1;          // direct_rl_env.py:343 episode_length_buf += 1, ahead of _get_dones
    if (obj_xf) {
        // RigidObjectData.root_state_w of this env (cat(pose with xyzw -> wxyz, velocity), as mdp_state_pack) assembled
        // here, stored for the kernels behind this one and used from registers
        const float* x = obj_xf + 7 * (int64_t)e;
        const float* w = obj_vel + 6 * (int64_t)e;
        row[0] = x[0]; row[1] = x[1]; row[2] = x[2]; row[3] = x[6]; row[4] = x[3]; row[5] = x[4]; row[6] = x[5];
#pragma unroll
        for (int k = 0; k < 6; ++k) row[7 + k] = w[k];
        float* o = obj_out + 13 * (int64_t)e;
#pragma unroll
        for (int k = 0; k < 13; ++k) o[k] = row[k];
    } else {
        const float* src = obj + 13 * (int64_t)e;
#pragma unroll
        for (int k = 0; k < 7; ++k) row[k] = src[k];      // position + orientation: all this and the rewards read
    }
    const float* r = row;
    V3 pos{r[0] - origins[3 * e], r[1] - origins[3 * e + 1], r[2] - origins[3 * e + 2]};
    V3 d{pos.x - in_hand[3 * e], pos.y - in_hand[3 * e + 1], pos.z - in_hand[3 * e + 2]};
    float goal_dist = norm3(d);
    out_of_reach[e] = goal_dist >= cfg.fall_dist ? 1 : 0;
    int64_t len = ep_len[e];
    bool max_succ = false;
    if (cfg.max_consecutive_success > 0) {
        float rot_dist = rotation_distance(ld4(r + 3), ld4(goal_rot + 4 * (int64_t)e));
        if (fabsf(rot_dist) <= cfg.success_tolerance) { len = 0; ep_len[e] = 0; }
        max_succ = successes[e] >= (float)cfg.max_consecutive_success;
    }
    bool to = len >= cfg.max_episode_length - 1;
    time_out[e] = (to || max_succ) ? 1 : 0;
    const uint8_t reset = (goal_dist >= cfg.fall_dist || to || max_succ) ? 1 : 0;   // direct_rl_env.py:346
    if (reset_buf) reset_buf[e] = reset;
    return reset;
}

__global__ void k_dones(mdp_inhand_cfg_t cfg, const float* obj, const float* __restrict__ origins,
                        const float* __restrict__ in_hand, const float* __restrict__ goal_rot,
                        const float* __restrict__ successes, int64_t* __restrict__ ep_len,
                        uint8_t* __restrict__ out_of_reach, uint8_t* __restrict__ time_out, int increment,
                        uint8_t* __restrict__ reset_buf, const float* __restrict__ obj_xf,
                        const float* __restrict__ obj_vel, float* obj_out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cfg.num_envs) return;
    float row[13];
    dones_env(cfg, e, obj, origins, in_hand, goal_rot, successes, ep_len, out_of_reach, time_out, increment, reset_buf,
              obj_xf, obj_vel, obj_out, row);
}

// :377-433 per-env part for env e (r = the object's root state: position + orientation are read); cross-env sums are
// accumulated per block, then atomically in fp64
__device__ __forceinline__ void rewards_env(const mdp_inhand_cfg_t& cfg, int e, const float* r,
                                            const float* __restrict__ origins, const float* __restrict__ in_hand,
                                            const float* __restrict__ goal_rot, const float* __restrict__ actions,
                                            uint8_t reset_flag, const uint8_t* reset_goal_buf,
                                            float* successes, float* __restrict__ reward,
                                            uint8_t* goal_resets, double& my_resets, double& my_succ) {
    V3 pos{r[0] - origins[3 * e], r[1] - origins[3 * e + 1], r[2] - origins[3 * e + 2]};
    V3 d{pos.x - in_hand[3 * e], pos.y - in_hand[3 * e + 1], pos.z - in_hand[3 * e + 2]};
    float goal_dist = norm3(d);
    float rot_dist = rotation_distance(ld4(r + 3), ld4(goal_rot + 4 * (int64_t)e));
    float dist_rew = goal_dist * cfg.dist_reward_scale;
    float rot_rew = 1.0f / (fabsf(rot_dist) + cfg.rot_eps) * cfg.rot_reward_scale;
    float ap = 0.0f;
    int A = cfg.num_actuated;
    const float* arow = actions + (int64_t)e * A;
    if ((A & 3) == 0 && ((uintptr_t)actions & 15) == 0) {      // same sum order, four columns per load
        for (int k = 0; k < A; k += 4) {
            const float4 a4 = *reinterpret_cast<const float4*>(arow + k);
            ap += a4.x * a4.x; ap += a4.y * a4.y; ap += a4.z * a4.z; ap += a4.w * a4.w;
        }
    } else {
        for (int k = 0; k < A; ++k) { float a = arow[k]; ap += a * a; }
    }
    float rew = (dist_rew + rot_rew) + ap * cfg.action_penalty_scale;
    uint8_t gr = (fabsf(rot_dist) <= cfg.success_tolerance) ? 1 : reset_goal_buf[e];
    float succ = successes[e] + (float)gr;
    if (gr == 1) rew = rew + cfg.reach_goal_bonus;
    bool fell = goal_dist >= cfg.fall_dist;
    if (fell) rew = rew + cfg.fall_penalty;
    uint8_t resets = fell ? 1 : reset_flag;
    successes[e] = succ;
    reward[e] = rew;
    goal_resets[e] = gr;
    my_resets = (double)resets;
    my_succ = (double)(succ * (float)resets);
}

// wave reduction, the block's waves through LDS, then one fp64 atomic pair per block (both sums are sums of
// integer-valued terms: exact in any order); all of them land on the same two words, so fewer is faster.
// Every thread of the block must call it.
__device__ __forceinline__ void rewards_block_sums(double my_resets, double my_succ, double* __restrict__ partials) {
    __shared__ double s_part[2][16];
    for (int off = 32; off > 0; off >>= 1) {
        my_resets += __shfl_down(my_resets, off);
        my_succ += __shfl_down(my_succ, off);
    }
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { s_part[0][wave] = my_resets; s_part[1][wave] = my_succ; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0, q = 0.0;
        for (int w = 0; w < nw; ++w) { r += s_part[0][w]; q += s_part[1][w]; }
        if (r != 0.0 || q != 0.0) {
            atomicAdd(&partials[0], r);
            atomicAdd(&partials[1], q);
        }
    }
}

__global__ void k_rewards(mdp_inhand_cfg_t cfg, const float* __restrict__ obj, const float* __restrict__ origins,
                          const float* __restrict__ in_hand, const float* __restrict__ goal_rot,
                          const float* __restrict__ actions, const uint8_t* __restrict__ reset_buf,
                          const uint8_t* reset_goal_buf, float* __restrict__ successes,
                          float* __restrict__ reward, uint8_t* goal_resets, double* __restrict__ partials) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    double my_resets = 0.0, my_succ = 0.0;
    if (e < cfg.num_envs)
        rewards_env(cfg, e, obj + 13 * (int64_t)e, origins, in_hand, goal_rot, actions, reset_buf[e], reset_goal_buf,
                    successes, reward, goal_resets, my_resets, my_succ);
    rewards_block_sums(my_resets, my_succ, partials);
}

// direct_rl_env.py:343-349 in one launch: the last sub-step's `hand.data.update(dt)` on blocks in front of the grid (nothing
// below reads the hand), then per env `episode_length_buf += 1`, _get_dones, reset_buf, _get_rewards -- the rewards take
// the env's reset flag and the object's state from the registers of the thread that just produced them
__global__ void __launch_bounds__(256) k_dones_rewards(mdp_inhand_cfg_t cfg, float* obj, const float* __restrict__ origins,
                                                       const float* __restrict__ in_hand, const float* __restrict__ goal_rot,
                                                       float* successes, int64_t* __restrict__ ep_len,
                                                       uint8_t* __restrict__ out_of_reach, uint8_t* __restrict__ time_out,
                                                       uint8_t* __restrict__ reset_buf, const float* __restrict__ obj_xf,
                                                       const float* __restrict__ obj_vel, const float* __restrict__ actions,
                                                       uint8_t* reset_goal_buf, float* __restrict__ reward,
                                                       double* __restrict__ partials, mdp_scene_update_t upd, float dt,
                                                       unsigned upd_total, unsigned nb_root, unsigned nb_joint,
                                                       unsigned nb_contact, int upd_envs) {
    // the per-env blocks are the longer chain (loads -> dones -> rewards -> block sums -> atomics): they go first in the
    // grid, the update blocks fill in behind them
    const unsigned task_blocks = gridDim.x - upd_total;
    if (blockIdx.x >= task_blocks) {
        scene_update_carried(upd, dt, blockIdx.x - task_blocks, nb_root, nb_joint, nb_contact, upd_envs);
        return;
    }
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    double my_resets = 0.0, my_succ = 0.0;
    if (e < cfg.num_envs) {
        float row[13];
        const uint8_t reset = dones_env(cfg, e, obj, origins, in_hand, goal_rot, successes, ep_len, out_of_reach, time_out,
                                        1, reset_buf, obj_xf, obj_vel, obj, row);
        rewards_env(cfg, e, row, origins, in_hand, goal_rot, actions, reset, reset_goal_buf, successes, reward,
                    reset_goal_buf, my_resets, my_succ);
    }
    rewards_block_sums(my_resets, my_succ, partials);
}

__global__ void k_finish_rewards(mdp_inhand_cfg_t cfg, const double* __restrict__ partials,
                                 float* __restrict__ cons, double* __restrict__ rezero) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float num_resets = (float)partials[0];
        float finished = (float)partials[1];
        if (num_resets > 0.0f)
            cons[0] = cfg.av_factor * finished / num_resets + (1.0f - cfg.av_factor) * cons[0];
        if (rezero) { rezero[0] = 0.0; rezero[1] = 0.0; }
    }
}

// :298-322.  A block owns FULL_OBS_EPB whole envs.  Which array, row stride, offset and operation a column takes does not
// depend on the env, so the block lowers the column list ONCE into descriptors in LDS (one pass of the branch chain per
// column) and then every (env, column) element costs a descriptor read, up to three loads issued together, a select
// and a store -- with a flat 64-bit index per element the division and the branch chain were ~200 vector instructions
// per element and the kernel was issue-bound (23 us for 20 MB).  v = scale * a | a - b | unscale(a, b, c) | one
// component of the quaternion difference; 1.0f * a == a bit for bit, so plain copies share the scaled form.
#ifndef MDP_FULL_OBS_EPB
#define MDP_FULL_OBS_EPB 8
#endif
constexpr int FULL_OBS_EPB = MDP_FULL_OBS_EPB;
struct FullObsCol {
    const float* pa; const float* pb; const float* pc;   // element of env 0 (pb, pc may be null)
    int sa, sb, sc;                                        // floats per env
    int kind;                                              // 0 scale * a, 1 a - b, 2 unscale(a, b, c), 3 quaternion difference
    float scale;
    int qk;                                                // kind 3: component
};

__global__ void k_full_obs(mdp_inhand_cfg_t cfg, const float* __restrict__ q, const float* __restrict__ qd,
                           const float* __restrict__ lower, const float* __restrict__ upper,
                           const float* __restrict__ obj, const float* __restrict__ origins,
                           const float* __restrict__ in_hand, const float* __restrict__ goal_rot,
                           const float* __restrict__ body, const int32_t* __restrict__ fingers,
                           const float* __restrict__ actions, int A, float* __restrict__ obs, int ld_obs, int O,
                           const float* __restrict__ link_xf, const float* __restrict__ link_vel,
                           double* __restrict__ partials, float* __restrict__ cons) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    if (partials && blockIdx.x == 0 && threadIdx.x == 0) {
        // mdp_inhand_finish_rewards_step riding along: the reward launch in front of this one left the two sums
        const float num_resets = (float)partials[0];
        const float finished = (float)partials[1];
        if (num_resets > 0.0f) cons[0] = cfg.av_factor * finished / num_resets + (1.0f - cfg.av_factor) * cons[0];
        partials[0] = 0.0; partials[1] = 0.0;
    }
    FullObsCol* cols = reinterpret_cast<FullObsCol*>(s_raw);
    const int D = cfg.num_dofs, F = cfg.num_fingertips, B = cfg.num_bodies;
    for (int c = threadIdx.x; c < O; c += blockDim.x) {
        FullObsCol d{obj, nullptr, nullptr, 13, 0, 0, 0, 1.0f, 0};
        int k = c;
        if (k < D) {
            d.pa = q + k; d.pb = lower + k; d.pc = upper + k; d.sa = d.sb = d.sc = D; d.kind = 2;
        } else if ((k -= D) < D) {
            d.pa = qd + k; d.sa = D; d.scale = cfg.vel_obs_scale;
        } else if ((k -= D) < 3) {
            d.pa = obj + k; d.pb = origins + k; d.sb = 3; d.kind = 1;          // object_pos
        } else if ((k -= 3) < 4) {
            d.pa = obj + 3 + k;                                                  // object_rot
        } else if ((k -= 4) < 3) {
            d.pa = obj + 7 + k;                                                  // object_linvel
        } else if ((k -= 3) < 3) {
            d.pa = obj + 10 + k; d.scale = cfg.vel_obs_scale;
        } else if ((k -= 3) < 3) {
            d.pa = in_hand + k; d.sa = 3;
        } else if ((k -= 3) < 4) {
            d.pa = goal_rot + k; d.sa = 4;
        } else if ((k -= 4) < 4) {
            d.pa = obj + 3; d.pb = goal_rot; d.sb = 4; d.kind = 3; d.qk = k;
        } else if ((k -= 4) < 3 * F) {
            int f = k / 3, a = k - 3 * f;
            // body == nullptr: the fingertip rows come from the physics view's link buffers (xyzw transforms, (N,B,7) and
            // (N,B,6)) -- the values body_state_w would hold (articulation_data.py:323-337) without packing all B bodies
            if (body) { d.pa = body + (int64_t)fingers[f] * 13 + a; d.sa = B * 13; }
            else { d.pa = link_xf + (int64_t)fingers[f] * 7 + a; d.sa = B * 7; }
            d.pb = origins + a; d.sb = 3; d.kind = 1;
        } else if ((k -= 3 * F) < 4 * F) {
            int f = k / 4, a = k - 4 * f;
            if (body) { d.pa = body + (int64_t)fingers[f] * 13 + 3 + a; d.sa = B * 13; }
            else { d.pa = link_xf + (int64_t)fingers[f] * 7 + (a == 0 ? 6 : 2 + a); d.sa = B * 7; }
        } else if ((k -= 4 * F) < 6 * F) {
            int f = k / 6, a = k - 6 * f;
            if (body) { d.pa = body + (int64_t)fingers[f] * 13 + 7 + a; d.sa = B * 13; }
            else { d.pa = link_vel + (int64_t)fingers[f] * 6 + a; d.sa = B * 6; }
        } else {
            k -= 6 * F;
            d.pa = actions + k; d.sa = A;
        }
        cols[c] = d;
    }
    __syncthreads();
    const int64_t e0 = (int64_t)blockIdx.x * FULL_OBS_EPB;
    const int64_t e1 = e0 + FULL_OBS_EPB < cfg.num_envs ? e0 + FULL_OBS_EPB : cfg.num_envs;
    for (int c = threadIdx.x; c < O; c += blockDim.x) {
        const FullObsCol d = cols[c];
#pragma unroll 4
        for (int64_t e = e0; e < e1; ++e) {
            const float* pa = d.pa + e * d.sa;
            const float va = *pa;
            const float vb = (d.pb && d.kind != 3) ? d.pb[e * d.sb] : 0.0f;
            const float vc = d.pc ? d.pc[e * d.sc] : 0.0f;
            float v;
            if (d.kind == 3) v = get(quat_mul(ld4(pa), quat_conjugate(ld4(d.pb + e * d.sb))), d.qk);
            else v = d.kind == 0 ? d.scale * va : (d.kind == 1 ? va - vb : ih_unscale(va, vb, vc));
            obs[e * ld_obs + c] = v;
        }
    }
}

}  // namespace

extern "C" int mdp_inhand_apply_action(const mdp_inhand_cfg_t* cfg, const float* actions,
                                       const int32_t* actuated_idx, const float* lower, const float* upper,
                                       float* cur_targets, float* prev_targets, void* stream) {
    MDP_REQUIRE(cfg && actions && actuated_idx && lower && upper && cur_targets && prev_targets, "null pointer");
    int64_t total = (int64_t)cfg->num_envs * cfg->num_actuated;
    if (total == 0) return MDP_OK;
    hipLaunchKernelGGL(k_apply_action, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), *cfg, actions,
                       actuated_idx, lower, upper, cur_targets, prev_targets, (float*)nullptr, 0, (float*)nullptr);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_apply_action_to(const mdp_inhand_cfg_t* cfg, const float* actions, float* actions_copy,
                                          const int32_t* actuated_idx, const float* lower, const float* upper,
                                          float* cur_targets, float* prev_targets, float* joint_pos_target,
                                          int ld_target, void* stream) {
    MDP_REQUIRE(cfg && actions && actuated_idx && lower && upper && cur_targets && prev_targets && joint_pos_target,
                "null pointer");
    MDP_REQUIRE(ld_target >= cfg->num_dofs, "joint_pos_target row stride %d < %d dofs", ld_target, cfg->num_dofs);
    int64_t total = (int64_t)cfg->num_envs * cfg->num_actuated;
    if (total == 0) return MDP_OK;
    hipLaunchKernelGGL(k_apply_action, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), *cfg, actions,
                       actuated_idx, lower, upper, cur_targets, prev_targets, joint_pos_target, ld_target,
                       actions_copy == actions ? (float*)nullptr : actions_copy);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_apply_action_to_with_update(const mdp_inhand_cfg_t* cfg, const float* actions,
                                                      float* actions_copy, const int32_t* actuated_idx,
                                                      const float* lower, const float* upper, float* cur_targets,
                                                      float* prev_targets, float* joint_pos_target, int ld_target,
                                                      const mdp_scene_update_t* update, float dt, void* stream) {
    MDP_REQUIRE(cfg && actions && actuated_idx && lower && upper && cur_targets && prev_targets && joint_pos_target && update,
                "null pointer");
    MDP_REQUIRE(ld_target >= cfg->num_dofs, "joint_pos_target row stride %d < %d dofs", ld_target, cfg->num_dofs);
    SceneUpdatePlan plan;
    int rc = scene_update_plan(update, &plan);
    if (rc != MDP_OK) return rc;
    MDP_REQUIRE(update->num_envs == cfg->num_envs, "scene update disagrees on num_envs");
    int64_t total = (int64_t)cfg->num_envs * cfg->num_actuated;
    if (total == 0) return MDP_OK;
    const unsigned upd_total = plan.carried_total;
    hipLaunchKernelGGL(k_apply_action_upd, dim3(upd_total + grid_for(total, 256)), dim3(256), 0, as_stream(stream), *cfg,
                       actions, actuated_idx, lower, upper, cur_targets, prev_targets, joint_pos_target, ld_target,
                       actions_copy == actions ? (float*)nullptr : actions_copy, plan.d, dt, upd_total, plan.nb_root,
                       plan.nb_joint, plan.nb_contact, plan.carried_envs);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_dones_rewards_step(const mdp_inhand_cfg_t* cfg, float* object_root_state,
                                             const float* env_origins, const float* in_hand_pos, const float* goal_rot,
                                             float* successes, int64_t* episode_length_buf, uint8_t* reset_terminated,
                                             uint8_t* reset_time_outs, uint8_t* reset_buf,
                                             const float* object_transforms, const float* object_velocities,
                                             const float* actions, uint8_t* reset_goal_buf, float* reward,
                                             double* partials, const mdp_scene_update_t* update, float dt, void* stream) {
    MDP_REQUIRE(cfg && object_root_state && env_origins && in_hand_pos && goal_rot && successes && episode_length_buf &&
                    reset_terminated && reset_time_outs && reset_buf && actions && reset_goal_buf && reward && partials,
                "null pointer");
    MDP_REQUIRE((object_transforms == nullptr) == (object_velocities == nullptr),
                "object_transforms and object_velocities go together");
    if (cfg->num_envs == 0) return MDP_OK;
    SceneUpdatePlan plan;
    unsigned upd_total = 0;
    if (update) {
        int rc = scene_update_plan(update, &plan);
        if (rc != MDP_OK) return rc;
        MDP_REQUIRE(update->num_envs == cfg->num_envs, "scene update disagrees on num_envs");
        upd_total = plan.carried_total;
    } else {
        plan.d = mdp_scene_update_t{};
    }
    hipLaunchKernelGGL(k_dones_rewards, dim3(upd_total + grid_for(cfg->num_envs, 256, 1 << 20)), dim3(256), 0,
                       as_stream(stream), *cfg, object_root_state, env_origins, in_hand_pos, goal_rot, successes,
                       episode_length_buf, reset_terminated, reset_time_outs, reset_buf, object_transforms,
                       object_velocities, actions, reset_goal_buf, reward, partials, plan.d, dt, upd_total, plan.nb_root,
                       plan.nb_joint, plan.nb_contact, plan.carried_envs);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_dones(const mdp_inhand_cfg_t* cfg, const float* object_root_state, const float* env_origins,
                                const float* in_hand_pos, const float* goal_rot, const float* successes,
                                int64_t* episode_length_buf, uint8_t* out_of_reach, uint8_t* time_out, void* stream) {
    MDP_REQUIRE(cfg && object_root_state && env_origins && in_hand_pos && goal_rot && successes &&
                    episode_length_buf && out_of_reach && time_out,
                "null pointer");
    if (cfg->num_envs == 0) return MDP_OK;
    hipLaunchKernelGGL(k_dones, dim3(grid_for(cfg->num_envs, 256, 1 << 20)), dim3(256), 0, as_stream(stream), *cfg,
                       object_root_state, env_origins, in_hand_pos, goal_rot, successes, episode_length_buf,
                       out_of_reach, time_out, 0, (uint8_t*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (float*)nullptr);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_dones_step(const mdp_inhand_cfg_t* cfg, float* object_root_state,
                                     const float* env_origins, const float* in_hand_pos, const float* goal_rot,
                                     const float* successes, int64_t* episode_length_buf, uint8_t* reset_terminated,
                                     uint8_t* reset_time_outs, uint8_t* reset_buf, const float* object_transforms,
                                     const float* object_velocities, void* stream) {
    MDP_REQUIRE(cfg && object_root_state && env_origins && in_hand_pos && goal_rot && successes &&
                    episode_length_buf && reset_terminated && reset_time_outs && reset_buf,
                "null pointer");
    MDP_REQUIRE((object_transforms == nullptr) == (object_velocities == nullptr),
                "object_transforms and object_velocities go together");
    if (cfg->num_envs == 0) return MDP_OK;
    hipLaunchKernelGGL(k_dones, dim3(grid_for(cfg->num_envs, 256, 1 << 20)), dim3(256), 0, as_stream(stream), *cfg,
                       object_root_state, env_origins, in_hand_pos, goal_rot, successes, episode_length_buf,
                       reset_terminated, reset_time_outs, 1, reset_buf, object_transforms, object_velocities,
                       object_root_state);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_rewards(const mdp_inhand_cfg_t* cfg, const float* object_root_state,
                                  const float* env_origins, const float* in_hand_pos, const float* goal_rot,
                                  const float* actions, const uint8_t* reset_buf, const uint8_t* reset_goal_buf,
                                  float* successes, float* reward, uint8_t* goal_resets, double* partials,
                                  void* stream) {
    MDP_REQUIRE(cfg && object_root_state && env_origins && in_hand_pos && goal_rot && actions && reset_buf &&
                    reset_goal_buf && successes && reward && goal_resets && partials,
                "null pointer");
    if (cfg->num_envs == 0) return MDP_OK;
    hipStream_t s = as_stream(stream);
    MDP_CHECK_HIP(hipMemsetAsync(partials, 0, 2 * sizeof(double), s));
    hipLaunchKernelGGL(k_rewards, dim3(grid_for(cfg->num_envs, 256, 1 << 20)), dim3(256), 0, s, *cfg,
                       object_root_state, env_origins, in_hand_pos, goal_rot, actions, reset_buf, reset_goal_buf,
                       successes, reward, goal_resets, partials);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_finish_rewards(const mdp_inhand_cfg_t* cfg, const double* partials,
                                         float* consecutive_successes, void* stream) {
    MDP_REQUIRE(cfg && partials && consecutive_successes, "null pointer");
    hipLaunchKernelGGL(k_finish_rewards, dim3(1), dim3(64), 0, as_stream(stream), *cfg, partials,
                       consecutive_successes, (double*)nullptr);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_rewards_step(const mdp_inhand_cfg_t* cfg, const float* object_root_state,
                                       const float* env_origins, const float* in_hand_pos, const float* goal_rot,
                                       const float* actions, const uint8_t* reset_buf, uint8_t* reset_goal_buf,
                                       float* successes, float* reward, double* partials, void* stream) {
    MDP_REQUIRE(cfg && object_root_state && env_origins && in_hand_pos && goal_rot && actions && reset_buf &&
                    reset_goal_buf && successes && reward && partials,
                "null pointer");
    if (cfg->num_envs == 0) return MDP_OK;
    // reset_goal_buf is read and written by the same thread of the same env: updated in place.  `partials` is zero on
    // entry: zero-initialised by its owner, left zeroed by mdp_inhand_finish_rewards_step of the step before
    hipLaunchKernelGGL(k_rewards, dim3(grid_for(cfg->num_envs, 256, 1 << 20)), dim3(256), 0, as_stream(stream), *cfg,
                       object_root_state, env_origins, in_hand_pos, goal_rot, actions, reset_buf, reset_goal_buf,
                       successes, reward, reset_goal_buf, partials);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_finish_rewards_step(const mdp_inhand_cfg_t* cfg, double* partials,
                                              float* consecutive_successes, void* stream) {
    MDP_REQUIRE(cfg && partials && consecutive_successes, "null pointer");
    hipLaunchKernelGGL(k_finish_rewards, dim3(1), dim3(64), 0, as_stream(stream), *cfg, partials,
                       consecutive_successes, partials);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_full_obs(const mdp_inhand_cfg_t* cfg, const float* joint_pos, const float* joint_vel,
                                   const float* lower, const float* upper, const float* object_root_state,
                                   const float* env_origins, const float* in_hand_pos, const float* goal_rot,
                                   const float* body_state_w, const int32_t* finger_bodies, const float* actions,
                                   int action_dim, float* obs, int ld_obs, void* stream) {
    MDP_REQUIRE(cfg && joint_pos && joint_vel && lower && upper && object_root_state && env_origins && in_hand_pos &&
                    goal_rot && body_state_w && finger_bodies && actions && obs,
                "null pointer");
    int O = 2 * cfg->num_dofs + 13 + 11 + 13 * cfg->num_fingertips + action_dim;
    MDP_REQUIRE(ld_obs >= O, "obs buffer too narrow (%d < %d)", ld_obs, O);
    if (cfg->num_envs == 0) return MDP_OK;
    const int threads = std::min(256, (O + 63) / 64 * 64);
    const size_t lds = (size_t)O * sizeof(FullObsCol);
    MDP_REQUIRE(lds <= 60 * 1024, "observation too wide for the column table (%d columns)", O);
    hipLaunchKernelGGL(k_full_obs, dim3(grid_for(cfg->num_envs, FULL_OBS_EPB)), dim3(threads), lds, as_stream(stream), *cfg, joint_pos,
                       joint_vel, lower, upper, object_root_state, env_origins, in_hand_pos, goal_rot, body_state_w,
                       finger_bodies, actions, action_dim, obs, ld_obs, O, (const float*)nullptr, (const float*)nullptr,
                       (double*)nullptr, (float*)nullptr);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_full_obs_from_view(const mdp_inhand_cfg_t* cfg, const float* joint_pos, const float* joint_vel,
                                             const float* lower, const float* upper, const float* object_root_state,
                                             const float* env_origins, const float* in_hand_pos, const float* goal_rot,
                                             const float* link_transforms, const float* link_velocities,
                                             const int32_t* finger_bodies, const float* actions, int action_dim,
                                             float* obs, int ld_obs, double* partials, float* consecutive_successes,
                                             void* stream) {
    MDP_REQUIRE(cfg && joint_pos && joint_vel && lower && upper && object_root_state && env_origins && in_hand_pos &&
                    goal_rot && link_transforms && link_velocities && finger_bodies && actions && obs,
                "null pointer");
    MDP_REQUIRE((partials == nullptr) == (consecutive_successes == nullptr), "partials and consecutive_successes go together");
    int O = 2 * cfg->num_dofs + 13 + 11 + 13 * cfg->num_fingertips + action_dim;
    MDP_REQUIRE(ld_obs >= O, "obs buffer too narrow (%d < %d)", ld_obs, O);
    if (cfg->num_envs == 0) return MDP_OK;
    const int threads = std::min(256, (O + 63) / 64 * 64);
    const size_t lds = (size_t)O * sizeof(FullObsCol);
    MDP_REQUIRE(lds <= 60 * 1024, "observation too wide for the column table (%d columns)", O);
    hipLaunchKernelGGL(k_full_obs, dim3(grid_for(cfg->num_envs, FULL_OBS_EPB)), dim3(threads), lds, as_stream(stream), *cfg, joint_pos,
                       joint_vel, lower, upper, object_root_state, env_origins, in_hand_pos, goal_rot,
                       (const float*)nullptr, finger_bodies, actions, action_dim, obs, ld_obs, O, link_transforms,
                       link_velocities, partials, consecutive_successes);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}
