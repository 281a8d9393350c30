// Shadow-hand in-hand cube reorientation (direct workflow).
// Reference: TASKS/direct/inhand_manipulation/inhand_manipulation_env.py:97-117,179-199,262-433
#include "mdp_device.hpp"
#include "mdp_host.hpp"
#include "scene_update_device.hpp"
#include "uniforms_device.hpp"

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

// :363-367 randomize_rotation; `np.pi` multiplies an fp32 tensor, i.e. the fp32 rounding of pi
MDP_DEV Q4 randomize_rotation(float rand0, float rand1, V3 x_unit, V3 y_unit) {
    const float pi = 3.14159265358979323846f;
    return quat_mul(quat_from_angle_axis(rand0 * pi, x_unit), quat_from_angle_axis(rand1 * pi, y_unit));
}
// sample_uniform(-1.0, 1.0, ...) (math.py:1331): rand * (upper - lower) + lower
MDP_DEV float ih_pm1(float u) { return u * 2.0f + -1.0f; }

constexpr uint32_t IH_STREAM = 0x49484e44u;
// uniforms of env e for the reset-path kernels (columns of the injected row, or the env's own generator epoch)
MDP_DEV Uniforms ih_uniforms(const mdp_inhand_goal_reset_t& g, int e) {
    return uniforms_for(g.injected_u, g.ld_injected, g.rng_state, IH_STREAM, e);
}
// :248-260 _reset_target_pose for env e, draws u(k0), u(k0 + 1)
MDP_DEV void goal_reset_env(const mdp_inhand_goal_reset_t& g, int e, const Uniforms& u, int k0) {
    const Q4 q = randomize_rotation(ih_pm1(u(k0)), ih_pm1(u(k0 + 1)), ld3(g.x_unit + 3 * (int64_t)e),
                                    ld3(g.y_unit + 3 * (int64_t)e));
    st4(g.goal_rot + 4 * (int64_t)e, q);
    g.reset_goal_buf[e] = 0;
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
                                                       const float* __restrict__ in_hand, const float* goal_rot,
                                                       float* successes, int64_t* __restrict__ ep_len,
                                                       uint8_t* __restrict__ out_of_reach, uint8_t* __restrict__ time_out,
                                                       uint8_t* __restrict__ reset_buf, const float* __restrict__ obj_xf,
                                                       const float* __restrict__ obj_vel, const float* __restrict__ actions,
                                                       uint8_t* reset_goal_buf, float* __restrict__ reward,
                                                       double* __restrict__ partials, mdp_scene_update_t upd, float dt,
                                                       unsigned upd_total, unsigned nb_root, unsigned nb_joint,
                                                       unsigned nb_contact, int upd_envs, int has_goal,
                                                       mdp_inhand_goal_reset_t goal) {
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
        // :172-175: `if len(goal_env_ids) > 0: self._reset_target_pose(goal_env_ids)` -- the env's own thread has just
        // written its goal_resets flag and is the only reader of its goal_rot in this launch
        if (has_goal && reset_goal_buf[e]) {
            const Uniforms u = ih_uniforms(goal, e);
            goal_reset_env(goal, e, u, 0);
        }
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

// :201-246 + direct_rl_env.py:560-581 for the envs of a mask.  One thread per env reads its flag (a launch with nothing
// to reset is N byte reads); a flagged lane does its env's goal and object itself (9 draws), then the wave walks its
// flagged envs one after the other with the lanes over the joints (2 D draws per env).
// Wave-level routine: lane `lane` holds env e (consecutive over the lanes) and whether it is flagged; every lane of the
// wave must call it.
__device__ __forceinline__ void inhand_reset_wave(const mdp_inhand_cfg_t& cfg, const mdp_inhand_reset_t& a, int e, bool flagged) {
    const int lane = threadIdx.x & 63;
    const mdp_inhand_goal_reset_t& g = a.goal;
    unsigned long long todo = __ballot(flagged);
    if (todo == 0ull) return;
    const int D = cfg.num_dofs;
    Uniforms u{g.injected_u, g.ld_injected, 0, 0, IH_STREAM, e};
    if (flagged) {
        if (!g.injected_u) {
            u.seed = g.rng_state[0];
            u.epoch = g.rng_state[1 + e];
            g.rng_state[1 + e] = u.epoch + 1;
        }
        goal_reset_env(g, e, u, 0);                                            // :208
        // :211-225 object: default state + position noise + origin, randomize_rotation, zero velocity
        const float* d = a.object_default_root_state + 13 * (int64_t)e;
        const float* org = a.env_origins + 3 * (int64_t)e;
        V3 pos;
        pos.x = (d[0] + a.reset_position_noise * ih_pm1(u(2))) + org[0];
        pos.y = (d[1] + a.reset_position_noise * ih_pm1(u(3))) + org[1];
        pos.z = (d[2] + a.reset_position_noise * ih_pm1(u(4))) + org[2];
        const Q4 q = randomize_rotation(ih_pm1(u(5)), ih_pm1(u(6)), ld3(g.x_unit + 3 * (int64_t)e), ld3(g.y_unit + 3 * (int64_t)e));
        float* r = a.object_root_state_w + 13 * (int64_t)e;
        st3(r, pos); st4(r + 3, q);
        for (int k = 7; k < 13; ++k) r[k] = 0.0f;
        float* t = a.sim_object_transforms + 7 * (int64_t)e;                  // (x, y, z, qx, qy, qz, qw)
        st3(t, pos); t[3] = q.x; t[4] = q.y; t[5] = q.z; t[6] = q.w;
        float* v = a.sim_object_velocities + 6 * (int64_t)e;
        for (int k = 0; k < 6; ++k) v[k] = 0.0f;
        a.successes[e] = 0.0f;                                                 // :245
        a.episode_length_buf[e] = 0;                                           // direct_rl_env.py:581
    }
    // :228-243 hand
    const int e_wave = e - lane;
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const int ee = e_wave + src;
        Uniforms ue{g.injected_u, g.ld_injected, 0, 0, IH_STREAM, ee};
        ue.seed = __shfl(u.seed, src);
        ue.epoch = __shfl(u.epoch, src);
        for (int j = lane; j < D; j += 64) {
            const int64_t i = (int64_t)ee * D + j;
            const float dq = a.default_joint_pos[i];
            const float delta_max = a.upper[i] - dq, delta_min = a.lower[i] - dq;
            const float rand_delta = delta_min + ((delta_max - delta_min) * 0.5f) * ih_pm1(ue(7 + j));
            const float q = dq + a.reset_dof_pos_noise * rand_delta;
            const float qd = a.default_joint_vel[i] + a.reset_dof_vel_noise * ih_pm1(ue(7 + D + j));
            a.prev_targets[i] = q; a.cur_targets[i] = q; a.hand_dof_targets[i] = q;
            a.joint_pos_target[(int64_t)ee * a.ld_target + j] = q;
            a.joint_pos[i] = q; a.joint_vel[i] = qd; a.prev_joint_vel[i] = qd; a.joint_acc[i] = 0.0f;
            a.sim_dof_positions[(int64_t)ee * a.ld_sim + j] = q;
            a.sim_dof_velocities[(int64_t)ee * a.ld_sim + j] = qd;
        }
    }
}

__global__ void __launch_bounds__(256) k_inhand_reset_idx(mdp_inhand_cfg_t cfg, mdp_inhand_reset_t a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    inhand_reset_wave(cfg, a, e, e < cfg.num_envs && (!a.goal.env_mask || a.goal.env_mask[e]));
}

// :281-350, the three observation layouts.  A block owns FULL_OBS_EPB whole envs.  Which array, row stride, offset and
// operation a column takes does not depend on the env, so the block lowers the column lists of its (up to two) outputs
// ONCE into descriptors in LDS (one pass of the branch chain per column) and then every (env, column) element costs a
// descriptor read, up to three loads issued together, a select and a store -- with a flat 64-bit index per element the
// division and the branch chain were ~200 vector instructions per element and the kernel was issue-bound (23 us for
// 20 MB).  v = scale * a | a - b | unscale(a, b, c) | one component of the quaternion difference; 1.0f * a == a bit
// for bit, so plain copies share the scaled form.
#ifndef MDP_FULL_OBS_EPB
#define MDP_FULL_OBS_EPB 8
#endif
constexpr int FULL_OBS_EPB = MDP_FULL_OBS_EPB;
struct FullObsCol {
    const float* pa; const float* pb; const float* pc;   // element of env 0 (pb, pc may be null)
    float* out;                                            // column of env 0 in its output
    int sa, sb, sc, so;                                    // floats per env
    int kind;                                              // 0 scale * a, 1 a - b, 2 unscale(a, b, c), 3 quaternion difference
    float scale;
    int qk;                                                // kind 3: component
};

// columns of a layout (mdp_inhand_obs_dim)
__host__ __device__ inline int inhand_obs_dim(int D, int F, int A, int kind) {
    if (kind == MDP_INHAND_OBS_FULL) return 2 * D + 13 + 11 + 13 * F + A;
    if (kind == MDP_INHAND_OBS_FULL_STATE) return 2 * D + 13 + 11 + 13 * F + 6 * F + A;
    if (kind == MDP_INHAND_OBS_REDUCED) return 3 * F + 3 + 4 + A;
    return 0;
}

// descriptor of column c of layout `kind`
__device__ __forceinline__ FullObsCol inhand_obs_col(const mdp_inhand_cfg_t& cfg, const mdp_inhand_obs_args_t& a, int kind,
                                                     int c) {
    const int D = cfg.num_dofs, F = cfg.num_fingertips, B = cfg.num_bodies, A = a.action_dim;
    const float* obj = a.object_root_state;
    FullObsCol d{obj, nullptr, nullptr, nullptr, 13, 0, 0, 0, 0, 1.0f, 0};
    auto tip_pos = [&](int k) {
        const int f = k / 3, x = k - 3 * f;
        // body_state_w == nullptr: the fingertip rows come from the physics view's link buffers (xyzw transforms, (N,B,7)
        // and (N,B,6)) -- the values body_state_w would hold (articulation_data.py:323-337) without packing all B bodies
        if (a.body_state_w) { d.pa = a.body_state_w + (int64_t)a.finger_bodies[f] * 13 + x; d.sa = B * 13; }
        else { d.pa = a.link_transforms + (int64_t)a.finger_bodies[f] * 7 + x; d.sa = B * 7; }
        d.pb = a.env_origins + x; d.sb = 3; d.kind = 1;
    };
    int k = c;
    if (kind == MDP_INHAND_OBS_REDUCED) {   // :286-294
        if (k < 3 * F) {
            tip_pos(k);
        } else if ((k -= 3 * F) < 3) {
            d.pa = obj + k; d.pb = a.env_origins + k; d.sb = 3; d.kind = 1;     // object_pos
        } else if ((k -= 3) < 4) {
            d.pa = obj + 3; d.pb = a.goal_rot; d.sb = 4; d.kind = 3; d.qk = k;
        } else {
            k -= 4;
            d.pa = a.actions + k; d.sa = A;
        }
        return d;
    }
    if (k < D) {
        d.pa = a.joint_pos + k; d.pb = a.lower + k; d.pc = a.upper + k; d.sa = d.sb = d.sc = D; d.kind = 2;
    } else if ((k -= D) < D) {
        d.pa = a.joint_vel + k; d.sa = D; d.scale = cfg.vel_obs_scale;
    } else if ((k -= D) < 3) {
        d.pa = obj + k; d.pb = a.env_origins + k; d.sb = 3; d.kind = 1;         // object_pos
    } else if ((k -= 3) < 4) {
        d.pa = obj + 3 + k;                                                       // object_rot
    } else if ((k -= 4) < 3) {
        d.pa = obj + 7 + k;                                                       // object_linvel
    } else if ((k -= 3) < 3) {
        d.pa = obj + 10 + k; d.scale = cfg.vel_obs_scale;
    } else if ((k -= 3) < 3) {
        d.pa = a.in_hand_pos + k; d.sa = 3;
    } else if ((k -= 3) < 4) {
        d.pa = a.goal_rot + k; d.sa = 4;
    } else if ((k -= 4) < 4) {
        d.pa = obj + 3; d.pb = a.goal_rot; d.sb = 4; d.kind = 3; d.qk = k;
    } else if ((k -= 4) < 3 * F) {
        tip_pos(k);
    } else if ((k -= 3 * F) < 4 * F) {
        const int f = k / 4, x = k - 4 * f;
        if (a.body_state_w) { d.pa = a.body_state_w + (int64_t)a.finger_bodies[f] * 13 + 3 + x; d.sa = B * 13; }
        else { d.pa = a.link_transforms + (int64_t)a.finger_bodies[f] * 7 + (x == 0 ? 6 : 2 + x); d.sa = B * 7; }
    } else if ((k -= 4 * F) < 6 * F) {
        const int f = k / 6, x = k - 6 * f;
        if (a.body_state_w) { d.pa = a.body_state_w + (int64_t)a.finger_bodies[f] * 13 + 7 + x; d.sa = B * 13; }
        else { d.pa = a.link_velocities + (int64_t)a.finger_bodies[f] * 6 + x; d.sa = B * 6; }
    } else if (kind == MDP_INHAND_OBS_FULL_STATE && (k -= 6 * F) < 6 * F) {
        // :343-344 force_torque_obs_scale * fingertip_force_sensors, the view's incoming joint wrench of the fingertip links
        const int f = k / 6, x = k - 6 * f;
        d.pa = a.link_incoming_joint_force + (int64_t)a.finger_bodies[f] * 6 + x; d.sa = B * 6;
        d.scale = a.force_torque_obs_scale;
    } else {
        k -= 6 * F;    // past the fingertip velocities (full) / the force sensors (full state)
        d.pa = a.actions + k; d.sa = A;
    }
    return d;
}

// (register budget for 6 waves per SIMD: 2048 blocks of 3 waves at 16,384 envs are one residency round; the reset
// variant's few extra values live in scratch on the rarely taken reset path)
template <bool has_reset>
__global__ void __launch_bounds__(256, 6) k_inhand_obs(mdp_inhand_cfg_t cfg, mdp_inhand_obs_args_t a, int O1, int O2, mdp_inhand_reset_t rst) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    // direct_rl_env.py:355-358 riding along: `_reset_idx` of the block's done envs in front of their observations -- every
    // buffer the reset writes and the observation reads is the env's own row, so a block barrier orders them.  The flags
    // are requested first and looked at behind the column lowering; a block without a done env only paid that load.
    bool rst_flag = false;
    if (has_reset && threadIdx.x < 64) {
        const int64_t er = (int64_t)blockIdx.x * FULL_OBS_EPB + threadIdx.x;
        rst_flag = threadIdx.x < FULL_OBS_EPB && er < cfg.num_envs && (!rst.goal.env_mask || rst.goal.env_mask[er]);
    }
    if (a.partials && blockIdx.x == 0 && threadIdx.x == 0) {
        // mdp_inhand_finish_rewards_step riding along: the reward launch in front of this one left the two sums
        const float num_resets = (float)a.partials[0];
        const float finished = (float)a.partials[1];
        if (num_resets > 0.0f)
            a.consecutive_successes[0] = cfg.av_factor * finished / num_resets + (1.0f - cfg.av_factor) * a.consecutive_successes[0];
        a.partials[0] = 0.0; a.partials[1] = 0.0;
    }
    FullObsCol* cols = reinterpret_cast<FullObsCol*>(s_raw);
    const int O = O1 + O2;
    for (int c = threadIdx.x; c < O; c += blockDim.x) {
        FullObsCol d;
        if (c < O1) { d = inhand_obs_col(cfg, a, a.policy_kind, c); d.out = a.policy + c; d.so = a.ld_policy; }
        else { d = inhand_obs_col(cfg, a, a.critic_kind, c - O1); d.out = a.critic + (c - O1); d.so = a.ld_critic; }
        cols[c] = d;
    }
    if (has_reset) {
        if (threadIdx.x < 64) {                       // wave 0
            inhand_reset_wave(cfg, rst, (int)((int64_t)blockIdx.x * FULL_OBS_EPB + threadIdx.x), rst_flag);
            __threadfence_block();
        }
    }
    __syncthreads();
    const int64_t e0 = (int64_t)blockIdx.x * FULL_OBS_EPB;
    const int64_t e1 = e0 + FULL_OBS_EPB < cfg.num_envs ? e0 + FULL_OBS_EPB : cfg.num_envs;
    const int ne = (int)(e1 - e0);
    for (int c = threadIdx.x; c < O; c += blockDim.x) {
        const FullObsCol d = cols[c];
        // rows of the block's first env: one 64-bit product per array and column, 32-bit offsets per env
        const float* pa0 = d.pa + e0 * d.sa;
        const float* pb0 = d.pb ? d.pb + e0 * d.sb : nullptr;
        const float* pc0 = d.pc ? d.pc + e0 * d.sc : nullptr;
        float* out0 = d.out + e0 * d.so;
#pragma unroll 4
        for (int le = 0; le < ne; ++le) {
            const float* pa = pa0 + le * d.sa;
            const float va = *pa;
            const float vb = (pb0 && d.kind != 3) ? pb0[le * d.sb] : 0.0f;
            const float vc = pc0 ? pc0[le * d.sc] : 0.0f;
            float v;
            if (d.kind == 3) v = get(quat_mul(ld4(pa), quat_conjugate(ld4(pb0 + le * d.sb))), d.qk);
            else v = d.kind == 0 ? d.scale * va : (d.kind == 1 ? va - vb : ih_unscale(va, vb, vc));
            out0[le * d.so] = v;
        }
    }
}

// :248-260 for the envs of a mask, one thread per env
__global__ void k_reset_target_pose(mdp_inhand_cfg_t cfg, mdp_inhand_goal_reset_t g) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cfg.num_envs || (g.env_mask && !g.env_mask[e])) return;
    const Uniforms u = ih_uniforms(g, e);
    goal_reset_env(g, e, u, 0);
}

}  // namespace

static int goal_reset_check(const mdp_inhand_cfg_t* cfg, const mdp_inhand_goal_reset_t* g, int draws) {
    MDP_REQUIRE(g->x_unit && g->y_unit && g->goal_rot && g->reset_goal_buf, "null pointer in the goal reset");
    MDP_REQUIRE(g->injected_u || g->rng_state, "goal reset: neither injected uniforms nor a generator state");
    if (g->injected_u) MDP_REQUIRE(g->ld_injected >= draws, "injected uniforms: %d columns, %d needed", g->ld_injected, draws);
    return MDP_OK;
}

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
                                             double* partials, const mdp_scene_update_t* update, float dt,
                                             const mdp_inhand_goal_reset_t* goal_reset, void* stream) {
    MDP_REQUIRE(cfg && object_root_state && env_origins && in_hand_pos && goal_rot && successes && episode_length_buf &&
                    reset_terminated && reset_time_outs && reset_buf && actions && reset_goal_buf && reward && partials,
                "null pointer");
    MDP_REQUIRE((object_transforms == nullptr) == (object_velocities == nullptr),
                "object_transforms and object_velocities go together");
    if (goal_reset) {
        int rc = goal_reset_check(cfg, goal_reset, 2);
        if (rc != MDP_OK) return rc;
        MDP_REQUIRE(goal_reset->goal_rot == goal_rot && goal_reset->reset_goal_buf == reset_goal_buf,
                    "the goal reset must name this launch's goal_rot / reset_goal_buf");
    }
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
                       plan.nb_joint, plan.nb_contact, plan.carried_envs, goal_reset ? 1 : 0,
                       goal_reset ? *goal_reset : mdp_inhand_goal_reset_t{});
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

static int inhand_reset_check(const mdp_inhand_cfg_t* cfg, const mdp_inhand_reset_t* a) {
    int rc = goal_reset_check(cfg, &a->goal, 7 + 2 * cfg->num_dofs);
    if (rc != MDP_OK) return rc;
    MDP_REQUIRE(a->object_default_root_state && a->env_origins && a->object_root_state_w && a->sim_object_transforms &&
                    a->sim_object_velocities && a->default_joint_pos && a->default_joint_vel && a->lower && a->upper &&
                    a->prev_targets && a->cur_targets && a->hand_dof_targets && a->joint_pos_target && a->joint_pos &&
                    a->joint_vel && a->prev_joint_vel && a->joint_acc && a->sim_dof_positions && a->sim_dof_velocities &&
                    a->successes && a->episode_length_buf,
                "null pointer");
    MDP_REQUIRE(a->ld_target >= cfg->num_dofs && a->ld_sim >= cfg->num_dofs, "row stride below num_dofs");
    // the device generator hands an env 64 * 4 draws per epoch (Uniforms: index e * 64 + k / 4)
    MDP_REQUIRE(a->goal.injected_u || 7 + 2 * cfg->num_dofs <= 256, "too many dofs for one generator epoch per env");
    return MDP_OK;
}

static int inhand_obs_launch(const mdp_inhand_cfg_t* cfg, const mdp_inhand_obs_args_t& a, void* stream) {
    auto kind_ok = [](int k) { return k == MDP_INHAND_OBS_FULL || k == MDP_INHAND_OBS_REDUCED || k == MDP_INHAND_OBS_FULL_STATE; };
    MDP_REQUIRE(a.policy_kind < 0 || kind_ok(a.policy_kind), "unknown policy observation kind %d", a.policy_kind);
    MDP_REQUIRE(a.critic_kind < 0 || kind_ok(a.critic_kind), "unknown critic observation kind %d", a.critic_kind);
    MDP_REQUIRE(a.action_dim >= 0, "negative action_dim");
    const int D = cfg->num_dofs, F = cfg->num_fingertips;
    const int O1 = a.policy_kind < 0 ? 0 : inhand_obs_dim(D, F, a.action_dim, a.policy_kind);
    const int O2 = a.critic_kind < 0 ? 0 : inhand_obs_dim(D, F, a.action_dim, a.critic_kind);
    MDP_REQUIRE(O1 + O2 > 0, "no observation requested");
    MDP_REQUIRE(a.object_root_state && a.env_origins && a.goal_rot && a.finger_bodies && a.actions, "null pointer");
    MDP_REQUIRE(a.body_state_w || a.link_transforms, "neither body_state_w nor link_transforms given");
    const bool wide = (a.policy_kind >= 0 && a.policy_kind != MDP_INHAND_OBS_REDUCED) ||
                      (a.critic_kind >= 0 && a.critic_kind != MDP_INHAND_OBS_REDUCED);
    if (wide) {
        MDP_REQUIRE(a.joint_pos && a.joint_vel && a.lower && a.upper && a.in_hand_pos, "null pointer");
        MDP_REQUIRE(a.body_state_w || a.link_velocities, "link_velocities missing");
    }
    if (a.policy_kind == MDP_INHAND_OBS_FULL_STATE || a.critic_kind == MDP_INHAND_OBS_FULL_STATE)
        MDP_REQUIRE(a.link_incoming_joint_force, "compute_full_state needs link_incoming_joint_force");
    if (O1) MDP_REQUIRE(a.policy && a.ld_policy >= O1, "policy buffer missing or too narrow (%d < %d)", a.ld_policy, O1);
    if (O2) MDP_REQUIRE(a.critic && a.ld_critic >= O2, "critic buffer missing or too narrow (%d < %d)", a.ld_critic, O2);
    MDP_REQUIRE((a.partials == nullptr) == (a.consecutive_successes == nullptr), "partials and consecutive_successes go together");
    if (cfg->num_envs == 0) return MDP_OK;
    const int O = O1 + O2;
    const int threads = std::min(256, (O + 63) / 64 * 64);
    const size_t lds = (size_t)O * sizeof(FullObsCol);
    MDP_REQUIRE(lds <= 60 * 1024, "observation too wide for the column table (%d columns)", O);
    mdp_inhand_reset_t rst{};
    if (a.reset) {
        int rc = inhand_reset_check(cfg, a.reset);
        if (rc != MDP_OK) return rc;
        rst = *a.reset;
    }
    if (a.reset)
        hipLaunchKernelGGL(k_inhand_obs<true>, dim3(grid_for(cfg->num_envs, FULL_OBS_EPB)), dim3(threads), lds,
                           as_stream(stream), *cfg, a, O1, O2, rst);
    else
        hipLaunchKernelGGL(k_inhand_obs<false>, dim3(grid_for(cfg->num_envs, FULL_OBS_EPB)), dim3(threads), lds,
                           as_stream(stream), *cfg, a, O1, O2, rst);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_obs_dim(const mdp_inhand_cfg_t* cfg, int kind, int action_dim) {
    if (!cfg) return 0;
    return inhand_obs_dim(cfg->num_dofs, cfg->num_fingertips, action_dim, kind);
}

extern "C" int mdp_inhand_observations(const mdp_inhand_cfg_t* cfg, const mdp_inhand_obs_args_t* args, void* stream) {
    MDP_REQUIRE(cfg && args, "null pointer");
    return inhand_obs_launch(cfg, *args, stream);
}

extern "C" int mdp_inhand_full_obs(const mdp_inhand_cfg_t* cfg, const float* joint_pos, const float* joint_vel,
                                   const float* lower, const float* upper, const float* object_root_state,
                                   const float* env_origins, const float* in_hand_pos, const float* goal_rot,
                                   const float* body_state_w, const int32_t* finger_bodies, const float* actions,
                                   int action_dim, float* obs, int ld_obs, void* stream) {
    MDP_REQUIRE(cfg && body_state_w, "null pointer");
    mdp_inhand_obs_args_t a{};
    a.joint_pos = joint_pos; a.joint_vel = joint_vel; a.lower = lower; a.upper = upper;
    a.object_root_state = object_root_state; a.env_origins = env_origins; a.in_hand_pos = in_hand_pos; a.goal_rot = goal_rot;
    a.body_state_w = body_state_w; a.finger_bodies = finger_bodies; a.actions = actions; a.action_dim = action_dim;
    a.policy_kind = MDP_INHAND_OBS_FULL; a.ld_policy = ld_obs; a.policy = obs; a.critic_kind = -1;
    return inhand_obs_launch(cfg, a, stream);
}

extern "C" int mdp_inhand_full_obs_from_view(const mdp_inhand_cfg_t* cfg, const float* joint_pos, const float* joint_vel,
                                             const float* lower, const float* upper, const float* object_root_state,
                                             const float* env_origins, const float* in_hand_pos, const float* goal_rot,
                                             const float* link_transforms, const float* link_velocities,
                                             const int32_t* finger_bodies, const float* actions, int action_dim,
                                             float* obs, int ld_obs, double* partials, float* consecutive_successes,
                                             void* stream) {
    MDP_REQUIRE(cfg && link_transforms && link_velocities, "null pointer");
    mdp_inhand_obs_args_t a{};
    a.joint_pos = joint_pos; a.joint_vel = joint_vel; a.lower = lower; a.upper = upper;
    a.object_root_state = object_root_state; a.env_origins = env_origins; a.in_hand_pos = in_hand_pos; a.goal_rot = goal_rot;
    a.link_transforms = link_transforms; a.link_velocities = link_velocities;
    a.finger_bodies = finger_bodies; a.actions = actions; a.action_dim = action_dim;
    a.policy_kind = MDP_INHAND_OBS_FULL; a.ld_policy = ld_obs; a.policy = obs; a.critic_kind = -1;
    a.partials = partials; a.consecutive_successes = consecutive_successes;
    return inhand_obs_launch(cfg, a, stream);
}

extern "C" int mdp_inhand_reset_target_pose(const mdp_inhand_cfg_t* cfg, const mdp_inhand_goal_reset_t* goal, void* stream) {
    MDP_REQUIRE(cfg && goal, "null pointer");
    int rc = goal_reset_check(cfg, goal, 2);
    if (rc != MDP_OK) return rc;
    if (cfg->num_envs == 0) return MDP_OK;
    hipLaunchKernelGGL(k_reset_target_pose, dim3(grid_for(cfg->num_envs, 256, 1 << 20)), dim3(256), 0, as_stream(stream),
                       *cfg, *goal);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_inhand_reset_idx(const mdp_inhand_cfg_t* cfg, const mdp_inhand_reset_t* a, void* stream) {
    MDP_REQUIRE(cfg && a, "null pointer");
    int rc = inhand_reset_check(cfg, a);
    if (rc != MDP_OK) return rc;
    if (cfg->num_envs == 0) return MDP_OK;
    hipLaunchKernelGGL(k_inhand_reset_idx, dim3(grid_for(cfg->num_envs, 256, 1 << 20)), dim3(256), 0, as_stream(stream),
                       *cfg, *a);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}
