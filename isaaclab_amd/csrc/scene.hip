// State-container derivations, sensor bookkeeping, contact sensor state machine, action processing
// and the uniform velocity command generator.
// Reference: LAB/assets/articulation/articulation_data.py, LAB/sensors/sensor_base.py,
// LAB/sensors/contact_sensor/contact_sensor.py, LAB/envs/mdp/actions/joint_actions.py,
// LAB/envs/mdp/commands/velocity_command.py, LAB/managers/command_manager.py
#include "mdp_device.hpp"
#include "mdp_host.hpp"
#include "scene_update_device.hpp"
#include "command_device.hpp"
#include "joint_action_device.hpp"

using namespace mdp;

namespace {

// articulation_data.py:446-455
__global__ void k_joint_acc(int64_t total, int J, const float* __restrict__ qd, int ld,
                            float* __restrict__ prev, float* __restrict__ acc, float elapsed) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t e = i / J;
        int j = (int)(i - e * J);
        float v = qd[e * ld + j];
        acc[i] = (v - prev[i]) / elapsed;
        prev[i] = v;
    }
}

// articulation_data.py:264-279: cat(pose with xyzw -> wxyz, velocity)
__global__ void k_state_pack(int64_t count, const float* __restrict__ pose, const float* __restrict__ vel,
                             float* __restrict__ out) {
    int64_t total = count * 13;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / 13;
        int k = (int)(i - r * 13);
        float v;
        if (k < 3) v = pose[7 * r + k];
        else if (k == 3) v = pose[7 * r + 6];       // w
        else if (k < 7) v = pose[7 * r + k - 1];    // x, y, z
        else v = vel[6 * r + (k - 7)];
        out[i] = v;
    }
}

// the same for `num_ids` of the B bodies of every env only (rows of the other bodies are left as they are)
__global__ void k_state_pack_rows(int64_t num_envs, int B, const int32_t* __restrict__ ids, int num_ids,
                                  const float* __restrict__ pose, const float* __restrict__ vel, float* __restrict__ out) {
    const int64_t total = num_envs * num_ids * 13;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / 13;
        const int k = (int)(i - row * 13);
        const int64_t e = row / num_ids;
        const int64_t r = e * B + ids[(int)(row - e * num_ids)];
        float v;
        if (k < 3) v = pose[7 * r + k];
        else if (k == 3) v = pose[7 * r + 6];       // w
        else if (k < 7) v = pose[7 * r + k - 1];    // x, y, z
        else v = vel[6 * r + (k - 7)];
        out[13 * r + k] = v;
    }
}

// articulation_data.py:412-425,503-518
__global__ void k_root_derive(int N, const float* __restrict__ st, int ld, float gx, float gy, float gz,
                              float* __restrict__ lin_b, float* __restrict__ ang_b, float* __restrict__ grav_b,
                              float* __restrict__ heading) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const float* r = st + (int64_t)e * ld;
    Q4 q = ld4(r + 3);
    if (lin_b) st3(lin_b + 3 * (int64_t)e, quat_rotate_inverse(q, ld3(r + 7)));
    if (ang_b) st3(ang_b + 3 * (int64_t)e, quat_rotate_inverse(q, ld3(r + 10)));
    if (grav_b) st3(grav_b + 3 * (int64_t)e, quat_rotate_inverse(q, V3{gx, gy, gz}));
    if (heading) {
        V3 f = quat_apply(q, V3{1.0f, 0.0f, 0.0f});
        heading[e] = atan2f(f.y, f.x);
    }
}

// articulation_data.py:288-302 / 355-370: v_link = v_com + w x R(q)(-com_b)
__global__ void k_link_from_com(int64_t count, const float* __restrict__ in, const float* __restrict__ com,
                                float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float* r = in + 13 * i;
        float* o = out + 13 * i;
        Q4 q = ld4(r + 3);
        V3 c = ld3(com + 3 * i);
        V3 off = quat_rotate(q, V3{-c.x, -c.y, -c.z});
        V3 w = ld3(r + 10);
        V3 cr = cross(w, off);
        for (int k = 0; k < 7; ++k) o[k] = r[k];
        o[7] = r[7] + cr.x;
        o[8] = r[8] + cr.y;
        o[9] = r[9] + cr.z;
        o[10] = w.x;
        o[11] = w.y;
        o[12] = w.z;
    }
}

// sensor_base.py:197-205 + contact_sensor.py:314-373.
// A block owns blockDim/B whole environments, so the per-env bookkeeping words (timestamp,
// last-update, outdated flag) are read by all B threads of an env, then -- after a block barrier --
// committed by the env's b == 0 thread: one launch, no inter-block hand-off.
template <bool COMMIT>
__global__ void k_contact_update(int N, int B, int H, const float* __restrict__ forces,
                                 float* __restrict__ net, float* __restrict__ hist,
                                 float* __restrict__ last_air, float* __restrict__ cur_air,
                                 float* __restrict__ last_con, float* __restrict__ cur_con,
                                 float* __restrict__ ts, float* __restrict__ ts_last, uint8_t* __restrict__ outdated,
                                 float dt, float period, float thresh, int eager) {
    if (COMMIT) {
        int epb = blockDim.x / B;  // whole envs per block (host guarantees B <= blockDim)
        int le = threadIdx.x / B;
        int b = threadIdx.x - le * B;
        for (int base = blockIdx.x * epb; base < N; base += gridDim.x * epb) {
            int e = base + le;
            bool active = (le < epb) && (e < N);
            float t_new = 0.0f, t_last = 0.0f;
            bool od = false;
            if (active) {
                t_new = ts[e] + dt;
                t_last = ts_last[e];
                od = (outdated[e] != 0) || ((t_new - t_last) + 1e-6f >= period);
            }
            __syncthreads();
            if (active) {
                if (od && eager)
                    contact_body_update((int64_t)e * B + b, e, b, B, H, forces, net, hist, last_air, cur_air,
                                        last_con, cur_con, t_new - t_last, thresh);
                if (b == 0) {
                    ts[e] = t_new;
                    if (od && eager) {
                        ts_last[e] = t_new;
                        od = false;
                    }
                    outdated[e] = od ? 1 : 0;
                }
            }
        }
    } else {
        int64_t total = (int64_t)N * B;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
             i += (int64_t)gridDim.x * blockDim.x) {
            int e = (int)(i / B);
            int b = (int)(i - (int64_t)e * B);
            float t_new = ts[e] + dt;
            float t_last = ts_last[e];
            bool od = (outdated[e] != 0) || ((t_new - t_last) + 1e-6f >= period);
            if (od && eager)
                contact_body_update(i, e, b, B, H, forces, net, hist, last_air, cur_air, last_con, cur_con,
                                    t_new - t_last, thresh);
        }
    }
}

// second phase: commit the per-env bookkeeping (separate kernel => no intra-kernel race)
__global__ void k_sensor_commit(int N, float* __restrict__ ts, float* __restrict__ ts_last,
                                uint8_t* __restrict__ outdated, float dt, float period, int eager) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    float t_new = ts[e] + dt;
    bool od = outdated[e] != 0;
    od = od || ((t_new - ts_last[e]) + 1e-6f >= period);
    ts[e] = t_new;
    if (od && eager) {
        ts_last[e] = t_new;
        od = false;
    }
    outdated[e] = od ? 1 : 0;
}

__global__ void k_contact_reset(int N, int B, int H, float* net, float* hist, float* last_air, float* cur_air,
                                float* last_con, float* cur_con, float* ts, float* ts_last, uint8_t* outdated,
                                const uint8_t* __restrict__ mask) {
    int64_t total = (int64_t)N * B;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int e = (int)(i / B);
        int b = (int)(i - (int64_t)e * B);
        if (mask && !mask[e]) continue;
        if (net) st3(net + 3 * i, V3{0, 0, 0});
        if (hist)
            for (int s = 0; s < H; ++s) st3(hist + (((int64_t)e * H + s) * B + b) * 3, V3{0, 0, 0});
        if (cur_air) {
            cur_air[i] = 0.0f;
            last_air[i] = 0.0f;
            cur_con[i] = 0.0f;
            last_con[i] = 0.0f;
        }
        if (b == 0) {
            ts[e] = 0.0f;
            ts_last[e] = 0.0f;
            outdated[e] = 1;
        }
    }
}

__global__ void k_sensor_tick(int N, float* __restrict__ ts, const float* __restrict__ ts_last,
                              uint8_t* __restrict__ outdated, float dt, float period) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    float t = ts[e] + dt;
    ts[e] = t;
    if ((t - ts_last[e]) + 1e-6f >= period) outdated[e] = 1;
}

// action_manager.py:331-332 + joint_actions.py:130-139
__global__ void k_joint_action(int64_t total, int A, mdp_joint_action_t a) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t e = i / A;
        joint_action_element(a, e, (int)(i - e * A), A, true);
    }
}

__global__ void k_velocity_command(mdp_velocity_command_cfg_t cfg, int N, const float* __restrict__ st, int ld,
                                   float* __restrict__ cmd, float* __restrict__ heading_target,
                                   uint8_t* __restrict__ is_heading, uint8_t* __restrict__ is_standing,
                                   float* __restrict__ time_left, int64_t* __restrict__ counter,
                                   float* __restrict__ m_xy, float* __restrict__ m_yaw, float dt,
                                   uint64_t* rng, const float* __restrict__ inj,
                                   const uint8_t* __restrict__ force) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    // generator words: {seed, epoch, ticket}.  Every block reads the epoch before it takes a
    // ticket; the block that draws the last ticket advances the epoch for the next launch
    // (stream order makes it visible), so hipGraph replays never repeat random numbers.
    uint64_t seed = 0, epoch = 0;
    if (rng) { seed = rng[0]; epoch = rng[1]; }
    if (e < N) velocity_command_env(cfg, e, st, ld, cmd, heading_target, is_heading, is_standing, time_left, counter,
                                    m_xy, m_yaw, dt, seed, epoch, inj, force);
    if (rng && !inj) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            unsigned long long t = atomicAdd((unsigned long long*)&rng[2], 1ull);
            if (t == (unsigned long long)gridDim.x - 1ull) {
                rng[2] = 0;
                rng[1] = epoch + 1;
            }
        }
    }
}

// interactive_scene.py:496-513 in one launch (body in scene_update_device.hpp, shared with the actuator kernels that
// carry the update of the previous physics step along)
__global__ void __launch_bounds__(256)
k_scene_update(mdp_scene_update_t d, float dt, unsigned nb_root, unsigned nb_joint, unsigned nb_contact) {
    scene_update_block(d, dt, blockIdx.x, nb_root, nb_joint, nb_contact);
}

}  // namespace

extern "C" int mdp_joint_acc_update(int num_envs, int num_joints, const float* joint_vel, int ld_vel,
                                    float* prev_joint_vel, float* joint_acc, float elapsed, void* stream) {
    MDP_REQUIRE(joint_vel && prev_joint_vel && joint_acc, "null pointer");
    int64_t total = (int64_t)num_envs * num_joints;
    if (total == 0) return MDP_OK;
    hipLaunchKernelGGL(k_joint_acc, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), total, num_joints,
                       joint_vel, ld_vel, prev_joint_vel, joint_acc, elapsed);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_state_pack(int64_t count, const float* pose_xyzw, const float* velocity, float* state_out,
                              void* stream) {
    MDP_REQUIRE(pose_xyzw && velocity && state_out, "null pointer");
    if (count == 0) return MDP_OK;
    hipLaunchKernelGGL(k_state_pack, dim3(grid_for(count * 13, 256)), dim3(256), 0, as_stream(stream), count,
                       pose_xyzw, velocity, state_out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_state_pack_rows(int64_t num_envs, int num_bodies, const int32_t* body_ids, int num_ids,
                                   const float* pose_xyzw, const float* velocity, float* state_out, void* stream) {
    MDP_REQUIRE(body_ids && pose_xyzw && velocity && state_out, "null pointer");
    MDP_REQUIRE(num_bodies > 0 && num_ids >= 0 && num_ids <= num_bodies, "bad body counts (%d of %d)", num_ids, num_bodies);
    if (num_envs <= 0 || num_ids == 0) return MDP_OK;
    hipLaunchKernelGGL(k_state_pack_rows, dim3(grid_for(num_envs * num_ids * 13, 256)), dim3(256), 0, as_stream(stream),
                       num_envs, num_bodies, body_ids, num_ids, pose_xyzw, velocity, state_out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_root_derive(int num_envs, const float* root_state, int ld_state, float gx, float gy, float gz,
                               float* root_lin_vel_b, float* root_ang_vel_b, float* projected_gravity_b,
                               float* heading_w, void* stream) {
    MDP_REQUIRE(root_state && ld_state >= 13, "root_state must be (N,13)");
    if (num_envs == 0) return MDP_OK;
    hipLaunchKernelGGL(k_root_derive, dim3(grid_for(num_envs, 256, 1 << 20)), dim3(256), 0, as_stream(stream),
                       num_envs, root_state, ld_state, gx, gy, gz, root_lin_vel_b, root_ang_vel_b,
                       projected_gravity_b, heading_w);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_link_state_from_com(int64_t count, const float* state_in, const float* com_pos_b,
                                       float* state_out, void* stream) {
    MDP_REQUIRE(state_in && com_pos_b && state_out, "null pointer");
    if (count == 0) return MDP_OK;
    hipLaunchKernelGGL(k_link_from_com, dim3(grid_for(count, 256)), dim3(256), 0, as_stream(stream), count, state_in,
                       com_pos_b, state_out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_contact_sensor_update(int num_envs, int num_bodies, int history_len,
                                         const float* net_contact_forces, float* net_forces_w,
                                         float* net_forces_w_history, float* last_air_time, float* current_air_time,
                                         float* last_contact_time, float* current_contact_time, float* timestamp,
                                         float* timestamp_last_update, uint8_t* is_outdated, float dt,
                                         float update_period, float force_threshold, int eager, void* stream) {
    MDP_REQUIRE(net_contact_forces && net_forces_w && timestamp && timestamp_last_update && is_outdated,
                "null pointer");
    MDP_REQUIRE((current_air_time == nullptr) ==
                    (last_air_time == nullptr && last_contact_time == nullptr && current_contact_time == nullptr) ||
                    (current_air_time && last_air_time && last_contact_time && current_contact_time),
                "air/contact timers must be all set or all null");
    if (num_envs == 0 || num_bodies == 0) return MDP_OK;
    hipStream_t s = as_stream(stream);
    int64_t total = (int64_t)num_envs * num_bodies;
    if (num_bodies <= 256) {
        int epb = 256 / num_bodies;
        int64_t blocks = ((int64_t)num_envs + epb - 1) / epb;
        hipLaunchKernelGGL(k_contact_update<true>, dim3(grid_for(blocks, 1)), dim3(256), 0, s, num_envs, num_bodies,
                           history_len, net_contact_forces, net_forces_w, net_forces_w_history, last_air_time,
                           current_air_time, last_contact_time, current_contact_time, timestamp,
                           timestamp_last_update, is_outdated, dt, update_period, force_threshold, eager);
    } else {
        hipLaunchKernelGGL(k_contact_update<false>, dim3(grid_for(total, 256)), dim3(256), 0, s, num_envs,
                           num_bodies, history_len, net_contact_forces, net_forces_w, net_forces_w_history,
                           last_air_time, current_air_time, last_contact_time, current_contact_time, timestamp,
                           timestamp_last_update, is_outdated, dt, update_period, force_threshold, eager);
        hipLaunchKernelGGL(k_sensor_commit, dim3(grid_for(num_envs, 256, 1 << 20)), dim3(256), 0, s, num_envs,
                           timestamp, timestamp_last_update, is_outdated, dt, update_period, eager);
    }
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_contact_sensor_reset(int num_envs, int num_bodies, int history_len, float* net_forces_w,
                                        float* net_forces_w_history, float* last_air_time, float* current_air_time,
                                        float* last_contact_time, float* current_contact_time, float* timestamp,
                                        float* timestamp_last_update, uint8_t* is_outdated, const uint8_t* env_mask,
                                        void* stream) {
    MDP_REQUIRE(timestamp && timestamp_last_update && is_outdated, "null pointer");
    if (num_envs == 0) return MDP_OK;
    int B = num_bodies < 1 ? 1 : num_bodies;
    int64_t total = (int64_t)num_envs * B;
    hipLaunchKernelGGL(k_contact_reset, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), num_envs, B,
                       history_len, num_bodies ? net_forces_w : nullptr, num_bodies ? net_forces_w_history : nullptr,
                       num_bodies ? last_air_time : nullptr, num_bodies ? current_air_time : nullptr,
                       last_contact_time, current_contact_time, timestamp, timestamp_last_update, is_outdated,
                       env_mask);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_sensor_tick(int num_envs, float* timestamp, const float* timestamp_last_update,
                               uint8_t* is_outdated, float dt, float update_period, void* stream) {
    MDP_REQUIRE(timestamp && timestamp_last_update && is_outdated, "null pointer");
    if (num_envs == 0) return MDP_OK;
    hipLaunchKernelGGL(k_sensor_tick, dim3(grid_for(num_envs, 256, 1 << 20)), dim3(256), 0, as_stream(stream),
                       num_envs, timestamp, timestamp_last_update, is_outdated, dt, update_period);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_joint_action_process(int num_envs, int action_dim, const float* actions_in, int ld_in,
                                        float* action, float* prev_action, float* processed,
                                        const float* scale_vec, float scale, const float* offset, const float* clip,
                                        void* stream) {
    MDP_REQUIRE(actions_in && action, "null pointer");
    MDP_REQUIRE(ld_in >= action_dim, "Invalid action shape, expected: %d, received: %d.", action_dim, ld_in);
    int64_t total = (int64_t)num_envs * action_dim;
    if (total == 0) return MDP_OK;
    mdp_joint_action_t a;
    std::memset(&a, 0, sizeof(a));
    a.actions_in = actions_in; a.ld_in = ld_in; a.action = action; a.prev_action = prev_action; a.processed = processed;
    a.scale_vec = scale_vec; a.scale = scale; a.offset = offset; a.clip = clip;
    hipLaunchKernelGGL(k_joint_action, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), total, action_dim, a);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_velocity_command_compute(const mdp_velocity_command_cfg_t* cfg, int num_envs,
                                            const float* root_state, int ld_state, float* vel_command_b,
                                            float* heading_target, uint8_t* is_heading_env, uint8_t* is_standing_env,
                                            float* time_left, int64_t* command_counter, float* metric_error_vel_xy,
                                            float* metric_error_vel_yaw, float dt, uint64_t* rng_state,
                                            const float* injected_u, const uint8_t* force_mask, void* stream) {
    MDP_REQUIRE(cfg && root_state && vel_command_b && heading_target && is_heading_env && is_standing_env &&
                    time_left && command_counter && metric_error_vel_xy && metric_error_vel_yaw,
                "null pointer");
    MDP_REQUIRE(rng_state || injected_u, "need rng_state or injected_u");
    MDP_REQUIRE(ld_state >= 13, "root_state must be (N,13)");
    if (num_envs == 0) return MDP_OK;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_velocity_command, dim3(grid_for(num_envs, 256, 1 << 20)), dim3(256), 0, s, *cfg, num_envs,
                       root_state, ld_state, vel_command_b, heading_target, is_heading_env, is_standing_env,
                       time_left, command_counter, metric_error_vel_xy, metric_error_vel_yaw, dt, rng_state,
                       injected_u, force_mask);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_scene_update(const mdp_scene_update_t* desc, float dt, void* stream) {
    SceneUpdatePlan plan;
    int rc = scene_update_plan(desc, &plan);
    if (rc != MDP_OK) return rc;
    if (plan.total == 0) return MDP_OK;
    hipLaunchKernelGGL(k_scene_update, dim3(plan.total), dim3(256), 0, as_stream(stream), plan.d, dt, plan.nb_root,
                       plan.nb_joint, plan.nb_contact);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}
