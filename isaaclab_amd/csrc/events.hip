// Event terms of the reset / interval modes as masked kernels: the reference calls them with
// an id list that comes from nonzero() on the host; here the selection is a byte mask on the device and nothing
// synchronises.  Reference arithmetic: LAB/envs/mdp/events.py:610-866, LAB/managers/event_manager.py:199-225,
// LAB/assets/articulation/articulation.py:316-341, 400-420, 472-508, LAB/utils/math.py:252-278, 463-499, 1331.
//
// One thread per env: an env that is not selected costs one byte read.  Random numbers are either injected (tests:
// the same uniforms go to the oracle) or drawn from Philox keyed by (seed, per-env epoch): the epoch word of an env
// advances only when that env draws, so a launch never races on a shared counter and a captured graph keeps drawing
// new numbers.
#include "mdp_device.hpp"
#include "mdp_host.hpp"
#include "uniforms_device.hpp"
#include "events_device.hpp"

using namespace mdp;

namespace {

// events.py:638-680 + articulation.py:316-341, 400-420
__global__ void k_reset_root_state_uniform(int N, mdp_event_root_state_t a) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N || (a.env_mask && !a.env_mask[e])) return;
    event_reset_root_state_env(a, e);
}

// events.py:802-866 + articulation.py:472-508, one thread per (env, joint).  The env's epoch word advances in the
// thread of joint 0; every thread of the env reads it first (the threads of one env sit in one block, barrier below).
__global__ void k_reset_joints(int N, mdp_event_joints_t a) {
    const int J = a.num_joints;
    const int epb = blockDim.x / J;                               // whole envs per block
    const int le = threadIdx.x / J, j = threadIdx.x - le * J;
    const int e = blockIdx.x * epb + le;
    const bool on = le < epb && e < N && (!a.env_mask || a.env_mask[e]);
    uint64_t seed = 0, epoch = 0;
    if (on && !a.injected_u) { seed = a.rng_state[0]; epoch = a.rng_state[1 + e]; }
    __syncthreads();
    if (!on) return;
    if (!a.injected_u && j == 0) a.rng_state[1 + e] = epoch + 1;
    event_reset_joint(a, e, j, seed, epoch);
}

// events.py:610-635 + articulation.py:400-420
__global__ void k_push_by_setting_velocity(int N, mdp_event_push_t a) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N || (a.env_mask && !a.env_mask[e])) return;
    event_push_env(a, e);
}

// event_manager.py:203-225, per-env timers
__global__ void k_interval_tick(int N, float* __restrict__ time_left, float dt, float lo, float scale,
                                const float* __restrict__ inj, uint64_t* rng_state, uint8_t* __restrict__ trigger) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    trigger[e] = event_interval_tick_env(time_left, dt, lo, scale, inj, rng_state, e, time_left[e]) ? 1 : 0;
}

// event_manager.py:209-216, one timer for all envs (is_global_time): a single block, the decision is taken once
__global__ void k_interval_tick_global(int N, float* time_left, float dt, float lo, float scale, const float* inj,
                                       uint64_t* rng_state, uint8_t* __restrict__ trigger) {
    __shared__ int s_fire;
    if (threadIdx.x == 0) {
        float t = time_left[0] - dt;
        bool fire = t < 1e-6f;
        if (fire) {
            Uniforms u = uniforms_for(inj, 1, rng_state, 0x45495647u, 0);
            t = u(0) * scale + lo;
        }
        time_left[0] = t;
        s_fire = fire ? 1 : 0;
    }
    __syncthreads();
    const uint8_t f = (uint8_t)s_fire;
    for (int e = threadIdx.x; e < N; e += blockDim.x) trigger[e] = f;
}

// With mean_level set the launch also returns what the term returns, torch.mean(terrain_levels.float()) over ALL envs
// (curriculums.py:55): every thread contributes its env's level (updated or not), blocks add up integers -- exact in any
// order -- and the last block to finish (a ticket) writes sum * (1 / N), torch's own expression, and clears the words.
__global__ void k_terrain_levels_vel(int N, mdp_curriculum_terrain_t a) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    long long level = 0;
    if (e < N) {
        if (a.env_mask && !a.env_mask[e]) level = a.mean_level ? (long long)a.terrain_levels[e] : 0;
        else level = (long long)terrain_levels_vel_env(e, a);
    }
    if (!a.mean_level) return;
    __shared__ long long s_sum[16];
    for (int off = 32; off > 0; off >>= 1) level += __shfl_down(level, off);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = level;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long block = 0;
        for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) block += s_sum[w];
        unsigned long long* ws = a.mean_ws;
        // the ticket's increment is computed from the value the first atomic returned (its top bit is never set): the
        // sum is in place before the ticket is taken, without a fence
        const unsigned long long before = atomicAdd(&ws[0], (unsigned long long)block);
        const unsigned long long ticket = atomicAdd(&ws[1], 1ull + (before >> 63));
        if (ticket == (unsigned long long)gridDim.x - 1) {
            const long long total = (long long)atomicExch(&ws[0], 0ull);
            atomicExch(&ws[1], 0ull);
            a.mean_level[0] = (float)total * (1.0f / (float)N);
        }
    }
}

// event_manager.py:235-240 for every reset term with min_step_count_between_reset == 0, one thread per env
__global__ void k_event_reset_mark(int N, const uint8_t* __restrict__ mask, mdp_event_reset_mark_t a) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N || (mask && !mask[e])) return;
    for (int t = 0; t < a.num_terms; ++t) {
        a.last_triggered_step[t][e] = a.step_count;
        a.triggered_once[t][e] = 1;
    }
}

// Imu._update_buffers_impl (LAB/sensors/imu/imu.py:142-182) for the outdated envs; one thread per env
__global__ void k_imu_update(int N, mdp_imu_args_t a) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N || (a.is_outdated && !a.is_outdated[e])) return;
    const float* t = a.transforms + 7 * (int64_t)e;            // x, y, z, qx, qy, qz, qw
    const V3 p{t[0], t[1], t[2]};
    const Q4 q{t[6], t[3], t[4], t[5]};
    const V3 off = ld3(a.offset_pos_b + 3 * (int64_t)e);
    const V3 r = quat_rotate(q, off);
    st3(a.pos_w + 3 * (int64_t)e, V3{p.x + r.x, p.y + r.y, p.z + r.z});
    const Q4 qs = quat_mul(q, ld4(a.offset_quat_b + 4 * (int64_t)e));
    st4(a.quat_w + 4 * (int64_t)e, qs);
    const float* v = a.velocities + 6 * (int64_t)e;
    V3 lv{v[0], v[1], v[2]};
    const V3 av{v[3], v[4], v[5]};
    // the velocity of the link's centre of mass, moved to the sensor: v + w x R (offset - com)
    const V3 com = ld3(a.coms + (int64_t)e * a.ld_coms);
    const V3 arm = quat_rotate(q, V3{off.x - com.x, off.y - com.y, off.z - com.z});
    const V3 c = cross(av, arm);
    lv = V3{lv.x + c.x, lv.y + c.y, lv.z + c.z};
    float* plv = a.prev_lin_vel_w + 3 * (int64_t)e;
    float* pav = a.prev_ang_vel_w + 3 * (int64_t)e;
    const float* g = a.gravity_bias_w + 3 * (int64_t)e;
    const V3 la{(lv.x - plv[0]) / a.dt + g[0], (lv.y - plv[1]) / a.dt + g[1], (lv.z - plv[2]) / a.dt + g[2]};
    const V3 aa{(av.x - pav[0]) / a.dt, (av.y - pav[1]) / a.dt, (av.z - pav[2]) / a.dt};
    st3(a.lin_vel_b + 3 * (int64_t)e, quat_rotate_inverse(qs, lv));
    st3(a.ang_vel_b + 3 * (int64_t)e, quat_rotate_inverse(qs, av));
    st3(a.lin_acc_b + 3 * (int64_t)e, quat_rotate_inverse(qs, la));
    st3(a.ang_acc_b + 3 * (int64_t)e, quat_rotate_inverse(qs, aa));
    st3(plv, lv);
    st3(pav, av);
    if (a.is_outdated) {
        a.is_outdated[e] = 0;
        if (a.timestamp_last_update && a.timestamp) a.timestamp_last_update[e] = a.timestamp[e];
    }
}

// FrameTransformer._update_buffers_impl (LAB/sensors/frame_transformer/frame_transformer.py:322-385): one thread per
// (env, target frame).  Like the reference, every env is recomputed as soon as one is outdated (`gate` = number of
// outdated envs, counted by the caller on the device); the bookkeeping is per env.
__global__ void k_frame_transformer_update(int N, int F, mdp_frame_transformer_args_t a) {
    if (a.gate != nullptr && *a.gate <= 0.0f) return;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * F) return;
    const int e = idx / F, f = idx - e * F;
    auto frame = [&](int64_t body) {
        const float* t = a.transforms + 7 * (int64_t)a.per_env_indices[body];
        return t;
    };
    // source frame (+ offset): combine_frame_transforms, math.py:749-781
    const float* ts = frame(a.source_frame_body_ids[e]);
    V3 sp{ts[0], ts[1], ts[2]};
    Q4 sq{ts[6], ts[3], ts[4], ts[5]};
    if (a.source_offset_pos) {
        const V3 r = quat_apply(sq, ld3(a.source_offset_pos + 3 * (int64_t)e));
        sp = V3{sp.x + r.x, sp.y + r.y, sp.z + r.z};
        sq = quat_mul(sq, ld4(a.source_offset_quat + 4 * (int64_t)e));
    }
    // target frame (+ offset)
    const float* tt = frame(a.target_frame_body_ids[a.duplicate_frame_indices[idx]]);
    V3 tp{tt[0], tt[1], tt[2]};
    Q4 tq{tt[6], tt[3], tt[4], tt[5]};
    if (a.target_offset_pos) {
        const V3 r = quat_apply(tq, ld3(a.target_offset_pos + 3 * (int64_t)idx));
        tp = V3{tp.x + r.x, tp.y + r.y, tp.z + r.z};
        tq = quat_mul(tq, ld4(a.target_offset_quat + 4 * (int64_t)idx));
    }
    // target in the source frame: subtract_frame_transforms, math.py:785-816 (quat_inv = normalised conjugate)
    const Q4 inv = normalize4(quat_conjugate(sq), 1e-9f);
    const Q4 rq = quat_mul(inv, tq);
    const V3 rp = quat_apply(inv, V3{tp.x - sp.x, tp.y - sp.y, tp.z - sp.z});
    if (f == 0) {
        st3(a.source_pos_w + 3 * (int64_t)e, sp);
        st4(a.source_quat_w + 4 * (int64_t)e, sq);
        if (a.is_outdated && a.is_outdated[e]) {
            a.is_outdated[e] = 0;
            if (a.timestamp && a.timestamp_last_update) a.timestamp_last_update[e] = a.timestamp[e];
        }
    }
    st3(a.target_pos_w + 3 * (int64_t)idx, tp);
    st4(a.target_quat_w + 4 * (int64_t)idx, tq);
    st3(a.target_pos_source + 3 * (int64_t)idx, rp);
    st4(a.target_quat_source + 4 * (int64_t)idx, rq);
}

}  // namespace

extern "C" int mdp_frame_transformer_update(int num_envs, int num_frames, const mdp_frame_transformer_args_t* a, void* stream) {
    MDP_REQUIRE(a, "null arguments");
    MDP_REQUIRE(a->transforms && a->per_env_indices && a->source_frame_body_ids && a->target_frame_body_ids &&
                    a->duplicate_frame_indices && a->source_pos_w && a->source_quat_w && a->target_pos_w &&
                    a->target_quat_w && a->target_pos_source && a->target_quat_source, "null pointer");
    MDP_REQUIRE((a->source_offset_pos == nullptr) == (a->source_offset_quat == nullptr), "source offset: give both or neither");
    MDP_REQUIRE((a->target_offset_pos == nullptr) == (a->target_offset_quat == nullptr), "target offset: give both or neither");
    if (num_envs <= 0 || num_frames <= 0) return MDP_OK;
    // one thread per (env, frame) pair, no grid stride: the grid must cover every pair
    const int64_t pairs = (int64_t)num_envs * num_frames;
    MDP_REQUIRE(pairs <= ((int64_t)1 << 30), "%lld (env, frame) pairs exceed one launch", (long long)pairs);
    hipLaunchKernelGGL(k_frame_transformer_update, dim3((unsigned)((pairs + 127) / 128)), dim3(128), 0,
                       as_stream(stream), num_envs, num_frames, *a);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_imu_update(int num_envs, const mdp_imu_args_t* a, void* stream) {
    MDP_REQUIRE(a, "null arguments");
    MDP_REQUIRE(a->transforms && a->velocities && a->coms && a->offset_pos_b && a->offset_quat_b && a->gravity_bias_w &&
                    a->prev_lin_vel_w && a->prev_ang_vel_w && a->pos_w && a->quat_w && a->lin_vel_b && a->ang_vel_b &&
                    a->lin_acc_b && a->ang_acc_b, "null pointer");
    MDP_REQUIRE(a->dt > 0.0f, "dt must be positive (the sensor needs one update() before its data is read)");
    MDP_REQUIRE(a->ld_coms >= 3, "coms rows must hold at least the position");
    if (num_envs <= 0) return MDP_OK;
    hipLaunchKernelGGL(k_imu_update, dim3((unsigned)((num_envs + 127) / 128)), dim3(128), 0, as_stream(stream), num_envs, *a);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_curriculum_terrain_levels_vel(int num_envs, const mdp_curriculum_terrain_t* a, void* stream) {
    MDP_REQUIRE(a, "null arguments");
    MDP_REQUIRE(a->root_state_w && a->command && a->terrain_levels && a->terrain_types && a->terrain_origins &&
                    a->env_origins, "null pointer");
    MDP_REQUIRE(a->ld_state >= 3 && a->max_terrain_level >= 1 && a->num_cols >= 1, "bad shape");
    MDP_REQUIRE(a->injected_levels || a->rng_state, "need rng_state or injected_levels");
    MDP_REQUIRE((a->mean_level == nullptr) == (a->mean_ws == nullptr), "mean_level and mean_ws go together");
    if (num_envs <= 0) return MDP_OK;
    hipLaunchKernelGGL(k_terrain_levels_vel, dim3(grid_for(num_envs, 128)), dim3(128), 0, as_stream(stream), num_envs, *a);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_event_reset_mark(int num_envs, const uint8_t* env_mask, const mdp_event_reset_mark_t* a, void* stream) {
    MDP_REQUIRE(a, "null arguments");
    MDP_REQUIRE(a->num_terms >= 0 && a->num_terms <= MDP_EVENT_MARK_MAX_TERMS, "at most %d terms per launch (got %d)",
                MDP_EVENT_MARK_MAX_TERMS, a->num_terms);
    for (int t = 0; t < a->num_terms; ++t)
        MDP_REQUIRE(a->last_triggered_step[t] && a->triggered_once[t], "term %d: null buffer", t);
    if (num_envs <= 0 || a->num_terms == 0) return MDP_OK;
    hipLaunchKernelGGL(k_event_reset_mark, dim3(grid_for(num_envs, 256, 1 << 20)), dim3(256), 0, as_stream(stream), num_envs,
                       env_mask, *a);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_event_reset_root_state_uniform(int num_envs, const mdp_event_root_state_t* a, void* stream) {
    MDP_REQUIRE(a, "null arguments");
    MDP_REQUIRE(a->default_root_state && a->env_origins && a->root_state_w && a->sim_root_transforms &&
                    a->sim_root_velocities, "null pointer");
    MDP_REQUIRE(a->injected_u || a->rng_state, "need rng_state or injected_u");
    MDP_REQUIRE(!a->body_acc_w || a->num_bodies > 0, "body_acc_w without num_bodies");
    if (num_envs <= 0) return MDP_OK;
    hipLaunchKernelGGL(k_reset_root_state_uniform, dim3(grid_for(num_envs, 128)), dim3(128), 0, as_stream(stream), num_envs, *a);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_event_reset_joints(int num_envs, const mdp_event_joints_t* a, void* stream) {
    MDP_REQUIRE(a, "null arguments");
    MDP_REQUIRE(a->mode == 0 || a->mode == 1, "mode must be 0 (by scale) or 1 (by offset)");
    MDP_REQUIRE(a->num_joints > 0 && a->num_joints <= 128, "num_joints out of range");
    MDP_REQUIRE(a->default_joint_pos && a->default_joint_vel && a->soft_joint_pos_limits && a->soft_joint_vel_limits &&
                    a->joint_pos && a->joint_vel && a->prev_joint_vel && a->joint_acc && a->sim_dof_positions &&
                    a->sim_dof_velocities, "null pointer");
    MDP_REQUIRE(a->ld_sim >= a->num_joints, "ld_sim < num_joints");
    MDP_REQUIRE(a->injected_u || a->rng_state, "need rng_state or injected_u");
    if (num_envs <= 0) return MDP_OK;
    const int epb = 256 / a->num_joints;                          // num_joints <= 128: at least two envs per block
    hipLaunchKernelGGL(k_reset_joints, dim3((unsigned)((num_envs + epb - 1) / epb)), dim3(256), 0, as_stream(stream), num_envs, *a);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_event_push_by_setting_velocity(int num_envs, const mdp_event_push_t* a, void* stream) {
    MDP_REQUIRE(a, "null arguments");
    MDP_REQUIRE(a->root_state_w && a->sim_root_velocities, "null pointer");
    MDP_REQUIRE(a->injected_u || a->rng_state, "need rng_state or injected_u");
    MDP_REQUIRE(!a->body_acc_w || a->num_bodies > 0, "body_acc_w without num_bodies");
    if (num_envs <= 0) return MDP_OK;
    hipLaunchKernelGGL(k_push_by_setting_velocity, dim3(grid_for(num_envs, 128)), dim3(128), 0, as_stream(stream), num_envs, *a);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_event_interval_tick(int num_envs, float* time_left, float dt, float lo, float scale, int is_global_time,
                                       const float* injected_u, uint64_t* rng_state, uint8_t* trigger_mask, void* stream) {
    MDP_REQUIRE(time_left && trigger_mask, "null pointer");
    MDP_REQUIRE(injected_u || rng_state, "need rng_state or injected_u");
    if (num_envs <= 0) return MDP_OK;
    if (is_global_time)
        hipLaunchKernelGGL(k_interval_tick_global, dim3(1), dim3(1024), 0, as_stream(stream), num_envs, time_left, dt, lo,
                           scale, injected_u, rng_state, trigger_mask);
    else
        hipLaunchKernelGGL(k_interval_tick, dim3(grid_for(num_envs, 256)), dim3(256), 0, as_stream(stream), num_envs,
                           time_left, dt, lo, scale, injected_u, rng_state, trigger_mask);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}
