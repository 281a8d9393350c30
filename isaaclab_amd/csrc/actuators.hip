// Explicit / implicit PD actuator models, delay buffers and the remotized-PD torque LUT.
// Reference: LAB/actuators/actuator_pd.py, LAB/utils/buffers/*.py, LAB/utils/interpolation/*.py
#include "mdp_device.hpp"
#include "mdp_host.hpp"
#include "scene_update_device.hpp"
#include "joint_action_device.hpp"

using namespace mdp;

namespace {

// One thread per (env, joint) element, flat index -> perfectly coalesced on contiguous (N,J).
// HBM-bound: reads q,qd,target(+gains), writes computed & applied.
// The first gridDim.x - nb_act blocks carry the scene update of the frame this launch reads (scene.update(k) + write_data_to_sim(k+1) in one
// launch; nb_act == gridDim.x when there is none).
template <int KIND>
__global__ void __launch_bounds__(256)
k_actuator_pd(mdp_scene_update_t upd, float dt, unsigned nb_act, unsigned nb_root, unsigned nb_joint, unsigned nb_contact, int upd_envs,
              mdp_joint_action_t act, int64_t total, int J,
                              const float* __restrict__ q_des, int ld_q_des,
                              const float* __restrict__ qd_des, int ld_qd_des,
                              const float* __restrict__ tau_ff, int ld_tau_ff,
                              const float* __restrict__ q, int ld_q,
                              const float* __restrict__ qd, int ld_qd,
                              const float* __restrict__ kp, const float* __restrict__ kd,
                              const float* __restrict__ eff_lim, const float* __restrict__ vel_lim,
                              float sat, float* __restrict__ computed, float* __restrict__ applied,
                              const float* __restrict__ gate) {
    if (gate != nullptr && *gate <= 0.0f) return;   // mdp_actuator_pd_compute_gated: nothing happens this launch
    const unsigned nb_upd = gridDim.x - nb_act;   // carried scene update: the first blocks of the grid
    if (blockIdx.x < nb_upd) {
        scene_update_carried(upd, dt, blockIdx.x, nb_root, nb_joint, nb_contact, upd_envs);
        return;
    }
    for (int64_t i = (blockIdx.x - nb_upd) * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)nb_act * blockDim.x) {
        int64_t e = i / J;
        int j = (int)(i - e * J);
        float qv = q[e * ld_q + j];
        float qdv = qd[e * ld_qd + j];
        // act.actions_in: this launch also processes the action of the step (its processed value is q_des)
        float des = act.actions_in ? joint_action_element(act, e, j, J, true)
                                   : (q_des ? q_des[e * ld_q_des + j] : 0.0f);
        float ddes = qd_des ? qd_des[e * ld_qd_des + j] : 0.0f;
        float ff = tau_ff ? tau_ff[e * ld_tau_ff + j] : 0.0f;
        // actuator_pd.py:135-138
        float error_pos = des - qv;
        float error_vel = ddes - qdv;
        float tau = (kp[i] * error_pos + kd[i] * error_vel) + ff;
        float lim = eff_lim[i];
        float out;
        if (KIND == MDP_ACT_DC_MOTOR) {
            // actuator_pd.py:223-233
            float vl = vel_lim[i];
            float max_eff = sat * (1.0f - qdv / vl);
            max_eff = tclamp(max_eff, 0.0f, lim);
            float min_eff = sat * (-1.0f - qdv / vl);
            min_eff = tclamp(min_eff, -lim, 0.0f);
            out = tclamp(tau, min_eff, max_eff);
        } else {
            // actuator_base.py:252-261
            out = tclamp(tau, -lim, lim);
        }
        computed[i] = tau;
        applied[i] = out;
    }
}

// circular_buffer.py:107-135 append (+ delay_buffer.py:159-177 delayed read), one thread per (env, d)
__global__ void k_cbuf_append(float* __restrict__ buffer, int64_t* __restrict__ num_pushes, int L, int N, int D,
                              int pointer, const float* __restrict__ data, int ld,
                              const int32_t* __restrict__ lags, float* __restrict__ out) {
    int64_t total = (int64_t)N * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int e = (int)(i / D);
        int d = (int)(i - (int64_t)e * D);
        float v = data[(int64_t)e * ld + d];
        int64_t pushes = num_pushes[e];
        if (pushes == 0) {
            for (int s = 0; s < L; ++s) buffer[((int64_t)s * N + e) * D + d] = v;
        } else {
            buffer[((int64_t)pointer * N + e) * D + d] = v;
        }
        if (out) {
            int64_t lag = lags ? (int64_t)lags[e] : 0;
            int64_t valid = lag < pushes ? lag : pushes;  // min(lag, (pushes+1)-1)
            int idx = (int)(((int64_t)pointer - valid) % L);
            if (idx < 0) idx += L;
            // the slot just written may be the one read back
            float r = (idx == pointer || pushes == 0) ? v : buffer[((int64_t)idx * N + e) * D + d];
            out[i] = r;
        }
    }
}
__global__ void k_cbuf_inc(int64_t* num_pushes, int N) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < N) num_pushes[e] += 1;
}

__global__ void k_cbuf_read(const float* __restrict__ buffer, int L, int N, int D, int pointer,
                            float* __restrict__ out) {
    int64_t total = (int64_t)N * L * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int d = (int)(i % D);
        int64_t t = i / D;
        int s = (int)(t % L);  // output slot (0 = oldest)
        int e = (int)(t / L);
        // roll by (L - pointer - 1): out slot s <- buffer slot (s + pointer + 1) mod L
        int src = (s + pointer + 1) % L;
        out[i] = buffer[((int64_t)src * N + e) * D + d];
    }
}

__global__ void k_cbuf_reset(float* __restrict__ buffer, int64_t* __restrict__ num_pushes, int L, int N, int D,
                             const uint8_t* __restrict__ mask) {
    int64_t total = (int64_t)N * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int e = (int)(i / D);
        int d = (int)(i - (int64_t)e * D);
        if (mask && !mask[e]) continue;
        if (buffer)
            for (int s = 0; s < L; ++s) buffer[((int64_t)s * N + e) * D + d] = 0.0f;
        if (d == 0) num_pushes[e] = 0;
    }
}

// linear_interpolation.py:54-85 (count of x < q by linear scan: S is small, the LUT sits in L1/K$)
__global__ void k_linear_interp(const float* __restrict__ x, const float* __restrict__ y, int S,
                                const float* __restrict__ q, int64_t count, float* __restrict__ fq,
                                float* __restrict__ efforts) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        float qv = q[i];
        int smaller = 0;
        for (int s = 0; s < S; ++s) smaller += (x[s] < qv) ? 1 : 0;
        int lb = smaller - 1 < 0 ? 0 : smaller - 1;
        int ub = smaller > S - 1 ? S - 1 : smaller;
        float w = (qv - x[lb]) / (x[ub] - x[lb]);
        if (ub == lb) w = 0.0f;
        float f = y[lb] + w * (y[ub] - y[lb]);
        if (fq) fq[i] = f;
        if (efforts) efforts[i] = tclamp(efforts[i], -f, f);
    }
}

}  // namespace

static int pd_launch(const mdp_joint_action_t* action, const mdp_scene_update_t* update, float dt, int kind, int num_envs,
                                                    int num_joints, const float* q_des, int ld_q_des,
                                                    const float* qd_des, int ld_qd_des, const float* tau_ff,
                                                    int ld_tau_ff, const float* joint_pos, int ld_pos,
                                                    const float* joint_vel, int ld_vel, const float* stiffness,
                                                    const float* damping, const float* effort_limit,
                                                    const float* velocity_limit, float saturation_effort,
                                                    float* computed_effort, float* applied_effort, void* stream,
                                                    const float* gate = nullptr) {
    MDP_REQUIRE(num_envs >= 0 && num_joints >= 0, "negative shape (%d,%d)", num_envs, num_joints);
    SceneUpdatePlan plan;
    if (update) {
        int rc = scene_update_plan(update, &plan);
        if (rc != MDP_OK) return rc;
    } else {
        std::memset(&plan.d, 0, sizeof(plan.d));
    }
    if (num_envs == 0 || num_joints == 0) {
        if (update) return mdp_scene_update(update, dt, stream);
        return MDP_OK;
    }
    MDP_REQUIRE(joint_pos && joint_vel && stiffness && damping && effort_limit && computed_effort && applied_effort,
                "mdp_actuator_pd_compute: null required pointer");
    MDP_REQUIRE(ld_pos >= num_joints && ld_vel >= num_joints, "row stride smaller than num_joints");
    MDP_REQUIRE(kind != MDP_ACT_DC_MOTOR || velocity_limit, "DC motor needs velocity_limit");
    mdp_joint_action_t act;
    std::memset(&act, 0, sizeof(act));
    if (action) {
        MDP_REQUIRE(action->actions_in && action->action && action->processed, "joint action: null pointer");
        MDP_REQUIRE(action->ld_in >= num_joints, "Invalid action shape, expected: %d, received: %d.", num_joints,
                    action->ld_in);
        act = *action;
    }
    int64_t total = (int64_t)num_envs * num_joints;
    const int block = 256;
    unsigned nb_act = grid_for(total, block);
    unsigned grid = nb_act + plan.carried_total;
    hipStream_t s = as_stream(stream);
#define LAUNCH(K)                                                                                              \
    hipLaunchKernelGGL(k_actuator_pd<K>, dim3(grid), dim3(block), 0, s, plan.d, dt, nb_act, plan.nb_root,      \
                       plan.nb_joint, plan.nb_contact, plan.carried_envs, act, total, num_joints, q_des, ld_q_des, qd_des, ld_qd_des,   \
                       tau_ff, ld_tau_ff, joint_pos, ld_pos, joint_vel, ld_vel, stiffness, damping,             \
                       effort_limit, velocity_limit, saturation_effort, computed_effort, applied_effort, gate)
    switch (kind) {
        case MDP_ACT_IMPLICIT:
        case MDP_ACT_IDEAL_PD: LAUNCH(MDP_ACT_IDEAL_PD); break;
        case MDP_ACT_DC_MOTOR: LAUNCH(MDP_ACT_DC_MOTOR); break;
        default: return fail(MDP_ERR_INVALID, "unknown actuator kind %d", kind);
    }
#undef LAUNCH
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_actuator_pd_compute_after_update(const mdp_scene_update_t* update, float dt, int kind, int num_envs,
                                                    int num_joints, const float* q_des, int ld_q_des,
                                                    const float* qd_des, int ld_qd_des, const float* tau_ff,
                                                    int ld_tau_ff, const float* joint_pos, int ld_pos,
                                                    const float* joint_vel, int ld_vel, const float* stiffness,
                                                    const float* damping, const float* effort_limit,
                                                    const float* velocity_limit, float saturation_effort,
                                                    float* computed_effort, float* applied_effort, void* stream) {
    return pd_launch(nullptr, update, dt, kind, num_envs, num_joints, q_des, ld_q_des, qd_des, ld_qd_des, tau_ff, ld_tau_ff,
                     joint_pos, ld_pos, joint_vel, ld_vel, stiffness, damping, effort_limit, velocity_limit,
                     saturation_effort, computed_effort, applied_effort, stream);
}

extern "C" int mdp_actuator_pd_compute_with_action(const mdp_joint_action_t* action, int kind, int num_envs,
                                                   int num_joints, const float* qd_des, int ld_qd_des,
                                                   const float* tau_ff, int ld_tau_ff, const float* joint_pos, int ld_pos,
                                                   const float* joint_vel, int ld_vel, const float* stiffness,
                                                   const float* damping, const float* effort_limit,
                                                   const float* velocity_limit, float saturation_effort,
                                                   float* computed_effort, float* applied_effort, void* stream) {
    MDP_REQUIRE(action, "null joint action");
    return pd_launch(action, nullptr, 0.0f, kind, num_envs, num_joints, action->processed, num_joints, qd_des, ld_qd_des,
                     tau_ff, ld_tau_ff, joint_pos, ld_pos, joint_vel, ld_vel, stiffness, damping, effort_limit,
                     velocity_limit, saturation_effort, computed_effort, applied_effort, stream);
}

extern "C" int mdp_actuator_pd_compute(int kind, int num_envs, int num_joints,
                                       const float* q_des, int ld_q_des, const float* qd_des, int ld_qd_des,
                                       const float* tau_ff, int ld_tau_ff, const float* joint_pos, int ld_pos,
                                       const float* joint_vel, int ld_vel, const float* stiffness,
                                       const float* damping, const float* effort_limit,
                                       const float* velocity_limit, float saturation_effort,
                                       float* computed_effort, float* applied_effort, void* stream) {
    return mdp_actuator_pd_compute_after_update(nullptr, 0.0f, kind, num_envs, num_joints, q_des, ld_q_des, qd_des,
                                                ld_qd_des, tau_ff, ld_tau_ff, joint_pos, ld_pos, joint_vel, ld_vel,
                                                stiffness, damping, effort_limit, velocity_limit, saturation_effort,
                                                computed_effort, applied_effort, stream);
}

extern "C" int mdp_actuator_pd_compute_gated(const float* gate, int kind, int num_envs, int num_joints,
                                             const float* q_des, int ld_q_des, const float* qd_des, int ld_qd_des,
                                             const float* tau_ff, int ld_tau_ff, const float* joint_pos, int ld_pos,
                                             const float* joint_vel, int ld_vel, const float* stiffness,
                                             const float* damping, const float* effort_limit,
                                             const float* velocity_limit, float saturation_effort,
                                             float* computed_effort, float* applied_effort, void* stream) {
    MDP_REQUIRE(gate, "null gate");
    return pd_launch(nullptr, nullptr, 0.0f, kind, num_envs, num_joints, q_des, ld_q_des, qd_des, ld_qd_des, tau_ff,
                     ld_tau_ff, joint_pos, ld_pos, joint_vel, ld_vel, stiffness, damping, effort_limit, velocity_limit,
                     saturation_effort, computed_effort, applied_effort, stream, gate);
}

extern "C" int mdp_circular_buffer_append(float* buffer, int64_t* num_pushes, int max_len, int num_envs, int dim,
                                          int pointer, const float* data, int ld_data, const int32_t* time_lags,
                                          float* out, void* stream) {
    MDP_REQUIRE(max_len >= 1, "The buffer size should be greater than zero. However, it is set to %d!", max_len);
    MDP_REQUIRE(pointer >= 0 && pointer < max_len, "pointer %d out of range [0,%d)", pointer, max_len);
    MDP_REQUIRE(buffer && num_pushes && data, "null pointer");
    if (num_envs == 0 || dim == 0) return MDP_OK;
    int64_t total = (int64_t)num_envs * dim;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_cbuf_append, dim3(grid_for(total, 256)), dim3(256), 0, s, buffer, num_pushes, max_len,
                       num_envs, dim, pointer, data, ld_data, time_lags, out);
    hipLaunchKernelGGL(k_cbuf_inc, dim3(grid_for(num_envs, 256, 1 << 20)), dim3(256), 0, s, num_pushes, num_envs);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_circular_buffer_read(const float* buffer, int max_len, int num_envs, int dim, int pointer,
                                        float* out, void* stream) {
    MDP_REQUIRE(buffer && out, "null pointer");
    MDP_REQUIRE(pointer >= 0 && pointer < max_len, "pointer out of range");
    int64_t total = (int64_t)num_envs * max_len * dim;
    if (total == 0) return MDP_OK;
    hipLaunchKernelGGL(k_cbuf_read, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), buffer, max_len,
                       num_envs, dim, pointer, out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_circular_buffer_reset(float* buffer, int64_t* num_pushes, int max_len, int num_envs, int dim,
                                         const uint8_t* env_mask, void* stream) {
    MDP_REQUIRE(num_pushes, "null pointer");
    if (num_envs == 0) return MDP_OK;
    int d = dim < 1 ? 1 : dim;
    int64_t total = (int64_t)num_envs * d;
    hipLaunchKernelGGL(k_cbuf_reset, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), buffer, num_pushes,
                       max_len, num_envs, d, env_mask);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_linear_interp(const float* x, const float* y, int num_samples, const float* q, int64_t count,
                                 float* fq, float* efforts, void* stream) {
    MDP_REQUIRE(num_samples > 0, "Input tensor x is empty!");
    MDP_REQUIRE(x && y && q, "null pointer");
    if (count == 0) return MDP_OK;
    hipLaunchKernelGGL(k_linear_interp, dim3(grid_for(count, 256)), dim3(256), 0, as_stream(stream), x, y,
                       num_samples, q, count, fq, efforts);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}
