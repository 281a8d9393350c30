// Post-physics scene update as a block-level device routine (interactive_scene.py:496-513): shared by the stand-alone
// launch (mdp_scene_update) and by the actuator kernels that carry the update of the frame they read along
// (scene.update(k) and write_data_to_sim(k+1) have no physics step between them).
#pragma once
#include "mdp_device.hpp"
#include "mdp_host.hpp"
#include <cstdlib>

namespace mdp {

__device__ __forceinline__ void contact_body_update(int64_t i, int e, int b, int B, int H,
                                                    const float* __restrict__ forces, float* __restrict__ net,
                                                    float* __restrict__ hist, float* __restrict__ last_air,
                                                    float* __restrict__ cur_air, float* __restrict__ last_con,
                                                    float* __restrict__ cur_con, float elapsed, float thresh) {
    V3 f = ld3(forces + 3 * i);
    st3(net + 3 * i, f);
    if (hist && H > 0) {
        // history[e, 1:] = history[e, :-1]; history[e, 0] = net  (contact_sensor.py:327-328)
        float* hb = hist + ((int64_t)e * H * B + b) * 3;
        for (int s = H - 1; s >= 1; --s) {
            float* dst = hb + (int64_t)s * B * 3;
            const float* src = hb + (int64_t)(s - 1) * B * 3;
            dst[0] = src[0];
            dst[1] = src[1];
            dst[2] = src[2];
        }
        st3(hb, f);
    }
    if (cur_air) {
        bool is_contact = norm3(f) > thresh;
        float ca = cur_air[i], cc = cur_con[i];
        bool first_contact = (ca > 0.0f) && is_contact;
        bool first_detached = (cc > 0.0f) && !is_contact;
        if (first_contact) last_air[i] = ca + elapsed;
        cur_air[i] = !is_contact ? (ca + elapsed) : 0.0f;
        if (first_detached) last_con[i] = cc + elapsed;
        cur_con[i] = is_contact ? (cc + elapsed) : 0.0f;
    }
}


// Blocks are split into four contiguous ranges, one per group:
//   [0, nb_root)      root_state_w assembly, one thread per (env, column)
//   [.., +nb_joint)   joint_pos / joint_vel refresh + joint_acc, one thread per (env, joint)
//   [.., +nb_contact) contact sensor, a block owns 256 / B whole envs
//   [.., +nb_tick)    lazily evaluated sensors, one thread per (sensor, env)
// `b` is the block index within the update's range; blockDim.x must be 256.
__device__ __forceinline__ void scene_update_block(const mdp_scene_update_t& d, float dt, unsigned b, unsigned nb_root,
                                                   unsigned nb_joint, unsigned nb_contact) {
    const int N = d.num_envs;
    if (b < nb_root) {
        int64_t i = (int64_t)b * 256 + threadIdx.x;
        if (i >= (int64_t)N * 13) return;
        int64_t r = i / 13;
        int k = (int)(i - r * 13);
        float v;
        if (k < 3) v = d.root_transforms[7 * r + k];
        else if (k == 3) v = d.root_transforms[7 * r + 6];
        else if (k < 7) v = d.root_transforms[7 * r + k - 1];
        else v = d.root_velocities[6 * r + (k - 7)];
        d.root_state_w[i] = v;
        return;
    }
    b -= nb_root;
    if (b < nb_joint) {
        int J = d.num_joints;
        int64_t i = (int64_t)b * 256 + threadIdx.x;
        if (i >= (int64_t)N * J) return;
        int64_t e = i / J;
        int j = (int)(i - e * J);
        float qd = d.joint_vel_src[e * d.ld_joint_src + j];
        if (d.joint_vel) d.joint_vel[i] = qd;
        if (d.joint_pos) d.joint_pos[i] = d.joint_pos_src[e * d.ld_joint_src + j];
        if (d.joint_acc) {
            d.joint_acc[i] = (qd - d.prev_joint_vel[i]) / d.acc_elapsed;
            d.prev_joint_vel[i] = qd;
        }
        return;
    }
    b -= nb_joint;
    if (b < nb_contact) {
        int B = d.num_contact_bodies;
        int epb = 256 / B;
        int le = threadIdx.x / B;
        int bb = threadIdx.x - le * B;
        int e = (int)b * epb + le;
        bool active = (le < epb) && (e < N);
        float t_new = 0.0f, t_last = 0.0f;
        bool od = false;
        if (active) {
            t_new = d.contact_timestamp[e] + dt;
            t_last = d.contact_timestamp_last_update[e];
            od = (d.contact_is_outdated[e] != 0) || ((t_new - t_last) + 1e-6f >= d.contact_update_period);
        }
        __syncthreads();
        if (active) {
            if (od && d.contact_eager)
                contact_body_update((int64_t)e * B + bb, e, bb, B, d.history_len, d.net_contact_forces, d.net_forces_w,
                                    d.net_forces_w_history, d.last_air_time, d.current_air_time, d.last_contact_time,
                                    d.current_contact_time, t_new - t_last, d.force_threshold);
            if (bb == 0) {
                d.contact_timestamp[e] = t_new;
                if (od && d.contact_eager) {
                    d.contact_timestamp_last_update[e] = t_new;
                    od = false;
                }
                d.contact_is_outdated[e] = od ? 1 : 0;
            }
        }
        return;
    }
    b -= nb_contact;
    {
        int64_t i = (int64_t)b * 256 + threadIdx.x;
        if (i >= (int64_t)N * d.num_ticks) return;
        int s = (int)(i / N);
        int e = (int)(i - (int64_t)s * N);
        float t = d.tick_timestamp[s][e] + dt;
        d.tick_timestamp[s][e] = t;
        if ((t - d.tick_timestamp_last_update[s][e]) + 1e-6f >= d.tick_update_period[s]) d.tick_is_outdated[s][e] = 1;
    }
}

// The same update restricted to the envs [e0, e1), executed by ONE whole block (any blockDim.x): used by the
// post-physics kernel, whose blocks each own a range of envs for every stage of the step.  Every thread of the block
// must call it (it contains a barrier).
//
// A thread handles one root-state element, one joint and one contact body per round and requests EVERYTHING those three
// need (sources, previous velocities, bookkeeping words, force, history rows, timers) before it waits for any of it, so
// a round costs about two memory round trips instead of a dozen chained ones -- the block is one link of a long chain
// of dependent phases, and every round trip is paid in full.
// barrier_first: the block barrier that separates the body threads' reads of the per-env words from the env threads'
// rewrite of them sits in the LAST round, where that round's loads have arrived, instead of behind the loop -- the
// results of the last round and the per-env words then go out together and the barrier does not wait for result stores
// to be acknowledged (callers whose next phase starts with a barrier of its own: the post-physics manager blocks).
__device__ __forceinline__ void scene_update_envs(const mdp_scene_update_t& d, float dt, int e0, int e1,
                                                  const bool barrier_first = false) {
    constexpr int TICKS = (int)(sizeof(d.tick_timestamp) / sizeof(d.tick_timestamp[0]));
    constexpr int HMAX = 4;  // history rows prefetched into registers; longer histories take the in-place shift
    const int n = e1 - e0;
    const int J = d.num_joints, B = d.num_contact_bodies, H = d.history_len;
    const bool has_root = d.root_state_w != nullptr;
    const bool has_joint = d.joint_vel_src != nullptr && J > 0;
    const bool has_contact = d.net_contact_forces != nullptr && B > 0;
    const bool hist_regs = d.net_forces_w_history != nullptr && H > 0 && H <= HMAX;
    const int n_root = has_root ? n * 13 : 0, n_joint = has_joint ? n * J : 0, n_contact = has_contact ? n * B : 0;
    int rounds_items = n_root > n_joint ? n_root : n_joint;
    if (n_contact > rounds_items) rounds_items = n_contact;
    // bookkeeping words: requested up front, together with the first round's inputs (one round trip, not two), committed
    // behind the barrier at the end (every body thread has read them by then; nobody else writes them)
    float bts = 0.0f, btl = 0.0f;
    uint8_t bflag = 0;
    const int ee = e0 + (int)threadIdx.x;
    const bool he = (int)threadIdx.x < n;
    if (has_contact && he) {
        bts = d.contact_timestamp[ee];
        btl = d.contact_timestamp_last_update[ee];
        bflag = d.contact_is_outdated[ee];
    }
    float tt[TICKS], ttl[TICKS];
#pragma unroll
    for (int s = 0; s < TICKS; ++s) {
        tt[s] = 0.0f; ttl[s] = 0.0f;
        if (s < d.num_ticks && he) { tt[s] = d.tick_timestamp[s][ee]; ttl[s] = d.tick_timestamp_last_update[s][ee]; }
    }
    for (int base = 0; base < rounds_items; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const bool hr = i < n_root, hj = i < n_joint, hc = i < n_contact;
        // ---- requests
        float rv = 0.0f;
        int64_t ro = 0;
        if (hr) {
            int64_t r = e0 + i / 13;
            int k = i % 13;
            ro = 13 * r + k;
            if (k < 3) rv = d.root_transforms[7 * r + k];
            else if (k == 3) rv = d.root_transforms[7 * r + 6];
            else if (k < 7) rv = d.root_transforms[7 * r + k - 1];
            else rv = d.root_velocities[6 * r + (k - 7)];
        }
        float qd = 0.0f, qp = 0.0f, pv = 0.0f;
        int64_t jo = 0;
        if (hj) {
            int64_t e = e0 + i / J;
            int j = i % J;
            jo = e * J + j;
            qd = d.joint_vel_src[e * d.ld_joint_src + j];
            if (d.joint_pos) qp = d.joint_pos_src[e * d.ld_joint_src + j];
            if (d.joint_acc) pv = d.prev_joint_vel[jo];
        }
        int ce = 0, cb = 0;
        int64_t ci = 0;
        float ts = 0.0f, tl = 0.0f, ca = 0.0f, cc = 0.0f;
        uint8_t flag = 0;
        V3 f = V3{0.0f, 0.0f, 0.0f};
        V3 hrow[HMAX];
        if (hc) {
            ce = e0 + i / B;
            cb = i % B;
            ci = (int64_t)ce * B + cb;
            ts = d.contact_timestamp[ce];
            tl = d.contact_timestamp_last_update[ce];
            flag = d.contact_is_outdated[ce];
            if (d.contact_eager) {
                f = ld3(d.net_contact_forces + 3 * ci);
                if (hist_regs) {
                    const float* hb = d.net_forces_w_history + ((int64_t)ce * H * B + cb) * 3;
#pragma unroll
                    for (int s = 0; s < HMAX - 1; ++s)
                        if (s < H - 1) hrow[s] = ld3(hb + (int64_t)s * B * 3);
                }
                if (d.current_air_time) { ca = d.current_air_time[ci]; cc = d.current_contact_time[ci]; }
            }
        }
        if (barrier_first && base + (int)blockDim.x >= rounds_items) {      // (block-uniform)
            asm volatile("" ::"v"(ts), "v"(tl), "v"((int)flag), "v"(bts), "v"(btl), "v"((int)bflag));
            __syncthreads();
        }
        // ---- results
        if (hr) d.root_state_w[ro] = rv;
        if (hj) {
            if (d.joint_vel) d.joint_vel[jo] = qd;
            if (d.joint_pos) d.joint_pos[jo] = qp;
            if (d.joint_acc) {
                d.joint_acc[jo] = (qd - pv) / d.acc_elapsed;
                d.prev_joint_vel[jo] = qd;
            }
        }
        if (hc) {
            float t_new = ts + dt;
            bool od = (flag != 0) || ((t_new - tl) + 1e-6f >= d.contact_update_period);
            if (od && d.contact_eager) {
                const float elapsed = t_new - tl;
                if (hist_regs || !d.net_forces_w_history || H <= 0) {
                    // contact_body_update with its inputs already in registers (contact_sensor.py:314-373)
                    st3(d.net_forces_w + 3 * ci, f);
                    if (hist_regs) {
                        float* hb = d.net_forces_w_history + ((int64_t)ce * H * B + cb) * 3;
#pragma unroll
                        for (int s = HMAX - 1; s >= 1; --s)
                            if (s < H) st3(hb + (int64_t)s * B * 3, hrow[s - 1]);
                        st3(hb, f);
                    }
                    if (d.current_air_time) {
                        bool is_contact = norm3(f) > d.force_threshold;
                        bool first_contact = (ca > 0.0f) && is_contact;
                        bool first_detached = (cc > 0.0f) && !is_contact;
                        if (first_contact) d.last_air_time[ci] = ca + elapsed;
                        d.current_air_time[ci] = !is_contact ? (ca + elapsed) : 0.0f;
                        if (first_detached) d.last_contact_time[ci] = cc + elapsed;
                        d.current_contact_time[ci] = is_contact ? (cc + elapsed) : 0.0f;
                    }
                } else {
                    contact_body_update(ci, ce, cb, B, H, d.net_contact_forces, d.net_forces_w, d.net_forces_w_history,
                                        d.last_air_time, d.current_air_time, d.last_contact_time, d.current_contact_time,
                                        elapsed, d.force_threshold);
                }
            }
        }
    }
    if (!barrier_first || rounds_items <= 0) __syncthreads();
    if (has_contact && he) {
        float t_new = bts + dt;
        bool od = (bflag != 0) || ((t_new - btl) + 1e-6f >= d.contact_update_period);
        d.contact_timestamp[ee] = t_new;
        if (od && d.contact_eager) {
            d.contact_timestamp_last_update[ee] = t_new;
            od = false;
        }
        d.contact_is_outdated[ee] = od ? 1 : 0;
    }
#pragma unroll
    for (int s = 0; s < TICKS; ++s) {
        if (s < d.num_ticks && he) {
            float t = tt[s] + dt;
            d.tick_timestamp[s][ee] = t;
            if ((t - ttl[s]) + 1e-6f >= d.tick_update_period[s]) d.tick_is_outdated[s][ee] = 1;
        }
    }
    // envs beyond blockDim.x (blocks that own more envs than they have threads)
    for (int i = threadIdx.x + blockDim.x; i < n; i += blockDim.x) {
        int e = e0 + i;
        if (has_contact) {
            float t_new = d.contact_timestamp[e] + dt;
            float t_last = d.contact_timestamp_last_update[e];
            bool od = (d.contact_is_outdated[e] != 0) || ((t_new - t_last) + 1e-6f >= d.contact_update_period);
            d.contact_timestamp[e] = t_new;
            if (od && d.contact_eager) {
                d.contact_timestamp_last_update[e] = t_new;
                od = false;
            }
            d.contact_is_outdated[e] = od ? 1 : 0;
        }
        for (int s = 0; s < d.num_ticks; ++s) {
            float t = d.tick_timestamp[s][e] + dt;
            d.tick_timestamp[s][e] = t;
            if ((t - d.tick_timestamp_last_update[s][e]) + 1e-6f >= d.tick_update_period[s]) d.tick_is_outdated[s][e] = 1;
        }
    }
}

// ---- the same update for a SLICE of envs whose items fit one round of the block's threads, in three parts, so that a
// kernel with work of its own (the actuator-net rows of k_lstm8_reg) can put its own requests between them: everything
// is requested up front, the results are stored once they are there, and the per-env bookkeeping words -- read by every
// body thread of the env -- are committed behind a block barrier the caller places at its own end.
// Requires max(13, J, B) * (e1 - e0) <= blockDim.x (host: scene_update_slice_envs).
struct SceneUpdateSlice {
    bool hr, hj, hc, he;
    float rv; int64_t ro;
    float qd, qp, pv; int64_t jo;
    int ce, cb; int64_t ci;
    float ts, tl, ca, cc; uint8_t flag;
    V3 f; V3 hrow[3];
    float ets, etl; uint8_t eflag;                 // env-level copies for the commit
    float tt[4], ttl[4];
};

__device__ __forceinline__ void scene_update_slice_request(const mdp_scene_update_t& d, int e0, int e1, SceneUpdateSlice& r) {
    const int n = e1 - e0;
    const int J = d.num_joints, B = d.num_contact_bodies, H = d.history_len;
    const bool has_root = d.root_state_w != nullptr;
    const bool has_joint = d.joint_vel_src != nullptr && J > 0;
    const bool has_contact = d.net_contact_forces != nullptr && B > 0;
    const bool hist_regs = d.net_forces_w_history != nullptr && H > 0 && H <= 4;
    const int i = threadIdx.x;
    r.hr = has_root && i < n * 13; r.hj = has_joint && i < n * J; r.hc = has_contact && i < n * B;
    r.he = i < n;
    r.rv = 0.0f; r.ro = 0;
    if (r.hr) {
        int64_t row = e0 + i / 13;
        int k = i % 13;
        r.ro = 13 * row + k;
        if (k < 3) r.rv = d.root_transforms[7 * row + k];
        else if (k == 3) r.rv = d.root_transforms[7 * row + 6];
        else if (k < 7) r.rv = d.root_transforms[7 * row + k - 1];
        else r.rv = d.root_velocities[6 * row + (k - 7)];
    }
    r.qd = 0.0f; r.qp = 0.0f; r.pv = 0.0f; r.jo = 0;
    if (r.hj) {
        int64_t e = e0 + i / J;
        int j = i % J;
        r.jo = e * J + j;
        r.qd = d.joint_vel_src[e * d.ld_joint_src + j];
        if (d.joint_pos) r.qp = d.joint_pos_src[e * d.ld_joint_src + j];
        if (d.joint_acc) r.pv = d.prev_joint_vel[r.jo];
    }
    r.ce = 0; r.cb = 0; r.ci = 0; r.ts = 0.0f; r.tl = 0.0f; r.ca = 0.0f; r.cc = 0.0f; r.flag = 0;
    r.f = V3{0.0f, 0.0f, 0.0f};
    if (r.hc) {
        r.ce = e0 + i / B;
        r.cb = i % B;
        r.ci = (int64_t)r.ce * B + r.cb;
        r.ts = d.contact_timestamp[r.ce];
        r.tl = d.contact_timestamp_last_update[r.ce];
        r.flag = d.contact_is_outdated[r.ce];
        if (d.contact_eager) {
            r.f = ld3(d.net_contact_forces + 3 * r.ci);
            if (hist_regs) {
                const float* hb = d.net_forces_w_history + ((int64_t)r.ce * H * B + r.cb) * 3;
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    if (s < H - 1) r.hrow[s] = ld3(hb + (int64_t)s * B * 3);
            }
            if (d.current_air_time) { r.ca = d.current_air_time[r.ci]; r.cc = d.current_contact_time[r.ci]; }
        }
    }
    r.ets = 0.0f; r.etl = 0.0f; r.eflag = 0;
    const int ee = e0 + i;
    if (has_contact && r.he) {
        r.ets = d.contact_timestamp[ee];
        r.etl = d.contact_timestamp_last_update[ee];
        r.eflag = d.contact_is_outdated[ee];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        r.tt[s] = 0.0f; r.ttl[s] = 0.0f;
        if (s < d.num_ticks && r.he) { r.tt[s] = d.tick_timestamp[s][ee]; r.ttl[s] = d.tick_timestamp_last_update[s][ee]; }
    }
}

// results of the round (same expressions as scene_update_envs); no barrier
__device__ __forceinline__ void scene_update_slice_store(const mdp_scene_update_t& d, float dt, const SceneUpdateSlice& r) {
    const int B = d.num_contact_bodies, H = d.history_len;
    const bool hist_regs = d.net_forces_w_history != nullptr && H > 0 && H <= 4;
    if (r.hr) d.root_state_w[r.ro] = r.rv;
    if (r.hj) {
        if (d.joint_vel) d.joint_vel[r.jo] = r.qd;
        if (d.joint_pos) d.joint_pos[r.jo] = r.qp;
        if (d.joint_acc) {
            d.joint_acc[r.jo] = (r.qd - r.pv) / d.acc_elapsed;
            d.prev_joint_vel[r.jo] = r.qd;
        }
    }
    if (r.hc) {
        float t_new = r.ts + dt;
        bool od = (r.flag != 0) || ((t_new - r.tl) + 1e-6f >= d.contact_update_period);
        if (od && d.contact_eager) {
            const float elapsed = t_new - r.tl;
            if (hist_regs || !d.net_forces_w_history || H <= 0) {
                st3(d.net_forces_w + 3 * r.ci, r.f);
                if (hist_regs) {
                    float* hb = d.net_forces_w_history + ((int64_t)r.ce * H * B + r.cb) * 3;
#pragma unroll
                    for (int s = 3; s >= 1; --s)
                        if (s < H) st3(hb + (int64_t)s * B * 3, r.hrow[s - 1]);
                    st3(hb, r.f);
                }
                if (d.current_air_time) {
                    bool is_contact = norm3(r.f) > d.force_threshold;
                    bool first_contact = (r.ca > 0.0f) && is_contact;
                    bool first_detached = (r.cc > 0.0f) && !is_contact;
                    if (first_contact) d.last_air_time[r.ci] = r.ca + elapsed;
                    d.current_air_time[r.ci] = !is_contact ? (r.ca + elapsed) : 0.0f;
                    if (first_detached) d.last_contact_time[r.ci] = r.cc + elapsed;
                    d.current_contact_time[r.ci] = is_contact ? (r.cc + elapsed) : 0.0f;
                }
            } else {
                contact_body_update(r.ci, r.ce, r.cb, B, H, d.net_contact_forces, d.net_forces_w, d.net_forces_w_history,
                                    d.last_air_time, d.current_air_time, d.last_contact_time, d.current_contact_time,
                                    elapsed, d.force_threshold);
            }
        }
    }
}

// per-env bookkeeping words; the caller has placed a block barrier between the request and this call
__device__ __forceinline__ void scene_update_slice_commit(const mdp_scene_update_t& d, float dt, int e0, const SceneUpdateSlice& r) {
    const bool has_contact = d.net_contact_forces != nullptr && d.num_contact_bodies > 0;
    const int ee = e0 + (int)threadIdx.x;
    if (!r.he) return;
    if (has_contact) {
        float t_new = r.ets + dt;
        bool od = (r.eflag != 0) || ((t_new - r.etl) + 1e-6f >= d.contact_update_period);
        d.contact_timestamp[ee] = t_new;
        if (od && d.contact_eager) {
            d.contact_timestamp_last_update[ee] = t_new;
            od = false;
        }
        d.contact_is_outdated[ee] = od ? 1 : 0;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s < d.num_ticks) {
            float t = r.tt[s] + dt;
            d.tick_timestamp[s][ee] = t;
            if ((t - r.ttl[s]) + 1e-6f >= d.tick_update_period[s]) d.tick_is_outdated[s][ee] = 1;
        }
    }
}

// envs per block of a slice layout over `blocks` blocks of `threads` threads, or 0 when a slice would not fit one round
inline int scene_update_slice_envs(const mdp_scene_update_t& d, int64_t blocks, int threads) {
    if (blocks <= 0 || d.num_envs <= 0) return 0;
    const int n = (int)((d.num_envs + blocks - 1) / blocks);
    int widest = 13;
    if (d.num_joints > widest) widest = d.num_joints;
    if (d.num_contact_bodies > widest) widest = d.num_contact_bodies;
    return (int64_t)n * widest <= threads ? n : 0;
}

struct SceneUpdatePlan {
    mdp_scene_update_t d;
    unsigned nb_root = 0, nb_joint = 0, nb_contact = 0, nb_tick = 0, total = 0;
    // layout of an update CARRIED by an actuator launch: carried_envs > 0 -> carried_total blocks that each run
    // scene_update_envs over carried_envs envs (a quarter of the blocks of the role-split layout: the actuator blocks
    // behind them start sooner); 0 -> the role-split layout above (MDP_UPD_ENVS=0)
    int carried_envs = 0;
    unsigned carried_total = 0;
};

// block b of the update carried by an actuator launch (see SceneUpdatePlan)
__device__ __forceinline__ void scene_update_carried(const mdp_scene_update_t& d, float dt, unsigned b, unsigned nb_root,
                                                     unsigned nb_joint, unsigned nb_contact, int carried_envs) {
    if (carried_envs > 0) {
        int64_t e0 = (int64_t)b * carried_envs;
        int64_t e1 = e0 + carried_envs < d.num_envs ? e0 + carried_envs : d.num_envs;
#ifndef MDP_UPD_CARRIED_TWO_PHASE
        int widest = 13;
        if (d.num_joints > widest) widest = d.num_joints;
        if (d.num_contact_bodies > widest) widest = d.num_contact_bodies;
        if (widest * carried_envs <= (int)blockDim.x) {
            // One round of items: requests -> barrier -> results AND the per-env words together.  The barrier sits where
            // the loads have arrived (a body thread's read of its env's timestamps must not see the env thread's
            // rewrite of them), not behind the result stores: scene_update_envs' barrier also waits for every result
            // store to be acknowledged, a round trip at the tail of the launch's longest blocks.
            SceneUpdateSlice us;
            scene_update_slice_request(d, (int)e0, (int)e1, us);
            asm volatile("" ::"v"(us.ts), "v"(us.tl), "v"((int)us.flag), "v"(us.ets), "v"(us.etl), "v"((int)us.eflag));
            __syncthreads();
            scene_update_slice_store(d, dt, us);
            scene_update_slice_commit(d, dt, (int)e0, us);
            return;
        }
#endif
        scene_update_envs(d, dt, (int)e0, (int)e1);
    } else {
        scene_update_block(d, dt, b, nb_root, nb_joint, nb_contact);
    }
}

// validates a descriptor and sizes the block ranges (host)
inline int scene_update_plan(const mdp_scene_update_t* desc, SceneUpdatePlan* plan) {
    MDP_REQUIRE(desc, "null scene-update descriptor");
    const mdp_scene_update_t& d = *desc;
    MDP_REQUIRE(d.num_envs >= 0 && d.num_joints >= 0 && d.num_contact_bodies >= 0 && d.history_len >= 0, "negative size");
    MDP_REQUIRE(d.num_ticks >= 0 && d.num_ticks <= 4, "at most 4 lazily evaluated sensors per launch (got %d)", d.num_ticks);
    const int N = d.num_envs;
    plan->d = d;
    if (N == 0) return MDP_OK;
    const bool has_root = d.root_state_w != nullptr;
    const bool has_joint = d.joint_vel_src != nullptr && d.num_joints > 0;
    const bool has_contact = d.net_contact_forces != nullptr && d.num_contact_bodies > 0;
    if (has_root) MDP_REQUIRE(d.root_transforms && d.root_velocities, "root_state_w needs root_transforms and root_velocities");
    if (has_joint) {
        MDP_REQUIRE(d.ld_joint_src >= d.num_joints, "joint source row stride %d < %d joints", d.ld_joint_src, d.num_joints);
        MDP_REQUIRE(!d.joint_pos || d.joint_pos_src, "joint_pos without a source");
        MDP_REQUIRE(!d.joint_acc || (d.prev_joint_vel && d.acc_elapsed > 0.0f), "joint_acc needs prev_joint_vel and elapsed > 0");
    }
    if (has_contact) {
        MDP_REQUIRE(d.num_contact_bodies <= 256, "fused scene update supports <= 256 contact bodies (got %d)", d.num_contact_bodies);
        MDP_REQUIRE(d.net_forces_w && d.contact_timestamp && d.contact_timestamp_last_update && d.contact_is_outdated,
                    "contact sensor buffers missing");
        MDP_REQUIRE((d.current_air_time == nullptr && d.last_air_time == nullptr && d.last_contact_time == nullptr &&
                     d.current_contact_time == nullptr) ||
                        (d.current_air_time && d.last_air_time && d.last_contact_time && d.current_contact_time),
                    "air/contact timers must be all set or all null");
    }
    for (int s = 0; s < d.num_ticks; ++s)
        MDP_REQUIRE(d.tick_timestamp[s] && d.tick_timestamp_last_update[s] && d.tick_is_outdated[s], "tick sensor %d: null buffer", s);
    auto blocks = [](int64_t work) { return (unsigned)((work + 255) / 256); };
    plan->nb_root = has_root ? blocks((int64_t)N * 13) : 0;
    plan->nb_joint = has_joint ? blocks((int64_t)N * d.num_joints) : 0;
    if (has_contact) {
        int epb = 256 / d.num_contact_bodies;
        plan->nb_contact = (unsigned)(((int64_t)N + epb - 1) / epb);
    }
    plan->nb_tick = d.num_ticks ? blocks((int64_t)N * d.num_ticks) : 0;
    plan->total = plan->nb_root + plan->nb_joint + plan->nb_contact + plan->nb_tick;
    static const int carried = [] { const char* v = std::getenv("MDP_UPD_ENVS"); return v ? std::atoi(v) : 8; }();  // one round of items per block
    plan->carried_envs = carried > 0 && N > 0 ? carried : 0;
    plan->carried_total = plan->carried_envs ? (unsigned)(((int64_t)N + plan->carried_envs - 1) / plan->carried_envs) : plan->total;
    if (!has_joint) { plan->d.joint_acc = nullptr; plan->d.joint_pos = nullptr; plan->d.joint_vel = nullptr; }
    return MDP_OK;
}

}  // namespace mdp
