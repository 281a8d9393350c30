// Fused manager-term programs: one kernel per manager evaluates every lowered term.
// Reference: LAB/managers/observation_manager.py:260-335, reward_manager.py:128-157,
// termination_manager.py:151-174 and the term libraries LAB/envs/mdp/{observations,rewards,
// terminations}.py, TASKS/manager_based/locomotion/velocity/mdp/{rewards,terminations}.py.
//
// Layout: the reference API hands out row-major (N, .) tensors, so the kernels keep that layout and
// get coalescing from the thread mapping instead: the observation kernel maps a thread to one
// (env, column) pair with columns fastest (a block owns whole environments: contiguous row writes),
// reward / termination kernels map a thread to one env and walk short rows (J <= 32) serially.
#include "mdp_device.hpp"
#include "mdp_host.hpp"
#include "raycast_device.hpp"
#include "mesh_handle.hpp"
#include "term_ops.h"
#include "obs_post_device.hpp"
#include "raycaster_update_device.hpp"
#include "scene_update_device.hpp"
#include "command_device.hpp"
#include "events_device.hpp"
#include "stamps_device.hpp"

#include <cstdlib>
#include <mutex>
#include <algorithm>
#include <vector>

using namespace mdp;

struct RayAux {
    mdp_mesh_view_t mesh;
    mdp_raycaster_desc_t d;
};

struct mdp_program {
    int kind = 0, device = 0, num_envs = 0, num_terms = 0;
    mdp_term_t* d_terms = nullptr;
    int32_t* d_pool = nullptr;
    RayAux* d_aux = nullptr;
    int16_t* d_col2term = nullptr;
    // observation columns by evaluation form, for the post-physics kernel: d_col2term + total_cols holds n_opcols
    // "operand" columns (obs_op_is_operand) followed by n_vterms indices of body-frame vector terms
    int n_opcols = 0, n_vterms = 0;
    uint64_t* d_rng = nullptr;  // {seed, per-block epoch counters of k_obs, per-64-env-block counters of k_post_physics}
    unsigned rng_post_base = 0; // first k_post_physics counter
    int total_cols = 0;
    int active_cols = 0;  // columns up to the last one this program writes itself (trailing MDP_OBS_SKIP terms excluded)
    int uses_rng = 0;
    int has_cast = 0;
    int cast_ti = -1;  // first fused ray-cast term (its sensor pose is staged in LDS)
    RayAux h_cast{};   // host copy of that term's descriptor (passed to k_obs by value)
    std::vector<mdp_term_t> h_terms;
};

// blocks of the observation kernel: a block owns 256 / O whole envs (one env when O >= 256)
static unsigned obs_grid(int num_envs, int O) {
    if (O <= 0 || num_envs <= 0) return 1;
    int epb = O >= 256 ? 1 : 256 / O;
    // a block stages the term table once and then walks its envs: cap the grid so the staging (one dependent global
    // round trip + barrier) is amortised over several envs instead of being paid by every env
    static const int64_t cap = [] { const char* e = std::getenv("MDP_OBS_GRID"); return e ? (int64_t)std::atoi(e) : (int64_t)2048; }();
    return grid_for(((int64_t)num_envs + epb - 1) / epb, 1, cap > 0 ? cap : 1 << 20);
}

namespace {

MDP_DEV int idx_at(const mdp_term_t& t, const int32_t* __restrict__ pool, int k) {
    return t.n_idx < 0 ? k : pool[t.idx_offset + k];
}
MDP_DEV int idx2_at(const mdp_term_t& t, const int32_t* __restrict__ pool, int k) {
    return t.n_idx2 < 0 ? k : pool[t.idx2_offset + k];
}
// number of reduced elements: explicit list length, or the full width stored in i1 for "all"
MDP_DEV int red_count(const mdp_term_t& t, int full) { return t.n_idx < 0 ? full : t.n_idx; }

// A term that reduces over joints / bodies is one link of the manager blocks' chain of dependent phases, and a loop
// that loads, waits and accumulates pays one memory round trip per element.  The reductions below therefore REQUEST
// RED_B elements at a time (indices clamped to the last element, so every request stays inside the row) and then
// consume them in the reference's left-to-right order: same sums bit for bit, one round trip per RED_B elements.
constexpr int RED_B = 8;
#define MDP_RED_IDX(j, k0, n)                                                          \
    int j[RED_B];                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < RED_B; ++i_) {                             \
        const int k_ = (k0) + i_ < (n) ? (k0) + i_ : (n) - 1;                          \
        j[i_] = idx_at(t, pool, k_);                                                   \
    }

// max over history of |f| for body b: hist (N,H,B,3)
constexpr int HIST_B = 4;   // history rows requested together (contact_sensor cfgs use history_length <= 3)
MDP_DEV float contact_hist_max(const float* __restrict__ hist, int e, int H, int B, int b) {
    float m = 0.0f;
    if (H <= HIST_B) {
        V3 f[HIST_B];
#pragma unroll
        for (int h = 0; h < HIST_B; ++h) {
            const int hh = h < H ? h : (H > 0 ? H - 1 : 0);
            if (H > 0) f[h] = ld3(hist + (((int64_t)e * H + hh) * B + b) * 3);
        }
#pragma unroll
        for (int h = 0; h < HIST_B; ++h) {
            if (h >= H) continue;
            float n = norm3(f[h]);
            m = (h == 0) ? n : tmax(m, n);
        }
        return m;
    }
    for (int h = 0; h < H; ++h) {
        const float* f = hist + (((int64_t)e * H + h) * B + b) * 3;
        float n = norm3(ld3(f));
        m = (h == 0) ? n : tmax(m, n);
    }
    return m;
}
// the same for two bodies at once (their history rows are requested together)
MDP_DEV void contact_hist_max2(const float* __restrict__ hist, int e, int H, int B, int b0, int b1, float* m0, float* m1) {
    if (H > 0 && H <= HIST_B) {
        V3 f[HIST_B], g[HIST_B];
#pragma unroll
        for (int h = 0; h < HIST_B; ++h) {
            const int hh = h < H ? h : H - 1;
            f[h] = ld3(hist + (((int64_t)e * H + hh) * B + b0) * 3);
            g[h] = ld3(hist + (((int64_t)e * H + hh) * B + b1) * 3);
        }
        float a = 0.0f, c = 0.0f;
#pragma unroll
        for (int h = 0; h < HIST_B; ++h) {
            if (h >= H) continue;
            float n = norm3(f[h]), q = norm3(g[h]);
            a = (h == 0) ? n : tmax(a, n);
            c = (h == 0) ? q : tmax(c, q);
        }
        *m0 = a; *m1 = c;
        return;
    }
    *m0 = contact_hist_max(hist, e, H, B, b0);
    *m1 = contact_hist_max(hist, e, H, B, b1);
}

// =============================================================================================
// Observations
// =============================================================================================
struct CastCtx {
    bool outdated;
};

MDP_DEV float obs_value(const mdp_term_t& t, const int32_t* __restrict__ pool, int e, int k) {
    switch (t.op) {
        case MDP_OBS_COPY: {
            const float* p = (const float*)t.p0;
            return p[(int64_t)e * t.i0 + t.i1 + idx_at(t, pool, k)];
        }
        case MDP_OBS_SUB: {
            const float* p = (const float*)t.p0;
            const float* q = (const float*)t.p1;
            int j = idx_at(t, pool, k);
            return p[(int64_t)e * t.i0 + t.i1 + j] - q[(int64_t)e * t.i2 + t.i3 + j];
        }
        case MDP_OBS_ADD_ROW_BIAS: {
            const float* p = (const float*)t.p0;
            const float* q = (const float*)t.p1;
            return p[(int64_t)e * t.i0 + t.i1 + idx_at(t, pool, k)] + q[(int64_t)e * t.i2 + t.i3];
        }
        case MDP_OBS_BASE_LIN_VEL:
        case MDP_OBS_BASE_ANG_VEL: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            V3 v = ld3(r + (t.op == MDP_OBS_BASE_LIN_VEL ? 7 : 10));
            return get(quat_rotate_inverse(ld4(r + 3), v), k);
        }
        case MDP_OBS_PROJECTED_GRAVITY: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            return get(quat_rotate_inverse(ld4(r + 3), V3{t.f0, t.f1, t.f2}), k);
        }
        case MDP_OBS_ROOT_QUAT_UNIQUE: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            float w = r[3];
            float v = r[3 + k];
            return w < 0.0f ? -v : v;
        }
        case MDP_OBS_JOINT_POS_LIMIT_NORMALIZED: {
            // math.py:22-40 scale_transform
            int j = idx_at(t, pool, k);
            float x = ((const float*)t.p0)[(int64_t)e * t.i0 + j];
            const float* lim = (const float*)t.p1 + ((int64_t)e * t.i1 + j) * 2;
            float lower = lim[0], upper = lim[1];
            float offset = (lower + upper) * 0.5f;
            return 2.0f * (x - offset) / (upper - lower);
        }
        case MDP_OBS_HEIGHT_SCAN: {
            const float* pos = (const float*)t.p0;
            const float* hits = (const float*)t.p1;
            return (pos[3 * (int64_t)e + 2] - hits[((int64_t)e * t.i0 + k) * 3 + 2]) - t.f0;
        }
        default: return 0.0f;
    }
}

// obs_value split in two for the columns whose value is a function of at most three loaded words ("operand" ops:
// everything but the body-frame vectors and the columns other launches write): the addresses, and what is done with the
// loaded words.  The post-physics kernel requests the operands of several columns before it uses any of them.
MDP_DEV bool obs_op_is_operand(int op) {
    return op == MDP_OBS_COPY || op == MDP_OBS_SUB || op == MDP_OBS_ADD_ROW_BIAS || op == MDP_OBS_ROOT_QUAT_UNIQUE ||
           op == MDP_OBS_JOINT_POS_LIMIT_NORMALIZED || op == MDP_OBS_HEIGHT_SCAN;
}
MDP_DEV bool obs_op_is_vector(int op) {
    return op == MDP_OBS_BASE_LIN_VEL || op == MDP_OBS_BASE_ANG_VEL || op == MDP_OBS_PROJECTED_GRAVITY;
}
MDP_DEV void obs_operand_addrs(const mdp_term_t& t, const int32_t* __restrict__ pool, int e, int k, const float** pa,
                               const float** pb, const float** pc) {
    const float* a;
    const float* b;
    const float* c;
    switch (t.op) {
        case MDP_OBS_SUB: {
            int j = idx_at(t, pool, k);
            a = (const float*)t.p0 + ((int64_t)e * t.i0 + t.i1 + j);
            b = (const float*)t.p1 + ((int64_t)e * t.i2 + t.i3 + j);
            c = a;
            break;
        }
        case MDP_OBS_ADD_ROW_BIAS: {
            a = (const float*)t.p0 + ((int64_t)e * t.i0 + t.i1 + idx_at(t, pool, k));
            b = (const float*)t.p1 + ((int64_t)e * t.i2 + t.i3);
            c = a;
            break;
        }
        case MDP_OBS_ROOT_QUAT_UNIQUE: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            a = r + 3 + k;
            b = r + 3;
            c = a;
            break;
        }
        case MDP_OBS_JOINT_POS_LIMIT_NORMALIZED: {
            int j = idx_at(t, pool, k);
            a = (const float*)t.p0 + ((int64_t)e * t.i0 + j);
            b = (const float*)t.p1 + ((int64_t)e * t.i1 + j) * 2;
            c = b + 1;
            break;
        }
        case MDP_OBS_HEIGHT_SCAN: {
            a = (const float*)t.p0 + (3 * (int64_t)e + 2);
            b = (const float*)t.p1 + (((int64_t)e * t.i0 + k) * 3 + 2);
            c = a;
            break;
        }
        default: {   // MDP_OBS_COPY
            a = (const float*)t.p0 + ((int64_t)e * t.i0 + t.i1 + idx_at(t, pool, k));
            b = a;
            c = a;
            break;
        }
    }
    *pa = a; *pb = b; *pc = c;
}
// same expressions as obs_value
MDP_DEV float obs_operand_finish(const mdp_term_t& t, float a, float b, float c) {
    switch (t.op) {
        case MDP_OBS_SUB: return a - b;
        case MDP_OBS_ADD_ROW_BIAS: return a + b;
        case MDP_OBS_ROOT_QUAT_UNIQUE: return b < 0.0f ? -a : a;
        case MDP_OBS_JOINT_POS_LIMIT_NORMALIZED: {
            float offset = (b + c) * 0.5f;
            return 2.0f * (a - offset) / (c - b);
        }
        case MDP_OBS_HEIGHT_SCAN: return (a - b) - t.f0;
        default: return a;
    }
}

// sensor pose of one env for the fused cast (ray_caster.py:226-252): position (+ drift) and the rotation
// applied to the ray pattern (yaw-only or full).  Computed once per env and shared through LDS.
MDP_DEV void cast_pose(const mdp_raycaster_desc_t& d, int e, V3* pos, Q4* qr, Q4* q) {
    const float* p = d.root_pose + (int64_t)e * d.ld_pose;
    *pos = ld3(p);
    if (d.drift) { pos->x += d.drift[3 * e]; pos->y += d.drift[3 * e + 1]; pos->z += d.drift[3 * e + 2]; }
    *q = ld4(p + 3);
    *qr = d.attach_yaw_only ? yaw_quat_halfangle(*q) : *q;   // (one form for every path of the sensor)
}

MDP_DEV float obs_height_scan_cast(const RayAux& aux, float offset, int e, int k, bool od, const float* pose) {
    const RayAux* a = &aux;
    const mdp_raycaster_desc_t& d = a->d;
    const float inf = __builtin_inff();
    int R = d.num_rays;
    float* o = d.ray_hits_w + ((int64_t)e * R + k) * 3;
    if (!od) return (d.pos_w[3 * (int64_t)e + 2] - o[2]) - offset;
    V3 pos;
    Q4 qr, q;
    if (pose) {
        pos = V3{pose[0], pose[1], pose[2]};
        qr = Q4{pose[3], pose[4], pose[5], pose[6]};
        q = qr;  // only read when the full rotation is attached, in which case qr == q
    } else {
        cast_pose(d, e, &pos, &qr, &q);
    }
    V3 s = quat_apply(qr, ld3(d.ray_starts_local + 3 * k));
    s = V3{s.x + pos.x, s.y + pos.y, s.z + pos.z};
    V3 dir = ld3(d.ray_dirs_local + 3 * k);
    if (!d.attach_yaw_only) dir = quat_apply(q, dir);
    mdp_ray_t ray = mdp_rc_make_ray(s.x, s.y, s.z, dir.x, dir.y, dir.z);
    mdp_hit_t h = mdp_rc_query(&a->mesh, &ray, d.max_dist, a->mesh.accel);
    float hz;
    if (h.face >= 0) {
        o[0] = s.x + h.t * dir.x; o[1] = s.y + h.t * dir.y; hz = s.z + h.t * dir.z; o[2] = hz;
    } else {
        o[0] = inf; o[1] = inf; o[2] = inf; hz = inf;
    }
    return (pos.z - hz) - offset;
}

MDP_DEV void obs_cast_commit(const mdp_term_t& t, int e) {
    const RayAux* a = (const RayAux*)t.p4;
    const mdp_raycaster_desc_t& d = a->d;
    const float* p = d.root_pose + (int64_t)e * d.ld_pose;
    V3 pos = ld3(p);
    if (d.drift) { pos.x += d.drift[3 * e]; pos.y += d.drift[3 * e + 1]; pos.z += d.drift[3 * e + 2]; }
    st3(d.pos_w + 3 * (int64_t)e, pos);
    st4(d.quat_w + 4 * (int64_t)e, ld4(p + 3));
    if (d.timestamp_last_update) d.timestamp_last_update[e] = d.timestamp[e];
    if (d.is_outdated) d.is_outdated[e] = 0;
}

constexpr int OBS_POSE_SLOTS = 32;  // env slots per block whose sensor pose is staged in LDS

// Every lane of a wave walks a different column, so term records would be fetched with per-lane (vector) global
// loads, one dependent round trip per field.  The block stages the term table and the column map in LDS
// once; the descriptor of the fused ray-cast term arrives by value (scalar registers).
template <bool CAST>
__global__ void k_obs(const mdp_term_t* __restrict__ g_terms, int T, const int32_t* __restrict__ pool,
                      const int16_t* __restrict__ g_col2term, int O, int N, int epb, int slot_w,
                      float* __restrict__ obs, int ld_obs, uint64_t* rng, int uses_rng, int has_cast, int cast_ti,
                      RayAux cast) {
    __shared__ float s_pose[OBS_POSE_SLOTS][8];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    mdp_term_t* terms = reinterpret_cast<mdp_term_t*>(s_dyn);
    int16_t* col2term = reinterpret_cast<int16_t*>(s_dyn + (size_t)T * sizeof(mdp_term_t));
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(g_terms);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_dyn);
        const int words = T * (int)(sizeof(mdp_term_t) / 4);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
        for (int i = threadIdx.x; i < O; i += blockDim.x) col2term[i] = g_col2term[i];
    }
    __syncthreads();
    int le = threadIdx.x / slot_w;
    int lc = threadIdx.x - le * slot_w;
    uint64_t seed = 0, epoch = 0;
    if (uses_rng) { seed = rng[0]; epoch = rng[1 + blockIdx.x]; }
    const bool stage_pose = CAST && has_cast && cast_ti >= 0 && epb <= OBS_POSE_SLOTS;
    for (int base = blockIdx.x * epb; base < N; base += gridDim.x * epb) {
        int e = base + le;
        bool active = le < epb && e < N;
        if (stage_pose) {
            if (active && lc == 0) {
                bool od = cast.d.is_outdated ? (cast.d.is_outdated[e] != 0) : true;
                s_pose[le][7] = od ? 1.0f : 0.0f;
                if (od) {
                    V3 pos;
                    Q4 qr, q;
                    cast_pose(cast.d, e, &pos, &qr, &q);
                    s_pose[le][0] = pos.x; s_pose[le][1] = pos.y; s_pose[le][2] = pos.z;
                    s_pose[le][3] = qr.w; s_pose[le][4] = qr.x; s_pose[le][5] = qr.y; s_pose[le][6] = qr.z;
                }
            }
            __syncthreads();
        }
        if (active) {
            for (int c = lc; c < O; c += slot_w) {
                int ti = col2term[c];
                const mdp_term_t& t = terms[ti];
                if (t.op == MDP_OBS_SKIP) continue;   // written by another launch (mdp_raycaster_update epilogue)
                int k = c - t.out_offset;
                float v;
                if (CAST && t.op == MDP_OBS_HEIGHT_SCAN_CAST) {
                    if (stage_pose && ti == cast_ti) {
                        v = obs_height_scan_cast(cast, t.f0, e, k, s_pose[le][7] != 0.0f, s_pose[le]);
                    } else {
                        const RayAux* a = (const RayAux*)t.p4;
                        bool od = a->d.is_outdated ? (a->d.is_outdated[e] != 0) : true;
                        v = obs_height_scan_cast(*a, t.f0, e, k, od, nullptr);
                    }
                } else {
                    v = obs_value(t, pool, e, k);
                }
                v = obs_post(t, v, e, k, seed, epoch, 0x4f425300u + (uint32_t)ti, (uint64_t)e * (uint64_t)O + c);
                obs[(int64_t)e * ld_obs + c] = v;
            }
        }
        if (CAST && has_cast) {
            __syncthreads();  // every ray thread has read the outdated flag before it is cleared
            if (active && lc == 0) {
                for (int ti = 0; ti < T; ++ti) {
                    const mdp_term_t& t = terms[ti];
                    if (t.op != MDP_OBS_HEIGHT_SCAN_CAST) continue;
                    const RayAux* a = (const RayAux*)t.p4;
                    bool od = a->d.is_outdated ? (a->d.is_outdated[e] != 0) : true;
                    if (od) obs_cast_commit(t, e);
                }
            }
        }
    }
    // Each block owns its epoch counter (all equal: one increment per launch).  A single shared counter would
    // need a last-block ticket, and thousands of same-address atomics serialise in L2 (measured: 150 us of a
    // 180 us launch at 4096 blocks).  Counters live on the device, so hipGraph replays draw fresh numbers.
    if (uses_rng && threadIdx.x == 0) rng[1 + blockIdx.x] = epoch + 1;
}

__global__ void k_obs_post(mdp_term_t t, int N, float* __restrict__ data, int ld, uint64_t seed, uint64_t offset) {
    int64_t total = (int64_t)N * t.dim;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int e = (int)(i / t.dim);
        int k = (int)(i - (int64_t)e * t.dim);
        float v = data[(int64_t)e * ld + k];
        data[(int64_t)e * ld + k] = obs_post(t, v, e, k, seed, offset, 0x504f5354u, (uint64_t)i);
    }
}

// =============================================================================================
// Rewards
// =============================================================================================
MDP_DEV float reward_value(const mdp_term_t& t, const int32_t* __restrict__ pool, int e, int N) {
    switch (t.op) {
        case MDP_REW_EXTERNAL: return ((const float*)t.p0)[(int64_t)e * (t.i0 > 0 ? t.i0 : 1)];
        case MDP_REW_IS_ALIVE: return ((const uint8_t*)t.p0)[e] ? 0.0f : 1.0f;
        case MDP_REW_IS_TERMINATED: return ((const uint8_t*)t.p0)[e] ? 1.0f : 0.0f;
        case MDP_REW_IS_TERMINATED_TERM: {
            const uint8_t* dones = (const uint8_t*)t.p0;
            float s = 0.0f;
            int n = red_count(t, t.i1);
            for (int k = 0; k < n; ++k) s += dones[(int64_t)idx_at(t, pool, k) * N + e] ? 1.0f : 0.0f;
            return s * (((const uint8_t*)t.p1)[e] ? 0.0f : 1.0f);
        }
        case MDP_REW_LIN_VEL_Z_L2: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            float z = quat_rotate_inverse(ld4(r + 3), ld3(r + 7)).z;
            return z * z;
        }
        case MDP_REW_ANG_VEL_XY_L2: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            V3 w = quat_rotate_inverse(ld4(r + 3), ld3(r + 10));
            return w.x * w.x + w.y * w.y;
        }
        case MDP_REW_FLAT_ORIENTATION_L2: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            V3 g = quat_rotate_inverse(ld4(r + 3), V3{t.f0, t.f1, t.f2});
            return g.x * g.x + g.y * g.y;
        }
        case MDP_REW_BASE_HEIGHT_L2: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            float target = t.f0;
            if (t.p1) {
                const float* hits = (const float*)t.p1;
                int R = t.i1;
                float s = 0.0f;
                for (int k = 0; k < R; ++k) s += hits[((int64_t)e * R + k) * 3 + 2];
                target = t.f0 + s / (float)R;
            }
            float d = r[2] - target;
            return d * d;
        }
        case MDP_REW_BODY_LIN_ACC_L2: {
            const float* acc = (const float*)t.p0;
            int B = t.i1, n = red_count(t, B);
            float s = 0.0f;
            for (int k0 = 0; k0 < n; k0 += RED_B) {
                MDP_RED_IDX(j, k0, n);
                V3 v[RED_B];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) v[i] = ld3(acc + ((int64_t)e * B + j[i]) * 6);
#pragma unroll
                for (int i = 0; i < RED_B; ++i) if (k0 + i < n) s += norm3(v[i]);
            }
            return s;
        }
        case MDP_REW_SUM_SQ: {
            const float* p = (const float*)t.p0 + (int64_t)e * t.i0;
            int n = red_count(t, t.i1);
            float s = 0.0f;
            for (int k0 = 0; k0 < n; k0 += RED_B) {
                MDP_RED_IDX(j, k0, n);
                float v[RED_B];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) v[i] = p[j[i]];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) if (k0 + i < n) s += v[i] * v[i];
            }
            return s;
        }
        case MDP_REW_SUM_ABS: {
            const float* p = (const float*)t.p0 + (int64_t)e * t.i0;
            int n = red_count(t, t.i1);
            float s = 0.0f;
            for (int k0 = 0; k0 < n; k0 += RED_B) {
                MDP_RED_IDX(j, k0, n);
                float v[RED_B];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) v[i] = p[j[i]];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) if (k0 + i < n) s += fabsf(v[i]);
            }
            return s;
        }
        case MDP_REW_SUM_ABS_DIFF:
        case MDP_REW_SUM_SQ_DIFF: {
            const float* p = (const float*)t.p0 + (int64_t)e * t.i0;
            const float* q = (const float*)t.p1 + (int64_t)e * t.i2;
            int n = red_count(t, t.i1);
            float s = 0.0f;
            for (int k0 = 0; k0 < n; k0 += RED_B) {
                MDP_RED_IDX(j, k0, n);
                float v[RED_B], w[RED_B];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) { v[i] = p[j[i]]; w[i] = q[j[i]]; }
#pragma unroll
                for (int i = 0; i < RED_B; ++i) {
                    if (k0 + i >= n) continue;
                    float d = v[i] - w[i];
                    s += (t.op == MDP_REW_SUM_ABS_DIFF) ? fabsf(d) : d * d;
                }
            }
            return s;
        }
        case MDP_REW_JOINT_POS_LIMITS: {
            const float* p = (const float*)t.p0 + (int64_t)e * t.i0;
            const float* lim = (const float*)t.p1 + (int64_t)e * t.i1 * 2;
            int n = red_count(t, t.i1);
            float s = 0.0f;
            for (int k0 = 0; k0 < n; k0 += RED_B) {
                MDP_RED_IDX(j, k0, n);
                float v[RED_B], lo[RED_B], hi[RED_B];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) { v[i] = p[j[i]]; lo[i] = lim[2 * j[i]]; hi[i] = lim[2 * j[i] + 1]; }
#pragma unroll
                for (int i = 0; i < RED_B; ++i) {
                    if (k0 + i >= n) continue;
                    float o = -tclamp_max(v[i] - lo[i], 0.0f);
                    o = o + tclamp_min(v[i] - hi[i], 0.0f);
                    s += o;
                }
            }
            return s;
        }
        case MDP_REW_JOINT_VEL_LIMITS: {
            const float* p = (const float*)t.p0 + (int64_t)e * t.i0;
            const float* lim = (const float*)t.p1 + (int64_t)e * t.i1;
            int n = red_count(t, t.i1);
            float s = 0.0f;
            for (int k0 = 0; k0 < n; k0 += RED_B) {
                MDP_RED_IDX(j, k0, n);
                float v[RED_B], l[RED_B];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) { v[i] = p[j[i]]; l[i] = lim[j[i]]; }
#pragma unroll
                for (int i = 0; i < RED_B; ++i) if (k0 + i < n) s += tclamp(fabsf(v[i]) - l[i] * t.f0, 0.0f, 1.0f);
            }
            return s;
        }
        case MDP_REW_UNDESIRED_CONTACTS:
        case MDP_REW_CONTACT_FORCES: {
            const float* hist = (const float*)t.p0;
            int H = t.i0, B = t.i1, n = red_count(t, B);
            float s = 0.0f;
            for (int k = 0; k < n; k += 2) {
                float m0, m1;
                contact_hist_max2(hist, e, H, B, idx_at(t, pool, k), idx_at(t, pool, k + 1 < n ? k + 1 : k), &m0, &m1);
                if (t.op == MDP_REW_UNDESIRED_CONTACTS) s += (m0 > t.f0) ? 1.0f : 0.0f;
                else s += tclamp_min(m0 - t.f0, 0.0f);
                if (k + 1 >= n) break;
                if (t.op == MDP_REW_UNDESIRED_CONTACTS) s += (m1 > t.f0) ? 1.0f : 0.0f;
                else s += tclamp_min(m1 - t.f0, 0.0f);
            }
            return s;
        }
        case MDP_REW_TRACK_LIN_VEL_XY_EXP: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            const float* c = (const float*)t.p1 + (int64_t)e * t.i1;
            V3 v = quat_rotate_inverse(ld4(r + 3), ld3(r + 7));
            float dx = c[0] - v.x, dy = c[1] - v.y;
            float err = dx * dx + dy * dy;
            return expf(-err / t.f0);
        }
        case MDP_REW_TRACK_ANG_VEL_Z_EXP: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            const float* c = (const float*)t.p1 + (int64_t)e * t.i1;
            V3 w = quat_rotate_inverse(ld4(r + 3), ld3(r + 10));
            float d = c[2] - w.z;
            return expf(-(d * d) / t.f0);
        }
        case MDP_REW_FEET_AIR_TIME: {
            const float* tc = (const float*)t.p0 + (int64_t)e * t.i1;
            const float* la = (const float*)t.p1 + (int64_t)e * t.i1;
            const float* c = (const float*)t.p2 + (int64_t)e * t.i2;
            int n = red_count(t, t.i1);
            float s = 0.0f;
            const float cx = c[0], cy = c[1];
            for (int k0 = 0; k0 < n; k0 += RED_B) {
                MDP_RED_IDX(j, k0, n);
                float v[RED_B], w[RED_B];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) { v[i] = tc[j[i]]; w[i] = la[j[i]]; }
#pragma unroll
                for (int i = 0; i < RED_B; ++i) {
                    if (k0 + i >= n) continue;
                    float first = ((v[i] > 0.0f) && (v[i] < t.f1)) ? 1.0f : 0.0f;
                    s += (w[i] - t.f0) * first;
                }
            }
            return s * ((norm2(cx, cy) > 0.1f) ? 1.0f : 0.0f);
        }
        case MDP_REW_FEET_AIR_TIME_POSITIVE_BIPED: {
            const float* air = (const float*)t.p0 + (int64_t)e * t.i1;
            const float* con = (const float*)t.p1 + (int64_t)e * t.i1;
            const float* c = (const float*)t.p2 + (int64_t)e * t.i2;
            int n = red_count(t, t.i1);
            int in_contact = 0;
            for (int k = 0; k < n; ++k) in_contact += con[idx_at(t, pool, k)] > 0.0f ? 1 : 0;
            bool single = in_contact == 1;
            float m = 0.0f;
            for (int k = 0; k < n; ++k) {
                int b = idx_at(t, pool, k);
                float mode = con[b] > 0.0f ? con[b] : air[b];
                float v = single ? mode : 0.0f;
                m = (k == 0) ? v : tmin(m, v);
            }
            m = tclamp_max(m, t.f0);
            return m * ((norm2(c[0], c[1]) > 0.1f) ? 1.0f : 0.0f);
        }
        case MDP_REW_FEET_SLIDE: {
            const float* hist = (const float*)t.p0;
            const float* body = (const float*)t.p1;
            int H = t.i0, Bs = t.i1, Ba = t.i2, n = red_count(t, Bs);
            float s = 0.0f;
            for (int k = 0; k < n; k += 2) {
                const int k1 = k + 1 < n ? k + 1 : k;
                const float* bs0 = body + ((int64_t)e * Ba + idx2_at(t, pool, k)) * 13;
                const float* bs1 = body + ((int64_t)e * Ba + idx2_at(t, pool, k1)) * 13;
                const float vx0 = bs0[7], vy0 = bs0[8], vx1 = bs1[7], vy1 = bs1[8];
                float m0, m1;
                contact_hist_max2(hist, e, H, Bs, idx_at(t, pool, k), idx_at(t, pool, k1), &m0, &m1);
                s += norm2(vx0, vy0) * (m0 > 1.0f ? 1.0f : 0.0f);
                if (k + 1 >= n) break;
                s += norm2(vx1, vy1) * (m1 > 1.0f ? 1.0f : 0.0f);
            }
            return s;
        }
        case MDP_REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            const float* c = (const float*)t.p1 + (int64_t)e * t.i1;
            V3 v = quat_rotate_inverse(yaw_quat(ld4(r + 3)), ld3(r + 7));
            float dx = c[0] - v.x, dy = c[1] - v.y;
            return expf(-(dx * dx + dy * dy) / t.f0);
        }
        case MDP_REW_TRACK_ANG_VEL_Z_WORLD_EXP: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            const float* c = (const float*)t.p1 + (int64_t)e * t.i1;
            float d = c[2] - r[12];
            return expf(-(d * d) / t.f0);
        }
        default: return 0.0f;
    }
}

// reward_manager.py:128-157
// One wave evaluates ONE term for 64 consecutive envs (coalesced state reads, wave-uniform opcode); the block
// covers all terms of its 64 envs, parks the weighted values in LDS and lane e of wave 0 adds them up in term
// order -- the same left-to-right sum as reward_manager.py:143-155.  (A thread-per-env kernel walks T terms
// of dependent loads on 64 waves only: 20 us at 4096 envs.)
__global__ void k_reward(const mdp_term_t* __restrict__ terms, int T, const int32_t* __restrict__ pool, int N,
                         float dt, float* __restrict__ reward, float* __restrict__ sums, float* __restrict__ step) {
    extern __shared__ float s_val[];  // [T][64]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    for (int base = blockIdx.x * 64; base < N; base += gridDim.x * 64) {
        int e = base + lane;
        for (int ti = wave; ti < T; ti += nwaves) {
            const mdp_term_t& t = terms[ti];
            float value = 0.0f;
            if (e < N && t.weight != 0.0f) {
                value = reward_value(t, pool, e, N) * t.weight * dt;
                if (sums) sums[(int64_t)t.out_offset * N + e] += value;
                if (step) step[(int64_t)e * T + t.out_offset] = value / dt;
            }
            s_val[ti * 64 + lane] = value;
        }
        __syncthreads();
        if (wave == 0 && e < N) {
            float total = 0.0f;
            for (int ti = 0; ti < T; ++ti)
                if (terms[ti].weight != 0.0f) total += s_val[ti * 64 + lane];
            reward[e] = total;
        }
        __syncthreads();
    }
}
// =============================================================================================
// Terminations
// =============================================================================================
MDP_DEV bool term_value(const mdp_term_t& t, const int32_t* __restrict__ pool, int e) {
    switch (t.op) {
        case MDP_TERM_EXTERNAL: return ((const uint8_t*)t.p0)[e] != 0;
        case MDP_TERM_TIME_OUT: return ((const int64_t*)t.p0)[e] >= t.l0;
        case MDP_TERM_COMMAND_RESAMPLE:
            return (((const float*)t.p0)[e] <= t.f0) && (((const int64_t*)t.p1)[e] == (int64_t)t.i0);
        case MDP_TERM_BAD_ORIENTATION: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            V3 g = quat_rotate_inverse(ld4(r + 3), V3{t.f0, t.f1, t.f2});
            return fabsf(acosf(-g.z)) > t.f3;
        }
        case MDP_TERM_ROOT_HEIGHT_BELOW_MINIMUM:
            return ((const float*)t.p0)[(int64_t)e * t.i0 + 2] < t.f0;
        case MDP_TERM_JOINT_POS_OUT_OF_MANUAL_LIMIT: {
            const float* p = (const float*)t.p0 + (int64_t)e * t.i0;
            int n = red_count(t, t.i1);
            bool up = false, lo = false;
            for (int k0 = 0; k0 < n; k0 += RED_B) {
                MDP_RED_IDX(j, k0, n);
                float v[RED_B];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) v[i] = p[j[i]];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) {
                    if (k0 + i >= n) continue;
                    up = up || (v[i] > t.f1);
                    lo = lo || (v[i] < t.f0);
                }
            }
            return up || lo;
        }
        case MDP_TERM_JOINT_VEL_OUT_OF_LIMIT: {
            const float* p = (const float*)t.p0 + (int64_t)e * t.i0;
            const float* lim = (const float*)t.p1 + (int64_t)e * t.i1;
            int n = red_count(t, t.i1);
            bool any = false;
            for (int k0 = 0; k0 < n; k0 += RED_B) {
                MDP_RED_IDX(j, k0, n);
                float v[RED_B], l[RED_B];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) { v[i] = p[j[i]]; l[i] = lim[j[i]]; }
#pragma unroll
                for (int i = 0; i < RED_B; ++i) if (k0 + i < n) any = any || (fabsf(v[i]) > l[i]);
            }
            return any;
        }
        case MDP_TERM_JOINT_VEL_OUT_OF_MANUAL_LIMIT: {
            const float* p = (const float*)t.p0 + (int64_t)e * t.i0;
            int n = red_count(t, t.i1);
            bool any = false;
            for (int k0 = 0; k0 < n; k0 += RED_B) {
                MDP_RED_IDX(j, k0, n);
                float v[RED_B];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) v[i] = p[j[i]];
#pragma unroll
                for (int i = 0; i < RED_B; ++i) if (k0 + i < n) any = any || (fabsf(v[i]) > t.f0);
            }
            return any;
        }
        case MDP_TERM_JOINT_EFFORT_OUT_OF_LIMIT: {
            // torch.isclose(computed, applied): |a-b| <= atol + rtol*|b| (rtol 1e-5, atol 1e-8), equal infs close
            const float* a = (const float*)t.p0 + (int64_t)e * t.i0;
            const float* b = (const float*)t.p1 + (int64_t)e * t.i0;
            int n = red_count(t, t.i1);
            bool any = false;
            for (int k0 = 0; k0 < n; k0 += RED_B) {
              MDP_RED_IDX(j, k0, n);
              float xs[RED_B], ys[RED_B];
#pragma unroll
              for (int i = 0; i < RED_B; ++i) { xs[i] = a[j[i]]; ys[i] = b[j[i]]; }
#pragma unroll
              for (int i = 0; i < RED_B; ++i) {
                if (k0 + i >= n) continue;
                float x = xs[i], y = ys[i];
                bool close = (x == y) || (isfinite(x) && isfinite(y) && (fabsf(x - y) <= 1e-8f + 1e-5f * fabsf(y)));
                any = any || close;
              }
            }
            return any;
        }
        case MDP_TERM_ILLEGAL_CONTACT: {
            const float* hist = (const float*)t.p0;
            int H = t.i0, B = t.i1, n = red_count(t, B);
            bool any = false;
            for (int k = 0; k < n; k += 2) {
                float m0, m1;
                contact_hist_max2(hist, e, H, B, idx_at(t, pool, k), idx_at(t, pool, k + 1 < n ? k + 1 : k), &m0, &m1);
                any = any || (m0 > t.f0);
                if (k + 1 < n) any = any || (m1 > t.f0);
            }
            return any;
        }
        case MDP_TERM_TERRAIN_OUT_OF_BOUNDS: {
            const float* r = (const float*)t.p0 + (int64_t)e * t.i0;
            return (fabsf(r[0]) > t.f0) || (fabsf(r[1]) > t.f1);
        }
        default: return false;
    }
}

// termination_manager.py:151-174
__global__ void k_termination(const mdp_term_t* __restrict__ terms, int T, const int32_t* __restrict__ pool, int N,
                              uint8_t* __restrict__ term_dones, uint8_t* __restrict__ truncated,
                              uint8_t* __restrict__ terminated, uint8_t* __restrict__ dones) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N; e += gridDim.x * blockDim.x) {
        bool trunc = false, term = false;
        for (int ti = 0; ti < T; ++ti) {
            const mdp_term_t& t = terms[ti];
            bool v = term_value(t, pool, e);
            if (t.i3) trunc = trunc || v; else term = term || v;
            if (term_dones) term_dones[(int64_t)t.out_offset * N + e] = v ? 1 : 0;
        }
        if (truncated) truncated[e] = trunc ? 1 : 0;
        if (terminated) terminated[e] = term ? 1 : 0;
        if (dones) dones[e] = (trunc || term) ? 1 : 0;
    }
}

// manager_based_rl_env.py:199-206 in one launch: episode_length_buf += 1, terminations, rewards.  A block owns 64
// consecutive envs; wave w evaluates term w, w + nwaves, ... for those envs (wave-uniform opcode, coalesced reads).
// The termination flags are committed to global memory and a block barrier separates them from the reward phase, so
// reward terms that read `terminated` / `term_dones` through their bound pointers see this step's values.
__global__ void k_term_reward(const mdp_term_t* __restrict__ tterms, int Tt, const int32_t* __restrict__ tpool,
                              const mdp_term_t* __restrict__ rterms, int Tr, const int32_t* __restrict__ rpool, int N,
                              float dt, int64_t* len_inc, uint8_t* term_dones, uint8_t* truncated, uint8_t* terminated,
                              uint8_t* dones, float* reward, float* sums, float* step) {
    extern __shared__ float s_val[];  // [max(Tt, Tr)][64]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    for (int base = blockIdx.x * 64; base < N; base += gridDim.x * 64) {
        int e = base + lane;
        if (len_inc) {
            if (wave == 0 && e < N) len_inc[e] += 1;
            __syncthreads();
        }
        for (int ti = wave; ti < Tt; ti += nwaves) {
            const mdp_term_t& t = tterms[ti];
            bool v = e < N ? term_value(t, tpool, e) : false;
            if (term_dones && e < N) term_dones[(int64_t)t.out_offset * N + e] = v ? 1 : 0;
            s_val[ti * 64 + lane] = v ? 1.0f : 0.0f;
        }
        __syncthreads();
        if (wave == 0 && e < N && Tt > 0) {
            bool trunc = false, term = false;
            for (int ti = 0; ti < Tt; ++ti) {
                bool v = s_val[ti * 64 + lane] != 0.0f;
                if (tterms[ti].i3) trunc = trunc || v; else term = term || v;
            }
            if (truncated) truncated[e] = trunc ? 1 : 0;
            if (terminated) terminated[e] = term ? 1 : 0;
            if (dones) dones[e] = (trunc || term) ? 1 : 0;
        }
        __threadfence_block();
        __syncthreads();
        for (int ti = wave; ti < Tr; ti += nwaves) {
            const mdp_term_t& t = rterms[ti];
            float value = 0.0f;
            if (e < N && t.weight != 0.0f) {
                value = reward_value(t, rpool, e, N) * t.weight * dt;
                if (sums) sums[(int64_t)t.out_offset * N + e] += value;
                if (step) step[(int64_t)e * Tr + t.out_offset] = value / dt;
            }
            s_val[ti * 64 + lane] = value;
        }
        __syncthreads();
        if (wave == 0 && e < N && Tr > 0) {
            float total = 0.0f;
            for (int ti = 0; ti < Tr; ++ti)
                if (rterms[ti].weight != 0.0f) total += s_val[ti * 64 + lane];
            reward[e] = total;
        }
        __syncthreads();
    }
}

// =============================================================================================
// Post-physics step in one launch (mdp_post_physics_step)
// =============================================================================================
constexpr int POST_ENVS_MAX = 64;  // envs per manager block: one wave lane per env in the term phases

struct PostArgs {
    int N;
    // ---- manager role: blocks [0, nb_post), post_envs envs each.  They come first in the grid: a manager block is a
    // chain of dependent phases (the long pole of the launch), so it has to be resident from the start
    unsigned nb_post;
    int post_envs;
    // ---- cast role: the remaining blocks
    unsigned nb_cast;
    int cast_epb, cast_slot_w;
    RaycasterUpdateArgs cast;
    int has_upd;
    float physics_dt;
    mdp_scene_update_t upd;
    const mdp_term_t* tterms; int Tt; const int32_t* tpool;
    const mdp_term_t* rterms; int Tr; const int32_t* rpool;
    float step_dt;
    int64_t* len_inc;
    uint8_t* term_dones; uint8_t* truncated; uint8_t* terminated; uint8_t* dones;
    float* reward; float* sums; float* step;
    int has_cmd;
    mdp_velocity_command_args_t cmd;
    const mdp_term_t* oterms; int To; const int32_t* opool; const int16_t* col2term; int O;
    const int16_t* olists; int n_opcols, n_vterms;   // operand columns, then vector terms (mdp_program::n_opcols)
    float* obs; int ld_obs; uint64_t* orng; unsigned orng_base; int ouses_rng;
    mdp_obs_normalize_t onorm;
    int obs_waves;   // > 0: that many of a manager block's waves write the observations next to the reward phase
    // ---- reset branch (k_post_physics<.., true> only)
    const mdp_fill_t* fills; int num_fills; float max_episode_length_s; double* log_sums; float* log;
    // ---- terrain curriculum and event terms inside the reset branch (mdp_post_physics_step_with_reset_events)
    int has_cur, has_ev_root, has_ev_joints, has_ev_mark, has_ev_push;
    mdp_curriculum_terrain_t cur;
    mdp_event_root_state_t ev_root;
    mdp_event_joints_t ev_joints;
    mdp_event_reset_mark_t ev_mark;
    mdp_event_push_t ev_push;
    float* push_time_left; float push_lo, push_scale, interval_dt;
    const float* push_tick_u; uint64_t* push_tick_rng; uint8_t* push_trigger;
};

// Both roles of the kernel are chains of dependent memory round trips, so the waves a CU can hold are its throughput:
// the register budget is set for MDP_POST_MIN_WAVES waves per SIMD (the few values that do not fit live in scratch).
#ifndef MDP_POST_MIN_WAVES
#define MDP_POST_MIN_WAVES 6
#endif
// the instantiation with the reset branch keeps more live state: at 6 waves it spilled 160 B per lane; 5 waves (96
// registers, 16 B of scratch) is the faster budget for it (--auto-reset masked: 117.6 -> 114.1 us/step, 4 waves 117.7)
#ifndef MDP_POST_MIN_WAVES_RESET
#define MDP_POST_MIN_WAVES_RESET 5
#endif
#ifndef MDP_POST_MAX_THREADS
#define MDP_POST_MAX_THREADS 768
#endif
// launches with at least this many envs take the DENSE instantiation (MDP_POST_DENSE_ENVS overrides, 0 = never)
static int64_t post_dense_envs() {
    const char* v = std::getenv("MDP_POST_DENSE_ENVS");   // read at every call, like the other A/B knobs
    const int64_t n = v ? std::atoll(v) : 12288;
    return n > 0 ? n : INT64_MAX;
}
// per-env part of the ray caster's update for every env, in front of k_post_physics when the launch has many envs
__global__ void __launch_bounds__(256) k_cast_pose(RaycasterUpdateArgs a) { raycaster_pose_pass(a); }

// ---- observations (observation_manager.py:260-335), columns fastest: contiguous row writes.  Thread otid of othreads;
// cmd_lds: operands that lie in the command buffer are taken from s_cmd (the new command, not stored yet)
__device__ __forceinline__ void post_observations(const PostArgs& a, const mdp_term_t* oterms, const int16_t* col2term,
                                                  const int16_t* opcols, const int16_t* vterms, const float (*s_cmd)[64],
                                                  const int e0, const int e1, const unsigned b, const int otid,
                                                  const int othreads, const bool cmd_lds) {
    if (!a.obs) return;
    uint64_t seed = 0, epoch = 0;
    // own counters, tagged so that a draw never repeats one made by k_obs for the same element
    if (a.ouses_rng) { seed = a.orng[0]; epoch = a.orng[a.orng_base + b]; }
    const int O = a.O;
    const int ne = e1 - e0;
    auto emit = [&](const mdp_term_t& t, int ti, int ee, int c, int k, float v) {
        v = obs_post(t, v, ee, k, seed, epoch | (1ull << 62), 0x4f425300u + (uint32_t)ti, (uint64_t)ee * (uint64_t)O + c);
        a.obs[(int64_t)ee * a.ld_obs + c] = v;
        if (a.onorm.out)     // the standardised copy for the observation's consumer (mdp_obs_normalize_t)
            a.onorm.out[(int64_t)ee * a.onorm.ld_out + c] =
                tclamp((v - a.onorm.mean32[c]) / a.onorm.denom[c], -a.onorm.clip, a.onorm.clip);
    };
    // (1) operand columns, columns fastest.  A thread requests the operands of OBS_U of its items before it
    // finishes the first: the phase costs one memory round trip per OBS_U items instead of one per item (items
    // past the end repeat the last one's requests and store nothing)
    constexpr int OBS_U = 4;
    const int total = ne * a.n_opcols;
    // (2) body-frame vector terms, one thread per (env, term): their root-state rows are requested HERE, in front of
    // the operand columns' round trip, and rotated behind it -- the phase is one memory round trip, not two
    const int nvec = ne * a.n_vterms;
    const bool has_vec = otid < nvec;
    Q4 vq{1.0f, 0.0f, 0.0f, 0.0f};
    V3 vv{0.0f, 0.0f, 0.0f};
    int v_ti = 0, v_ee = 0;
    if (has_vec) {
        const int le = otid / a.n_vterms;
        v_ti = vterms[otid - le * a.n_vterms];
        v_ee = e0 + le;
        const mdp_term_t& t = oterms[v_ti];
        const float* r = (const float*)t.p0 + (int64_t)v_ee * t.i0;
        vq = ld4(r + 3);
        vv = t.op == MDP_OBS_PROJECTED_GRAVITY ? V3{t.f0, t.f1, t.f2} : ld3(r + (t.op == MDP_OBS_BASE_LIN_VEL ? 7 : 10));
    }
    for (int i0 = otid; i0 < total; i0 += OBS_U * othreads) {
        float va[OBS_U], vb[OBS_U], vc[OBS_U];
        int ees[OBS_U], cs[OBS_U];
#pragma unroll
        for (int u = 0; u < OBS_U; ++u) {
            int i = i0 + u * othreads;
            if (i >= total) i = total - 1;
            const int le = i / a.n_opcols;
            const int c = opcols[i - le * a.n_opcols];
            const mdp_term_t& t = oterms[col2term[c]];
            const float *pa, *pb, *pc;
            obs_operand_addrs(t, a.opool, e0 + le, c - t.out_offset, &pa, &pb, &pc);
            va[u] = *pa; vb[u] = *pb; vc[u] = *pc;
            if (cmd_lds) {          // (block-uniform) the new command waits in s_cmd until the rewards are through
                const float* cb = a.cmd.vel_command_b + 3 * (int64_t)e0;
                const uint64_t span = 3ull * (uint64_t)ne;
                const uint64_t oa = (uint64_t)(pa - cb), ob = (uint64_t)(pb - cb), oc = (uint64_t)(pc - cb);
                if (oa < span) va[u] = s_cmd[(unsigned)oa % 3u][(unsigned)oa / 3u];
                if (ob < span) vb[u] = s_cmd[(unsigned)ob % 3u][(unsigned)ob / 3u];
                if (oc < span) vc[u] = s_cmd[(unsigned)oc % 3u][(unsigned)oc / 3u];
            }
            ees[u] = e0 + le; cs[u] = c;
        }
#pragma unroll
        for (int u = 0; u < OBS_U; ++u) {
            if (i0 + u * othreads >= total) continue;
            const int ti = col2term[cs[u]];
            const mdp_term_t& t = oterms[ti];
            emit(t, ti, ees[u], cs[u], cs[u] - t.out_offset, obs_operand_finish(t, va[u], vb[u], vc[u]));
        }
    }
    if (has_vec) {
        const mdp_term_t& t = oterms[v_ti];
        const V3 v = quat_rotate_inverse(vq, vv);
        for (int k = 0; k < t.dim && k < 3; ++k)
            if (t.out_offset + k < O) emit(t, v_ti, v_ee, t.out_offset + k, k, get(v, k));
    }
    // (more (env, term) pairs than threads: the rest one round trip at a time)
    for (int i = otid + othreads; i < nvec; i += othreads) {
        const int le = i / a.n_vterms;
        const int ti = vterms[i - le * a.n_vterms];
        const mdp_term_t& t = oterms[ti];
        const int ee = e0 + le;
        const float* r = (const float*)t.p0 + (int64_t)ee * t.i0;
        V3 v = t.op == MDP_OBS_PROJECTED_GRAVITY ? V3{t.f0, t.f1, t.f2} : ld3(r + (t.op == MDP_OBS_BASE_LIN_VEL ? 7 : 10));
        v = quat_rotate_inverse(ld4(r + 3), v);
        for (int k = 0; k < t.dim && k < 3; ++k)
            if (t.out_offset + k < O) emit(t, ti, ee, t.out_offset + k, k, get(v, k));
    }
    if (a.ouses_rng && otid == 0) a.orng[a.orng_base + b] = epoch + 1;
}

// ROLES: the instantiation with wave roles in the manager blocks (a.obs_waves observation waves next to the reward
// phase); launches with cast blocks keep the instantiation without them (three-wave blocks: nothing to split)
// (used by launches without cast blocks, one 12-wave block per CU: a 128-register budget, no spills -- flat Anymal-D,
// 4096 envs: post-physics launch 16.8 us at 6 waves per SIMD, 15.9 us at 4)
#ifndef MDP_POST_MIN_WAVES_ROLES
#define MDP_POST_MIN_WAVES_ROLES 4
#endif
// ONE: the cast blocks are one env each with one ray per lane (raycaster_update_down_block_one)
// DENSE: the ONE instantiation with a register budget for MDP_POST_MIN_WAVES_DENSE waves per SIMD, for launches of many
// rounds of cast blocks (>= MDP_POST_DENSE_ENVS envs): there the launch is the cast's throughput = resident cast blocks
// / block life, the manager chain (which pays for the smaller budget with 56 B of scratch per lane) is hidden behind it.
// Same-box A/B (exp54): 16,384 envs post-physics 77.4 -> 71.9 us, step 181.3 -> 178.3 us; 65,536 envs 302 -> 288 us,
// step 598.7 -> 585.8 us; 8 waves the same or worse; at 8192 envs (H1) and below 6 waves stay faster.
#ifndef MDP_POST_MIN_WAVES_DENSE
#define MDP_POST_MIN_WAVES_DENSE 7
#endif
template <int BATCH, bool RESET, bool STAGED, bool ROLES, bool ONE = false, bool DENSE = false>
__global__ void __launch_bounds__(MDP_POST_MAX_THREADS, RESET ? MDP_POST_MIN_WAVES_RESET
                                                        : (ROLES ? MDP_POST_MIN_WAVES_ROLES
                                                                 : (DENSE ? MDP_POST_MIN_WAVES_DENSE : MDP_POST_MIN_WAVES)))
k_post_physics(PostArgs a) {
    static_assert(!DENSE || ONE, "the dense register budget is an instantiation of the one-env cast launch");
    static_assert(!(RESET && ROLES), "the reset branch orders rewards -> reset -> observations");
    static_assert(!(ONE && (STAGED || ROLES)), "one-env cast blocks: the per-ray walk, launches with a cast");
    extern __shared__ __align__(16) unsigned char s_dyn[];
    if (blockIdx.x >= a.nb_post) {
        MDP_STAMP(0, 1);
        if constexpr (STAGED)
            raycaster_update_down_blocks_staged(a.cast, a.cast_epb, a.cast_slot_w, blockIdx.x - a.nb_post, a.nb_cast, s_dyn);
        else if constexpr (ONE)
            raycaster_update_down_block_one<BATCH>(a.cast, blockIdx.x - a.nb_post);
        else
            raycaster_update_down_blocks<BATCH>(a.cast, a.cast_epb, a.cast_slot_w, blockIdx.x - a.nb_post, a.nb_cast);
        MDP_STAMP(1);
        return;
    }
    MDP_STAMP(0, 2);
    const unsigned b = blockIdx.x;
    const int N = a.N;
    const int E = a.post_envs;
    const int e0 = (int)b * E;
    const int e1 = e0 + E < N ? e0 + E : N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int tmax = a.Tt > a.Tr ? a.Tt : a.Tr;
    // LDS: [tmax][64] term values | observation, termination and reward term tables | column -> term map.  The tables
    // are staged first (their first use is behind the scene update), so no phase starts with a round trip for its terms
    float* s_val = reinterpret_cast<float*>(s_dyn);
    mdp_term_t* oterms = reinterpret_cast<mdp_term_t*>(s_dyn + (size_t)(tmax > 0 ? tmax : 1) * 64 * sizeof(float));
    mdp_term_t* tterms = oterms + a.To;
    mdp_term_t* rterms = tterms + a.Tt;
    int16_t* col2term = reinterpret_cast<int16_t*>(rterms + a.Tr);
    {
        constexpr int W = (int)(sizeof(mdp_term_t) / 4);
        const uint32_t* so = reinterpret_cast<const uint32_t*>(a.oterms);
        const uint32_t* st = reinterpret_cast<const uint32_t*>(a.tterms);
        const uint32_t* sr = reinterpret_cast<const uint32_t*>(a.rterms);
        uint32_t* dst = reinterpret_cast<uint32_t*>(oterms);
        const int wo = a.To * W, wt = a.Tt * W, wr = a.Tr * W;
        for (int i = threadIdx.x; i < wo + wt + wr; i += blockDim.x)
            dst[i] = i < wo ? so[i] : (i < wo + wt ? st[i - wo] : sr[i - wo - wt]);
        for (int i = threadIdx.x; i < a.O; i += blockDim.x) col2term[i] = a.col2term[i];
        for (int i = threadIdx.x; i < a.n_opcols + a.n_vterms; i += blockDim.x) col2term[a.O + i] = a.olists[i];
    }
    const int16_t* opcols = col2term + a.O;
    const int16_t* vterms = opcols + a.n_opcols;
    // reset variant: the fill records behind the column map (8-byte aligned), the block's done flags in s_done
    mdp_fill_t* fills = reinterpret_cast<mdp_fill_t*>(
        s_dyn + (((size_t)(reinterpret_cast<unsigned char*>(col2term + a.O + a.n_opcols + a.n_vterms) - s_dyn) + 7) & ~(size_t)7));
    __shared__ uint8_t s_done[64];
    __shared__ int s_last;
    if (RESET) {
        constexpr int FW = (int)(sizeof(mdp_fill_t) / 4);
        const uint32_t* sf = reinterpret_cast<const uint32_t*>(a.fills);
        uint32_t* df = reinterpret_cast<uint32_t*>(fills);
        for (int i = threadIdx.x; i < a.num_fills * FW; i += blockDim.x) df[i] = sf[i];
    }
    // ---- scene.update(physics_dt) for this block's envs (interactive_scene.py:496-513)
    if (a.has_upd) scene_update_envs(a.upd, a.physics_dt, e0, e1, true);
    MDP_STAMP(1);
    const int e = lane < E ? e0 + lane : N;   // lanes beyond the block's envs idle through the term phases
    // (reset branch with a terrain curriculum: the levels of ALL the block's envs go into the statistics; requested here,
    // by the wave that runs the curriculum, several phases ahead of their use)
    long long lvl_pre = 0;
    if (RESET && a.has_cur && wave == (nwaves > 1 ? 1 : 0) && e < N) lvl_pre = (long long)a.cur.terrain_levels[e];
    // ---- episode_length_buf += 1 (manager_based_rl_env.py:199)
    if (a.len_inc && wave == 0 && e < N) a.len_inc[e] += 1;
    __threadfence_block();
    __syncthreads();
    __shared__ float s_cmd[3][64];      // the new command of the block's envs until the rewards are through
    auto observations = [&](const int otid, const int othreads, const bool cmd_lds) {
        post_observations(a, oterms, col2term, opcols, vterms, s_cmd, e0, e1, b, otid, othreads, cmd_lds);
    };
    // ---- terminations (termination_manager.py:151-174): wave w evaluates terms w, w + nwaves, ...
    // Without the reset branch the command update (command_manager.py:151-187) is evaluated HERE, by the last wave, next
    // to the termination terms -- it needs the updated root state and nothing the terminations or rewards produce --
    // and only its result, the new command, is held back in registers until the rewards have read the previous one
    // (stored where the command phase used to be).  One dependent phase less in the block's chain.
    const bool cmd_early = !RESET && a.has_cmd;
    // Wave roles (no reset branch): the observations need the updated scene and the new command, nothing the terminations
    // or rewards produce, so the block's last a.obs_waves waves write them WHILE the other waves evaluate the reward terms
    // (they read the new command from s_cmd: the command buffer itself changes only once the rewards are through).
    // The chain update -> terminations -> rewards -> observations loses its last link.
    const int nB = (ROLES && a.obs && a.obs_waves > 0 && a.obs_waves < nwaves) ? a.obs_waves : 0;
    const int nA = nwaves - nB;          // waves of the termination / reward phases
    const bool obs_wave = nB > 0 && wave >= nA;
    if (cmd_early && wave == nwaves - 1) {
        uint64_t seed = 0, epoch = 0;
        const bool gen = a.cmd.rng_state != nullptr && a.cmd.injected_u == nullptr;
        if (a.cmd.rng_state) { seed = a.cmd.rng_state[0]; epoch = a.cmd.rng_state[4 + b]; }
        if (e < N) {
            float cmd_new[3];
            velocity_command_env(a.cmd.cfg, e, a.cmd.root_state, a.cmd.ld_state, a.cmd.vel_command_b, a.cmd.heading_target,
                                 a.cmd.is_heading_env, a.cmd.is_standing_env, a.cmd.time_left, a.cmd.command_counter,
                                 a.cmd.metric_error_vel_xy, a.cmd.metric_error_vel_yaw, a.cmd.dt, seed, epoch | (1ull << 62),
                                 a.cmd.injected_u, nullptr, cmd_new);
            s_cmd[0][lane] = cmd_new[0]; s_cmd[1][lane] = cmd_new[1]; s_cmd[2][lane] = cmd_new[2];
        }
        if (gen && lane == 0) a.cmd.rng_state[4 + b] = epoch + 1;
    }
    for (int ti = obs_wave ? a.Tt : wave; ti < a.Tt; ti += nA) {
        const mdp_term_t& t = tterms[ti];
        bool v = e < N ? term_value(t, a.tpool, e) : false;
        if (a.term_dones && e < N) a.term_dones[(int64_t)t.out_offset * N + e] = v ? 1 : 0;
        s_val[ti * 64 + lane] = v ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (wave == 0 && e < N && a.Tt > 0) {
        bool trunc = false, term = false;
        for (int ti = 0; ti < a.Tt; ++ti) {
            bool v = s_val[ti * 64 + lane] != 0.0f;
            if (tterms[ti].i3) trunc = trunc || v; else term = term || v;
        }
        if (a.truncated) a.truncated[e] = trunc ? 1 : 0;
        if (a.terminated) a.terminated[e] = term ? 1 : 0;
        if (a.dones) a.dones[e] = (trunc || term) ? 1 : 0;
        if (RESET) s_done[lane] = (trunc || term) ? 1 : 0;
    }
    __threadfence_block();
    __syncthreads();
    MDP_STAMP(2);
    // ---- rewards (reward_manager.py:128-157)
    for (int ti = obs_wave ? a.Tr : wave; ti < a.Tr; ti += nA) {
        const mdp_term_t& t = rterms[ti];
        float value = 0.0f;
        if (e < N && t.weight != 0.0f) {
            value = reward_value(t, a.rpool, e, N) * t.weight * a.step_dt;
            if (a.sums) a.sums[(int64_t)t.out_offset * N + e] += value;
            if (a.step) a.step[(int64_t)e * a.Tr + t.out_offset] = value / a.step_dt;
        }
        s_val[ti * 64 + lane] = value;
    }
    // ---- observations (observation_manager.py:260-335) with wave roles: by the observation waves next to the reward
    // terms (or, a.obs_waves == 0, by every wave behind its reward terms); the new command comes from s_cmd
    if constexpr (ROLES) {
        if (nB == 0 || obs_wave)
            observations((int)threadIdx.x - (nB == 0 ? 0 : nA * 64), nB == 0 ? (int)blockDim.x : nB * 64, cmd_early);
    }
    __syncthreads();
    if (wave == 0 && e < N && a.Tr > 0) {
        float total = 0.0f;
        for (int ti = 0; ti < a.Tr; ++ti)
            if (rterms[ti].weight != 0.0f) total += s_val[ti * 64 + lane];
        a.reward[e] = total;
    }
    // ---- reset of the envs that are done (manager_based_rl_env.py:214-229, 346-391): between rewards and commands
    if (RESET) {
        __threadfence_block();
        __syncthreads();                                  // episode sums and reward of every env of the block are written
        // statistics of the reset envs: row r of {count, Tr episode sums, 2 command metrics, Tt termination counts},
        // rows over the waves, envs over the lanes (one load per lane, no chain), wave sum
        const int rows = 1 + a.Tr + 2 + a.Tt;
        const int slot_rows = rows + (a.has_cur ? 1 : 0);  // + the terrain levels of ALL the block's envs (curriculums.py:55)
        const bool mine = e < N && s_done[lane] != 0;
        __shared__ uint8_t s_list[64];                    // local ids of the block's done envs, s_nd of them
        __shared__ int s_nd;
        // every block stores its row sums in a slot of its own (no atomics: the sums do not depend on block timing)
        double* part = a.log_sums + 1 + (size_t)b * slot_rows;
        if (wave == 0) {
            const unsigned long long m = __ballot(mine);
            if (mine) s_list[__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)lane;
            if (lane == 0) s_nd = __popcll(m);
        }
        // CurriculumManager.compute(env_ids) comes first in _reset_idx (manager_based_rl_env.py:353): it reads the root
        // position and the command of the episode that ends, and moves the env origin the root-state event reads.  Both,
        // and EventManager's bookkeeping (event_manager.py:235-240), run HERE for the done envs, on the lanes of the
        // block's second wave while the other waves fetch the statistics rows: nothing they write is read by the
        // statistics, the fills or CommandTerm.reset, and the command update / observations sit behind two barriers
        if (wave == (nwaves > 1 ? 1 : 0)) {
            // everything the two routines read that depends on nothing they write is requested first: the done envs'
            // chain is loads -> terrain origin of the new level -> stores, not one round trip per statement
            TerrainLevelsIn cin{};
            RootStateIn rin{};
            const int64_t lvl_old = (int64_t)lvl_pre;
            if (mine && a.has_cur) cin = terrain_levels_vel_load(e, a.cur);
            if (mine && a.has_ev_root) rin = event_reset_root_state_load(a.ev_root, e);
            V3 org{0.0f, 0.0f, 0.0f};
            if (mine && a.has_ev_root && !a.has_cur) org = ld3(a.ev_root.env_origins + 3 * (int64_t)e);
            // the new level's origin row is requested by the decision; the root-state event's draws (three Philox blocks,
            // three sincos, a quaternion product: ~3 us of dependent arithmetic on one lane) run while it is in flight
            int64_t lvl = lvl_old;
            if (mine && a.has_cur) lvl = terrain_levels_vel_decide(e, a.cur, cin, &org);
            RootStateDraw rdraw{};
            if (mine && a.has_ev_root) rdraw = event_reset_root_state_draw(a.ev_root, e, rin);
            if (mine && a.has_cur) terrain_levels_vel_commit(e, a.cur, org);
            if (a.has_cur) {
                double lv = e < N ? (double)lvl : 0.0;
                for (int o = 32; o > 0; o >>= 1) lv += __shfl_xor(lv, o);
                if (lane == 0)
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(part + rows),
                                       (unsigned long long)__double_as_longlong(lv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (mine) {
                if (a.has_ev_root) event_reset_root_state_finish(a.ev_root, e, rin, rdraw, org);     // manager_based_rl_env.py:357-359
                if (a.has_ev_mark) {
                    // (constant indices into the argument struct: a runtime index would move the whole struct to scratch)
#pragma unroll
                    for (int t = 0; t < MDP_EVENT_MARK_MAX_TERMS; ++t)
                        if (t < a.ev_mark.num_terms) {
                            a.ev_mark.last_triggered_step[t][e] = a.ev_mark.step_count;
                            a.ev_mark.triggered_once[t][e] = 1;
                        }
                }
            }
        }
#ifndef MDP_POST_ROWS_B
#define MDP_POST_ROWS_B 4
#endif
        constexpr int ROWS_B = MDP_POST_ROWS_B;
        for (int r0 = wave; r0 < rows; r0 += ROWS_B * nwaves) {
            // ROWS_B rows at a time: their loads are issued back to back, then the wave sums are taken (same box, us/step
            // with masked resets / with events: 105.0 / 112.1 with 4 rows, 105.7 / 112.6 with 12)
            float v[ROWS_B];
#pragma unroll
            for (int j = 0; j < ROWS_B; ++j) {
                const int r = r0 + j * nwaves;
                v[j] = 0.0f;
                if (mine && r < rows) {
                    if (r == 0) v[j] = 1.0f;
                    else if (r <= a.Tr) v[j] = a.sums[(int64_t)rterms[r - 1].out_offset * N + e];
                    else if (r == a.Tr + 1) v[j] = a.has_cmd ? a.cmd.metric_error_vel_xy[e] : 0.0f;
                    else if (r == a.Tr + 2) v[j] = a.has_cmd ? a.cmd.metric_error_vel_yaw[e] : 0.0f;
                    else v[j] = a.term_dones[(int64_t)tterms[r - a.Tr - 3].out_offset * N + e] ? 1.0f : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < ROWS_B; ++j) {
                const int r = r0 + j * nwaves;
                double d = (double)v[j];
                for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
                // device-scope store: written through to where the last block reads it.  (A __threadfence() here would
                // write back the whole XCD's L2 -- everything the cast blocks have produced so far -- once per block.)
                if (lane == 0 && r < rows)
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(part + r),
                                       (unsigned long long)__double_as_longlong(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);                    // those stores are acknowledged before the block takes its ticket
        __syncthreads();                                  // statistics read before anything is zeroed
#ifdef MDP_STAMPS_RESET_FINE
        MDP_STAMP(1);                                     // (diagnostic: slots 1 / 2 re-used for the reset section)
#endif
        // the last manager block to get here will add the slots up and publish the log values -- at the very end of its
        // own work: the sums sit in memory, several round trips away, and nothing in this launch waits for the log
        unsigned long long* ticket = reinterpret_cast<unsigned long long*>(a.log_sums);
        if (threadIdx.x == 0)
            s_last = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull ==
                     (unsigned long long)a.nb_post;
        // the buffers _reset_idx zeroes, for the done envs of this block (mdp_fill_t records, see mdp_masked_fill)
        // (done env, buffer) pairs over the waves, lanes over the elements (nothing to do in a block without done envs)
        // Wave 0 goes straight to CommandTerm.reset below (a round trip of its own); the other waves share the fills and
        // the joint-state event
        const int nd = s_nd;
        const int fwaves = (a.has_cmd && nwaves > 1) ? nwaves - 1 : nwaves;
        const int fwave = fwaves < nwaves ? wave - 1 : wave;
        for (int idx = fwave < 0 ? nd * a.num_fills : fwave; idx < nd * a.num_fills; idx += fwaves) {
            const int j = idx / a.num_fills, fi = idx - j * a.num_fills;
            const int ee = e0 + s_list[j];
            const mdp_fill_t& f = fills[fi];
            const int64_t base = (int64_t)ee * f.env_stride;
            for (int k = lane; k < f.count; k += 64) {
                const int64_t at = base + (int64_t)k * f.elem_stride;
                if (f.elem_bytes == 4) reinterpret_cast<uint32_t*>(f.ptr)[at] = (uint32_t)f.value_bits;
                else if (f.elem_bytes == 8) reinterpret_cast<uint64_t*>(f.ptr)[at] = f.value_bits;
                else reinterpret_cast<uint8_t*>(f.ptr)[at] = (uint8_t)f.value_bits;
            }
        }
        // the joint-state event of the done envs (events.py:802-866): the same waves over the envs (last wave first),
        // lanes over the joints
        if (a.has_ev_joints && fwave >= 0) {
            for (int j0 = fwaves - 1 - fwave; j0 < nd; j0 += fwaves) {
                const int ee = e0 + s_list[j0];
                uint64_t jseed = 0, jepoch = 0;
                if (!a.ev_joints.injected_u) { jseed = a.ev_joints.rng_state[0]; jepoch = a.ev_joints.rng_state[1 + ee]; }
                // the first 64 joints' inputs are requested with the generator words (one round trip)
                const bool first = lane < a.ev_joints.num_joints;
                JointResetIn jin{};
                if (first) jin = event_reset_joint_load(a.ev_joints, ee, lane);
                // (the store below carries the loaded epoch: every lane's load has returned before it is issued)
                if (!a.ev_joints.injected_u && lane == 0) a.ev_joints.rng_state[1 + ee] = jepoch + 1;
                if (first) event_reset_joint_apply(a.ev_joints, ee, lane, jseed, jepoch, jin);
                for (int j = lane + 64; j < a.ev_joints.num_joints; j += 64) event_reset_joint(a.ev_joints, ee, j, jseed, jepoch);
            }
        }
        // CommandTerm.reset (command_manager.py:120-149): metrics, counter, resample
        if (a.has_cmd && wave == 0 && e < N && s_done[lane]) {
            uint64_t seed = 0, epoch = 0;
            if (a.cmd.rng_state) { seed = a.cmd.rng_state[0]; epoch = a.cmd.rng_state[4 + b]; }
            velocity_command_env(a.cmd.cfg, e, a.cmd.root_state, a.cmd.ld_state, a.cmd.vel_command_b, a.cmd.heading_target,
                                 a.cmd.is_heading_env, a.cmd.is_standing_env, a.cmd.time_left, a.cmd.command_counter,
                                 a.cmd.metric_error_vel_xy, a.cmd.metric_error_vel_yaw, 0.0f, seed, epoch | (3ull << 61),
                                 a.cmd.injected_u, a.dones);
        }
        __threadfence_block();
#ifdef MDP_STAMPS_RESET_FINE
        MDP_STAMP(2);
#endif
    }
    // ---- command_manager.compute(step_dt) (command_manager.py:151-187, velocity_command.py:111-160): after the
    // rewards, which read the previous command
    __syncthreads();
    MDP_STAMP(3);
    if (cmd_early) {
        if (wave == nwaves - 1 && e < N) {
            a.cmd.vel_command_b[3 * e + 0] = s_cmd[0][lane];
            a.cmd.vel_command_b[3 * e + 1] = s_cmd[1][lane];
            a.cmd.vel_command_b[3 * e + 2] = s_cmd[2][lane];
        }
        __threadfence_block();
        __syncthreads();
    } else if (RESET && a.has_cmd) {
        uint64_t seed = 0, epoch = 0;
        const bool gen = a.cmd.rng_state != nullptr && a.cmd.injected_u == nullptr;
        if (a.cmd.rng_state) { seed = a.cmd.rng_state[0]; epoch = a.cmd.rng_state[4 + b]; }
        // (the interval term's timer is requested in front of the command update, used behind it)
        const bool ticks = a.has_ev_push && wave == 0 && e < N;
        const float push_tl = ticks ? a.push_time_left[e] : 0.0f;
        PushIn pin{};
        if (ticks) pin = event_push_load(a.ev_push, e);      // (the command update does not write the root state)
        if (wave == 0 && e < N)
            velocity_command_env(a.cmd.cfg, e, a.cmd.root_state, a.cmd.ld_state, a.cmd.vel_command_b, a.cmd.heading_target,
                                 a.cmd.is_heading_env, a.cmd.is_standing_env, a.cmd.time_left, a.cmd.command_counter,
                                 a.cmd.metric_error_vel_xy, a.cmd.metric_error_vel_yaw, a.cmd.dt, seed, epoch | (1ull << 62),
                                 a.cmd.injected_u, nullptr);
        if (gen && threadIdx.x == 0) a.cmd.rng_state[4 + b] = epoch + 1;
        // EventManager.apply(mode="interval") (manager_based_rl_env.py:233-235): behind the command update, in front of
        // the observations; per-env timers (event_manager.py:218-225)
        if (a.has_ev_push && wave == 0 && e < N) {
            const bool fire = event_interval_tick_env(a.push_time_left, a.interval_dt, a.push_lo, a.push_scale, a.push_tick_u,
                                                      a.push_tick_rng, e, push_tl);
            if (a.push_trigger) a.push_trigger[e] = fire ? 1 : 0;
            if (fire) event_push_apply(a.ev_push, e, pin);
        }
        __threadfence_block();
        __syncthreads();
    }
    MDP_STAMP(4);
    // ---- observations without wave roles: behind the (reset and the) command update, from the buffers
    if constexpr (!ROLES) observations((int)threadIdx.x, (int)blockDim.x, false);
    MDP_STAMP(6);
    if (RESET) {
        if (s_last) {                                     // block-uniform, written before the command phase's barriers
            // thread (r, p) adds the slots of blocks p, p + P, ... for row r (independent, coalesced loads, 32 in flight),
            // then the P partial sums of a row are added in order: the log does not depend on block timing
            __shared__ double s_red[256];
            const int stat_rows = 1 + a.Tr + 2 + a.Tt;
            const int rows = stat_rows + (a.has_cur ? 1 : 0);   // slot layout of the blocks: + the terrain-level sums
            double* s_sum = reinterpret_cast<double*>(s_val);
            const unsigned long long* all = reinterpret_cast<const unsigned long long*>(a.log_sums + 1);
            int P = (blockDim.x < 256 ? (int)blockDim.x : 256) / rows;
            if (P < 1) P = 1;
            __syncthreads();                              // s_val: the last reader was the reward total
            if ((int)threadIdx.x < rows * P) {
                const int p = threadIdx.x / rows, r = threadIdx.x - p * rows;
                double v = 0.0;
                for (unsigned b0 = p; b0 < a.nb_post; b0 += 8u * P) {
                    unsigned long long w[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned bb = b0 + (unsigned)j * P;
                        w[j] = bb < a.nb_post ? __hip_atomic_load(all + (size_t)bb * rows + r, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v += __longlong_as_double((long long)w[j]);
                }
                s_red[threadIdx.x] = v;
            }
            __syncthreads();
            if ((int)threadIdx.x < rows) {
                double v = 0.0;
                for (int p = 0; p < P; ++p) v += s_red[p * rows + threadIdx.x];
                s_sum[threadIdx.x] = v;
            }
            __syncthreads();
            if ((int)threadIdx.x < rows) {
                const int r = threadIdx.x;
                const double cnt = s_sum[0], sum = s_sum[r];
                float v;
                if (r == stat_rows) v = (float)sum * (1.0f / (float)N);     // torch.mean(terrain_levels.float()), curriculums.py:55
                else if (r == 0 || r > a.Tr + 2) v = (float)sum;            // counts
                else {
                    v = cnt > 0.0 ? (float)(sum / cnt) : 0.0f;              // torch.mean over the reset envs
                    if (r <= a.Tr) v = v / a.max_episode_length_s;
                }
                a.log[r] = v;
            }
            if (threadIdx.x == 0)                                           // ready for the next launch
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.log_sums), 0ull, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    MDP_STAMP(5);
}

// ---- single-term evaluation: the term travels as a kernel argument (no allocation, no sync) ----
__global__ void k_obs_single(mdp_term_t t, const int32_t* __restrict__ pool, int N, float* __restrict__ out, int ld,
                             uint64_t seed, uint64_t epoch) {
    int64_t total = (int64_t)N * t.dim;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int e = (int)(i / t.dim);
        int k = (int)(i - (int64_t)e * t.dim);
        float v = obs_value(t, pool, e, k);
        out[(int64_t)e * ld + k] = obs_post(t, v, e, k, seed, epoch, 0x53494e47u, (uint64_t)i);
    }
}
__global__ void k_reward_single(mdp_term_t t, const int32_t* __restrict__ pool, int N, float* __restrict__ out) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N; e += gridDim.x * blockDim.x)
        out[e] = reward_value(t, pool, e, N);
}
__global__ void k_term_single(mdp_term_t t, const int32_t* __restrict__ pool, int N, uint8_t* __restrict__ out) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N; e += gridDim.x * blockDim.x)
        out[e] = term_value(t, pool, e) ? 1 : 0;
}

}  // namespace

// =============================================================================================
// Host API
// =============================================================================================
static bool obs_op_is_operand_host(int op) {
    return op == MDP_OBS_COPY || op == MDP_OBS_SUB || op == MDP_OBS_ADD_ROW_BIAS || op == MDP_OBS_ROOT_QUAT_UNIQUE ||
           op == MDP_OBS_JOINT_POS_LIMIT_NORMALIZED || op == MDP_OBS_HEIGHT_SCAN;
}
static bool obs_op_is_vector_host(int op) {
    return op == MDP_OBS_BASE_LIN_VEL || op == MDP_OBS_BASE_ANG_VEL || op == MDP_OBS_PROJECTED_GRAVITY;
}

extern "C" int mdp_program_create(int kind, int device, int num_envs, const mdp_term_t* terms, int num_terms,
                                  const int32_t* index_pool, int index_pool_len, uint64_t seed,
                                  mdp_program_t** out) {
    MDP_REQUIRE(out, "null out pointer");
    MDP_REQUIRE(kind >= MDP_PROG_OBS && kind <= MDP_PROG_TERMINATION, "unknown program kind %d", kind);
    MDP_REQUIRE(num_envs >= 0 && num_terms >= 0, "negative size");
    MDP_REQUIRE(num_terms == 0 || terms, "null terms");
    MDP_CHECK_HIP(hipSetDevice(device));
    mdp_program* p = new mdp_program();
    p->kind = kind;
    p->device = device;
    p->num_envs = num_envs;
    p->num_terms = num_terms;
    p->h_terms.assign(terms, terms + num_terms);
    // validate + derived tables
    std::vector<int16_t> col2term;
    std::vector<RayAux> aux;
    const int nops = kind == MDP_PROG_OBS ? MDP_OBS_NUM_OPS : (kind == MDP_PROG_REWARD ? MDP_REW_NUM_OPS : MDP_TERM_NUM_OPS);
    for (int i = 0; i < num_terms; ++i) {
        mdp_term_t& t = p->h_terms[i];
        if (t.op < 0 || t.op >= nops) { delete p; return fail(MDP_ERR_INVALID, "term %d: unknown opcode %d", i, t.op); }
        if (t.n_idx > 0 && (t.idx_offset < 0 || t.idx_offset + t.n_idx > index_pool_len)) {
            delete p;
            return fail(MDP_ERR_INVALID, "term %d: index list out of the pool", i);
        }
        if (t.n_idx2 > 0 && (t.idx2_offset < 0 || t.idx2_offset + t.n_idx2 > index_pool_len)) {
            delete p;
            return fail(MDP_ERR_INVALID, "term %d: second index list out of the pool", i);
        }
        if (kind == MDP_PROG_OBS) {
            if (t.dim <= 0 || t.out_offset < 0) { delete p; return fail(MDP_ERR_INVALID, "term %d: bad dim/offset", i); }
            if (t.out_offset + t.dim > 32767 * 4) { delete p; return fail(MDP_ERR_INVALID, "observation too wide"); }
            if ((int)col2term.size() < t.out_offset + t.dim) col2term.resize(t.out_offset + t.dim, -1);
            for (int k = 0; k < t.dim; ++k) col2term[t.out_offset + k] = (int16_t)i;
            if (t.op != MDP_OBS_SKIP && t.out_offset + t.dim > p->active_cols) p->active_cols = t.out_offset + t.dim;
            if (t.noise_kind == 1 || t.noise_kind == 2) p->uses_rng |= (t.noise_u == nullptr);
            if (t.op == MDP_OBS_HEIGHT_SCAN_CAST) {
                const mdp_raycaster_desc_t* d = (const mdp_raycaster_desc_t*)t.p4;
                if (!d || !d->mesh || !d->root_pose || !d->ray_starts_local || !d->ray_dirs_local || !d->pos_w ||
                    !d->quat_w || !d->ray_hits_w || d->num_rays != t.dim || d->ld_pose < 7) {
                    delete p;
                    return fail(MDP_ERR_INVALID, "term %d: invalid ray-caster binding", i);
                }
                RayAux a;
                a.mesh = d->mesh->view;
                a.d = *d;
                a.d.mesh = nullptr;
                aux.push_back(a);
                if (!p->has_cast) { p->cast_ti = i; p->h_cast = a; }
                p->has_cast = 1;
            }
        } else {
            if (t.out_offset < 0 || t.out_offset >= num_terms) {
                delete p;
                return fail(MDP_ERR_INVALID, "term %d: slot %d out of range", i, t.out_offset);
            }
        }
    }
    for (size_t c = 0; c < col2term.size(); ++c)
        if (col2term[c] < 0) { delete p; return fail(MDP_ERR_INVALID, "observation column %zu not covered by a term", c); }
    p->total_cols = (int)col2term.size();
    // upload
    auto dmalloc = [](void** d, size_t n) { return hipMalloc(d, n ? n : 16); };
    if (!aux.empty()) {
        MDP_CHECK_HIP(dmalloc((void**)&p->d_aux, aux.size() * sizeof(RayAux)));
        MDP_CHECK_HIP(hipMemcpy(p->d_aux, aux.data(), aux.size() * sizeof(RayAux), hipMemcpyHostToDevice));
        size_t ai = 0;
        for (auto& t : p->h_terms)
            if (kind == MDP_PROG_OBS && t.op == MDP_OBS_HEIGHT_SCAN_CAST) t.p4 = p->d_aux + ai++;
    }
    MDP_CHECK_HIP(dmalloc((void**)&p->d_terms, sizeof(mdp_term_t) * (size_t)num_terms));
    if (num_terms)
        MDP_CHECK_HIP(hipMemcpy(p->d_terms, p->h_terms.data(), sizeof(mdp_term_t) * (size_t)num_terms,
                                hipMemcpyHostToDevice));
    MDP_CHECK_HIP(dmalloc((void**)&p->d_pool, sizeof(int32_t) * (size_t)index_pool_len));
    if (index_pool_len)
        MDP_CHECK_HIP(hipMemcpy(p->d_pool, index_pool, sizeof(int32_t) * (size_t)index_pool_len, hipMemcpyHostToDevice));
    if (!col2term.empty()) {
        std::vector<int16_t> tab(col2term);
        for (int c = 0; c < p->active_cols; ++c)
            if (obs_op_is_operand_host(p->h_terms[col2term[c]].op)) { tab.push_back((int16_t)c); ++p->n_opcols; }
        for (int i = 0; i < num_terms; ++i)
            if (obs_op_is_vector_host(p->h_terms[i].op) && p->h_terms[i].out_offset < p->active_cols) {
                tab.push_back((int16_t)i);
                ++p->n_vterms;
            }
        MDP_CHECK_HIP(dmalloc((void**)&p->d_col2term, sizeof(int16_t) * tab.size()));
        MDP_CHECK_HIP(hipMemcpy(p->d_col2term, tab.data(), sizeof(int16_t) * tab.size(), hipMemcpyHostToDevice));
    }
    {
        // {seed, epoch of block 0, epoch of block 1, ...}: one counter per block of the observation grid
        p->rng_post_base = 1 + std::max(obs_grid(p->num_envs, p->total_cols), obs_grid(p->num_envs, p->active_cols));
        std::vector<uint64_t> rng((size_t)p->rng_post_base + (size_t)p->num_envs + 1, 0);  // post_envs >= 1
        rng[0] = seed;
        MDP_CHECK_HIP(dmalloc((void**)&p->d_rng, sizeof(uint64_t) * rng.size()));
        MDP_CHECK_HIP(hipMemcpy(p->d_rng, rng.data(), sizeof(uint64_t) * rng.size(), hipMemcpyHostToDevice));
    }
    *out = p;
    return MDP_OK;
}

extern "C" int mdp_program_destroy(mdp_program_t* p) {
    if (!p) return MDP_OK;
    (void)hipSetDevice(p->device);
    (void)hipFree(p->d_terms);
    (void)hipFree(p->d_pool);
    (void)hipFree(p->d_aux);
    (void)hipFree(p->d_col2term);
    (void)hipFree(p->d_rng);
    delete p;
    return MDP_OK;
}

extern "C" int mdp_obs_compute(mdp_program_t* p, float* obs, int ld_obs, void* stream) {
    MDP_REQUIRE(p && p->kind == MDP_PROG_OBS, "not an observation program");
    MDP_REQUIRE(obs && ld_obs >= p->total_cols, "obs buffer too narrow (%d < %d)", ld_obs, p->total_cols);
    if (p->num_envs == 0 || p->total_cols == 0 || p->active_cols == 0) return MDP_OK;
    // columns behind the last one this program writes (height scans written by their cast kernels) get no threads
    int O = p->active_cols;
    int epb = O >= 256 ? 1 : 256 / O;
    int slot_w = O >= 256 ? 256 : O;
    int threads = ((epb * slot_w + 63) / 64) * 64;
    size_t lds = (size_t)p->num_terms * sizeof(mdp_term_t) + (size_t)O * sizeof(int16_t);
    if (lds > 48 * 1024) return fail(MDP_ERR_UNSUPPORTED, "observation program too large for the LDS term table (%zu B)", lds);
    // programs without a fused ray cast run the lean instantiation (no ray code: 1/3 of the registers)
    if (p->has_cast)
        hipLaunchKernelGGL(k_obs<true>, dim3(obs_grid(p->num_envs, O)), dim3(threads), lds, as_stream(stream), p->d_terms,
                           p->num_terms, p->d_pool, p->d_col2term, O, p->num_envs, epb, slot_w, obs, ld_obs, p->d_rng,
                           p->uses_rng, p->has_cast, p->cast_ti, p->h_cast);
    else
        hipLaunchKernelGGL(k_obs<false>, dim3(obs_grid(p->num_envs, O)), dim3(threads), lds, as_stream(stream), p->d_terms,
                           p->num_terms, p->d_pool, p->d_col2term, O, p->num_envs, epb, slot_w, obs, ld_obs, p->d_rng,
                           p->uses_rng, 0, -1, p->h_cast);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_reward_compute(mdp_program_t* p, float dt, float* reward_buf, float* episode_sums,
                                  float* step_reward, void* stream) {
    MDP_REQUIRE(p && p->kind == MDP_PROG_REWARD, "not a reward program");
    MDP_REQUIRE(reward_buf, "null reward buffer");
    if (p->num_envs == 0) return MDP_OK;
    hipStream_t s = as_stream(stream);
    int waves = p->num_terms < 1 ? 1 : (p->num_terms > 16 ? 16 : p->num_terms);
    size_t lds = (size_t)(p->num_terms > 0 ? p->num_terms : 1) * 64 * sizeof(float);
    if (lds > 60 * 1024) return fail(MDP_ERR_UNSUPPORTED, "reward program with %d terms exceeds the LDS staging area", p->num_terms);
    hipLaunchKernelGGL(k_reward, dim3(grid_for(p->num_envs, 64, 1 << 20)), dim3(64 * waves), lds, s, p->d_terms,
                       p->num_terms, p->d_pool, p->num_envs, dt, reward_buf, episode_sums, step_reward);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_termination_compute(mdp_program_t* p, uint8_t* term_dones, uint8_t* truncated,
                                       uint8_t* terminated, uint8_t* dones, void* stream) {
    MDP_REQUIRE(p && p->kind == MDP_PROG_TERMINATION, "not a termination program");
    if (p->num_envs == 0) return MDP_OK;
    hipLaunchKernelGGL(k_termination, dim3(grid_for(p->num_envs, 64, 1 << 20)), dim3(64), 0, as_stream(stream),
                       p->d_terms, p->num_terms, p->d_pool, p->num_envs, term_dones, truncated, terminated, dones);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_termination_reward_compute(mdp_program_t* tp, mdp_program_t* rp, float dt,
                                              int64_t* episode_length_buf_inc, uint8_t* term_dones,
                                              uint8_t* truncated, uint8_t* terminated, uint8_t* dones,
                                              float* reward_buf, float* episode_sums, float* step_reward,
                                              void* stream) {
    MDP_REQUIRE(tp && tp->kind == MDP_PROG_TERMINATION, "not a termination program");
    MDP_REQUIRE(rp && rp->kind == MDP_PROG_REWARD, "not a reward program");
    MDP_REQUIRE(tp->num_envs == rp->num_envs && tp->device == rp->device, "programs disagree on num_envs / device");
    MDP_REQUIRE(reward_buf || rp->num_terms == 0, "null reward buffer");
    if (tp->num_envs == 0) return MDP_OK;
    int tmax = tp->num_terms > rp->num_terms ? tp->num_terms : rp->num_terms;
    int waves = tmax < 1 ? 1 : (tmax > 16 ? 16 : tmax);
    size_t lds = (size_t)(tmax > 0 ? tmax : 1) * 64 * sizeof(float);
    if (lds > 60 * 1024) return fail(MDP_ERR_UNSUPPORTED, "%d terms exceed the LDS staging area", tmax);
    hipLaunchKernelGGL(k_term_reward, dim3(grid_for(tp->num_envs, 64, 1 << 20)), dim3(64 * waves), lds,
                       as_stream(stream), tp->d_terms, tp->num_terms, tp->d_pool, rp->d_terms, rp->num_terms, rp->d_pool,
                       tp->num_envs, dt, episode_length_buf_inc, term_dones, truncated, terminated, dones, reward_buf,
                       episode_sums, step_reward);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

MDP_STAMP_SETTER(mdp_debug_stamps_programs)

static int post_physics_launch(int num_envs, const mdp_scene_update_t* update, float physics_dt,
                               const mdp_termination_reward_args_t* tr, const mdp_velocity_command_args_t* command,
                               const mdp_obs_args_t* observation, const mdp_raycaster_update_args_t* cast,
                               const mdp_post_reset_args_t* reset, void* stream, bool clear_ticket = false,
                               const mdp_post_events_t* events = nullptr) {
    MDP_REQUIRE(num_envs >= 0, "negative num_envs");
    MDP_REQUIRE(tr && tr->termination && tr->reward, "termination / reward programs required");
    mdp_program_t* tp = tr->termination;
    mdp_program_t* rp = tr->reward;
    MDP_REQUIRE(tp->kind == MDP_PROG_TERMINATION && rp->kind == MDP_PROG_REWARD, "wrong program kinds");
    MDP_REQUIRE(tp->num_envs == num_envs && rp->num_envs == num_envs, "programs disagree on num_envs");
    MDP_REQUIRE(tr->reward_buf || rp->num_terms == 0, "null reward buffer");
    if (num_envs == 0) return MDP_OK;
    PostArgs a;
    std::memset(&a, 0, sizeof(a));
    a.N = num_envs;
    static const int threads_env = [] {
        const char* v = std::getenv("MDP_POST_THREADS");
        int t = v ? std::atoi(v) : 0;      // 0: whole waves covering one cast slot group (<= 256 threads)
        return t <= 0 ? 0 : (t < 64 ? 64 : (t > MDP_POST_MAX_THREADS ? MDP_POST_MAX_THREADS : (t / 64) * 64));
    }();
    static const int post_envs_forced = [] {
        const char* v = std::getenv("MDP_POST_ENVS");
        int e = v ? std::atoi(v) : 0;
        return e < 1 ? 0 : (e > POST_ENVS_MAX ? POST_ENVS_MAX : e);
    }();
    // envs per manager block, measured on MI355X with the whole-step bench: 16 at 4096 envs (in
    // the step the manager chain outlasts the cast: 74.6 us/step with 24, 73.3 with 16, 73.5 with 12, 74.9 with 8), a full wave of lanes per term once the cast runs for several rounds (fewer manager blocks
    // holding wave slots the cast could use; whole-step bench: 204.2 -> 202.0 us at 16384 envs with 32, 743 -> 729 us at
    // 65536 envs with 64; at 8192 envs 24 is still best)
    // Without a cast there is nobody to take wave slots from, and a manager block's phases are loops of one term per
    // wave and round trip (17 reward terms on 4 waves: five round trips one after the other): blocks of 12 waves with
    // 16 envs each (flat Anymal-D, 4096 envs, us/step: 256 threads x 8 envs 49.0, x 16 49.3, x 24 51.1; 384 x 8 46.7;
    // 768 x 12 47.2, x 16 45.6 - 45.8, x 24 46.5, x 32 46.0; post-physics launch 22.1 -> 17.8 us)
    const int post_envs = post_envs_forced > 0 ? post_envs_forced
                          : (!cast && num_envs < 12288) ? 16
                          : (num_envs >= 32768 ? 64 : (num_envs >= 12288 ? 32 : (num_envs >= 6144 ? 24 : 16)));
    int threads = threads_env > 0 ? threads_env : (cast ? 256 : MDP_POST_MAX_THREADS);
    if (threads_env <= 0 && cast && cast->num_rays > 0 && cast->num_rays < 256)
        threads = (((256 / cast->num_rays) * cast->num_rays + 63) / 64) * 64;   // 187 rays -> 192 threads
    // ---- cast
    bool cast_one = false;
    if (cast) {
        MDP_REQUIRE(cast->mesh && cast->root_pose && cast->ray_starts_local && cast->ray_dirs_local && cast->pos_w &&
                        cast->quat_w && cast->ray_hits_w && cast->num_rays > 0, "cast: null pointer");
        MDP_REQUIRE(cast->ld_pose >= 7, "cast: root_pose must be (N,>=7)");
        MDP_REQUIRE(!(cast->tick_dt > 0.0f) || (cast->timestamp && cast->timestamp_last_update && cast->is_outdated),
                    "cast: tick_dt needs the bookkeeping buffers");
        if (!(cast->pattern_vertical_down && cast->attach_yaw_only && cast->mesh->view.accel == 1 && cast->mesh->view.cell_recs))
            return fail(MDP_ERR_UNSUPPORTED, "post-physics launch needs a cast that qualifies for the vertical-ray kernel");
        RaycasterUpdateArgs& ra = a.cast;
        ra.mesh = cast->mesh->view; ra.N = num_envs; ra.R = cast->num_rays; ra.pose = cast->root_pose;
        ra.ld_pose = cast->ld_pose; ra.pose_xyzw = cast->pose_xyzw; ra.drift = cast->drift;
        ra.starts_l = cast->ray_starts_local; ra.dirs_l = cast->ray_dirs_local; ra.max_dist = cast->max_dist;
        ra.pos_w = cast->pos_w; ra.quat_w = cast->quat_w; ra.hits = cast->ray_hits_w; ra.ts = cast->timestamp;
        ra.ts_last = cast->timestamp_last_update; ra.outdated = cast->is_outdated; ra.tick_dt = cast->tick_dt;
        ra.period = cast->update_period;
        // Many envs: the cast runs for several rounds of resident blocks and a block's life is what the launch costs, a
        // third of it one lane's bookkeeping chain (cold loads, atan2f / cosf / sinf) with the ray lanes parked at the
        // barrier.  From MDP_CAST_POSE_MIN_ENVS envs on that part runs as a pass of its own, one thread per env, in front
        // of this launch, and the cast blocks -- every ray lane reads its env's row, no LDS hand-over, no barrier -- start
        // with their rays.  Whole-step bench: 16384 envs 204.0 -> 196.3 us/step (this launch 112.7 -> 98.7 us), 65536
        // envs 742.8 -> 695.4 (454 -> 389), 8192 envs (H1) 101.0 -> 100.0; at 4096 envs the extra launch costs more than
        // the blocks gain (73.7 -> 75.8 us/step), hence the default of 8192.
        static const int pose_min_envs = [] {
            const char* v = std::getenv("MDP_CAST_POSE_MIN_ENVS");
            return v ? std::atoi(v) : 8192;
        }();
        // (read at every call, i.e. at capture time under a hipGraph: tests flip it inside one process)
        const char* stage_v = std::getenv("MDP_RC_STAGE");
        const int stage_env = stage_v ? std::atoi(stage_v) : 0;
        // (the reset-carrying instantiation and meshes with non-finite vertices keep the per-ray walk)
        ra.stage = (stage_env != 0 && !reset && cast->mesh->view.finite) ? 1 : 0;
        ra.pose_ws = cast->pose_ws;
        ra.pose_ready = (cast->pose_ws != nullptr && pose_min_envs >= 0 && num_envs >= pose_min_envs) ? 1 : 0;
        if (cast->scan && cast->scan->obs) {
            const mdp_height_scan_obs_t* sc = cast->scan;
            MDP_REQUIRE(sc->ld_obs >= sc->col_offset + cast->num_rays && sc->col_offset >= 0, "height-scan columns do not fit");
            const bool generated = (sc->post.noise_kind == 1 || sc->post.noise_kind == 2) && !sc->post.noise_u;
            MDP_REQUIRE(!generated || sc->rng, "generated noise needs the rng words");
            MDP_REQUIRE(!sc->normalize.out || (sc->normalize.mean32 && sc->normalize.denom &&
                                                  sc->normalize.ld_out >= sc->col_offset + cast->num_rays),
                        "height-scan normalize: bad tables / ld_out");
            ra.scan = *sc;
            if (!generated) ra.scan.rng = nullptr;
        }
        const int max_epb = ra.stage ? (int)RCU_MAX_EPB_STAGED : (int)RCU_MAX_EPB;
        a.cast_epb = cast->num_rays >= threads ? 1 : std::min(threads / cast->num_rays, max_epb);
        a.cast_slot_w = cast->num_rays >= threads ? threads : cast->num_rays;
        static const int slot_w_env = [] { const char* v = std::getenv("MDP_POST_CAST_SLOT_W"); return v ? std::atoi(v) : 0; }();
        if (slot_w_env > 0 && slot_w_env <= threads) {      // threads per env: an env's rays are looped over by its slot
            a.cast_slot_w = slot_w_env;
            a.cast_epb = std::min(threads / slot_w_env, max_epb);
        }
        static const int64_t cast_cap = [] {
            const char* v = std::getenv("MDP_POST_CAST_BLOCKS");   // > 0: at most this many cast blocks, each strides over its envs
            int64_t c = v ? std::atoll(v) : 0;
            return c > 0 ? c : (int64_t)(1 << 20);
        }();
        a.nb_cast = (unsigned)std::min<int64_t>(((int64_t)num_envs + a.cast_epb - 1) / a.cast_epb, cast_cap);
        // one env per block, one ray per lane, a block per env: the loop-free cast routine (MDP_POST_CAST_ONE=0: the general one)
        const char* one_v = std::getenv("MDP_POST_CAST_ONE");   // read at every call, like MDP_RC_STAGE
        const bool one_env = !one_v || std::atoi(one_v) != 0;
        cast_one = one_env && a.cast_epb == 1 && cast->num_rays <= threads && a.nb_cast == (unsigned)num_envs;
    }
    // ---- scene update
    if (update) {
        SceneUpdatePlan plan;
        int rc = scene_update_plan(update, &plan);
        if (rc != MDP_OK) return rc;
        MDP_REQUIRE(update->num_envs == num_envs, "scene update disagrees on num_envs");
        a.upd = plan.d;
        a.has_upd = 1;
        a.physics_dt = physics_dt;
    }
    // ---- terminations / rewards
    a.tterms = tp->d_terms; a.Tt = tp->num_terms; a.tpool = tp->d_pool;
    a.rterms = rp->d_terms; a.Tr = rp->num_terms; a.rpool = rp->d_pool;
    a.step_dt = tr->dt; a.len_inc = tr->episode_length_buf_inc;
    a.term_dones = tr->term_dones; a.truncated = tr->truncated; a.terminated = tr->terminated; a.dones = tr->dones;
    a.reward = tr->reward_buf; a.sums = tr->episode_sums; a.step = tr->step_reward;
    // ---- command
    if (command) {
        MDP_REQUIRE(command->root_state && command->vel_command_b && command->heading_target && command->is_heading_env &&
                        command->is_standing_env && command->time_left && command->command_counter &&
                        command->metric_error_vel_xy && command->metric_error_vel_yaw, "command: null pointer");
        MDP_REQUIRE(command->rng_state || command->injected_u, "command: need rng_state or injected_u");
        MDP_REQUIRE(command->ld_state >= 13, "command: root_state must be (N,13)");
        a.cmd = *command;
        a.has_cmd = 1;
    }
    // ---- observations
    size_t lds = (size_t)((a.Tt > a.Tr ? a.Tt : a.Tr) > 0 ? (a.Tt > a.Tr ? a.Tt : a.Tr) : 1) * 64 * sizeof(float);
    lds += (size_t)(a.Tt + a.Tr) * sizeof(mdp_term_t) + 16;
    if (observation && observation->program) {
        mdp_program_t* op = observation->program;
        MDP_REQUIRE(op->kind == MDP_PROG_OBS && op->num_envs == num_envs, "not an observation program of this size");
        MDP_REQUIRE(observation->obs && observation->ld_obs >= op->total_cols, "obs buffer too narrow");
        if (op->active_cols > 0) {
            a.oterms = op->d_terms; a.To = op->num_terms; a.opool = op->d_pool; a.col2term = op->d_col2term;
            a.olists = op->d_col2term + op->total_cols; a.n_opcols = op->n_opcols; a.n_vterms = op->n_vterms;
            a.O = op->active_cols; a.obs = observation->obs; a.ld_obs = observation->ld_obs; a.orng = op->d_rng; a.orng_base = op->rng_post_base;
            a.ouses_rng = op->uses_rng;
            if (observation->normalize.out) {
                MDP_REQUIRE(observation->normalize.mean32 && observation->normalize.denom &&
                                observation->normalize.ld_out >= op->total_cols, "observation normalize: bad tables / ld_out");
                a.onorm = observation->normalize;
            }
            lds += (size_t)op->num_terms * sizeof(mdp_term_t) +
                   (size_t)(a.O + a.n_opcols + a.n_vterms) * sizeof(int16_t) + 16;
        }
    }
    if (reset && reset->num_fills > 0) lds += (size_t)reset->num_fills * sizeof(mdp_fill_t) + 8;
    if (lds > 60 * 1024) return fail(MDP_ERR_UNSUPPORTED, "programs too large for the LDS staging area (%zu B)", lds);
    // the cast blocks' record table lives in the same dynamic segment (a block has one role)
    if (cast && a.cast.stage && lds < rc_stage_lds_bytes()) lds = rc_stage_lds_bytes();
    // waves of a manager block that write the observations next to the reward phase (0: every wave does every phase)
    const char* obs_waves_v = std::getenv("MDP_POST_OBS_WAVES");      // read at every call, like MDP_RC_STAGE
    const int obs_waves_env = obs_waves_v ? std::atoi(obs_waves_v) : -1;
    // default: half the waves of the wide blocks of a launch without cast blocks (flat Anymal-D, 4096 envs, 12 waves:
    // 46.0 us/step without roles, 44.9 / 43.7 / 44.0 with 4 / 6 / 8 observation waves); three-wave blocks keep one role
    // -- and so do launches with more manager blocks than CUs: the roles' instantiation holds one wide block per CU (flat
    // Anymal-D at 16,384 envs, 512 blocks: 78.3 us/step without roles, 91.3 with)
    const int64_t nb_manager = ((int64_t)num_envs + post_envs - 1) / post_envs;
    a.obs_waves = obs_waves_env >= 0 ? obs_waves_env : ((!cast && threads >= 512 && nb_manager <= 256) ? threads / 128 : 0);
    if (a.obs_waves >= threads / 64 || reset || (cast && a.cast.stage)) a.obs_waves = 0;
    a.post_envs = post_envs;
    a.nb_post = (unsigned)(((int64_t)num_envs + post_envs - 1) / post_envs);
    if (cast && a.cast.pose_ready) MDP_REQUIRE(((uintptr_t)a.cast.pose_ws & 15) == 0, "cast: pose_ws must be 16-byte aligned");
    // every argument check is behind us: the pass in front of the kernel may write the sensor's buffers now
    const bool staged = cast && a.cast.stage;
    const dim3 grid(a.nb_post + a.nb_cast);
    auto pose_pass = [&]() {
        if (cast && a.cast.pose_ready)
            hipLaunchKernelGGL(k_cast_pose, dim3(grid_for(num_envs, 256, 1 << 20)), dim3(256), 0, as_stream(stream), a.cast);
    };
    if (reset) {
        MDP_REQUIRE(reset->log && reset->log_sums && (reset->fills || reset->num_fills == 0) && reset->num_fills >= 0,
                    "reset: null pointer");
        MDP_REQUIRE(reset->max_episode_length_s > 0.0f, "reset: max_episode_length_s must be positive");
        MDP_REQUIRE(tr->dones && tr->term_dones && tr->episode_sums, "reset needs dones, term_dones and episode_sums");
        if (events) {
            MDP_REQUIRE(command, "events: the launch needs the command term (the interval term runs behind its update)");
            if (events->curriculum) {
                const mdp_curriculum_terrain_t* c = events->curriculum;
                MDP_REQUIRE(c->root_state_w && c->command && c->terrain_levels && c->terrain_types && c->terrain_origins &&
                                c->env_origins, "events: curriculum: null pointer");
                MDP_REQUIRE(c->ld_state >= 3 && c->max_terrain_level >= 1 && c->num_cols >= 1, "events: curriculum: bad shape");
                MDP_REQUIRE(c->injected_levels || c->rng_state, "events: curriculum: need rng_state or injected_levels");
                a.cur = *c; a.has_cur = 1;
            }
            if (events->reset_root_state) {
                const mdp_event_root_state_t* r = events->reset_root_state;
                MDP_REQUIRE(r->default_root_state && r->env_origins && r->root_state_w && r->sim_root_transforms &&
                                r->sim_root_velocities, "events: root state: null pointer");
                MDP_REQUIRE(r->injected_u || r->rng_state, "events: root state: need rng_state or injected_u");
                MDP_REQUIRE(!r->body_acc_w || r->num_bodies > 0, "events: root state: body_acc_w without num_bodies");
                a.ev_root = *r; a.has_ev_root = 1;
            }
            if (events->reset_joints) {
                const mdp_event_joints_t* j = events->reset_joints;
                MDP_REQUIRE(j->mode == 0 || j->mode == 1, "events: joints: mode must be 0 (by scale) or 1 (by offset)");
                MDP_REQUIRE(j->num_joints > 0 && j->num_joints <= 128, "events: joints: num_joints out of range");
                MDP_REQUIRE(j->default_joint_pos && j->default_joint_vel && j->soft_joint_pos_limits && j->soft_joint_vel_limits &&
                                j->joint_pos && j->joint_vel && j->prev_joint_vel && j->joint_acc && j->sim_dof_positions &&
                                j->sim_dof_velocities, "events: joints: null pointer");
                MDP_REQUIRE(j->ld_sim >= j->num_joints, "events: joints: ld_sim < num_joints");
                MDP_REQUIRE(j->injected_u || j->rng_state, "events: joints: need rng_state or injected_u");
                a.ev_joints = *j; a.has_ev_joints = 1;
            }
            if (events->reset_mark) {
                const mdp_event_reset_mark_t* m = events->reset_mark;
                MDP_REQUIRE(m->num_terms >= 0 && m->num_terms <= MDP_EVENT_MARK_MAX_TERMS, "events: mark: at most %d terms",
                            MDP_EVENT_MARK_MAX_TERMS);
                for (int t = 0; t < m->num_terms; ++t)
                    MDP_REQUIRE(m->last_triggered_step[t] && m->triggered_once[t], "events: mark: term %d: null buffer", t);
                a.ev_mark = *m; a.has_ev_mark = m->num_terms > 0 ? 1 : 0;
            }
            if (events->push) {
                const mdp_event_push_t* p = events->push;
                MDP_REQUIRE(p->root_state_w && p->sim_root_velocities && events->push_time_left, "events: push: null pointer");
                MDP_REQUIRE(p->injected_u || p->rng_state, "events: push: need rng_state or injected_u");
                MDP_REQUIRE(events->push_tick_u || events->push_tick_rng, "events: push: the timers need push_tick_rng or push_tick_u");
                MDP_REQUIRE(!p->body_acc_w || p->num_bodies > 0, "events: push: body_acc_w without num_bodies");
                MDP_REQUIRE(events->interval_dt > 0.0f, "events: interval_dt must be positive");
                a.ev_push = *p; a.has_ev_push = 1;
                a.push_time_left = events->push_time_left; a.push_lo = events->push_lo; a.push_scale = events->push_scale;
                a.interval_dt = events->interval_dt; a.push_tick_u = events->push_tick_u;
                a.push_tick_rng = events->push_tick_rng; a.push_trigger = events->push_trigger;
            }
        }
        MDP_REQUIRE(1 + a.Tr + 2 + a.Tt + a.has_cur <= std::min(threads, 256), "reset: more statistics rows than a block reduces");
        MDP_REQUIRE(reset->num_fills <= MDP_MAX_FILLS, "reset: more than MDP_MAX_FILLS buffers");
        MDP_REQUIRE(post_envs >= 8, "reset: the log_sums workspace is sized for blocks of >= 8 envs (MDP_POST_ENVS=%d)", post_envs);
        a.fills = reset->fills; a.num_fills = reset->num_fills; a.log = reset->log;
        a.log_sums = reset->log_sums; a.max_episode_length_s = reset->max_episode_length_s;
        // (noalloc entry point) the shared scratch array keeps the last call's contents: the block ticket starts at zero.
        // Enqueued here, behind every argument check, so a rejected call leaves nothing in the stream / a capture
        if (clear_ticket) MDP_CHECK_HIP(hipMemsetAsync(a.log_sums, 0, sizeof(double), as_stream(stream)));
        pose_pass();
        // (one-env cast blocks in the reset instantiation too: its launch is bound by the blocks with done envs, so they
        // are worth little there -- 109.8 -> 109.3 us/step with events, exp56; MDP_POST_RESET_ONE=0: the general routine.
        // 6 waves per SIMD for this instantiation stay slower, with either routine: 111.4 / 109.9)
        {
            const char* ro = std::getenv("MDP_POST_RESET_ONE");     // A/B knob, read at every call
            if (cast_one && !(ro && std::atoi(ro) == 0))
                hipLaunchKernelGGL((k_post_physics<1, true, false, false, true>), grid, dim3(threads), lds, as_stream(stream), a);
            else
                hipLaunchKernelGGL((k_post_physics<1, true, false, false>), grid, dim3(threads), lds, as_stream(stream), a);
        }
    } else {
        pose_pass();
        if (staged) hipLaunchKernelGGL((k_post_physics<1, false, true, false>), grid, dim3(threads), lds, as_stream(stream), a);
        else if (a.obs_waves > 0) hipLaunchKernelGGL((k_post_physics<1, false, false, true>), grid, dim3(threads), lds, as_stream(stream), a);
        else if (cast_one && num_envs >= post_dense_envs()) hipLaunchKernelGGL((k_post_physics<1, false, false, false, true, true>), grid, dim3(threads), lds, as_stream(stream), a);
        else if (cast_one) hipLaunchKernelGGL((k_post_physics<1, false, false, false, true>), grid, dim3(threads), lds, as_stream(stream), a);
        else hipLaunchKernelGGL((k_post_physics<1, false, false, false>), grid, dim3(threads), lds, as_stream(stream), a);
    }
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_post_physics_step(int num_envs, const mdp_scene_update_t* update, float physics_dt,
                                     const mdp_termination_reward_args_t* tr, const mdp_velocity_command_args_t* command,
                                     const mdp_obs_args_t* observation, const mdp_raycaster_update_args_t* cast,
                                     void* stream) {
    return post_physics_launch(num_envs, update, physics_dt, tr, command, observation, cast, nullptr, stream);
}

extern "C" int mdp_post_physics_step_with_reset_events(int num_envs, const mdp_scene_update_t* update, float physics_dt,
                                                       const mdp_termination_reward_args_t* tr,
                                                       const mdp_velocity_command_args_t* command,
                                                       const mdp_obs_args_t* observation,
                                                       const mdp_raycaster_update_args_t* cast,
                                                       const mdp_post_reset_args_t* reset, const mdp_post_events_t* events,
                                                       void* stream) {
    MDP_REQUIRE(reset, "reset arguments required");
    MDP_REQUIRE(events, "event arguments required (mdp_post_physics_step_with_reset is the launch without them)");
    return post_physics_launch(num_envs, update, physics_dt, tr, command, observation, cast, reset, stream, false, events);
}

extern "C" int mdp_post_physics_step_with_reset(int num_envs, const mdp_scene_update_t* update, float physics_dt,
                                                const mdp_termination_reward_args_t* tr,
                                                const mdp_velocity_command_args_t* command,
                                                const mdp_obs_args_t* observation,
                                                const mdp_raycaster_update_args_t* cast,
                                                const mdp_post_reset_args_t* reset, void* stream) {
    MDP_REQUIRE(reset, "null reset arguments");
    return post_physics_launch(num_envs, update, physics_dt, tr, command, observation, cast, reset, stream);
}

// ---- the same entry point for callers that cannot allocate (HIP graph capture, serving loops): the block-statistics
// workspace (`log_sums`) is a file-scope device array instead of a caller allocation.  Each GPU holds its own copy of the
// array at its own address: the address is looked up once per device, with that device current, and host code only
// ever uses that pointer (never the array's name, which is a host-side stand-in).
static __device__ __align__(256) double scratch[MDP_POST_NOALLOC_SCRATCH_DOUBLES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
double* g_scratch_ptr[kMaxDevices] = {};

// `program_device`: the device the caller's programs (and so its buffers and stream) live on.  Each GPU holds its own
// copy of the array at its own address, and hipGetSymbolAddress answers for the CURRENT device: the two must agree,
// else the kernel would be handed another GPU's address.
int post_noalloc_scratch(int program_device, double** out) {
    int dev = -1;
    MDP_CHECK_HIP(hipGetDevice(&dev));
    MDP_REQUIRE(dev >= 0 && dev < kMaxDevices, "device index %d out of range", dev);
    MDP_REQUIRE(dev == program_device,
                "noalloc: the current device (%d) is not the device the programs were created on (%d); call hipSetDevice first",
                dev, program_device);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        double* p = nullptr;
        MDP_CHECK_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
        MDP_REQUIRE(p != nullptr, "hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return MDP_OK;
}
}  // namespace

extern "C" int mdp_post_physics_step_with_reset_noalloc(int num_envs, const mdp_scene_update_t* update, float physics_dt,
                                                        const mdp_termination_reward_args_t* tr,
                                                        const mdp_velocity_command_args_t* command,
                                                        const mdp_obs_args_t* observation,
                                                        const mdp_raycaster_update_args_t* cast,
                                                        const mdp_post_reset_args_t* reset, void* stream) {
    MDP_REQUIRE(reset, "null reset arguments");
    MDP_REQUIRE(tr && tr->termination && tr->reward, "null termination / reward programs");
    MDP_REQUIRE(reset->log_sums == nullptr, "noalloc: log_sums must be NULL (the library's own scratch array is used)");
    // workspace the launch will use: the ticket word + one slot of `rows` doubles per manager block of >= 8 envs
    const int64_t rows = 1 + (int64_t)tr->reward->num_terms + 2 + (int64_t)tr->termination->num_terms;
    const int64_t need = 1 + rows * (((int64_t)num_envs + 7) / 8);
    MDP_REQUIRE(num_envs >= 0 && need <= (int64_t)MDP_POST_NOALLOC_SCRATCH_DOUBLES,
                "noalloc: %lld envs x %lld statistics rows need %lld doubles of scratch, the static array holds %d"
                " (use mdp_post_physics_step_with_reset with a workspace of your own)",
                (long long)num_envs, (long long)rows, (long long)need, (int)MDP_POST_NOALLOC_SCRATCH_DOUBLES);
    MDP_REQUIRE(tr->termination->device == tr->reward->device, "noalloc: programs of different devices");
    double* ws = nullptr;
    int rc = post_noalloc_scratch(tr->termination->device, &ws);
    if (rc != MDP_OK) return rc;
    // The array keeps whatever the last call on this GPU left.  Only word 0 (the block ticket) must start at zero -- the
    // launch routine clears it once it has accepted the arguments; the slots are fully rewritten by the blocks of this
    // launch before its last block reads them.
    mdp_post_reset_args_t r = *reset;
    r.log_sums = ws;
    return post_physics_launch(num_envs, update, physics_dt, tr, command, observation, cast, &r, stream, true);
}

extern "C" int mdp_obs_postprocess(const mdp_term_t* term, int num_envs, float* data, int ld, uint64_t seed,
                                   uint64_t offset, void* stream) {
    MDP_REQUIRE(term && data, "null pointer");
    MDP_REQUIRE(term->dim > 0 && ld >= term->dim, "bad dim/ld");
    if (num_envs == 0) return MDP_OK;
    int64_t total = (int64_t)num_envs * term->dim;
    hipLaunchKernelGGL(k_obs_post, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), *term, num_envs, data,
                       ld, seed, offset);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_term_eval(int kind, const mdp_term_t* term, const int32_t* index_pool, int num_envs, void* out,
                             int ld_out, uint64_t seed, uint64_t epoch, void* stream) {
    MDP_REQUIRE(term && out, "null pointer");
    MDP_REQUIRE(kind >= MDP_PROG_OBS && kind <= MDP_PROG_TERMINATION, "unknown program kind %d", kind);
    MDP_REQUIRE(term->n_idx <= 0 || index_pool, "index list without a pool");
    if (num_envs == 0) return MDP_OK;
    hipStream_t s = as_stream(stream);
    if (kind == MDP_PROG_OBS) {
        MDP_REQUIRE(term->op >= 0 && term->op < MDP_OBS_NUM_OPS && term->op != MDP_OBS_HEIGHT_SCAN_CAST,
                    "opcode %d cannot be evaluated stand-alone", term->op);
        MDP_REQUIRE(term->dim > 0 && ld_out >= term->dim, "bad dim/ld");
        int64_t total = (int64_t)num_envs * term->dim;
        hipLaunchKernelGGL(k_obs_single, dim3(grid_for(total, 256)), dim3(256), 0, s, *term, index_pool, num_envs,
                           (float*)out, ld_out, seed, epoch);
    } else if (kind == MDP_PROG_REWARD) {
        MDP_REQUIRE(term->op >= 0 && term->op < MDP_REW_NUM_OPS, "unknown reward opcode %d", term->op);
        hipLaunchKernelGGL(k_reward_single, dim3(grid_for(num_envs, 64, 1 << 20)), dim3(64), 0, s, *term, index_pool,
                           num_envs, (float*)out);
    } else {
        MDP_REQUIRE(term->op >= 0 && term->op < MDP_TERM_NUM_OPS, "unknown termination opcode %d", term->op);
        hipLaunchKernelGGL(k_term_single, dim3(grid_for(num_envs, 64, 1 << 20)), dim3(64), 0, s, *term, index_pool,
                           num_envs, (uint8_t*)out);
    }
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}
