// Height-scan ray caster update as a block-level device routine (ray_caster.py:220-260 + sensor_base.py:197-205,
// 284-293 + observations.py:165-173): shared by mdp_raycaster_update's vertical-ray kernel and by the post-physics
// kernel (programs.hip), where cast blocks run next to the blocks that evaluate the managers.
#pragma once
#include "mdp_device.hpp"
#include "raycast_device.hpp"
#include "obs_post_device.hpp"
#include "stamps_device.hpp"

namespace mdp {

struct RaycasterUpdateArgs {
    mdp_mesh_view_t mesh;
    int N, R;
    const float* pose;
    int ld_pose, pose_xyzw;
    const float* drift;
    const float* starts_l;
    const float* dirs_l;
    float max_dist;
    float* pos_w;
    float* quat_w;
    float* hits;
    float* ts;
    float* ts_last;
    uint8_t* outdated;
    mdp_height_scan_obs_t scan;
    float tick_dt, period;
    const uint8_t* reset_mask;   // FLAGGED instantiation only (mdp_raycaster_update_flagged)
    float* pose_ws;              // (N, 8) {pos.xyz, yaw quaternion (w, z), cast flag, -, -}: filled by raycaster_pose_pass
    int pose_ready;              // 1: pose_ws holds this step's values (the pass ran in front of this launch)
};

// height_scan observation column of ray r of env e, written by the cast kernels (mdp_height_scan_obs_t)
// Generator stream of the height-scan noise: the per-step cast and the flagged recast after a reset read the same
// per-block epoch words with different block -> env maps, so they draw from DIFFERENT streams -- with one stream the
// recast of env e at epoch k would repeat, sample for sample, what the per-step cast of env e draws a few steps later.
constexpr uint32_t SCAN_STREAM = 0x48534341u, SCAN_STREAM_FLAGGED = 0x48534346u;

__device__ __forceinline__ void scan_store(const mdp_height_scan_obs_t& sc, int e, int r, int R, float sensor_z, float hit_z,
                                           uint64_t seed, uint64_t epoch, uint32_t stream = SCAN_STREAM) {
    float v = (sensor_z - hit_z) - sc.offset;   // observations.py:173
    v = obs_post(sc.post, v, e, r, seed, epoch, stream, (uint64_t)e * (uint64_t)R + (uint64_t)r);
    sc.obs[(int64_t)e * sc.ld_obs + sc.col_offset + r] = v;
    if (sc.normalize.out) {      // the standardised copy for the observation's consumer (mdp_obs_normalize_t)
        const int c = sc.col_offset + r;
        sc.normalize.out[(int64_t)e * sc.normalize.ld_out + c] =
            tclamp((v - sc.normalize.mean32[c]) / sc.normalize.denom[c], -sc.normalize.clip, sc.normalize.clip);
    }
}

// yaw-only attachment and a pattern whose directions are all (0, 0, -c), on a mesh with cell records.  A block owns
// epb whole environments (slot_w threads each, blockDim.x >= epb * slot_w) and strides over the envs by nblocks.
//
// The kernel is bound by vector instructions, so everything that is per ENV runs on one lane per env -- the sensor
// bookkeeping of this physics step, the pose, the yaw quaternion (atan2f / cosf / sinf) -- and reaches the env's ray
// lanes through LDS behind the block's single barrier; the ray lanes only rotate their start, walk one grid cell and
// post-process their height.
constexpr int RCU_MAX_EPB = 256;   // blockDim.x <= 1024 and num_rays >= 1; callers keep epb <= this

// ray r of env e: start rotated by the yaw quaternion, one grid-cell walk, hit point and height_scan column stored
template <int BATCH>
__device__ __forceinline__ void raycaster_down_ray(const RaycasterUpdateArgs& a, int e, int r, V3 pos, Q4 qr, uint64_t seed,
                                                   uint64_t epoch, uint32_t stream = SCAN_STREAM) {
    const mdp_mesh_view_t& mesh = a.mesh;
    const int R = a.R;
    const float inf = __builtin_inff();
    V3 s = quat_apply(qr, ld3(a.starts_l + 3 * r));
    s = V3{s.x + pos.x, s.y + pos.y, s.z + pos.z};
    V3 d = ld3(a.dirs_l + 3 * r);
    mdp_hit_t h;
    if (mesh.finite) {
#ifdef MDP_STAMPS
        int walk = 0;   // diagnostic: records in the cell / batches read by this ray, 0 = lattice patch (tools/kernel_stamps.py)
        if (!(mesh.patch_cells && mdp_rc_patch_down(&mesh, s.x, s.y, s.z, d.z, a.max_dist, &h)))
            h = mdp_rc_grid_down_exact<BATCH>(&mesh, s.x, s.y, s.z, d.z, a.max_dist, &walk);
        if (g_stamps && e < 4096) g_stamps[8192 * 8 + (size_t)e * R + r] = (unsigned long long)walk;
#else
        if (!(mesh.patch_cells && mdp_rc_patch_down(&mesh, s.x, s.y, s.z, d.z, a.max_dist, &h)))
            h = mdp_rc_grid_down_exact<BATCH>(&mesh, s.x, s.y, s.z, d.z, a.max_dist);
#endif
    } else {
        mdp_vray_t ray = mdp_rc_make_vray(s.x, s.y, s.z, d.x, d.y, d.z);
        h = mdp_rc_grid_down<BATCH>(&mesh, &ray, a.max_dist);
    }
    MDP_STAMP(4);
    float* o = a.hits + ((int64_t)e * R + r) * 3;
    float hz = inf;
    if (h.face >= 0) {
        hz = s.z + h.t * d.z;
        o[0] = s.x + h.t * d.x; o[1] = s.y + h.t * d.y; o[2] = hz;
    } else {
        o[0] = inf; o[1] = inf; o[2] = inf;
    }
    if (a.scan.obs) scan_store(a.scan, e, r, R, pos.z, hz, seed, epoch, stream);
}

// Per-env part of the update, one lane per env: SensorBase.update of this physics step (sensor_base.py:197-205), the
// lazy-evaluation test (:284-293), sensor pose with drift and the yaw-only quaternion (ray_caster.py:220-240); writes
// the sensor's own buffers and returns {pos.xyz, qr.w, qr.z, cast flag} in se[0..5].
__device__ __forceinline__ void raycaster_env_bookkeeping(const RaycasterUpdateArgs& a, int e, float* se) {
    // everything whose address is known up front is requested together (flags, timestamps, pose, drift): one
    // round trip instead of flags -> pose; envs that turn out not to be outdated only wasted two lines
    const float* p = a.pose + (int64_t)e * a.ld_pose;
    const uint8_t od_in = a.outdated ? a.outdated[e] : (uint8_t)1;
    const float ts_in = a.ts ? a.ts[e] : 0.0f;
    const float last_in = (a.tick_dt > 0.0f) ? a.ts_last[e] : 0.0f;
    V3 pos = ld3(p);
    // physics transforms are (x, y, z, w) (ray_caster.py:226-232 converts them); root_state rows (w, x, y, z)
    const Q4 q = a.pose_xyzw ? Q4{p[6], p[3], p[4], p[5]} : ld4(p + 3);
    V3 dr{0.0f, 0.0f, 0.0f};
    if (a.drift) dr = ld3(a.drift + 3 * (int64_t)e);
    // SensorBase.update of this physics step (sensor_base.py:197-205) + the lazy-evaluation test (:284-293)
    bool od = od_in != 0;
    float t_now = ts_in;
#ifdef MDP_STAMPS
    if (pos.x == 1.2345e30f && dr.x == 1.0f && ts_in == last_in) return;   // waits for every input
    MDP_STAMP(5);
#endif
    if (a.tick_dt > 0.0f) {
        t_now += a.tick_dt;
        a.ts[e] = t_now;
        if ((t_now - last_in) + 1e-6f >= a.period) od = true;
    }
    se[5] = od ? 1.0f : 0.0f;
    if (od) {
        if (a.drift) { pos.x += dr.x; pos.y += dr.y; pos.z += dr.z; }
        Q4 qr = yaw_quat(q);
#ifdef MDP_STAMPS
        if (qr.w == 1.2345e30f) return;
        MDP_STAMP(6);
#endif
        se[0] = pos.x; se[1] = pos.y; se[2] = pos.z; se[3] = qr.w; se[4] = qr.z;
        st3(a.pos_w + 3 * (int64_t)e, pos);
        st4(a.quat_w + 4 * (int64_t)e, q);
        if (a.ts_last) a.ts_last[e] = t_now;
        if (a.outdated) a.outdated[e] = 0;
    }
}

// The same for every env in a pass of its own (one thread per env) in front of a launch with many envs: the results go
// to pose_ws (N, 8) and the cast blocks of that launch start with their rays (RaycasterUpdateArgs::pose_ready).
__device__ __forceinline__ void raycaster_pose_pass(const RaycasterUpdateArgs& a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.N) return;
    float se[6] = {0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f};
    raycaster_env_bookkeeping(a, e, se);
    float4* w = reinterpret_cast<float4*>(a.pose_ws + 8 * (int64_t)e);
    w[0] = float4{se[0], se[1], se[2], se[3]};
    w[1] = float4{se[4], se[5], 0.0f, 0.0f};
}

template <int BATCH>
__device__ __forceinline__ void raycaster_update_down_blocks(const RaycasterUpdateArgs& a, int epb, int slot_w,
                                                             unsigned block, unsigned nblocks) {
    __shared__ float s_env[RCU_MAX_EPB][6];   // pos.xyz, yaw quaternion (w, z), cast flag
    const int N = a.N, R = a.R;
    float* __restrict__ hits = a.hits;
    const mdp_height_scan_obs_t& scan = a.scan;
    const int le = threadIdx.x / slot_w;
    const int lr = threadIdx.x - le * slot_w;
    uint64_t seed = 0, epoch = 0;
    if (scan.obs && scan.rng) { seed = scan.rng[0]; epoch = scan.rng[1 + block]; }
    for (int base = block * epb; base < N; base += nblocks * epb) {
        const int e = base + le;
        const bool active = le < epb && e < N;
        float sv[6] = {0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f};
        if (a.pose_ready) {
            // the pass in front of this launch did the per-env part: every ray lane reads the env's 32-byte row itself
            // (one request per wave, the line is still on chip) -- no LDS hand-over, no barrier, the waves run free
            if (active) {
                const float4* w = reinterpret_cast<const float4*>(a.pose_ws + 8 * (int64_t)e);
                const float4 w0 = w[0], w1 = w[1];
                sv[0] = w0.x; sv[1] = w0.y; sv[2] = w0.z; sv[3] = w0.w; sv[4] = w1.x; sv[5] = w1.y;
            }
        } else {
            if (active && lr == 0) raycaster_env_bookkeeping(a, e, s_env[le]);
            __syncthreads();
            const float* sl = s_env[active ? le : 0];
#pragma unroll
            for (int k = 0; k < 6; ++k) sv[k] = sl[k];
        }
        MDP_STAMP(2);
        const float* se = sv;
        const bool od = active && se[5] != 0.0f;
        if (active && !od && scan.obs) {
            float sz = a.pos_w[3 * (int64_t)e + 2];
            for (int r = lr; r < R; r += slot_w) scan_store(scan, e, r, R, sz, hits[((int64_t)e * R + r) * 3 + 2], seed, epoch);
        }
        if (od) {
            const V3 pos = V3{se[0], se[1], se[2]};
            const Q4 qr = Q4{se[3], 0.0f, 0.0f, se[4]};
            MDP_STAMP(3);
            for (int r = lr; r < R; r += slot_w) raycaster_down_ray<BATCH>(a, e, r, pos, qr, seed, epoch);
        }
        if (!a.pose_ready) __syncthreads();   // s_env is rewritten by the next env group
    }
    if (scan.obs && scan.rng && threadIdx.x == 0) scan.rng[1 + block] = epoch + 1;
}

// The reset + lazy re-evaluation form (mdp_raycaster_update_flagged): envs of reset_mask get their timestamps zeroed and
// count as outdated, there is no tick, and only outdated envs are cast (the others keep their observation columns).
// Few envs are flagged in a step, so a block looks at RCF_ENVS envs with one lane each -- bookkeeping, pose, yaw
// quaternion -- and then casts the flagged ones one after the other with all its threads: a launch with nothing to do
// is N / RCF_ENVS blocks that read three flags per env and leave (4 envs per block: --auto-reset masked 113.9 -> 112.6 us/step
// against 16; a block with two flagged envs casts them one after the other).
#ifndef MDP_RCF_ENVS
#define MDP_RCF_ENVS 4
#endif
constexpr int RCF_ENVS = MDP_RCF_ENVS;

template <int BATCH>
__device__ __forceinline__ void raycaster_update_flagged_blocks(const RaycasterUpdateArgs& a, unsigned block) {
    __shared__ float s_fenv[RCF_ENVS][6];
    const int N = a.N, R = a.R;
    const mdp_height_scan_obs_t& scan = a.scan;
    uint64_t seed = 0, epoch = 0;
    const int base = (int)block * RCF_ENVS;
    // the launch is a chain of dependent cold loads for the few envs it casts: everything whose address is known up
    // front is requested together with the flags (pose, drift, generator words) instead of behind them
    if (scan.obs && scan.rng) { seed = scan.rng[0]; epoch = scan.rng[1 + block]; }
    if ((int)threadIdx.x < RCF_ENVS) {
        const int le = threadIdx.x, e = base + le;
        bool od = false;
        if (e < N) {
            const float* p = a.pose + (int64_t)e * a.ld_pose;
            V3 pos = ld3(p);
            Q4 q = a.pose_xyzw ? Q4{p[6], p[3], p[4], p[5]} : ld4(p + 3);
            V3 dr{0.0f, 0.0f, 0.0f};
            if (a.drift) dr = ld3(a.drift + 3 * (int64_t)e);
            const bool reset = a.reset_mask && a.reset_mask[e];
            od = a.outdated[e] != 0;
            float t_now = a.ts[e];
            if (reset) {                                  // sensor_base.py:182-195
                t_now = 0.0f;
                a.ts[e] = 0.0f;
                od = true;
            }
            if (od) {
                if (a.drift) { pos.x += dr.x; pos.y += dr.y; pos.z += dr.z; }
                Q4 qr = yaw_quat(q);
                float* se = s_fenv[le];
                se[0] = pos.x; se[1] = pos.y; se[2] = pos.z; se[3] = qr.w; se[4] = qr.z;
                st3(a.pos_w + 3 * (int64_t)e, pos);
                st4(a.quat_w + 4 * (int64_t)e, q);
                a.ts_last[e] = t_now;
                a.outdated[e] = 0;
            }
        }
        s_fenv[le][5] = od ? 1.0f : 0.0f;
    }
    __syncthreads();
    bool any = false;
    for (int le = 0; le < RCF_ENVS; ++le) {
        const float* se = s_fenv[le];
        if (se[5] == 0.0f) continue;                     // block-uniform
        any = true;
        const V3 pos = V3{se[0], se[1], se[2]};
        const Q4 qr = Q4{se[3], 0.0f, 0.0f, se[4]};
        for (int r = threadIdx.x; r < R; r += blockDim.x)
            raycaster_down_ray<BATCH>(a, base + le, r, pos, qr, seed, epoch, SCAN_STREAM_FLAGGED);
    }
    if (any && scan.obs && scan.rng && threadIdx.x == 0) scan.rng[1 + block] = epoch + 1;
}

}  // namespace mdp
