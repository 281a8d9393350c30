// Height-scan ray caster update as a block-level device routine (ray_caster.py:220-260 + sensor_base.py:197-205,
// 284-293 + observations.py:165-173): shared by mdp_raycaster_update's vertical-ray kernel and by the post-physics
// kernel (programs.hip), where cast blocks run next to the blocks that evaluate the managers.
#pragma once
#include "mdp_device.hpp"
#include "raycast_device.hpp"
#include "obs_post_device.hpp"

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
};

// height_scan observation column of ray r of env e, written by the cast kernels (mdp_height_scan_obs_t)
__device__ __forceinline__ void scan_store(const mdp_height_scan_obs_t& sc, int e, int r, int R, float sensor_z, float hit_z,
                                           uint64_t seed, uint64_t epoch) {
    float v = (sensor_z - hit_z) - sc.offset;   // observations.py:173
    v = obs_post(sc.post, v, e, r, seed, epoch, 0x48534341u, (uint64_t)e * (uint64_t)R + (uint64_t)r);
    sc.obs[(int64_t)e * sc.ld_obs + sc.col_offset + r] = v;
}

// yaw-only attachment and a pattern whose directions are all (0, 0, -c), on a mesh with cell records.  A block owns
// epb whole environments (slot_w threads each, blockDim.x >= epb * slot_w) and strides over the envs by nblocks.
template <int BATCH>
__device__ __forceinline__ void raycaster_update_down_blocks(const RaycasterUpdateArgs& a, int epb, int slot_w,
                                                             unsigned block, unsigned nblocks) {
    const mdp_mesh_view_t& mesh = a.mesh;
    const int N = a.N, R = a.R, ld_pose = a.ld_pose, pose_xyzw = a.pose_xyzw;
    const float* __restrict__ pose = a.pose;
    const float* __restrict__ drift = a.drift;
    const float* __restrict__ starts_l = a.starts_l;
    const float* __restrict__ dirs_l = a.dirs_l;
    const float max_dist = a.max_dist, tick_dt = a.tick_dt, period = a.period;
    float* __restrict__ pos_w = a.pos_w;
    float* __restrict__ quat_w = a.quat_w;
    float* __restrict__ hits = a.hits;
    float* __restrict__ ts = a.ts;
    float* __restrict__ ts_last = a.ts_last;
    uint8_t* __restrict__ outdated = a.outdated;
    const mdp_height_scan_obs_t& scan = a.scan;
    const float inf = __builtin_inff();
    int le = threadIdx.x / slot_w;
    int lr = threadIdx.x - le * slot_w;
    uint64_t seed = 0, epoch = 0;
    if (scan.obs && scan.rng) { seed = scan.rng[0]; epoch = scan.rng[1 + block]; }
    for (int base = block * epb; base < N; base += nblocks * epb) {
        int e = base + le;
        bool active = le < epb && e < N;
        if (tick_dt > 0.0f) {
            // SensorBase.update of this physics step (sensor_base.py:197-205) done here, so the cast does not have
            // to wait for the launch that updates the rest of the scene
            if (active && lr == 0) {
                float t = ts[e] + tick_dt;
                ts[e] = t;
                if ((t - ts_last[e]) + 1e-6f >= period) outdated[e] = 1;
            }
            __syncthreads();
        }
        bool od = active && (outdated ? outdated[e] != 0 : true);
        __syncthreads();
        if (active && !od && scan.obs) {
            float sz = pos_w[3 * (int64_t)e + 2];
            for (int r = lr; r < R; r += slot_w) scan_store(scan, e, r, R, sz, hits[((int64_t)e * R + r) * 3 + 2], seed, epoch);
        }
        if (od) {
            const float* p = pose + (int64_t)e * ld_pose;
            V3 pos = ld3(p);
            if (drift) { pos.x += drift[3 * e]; pos.y += drift[3 * e + 1]; pos.z += drift[3 * e + 2]; }
            // physics transforms are (x, y, z, w) (ray_caster.py:226-232 converts them); root_state rows are (w, x, y, z)
            Q4 q = pose_xyzw ? Q4{p[6], p[3], p[4], p[5]} : ld4(p + 3);
            Q4 qr = yaw_quat(q);
            for (int r = lr; r < R; r += slot_w) {
                V3 s = quat_apply(qr, ld3(starts_l + 3 * r));
                s = V3{s.x + pos.x, s.y + pos.y, s.z + pos.z};
                V3 d = ld3(dirs_l + 3 * r);
                mdp_vray_t ray = mdp_rc_make_vray(s.x, s.y, s.z, d.x, d.y, d.z);
                mdp_hit_t h = mdp_rc_grid_down<BATCH>(&mesh, &ray, max_dist);
                float* o = hits + ((int64_t)e * R + r) * 3;
                float hz = inf;
                if (h.face >= 0) {
                    hz = s.z + h.t * d.z;
                    o[0] = s.x + h.t * d.x; o[1] = s.y + h.t * d.y; o[2] = hz;
                } else {
                    o[0] = inf; o[1] = inf; o[2] = inf;
                }
                if (scan.obs) scan_store(scan, e, r, R, pos.z, hz, seed, epoch);
            }
            if (lr == 0) {
                st3(pos_w + 3 * (int64_t)e, pos);
                st4(quat_w + 4 * (int64_t)e, q);
                if (ts_last) ts_last[e] = ts[e];
                if (outdated) outdated[e] = 0;
            }
        }
    }
    if (scan.obs && scan.rng && threadIdx.x == 0) scan.rng[1 + block] = epoch + 1;
}

}  // namespace mdp
