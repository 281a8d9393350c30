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
    int stage;                   // 1: the block stages the cell records its rays name in LDS (RcStage; the launch
                                 // provides rc_stage_lds_bytes() of dynamic LDS), 0: every ray walks its cell on its own
};

// ---- cell records staged in LDS ----------------------------------------------------------------------------------
// The rays of one scan fall on a patch of ~2 m^2: on stair treads, box tops and the border most of them name the SAME
// few cell lists (cells with identical lists share one run of records, raycast.hip), and a ray that walks its list on
// its own pays one dependent memory round trip per record -- with the slowest lane of the wave setting the pace -- for
// bytes its neighbours fetch as well.  Staged form: every ray that needs records enters its list (keyed by the list's
// offset) in a small hash table in LDS, the block then fetches the first RC_STAGE_K records of every DISTINCT list in one
// cooperative, coalesced round trip (16 B per thread and load), and the rays walk out of LDS: same records, same order,
// same arithmetic (mdp_rc_down_exact_record), so hits stay bit-identical.  A ray that finds no free slot, or whose walk
// goes past the staged records, continues from global memory as before.
#ifndef MDP_RC_STAGE_SLOTS
#define MDP_RC_STAGE_SLOTS 64          // power of two
#endif
#ifndef MDP_RC_STAGE_K
#define MDP_RC_STAGE_K 3
#endif
constexpr int RC_STAGE_SLOTS = MDP_RC_STAGE_SLOTS, RC_STAGE_K = MDP_RC_STAGE_K, RC_STAGE_PROBES = 4;
struct RcStage {
    float4 rec[RC_STAGE_SLOTS][RC_STAGE_K][4];   // 64-byte records, list order
    int key[RC_STAGE_SLOTS];                     // list offset into cell_recs_down (records), -1 = free
    int cnt[RC_STAGE_SLOTS];                     // records staged for the slot: min(list length, RC_STAGE_K)
};
constexpr size_t rc_stage_lds_bytes() { return sizeof(RcStage); }

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
constexpr int RCU_MAX_EPB_STAGED = 32;   // env slots per block of the staged routine

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

// The same ray for a launch that casts a handful of envs (the flagged recast after a reset): the dependent chain is what
// such a launch costs, not its instruction count, so everything is requested as early as its address is known -- the
// cell's entry together with the coarse patch entry, then the patch's four vertices together with the cell's first two
// records -- and the walk of mdp_rc_grid_down_exact runs over what has arrived (further records, rare, one round trip
// each): sensor pose -> two round trips instead of up to six.  Same records in the same order through the same
// routines: the hit is bit-identical to raycaster_down_ray's.  Meshes with non-finite vertices take that routine.
__device__ __forceinline__ void raycaster_down_ray_eager(const RaycasterUpdateArgs& a, int e, int r, V3 pos, Q4 qr,
                                                         uint64_t seed, uint64_t epoch, uint32_t stream) {
    const mdp_mesh_view_t& m = a.mesh;
    const int R = a.R;
    const float inf = __builtin_inff();
    V3 s = quat_apply(qr, ld3(a.starts_l + 3 * r));
    s = V3{s.x + pos.x, s.y + pos.y, s.z + pos.z};
    const V3 d = ld3(a.dirs_l + 3 * r);
    // ---- round trip 1: cell entry + patch entry
    int32_t c = 0;
    const bool cell_ok = mdp_rc_down_cell(&m, s.x, s.y, &c);
    int2 cref{0, 0};
    if (cell_ok) cref = reinterpret_cast<const int2*>(m.cell_start_down)[c];
    int pcell = 0;
    const bool pc_ok = m.patch_cells && mdp_rc_patch_cell(&m, s.x, s.y, &pcell);
    int4 c0{0, 0, 0, -1}, c1{0, 0, 0, 0};
    if (pc_ok) {
        const int4* pc = reinterpret_cast<const int4*>(m.patch_cells) + 2 * pcell;
        c0 = pc[0]; c1 = pc[1];
    }
    // ---- round trip 2: lattice vertices + the first two records
    int vidx = 0, ncols = 0;
    const bool sq = pc_ok && mdp_rc_patch_square(c0, c1, s.x, s.y, &vidx, &ncols);
    float4 v00{0, 0, 0, 0}, v01 = v00, v10 = v00, v11 = v00;
    if (sq) {
        const float4* V = reinterpret_cast<const float4*>(m.patch_verts) + vidx;
        v00 = V[0]; v01 = V[1]; v10 = V[ncols]; v11 = V[ncols + 1];
    }
    const bool disjoint = cref.x < 0;
    const int32_t b = MDP_RC_CELL_OFFSET(cref.x), n = cell_ok ? cref.y : 0;
    mdp_rec_t R0{}, R1{};
    if (n > 0) R0 = mdp_rc_load_rec(m.cell_recs_down + MDP_RC_REC_STRIDE * (int64_t)b);
    if (n > 1) R1 = mdp_rc_load_rec(m.cell_recs_down + MDP_RC_REC_STRIDE * (int64_t)(b + 1));
    // ---- the patch's answer, else the walk
    mdp_hit_t h; h.t = a.max_dist; h.face = -1;
    const bool patched = sq && mdp_rc_patch_eval(v00, v01, v10, v11, s.x, s.y, s.z, d.z, a.max_dist, &h) != 0;
    if (!patched) {
        h.t = a.max_dist; h.face = -1;
        const float Sz = 1.0f / d.z;
        bool stop = false;
        for (int32_t k = 0; k < n && !stop; ++k) {
            mdp_rec_t rec = k == 0 ? R0 : R1;
            if (k > 1) rec = mdp_rc_load_rec(m.cell_recs_down + MDP_RC_REC_STRIDE * (int64_t)(b + k));
            stop = mdp_rc_down_exact_record(rec.v, s.x, s.y, s.z, Sz, disjoint, &h);
        }
    }
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
    if (!a.stage) MDP_STAMP(5);     // (the staged walk uses stamps 5 and 6 for its two barriers)
#endif
    if (a.tick_dt > 0.0f) {
        t_now += a.tick_dt;
        a.ts[e] = t_now;
        if ((t_now - last_in) + 1e-6f >= a.period) od = true;
    }
    se[5] = od ? 1.0f : 0.0f;
    if (od) {
        if (a.drift) { pos.x += dr.x; pos.y += dr.y; pos.z += dr.z; }
        Q4 qr = yaw_quat_halfangle(q);     // (mdp_device.hpp: the libm-free form, this lane is the block's critical path)
#ifdef MDP_STAMPS
        if (qr.w == 1.2345e30f) return;
        if (!a.stage) MDP_STAMP(6);
#endif
        se[0] = pos.x; se[1] = pos.y; se[2] = pos.z; se[3] = qr.w; se[4] = qr.z;
        st3(a.pos_w + 3 * (int64_t)e, pos);
        st4(a.quat_w + 4 * (int64_t)e, q);
        if (a.ts_last) a.ts_last[e] = t_now;
        if (a.outdated) a.outdated[e] = 0;
    }
}

// raycaster_env_bookkeeping with its global stores held back (RcBook) until the caller commits them: in a block whose
// ray lanes wait at a barrier for this lane, stores in front of the barrier make the barrier wait for their
// acknowledgements as well.
struct RcBook { V3 pos; Q4 q; float t_now; bool ticked, cast; };
__device__ __forceinline__ void raycaster_env_bookkeeping_deferred(const RaycasterUpdateArgs& a, int e, float* se, RcBook& bk) {
    const float* p = a.pose + (int64_t)e * a.ld_pose;
    const uint8_t od_in = a.outdated ? a.outdated[e] : (uint8_t)1;
    const float ts_in = a.ts ? a.ts[e] : 0.0f;
    const float last_in = (a.tick_dt > 0.0f) ? a.ts_last[e] : 0.0f;
    V3 pos = ld3(p);
    const Q4 q = a.pose_xyzw ? Q4{p[6], p[3], p[4], p[5]} : ld4(p + 3);
    V3 dr{0.0f, 0.0f, 0.0f};
    if (a.drift) dr = ld3(a.drift + 3 * (int64_t)e);
    bool od = od_in != 0;
    float t_now = ts_in;
#ifdef MDP_STAMPS
    if (pos.x == 1.2345e30f && dr.x == 1.0f && ts_in == last_in) return;   // waits for every input
    MDP_STAMP(5);
#endif
    bk.ticked = a.tick_dt > 0.0f;
    if (bk.ticked) {
        t_now += a.tick_dt;
        if ((t_now - last_in) + 1e-6f >= a.period) od = true;
    }
    se[5] = od ? 1.0f : 0.0f;
    if (od) {
        if (a.drift) { pos.x += dr.x; pos.y += dr.y; pos.z += dr.z; }
        Q4 qr = yaw_quat_halfangle(q);
#ifdef MDP_STAMPS
        if (qr.w == 1.2345e30f) return;
        MDP_STAMP(6);
#endif
        se[0] = pos.x; se[1] = pos.y; se[2] = pos.z; se[3] = qr.w; se[4] = qr.z;
    }
    bk.pos = pos; bk.q = q; bk.t_now = t_now; bk.cast = od;
}
__device__ __forceinline__ void raycaster_env_bookkeeping_commit(const RaycasterUpdateArgs& a, int e, const RcBook& bk) {
    if (bk.ticked) a.ts[e] = bk.t_now;
    if (bk.cast) {
        st3(a.pos_w + 3 * (int64_t)e, bk.pos);
        st4(a.quat_w + 4 * (int64_t)e, bk.q);
        if (a.ts_last) a.ts_last[e] = bk.t_now;
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

// The rays [r0, r0 + slot_w) of the block's envs through the staged walk (RcStage above).  Every thread of the block
// calls this (the barriers inside are block-wide); `live`: this lane casts ray r of env e.  The table's keys are free
// (-1) on entry, behind a barrier, and are left in use: the caller frees them behind the barrier that ends its iteration.
__device__ __forceinline__ void raycaster_down_rays_staged(const RaycasterUpdateArgs& a, RcStage* st, bool live, int e, int r,
                                                           V3 pos, Q4 qr, uint64_t seed, uint64_t epoch) {
    const mdp_mesh_view_t& m = a.mesh;
    const int R = a.R;
    const float inf = __builtin_inff();
    V3 s{0.0f, 0.0f, 0.0f}, d{0.0f, 0.0f, -1.0f};
    mdp_hit_t h; h.t = a.max_dist; h.face = -1;
    int32_t b = 0, n = 0;
    bool disjoint = false;
    if (live) {
        s = quat_apply(qr, ld3(a.starts_l + 3 * r));
        s = V3{s.x + pos.x, s.y + pos.y, s.z + pos.z};
        d = ld3(a.dirs_l + 3 * r);
        // the cell's entry and the coarse patch entry are requested together: one round trip, not patch -> cell
        int32_t c;
        const bool cell_ok = mdp_rc_down_cell(&m, s.x, s.y, &c);
        int2 cref{0, 0};
        if (cell_ok) cref = reinterpret_cast<const int2*>(m.cell_start_down)[c];
        bool patched = false;
        int pcell;
        if (m.patch_cells && mdp_rc_patch_cell(&m, s.x, s.y, &pcell)) {
            const int4* pc = reinterpret_cast<const int4*>(m.patch_cells) + 2 * pcell;
            const int4 c0 = pc[0], c1 = pc[1];
            int vidx, ncols;
            if (mdp_rc_patch_square(c0, c1, s.x, s.y, &vidx, &ncols)) {
                const float4* V = reinterpret_cast<const float4*>(m.patch_verts) + vidx;
                patched = mdp_rc_patch_eval(V[0], V[1], V[ncols], V[ncols + 1], s.x, s.y, s.z, d.z, a.max_dist, &h) != 0;
            }
        }
        if (!patched) {
            h.t = a.max_dist; h.face = -1;
            if (cell_ok) { disjoint = cref.x < 0; b = MDP_RC_CELL_OFFSET(cref.x); n = cref.y; }
        }
    }
    // ---- this ray's list into the table (linear probing; the first ray of a list owns the slot's count)
    int slot = -1;
    if (n > 0) {
        unsigned hsh = ((unsigned)b * 2654435761u) >> 16;
        for (int p = 0; p < RC_STAGE_PROBES; ++p) {
            const int at = (int)((hsh + (unsigned)p) & (unsigned)(RC_STAGE_SLOTS - 1));
            const int old = atomicCAS(&st->key[at], -1, (int)b);
            if (old == -1) st->cnt[at] = n < RC_STAGE_K ? n : RC_STAGE_K;
            if (old == -1 || old == (int)b) { slot = at; break; }
        }
    }
    __syncthreads();
    MDP_STAMP(5);
    // ---- the distinct lists' first records, 16 bytes per thread and load, four loads of a thread in flight together
    {
        constexpr int CH = RC_STAGE_SLOTS * RC_STAGE_K * 4;
        const float4* recs = reinterpret_cast<const float4*>(m.cell_recs_down);
        float4* dst = &st->rec[0][0][0];
        auto src_of = [&](int i) -> const float4* {     // chunk i of the table: its address in the record array, or null
            if (i >= CH) return nullptr;
            const int sl = i / (RC_STAGE_K * 4), ch = i - sl * (RC_STAGE_K * 4);
            const int key = st->key[sl];
            return (key >= 0 && (ch >> 2) < st->cnt[sl]) ? recs + ((int64_t)key * 4 + ch) : nullptr;
        };
        const int step = (int)blockDim.x;
        for (int i = threadIdx.x; i < CH; i += 4 * step) {
            const float4 *p0 = src_of(i), *p1 = src_of(i + step), *p2 = src_of(i + 2 * step), *p3 = src_of(i + 3 * step);
            float4 v0, v1, v2, v3;
            if (p0) v0 = *p0;
            if (p1) v1 = *p1;
            if (p2) v2 = *p2;
            if (p3) v3 = *p3;
            if (p0) dst[i] = v0;
            if (p1) dst[i + step] = v1;
            if (p2) dst[i + 2 * step] = v2;
            if (p3) dst[i + 3 * step] = v3;
        }
    }
    __syncthreads();
    MDP_STAMP(6);
    // ---- the walk of mdp_rc_grid_down_exact, record k from LDS while it is staged
    if (n > 0) {
        const float Sz = 1.0f / d.z;
        bool stop = false;
        for (int32_t k = 0; k < n && !stop; ++k) {
            mdp_rec_t rec;
            if (slot >= 0 && k < RC_STAGE_K) {
                const float4* q = st->rec[slot][k];
                const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                rec.v[0] = q0.x; rec.v[1] = q0.y; rec.v[2] = q0.z; rec.v[3] = q0.w;
                rec.v[4] = q1.x; rec.v[5] = q1.y; rec.v[6] = q1.z; rec.v[7] = q1.w;
                rec.v[8] = q2.x; rec.v[9] = q2.y; rec.v[10] = q2.z; rec.v[11] = q2.w;
                rec.v[12] = q3.x;
            } else {
                rec = mdp_rc_load_rec(m.cell_recs_down + MDP_RC_REC_STRIDE * (int64_t)(b + k));
            }
            stop = mdp_rc_down_exact_record(rec.v, s.x, s.y, s.z, Sz, disjoint, &h);
        }
    }
    MDP_STAMP(4);
    if (live) {
        float* o = a.hits + ((int64_t)e * R + r) * 3;
        float hz = inf;
        if (h.face >= 0) {
            hz = s.z + h.t * d.z;
            o[0] = s.x + h.t * d.x; o[1] = s.y + h.t * d.y; o[2] = hz;
        } else {
            o[0] = inf; o[1] = inf; o[2] = inf;
        }
        if (a.scan.obs) scan_store(a.scan, e, r, R, pos.z, hz, seed, epoch);
    }
}

// `stage_lds`: rc_stage_lds_bytes() of (dynamic) LDS, 16-byte aligned, when a.stage is set
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

// The same for the launch shape of the 4096-env step: ONE env per block, one ray per lane (R <= blockDim.x, as many
// blocks as envs).  No env / ray loops, no lane -> (env slot, ray) division, the env index is the block's: the per-env
// addresses are wave-uniform and the routine needs a handful of vector registers less than the general one above --
// which is what keeps the post-physics kernel's 80-register instantiation out of scratch.  Same statements per ray.
template <int BATCH>
__device__ __forceinline__ void raycaster_update_down_block_one(const RaycasterUpdateArgs& a, unsigned block) {
    __shared__ float s_env1[6];   // pos.xyz, yaw quaternion (w, z), cast flag
    const int R = a.R;
    // XCD-aware env order.  The launch's blocks are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, each
    // XCD has its own L2), so with env = block the envs e .. e + 7 -- neighbours in every per-env array (pose rows,
    // time stamps, flags: several envs per 64-byte sector) -- are fetched into eight L2s.  Blocks of one residue class
    // take RUNS of 2^MDP_CAST_XCD_RUN consecutive envs instead, the runs dealt round-robin (a bijection of [0, N) whenever
    // 8 << RUN divides N; affinity only, nothing depends on where a block runs; the first 8 << RUN blocks still cover the
    // first 8 << RUN envs, so the launch keeps its env order at that grain).  Measured at 4096 envs (exp51): runs of
    // 8 / 16 / 32 / 64 envs: post-physics launch 29.55 -> 29.3 / 28.95 / 29.1 / 29.7 us.  MDP_CAST_XCD_ORDER=1, each XCD
    // one contiguous EIGHTH of the envs, is 11 us SLOWER (40.8 us): TerrainImporter maps consecutive envs to the same
    // terrain column, a column is one terrain type, and an XCD that draws the stair columns casts for 40 us while the
    // XCDs with the height-field columns idle -- blocks are dealt to XCDs statically, the launch ends with its slowest XCD.
#ifndef MDP_CAST_XCD_ORDER
#define MDP_CAST_XCD_ORDER 2
#endif
#ifndef MDP_CAST_XCD_RUN
#define MDP_CAST_XCD_RUN 4
#endif
    const unsigned nenv = (unsigned)a.N;
    int e = (int)block;
    if (MDP_CAST_XCD_ORDER == 1 && (nenv & 7u) == 0) e = (int)((block & 7u) * (nenv >> 3) + (block >> 3));
    if (MDP_CAST_XCD_ORDER == 2 && (nenv & ((8u << MDP_CAST_XCD_RUN) - 1u)) == 0) {
        const unsigned k = block >> 3;
        e = (int)(((k >> MDP_CAST_XCD_RUN) << (MDP_CAST_XCD_RUN + 3)) + ((block & 7u) << MDP_CAST_XCD_RUN) + (k & ((1u << MDP_CAST_XCD_RUN) - 1u)));
    }
    const int r = (int)threadIdx.x;
    const mdp_height_scan_obs_t& scan = a.scan;
    uint64_t seed = 0, epoch = 0;
    if (scan.obs && scan.rng) { seed = scan.rng[0]; epoch = scan.rng[1 + block]; }
    float sv[6] = {0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f};
    RcBook bk{};
    if (a.pose_ready) {
        const float4* w = reinterpret_cast<const float4*>(a.pose_ws + 8 * (int64_t)e);
        const float4 w0 = w[0], w1 = w[1];
        sv[0] = w0.x; sv[1] = w0.y; sv[2] = w0.z; sv[3] = w0.w; sv[4] = w1.x; sv[5] = w1.y;
    } else {
        // (lane 0's stores to the sensor's buffers go out at the END of the routine: in front of the barrier the barrier
        // would wait for their acknowledgements, right behind it the lane's own ray would)
        if (r == 0) raycaster_env_bookkeeping_deferred(a, e, s_env1, bk);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 6; ++k) sv[k] = s_env1[k];
    }
    MDP_STAMP(2);
    const bool od = sv[5] != 0.0f;
    if (r < R) {
        if (od) {
            MDP_STAMP(3);
            raycaster_down_ray<BATCH>(a, e, r, V3{sv[0], sv[1], sv[2]}, Q4{sv[3], 0.0f, 0.0f, sv[4]}, seed, epoch);
        } else if (scan.obs) {
            scan_store(scan, e, r, R, a.pos_w[3 * (int64_t)e + 2], a.hits[((int64_t)e * R + r) * 3 + 2], seed, epoch);
        }
    }
    if (!a.pose_ready && r == 0) raycaster_env_bookkeeping_commit(a, e, bk);
    if (scan.obs && scan.rng && r == 0) scan.rng[1 + block] = epoch + 1;
}

// The same block routine with the staged walk (the host sets a.stage, picks the kernel instantiation that calls this
// one and provides rc_stage_lds_bytes() of dynamic LDS at `stage_lds`, 16-byte aligned; meshes with non-finite
// vertices are not staged).  A function of its own: the per-ray routine above stays exactly as it was measured.
__device__ __forceinline__ void raycaster_update_down_blocks_staged(const RaycasterUpdateArgs& a, int epb, int slot_w,
                                                                    unsigned block, unsigned nblocks,
                                                                    unsigned char* stage_lds) {
    __shared__ float s_senv[RCU_MAX_EPB_STAGED][6];   // pos.xyz, yaw quaternion (w, z), cast flag
    RcStage* st = reinterpret_cast<RcStage*>(stage_lds);
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
        // free the table (visible behind the barrier below)
        for (int i = threadIdx.x; i < RC_STAGE_SLOTS; i += blockDim.x) st->key[i] = -1;
        if (a.pose_ready) {
            if (active) {
                const float4* w = reinterpret_cast<const float4*>(a.pose_ws + 8 * (int64_t)e);
                const float4 w0 = w[0], w1 = w[1];
                sv[0] = w0.x; sv[1] = w0.y; sv[2] = w0.z; sv[3] = w0.w; sv[4] = w1.x; sv[5] = w1.y;
            }
            __syncthreads();
        } else {
            if (active && lr == 0) raycaster_env_bookkeeping(a, e, s_senv[le]);
            __syncthreads();
            const float* sl = s_senv[active ? le : 0];
#pragma unroll
            for (int k = 0; k < 6; ++k) sv[k] = sl[k];
        }
        MDP_STAMP(2);
        const bool od = active && sv[5] != 0.0f;
        if (active && !od && scan.obs) {
            float sz = a.pos_w[3 * (int64_t)e + 2];
            for (int r = lr; r < R; r += slot_w) scan_store(scan, e, r, R, sz, hits[((int64_t)e * R + r) * 3 + 2], seed, epoch);
        }
        const V3 pos = V3{sv[0], sv[1], sv[2]};
        const Q4 qr = Q4{sv[3], 0.0f, 0.0f, sv[4]};
        MDP_STAMP(3);
        for (int r0 = 0; r0 < R; r0 += slot_w) {          // the same trip count for every thread of the block
            if (r0 > 0) {
                __syncthreads();                          // the previous group's walks are through: free the table
                for (int i = threadIdx.x; i < RC_STAGE_SLOTS; i += blockDim.x) st->key[i] = -1;
                __syncthreads();
            }
            raycaster_down_rays_staged(a, st, od && r0 + lr < R, e, r0 + lr, pos, qr, seed, epoch);
        }
        __syncthreads();                                  // s_senv and the table are rewritten by the next env group
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
                Q4 qr = yaw_quat_halfangle(q);        // (the per-step cast's form: both give the same bits)
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
        if (a.mesh.finite) {
            for (int r = threadIdx.x; r < R; r += blockDim.x)
                raycaster_down_ray_eager(a, base + le, r, pos, qr, seed, epoch, SCAN_STREAM_FLAGGED);
        } else {
            for (int r = threadIdx.x; r < R; r += blockDim.x)
                raycaster_down_ray<BATCH>(a, base + le, r, pos, qr, seed, epoch, SCAN_STREAM_FLAGGED);
        }
    }
    if (any && scan.obs && scan.rng && threadIdx.x == 0) scan.rng[1 + block] = epoch + 1;
}

}  // namespace mdp
