// Host side of the flagged recast (mdp_raycaster_update_flagged): argument checks and the kernel argument block, shared
// by the stand-alone launch (raycast.hip) and the fused reset tail of the LSTM actuator launch (actuator_nets.hip).
#pragma once
#include <cstring>
#include "mdp_host.hpp"
#include "mesh_handle.hpp"
#include "raycaster_update_device.hpp"

namespace mdp {

inline int raycaster_flagged_args(int num_envs, const mdp_raycaster_update_args_t* cast, const uint8_t* reset_mask,
                                  RaycasterUpdateArgs* out) {
    MDP_REQUIRE(cast, "null cast arguments");
    MDP_REQUIRE(cast->mesh && cast->root_pose && cast->ray_starts_local && cast->ray_dirs_local && cast->pos_w &&
                    cast->quat_w && cast->ray_hits_w && cast->num_rays > 0, "cast: null pointer");
    MDP_REQUIRE(cast->timestamp && cast->timestamp_last_update && cast->is_outdated, "cast: bookkeeping buffers required");
    MDP_REQUIRE(cast->ld_pose >= 7, "cast: root_pose must be (N,>=7)");
    if (!(cast->pattern_vertical_down && cast->attach_yaw_only && cast->mesh->view.accel == 1 && cast->mesh->view.cell_recs))
        return fail(MDP_ERR_UNSUPPORTED, "flagged recast needs a cast that qualifies for the vertical-ray kernel");
    RaycasterUpdateArgs& ra = *out;
    std::memset(&ra, 0, sizeof(ra));
    ra.mesh = cast->mesh->view; ra.N = num_envs; ra.R = cast->num_rays; ra.pose = cast->root_pose; ra.ld_pose = cast->ld_pose;
    ra.pose_xyzw = cast->pose_xyzw; ra.drift = cast->drift; ra.starts_l = cast->ray_starts_local;
    ra.dirs_l = cast->ray_dirs_local; ra.max_dist = cast->max_dist; ra.pos_w = cast->pos_w; ra.quat_w = cast->quat_w;
    ra.hits = cast->ray_hits_w; ra.ts = cast->timestamp; ra.ts_last = cast->timestamp_last_update;
    ra.outdated = cast->is_outdated; ra.period = cast->update_period; ra.reset_mask = reset_mask;
    ra.pose_ws = nullptr; ra.pose_ready = 0;
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
    return MDP_OK;
}

}  // namespace mdp
