// Observation post-processing shared by the observation program (programs.hip) and the height-scan epilogue of the ray
// caster (raycast.hip).
#pragma once
#include "mdp_device.hpp"
#include "../../include/mdp_hip.h"

namespace mdp {

// observation_manager.py:313-318: noise -> clip -> scale
MDP_DEV float obs_post(const mdp_term_t& t, float v, int e, int k, uint64_t seed, uint64_t epoch, uint32_t stream,
                       uint64_t index) {
    if (t.noise_kind == 1) {
        float u = t.noise_u ? t.noise_u[(int64_t)e * t.dim + k] : rng_uniform(seed, epoch, stream, index);
        float n = u * t.n1;  // n1 = (n_max - n_min)
        if (t.noise_op == 0) v = (v + n) + t.n0;
        else if (t.noise_op == 1) v = v * (n + t.n0);
        else v = n + t.n0;
    } else if (t.noise_kind == 2) {
        float z = t.noise_u ? t.noise_u[(int64_t)e * t.dim + k] : rng_normal(seed, epoch, stream, index);
        float n = t.n1 * z;  // std * randn
        if (t.noise_op == 0) v = (v + t.n0) + n;
        else if (t.noise_op == 1) v = v * (t.n0 + n);
        else v = t.n0 + n;
    } else if (t.noise_kind == 3) {
        if (t.noise_op == 0) v = v + t.n0;
        else if (t.noise_op == 1) v = v * t.n0;
        else v = 0.0f + t.n0;
    }
    if (t.has_clip) v = tclamp(v, t.clip_lo, t.clip_hi);
    if (t.has_scale == 1) v = v * t.scale;
    else if (t.has_scale == 2) v = v * t.scale_vec[k];
    return v;
}

}  // namespace mdp
