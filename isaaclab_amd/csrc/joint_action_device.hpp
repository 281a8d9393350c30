// ActionManager.process_action + JointAction.process_actions for one (env, action column) element
// (action_manager.py:318-339, joint_actions.py:130-139): shared by mdp_joint_action_process (scene.hip) and by the
// actuator kernels that process the action of the step inside the first actuator launch (*_compute_with_action).
#pragma once
#include "mdp_device.hpp"
#include "../../include/mdp_hip.h"

namespace mdp {

// returns the processed action; `store` = this lane owns the element (writes prev_action / action / processed)
__device__ __forceinline__ float joint_action_element(const mdp_joint_action_t& a, int64_t e, int col, int A, bool store) {
    const int64_t i = e * A + col;
    const float raw = a.actions_in[e * a.ld_in + col];
    if (store) {
        if (a.prev_action) a.prev_action[i] = a.action[i];
        a.action[i] = raw;
    }
    const float sc = a.scale_vec ? a.scale_vec[col] : a.scale;
    float p = raw * sc + (a.offset ? a.offset[i] : 0.0f);
    if (a.clip) p = tclamp(p, a.clip[2 * i], a.clip[2 * i + 1]);
    if (store && a.processed) a.processed[i] = p;
    return p;
}

}  // namespace mdp
