// Uniform draws of the reset-path kernels (events.hip, inhand.hip): either injected (parity tests hand the oracle and the
// kernel the same numbers) or Philox keyed by (seed, per-env epoch).  The epoch word of an env advances only when that
// env draws, so a launch never races on a shared counter and a captured hipGraph keeps drawing new numbers.
#pragma once
#include "mdp_device.hpp"

namespace mdp {

// k-th uniform of env e: injected row or Philox(seed, epoch, stream, e * 64 + k / 4)
struct Uniforms {
    const float* inj;
    int per_env;
    uint64_t seed, epoch;
    uint32_t stream;
    int e;
    __device__ __forceinline__ float operator()(int k) const {
        if (inj) return inj[(int64_t)e * per_env + k];
        U4 r = rng4(seed, epoch, stream, (uint64_t)e * 64u + (uint64_t)(k >> 2));
        uint32_t w = (k & 3) == 0 ? r.x : ((k & 3) == 1 ? r.y : ((k & 3) == 2 ? r.z : r.w));
        return u01(w);
    }
};

// rng_state = {seed, epoch of env 0, epoch of env 1, ...}; the calling thread owns env e's epoch word
__device__ __forceinline__ Uniforms uniforms_for(const float* inj, int per_env, uint64_t* rng_state, uint32_t stream, int e) {
    Uniforms u{inj, per_env, 0, 0, stream, e};
    if (!inj && rng_state) {
        u.seed = rng_state[0];
        u.epoch = rng_state[1 + e];
        rng_state[1 + e] = u.epoch + 1;
    }
    return u;
}

}  // namespace mdp
