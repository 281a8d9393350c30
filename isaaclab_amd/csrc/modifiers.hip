// Class-based observation modifiers.  Reference: LAB/utils/modifiers/modifier.py:67-259
#include "mdp_device.hpp"
#include "mdp_host.hpp"

using namespace mdp;

namespace {

__global__ void k_digital_filter(int64_t count, const float* __restrict__ data, float* __restrict__ xh, int nb,
                                 float* __restrict__ yh, int na, const float* __restrict__ B,
                                 const float* __restrict__ A, float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        float* x = xh + i * nb;
        float* y = yh + i * na;
        // roll input window, newest first (modifier.py:180-181)
        for (int j = nb - 1; j >= 1; --j) x[j] = x[j - 1];
        x[0] = data[i];
        float xb = 0.0f, ya = 0.0f;
        for (int j = 0; j < nb; ++j) xb += x[j] * B[j];
        for (int j = 0; j < na; ++j) ya += y[j] * A[j];
        float yi = xb - ya;
        for (int j = na - 1; j >= 1; --j) y[j] = y[j - 1];
        y[0] = yi;
        out[i] = yi;
    }
}

__global__ void k_integrator(int64_t count, const float* __restrict__ data, float* __restrict__ integral,
                             float* __restrict__ y_prev, float dt, float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        float d = data[i];
        float v = integral[i] + (d + y_prev[i]) / 2.0f * dt;
        integral[i] = v;
        y_prev[i] = d;
        out[i] = v;
    }
}

__global__ void k_rows_zero(float* __restrict__ buf, int N, int64_t per_env, const uint8_t* __restrict__ mask) {
    int64_t total = (int64_t)N * per_env;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int e = (int)(i / per_env);
        if (mask && !mask[e]) continue;
        buf[i] = 0.0f;
    }
}

}  // namespace

extern "C" int mdp_modifier_digital_filter(int64_t count, const float* data, float* x_hist, int nb, float* y_hist,
                                           int na, const float* B, const float* A, float* out, void* stream) {
    MDP_REQUIRE(data && x_hist && y_hist && B && A && out, "null pointer");
    MDP_REQUIRE(nb >= 1 && na >= 1, "Digital filter coefficients A and B must not be None.");
    if (count == 0) return MDP_OK;
    hipLaunchKernelGGL(k_digital_filter, dim3(grid_for(count, 256)), dim3(256), 0, as_stream(stream), count, data,
                       x_hist, nb, y_hist, na, B, A, out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_modifier_integrator(int64_t count, const float* data, float* integral, float* y_prev, float dt,
                                       float* out, void* stream) {
    MDP_REQUIRE(data && integral && y_prev && out, "null pointer");
    if (count == 0) return MDP_OK;
    hipLaunchKernelGGL(k_integrator, dim3(grid_for(count, 256)), dim3(256), 0, as_stream(stream), count, data,
                       integral, y_prev, dt, out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_rows_zero(float* buf, int num_envs, int64_t per_env, const uint8_t* env_mask, void* stream) {
    MDP_REQUIRE(buf, "null pointer");
    int64_t total = (int64_t)num_envs * per_env;
    if (total == 0) return MDP_OK;
    hipLaunchKernelGGL(k_rows_zero, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), buf, num_envs,
                       per_env, env_mask);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}
