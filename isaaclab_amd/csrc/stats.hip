// Per-rank partial statistics for the two cross-GPU reductions:
// obs-normaliser moments (Chan-mergeable, fp64) and masked episode-statistic sums.
// Reference arithmetic: my_code/models/my_running_standard_scaler.py:57-80,
// LAB/managers/reward_manager.py:115-121, termination_manager.py:142-144, command_manager.py:136-142.
#include "mdp_device.hpp"
#include "mdp_host.hpp"

using namespace mdp;

namespace {

// one block per column tile of 64 columns x all rows: two passes (mean, then centred M2) in fp64.
// D <= 256, N up to 1e6: each thread strides over rows for one column; rows are reduced through LDS.
__global__ void k_moments(const float* __restrict__ x, int N, int D, int ld, double* __restrict__ out) {
    // grid.x = D (one block per column), block = 256 threads striding over rows
    int c = blockIdx.x;
    __shared__ double red[256];
    double s = 0.0;
    for (int e = threadIdx.x; e < N; e += blockDim.x) s += (double)x[(int64_t)e * ld + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    double sum = red[0];
    double mean = N > 0 ? sum / (double)N : 0.0;
    __syncthreads();
    double m2 = 0.0;
    for (int e = threadIdx.x; e < N; e += blockDim.x) {
        double d = (double)x[(int64_t)e * ld + c] - mean;
        m2 += d * d;
    }
    red[threadIdx.x] = m2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (c == 0) out[0] = (double)N;
        out[1 + c] = sum;
        out[1 + D + c] = red[0];
    }
}

__global__ void k_masked_sums(const float* __restrict__ v, int rows, int N, const uint8_t* __restrict__ mask,
                              double* __restrict__ out) {
    int k = blockIdx.x;
    __shared__ double rs[256];
    __shared__ double rc[256];
    double s = 0.0, c = 0.0;
    for (int e = threadIdx.x; e < N; e += blockDim.x) {
        if (!mask || mask[e]) { s += (double)v[(int64_t)k * N + e]; c += 1.0; }
    }
    rs[threadIdx.x] = s;
    rc[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { rs[threadIdx.x] += rs[threadIdx.x + off]; rc[threadIdx.x] += rc[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[2 * k] = rs[0]; out[2 * k + 1] = rc[0]; }
}

}  // namespace

extern "C" int mdp_stats_moments(const float* x, int num_envs, int dim, int ld, double* out, void* stream) {
    MDP_REQUIRE(x && out, "null pointer");
    MDP_REQUIRE(dim > 0 && ld >= dim, "bad dim/ld");
    hipLaunchKernelGGL(k_moments, dim3(dim), dim3(256), 0, as_stream(stream), x, num_envs, dim, ld, out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_stats_masked_sums(const float* values, int rows, int num_envs, const uint8_t* mask, double* out,
                                     void* stream) {
    MDP_REQUIRE(values && out, "null pointer");
    if (rows == 0) return MDP_OK;
    hipLaunchKernelGGL(k_masked_sums, dim3(rows), dim3(256), 0, as_stream(stream), values, rows, num_envs, mask, out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}
