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

struct FillTable { mdp_fill_t f[MDP_MAX_FILLS]; };

// one wave per env: a wave whose env is not flagged leaves after one byte; a flagged one walks the buffers with its
// lanes over the elements (few envs are flagged in a step, so the launch is mostly the mask read)
__global__ void k_masked_fill(FillTable t, int num_fills, const uint8_t* __restrict__ mask, int N) {
    const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (e >= N || !mask[e]) return;
    for (int fi = 0; fi < num_fills; ++fi) {
        const mdp_fill_t& f = t.f[fi];
        const int64_t base = (int64_t)e * f.env_stride;
        for (int k = lane; k < f.count; k += 64) {
            const int64_t at = base + (int64_t)k * f.elem_stride;
            if (f.elem_bytes == 4) reinterpret_cast<uint32_t*>(f.ptr)[at] = (uint32_t)f.value_bits;
            else if (f.elem_bytes == 8) reinterpret_cast<uint64_t*>(f.ptr)[at] = f.value_bits;
            else reinterpret_cast<uint8_t*>(f.ptr)[at] = (uint8_t)f.value_bits;
        }
    }
}

// The values the managers' reset() log for the envs of `dones` (reward_manager.py:115-121, command_manager.py:131-136,
// termination_manager.py:149-150), one block per row: row 0 = number of reset envs, rows 1..Tr = mean episode sum /
// max_episode_length_s, then the means of the two command metrics, then one count per termination term.
__global__ void k_reset_log(int N, const uint8_t* __restrict__ dones, const float* __restrict__ sums, int Tr,
                            const float* __restrict__ metric_a, const float* __restrict__ metric_b,
                            const uint8_t* __restrict__ term_dones, int Tt, float max_episode_length_s,
                            float* __restrict__ log) {
    const int r = blockIdx.x;
    __shared__ double rs[256];
    __shared__ double rc[256];
    double s = 0.0, c = 0.0;
    for (int e = threadIdx.x; e < N; e += blockDim.x) {
        if (!dones[e]) continue;
        c += 1.0;
        if (r == 0) s += 1.0;
        else if (r <= Tr) s += (double)sums[(int64_t)(r - 1) * N + e];
        else if (r == Tr + 1) s += metric_a ? (double)metric_a[e] : 0.0;
        else if (r == Tr + 2) s += metric_b ? (double)metric_b[e] : 0.0;
        else s += term_dones[(int64_t)(r - Tr - 3) * N + e] ? 1.0 : 0.0;
    }
    rs[threadIdx.x] = s;
    rc[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { rs[threadIdx.x] += rs[threadIdx.x + off]; rc[threadIdx.x] += rc[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double sum = rs[0], cnt = rc[0];
        float v;
        if (r == 0 || r > Tr + 2) v = (float)sum;
        else {
            v = cnt > 0.0 ? (float)(sum / cnt) : 0.0f;
            if (r <= Tr) v = v / max_episode_length_s;
        }
        log[r] = v;
    }
}

}  // namespace

extern "C" int mdp_masked_fill(const mdp_fill_t* fills, int num_fills, const uint8_t* env_mask, int num_envs, void* stream) {
    MDP_REQUIRE(num_fills >= 0 && num_fills <= MDP_MAX_FILLS, "at most %d buffers per call (got %d)", MDP_MAX_FILLS, num_fills);
    MDP_REQUIRE(num_envs >= 0, "negative num_envs");
    if (num_fills == 0 || num_envs == 0) return MDP_OK;
    MDP_REQUIRE(fills && env_mask, "null pointer");
    FillTable t;
    std::memset(&t, 0, sizeof(t));
    int64_t largest = 0;
    for (int i = 0; i < num_fills; ++i) {
        const mdp_fill_t& f = fills[i];
        MDP_REQUIRE(f.ptr && f.count >= 0, "fill %d: null pointer or negative count", i);
        MDP_REQUIRE(f.elem_bytes == 1 || f.elem_bytes == 4 || f.elem_bytes == 8, "fill %d: elem_bytes %d", i, f.elem_bytes);
        t.f[i] = f;
        largest = std::max<int64_t>(largest, (int64_t)num_envs * f.count);
    }
    if (largest == 0) return MDP_OK;
    hipLaunchKernelGGL(k_masked_fill, dim3((unsigned)((num_envs + 3) / 4)), dim3(256), 0, as_stream(stream), t, num_fills,
                       env_mask, num_envs);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_reset_log(int num_envs, const uint8_t* dones, const float* episode_sums, int num_reward_terms,
                             const float* metric_a, const float* metric_b, const uint8_t* term_dones,
                             int num_termination_terms, float max_episode_length_s, float* log, void* stream) {
    MDP_REQUIRE(dones && log, "null pointer");
    MDP_REQUIRE(num_reward_terms >= 0 && num_termination_terms >= 0, "negative term count");
    MDP_REQUIRE(num_reward_terms == 0 || episode_sums, "null episode_sums");
    MDP_REQUIRE(num_termination_terms == 0 || term_dones, "null term_dones");
    MDP_REQUIRE(max_episode_length_s > 0.0f, "max_episode_length_s must be positive");
    const int rows = 1 + num_reward_terms + 2 + num_termination_terms;
    hipLaunchKernelGGL(k_reset_log, dim3(rows), dim3(256), 0, as_stream(stream), num_envs, dones, episode_sums,
                       num_reward_terms, metric_a, metric_b, term_dones, num_termination_terms, max_episode_length_s, log);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

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
