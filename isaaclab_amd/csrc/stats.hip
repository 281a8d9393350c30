// Per-rank partial statistics for the two cross-GPU reductions:
// obs-normaliser moments (Chan-mergeable, fp64) and masked episode-statistic sums.
// Reference arithmetic: my_code/models/my_running_standard_scaler.py:57-80,
// LAB/managers/reward_manager.py:115-121, termination_manager.py:142-144, command_manager.py:136-142.
#include "mdp_device.hpp"
#include "mdp_host.hpp"
#include <cstring>

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

template <typename T>
__global__ void k_masked_sums(const T* __restrict__ v, int rows, int N, const uint8_t* __restrict__ mask,
                              double* __restrict__ out) {
    int k = blockIdx.x;
    __shared__ double rs[256];
    __shared__ double rc[256];
    double s = 0.0, c = 0.0;
    // eight elements per trip, their loads issued before the first add (with a branch per element every load is its own
    // round trip, N / 256 of them per thread); a masked-out element adds an exact zero: same order, same sums
    const T* row = v + (int64_t)k * N;
    for (int e0 = threadIdx.x; e0 < N; e0 += 8 * blockDim.x) {
        T x[8];
        uint8_t m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + j * blockDim.x;
            const bool in = e < N;
            x[j] = in ? row[e] : (T)0;
            m[j] = in ? (mask ? mask[e] : (uint8_t)1) : (uint8_t)0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s += m[j] ? (double)x[j] : 0.0;
            c += m[j] ? 1.0 : 0.0;
        }
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

// The same for several (K_i, N) blocks of different element types in ONE launch (mdp_stats_masked_sums_multi): block b
// finds its source and row from the table's running row count; the per-row routine, its order of additions and so its
// sums are k_masked_sums' own.
struct StatSources { mdp_stats_source_t s[MDP_STATS_MAX_SOURCES]; int n; };

template <typename T>
__device__ __forceinline__ void masked_sums_row(const T* __restrict__ row, int N, const uint8_t* __restrict__ mask,
                                                double* __restrict__ out2) {
    __shared__ double rs[256];
    __shared__ double rc[256];
    double s = 0.0, c = 0.0;
    for (int e0 = threadIdx.x; e0 < N; e0 += 8 * blockDim.x) {
        T x[8];
        uint8_t m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + j * blockDim.x;
            const bool in = e < N;
            x[j] = in ? row[e] : (T)0;
            m[j] = in ? (mask ? mask[e] : (uint8_t)1) : (uint8_t)0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s += m[j] ? (double)x[j] : 0.0;
            c += m[j] ? 1.0 : 0.0;
        }
    }
    rs[threadIdx.x] = s;
    rc[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { rs[threadIdx.x] += rs[threadIdx.x + off]; rc[threadIdx.x] += rc[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = rs[0]; out2[1] = rc[0]; }
}

__global__ void __launch_bounds__(256) k_masked_sums_multi(StatSources t, int N, const uint8_t* __restrict__ mask,
                                                            double* __restrict__ out) {
    int k = blockIdx.x, i = 0;
    while (i < t.n - 1 && k >= t.s[i].rows) { k -= t.s[i].rows; ++i; }     // (block-uniform)
    const mdp_stats_source_t src = t.s[i];
    const uint8_t* m = src.masked ? mask : nullptr;
    double* o = out + 2 * (int64_t)blockIdx.x;
    if (src.elem_kind == 0) masked_sums_row(reinterpret_cast<const float*>(src.values) + (int64_t)k * N, N, m, o);
    else if (src.elem_kind == 1) masked_sums_row(reinterpret_cast<const uint8_t*>(src.values) + (int64_t)k * N, N, m, o);
    else masked_sums_row(reinterpret_cast<const int64_t*>(src.values) + (int64_t)k * N, N, m, o);
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

// Batch moments in row chunks: block (column group of 32, chunk) reads its rows coalesced (32 consecutive floats per
// row), two passes over its own chunk (sum -> chunk mean -> centred M2, the second pass hits L2), and writes one
// Chan-mergeable partial row [n, sum x (D), M2 (D)] per chunk.  gridDim = (ceil(D / 32), num_chunks), block (32, 8).
__global__ void k_moments_chunks(const float* __restrict__ x, int64_t rows, int D, int64_t ld, int num_chunks,
                                 double* __restrict__ partials) {
    const int c = blockIdx.x * 32 + threadIdx.x;
    const int chunk = blockIdx.y;
    const int64_t per = (rows + num_chunks - 1) / num_chunks;
    const int64_t r0 = chunk * per, r1 = r0 + per < rows ? r0 + per : rows;
    const int64_t n = r1 > r0 ? r1 - r0 : 0;
    __shared__ double red[8][33];
    double s = 0.0;
    if (c < D)
        for (int64_t r = r0 + threadIdx.y; r < r1; r += 8) s += (double)x[r * ld + c];
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    double sum = 0.0;
    for (int k = 0; k < 8; ++k) sum += red[k][threadIdx.x];
    const double mean = n > 0 ? sum / (double)n : 0.0;
    __syncthreads();
    double m2 = 0.0;
    if (c < D)
        for (int64_t r = r0 + threadIdx.y; r < r1; r += 8) {
            double d = (double)x[r * ld + c] - mean;
            m2 += d * d;
        }
    red[threadIdx.y][threadIdx.x] = m2;
    __syncthreads();
    if (threadIdx.y == 0 && c < D) {
        double t = 0.0;
        for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
        double* row = partials + (int64_t)chunk * (1 + 2 * D);
        if (c == 0) row[0] = (double)n;
        row[1 + c] = sum;
        row[1 + D + c] = t;
    }
}

// RunningStandardScaler._parallel_variance (my_running_standard_scaler.py:57-80) fed by partial rows (row chunks of this
// rank's batch, and the other ranks' chunks behind them after an all-gather): Chan merge of the rows in order -> batch
// mean and Bessel-corrected variance (torch.var) -> running update -> the fp32 casts mdp_scaler_apply reads.
// One block; thread per column.
__global__ void k_scaler_update(int P, int D, const double* __restrict__ partials, double* __restrict__ running_mean,
                                double* __restrict__ running_var, double* __restrict__ count, float eps,
                                float* __restrict__ mean32, float* __restrict__ std32, float* __restrict__ denom) {
    const double cur = count[0];
    double total_n = 0.0;
    for (int p = 0; p < P; ++p) total_n += partials[(int64_t)p * (1 + 2 * D)];
    __syncthreads();                                   // every thread has read the count before thread 0 replaces it
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        double n = 0.0, mean = 0.0, m2 = 0.0;
        for (int p = 0; p < P; ++p) {
            const double* row = partials + (int64_t)p * (1 + 2 * D);
            const double nb = row[0];
            if (nb <= 0.0) continue;
            const double mb = row[1 + c] / nb;
            const double delta = mb - mean, tot = n + nb;
            mean += delta * nb / tot;
            m2 += row[1 + D + c] + delta * delta * n * nb / tot;
            n = tot;
        }
        if (n > 0.0) {
            // torch.var (Bessel): a single-row batch gives 0 / 0 = NaN, which the reference lets into its running variance
            // (my_running_standard_scaler.py:74-80); reproduced, not repaired
            const double var_b = m2 / (n - 1.0);
            const double delta = mean - running_mean[c];
            const double tot = cur + n;
            const double M2 = running_var[c] * cur + var_b * n + delta * delta * cur * n / tot;
            running_mean[c] = running_mean[c] + delta * n / tot;
            running_var[c] = M2 / tot;
        }
        const float sd = sqrtf((float)running_var[c]);
        mean32[c] = (float)running_mean[c];
        std32[c] = sd;
        denom[c] = sd + eps;
    }
    if (threadIdx.x == 0 && total_n > 0.0) count[0] = cur + total_n;
}

// my_running_standard_scaler.py:108-119: the fp64 running statistics are cast to fp32 once per update, not per element
__global__ void k_scaler_prepare(int D, const double* __restrict__ mean, const double* __restrict__ var, float eps,
                                 float* __restrict__ mean32, float* __restrict__ std32, float* __restrict__ denom) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    float s = sqrtf((float)var[c]);
    mean32[c] = (float)mean[c];
    std32[c] = s;
    denom[c] = s + eps;
}

// clamp((x - mean) / (sqrt(var) + eps), -clip, clip) -- or the inverse map -- written wherever the caller wants it
// (a slot of the rollout storage): one pass over (rows, D)
template <bool INVERSE>
__global__ void k_scaler_apply(int64_t rows, int D, const float* __restrict__ x, int64_t ld, const float* __restrict__ mean32,
                               const float* __restrict__ std32, const float* __restrict__ denom, float clip,
                               float* __restrict__ out, int64_t ld_out) {
    const int64_t total = rows * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        const int c = (int)(i - r * D);
        const float v = x[r * ld + c];
        float y;
        if (INVERSE) y = std32[c] * tclamp(v, -clip, clip) + mean32[c];
        else y = tclamp((v - mean32[c]) / denom[c], -clip, clip);
        out[r * ld_out + c] = y;
    }
}

}  // namespace

extern "C" int mdp_stats_moments_chunked(const float* x, int64_t rows, int dim, int64_t ld, int num_chunks,
                                         double* partials, void* stream) {
    MDP_REQUIRE(x && partials, "null pointer");
    MDP_REQUIRE(rows >= 0 && dim > 0 && ld >= dim, "bad dim/ld");
    MDP_REQUIRE(num_chunks >= 1 && num_chunks <= 65535, "num_chunks out of range");
    hipLaunchKernelGGL(k_moments_chunks, dim3((unsigned)((dim + 31) / 32), (unsigned)num_chunks), dim3(32, 8), 0,
                       as_stream(stream), x, rows, dim, ld, num_chunks, partials);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_scaler_update(int num_partials, int dim, const double* partials, double* running_mean,
                                 double* running_variance, double* current_count, float epsilon, float* mean32,
                                 float* std32, float* denom, void* stream) {
    MDP_REQUIRE(partials && running_mean && running_variance && current_count && mean32 && std32 && denom, "null pointer");
    MDP_REQUIRE(num_partials >= 0 && dim > 0, "bad shape");
    hipLaunchKernelGGL(k_scaler_update, dim3(1), dim3(256), 0, as_stream(stream), num_partials, dim, partials,
                       running_mean, running_variance, current_count, epsilon, mean32, std32, denom);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_scaler_prepare(int dim, const double* running_mean, const double* running_variance, float epsilon,
                                  float* mean32, float* std32, float* denom, void* stream) {
    MDP_REQUIRE(running_mean && running_variance && mean32 && std32 && denom, "null pointer");
    if (dim <= 0) return MDP_OK;
    hipLaunchKernelGGL(k_scaler_prepare, dim3(grid_for(dim, 256)), dim3(256), 0, as_stream(stream), dim, running_mean,
                       running_variance, epsilon, mean32, std32, denom);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_scaler_apply(int64_t rows, int dim, const float* x, int64_t ld, const float* mean32, const float* std32,
                                const float* denom, float clip_threshold, int inverse, float* out, int64_t ld_out,
                                void* stream) {
    MDP_REQUIRE(x && mean32 && std32 && denom && out, "null pointer");
    MDP_REQUIRE(rows >= 0 && dim > 0 && ld >= dim && ld_out >= dim, "bad shape");
    if (rows == 0) return MDP_OK;
    const unsigned grid = grid_for(rows * dim, 256, 1 << 16);
    if (inverse)
        hipLaunchKernelGGL(k_scaler_apply<true>, dim3(grid), dim3(256), 0, as_stream(stream), rows, dim, x, ld, mean32, std32,
                           denom, clip_threshold, out, ld_out);
    else
        hipLaunchKernelGGL(k_scaler_apply<false>, dim3(grid), dim3(256), 0, as_stream(stream), rows, dim, x, ld, mean32, std32,
                           denom, clip_threshold, out, ld_out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

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
    hipLaunchKernelGGL(k_masked_sums<float>, dim3(rows), dim3(256), 0, as_stream(stream), values, rows, num_envs, mask, out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_stats_masked_sums_typed(const void* values, int elem_kind, int rows, int num_envs, const uint8_t* mask,
                                           double* out, void* stream) {
    MDP_REQUIRE(values && out, "null pointer");
    MDP_REQUIRE(elem_kind >= 0 && elem_kind <= 2, "unknown element kind %d", elem_kind);
    if (rows == 0) return MDP_OK;
    hipStream_t st = as_stream(stream);
    if (elem_kind == 0)
        hipLaunchKernelGGL(k_masked_sums<float>, dim3(rows), dim3(256), 0, st, (const float*)values, rows, num_envs, mask, out);
    else if (elem_kind == 1)
        hipLaunchKernelGGL(k_masked_sums<uint8_t>, dim3(rows), dim3(256), 0, st, (const uint8_t*)values, rows, num_envs, mask, out);
    else
        hipLaunchKernelGGL(k_masked_sums<int64_t>, dim3(rows), dim3(256), 0, st, (const int64_t*)values, rows, num_envs, mask, out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_stats_masked_sums_multi(const mdp_stats_source_t* sources, int num_sources, int num_envs,
                                           const uint8_t* mask, double* out, void* stream) {
    MDP_REQUIRE(sources && out, "null pointer");
    MDP_REQUIRE(num_sources >= 0 && num_sources <= MDP_STATS_MAX_SOURCES, "at most %d sources (got %d)", MDP_STATS_MAX_SOURCES,
                num_sources);
    StatSources t;
    std::memset(&t, 0, sizeof(t));
    int64_t rows = 0;
    for (int i = 0; i < num_sources; ++i) {
        const mdp_stats_source_t& s = sources[i];
        MDP_REQUIRE(s.elem_kind >= 0 && s.elem_kind <= 2, "source %d: unknown element kind %d", i, s.elem_kind);
        MDP_REQUIRE(s.rows >= 0, "source %d: negative row count", i);
        if (s.rows == 0) continue;
        MDP_REQUIRE(s.values, "source %d: null pointer", i);
        MDP_REQUIRE(!s.masked || mask, "source %d follows the mask, but mask is NULL", i);
        t.s[t.n++] = s;
        rows += s.rows;
    }
    if (rows == 0) return MDP_OK;
    MDP_REQUIRE(rows < (1 << 30), "too many rows");
    hipLaunchKernelGGL(k_masked_sums_multi, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), t, num_envs, mask, out);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}
