// How should one lane fetch ONE private 64-byte record (a grid-cell record of the ray caster) from a table far larger than
// the caches?  Each of R lanes reads the record idx[lane] of a T-record table and adds its 16 floats up.
//   A  per lane, four global_load_dwordx4 to its own line (what mdp_rc_load_rec did)
//   B  per lane, first 16 B only (reference: one request per line, a quarter of the data)
//   C  quad-cooperative: lane 4k+q loads piece q of the record of ray 4k+i in instruction i (each instruction covers 16
//      whole records, no line is touched twice), transposed through LDS (ds_write_b128 / ds_read_b128)
//   D  the same through LDS-DMA (global_load_lds_dwordx4): the [instruction][lane] image IS record-contiguous
//   E  per lane, dependent index load + A (the cell_start -> record chain)
// hipcc --offload-arch=gfx950 -O3 tools/micro/gather64.hip -o /tmp/gather64 && /tmp/gather64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

__global__ void __launch_bounds__(192) k_A(const float4* __restrict__ tab, const int* __restrict__ idx, float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* r = tab + 4 * (size_t)idx[i];
    float4 a = r[0], b = r[1], c = r[2], d = r[3];
    out[i] = (sum4(a) + sum4(b)) + (sum4(c) + sum4(d));
}
__global__ void __launch_bounds__(192) k_B(const float4* __restrict__ tab, const int* __restrict__ idx, float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = sum4(tab[4 * (size_t)idx[i]]);
}
__global__ void __launch_bounds__(192) k_C(const float4* __restrict__ tab, const int* __restrict__ idx, float* __restrict__ out, int n) {
    __shared__ float4 s[3][4][64];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, q = lane & 3, base = lane & ~3;
    int my = i < n ? idx[i] : 0;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int rec = __shfl(my, base + k);          // record of ray base + k
        v[k] = tab[4 * (size_t)rec + q];         // its piece q
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[w][k][lane] = v[k];   // [instruction][lane]: ray base+k, piece q at [k][base+q]
    // ray `lane` = base + q: its pieces p are at [q][base + p]
    float4 a = s[w][q][base + 0], b = s[w][q][base + 1], c = s[w][q][base + 2], d = s[w][q][base + 3];
    if (i < n) out[i] = (sum4(a) + sum4(b)) + (sum4(c) + sum4(d));
}
__global__ void __launch_bounds__(192) k_D(const float4* __restrict__ tab, const int* __restrict__ idx, float* __restrict__ out, int n) {
    __shared__ float4 s[3][4][64];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, q = lane & 3, base = lane & ~3;
    int my = i < n ? idx[i] : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int rec = __shfl(my, base + k);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tab + 4 * (size_t)rec + q),
                                         (__attribute__((address_space(3))) void*)&s[w][k][0], 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float4 a = s[w][q][base + 0], b = s[w][q][base + 1], c = s[w][q][base + 2], d = s[w][q][base + 3];
    if (i < n) out[i] = (sum4(a) + sum4(b)) + (sum4(c) + sum4(d));
}
__global__ void __launch_bounds__(192) k_E(const float4* __restrict__ tab, const int* __restrict__ idx, const int* __restrict__ idx2,
                                           float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = idx2[i];                      // random position in a 10 MB int table ...
    int lo = idx[c], hi = idx[c + 1];     // ... whose two neighbouring entries are read by two loads (cell_start pair)
    const float4* r = tab + 4 * (size_t)(lo + (hi & 1));
    float4 a = r[0], b = r[1], cc = r[2], d = r[3];
    out[i] = (sum4(a) + sum4(b)) + (sum4(cc) + sum4(d));
}
__global__ void __launch_bounds__(192) k_F(const float4* __restrict__ tab, const int2* __restrict__ idxp, const int* __restrict__ idx2,
                                           float* __restrict__ out, int n) {
    __shared__ float4 s[3][4][64];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, q = lane & 3, base = lane & ~3;
    int c = i < n ? idx2[i] : 0;
    int2 p = idxp[c];                     // one aligned 8-byte load
    int my = p.x + (p.y & 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int rec = __shfl(my, base + k);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tab + 4 * (size_t)rec + q),
                                         (__attribute__((address_space(3))) void*)&s[w][k][0], 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float4 a = s[w][q][base + 0], b = s[w][q][base + 1], cc = s[w][q][base + 2], d = s[w][q][base + 3];
    if (i < n) out[i] = (sum4(a) + sum4(b)) + (sum4(cc) + sum4(d));
}

template <class F>
static float graph_us(F launch, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 20; ++i) launch(i);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, st)); CHECK(hipStreamSynchronize(st));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.0f / 200.0f;
}

int main() {
    hipStream_t st; CHECK(hipStreamCreate(&st));
    const int n = 4096 * 187;
    const size_t T = 15u << 20;                 // 15 M records x 64 B = 1 GB
    const int NC = 2400000;                     // "cells"
    float4* tab; int* idx; int* cells; int2* cellp; int* idx2; float* out;
    CHECK(hipMalloc(&tab, T * 64)); CHECK(hipMemset(tab, 0, T * 64));
    const int sets = 20;
    std::vector<int> h((size_t)n * sets), h2((size_t)n * sets), hc(NC + 1);
    std::vector<int2> hp(NC);
    srand(1);
    auto rnd = [] { return ((unsigned)rand() << 15) ^ (unsigned)rand(); };
    for (auto& v : h) v = (int)(rnd() % T);
    for (auto& v : h2) v = (int)(rnd() % NC);
    for (int c = 0; c <= NC; ++c) hc[c] = (int)((size_t)c * 6 % (T - 8));
    for (int c = 0; c < NC; ++c) hp[c] = int2{hc[c], hc[c + 1]};
    CHECK(hipMalloc(&idx, h.size() * 4)); CHECK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&idx2, h2.size() * 4)); CHECK(hipMemcpy(idx2, h2.data(), h2.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&cells, hc.size() * 4)); CHECK(hipMemcpy(cells, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&cellp, hp.size() * 8)); CHECK(hipMemcpy(cellp, hp.data(), hp.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, n * 4));
    const int blocks = (n + 191) / 192;
    // every launch of a graph uses its own index set: nothing is warm from the launch before
    printf("A per-lane 4 x dwordx4      %.2f us\n", graph_us([&](int i) { hipLaunchKernelGGL(k_A, dim3(blocks), dim3(192), 0, st, tab, idx + (size_t)i * n, out, n); }, st));
    printf("B per-lane 1 x dwordx4      %.2f us\n", graph_us([&](int i) { hipLaunchKernelGGL(k_B, dim3(blocks), dim3(192), 0, st, tab, idx + (size_t)i * n, out, n); }, st));
    printf("C quad-coop + LDS transpose %.2f us\n", graph_us([&](int i) { hipLaunchKernelGGL(k_C, dim3(blocks), dim3(192), 0, st, tab, idx + (size_t)i * n, out, n); }, st));
    printf("D quad-coop LDS-DMA         %.2f us\n", graph_us([&](int i) { hipLaunchKernelGGL(k_D, dim3(blocks), dim3(192), 0, st, tab, idx + (size_t)i * n, out, n); }, st));
    printf("E cell pair (2 loads) + A   %.2f us\n", graph_us([&](int i) { hipLaunchKernelGGL(k_E, dim3(blocks), dim3(192), 0, st, tab, cells, idx2 + (size_t)i * n, out, n); }, st));
    printf("F cell int2 + D             %.2f us\n", graph_us([&](int i) { hipLaunchKernelGGL(k_F, dim3(blocks), dim3(192), 0, st, tab, cellp, idx2 + (size_t)i * n, out, n); }, st));
    return 0;
}
