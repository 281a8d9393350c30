// Duration of a near-empty kernel vs. grid size and block size (hipGraph of 20 launches, HIP events): calibrates the
// workgroup dispatch cost that bounds short kernels on MI355X.   hipcc --offload-arch=gfx950 -O3 dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_touch(const float* __restrict__ in, float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0f;
}

__global__ void k_touch_barrier(const float* __restrict__ in, float* __restrict__ out, int n) {
    __shared__ float s[64];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? in[i] : 0.0f;
    if (threadIdx.x < 64) s[threadIdx.x] = v;
    __syncthreads();
    if (i < n) out[i] = v + s[threadIdx.x & 63];
}

template <class F>
static float graph_us(F launch, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 20; ++i) launch();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.0f / 200.0f;
}

int main() {
    hipStream_t st; hipStreamCreate(&st);
    const int nmax = 65536 * 256;
    float *in, *out; hipMalloc(&in, nmax * 4); hipMalloc(&out, nmax * 4); hipMemset(in, 0, nmax * 4);
    for (int threads : {64, 192, 256, 1024}) {
        for (int blocks : {256, 1024, 2048, 4096, 8192, 16384, 65536}) {
            if ((long)blocks * threads > nmax) continue;
            int n = blocks * threads;
            float t0 = graph_us([&] { hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(threads), 0, st, in, out, n); }, st);
            float t1 = graph_us([&] { hipLaunchKernelGGL(k_touch_barrier, dim3(blocks), dim3(threads), 0, st, in, out, n); }, st);
            printf("threads %4d blocks %6d  touch %7.2f us  touch+barrier %7.2f us\n", threads, blocks, t0, t1);
        }
    }
    return 0;
}
