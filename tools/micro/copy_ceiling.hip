// Measured HBM ceiling of one MI355X for the roofline of SURVEY section 8(d): a grid-stride float4 copy, read-only sum
// and write-only fill over buffers far larger than the 256 MB Infinity Cache.  GB/s counts bytes read + bytes written.
// hipcc --offload-arch=gfx950 -O3 tools/micro/copy_ceiling.hip -o tools/micro/bin/copy_ceiling && tools/micro/bin/copy_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
    float s = 0.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    if (s == 12345.678f) out[0] = s;   // never true for the zero-filled buffer: keeps the loads alive
}
__global__ void __launch_bounds__(256) k_fill(float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = float4{1.0f, 2.0f, 3.0f, 4.0f};
}

template <class F>
static float time_us(F launch, hipStream_t st, int reps) {
    launch(); CHECK(hipStreamSynchronize(st));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / reps;
}

int main() {
    hipStream_t st; CHECK(hipStreamCreate(&st));
    float* out; CHECK(hipMalloc(&out, 4));
    for (size_t mb : {64, 1024, 4096}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        float4 *a, *b;
        CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
        CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
        for (int blocks : {2048, 8192}) {
            float tc = time_us([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, a, b, n); }, st, 20);
            float tr = time_us([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, st, a, out, n); }, st, 20);
            float tw = time_us([&] { hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, st, b, n); }, st, 20);
            printf("%5zu MiB buffers, %5d blocks: copy %8.1f us = %6.0f GB/s (read + write) | read %8.1f us = %6.0f GB/s | fill %8.1f us = %6.0f GB/s\n",
                   mb, blocks, tc, 2.0 * bytes / tc * 1e-3, tr, bytes / tr * 1e-3, tw, bytes / tw * 1e-3);
        }
        CHECK(hipFree(a)); CHECK(hipFree(b));
    }
    return 0;
}
