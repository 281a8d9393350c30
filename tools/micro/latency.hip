// What does ONE dependent memory round trip cost on this machine?  A single wave chases a random cyclic permutation
// through a table of a given size (one 4-byte index per 128-byte line), so every load misses whatever is smaller than the
// table: L2 (4 MiB per XCD), Infinity Cache (256 MiB), HBM.  Reported per table size: ns per load with the chip idle
// (1 block) and with the chip busy (the same chase in 2048 blocks of 192 threads, i.e. the residency of the cast).
// Also: block start -> first load answered for a block that was just dispatched (what a cast block's bookkeeping lane
// waits for), from the 100 MHz wall clock.
//   gpurun -- 'make -C tools/micro && tools/micro/bin/latency'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int LINE_WORDS = 32;   // 128-byte lines

// every lane of the block follows its own chain (lane-private start), `hops` dependent loads
__global__ void __launch_bounds__(192) k_chase(const int* __restrict__ tab, int lines, int hops, int* __restrict__ out,
                                               unsigned long long* __restrict__ first) {
    const unsigned long long t0 = wall_clock64();
    int at = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 7919u % (unsigned)lines);
    at = tab[(size_t)at * LINE_WORDS];
    if (first && threadIdx.x == 0 && at >= 0) first[blockIdx.x] = wall_clock64() - t0;
    for (int h = 1; h < hops; ++h) at = tab[(size_t)at * LINE_WORDS];
    if (at == -7) out[0] = at;
}

int main() {
    const size_t sizes_mb[] = {1, 2, 16, 64, 192, 1024, 4096};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int* out;
    CHECK(hipMalloc(&out, 64));
    unsigned long long* first;
    CHECK(hipMalloc(&first, 4096 * sizeof(unsigned long long)));
    std::mt19937 rng(5);
    printf("%10s %22s %22s %30s\n", "table", "idle: ns per hop", "busy: ns per hop", "busy: start->first answer ns");
    for (size_t mb : sizes_mb) {
        const int lines = (int)(mb * 1024 * 1024 / 128);
        std::vector<int> perm(lines), host((size_t)lines * LINE_WORDS, 0);
        std::iota(perm.begin(), perm.end(), 0);
        std::shuffle(perm.begin(), perm.end(), rng);
        for (int i = 0; i < lines; ++i) host[(size_t)perm[i] * LINE_WORDS] = perm[(i + 1) % lines];   // one cycle
        int* tab;
        CHECK(hipMalloc(&tab, host.size() * sizeof(int)));
        CHECK(hipMemcpy(tab, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice));
        auto run = [&](int blocks, int threads, int hops, unsigned long long* f) {
            hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(threads), 0, 0, tab, lines, hops, out, f);   // warm-up (clocks)
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(threads), 0, 0, tab, lines, hops, out, f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            return 1e6 * ms / hops;
        };
        const double idle = run(1, 64, 4000, nullptr);
        const double busy = run(2048, 192, 200, first);
        std::vector<unsigned long long> f(2048);
        CHECK(hipMemcpy(f.data(), first, 2048 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(f.begin(), f.end());
        printf("%7zu MiB %22.0f %22.0f %14llu med %10llu p90\n", mb, idle, busy, f[1024] * 10ull, f[1843] * 10ull);
        CHECK(hipFree(tab));
    }
    return 0;
}
