// Diagnostic in-kernel wall-clock stamps (compiled only with -DMDP_STAMPS, see Makefile STAMPS=1 and
// tools/kernel_stamps.py).  Each block records up to 8 stamps of the 100 MHz constant clock in row blockIdx.x of a
// buffer nobody else reads; the shipped library contains none of this.
#pragma once
#include <hip/hip_runtime.h>

#ifdef MDP_STAMPS
namespace mdp {
static __device__ unsigned long long* g_stamps = nullptr;   // [blocks][8], one copy per translation unit
__device__ __forceinline__ void stamp(int slot, unsigned long long tag = 0) {
    if (g_stamps && threadIdx.x == 0) {
        g_stamps[(size_t)blockIdx.x * 8 + slot] = wall_clock64();
        // tag | HW_ID (wave / simd / cu / sh / se ids) << 8 | XCC_ID << 40: where the block ran
        if (slot == 0)
            g_stamps[(size_t)blockIdx.x * 8 + 7] =
                tag | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8) |
                ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) << 40);
    }
}
}  // namespace mdp
#define MDP_STAMP(slot, ...) ::mdp::stamp(slot, ##__VA_ARGS__)
#define MDP_STAMP_SETTER(fn)                                                                          \
    extern "C" int fn(void* buffer) {                                                                 \
        unsigned long long* p = (unsigned long long*)buffer;                                          \
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(::mdp::g_stamps), &p, sizeof(p));                    \
    }
#else
#define MDP_STAMP(slot, ...) ((void)0)
#define MDP_STAMP_SETTER(fn)
#endif
