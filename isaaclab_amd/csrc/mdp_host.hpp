// Host-side helpers shared by the translation units of libmdp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/mdp_hip.h"

namespace mdp {

char* last_error_buf();  // defined in capi.hip (thread local)

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define MDP_CHECK_HIP(expr)                                                                          \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return mdp::fail(MDP_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                                    \
    } while (0)

#define MDP_REQUIRE(cond, ...)                                      \
    do {                                                            \
        if (!(cond)) return mdp::fail(MDP_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// hipLaunchKernelGGL leaves errors in the sticky slot; fetch them without synchronising.
#define MDP_LAUNCH_CHECK()                                                                       \
    do {                                                                                         \
        hipError_t _e = hipGetLastError();                                                       \
        if (_e != hipSuccess)                                                                    \
            return mdp::fail(MDP_ERR_RUNTIME, "kernel launch failed: %s (%s:%d)",                \
                             hipGetErrorString(_e), __FILE__, __LINE__);                         \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// grid size for an elementwise kernel: one thread per element up to `cap` blocks.  The default cap (2^20 blocks) is only
// a guard for grid-stride kernels; kernels WITHOUT a grid stride (one thread per element, early return) must be
// launched with grids that cover their work -- 2^20 blocks x 128 threads covers 134 M elements, far beyond any per-env
// launch here; a smaller cap is passed explicitly where a kernel strides.
inline unsigned grid_for(int64_t work, int block, int64_t cap = (int64_t)1 << 20) {
    int64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

}  // namespace mdp
