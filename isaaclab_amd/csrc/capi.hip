// Library-level entry points: version, error string, device probe.
#include "mdp_host.hpp"

namespace mdp {
char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace mdp

extern "C" int mdp_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char* mdp_last_error(void) { return mdp::last_error_buf(); }
extern "C" int mdp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
