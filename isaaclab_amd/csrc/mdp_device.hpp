// Device-side building blocks shared by every kernel of libmdp_hip.so (gfx950 only).
//
// The arithmetic mirrors the expression trees of the reference torch code op by op (one rounding
// per torch op); the library is compiled with -ffp-contract=off so hipcc never fuses a*b+c.
// Citations: LAB = /root/reference/source/isaaclab/isaaclab.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MDP_DEV __device__ __forceinline__

namespace mdp {

struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };

MDP_DEV V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
MDP_DEV Q4 ld4(const float* p) { return Q4{p[0], p[1], p[2], p[3]}; }
MDP_DEV void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
MDP_DEV void st4(float* p, Q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
MDP_DEV float get(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
MDP_DEV float get(Q4 q, int k) { return k == 0 ? q.w : (k == 1 ? q.x : (k == 2 ? q.y : q.z)); }

// torch.clamp semantics: NaN in x propagates, min then max (max wins when lo > hi).
MDP_DEV float tclamp(float x, float lo, float hi) {
    if (x != x) return x;
    float r = x < lo ? lo : x;
    r = r > hi ? hi : r;
    return r;
}
MDP_DEV float tclamp_min(float x, float lo) { return (x != x) ? x : (x < lo ? lo : x); }
MDP_DEV float tclamp_max(float x, float hi) { return (x != x) ? x : (x > hi ? hi : x); }
// torch.max/min(a,b) elementwise: NaN propagates
MDP_DEV float tmax(float a, float b) { return (a != a || b != b) ? (a + b) : (a > b ? a : b); }
MDP_DEV float tmin(float a, float b) { return (a != a || b != b) ? (a + b) : (a < b ? a : b); }

// torch.cross (dim=-1)
MDP_DEV V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
MDP_DEV float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// torch.norm(p=2, dim=-1) over 3 / 4 elements
MDP_DEV float norm3(V3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }
MDP_DEV float norm2(float a, float b) { return sqrtf(a * a + b * b); }
MDP_DEV float norm4(Q4 q) { return sqrtf(((q.w * q.w + q.x * q.x) + q.y * q.y) + q.z * q.z); }

// math.py:582-602 quat_rotate / :605-625 quat_rotate_inverse:  a (+/-) b + c
MDP_DEV V3 quat_rotate_impl(Q4 q, V3 v, bool inverse) {
    float s = 2.0f * (q.w * q.w) - 1.0f;
    V3 a{v.x * s, v.y * s, v.z * s};
    V3 qv{q.x, q.y, q.z};
    V3 cr = cross(qv, v);
    V3 b{cr.x * q.w * 2.0f, cr.y * q.w * 2.0f, cr.z * q.w * 2.0f};
    float d = dot3(qv, v);
    V3 c{qv.x * d * 2.0f, qv.y * d * 2.0f, qv.z * d * 2.0f};
    if (inverse) return V3{(a.x - b.x) + c.x, (a.y - b.y) + c.y, (a.z - b.z) + c.z};
    return V3{(a.x + b.x) + c.x, (a.y + b.y) + c.y, (a.z + b.z) + c.z};
}
MDP_DEV V3 quat_rotate(Q4 q, V3 v) { return quat_rotate_impl(q, v, false); }
MDP_DEV V3 quat_rotate_inverse(Q4 q, V3 v) { return quat_rotate_impl(q, v, true); }

// math.py:545-564 quat_apply: t = 2 * (xyz x v); v + w*t + xyz x t
MDP_DEV V3 quat_apply(Q4 q, V3 v) {
    V3 xyz{q.x, q.y, q.z};
    V3 t = cross(xyz, v);
    t = V3{t.x * 2.0f, t.y * 2.0f, t.z * 2.0f};
    V3 c2 = cross(xyz, t);
    return V3{(v.x + q.w * t.x) + c2.x, (v.y + q.w * t.y) + c2.y, (v.z + q.w * t.z) + c2.z};
}

// math.py:81-92 normalize (eps clamp on the norm)
MDP_DEV Q4 normalize4(Q4 q, float eps) {
    float n = tclamp_min(norm4(q), eps);
    return Q4{q.w / n, q.x / n, q.y / n, q.z / n};
}
MDP_DEV V3 normalize3(V3 v, float eps) {
    float n = tclamp_min(norm3(v), eps);
    return V3{v.x / n, v.y / n, v.z / n};
}

// math.py:521-542 yaw_quat
MDP_DEV Q4 yaw_quat(Q4 q) {
    float yaw = atan2f(2.0f * (q.w * q.z + q.x * q.y), 1.0f - 2.0f * (q.y * q.y + q.z * q.z));
    Q4 r{cosf(yaw / 2.0f), 0.0f, 0.0f, sinf(yaw / 2.0f)};
    return normalize4(r, 1e-9f);
}

// The same quaternion without the three libm calls, for the one caller where they sit on a launch's critical path (the
// per-env part of the height-scan cast: one lane computes it while the env's 187 ray lanes wait at a barrier, ~1.2 us of
// a 12 us block life).  With (s, c) the reference's atan2 arguments and (sn, cn) = (s, c) / hypot(s, c) = (sin yaw, cos yaw):
//   cn >= 0:  cos(yaw/2) = sqrt((1 + cn) / 2),  sin(yaw/2) = sn / (2 cos(yaw/2))
//   cn <  0:  |sin(yaw/2)| = sqrt((1 - cn) / 2) with the sign of s (atan2's branch: -0 -> -pi),  cos(yaw/2) = sn / (2 sin(yaw/2))
// -- no cancellation in either branch.  Evaluated in fp64 from the fp32 (s, c) the reference forms and rounded once: the
// result is within half an fp32 ulp of the exact value for those arguments, i.e. closer to it than the reference's own
// atan2 -> cos / sin chain (each step ~1 ulp), and within ~2e-7 of what torch returns.
MDP_DEV Q4 yaw_quat_halfangle(Q4 q) {
    const float s = 2.0f * (q.w * q.z + q.x * q.y), c = 1.0f - 2.0f * (q.y * q.y + q.z * q.z);
    const double sd = (double)s, cd = (double)c;
    const double r = sqrt(sd * sd + cd * cd);
    if (!(r > 0.0)) return Q4{1.0f, 0.0f, 0.0f, 0.0f};       // atan2(0, 0) = 0
    const double sn = sd / r, cn = cd / r;
    double ch, sh;
    if (cn >= 0.0) {
        ch = sqrt(0.5 * (1.0 + cn));
        sh = sn / (2.0 * ch);
    } else {
        sh = copysign(sqrt(0.5 * (1.0 - cn)), sd);
        ch = sn / (2.0 * sh);
    }
    return normalize4(Q4{(float)ch, 0.0f, 0.0f, (float)sh}, 1e-9f);
}

// math.py:223-235
MDP_DEV Q4 quat_conjugate(Q4 q) { return Q4{q.w, -q.x, -q.y, -q.z}; }

// math.py:463-499 (8-multiply form, expression tree preserved)
MDP_DEV Q4 quat_mul(Q4 a, Q4 b) {
    float ww = (a.z + a.x) * (b.x + b.y);
    float yy = (a.w - a.y) * (b.w + b.z);
    float zz = (a.w + a.y) * (b.w - b.z);
    float xx = (ww + yy) + zz;
    float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
    float w = (qq - ww) + (a.z - a.y) * (b.y - b.z);
    float x = (qq - xx) + (a.x + a.w) * (b.x + b.w);
    float y = (qq - yy) + (a.w - a.x) * (b.y + b.z);
    float z = (qq - zz) + (a.z + a.y) * (b.w - b.x);
    return Q4{w, x, y, z};
}

// python-style float modulo (torch `%` / torch.remainder): result has the sign of the divisor
MDP_DEV float pymod(float a, float b) {
    float r = fmodf(a, b);
    if (r != 0.0f && ((r < 0.0f) != (b < 0.0f))) r += b;
    return r;
}

// math.py:95-117 wrap_to_pi
MDP_DEV float wrap_to_pi(float angle) {
    const float pi = 3.14159265358979323846f;
    float wrapped = pymod(angle + pi, 2.0f * pi);
    return (wrapped == 0.0f && angle > 0.0f) ? pi : (wrapped - pi);
}

// math.py:645-674 axis_angle_from_quat
MDP_DEV V3 axis_angle_from_quat(Q4 q, float eps) {
    float sgn = 1.0f - 2.0f * (q.w < 0.0f ? 1.0f : 0.0f);
    q = Q4{q.w * sgn, q.x * sgn, q.y * sgn, q.z * sgn};
    float mag = norm3(V3{q.x, q.y, q.z});
    float half = atan2f(mag, q.w);
    float angle = 2.0f * half;
    float s = fabsf(angle) > eps ? (sinf(half) / angle) : (0.5f - angle * angle / 48.0f);
    return V3{q.x / s, q.y / s, q.z / s};
}

// math.py:628-642 quat_from_angle_axis
MDP_DEV Q4 quat_from_angle_axis(float angle, V3 axis) {
    float theta = angle / 2.0f;
    V3 n = normalize3(axis, 1e-9f);
    float s = sinf(theta);
    Q4 q{cosf(theta), n.x * s, n.y * s, n.z * s};
    return normalize4(q, 1e-9f);
}

// ---------------------------------------------------------------------------------------------
// Counter-based generator: Philox4x32-10 (Salmon et al. 2011).  Keyed by (seed), counter
// (epoch, stream id, element index).  Stateless -> results are independent of launch geometry
// and of how environments are sharded over GPUs.
// ---------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
MDP_DEV U4 philox4x32_10(U4 ctr, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = U4{hi1 ^ ctr.y ^ k0, lo1, hi0 ^ ctr.w ^ k1, lo0};
        k0 += W0;
        k1 += W1;
    }
    return ctr;
}
// uniform in [0,1) with 24 random bits (same granularity as torch.rand for fp32)
MDP_DEV float u01(uint32_t bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }

MDP_DEV U4 rng4(uint64_t seed, uint64_t epoch, uint32_t stream, uint64_t index) {
    U4 ctr{(uint32_t)index, (uint32_t)(index >> 32), (uint32_t)epoch, stream ^ (uint32_t)(epoch >> 32) * 0x9E3779B9u};
    return philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
}
// Per-element observation noise draws one number for every output element of every step (and one per ray in the
// height scan): Philox's 40 quarter-rate integer multiplies per draw were a fifth of the cast kernel's vector time.
// These draws use a two-round multiply-xorshift finaliser (constants from the hash-prospector search, avalanche bias
// ~0.1) over (element index ^ key), the key being the same finaliser chained over seed, stream id and epoch -- still
// counter-based and stateless, ~1/8 of the instructions.  The command sampler (rare, few numbers) stays on Philox.
MDP_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
MDP_DEV uint32_t rng_key(uint64_t seed, uint64_t epoch, uint32_t stream) {   // uniform over a launch
    uint32_t k = mix32((uint32_t)seed ^ 0x9E3779B9u);
    k = mix32(k ^ (uint32_t)(seed >> 32));
    k = mix32(k ^ stream);
    k = mix32(k ^ (uint32_t)epoch);
    k = mix32(k ^ (uint32_t)(epoch >> 32));
    return k;
}
MDP_DEV uint32_t rng_bits(uint32_t key, uint64_t index, uint32_t salt) {
    uint32_t x = mix32((uint32_t)index ^ key);
    return mix32(x ^ ((uint32_t)(index >> 32) * 0x85EBCA6Bu + salt));
}
MDP_DEV float rng_uniform(uint64_t seed, uint64_t epoch, uint32_t stream, uint64_t index) {
    return u01(rng_bits(rng_key(seed, epoch, stream), index, 0x68E31DA4u));
}
MDP_DEV float rng_normal(uint64_t seed, uint64_t epoch, uint32_t stream, uint64_t index) {
    uint32_t key = rng_key(seed, epoch, stream);
    float u1 = 1.0f - u01(rng_bits(key, index, 0x68E31DA4u));  // (0,1]
    float u2 = u01(rng_bits(key, index, 0xB5297A4Du));
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

}  // namespace mdp
