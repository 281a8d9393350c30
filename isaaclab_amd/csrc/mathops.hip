// isaaclab.utils.math as batched row kernels (one thread per row).
// Reference: LAB/utils/math.py (line numbers next to each op in include/mdp_hip.h).
#include "mdp_device.hpp"
#include "mdp_host.hpp"

using namespace mdp;

namespace {

struct MathArgs {
    const float* a; int lda;
    const float* b; int ldb;
    const float* c; int ldc;
    const float* d; int ldd;
    float* out; int ldo;
    int i0; float f0;
};

MDP_DEV void quat_from_matrix_row(const float* m, float* o) {
    // math.py:294-349
    float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    float qa[4] = {((1.0f + m00) + m11) + m22, ((1.0f + m00) - m11) - m22, ((1.0f - m00) + m11) - m22,
                   ((1.0f - m00) - m11) + m22};
    for (int i = 0; i < 4; ++i) qa[i] = qa[i] > 0.0f ? sqrtf(qa[i]) : 0.0f;
    float cand[4][4] = {
        {qa[0] * qa[0], m21 - m12, m02 - m20, m10 - m01},
        {m21 - m12, qa[1] * qa[1], m10 + m01, m02 + m20},
        {m02 - m20, m10 + m01, qa[2] * qa[2], m12 + m21},
        {m10 - m01, m20 + m02, m21 + m12, qa[3] * qa[3]},
    };
    int best = 0;  // argmax, first occurrence on ties (torch.argmax)
    for (int i = 1; i < 4; ++i)
        if (qa[i] > qa[best]) best = i;
    float den = 2.0f * tmax(qa[best], 0.1f);
    for (int k = 0; k < 4; ++k) o[k] = cand[best][k] / den;
}

template <int OP>
__global__ void k_math(int64_t n, MathArgs g) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float* a = g.a ? g.a + i * g.lda : nullptr;
        const float* b = g.b ? g.b + i * g.ldb : nullptr;
        const float* c = g.c ? g.c + i * g.ldc : nullptr;
        const float* d = g.d ? g.d + i * g.ldd : nullptr;
        float* o = g.out + i * g.ldo;
        if (OP == MDP_MATH_QUAT_MUL) st4(o, quat_mul(ld4(a), ld4(b)));
        else if (OP == MDP_MATH_QUAT_CONJUGATE) st4(o, quat_conjugate(ld4(a)));
        else if (OP == MDP_MATH_QUAT_INV) st4(o, normalize4(quat_conjugate(ld4(a)), 1e-9f));
        else if (OP == MDP_MATH_QUAT_UNIQUE) { Q4 q = ld4(a); if (q.w < 0.0f) q = Q4{-q.w, -q.x, -q.y, -q.z}; st4(o, q); }
        else if (OP == MDP_MATH_QUAT_ROTATE) st3(o, quat_rotate(ld4(a), ld3(b)));
        else if (OP == MDP_MATH_QUAT_ROTATE_INVERSE) st3(o, quat_rotate_inverse(ld4(a), ld3(b)));
        else if (OP == MDP_MATH_QUAT_APPLY) st3(o, quat_apply(ld4(a), ld3(b)));
        else if (OP == MDP_MATH_QUAT_APPLY_YAW) st3(o, quat_apply(yaw_quat(ld4(a)), ld3(b)));
        else if (OP == MDP_MATH_YAW_QUAT) st4(o, yaw_quat(ld4(a)));
        else if (OP == MDP_MATH_QUAT_FROM_EULER_XYZ) {
            // math.py:251-278
            float roll = a[0], pitch = a[1], yaw = a[2];
            float cy = cosf(yaw * 0.5f), sy = sinf(yaw * 0.5f);
            float cr = cosf(roll * 0.5f), sr = sinf(roll * 0.5f);
            float cp = cosf(pitch * 0.5f), sp = sinf(pitch * 0.5f);
            o[0] = cy * cr * cp + sy * sr * sp;
            o[1] = cy * sr * cp - sy * cr * sp;
            o[2] = cy * cr * sp + sy * sr * cp;
            o[3] = sy * cr * cp - cy * sr * sp;
        } else if (OP == MDP_MATH_EULER_XYZ_FROM_QUAT) {
            // math.py:413-444 (angles returned in [0, 2pi))
            const float pi = 3.14159265358979323846f;
            Q4 q = ld4(a);
            float sin_roll = 2.0f * (q.w * q.x + q.y * q.z);
            float cos_roll = 1.0f - 2.0f * (q.x * q.x + q.y * q.y);
            float roll = atan2f(sin_roll, cos_roll);
            float sin_pitch = 2.0f * (q.w * q.y - q.z * q.x);
            float sgn = sin_pitch > 0.0f ? 1.0f : (sin_pitch < 0.0f ? -1.0f : 0.0f);
            float pitch = fabsf(sin_pitch) >= 1.0f ? fabsf(pi / 2.0f) * sgn : asinf(sin_pitch);
            float sin_yaw = 2.0f * (q.w * q.z + q.x * q.y);
            float cos_yaw = 1.0f - 2.0f * (q.y * q.y + q.z * q.z);
            float yaw = atan2f(sin_yaw, cos_yaw);
            o[0] = pymod(roll, 2.0f * pi);
            o[1] = pymod(pitch, 2.0f * pi);
            o[2] = pymod(yaw, 2.0f * pi);
        } else if (OP == MDP_MATH_AXIS_ANGLE_FROM_QUAT) st3(o, axis_angle_from_quat(ld4(a), g.f0));
        else if (OP == MDP_MATH_QUAT_ERROR_MAGNITUDE) {
            Q4 qd = quat_mul(ld4(a), quat_conjugate(ld4(b)));
            o[0] = norm3(axis_angle_from_quat(qd, 1.0e-6f));
        } else if (OP == MDP_MATH_QUAT_BOX_MINUS) {
            // math.py:502-518
            Q4 qd = quat_mul(ld4(a), quat_conjugate(ld4(b)));
            float re = qd.w;
            V3 im{qd.x, qd.y, qd.z};
            float nim = norm3(im);
            float sgn = re > 0.0f ? 1.0f : (re < 0.0f ? -1.0f : 0.0f);
            float scale = 2.0f * (nim > 1.0e-7f ? atan2f(nim, re) / nim : sgn);
            st3(o, V3{scale * im.x, scale * im.y, scale * im.z});
        } else if (OP == MDP_MATH_QUAT_FROM_ANGLE_AXIS) st4(o, quat_from_angle_axis(a[0], ld3(b)));
        else if (OP == MDP_MATH_MATRIX_FROM_QUAT) {
            // math.py:143-174
            float r = a[0], x = a[1], y = a[2], z = a[3];
            float two_s = 2.0f / (((r * r + x * x) + y * y) + z * z);
            o[0] = 1.0f - two_s * (y * y + z * z);
            o[1] = two_s * (x * y - z * r);
            o[2] = two_s * (x * z + y * r);
            o[3] = two_s * (x * y + z * r);
            o[4] = 1.0f - two_s * (x * x + z * z);
            o[5] = two_s * (y * z - x * r);
            o[6] = two_s * (x * z - y * r);
            o[7] = two_s * (y * z + x * r);
            o[8] = 1.0f - two_s * (x * x + y * y);
        } else if (OP == MDP_MATH_QUAT_FROM_MATRIX) quat_from_matrix_row(a, o);
        else if (OP == MDP_MATH_WRAP_TO_PI) o[0] = wrap_to_pi(a[0]);
        else if (OP == MDP_MATH_NORMALIZE) {
            int w = g.i0;
            float s = 0.0f;
            for (int k = 0; k < w; ++k) s += a[k] * a[k];
            float nrm = tclamp_min(sqrtf(s), g.f0);
            for (int k = 0; k < w; ++k) o[k] = a[k] / nrm;
        } else if (OP == MDP_MATH_SCALE_TRANSFORM) {
            float offset = (b[0] + c[0]) * 0.5f;
            o[0] = 2.0f * (a[0] - offset) / (c[0] - b[0]);
        } else if (OP == MDP_MATH_UNSCALE_TRANSFORM) {
            float offset = (b[0] + c[0]) * 0.5f;
            o[0] = a[0] * (c[0] - b[0]) * 0.5f + offset;
        } else if (OP == MDP_MATH_SATURATE) o[0] = tmax(tmin(a[0], c[0]), b[0]);
        else if (OP == MDP_MATH_COMBINE_FRAME) {
            // math.py:749-781
            Q4 q01 = ld4(b);
            V3 t01 = ld3(a);
            V3 t02 = t01;
            if (c) { V3 r = quat_apply(q01, ld3(c)); t02 = V3{t01.x + r.x, t01.y + r.y, t01.z + r.z}; }
            Q4 q02 = d ? quat_mul(q01, ld4(d)) : q01;
            st3(o, t02);
            st4(o + 3, q02);
        } else if (OP == MDP_MATH_SUBTRACT_FRAME) {
            // math.py:785-816
            Q4 q10 = normalize4(quat_conjugate(ld4(b)), 1e-9f);
            V3 t01 = ld3(a);
            V3 diff = c ? V3{c[0] - t01.x, c[1] - t01.y, c[2] - t01.z} : V3{-t01.x, -t01.y, -t01.z};
            st3(o, quat_apply(q10, diff));
            st4(o + 3, d ? quat_mul(q10, ld4(d)) : q10);
        } else if (OP == MDP_MATH_COMPUTE_POSE_ERROR) {
            // math.py:820-869
            Q4 q01 = ld4(b), q02 = ld4(d);
            Q4 cj = quat_conjugate(q01);
            float nrm = quat_mul(q01, cj).w;
            Q4 inv{cj.w / nrm, cj.x / nrm, cj.y / nrm, cj.z / nrm};
            Q4 qe = quat_mul(q02, inv);
            o[0] = c[0] - a[0]; o[1] = c[1] - a[1]; o[2] = c[2] - a[2];
            if (g.i0 == 1) st4(o + 3, qe);
            else st3(o + 3, axis_angle_from_quat(qe, 1.0e-6f));
        } else if (OP == MDP_MATH_APPLY_DELTA_POSE) {
            // math.py:872-910
            V3 rot{c[3], c[4], c[5]};
            float angle = norm3(rot);
            V3 axis{rot.x / angle, rot.y / angle, rot.z / angle};
            Q4 dq = angle > g.f0 ? quat_from_angle_axis(angle, axis) : Q4{1.0f, 0.0f, 0.0f, 0.0f};
            o[0] = a[0] + c[0]; o[1] = a[1] + c[1]; o[2] = a[2] + c[2];
            st4(o + 3, quat_mul(dq, ld4(b)));
        } else if (OP == MDP_MATH_SKEW) {
            o[0] = 0.0f; o[1] = -a[2]; o[2] = a[1];
            o[3] = a[2]; o[4] = 0.0f; o[5] = -a[0];
            o[6] = -a[1]; o[7] = a[0]; o[8] = 0.0f;
        } else if (OP == MDP_MATH_COPYSIGN) {
            float s = a[0] > 0.0f ? 1.0f : (a[0] < 0.0f ? -1.0f : 0.0f);
            o[0] = fabsf(g.f0) * s;
        }
    }
}

// ---- samplers (math.py:1233-1421) on the counter-based generator -------------------------------------------------
// One thread per sample.  The two uniforms of a sample come from one Philox4x32-10 block keyed by (seed, epoch, kind,
// sample index) -- or from `inject` (2 floats per sample; 4 standard normals per sample for MDP_SAMPLE_QUAT) so the
// reference's own transformation of a recorded torch.rand draw can be compared number for number.
struct SampleArgs {
    const float* a; const float* b;   // per-sample lower / upper (mean / std) or null -> the scalars
    float as, bs, cs;                  // scalars: lower, upper (mean, std; cylinder: radius, h_min, h_max)
    const float* inject;
    float* out;
    const unsigned long long* state;   // {seed, epoch}
};

template <int KIND>
__global__ void k_sample(int64_t n, SampleArgs g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float u1, u2, z[4];
    if (g.inject) {
        if (KIND == MDP_SAMPLE_QUAT) { for (int k = 0; k < 4; ++k) z[k] = g.inject[4 * i + k]; u1 = u2 = 0.0f; }
        else { u1 = g.inject[2 * i]; u2 = g.inject[2 * i + 1]; }
    } else {
        const U4 r = rng4(g.state[0], g.state[1], 0x53414d50u + (uint32_t)KIND, (uint64_t)i);
        u1 = u01(r.x); u2 = u01(r.y);
        if (KIND == MDP_SAMPLE_QUAT) {     // two Box-Muller pairs
            const float r1 = sqrtf(-2.0f * logf(1.0f - u1)), r2 = sqrtf(-2.0f * logf(1.0f - u01(r.z)));
            const float t1 = 6.28318530717958647692f * u2, t2 = 6.28318530717958647692f * u01(r.w);
            z[0] = r1 * cosf(t1); z[1] = r1 * sinf(t1); z[2] = r2 * cosf(t2); z[3] = r2 * sinf(t2);
        }
    }
    const float lo = g.a ? g.a[i] : g.as, hi = g.b ? g.b[i] : g.bs;
    if (KIND == MDP_SAMPLE_UNIFORM) {
        g.out[i] = u1 * (hi - lo) + lo;                                  // :1331
    } else if (KIND == MDP_SAMPLE_LOG_UNIFORM) {
        const float ll = logf(lo), lh = logf(hi);                        // :1362
        g.out[i] = expf(u1 * (lh - ll) + ll);
    } else if (KIND == MDP_SAMPLE_GAUSSIAN) {
        // torch.normal(mean, std): mean + std * z, z from Box-Muller over u1 in (0, 1]
        const float zz = g.inject ? u1 : sqrtf(-2.0f * logf(1.0f - u1)) * cosf(6.28318530717958647692f * u2);
        g.out[i] = lo + hi * zz;
    } else if (KIND == MDP_SAMPLE_TRIANGLE) {
        float r = 2.0f * u1 - 1.0f;                                      // :1304-1310
        r = r < 0.0f ? -sqrtf(-r) : sqrtf(r);
        r = (r + 1.0f) / 2.0f;
        g.out[i] = (hi - lo) * r + lo;
    } else if (KIND == MDP_SAMPLE_CYLINDER) {
        const float ang = (u1 * 2.0f - 1.0f) * 3.14159265358979323846f;  // :1408-1419 (as = radius, bs..cs = h range)
        g.out[3 * i + 0] = g.as * cosf(ang);
        g.out[3 * i + 1] = g.as * sinf(ang);
        g.out[3 * i + 2] = u2 * (g.cs - g.bs) + g.bs;
    } else if (KIND == MDP_SAMPLE_YAW_QUAT) {
        const float yaw = 2.0f * 3.14159265358979323846f * u1;           // :1281-1285
        // quat_from_euler_xyz(0, 0, yaw) (math.py:251-278) with the roll / pitch factors evaluated: cos 0 = 1, sin 0 = 0
        const float cy = cosf(yaw * 0.5f), sy = sinf(yaw * 0.5f);
        const float cr = 1.0f, sr = 0.0f, cp = 1.0f, sp = 0.0f;
        st4(g.out + 4 * i, Q4{cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp,
                              sy * cr * cp - cy * sr * sp});
    } else if (KIND == MDP_SAMPLE_QUAT) {
        const float nrm = sqrtf(((z[0] * z[0] + z[1] * z[1]) + z[2] * z[2]) + z[3] * z[3]);   // :1265-1267
        const float d = tmax(nrm, 1e-12f);
        for (int k = 0; k < 4; ++k) g.out[4 * i + k] = z[k] / d;
    }
}

__global__ void k_sample_advance(unsigned long long* state) { state[1] += 1ull; }

}  // namespace

extern "C" int mdp_math_op(int op, int64_t n, const float* a, int lda, const float* b, int ldb, const float* c,
                           int ldc, const float* d, int ldd, float* out, int ldo, int i0, float f0, void* stream) {
    MDP_REQUIRE(op >= 0 && op < MDP_MATH_NUM_OPS, "unknown math op %d", op);
    MDP_REQUIRE(n >= 0, "negative row count %lld", (long long)n);
    if (n == 0) return MDP_OK;
    MDP_REQUIRE(a && out, "null pointer");
    MathArgs g{a, lda, b, ldb, c, ldc, d, ldd, out, ldo, i0, f0};
    dim3 grid(grid_for(n, 256)), block(256);
    hipStream_t s = as_stream(stream);
#define CASE(OP) case OP: hipLaunchKernelGGL(k_math<OP>, grid, block, 0, s, n, g); break;
    switch (op) {
        CASE(MDP_MATH_QUAT_MUL) CASE(MDP_MATH_QUAT_CONJUGATE) CASE(MDP_MATH_QUAT_INV) CASE(MDP_MATH_QUAT_UNIQUE)
        CASE(MDP_MATH_QUAT_ROTATE) CASE(MDP_MATH_QUAT_ROTATE_INVERSE) CASE(MDP_MATH_QUAT_APPLY)
        CASE(MDP_MATH_QUAT_APPLY_YAW) CASE(MDP_MATH_YAW_QUAT) CASE(MDP_MATH_QUAT_FROM_EULER_XYZ)
        CASE(MDP_MATH_EULER_XYZ_FROM_QUAT) CASE(MDP_MATH_AXIS_ANGLE_FROM_QUAT) CASE(MDP_MATH_QUAT_ERROR_MAGNITUDE)
        CASE(MDP_MATH_QUAT_BOX_MINUS) CASE(MDP_MATH_QUAT_FROM_ANGLE_AXIS) CASE(MDP_MATH_MATRIX_FROM_QUAT)
        CASE(MDP_MATH_QUAT_FROM_MATRIX) CASE(MDP_MATH_WRAP_TO_PI) CASE(MDP_MATH_NORMALIZE)
        CASE(MDP_MATH_SCALE_TRANSFORM) CASE(MDP_MATH_UNSCALE_TRANSFORM) CASE(MDP_MATH_SATURATE)
        CASE(MDP_MATH_COMBINE_FRAME) CASE(MDP_MATH_SUBTRACT_FRAME) CASE(MDP_MATH_COMPUTE_POSE_ERROR)
        CASE(MDP_MATH_APPLY_DELTA_POSE) CASE(MDP_MATH_SKEW) CASE(MDP_MATH_COPYSIGN)
    }
#undef CASE
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_sample(int kind, int64_t n, const float* a, const float* b, float a_scalar, float b_scalar,
                          float c_scalar, const float* inject, float* out, uint64_t* state, void* stream) {
    MDP_REQUIRE(kind >= 0 && kind < MDP_SAMPLE_NUM, "unknown sampler %d", kind);
    MDP_REQUIRE(n >= 0, "negative sample count %lld", (long long)n);
    if (n == 0) return MDP_OK;
    MDP_REQUIRE(out, "null output");
    MDP_REQUIRE(inject || state, "a sampler needs its generator state {seed, epoch} (or injected numbers)");
    MDP_REQUIRE(n <= ((int64_t)1 << 31) * 256, "sample count %lld beyond one launch", (long long)n);
    SampleArgs g{a, b, a_scalar, b_scalar, c_scalar, inject, out, (const unsigned long long*)state};
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t s = as_stream(stream);
#define CASE(K) case K: hipLaunchKernelGGL(k_sample<K>, grid, block, 0, s, n, g); break;
    switch (kind) {
        CASE(MDP_SAMPLE_UNIFORM) CASE(MDP_SAMPLE_LOG_UNIFORM) CASE(MDP_SAMPLE_GAUSSIAN) CASE(MDP_SAMPLE_TRIANGLE)
        CASE(MDP_SAMPLE_CYLINDER) CASE(MDP_SAMPLE_YAW_QUAT) CASE(MDP_SAMPLE_QUAT)
    }
#undef CASE
    // the epoch lives on the device and moves in stream order: a captured call draws fresh numbers on every replay
    if (!inject) hipLaunchKernelGGL(k_sample_advance, dim3(1), dim3(1), 0, s, (unsigned long long*)state);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}
