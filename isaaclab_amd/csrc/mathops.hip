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
