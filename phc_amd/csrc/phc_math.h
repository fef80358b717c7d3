// phc_math.h -- fp32 quaternion / vector primitives shared by every kernel of the path.
//
// Restates, per lane, the elementwise quaternion library the reference runs as chains of
// torch ops: reference phc/utils/isaacgym_torch_utils.py (R12 in SURVEY.md section 8a).
// Convention: xyzw, w last.  Every function cites the reference lines it follows; operation
// ORDER is kept the same as the reference so that fp32 results stay within a few ulp of it.
//
// The header is PHC_HD (host + device) so that oracle/hostemu can compile the very same
// per-lane math with g++ and check it against the oracle on a machine without a GPU.
// That host build is test infrastructure only; the product always runs the HIP kernels.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PHC_HD __host__ __device__ __forceinline__
#define PHC_NOINLINE_HD __host__ __device__ __attribute__((noinline))   // rarely taken paths: kept out of the callers' register budget
#else
#define PHC_HD inline
#define PHC_NOINLINE_HD inline
struct float4 { float x, y, z, w; };  // host build only (oracle/hostemu)
#endif

namespace phc {

struct V3 { float x, y, z; };
struct Q4 { float x, y, z, w; };

PHC_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PHC_HD Q4 q4(float x, float y, float z, float w) { Q4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
PHC_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PHC_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PHC_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
PHC_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PHC_HD V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
PHC_HD V3& operator+=(V3& a, V3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
PHC_HD V3& operator-=(V3& a, V3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
PHC_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PHC_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PHC_HD float norm2(V3 a) { return dot(a, a); }
PHC_HD float norm(V3 a) { return sqrtf(dot(a, a)); }

// ---- elementary functions of the reference-library restatements below ----
// The reference evaluates these as torch CPU / CUDA ops (libm accuracy).  On the device the libm expansions are most of the task kernels'
// instruction stream (round 3 count: sinf ~50 executed / 120 static instructions with its huge-argument path, IEEE division 12, IEEE sqrt 18), and
// the kernels are bound by the length of one wavefront's stream (profiles/r03_stepper/README.md).  Device versions with <= 2 ulp error for the
// bounded arguments of this path (angles of a few turns at most, normalised quaternions); parity with the reference is pinned at 1e-5 absolute by
// tests/test_task_parity.py, five orders above their error.  INDEX arithmetic (frame_ref, sample_time_interval: bit-exact contract) does not
// use them.  Host build (oracle/hostemu): plain libm.
#if defined(__HIP_DEVICE_COMPILE__)
PHC_HD float t_rcp(float b) {   // 1 / b, one Newton step on v_rcp_f32
    float r = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
}
PHC_HD float t_div(float a, float b) {
    const float r = t_rcp(b);
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
PHC_HD float t_sqrt(float x) {   // v_rsq_f32 + one Heron step; sqrt(0) = 0, sqrt(negative) = NaN like the reference
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    s = __builtin_fmaf(__builtin_fmaf(-s, s, x) * 0.5f, y, s);
    return x == 0.0f ? 0.0f : s;
}
// sin and cos of one argument: Cody-Waite reduction to [-pi/4, pi/4] (three-part pi/2, exact products for |x| < ~1e4), Cephes sinf / cosf kernels
PHC_HD void t_sincos(float x, float* sn, float* cs) {
    const float k = rintf(x * 0.636619772f);
    float r = __builtin_fmaf(-k, 1.57073974609375f, x);
    r = __builtin_fmaf(-k, 5.657970905303955078125e-05f, r);
    r = __builtin_fmaf(-k, 9.920936294705029468e-10f, r);
    const float z = r * r;
    const float ps = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float pc = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                                    __builtin_fmaf(-0.5f, z, 1.0f));
    const int n = (int)k;
    const float s0 = (n & 1) ? pc : ps, c0 = (n & 1) ? ps : pc;
    *sn = (n & 2) ? -s0 : s0;
    *cs = ((n + 1) & 2) ? -c0 : c0;
}
PHC_HD float t_sin(float x) { float s, c; t_sincos(x, &s, &c); return s; }
// atan2(sin x, cos x): x brought into (-pi, pi] by subtracting the nearest multiple of 2 pi (identity for |x| < pi)
PHC_HD float t_normalize_angle(float x) {
    const float k = rintf(x * 0.159154937f);
    return __builtin_fmaf(k, 1.7484555e-7f, __builtin_fmaf(-k, 6.28318548f, x));
}
#else
PHC_HD float t_rcp(float b) { return 1.0f / b; }
PHC_HD float t_div(float a, float b) { return a / b; }
PHC_HD float t_sqrt(float x) { return sqrtf(x); }
PHC_HD void t_sincos(float x, float* sn, float* cs) { *sn = sinf(x); *cs = cosf(x); }
PHC_HD float t_sin(float x) { return sinf(x); }
PHC_HD float t_normalize_angle(float x) { return atan2f(sinf(x), cosf(x)); }
#endif
PHC_HD float t_norm(V3 a) { return t_sqrt(dot(a, a)); }

// isaacgym_torch_utils.py:25-45 -- the 8-multiplication form
PHC_HD Q4 quat_mul(Q4 a, Q4 b) {
    float ww = (a.z + a.x) * (b.x + b.y);
    float yy = (a.w - a.y) * (b.w + b.z);
    float zz = (a.w + a.y) * (b.w - b.z);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
    Q4 r;
    r.w = qq - ww + (a.z - a.y) * (b.y - b.z);
    r.x = qq - xx + (a.x + a.w) * (b.x + b.w);
    r.y = qq - yy + (a.w - a.x) * (b.y + b.z);
    r.z = qq - zz + (a.z + a.y) * (b.w - b.x);
    return r;
}

// isaacgym_torch_utils.py:90-93
PHC_HD Q4 quat_conjugate(Q4 a) { return q4(-a.x, -a.y, -a.z, a.w); }

// isaacgym_torch_utils.py:238-247 (my_quat_rotate == quat_rotate)
PHC_HD V3 quat_rotate(Q4 q, V3 v) {
    V3 qv = v3(q.x, q.y, q.z);
    float s = 2.0f * q.w * q.w - 1.0f;
    V3 a = v * s;
    V3 b = cross(qv, v) * q.w * 2.0f;
    V3 c = qv * dot(qv, v) * 2.0f;
    return a + b + c;
}

// isaacgym_torch_utils.py:110-111
PHC_HD float normalize_angle(float x) { return t_normalize_angle(x); }

// isaacgym_torch_utils.py:250-271 -- (angle, axis); min_theta 1e-5, default axis z
PHC_HD float quat_to_angle_axis(Q4 q, V3* axis) {
    const float min_theta = 1e-5f;
    float sin_theta = t_sqrt(1.0f - q.w * q.w);
    float angle = normalize_angle(2.0f * acosf(q.w));
    bool mask = fabsf(sin_theta) > min_theta;  // NaN (|w|>1) -> false, as torch.abs(nan) > x
    if (axis) { const float is = t_rcp(sin_theta); *axis = mask ? v3(q.x * is, q.y * is, q.z * is) : v3(0.f, 0.f, 1.f); }
    return mask ? angle : 0.0f;
}

// isaacgym_torch_utils.py:284-290
PHC_HD V3 quat_to_exp_map(Q4 q) {
    V3 axis;
    float angle = quat_to_angle_axis(q, &axis);
    return axis * angle;
}

// isaacgym_torch_utils.py:294-306 -- 6-D "tangent, normal" = q * x-axis, q * z-axis
PHC_HD void quat_to_tan_norm(Q4 q, float* out6) {
    V3 t = quat_rotate(q, v3(1.f, 0.f, 0.f));
    V3 n = quat_rotate(q, v3(0.f, 0.f, 1.f));
    out6[0] = t.x; out6[1] = t.y; out6[2] = t.z; out6[3] = n.x; out6[4] = n.y; out6[5] = n.z;
}

// isaacgym_torch_utils.py:49-50,97-106 -- normalises the axis, then the quaternion
PHC_HD Q4 quat_from_angle_axis(float angle, V3 axis) {
    float theta = angle / 2.0f;
    float an = fmaxf(t_norm(axis), 1e-9f);
    float s, c;
    t_sincos(theta, &s, &c);
    const float ia = t_rcp(an);
    Q4 q = q4(axis.x * ia * s, axis.y * ia * s, axis.z * ia * s, c);
    float qn = fmaxf(t_sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-9f);
    const float iq = t_rcp(qn);
    return q4(q.x * iq, q.y * iq, q.z * iq, q.w * iq);
}

// isaacgym_torch_utils.py:342-365
PHC_HD Q4 exp_map_to_quat(V3 e) {
    const float min_theta = 1e-5f;
    float angle = t_norm(e);
    const float ian = t_rcp(angle);     // (angle 0: inf * 0 = NaN axis, replaced by the mask below -- as the reference's 0 / 0)
    V3 axis = v3(e.x * ian, e.y * ian, e.z * ian);
    angle = normalize_angle(angle);
    bool mask = fabsf(angle) > min_theta;
    if (!mask) { angle = 0.f; axis = v3(0.f, 0.f, 1.f); }
    return quat_from_angle_axis(angle, axis);
}

// isaacgym_torch_utils.py:369-390 -- thresholds |sin| < 0.001 -> midpoint, |cos| >= 1 -> q0
PHC_HD Q4 slerp(Q4 q0, Q4 q1, float t) {
    float c = q0.x * q1.x + q0.y * q1.y + q0.z * q1.z + q0.w * q1.w;
    if (c < 0.f) { q1 = q4(-q1.x, -q1.y, -q1.z, -q1.w); }
    c = fabsf(c);
    float half_theta = acosf(c);
    float s = t_sqrt(1.0f - c * c);
    const float is = t_rcp(s);
    float ra = t_sin((1.0f - t) * half_theta) * is;
    float rb = t_sin(t * half_theta) * is;
    Q4 r = q4(ra * q0.x + rb * q1.x, ra * q0.y + rb * q1.y, ra * q0.z + rb * q1.z, ra * q0.w + rb * q1.w);
    if (fabsf(s) < 0.001f) r = q4(0.5f * q0.x + 0.5f * q1.x, 0.5f * q0.y + 0.5f * q1.y, 0.5f * q0.z + 0.5f * q1.z, 0.5f * q0.w + 0.5f * q1.w);
    if (fabsf(c) >= 1.f) r = q0;
    return r;
}

// isaacgym_torch_utils.py:394-405 -- heading = atan2 of the rotated x axis
PHC_HD float calc_heading(Q4 q) {
    V3 d = quat_rotate(q, v3(1.f, 0.f, 0.f));
    return atan2f(d.y, d.x);
}
// isaacgym_torch_utils.py:409-433
PHC_HD Q4 calc_heading_quat(Q4 q) { return quat_from_angle_axis(calc_heading(q), v3(0.f, 0.f, 1.f)); }
PHC_HD Q4 calc_heading_quat_inv(Q4 q) { return quat_from_angle_axis(-calc_heading(q), v3(0.f, 0.f, 1.f)); }

// ---- helpers that are not in the reference library (used by the stepper) ----
PHC_HD Q4 quat_normalize(Q4 q) {
    float n = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return q4(q.x * n, q.y * n, q.z * n, q.w * n);
}
// Hamilton product, 16-multiplication form (cheaper to schedule than the 8-mul form above)
PHC_HD Q4 quat_mul16(Q4 a, Q4 b) {
    return q4(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
              a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
struct M3 { float m[9]; };  // row-major
PHC_HD M3 quat_to_mat(Q4 q) {
    float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
    float xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z, wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    M3 r;
    r.m[0] = 1.f - 2.f * (yy + zz); r.m[1] = 2.f * (xy - wz);       r.m[2] = 2.f * (xz + wy);
    r.m[3] = 2.f * (xy + wz);       r.m[4] = 1.f - 2.f * (xx + zz); r.m[5] = 2.f * (yz - wx);
    r.m[6] = 2.f * (xz - wy);       r.m[7] = 2.f * (yz + wx);       r.m[8] = 1.f - 2.f * (xx + yy);
    return r;
}
PHC_HD V3 mat_mul(const M3& a, V3 v) {
    return v3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
              a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
PHC_HD V3 mat_tmul(const M3& a, V3 v) {
    return v3(a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
              a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z);
}
// exp map (rotation vector) -> quaternion, numerically safe near 0 (stepper integrator)
PHC_HD Q4 quat_from_rotvec(V3 e) {
    float a2 = norm2(e);
    float a = sqrtf(a2);
    float k = (a < 1e-4f) ? (0.5f - a2 * (1.0f / 48.0f)) : (sinf(0.5f * a) / a);
    return q4(e.x * k, e.y * k, e.z * k, cosf(0.5f * a));
}
// quaternion -> rotation vector with the shortest arc (|angle| <= pi); used for PD error
PHC_HD V3 quat_to_rotvec(Q4 q) {
    if (q.w < 0.f) q = q4(-q.x, -q.y, -q.z, -q.w);
    float s2 = q.x * q.x + q.y * q.y + q.z * q.z;
    float s = sqrtf(s2);
    float k = (s < 1e-5f) ? 2.0f : (2.0f * atan2f(s, q.w) / s);
    return v3(q.x * k, q.y * k, q.z * k);
}

}  // namespace phc
