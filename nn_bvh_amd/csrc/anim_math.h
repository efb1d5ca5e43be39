// anim_math.h — AnimatedTransform::Interpolate (util/transform.cpp:1062-1081) on the device, for the
// AnimatedPrimitive path of the two-level kernels (cpu/primitive.cpp:133-158).
//
// Produces rows 0..2 of Interpolate(time).mInv, the matrix Transform::ApplyInverse(Ray) reads:
//   Translate(lerp T) * Transform(Slerp(dt, R0, R1)) * Transform(lerp S)
//   (A * B).mInv = B.mInv * A.mInv with SquareMatrix::operator* as FMA chains (util/math.h:1499-1509),
//   Transform(SquareMatrix).mInv = Inverse(m) (util/math.h:1572-1625: DifferenceOfProducts cofactors,
//   compensated InnerProduct sums), Transform(Quaternion) (util/transform.h:367-385).
// Everything is the reference's fp32 arithmetic operation for operation EXCEPT the two per-ray sines
// of Slerp (util/vecmath.h:1146-1151, SinXOverX util/math.h:340-344): the reference calls libm's sinf,
// the device evaluates sin in fp64 and rounds to fp32 (= the correctly rounded value; glibc's sinf
// differs from it on roughly one input in 10^5, by one ulp).  theta = AngleBetween(R0, R1) and
// SinXOverX(theta) depend on the transform only and are computed on the host with libm.  This is the
// one documented tolerance exception of the traversal path.
#pragma once
#include <hip/hip_runtime.h>

#include "nnbvh_internal.h"
#include "trace_math.h"

namespace nnbvh {

// device table entry: kAnimStride = 76 floats (nnbvh_internal.h)
// [0..2] T0, [3..5] T1, [6..9] R0 (v.xyz, w), [10..13] R1, [14..29] S0, [30..45] S1, [46] startTime,
// [47] endTime, [48] theta, [49] SinXOverX(theta), [50..61] startTransform.mInv rows 0..2,
// [62..73] endTransform.mInv rows 0..2, [74] actuallyAnimated (0 / 1), [75] pad

struct Cf {
    float v, err;
};
DEV Cf two_prod(float a, float b) {  // util/math.h:559-562
    const float ab = a * b;
    return {ab, __builtin_fmaf(a, b, -ab)};
}
DEV Cf two_sum(float a, float b) {  // util/math.h:564-567
    const float s = a + b, delta = s - a;
    return {s, (a - (s - delta)) + (b - delta)};
}
// internal::InnerProduct over three / six products (util/math.h:585-612)
DEV float inner3(float a0, float b0, float a1, float b1, float a2, float b2) {
    const Cf p2 = two_prod(a2, b2);
    const Cf p1 = two_prod(a1, b1);
    const Cf s1 = two_sum(p1.v, p2.v);
    const Cf t1 = {s1.v, p1.err + (p2.err + s1.err)};
    const Cf p0 = two_prod(a0, b0);
    const Cf s0 = two_sum(p0.v, t1.v);
    return s0.v + (p0.err + (t1.err + s0.err));
}
DEV float inner6(const float a[6], const float b[6]) {
    Cf t = two_prod(a[5], b[5]);
#pragma unroll
    for (int k = 4; k >= 0; --k) {
        const Cf p = two_prod(a[k], b[k]);
        const Cf s = two_sum(p.v, t.v);
        t = {s.v, p.err + (t.err + s.err)};
    }
    return t.v + t.err;
}

DEV float sin_x_over_x_dev(float x) {  // util/math.h:340-344; sine in fp64, rounded once
    if (1 - x * x == 1) return 1;
    return (float)sin((double)x) / x;
}

// rows 0..2 of Interpolate(time).mInv (r0..r2) and, with WITH_FORWARD, of Interpolate(time).m (f0..f2: what
// Transform::operator()(SurfaceInteraction) applies to points and vectors; AnimatedPrimitive::Intersect,
// cpu/primitive.cpp:146-157); a = the instance's table entry, fwd = rows 0..2 of startTransform.m and of
// endTransform.m (24 floats), read at and beyond the ends of the time range
template <bool WITH_FORWARD>
DEV void anim_rows(const float *__restrict__ a, const float *__restrict__ fwd, float time, float4 &r0, float4 &r1,
                   float4 &r2, float4 &f0, float4 &f1, float4 &f2) {
    const float startTime = a[46], endTime = a[47];
    if (time <= startTime) {  // transform.cpp:1064-1065
        r0 = {a[50], a[51], a[52], a[53]};
        r1 = {a[54], a[55], a[56], a[57]};
        r2 = {a[58], a[59], a[60], a[61]};
        if (WITH_FORWARD) {
            f0 = {fwd[0], fwd[1], fwd[2], fwd[3]};
            f1 = {fwd[4], fwd[5], fwd[6], fwd[7]};
            f2 = {fwd[8], fwd[9], fwd[10], fwd[11]};
        }
        return;
    }
    if (time >= endTime) {  // :1066-1067
        r0 = {a[62], a[63], a[64], a[65]};
        r1 = {a[66], a[67], a[68], a[69]};
        r2 = {a[70], a[71], a[72], a[73]};
        if (WITH_FORWARD) {
            f0 = {fwd[12], fwd[13], fwd[14], fwd[15]};
            f1 = {fwd[16], fwd[17], fwd[18], fwd[19]};
            f2 = {fwd[20], fwd[21], fwd[22], fwd[23]};
        }
        return;
    }
    const float dt = (time - startTime) / (endTime - startTime);
    const float tx = (1 - dt) * a[0] + dt * a[3], ty = (1 - dt) * a[1] + dt * a[4], tz = (1 - dt) * a[2] + dt * a[5];
    // Slerp
    const float theta = a[48], sTT = a[49];
    const float w1 = sin_x_over_x_dev((1 - dt) * theta), w2 = sin_x_over_x_dev(dt * theta);
    float q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = a[6 + k] * (1 - dt) * w1 / sTT + a[10 + k] * dt * w2 / sTT;
    float S[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) S[k] = a[14 + k] * (1 - dt) + a[30 + k] * dt;
    // Transform(Quaternion): mInv (transform.h:367-385)
    const float xx = q[0] * q[0], yy = q[1] * q[1], zz = q[2] * q[2];
    const float xy = q[0] * q[1], xz = q[0] * q[2], yz = q[1] * q[2];
    const float wx = q[0] * q[3], wy = q[1] * q[3], wz = q[2] * q[3];
    const float rmi[16] = {1 - 2 * (yy + zz), 2 * (xy + wz), 2 * (xz - wy), 0,
                           2 * (xy - wz), 1 - 2 * (xx + zz), 2 * (yz + wx), 0,
                           2 * (xz + wy), 2 * (yz - wx), 1 - 2 * (xx + yy), 0,
                           0, 0, 0, 1};
    const float tmi[16] = {1, 0, 0, -tx, 0, 1, 0, -ty, 0, 0, 1, -tz, 0, 0, 0, 1};  // Translate().mInv
    // tri = rmi * tmi (the inverse of Translate * Rotate)
    float tri[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_fmaf(rmi[4 * i + k], tmi[4 * k + j], acc);
            tri[4 * i + j] = acc;
        }
    // smi = Inverse(S), rows 0..2 (util/math.h:1572-1625)
#define SM(i, j) S[4 * (i) + (j)]
    const float s0 = dop(SM(0, 0), SM(1, 1), SM(1, 0), SM(0, 1)), s1 = dop(SM(0, 0), SM(1, 2), SM(1, 0), SM(0, 2));
    const float s2 = dop(SM(0, 0), SM(1, 3), SM(1, 0), SM(0, 3)), s3 = dop(SM(0, 1), SM(1, 2), SM(1, 1), SM(0, 2));
    const float s4 = dop(SM(0, 1), SM(1, 3), SM(1, 1), SM(0, 3)), s5 = dop(SM(0, 2), SM(1, 3), SM(1, 2), SM(0, 3));
    const float c0 = dop(SM(2, 0), SM(3, 1), SM(3, 0), SM(2, 1)), c1 = dop(SM(2, 0), SM(3, 2), SM(3, 0), SM(2, 2));
    const float c2 = dop(SM(2, 0), SM(3, 3), SM(3, 0), SM(2, 3)), c3 = dop(SM(2, 1), SM(3, 2), SM(3, 1), SM(2, 2));
    const float c4 = dop(SM(2, 1), SM(3, 3), SM(3, 1), SM(2, 3)), c5 = dop(SM(2, 2), SM(3, 3), SM(3, 2), SM(2, 3));
    const float da[6] = {s0, -s1, s2, s3, s5, -s4}, db[6] = {c5, c4, c3, c2, c0, c1};
    const float determinant = inner6(da, db);
    float smi[12];
    if (determinant == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) smi[k] = __builtin_nanf("");
    } else {
        const float s = 1 / determinant;
        smi[0] = s * inner3(SM(1, 1), c5, SM(1, 3), c3, -SM(1, 2), c4);
        smi[1] = s * inner3(-SM(0, 1), c5, SM(0, 2), c4, -SM(0, 3), c3);
        smi[2] = s * inner3(SM(3, 1), s5, SM(3, 3), s3, -SM(3, 2), s4);
        smi[3] = s * inner3(-SM(2, 1), s5, SM(2, 2), s4, -SM(2, 3), s3);
        smi[4] = s * inner3(-SM(1, 0), c5, SM(1, 2), c2, -SM(1, 3), c1);
        smi[5] = s * inner3(SM(0, 0), c5, SM(0, 3), c1, -SM(0, 2), c2);
        smi[6] = s * inner3(-SM(3, 0), s5, SM(3, 2), s2, -SM(3, 3), s1);
        smi[7] = s * inner3(SM(2, 0), s5, SM(2, 3), s1, -SM(2, 2), s2);
        smi[8] = s * inner3(SM(1, 0), c4, SM(1, 3), c0, -SM(1, 1), c2);
        smi[9] = s * inner3(-SM(0, 0), c4, SM(0, 1), c2, -SM(0, 3), c0);
        smi[10] = s * inner3(SM(3, 0), s4, SM(3, 3), s0, -SM(3, 1), s2);
        smi[11] = s * inner3(-SM(2, 0), s4, SM(2, 1), s2, -SM(2, 3), s0);
    }
#undef SM
    // mInv = smi * tri, rows 0..2
    float o[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_fmaf(smi[4 * i + k], tri[4 * k + j], acc);
            o[4 * i + j] = acc;
        }
    r0 = {o[0], o[1], o[2], o[3]};
    r1 = {o[4], o[5], o[6], o[7]};
    r2 = {o[8], o[9], o[10], o[11]};
    if (WITH_FORWARD) {
        // m = (Translate.m * Rotate.m) * Scale.m (Transform::operator*, util/transform.cpp:141-143), with
        // Rotate.m = Transpose(Rotate.mInv) (transform.h:382-383) and SquareMatrix::operator* as FMA chains
        const float tm[16] = {1, 0, 0, tx, 0, 1, 0, ty, 0, 0, 1, tz, 0, 0, 0, 1};
        float tr[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = __builtin_fmaf(tm[4 * i + k], rmi[4 * j + k], acc);  // rm[k][j] = rmi[j][k]
                tr[4 * i + j] = acc;
            }
        float fo[12];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = __builtin_fmaf(tr[4 * i + k], S[4 * k + j], acc);
                fo[4 * i + j] = acc;
            }
        f0 = {fo[0], fo[1], fo[2], fo[3]};
        f1 = {fo[4], fo[5], fo[6], fo[7]};
        f2 = {fo[8], fo[9], fo[10], fo[11]};
    }
}

// rows 0..2 of Interpolate(time).mInv, the matrix Transform::ApplyInverse(Ray) reads (the traversal kernels)
DEV void anim_inverse_rows(const float *__restrict__ a, float time, float4 &r0, float4 &r1, float4 &r2) {
    float4 f0, f1, f2;
    anim_rows<false>(a, nullptr, time, r0, r1, r2, f0, f1, f2);
}

}  // namespace nnbvh
